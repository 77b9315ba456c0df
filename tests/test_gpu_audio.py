"""GPU: audio pre-processing (reference utils.py:313-349 from the decoded waveform on).

* the per-row cubic resample (utils.py:337-343) against vectors produced by `scipy.interpolate.interp1d(kind='cubic')` itself
  (tests/golden/resample_*.npz, written by tests/golden/make_golden_audio.py): the device computes the spline as a (B, T)
  matrix product in double and rounds once to float32 — within 1 float32 ulp of float32(scipy's float64 result);
* the MFCC pipeline against oracle/audio_ref.py (float64 restatement of librosa's documented defaults) — PARITY UNPINNED with
  respect to librosa itself (absent from the image); both sides are double arithmetic on the same float32 samples, so the
  tolerance only covers summation order and the final float32 rounding (values reach ~600: ulp 6e-5)."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cvml_goalnet_amd import AVM  # noqa: E402
from cvml_goalnet_amd.preprocess import cubic_resample, extract_audio_features  # noqa: E402
from oracle import audio_ref  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "resample_*.npz"))), ids=lambda p: os.path.basename(p)[:-4])
def test_cubic_resample_matches_scipy_interp1d(path):
    z = np.load(path, allow_pickle=False)
    got = cubic_resample(z["rows"], int(z["b"][0])).cpu().numpy()
    want = z["out"]
    assert got.shape == want.shape and got.dtype == np.float32
    ulp = np.spacing(np.abs(want).astype(np.float32)).astype(np.float64)
    assert (np.abs(got.astype(np.float64) - want) <= ulp).all(), float(np.abs(got - want).max())
    with pytest.raises(ValueError):
        cubic_resample(z["rows"][:, :3], 30)                              # fewer than 4 points: scipy raises too


def _waveform(n, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / audio_ref.SR
    y = 0.3 * np.sin(2 * np.pi * (200.0 + 900.0 * t) * t) + 0.05 * rng.standard_normal(n)      # a chirp over noise
    y[n // 3: n // 3 + 4000] *= 0.001                                     # a near-silent stretch: exercises the top_db clip
    return y.astype(np.float32)


@pytest.mark.parametrize("seconds,n_frames,b", [(3.0, 3, 30), (5.37, 5, 30), (4.0, 2, 60)])
def test_mfcc_features_match_the_float64_restatement(seconds, n_frames, b):
    y = _waveform(int(seconds * audio_ref.SR), int(seconds * 100))
    got = extract_audio_features(y, n_frames, b).cpu().numpy()
    want = audio_ref.extract_audio_features(y, n_frames, b)
    assert got.shape == (n_frames, 30, b) and got.dtype == np.float32
    err = np.abs(got.astype(np.float64) - want)
    print(f"[parity] mfcc {seconds}s / {n_frames} slots: max |err| {err.max():.3e} (max |mfcc| {np.abs(want).max():.1f})")
    assert err.max() <= 2e-4
    assert np.abs(want).max() > 50.0                                       # coefficient 0 of a real MFCC: the comparison is not vacuous


def test_slot_edges_and_errors():
    # round-half-to-even slot boundaries (utils.py:325-326) and a clipped, shorter last slot with its own spline matrix
    y = _waveform(3 * 22050 + 1000, 7)
    got = extract_audio_features(y, 3, 30).cpu().numpy()
    want = audio_ref.extract_audio_features(y, 3, 30)
    assert np.abs(got - want).max() <= 2e-4
    with pytest.raises(ValueError):
        extract_audio_features(y[:4000], 3, 30)                            # 1333 samples per slot: 3 STFT frames < 4
    # feeds AudBl: (N, 30, B) float32 on the device
    n = 4
    feats = extract_audio_features(_waveform(n * 22050, 9), n, 30)
    m = AVM(audio_included=True, device="cuda:0")
    with torch.no_grad():
        out = m(feats, torch.rand(n, 3, 40, 40))
    assert out.shape == (n, 1) and bool(torch.isfinite(out).all())
