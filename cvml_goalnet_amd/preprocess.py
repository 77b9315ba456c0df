"""Frame pre-processing on the GPU: what `extract_condensed_frame_tensor`
(`/root/reference/utils.py:274-292`) does to every decoded frame it keeps — per-frame min-max normalisation,
`cv2.resize(image, (40, 40))`, channel-first — for frames that are already decoded (`cv2.VideoCapture` stays the
reference's I/O). Output is the `(N, 3, H, W)` float32 tensor `AVM` takes, left on the device.

PARITY UNPINNED: OpenCV is not in the build image; kernel and oracle (oracle/preproc_ref.py) restate its documented
INTER_LINEAR algorithm and agree with each other bit for bit, but neither has been compared with cv2 itself.
The MFCC half of the reference's pre-processing (utils.py:313-349) is not built (librosa absent).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from ._lib import GoalnetError, check
from .ops import _s


def frames_to_tensor(frames, size=(40, 40), device=None) -> torch.Tensor:
    """frames: (N, H0, W0, 3) uint8 (numpy or torch, BGR as cv2 decodes); size = (width, height) as `cv2.resize` takes it.
    Returns the (N, 3, height, width) float32 GPU tensor of utils.py:291."""
    if not torch.cuda.is_available():
        raise GoalnetError("frame pre-processing runs on the GPU (torch.cuda.is_available() is False); there is no CPU fallback")
    dev = torch.device(device if device is not None else "cuda:0")
    t = frames if torch.is_tensor(frames) else torch.from_numpy(np.ascontiguousarray(frames))
    if t.dtype != torch.uint8 or t.dim() != 4 or t.shape[3] != 3:
        raise ValueError("frames must be uint8 (N, H0, W0, 3)")
    t = t.to(dev).contiguous()
    n, h0, w0, _ = t.shape
    if n < 1:
        raise ValueError("no frames")
    w, h = int(size[0]), int(size[1])
    out = torch.empty(n, 3, h, w, dtype=torch.float32, device=dev)
    scratch = torch.empty(n, 2, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(_lib.load().goalnet_frames_preprocess(t.data_ptr(), n, h0, w0, out.data_ptr(), h, w, scratch.data_ptr(), _s()),
              "frames_preprocess")
    return out


# ---------------------------------------------------------------------------------------------------------------------
# audio: per-slot MFCC + cubic resample (utils.py:313-349 from the decoded waveform on)
# ---------------------------------------------------------------------------------------------------------------------
SR, N_FFT, HOP, N_MELS, N_MFCC, TOP_DB = 22050, 2048, 512, 128, 30, 80.0
_CONST = {}        # (device, sr) -> device constants;  (device, T, B) -> resample matrix


def spline_matrix(t: int, b: int) -> np.ndarray:
    """(b, t) float64 matrix of `interp1d(arange(t), ., kind='cubic')(linspace(0, t-1, b))` (utils.py:337-343): the cubic
    spline through t uniform knots with not-a-knot ends, in the second-derivative form. A host-side constant (t <= ~100)."""
    if t < 4:
        raise ValueError(f"a cubic spline needs at least 4 points, got {t} (scipy's interp1d raises as well)")
    lhs = np.diag(np.full(t, 4.0)) + np.diag(np.ones(t - 1), 1) + np.diag(np.ones(t - 1), -1)
    rhs = 6.0 * (np.diag(np.full(t, -2.0)) + np.diag(np.ones(t - 1), 1) + np.diag(np.ones(t - 1), -1))
    lhs[0], lhs[-1], rhs[0], rhs[-1] = 0.0, 0.0, 0.0, 0.0
    lhs[0, :3] = (1.0, -2.0, 1.0)                      # third derivative continuous at knot 1 ...
    lhs[-1, -3:] = (1.0, -2.0, 1.0)                    # ... and at knot t-2
    second = np.linalg.solve(lhs, rhs)                 # second derivatives at the knots as a linear map of the values
    x = np.linspace(0.0, t - 1.0, b)
    k = np.clip(np.floor(x).astype(np.int64), 0, t - 2)
    u = (x - k)[:, None]
    eye = np.eye(t)
    return (1.0 - u) * eye[k] + u * eye[k + 1] + ((1.0 - u) ** 3 - (1.0 - u)) / 6.0 * second[k] + (u ** 3 - u) / 6.0 * second[k + 1]


def _slaney_mel_weights(sr: int):
    """librosa.filters.mel(sr=sr, n_fft=2048, n_mels=128) (htk=False, norm='slaney', fmin 0, fmax sr/2) as sparse triangles:
    per band the first FFT bin, the number of bins and the weights"""
    def to_hz(m):
        lin = m * (200.0 / 3)
        return np.where(m >= 15.0, 1000.0 * np.exp((m - 15.0) * (np.log(6.4) / 27.0)), lin)     # 1 kHz = mel 15: log above
    top = 15.0 + np.log((sr / 2.0) / 1000.0) / (np.log(6.4) / 27.0) if sr / 2.0 >= 1000.0 else (sr / 2.0) / (200.0 / 3)
    edges = to_hz(np.linspace(0.0, top, N_MELS + 2))
    freqs = np.linspace(0.0, sr / 2.0, N_FFT // 2 + 1)
    starts, counts, offs, weights = [], [], [], []
    for m in range(N_MELS):
        lo, ce, hi = edges[m], edges[m + 1], edges[m + 2]
        tri = np.maximum(0.0, np.minimum((freqs - lo) / (ce - lo), (hi - freqs) / (hi - ce))) * (2.0 / (hi - lo))
        nz = np.nonzero(tri)[0]
        k0, k1 = (int(nz[0]), int(nz[-1]) + 1) if nz.size else (0, 0)
        starts.append(k0); counts.append(k1 - k0); offs.append(sum(counts[:-1])); weights.append(tri[k0:k1])
    return (np.array(starts, np.int32), np.array(counts, np.int32), np.array(offs, np.int32),
            np.concatenate(weights) if weights else np.zeros(0))


def _audio_consts(dev, sr):
    key = (str(dev), sr)
    if key not in _CONST:
        j = np.arange(N_FFT)
        window = 0.5 - 0.5 * np.cos(2.0 * np.pi * j / N_FFT)                           # periodic Hann (fftbins=True)
        k = np.arange(N_FFT // 2)
        tw = np.stack([np.cos(2.0 * np.pi * k / N_FFT), -np.sin(2.0 * np.pi * k / N_FFT)], axis=1)
        ms, mc, mo, mw = _slaney_mel_weights(sr)
        f = np.arange(N_MFCC)[:, None]
        n = np.arange(N_MELS)[None, :]
        dct = np.cos(np.pi * f * (2 * n + 1) / (2.0 * N_MELS)) * np.sqrt(2.0 / N_MELS)   # DCT-II, norm='ortho'
        dct[0] *= np.sqrt(0.5)
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)              # noqa: E731
        _CONST[key] = {"window": up(window), "twiddle": up(tw), "mel_start": up(ms), "mel_count": up(mc), "mel_off": up(mo),
                       "mel_w": up(mw), "dct": up(dct)}
    return _CONST[key]


def cubic_resample(rows, b: int, device=None) -> torch.Tensor:
    """utils.py:337-343 for every row of `rows` (..., T) -> (..., b) float32 on the GPU (cubic spline, not-a-knot ends)."""
    if not torch.cuda.is_available():
        raise GoalnetError("audio pre-processing runs on the GPU (torch.cuda.is_available() is False); there is no CPU fallback")
    dev = torch.device(device if device is not None else "cuda:0")
    x = (rows if torch.is_tensor(rows) else torch.from_numpy(np.ascontiguousarray(rows))).to(device=dev, dtype=torch.float32).contiguous()
    t = x.shape[-1]
    r = torch.from_numpy(spline_matrix(t, b)).to(dev)
    out = torch.empty(*x.shape[:-1], b, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(_lib.load().goalnet_cubic_resample(x.data_ptr(), r.data_ptr(), out.data_ptr(), x.numel() // t, t, b, _s()), "cubic_resample")
    return out


def slot_bounds(n_samples: int, n_frames: int):
    """utils.py:322-330: (start, length) of every frame slot; Python's round() on floats (half to even), end clipped"""
    spf = n_samples / n_frames
    starts, lens = [], []
    for i in range(n_frames):
        a = round(i * spf)
        e = min(round(a + spf), n_samples)
        starts.append(a); lens.append(max(e - a, 0))
    return starts, lens


def extract_audio_features(y, n_frames: int, bin_length: int, sr: int = SR, device=None) -> torch.Tensor:
    """`utils.extract_audio_features(audio_fp, n_frames, bin_length)` (utils.py:313-349) from the decoded waveform on:
    `y` is what `librosa.load(audio_fp)` returns (mono float32 at `sr` = 22 050 Hz; decoding and resampling the file stay
    the reference's I/O). Returns the (n_frames, 30, bin_length) float32 tensor AudBl takes, on the GPU.
    PARITY UNPINNED for the MFCC values (librosa is not installed in the build image); the resample is pinned against scipy."""
    if not torch.cuda.is_available():
        raise GoalnetError("audio pre-processing runs on the GPU (torch.cuda.is_available() is False); there is no CPU fallback")
    dev = torch.device(device if device is not None else "cuda:0")
    wav = (y if torch.is_tensor(y) else torch.from_numpy(np.ascontiguousarray(y))).to(device=dev, dtype=torch.float32).contiguous().view(-1)
    if n_frames < 1 or wav.numel() < 1:
        raise ValueError("no audio / no frames")
    starts, lens = slot_bounds(wav.numel(), n_frames)
    ts = [1 + n // HOP for n in lens]
    if min(ts) < 4:
        raise ValueError(f"a slot has {min(ts)} STFT frames: a cubic spline needs at least 4 (scipy's interp1d raises as well)")
    tmax = max(ts)
    c = _audio_consts(dev, sr)
    mats, offs, off_of = [], [], {}
    for t in ts:                                       # one (B, T) matrix per distinct T (the last slot may be shorter)
        if t not in off_of:
            off_of[t] = sum(m.size for m in mats)
            mats.append(spline_matrix(t, bin_length))
        offs.append(off_of[t])
    r_all = torch.from_numpy(np.concatenate([m.reshape(-1) for m in mats])).to(dev)
    r_off = torch.tensor(offs, dtype=torch.int64, device=dev)
    start_t = torch.tensor(starts, dtype=torch.int64, device=dev)
    len_t = torch.tensor(lens, dtype=torch.int32, device=dev)
    logmel = torch.empty(n_frames, tmax, N_MELS, dtype=torch.float64, device=dev)
    out = torch.empty(n_frames, N_MFCC, bin_length, dtype=torch.float32, device=dev)
    lib = _lib.load()
    with torch.cuda.device(dev):
        for s0 in range(0, n_frames, 65535):           # grid.y limit
            s1 = min(n_frames, s0 + 65535)
            check(lib.goalnet_logmel_slots(wav.data_ptr(), start_t[s0:].data_ptr(), len_t[s0:].data_ptr(), s1 - s0, tmax,
                                           c["window"].data_ptr(), c["twiddle"].data_ptr(), c["mel_start"].data_ptr(),
                                           c["mel_count"].data_ptr(), c["mel_off"].data_ptr(), c["mel_w"].data_ptr(),
                                           logmel[s0:].data_ptr(), _s()), "logmel_slots")
        check(lib.goalnet_mfcc_from_logmel(logmel.data_ptr(), len_t.data_ptr(), n_frames, tmax, c["dct"].data_ptr(), r_all.data_ptr(),
                                           r_off.data_ptr(), out.data_ptr(), N_MFCC, bin_length, TOP_DB, _s()), "mfcc_from_logmel")
    return out
