"""ORACLE — test infrastructure only. Never imported by the product path.

CPU restatement (numpy, float64) of the reference's audio pre-processing, /root/reference/utils.py:313-349, for a
waveform that is already decoded and resampled (`y, sr = librosa.load(audio_fp)`, utils.py:320, stays the reference's
I/O: mono, 22 050 Hz, float32):

    audio_samples_per_frame = len(y) / n_frames                                          utils.py:322
    start = round(frame_idx * audio_samples_per_frame); end = round(start + ...)         utils.py:325-330 (Python round: half-to-even)
    mfccs = librosa.feature.mfcc(y = y[start:end], sr = sr, n_mfcc = 30)                 utils.py:333
    interp1d(np.arange(T), mfccs[f], kind = 'cubic', fill_value = "extrapolate")         utils.py:337-342
        (np.linspace(0, T - 1, B))                                                        utils.py:343

Two parts with different pinning:

* `cubic_resample_matrix` / `cubic_resample` — the not-a-knot cubic spline `scipy.interpolate.interp1d(kind='cubic')`
  builds (make_interp_spline, k = 3, default boundary conditions), written as a (B, T) matrix in float64. **Pinned**: scipy
  IS importable in the build image; tests/golden/make_golden_audio.py runs `interp1d` itself and commits the vectors
  (tests/golden/resample_*.npz); tests/test_oracle_golden.py checks this restatement against them.
* `mfcc` — librosa 0.10's documented defaults for `librosa.feature.mfcc` -> `melspectrogram` -> `stft`:
  n_fft 2048, hop 512, periodic Hann window, center = True with zero padding (pad_mode = "constant"), power 2.0,
  128 Slaney mel bands (htk = False, norm = "slaney", fmin 0, fmax sr/2), `power_to_db` (ref 1.0, amin 1e-10, top_db 80),
  orthonormal DCT-II over the mel axis, first 30 coefficients. **PARITY UNPINNED**: librosa is not installed here and
  the reference holds no fixture for this step, so this restatement has never been compared with librosa's output
  (which also runs in float32; this one in float64).
"""
from __future__ import annotations

import numpy as np

SR = 22050
N_FFT = 2048
HOP = 512
N_MELS = 128
N_MFCC = 30


# ------------------------------------------------------------------------------------------------------------------
# cubic spline (pinned against scipy)
# ------------------------------------------------------------------------------------------------------------------
def cubic_resample_matrix(t: int, b: int) -> np.ndarray:
    """R (b, t) float64 with R @ y == interp1d(arange(t), y, kind='cubic')(linspace(0, t - 1, b)) for any y (t,).

    Uniform knots 0..t-1, second derivatives M from  M[i-1] + 4 M[i] + M[i+1] = 6 (y[i-1] - 2 y[i] + y[i+1])  (i = 1..t-2)
    with the not-a-knot conditions M[0] - 2 M[1] + M[2] = 0 and M[t-3] - 2 M[t-2] + M[t-1] = 0 (third derivative
    continuous across the first and last interior knots); on [i, i+1] with u = x - i:
        s(x) = (1-u) y[i] + u y[i+1] + ((1-u)^3 - (1-u)) M[i] / 6 + (u^3 - u) M[i+1] / 6."""
    if t < 4:
        raise ValueError("a cubic spline needs at least 4 points (scipy's interp1d raises as well)")
    a = np.zeros((t, t))
    d = np.zeros((t, t))
    for i in range(1, t - 1):
        a[i, i - 1], a[i, i], a[i, i + 1] = 1.0, 4.0, 1.0
        d[i, i - 1], d[i, i], d[i, i + 1] = 6.0, -12.0, 6.0
    a[0, 0], a[0, 1], a[0, 2] = 1.0, -2.0, 1.0
    a[t - 1, t - 3], a[t - 1, t - 2], a[t - 1, t - 1] = 1.0, -2.0, 1.0
    s = np.linalg.solve(a, d)                                 # M = s @ y
    x = np.linspace(0.0, t - 1.0, b)
    i = np.minimum(np.floor(x).astype(np.int64), t - 2)
    u = x - i
    r = np.zeros((b, t))
    rows = np.arange(b)
    r[rows, i] += 1.0 - u
    r[rows, i + 1] += u
    r += (((1.0 - u) ** 3 - (1.0 - u)) / 6.0)[:, None] * s[i] + ((u ** 3 - u) / 6.0)[:, None] * s[i + 1]
    return r


def cubic_resample(rows: np.ndarray, b: int) -> np.ndarray:
    """(..., T) -> (..., b) float64: every row through utils.py:337-343"""
    rows = np.asarray(rows, dtype=np.float64)
    return rows @ cubic_resample_matrix(rows.shape[-1], b).T


# ------------------------------------------------------------------------------------------------------------------
# MFCC (librosa's documented defaults; parity unpinned)
# ------------------------------------------------------------------------------------------------------------------
def hann_periodic(n: int) -> np.ndarray:
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)          # scipy.signal.get_window("hann", n, fftbins=True)


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr: int = SR, n_fft: int = N_FFT, n_mels: int = N_MELS) -> np.ndarray:
    """librosa.filters.mel(sr, n_fft, n_mels, fmin=0, fmax=sr/2, htk=False, norm='slaney'): (n_mels, 1 + n_fft/2) float64"""
    fftfreqs = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(sr / 2.0), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0.0, np.minimum(lower, upper))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]       # Slaney: constant energy per band
    return w


def dct_matrix(n_out: int = N_MFCC, n_in: int = N_MELS) -> np.ndarray:
    """scipy.fftpack.dct(x, type=2, norm='ortho') along an axis of length n_in, first n_out outputs: (n_out, n_in)"""
    k = np.arange(n_out)[:, None]
    n = np.arange(n_in)[None, :]
    m = np.cos(np.pi * k * (2 * n + 1) / (2.0 * n_in)) * np.sqrt(2.0 / n_in)
    m[0] *= np.sqrt(0.5)
    return m


def n_stft_frames(n_samples: int) -> int:
    return 1 + n_samples // HOP                                           # center=True


def log_mel(seg: np.ndarray, sr: int = SR) -> np.ndarray:
    """power_to_db(melspectrogram(seg)) BEFORE the top_db clip: (n_mels, T) float64"""
    seg = np.asarray(seg, dtype=np.float64)
    t = n_stft_frames(len(seg))
    pad = np.concatenate([np.zeros(N_FFT // 2), seg, np.zeros(N_FFT // 2)])
    win = hann_periodic(N_FFT)
    frames = np.stack([pad[i * HOP:i * HOP + N_FFT] * win for i in range(t)])          # (T, n_fft)
    power = np.abs(np.fft.rfft(frames, axis=1)) ** 2                                    # (T, 1025)
    mel = mel_filterbank(sr) @ power.T                                                   # (n_mels, T)
    return 10.0 * np.log10(np.maximum(1e-10, mel))                                       # ref = 1.0 -> - 10 log10(1) = 0


def mfcc(seg: np.ndarray, sr: int = SR, n_mfcc: int = N_MFCC) -> np.ndarray:
    """librosa.feature.mfcc(y=seg, sr=sr, n_mfcc=n_mfcc) with librosa 0.10 defaults: (n_mfcc, T) float64"""
    s_db = log_mel(seg, sr)
    s_db = np.maximum(s_db, s_db.max() - 80.0)                                          # top_db = 80
    return dct_matrix(n_mfcc, N_MELS) @ s_db


def slot_bounds(n_samples: int, n_frames: int):
    """utils.py:322-330: [(start, end)] per frame slot; Python's round (half to even) on floats"""
    spf = n_samples / n_frames
    out = []
    for i in range(n_frames):
        start = round(i * spf)
        end = round(start + spf)
        out.append((start, min(end, n_samples)))
    return out


def extract_audio_features(y: np.ndarray, n_frames: int, bin_length: int, sr: int = SR) -> np.ndarray:
    """utils.py:313-349 from the decoded waveform on: (n_frames, 30, bin_length) float64"""
    out = []
    for a, b in slot_bounds(len(y), n_frames):
        out.append(cubic_resample(mfcc(y[a:b], sr), bin_length))
    return np.array(out)
