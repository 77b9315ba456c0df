"""Deterministic synthetic inputs / weights / dropout masks (counter-based splitmix64).

SURVEY.md §8(d): the reference ships no dataset and never seeds torch (only numpy, with
12344321 at /root/reference/main.py:53), so every parity and bench input is regenerated from a
seed formula on both sides instead of being committed (weights are 94 MB at 40x40, 5 GB at 224x224).

The same formula is implemented on the device by `goalnet_fill_uniform` (csrc/fill.hip); the GPU
tests check the two agree bit for bit.

    key(tensor_id)   = mix64(seed + (tensor_id + 1) * 0xD1342543DE82EF95)
    bits(i)          = mix64(key + (i + 1) * 0x9E3779B97F4A7C15)
    u(i)             = float32(bits(i) >> 40) * 2^-24            in [0, 1), exact in fp32
    uniform(lo, hi)  = lo + (hi - lo) * u       (fp32 multiply, then fp32 add: two roundings)

Element index i is always the flat index of the torch-native (logical, contiguous) layout of the
tensor (OIHW weights, NCHW frames), so fixtures do not depend on the device layout.
"""
from __future__ import annotations

import math

import numpy as np

BASE_SEED = 12344321  # /root/reference/main.py:53
_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_STREAM = np.uint64(0xD1342543DE82EF95)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_CHUNK = 1 << 24


def _mix64(z: np.ndarray) -> np.ndarray:
    z = z.astype(np.uint64, copy=True)
    z ^= z >> np.uint64(30)
    z *= _M1
    z ^= z >> np.uint64(27)
    z *= _M2
    z ^= z >> np.uint64(31)
    return z


def stream_key(tensor_id: int, seed: int = BASE_SEED) -> np.uint64:
    with np.errstate(over="ignore"):
        z = np.array([np.uint64(seed) + np.uint64(tensor_id + 1) * _STREAM], dtype=np.uint64)
    return _mix64(z)[0]


def raw_bits(tensor_id: int, n: int, seed: int = BASE_SEED, offset: int = 0) -> np.ndarray:
    """uint64 stream bits(offset) .. bits(offset+n-1)."""
    key = stream_key(tensor_id, seed)
    out = np.empty(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        for s in range(0, n, _CHUNK):
            e = min(n, s + _CHUNK)
            idx = np.arange(offset + s + 1, offset + e + 1, dtype=np.uint64)
            out[s:e] = _mix64(key + idx * _GAMMA)
    return out


def unit(tensor_id: int, n: int, seed: int = BASE_SEED, offset: int = 0) -> np.ndarray:
    """float32 u(i) in [0,1), 24 random bits each."""
    out = np.empty(n, dtype=np.float32)
    for s in range(0, n, _CHUNK):
        e = min(n, s + _CHUNK)
        b = raw_bits(tensor_id, e - s, seed, offset + s)
        out[s:e] = (b >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)
    return out


def uniform(tensor_id: int, shape, lo: float, hi: float, seed: int = BASE_SEED) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    u = unit(tensor_id, n, seed)
    lo32 = np.float32(lo)
    span = np.float32(np.float32(hi) - lo32)
    return (lo32 + span * u).astype(np.float32).reshape(shape)


# ----------------------------------------------------------------------------------------------
# tensor ids (fixed; the goldens depend on them)
# ----------------------------------------------------------------------------------------------
TID_VISUAL = 1
TID_AUDIO = 2  # uses 2..5 (four uniform streams summed)
TID_LABELS = 6
TID_DROP = 4096  # + 8*step + layer: visbl.drop5, fusion.2, fusion.5, fusion.8, fusion.11
TID_PARAM = 64  # 64 + index into PARAM_ORDER
TID_GRADOUT = 40

# torch-native state_dict parameter order (probed from the reference instance)
PARAM_ORDER = [
    "visbl.conv1.weight", "visbl.conv1.bias", "visbl.bnorm1.weight", "visbl.bnorm1.bias",
    "visbl.conv2.weight", "visbl.conv2.bias", "visbl.bnorm2.weight", "visbl.bnorm2.bias",
    "visbl.conv3.weight", "visbl.conv3.bias", "visbl.bnorm3.weight", "visbl.bnorm3.bias",
    "visbl.linear5.weight", "visbl.linear5.bias",
    "audbl.conv1.weight", "audbl.conv1.bias", "audbl.conv2.weight", "audbl.conv2.bias",
    "audbl.linear3.weight", "audbl.linear3.bias",
    "fusion.0.weight", "fusion.0.bias", "fusion.3.weight", "fusion.3.bias",
    "fusion.6.weight", "fusion.6.bias", "fusion.9.weight", "fusion.9.bias",
    "fusion.12.weight", "fusion.12.bias",
]

DROP_P = 0.2  # /root/reference/utils.py:170, 245-254


def conv_out_hw(h: int, w: int):
    """Spatial sizes through VisBl (/root/reference/utils.py:151-164)."""
    h1, w1 = (h + 2 * 3 - 3) // 3 + 1, (w + 2 * 3 - 3) // 3 + 1  # conv1 k3 s3 p3
    p1 = (h1 - 2, w1 - 2)  # maxpool k3 s1
    p2 = (p1[0] - 2, p1[1] - 2)  # conv2 keeps size, pool shrinks by 2
    p3 = (p2[0] - 2, p2[1] - 2)
    return (h1, w1), p1, p2, p3


def param_shapes(h: int, w: int, bins: int, audio_included: bool = True) -> dict:
    """Logical (torch-native) parameter shapes for an H x W frame and B audio bins."""
    _, _, _, p3 = conv_out_hw(h, w)
    l1 = (bins + 2 - 3) // 2 + 1
    l2 = (l1 + 2 - 3) // 2 + 1
    shapes = {
        "visbl.conv1.weight": (64, 3, 3, 3), "visbl.conv1.bias": (64,),
        "visbl.bnorm1.weight": (64,), "visbl.bnorm1.bias": (64,),
        "visbl.conv2.weight": (256, 64, 3, 3), "visbl.conv2.bias": (256,),
        "visbl.bnorm2.weight": (256,), "visbl.bnorm2.bias": (256,),
        "visbl.conv3.weight": (512, 256, 3, 3), "visbl.conv3.bias": (512,),
        "visbl.bnorm3.weight": (512,), "visbl.bnorm3.bias": (512,),
        "visbl.linear5.weight": (512, 512 * p3[0] * p3[1]), "visbl.linear5.bias": (512,),
    }
    if audio_included:
        shapes.update({
            "audbl.conv1.weight": (64, 30, 3), "audbl.conv1.bias": (64,),
            "audbl.conv2.weight": (128, 64, 3), "audbl.conv2.bias": (128,),
            "audbl.linear3.weight": (128, 128 * l2), "audbl.linear3.bias": (128,),
        })
    f0_in = 640 if audio_included else 512
    shapes.update({
        "fusion.0.weight": (512, f0_in), "fusion.0.bias": (512,),
        "fusion.3.weight": (512, 512), "fusion.3.bias": (512,),
        "fusion.6.weight": (256, 512), "fusion.6.bias": (256,),
        "fusion.9.weight": (128, 256), "fusion.9.bias": (128,),
        "fusion.12.weight": (1, 128), "fusion.12.bias": (1,),
    })
    return shapes


def _fan_in(name: str, shape) -> int:
    if name.endswith(".bias"):
        raise ValueError
    return int(np.prod(shape[1:]))


def param_range(name: str, shapes: dict, bn_affine: str = "random"):
    """(lo, hi) of the uniform a parameter is drawn from.

    Weights and biases follow torch's default init bound 1/sqrt(fan_in) (kaiming_uniform a=sqrt(5),
    SURVEY.md §8(a) row 1). BatchNorm affine is (1, 0) in the reference; `bn_affine="random"` draws
    gamma in [0.5, 1.5) and beta in [-0.5, 0.5) so the parity tests exercise them.
    """
    if ".bnorm" in name:
        if bn_affine == "random":
            return (0.5, 1.5) if name.endswith("weight") else (-0.5, 0.5)
        return (1.0, 1.0) if name.endswith("weight") else (0.0, 0.0)
    wname = name.rsplit(".", 1)[0] + ".weight"
    bound = 1.0 / math.sqrt(_fan_in(wname, shapes[wname]))
    return (-bound, bound)


def make_params(h: int, w: int, bins: int = 30, audio_included: bool = True,
                bn_affine: str = "random", seed: int = BASE_SEED, only=None) -> dict:
    """name -> float32 ndarray in torch-native layout."""
    shapes = param_shapes(h, w, bins, audio_included)
    out = {}
    for name, shape in shapes.items():
        if only is not None and name not in only:
            continue
        lo, hi = param_range(name, shapes, bn_affine)
        tid = TID_PARAM + PARAM_ORDER.index(name)
        if lo == hi:
            out[name] = np.full(shape, lo, dtype=np.float32)
        else:
            out[name] = uniform(tid, shape, lo, hi, seed)
    return out


def make_visual(n: int, h: int, w: int, seed: int = BASE_SEED) -> np.ndarray:
    """(N,3,H,W) fp32: U[0,1) then per-frame min-max rescaled to exactly [0,1]
    (mimics /root/reference/utils.py:284 without its 1e-7 guard)."""
    x = uniform(TID_VISUAL, (n, 3 * h * w), 0.0, 1.0, seed)
    mn = x.min(axis=1, keepdims=True)
    mx = x.max(axis=1, keepdims=True)
    x = ((x - mn) / (mx - mn)).astype(np.float32)
    return x.reshape(n, 3, h, w)


def make_audio(n: int, bins: int = 30, seed: int = BASE_SEED) -> np.ndarray:
    """(N,30,B) fp32 MFCC-like: ~N(0,1)*20 (Irwin-Hall of 4 uniforms, arithmetic only), coefficient 0
    offset by -200."""
    cnt = n * 30 * bins
    s = np.zeros(cnt, dtype=np.float32)
    for k in range(4):
        s += unit(TID_AUDIO + k, cnt, seed)
    g = (s - np.float32(2.0)) * np.float32(math.sqrt(3.0)) * np.float32(20.0)
    g = g.reshape(n, 30, bins).astype(np.float32)
    g[:, 0, :] -= np.float32(200.0)
    return g


def make_labels(n: int, seed: int = BASE_SEED) -> np.ndarray:
    """(N,) fp32 drawn uniformly from {1..5} (/root/reference/utils.py:394 rounds annotator means)."""
    u = unit(TID_LABELS, n, seed)
    return (np.float32(1.0) + np.floor(u * np.float32(5.0))).astype(np.float32)


DROP_WIDTHS = (512, 512, 512, 256, 128)  # visbl.drop5, fusion.2, .5, .8, .11


def make_drop_masks(n: int, seed: int = BASE_SEED, p: float = DROP_P, step: int = 0):
    """Five (N,width) fp32 multipliers, 0 or 1/(1-p): keep iff u >= p.

    Same formula as the device generator `goalnet_dropout_mask` (csrc/fill.hip). `step` selects an
    independent stream per train step (tensor id TID_DROP + 8*step + layer)."""
    out = []
    scale = np.float32(1.0 / (1.0 - p))
    for li, wdt in enumerate(DROP_WIDTHS):
        u = unit(TID_DROP + 8 * step + li, n * wdt, seed)
        out.append(np.where(u >= np.float32(p), scale, np.float32(0.0)).astype(np.float32).reshape(n, wdt))
    return out


def sample_indices(numel: int, k: int = 16, salt: int = 0) -> np.ndarray:
    """k deterministic flat indices into a tensor of `numel` elements (golden sampling)."""
    b = raw_bits(1000 + salt, k, BASE_SEED)
    return (b % np.uint64(max(numel, 1))).astype(np.int64)
