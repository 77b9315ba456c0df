"""ORACLE — test infrastructure only. Never imported by the product path.

CPU restatement (numpy) of the reference's frame pre-processing, /root/reference/utils.py:274-292, for frames that are
already decoded:

    image = ((image - image.min()) / (image.max() - image.min() + 1e-7)).astype(np.float32)   # utils.py:284
    image = cv2.resize(image, (40, 40))                                                       # utils.py:285
    np.transpose(np.array(frames), axes=(0, 3, 1, 2))                                         # utils.py:291

PARITY UNPINNED. OpenCV (`cv2`) is not installed in the build image and the reference holds no fixture for this step, so
`cv2.resize` itself could not be run: the bilinear resize below restates OpenCV's documented algorithm for float32
images (modules/imgproc/src/resize.cpp, INTER_LINEAR: inv_scale = dsize / ssize, scale = 1 / inv_scale;
fx = (float)((dx + 0.5) * scale - 0.5); sx = floor(fx); fx -= sx; coordinates clamped to the border with weight 0;
a horizontal pass over the two source rows followed by the vertical pass; no antialiasing when shrinking). The numpy
line before it is executed as written (uint8 arithmetic for the differences, float64 division, float32 cast). The
MFCC half of the reference's pre-processing (librosa.load's resampler, librosa.feature.mfcc, scipy cubic interp1d,
utils.py:313-349) is not restated: librosa is absent and its resampler has no short published form.
"""
from __future__ import annotations

import numpy as np


def normalise_frame(image_u8: np.ndarray) -> np.ndarray:
    """utils.py:284, verbatim semantics: uint8 differences, float64 division, float32 result"""
    image = np.asarray(image_u8, dtype=np.uint8)
    return ((image - image.min()) / (image.max() - image.min() + 1e-7)).astype(np.float32)


def resize_bilinear_f32(img: np.ndarray, width: int, height: int) -> np.ndarray:
    """cv2.resize(img, (width, height)) for a float32 HWC image, INTER_LINEAR (see the header)"""
    h0, w0 = img.shape[:2]
    scale_x = 1.0 / (width / w0)
    scale_y = 1.0 / (height / h0)

    def coords(n_dst, n_src, scale):
        f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = f - s.astype(np.float32)
        lo = s < 0
        s[lo] = 0; f[lo] = 0.0
        hi = s >= n_src - 1
        s[hi] = n_src - 1; f[hi] = 0.0
        s1 = np.minimum(s + 1, n_src - 1)
        return s, s1, f.astype(np.float32)

    sx, sx1, fx = coords(width, w0, scale_x)
    sy, sy1, fy = coords(height, h0, scale_y)
    a0, a1 = (np.float32(1.0) - fx)[None, :, None], fx[None, :, None]
    b0, b1 = (np.float32(1.0) - fy)[:, None, None], fy[:, None, None]
    img = img.astype(np.float32)
    r0, r1 = img[sy], img[sy1]                                     # (height, w0, C)
    h0_ = r0[:, sx] * a0 + r0[:, sx1] * a1                         # float32 products and sums, in this order
    h1_ = r1[:, sx] * a0 + r1[:, sx1] * a1
    return (h0_ * b0 + h1_ * b1).astype(np.float32)


def frames_to_tensor(frames_u8: np.ndarray, size=(40, 40)) -> np.ndarray:
    """(N, H0, W0, 3) uint8 -> (N, 3, H, W) float32, utils.py:284-291; size = (width, height) as cv2 takes it"""
    out = [resize_bilinear_f32(normalise_frame(f), size[0], size[1]) for f in frames_u8]
    return np.transpose(np.array(out), axes=(0, 3, 1, 2))
