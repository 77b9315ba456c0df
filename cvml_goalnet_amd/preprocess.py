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
