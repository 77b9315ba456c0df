"""Post-processing and F-score on the GPU behind the reference's function names.

The reference runs, after every video of every epoch (`main.py:100, 115, 207, 227`),

    postprocess_and_get_fscores(video_id, batch_predictions, full_n_batch_frames, gd_summarized_video_frame_indices,
                                h5_file_path, mat_file_path, skip_frames)            # utils.py:586-604

which re-opens two HDF5 files to fetch the video's KTS change points (`utils.py:617-629`) and then works in pure
Python lists: round -> int8, `expand_array`, `get_clip_information`, a list-of-lists 0/1 `knapsack`, a frame-by-frame mask
loop and `get_fscore`. Here the same functions take the change points as an argument (reading the dataset's HDF5 files
stays the reference's job — h5py is I/O, not the hot path) and run as a handful of kernels in libgoalnet_hip.so
(csrc/postproc.hip), bit-exact with the reference's integers and doubles. `SummaryEvaluator` keeps a video's change
points and annotator summaries resident on the device, so the per-epoch call is: predictions stay on the GPU, five
small launches, 24 bytes back.

No CPU fallback: without the library / a GPU these functions raise.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import GoalnetError, check
from .ops import _s

I32, I64, U8, F32, F64 = torch.int32, torch.int64, torch.uint8, torch.float32, torch.float64


def _dev(device=None):
    if not torch.cuda.is_available():
        raise GoalnetError("post-processing runs on the GPU (torch.cuda.is_available() is False); there is no CPU fallback")
    return torch.device(device if device is not None else "cuda:0")


def _importances_1d(batch_importances) -> torch.Tensor:
    """utils.py:608-610"""
    t = batch_importances if torch.is_tensor(batch_importances) else torch.as_tensor(np.asarray(batch_importances))
    if t.dim() != 1:
        assert t.dim() == 2 and t.shape[-1] == 1, "E: Invalid shape for importance tensor"
        t = t[:, 0]
    return t


def knapsack(values: Sequence[int], weights: Sequence[float], capacity, scale_factor=5, device=None) -> List[int]:
    """utils.py:465-510. Scaling of weights and capacity (`int(w * scale_factor)`) is the reference's host arithmetic;
    the DP table and the back-tracking run on the device."""
    dev = _dev(device)
    lib = _lib.load()
    w = [int(x * scale_factor) for x in weights]
    cap = int(capacity * scale_factor)
    n = len(values)
    if n == 0:
        return []
    if min(w) < 0 or cap < 0:
        raise ValueError("knapsack: negative weight or capacity")
    vals = torch.tensor([int(v) for v in values], dtype=I64, device=dev)
    wts = torch.tensor(w, dtype=I32, device=dev)
    sel = torch.empty(n, dtype=I32, device=dev)
    nbytes = lib.goalnet_knapsack_ws_bytes(n, cap)
    ws = torch.empty(nbytes // 8, dtype=I64, device=dev)
    with torch.cuda.device(dev):
        check(lib.goalnet_knapsack(vals.data_ptr(), wts.data_ptr(), n, cap, sel.data_ptr(), ws.data_ptr(), nbytes, _s()), "knapsack")
    return torch.nonzero(sel).flatten().tolist()


def get_fscore(gd_summary_indices, predicted_summary_indices, device=None) -> Tuple[float, float]:
    """utils.py:552-580 on 0/1 arrays: (mean, max) of the per-annotator F-scores, exact integer sums."""
    dev = _dev(device)
    lib = _lib.load()
    gd = torch.as_tensor(np.ascontiguousarray(np.asarray(gd_summary_indices) != 0).astype(np.uint8)).to(dev) \
        if not torch.is_tensor(gd_summary_indices) else (gd_summary_indices != 0).to(device=dev, dtype=U8).contiguous()
    S = torch.as_tensor(np.ascontiguousarray(np.asarray(predicted_summary_indices) != 0).astype(np.uint8)).to(dev) \
        if not torch.is_tensor(predicted_summary_indices) else (predicted_summary_indices != 0).to(device=dev, dtype=U8).contiguous()
    assert gd.dim() == 2 and gd.shape[1] == S.numel()
    n_users, n = gd.shape
    out = torch.empty(2, dtype=F64, device=dev)
    counts = torch.empty(2 * (n_users + 1), dtype=I64, device=dev)
    with torch.cuda.device(dev):
        check(lib.goalnet_fscore(gd.data_ptr(), S.data_ptr(), n_users, n, out.data_ptr(), counts.data_ptr(), _s()), "fscore")
    a, m = out.tolist()
    return a, m


class SummaryEvaluator:
    """One video's static inputs (change points from the dataset's HDF5 file, annotator summaries, frame counts) kept on
    the device; `__call__(pred)` = postprocess_and_get_fscores, `postprocess(pred)` = postprocess."""

    def __init__(self, change_points, full_n_frames: int, skip_frames: int, gd_summarized_video_frame_indices=None, device=None):
        self.device = _dev(device)
        self.lib = _lib.load()
        cps = np.asarray(change_points)
        if cps.ndim != 2 or cps.shape[1] != 2 or cps.shape[0] < 1:
            raise ValueError("change_points must be [n_clips][2]")
        self.n_clips = int(cps.shape[0])
        self.full_n = int(full_n_frames)
        self.skip = int(skip_frames)
        if self.full_n < 1 or self.skip < 1:
            raise ValueError("full_n_frames and skip_frames must be positive")
        self.cps = torch.as_tensor(np.ascontiguousarray(cps.astype(np.int32))).to(self.device)
        self.capacity = int(0.15 * self.full_n)                        # utils.py:633
        self.cap_scaled = int(self.capacity * 5)                       # utils.py:478 (scale_factor = 5)
        self.gd = None
        self.n_users = 0
        if gd_summarized_video_frame_indices is not None:
            gd = np.asarray(gd_summarized_video_frame_indices)
            assert gd.ndim == 2 and gd.shape[1] == self.full_n, "gd_summary_indices must be (n_users, full_n_frames)"
            self.n_users = int(gd.shape[0])
            self.gd = torch.as_tensor(np.ascontiguousarray(gd != 0).astype(np.uint8)).to(self.device)
        dev = self.device
        self.mask = torch.empty(self.full_n, dtype=U8, device=dev)
        self.selected = torch.empty(self.n_clips, dtype=I32, device=dev)
        self.clip_values = torch.empty(self.n_clips, dtype=I64, device=dev)
        self.clip_lengths = torch.empty(self.n_clips, dtype=I32, device=dev)
        self.result = torch.zeros(3, dtype=F64, device=dev)           # [f_avg, f_max, status (int32 in the first 4 bytes)]
        self.ws_bytes = self.lib.goalnet_postprocess_ws_bytes(self.n_clips, self.cap_scaled, self.n_users)
        self.ws = torch.empty(self.ws_bytes // 8, dtype=I64, device=dev)

    def _launch(self, batch_importances, with_fscore: bool):
        pred = _importances_1d(batch_importances).detach().to(device=self.device, dtype=F32).contiguous()
        if pred.numel() < 1:
            raise IndexError("list index out of range")                 # expand_array on an empty list, utils.py:408
        gd = self.gd if with_fscore else None
        if with_fscore and gd is None:
            raise ValueError("this evaluator was built without annotator summaries")
        with torch.cuda.device(self.device):
            check(self.lib.goalnet_postprocess(
                pred.data_ptr(), pred.numel(), self.skip, self.full_n, self.cps.data_ptr(), self.n_clips, 5, self.cap_scaled,
                0 if gd is None else gd.data_ptr(), self.n_users if gd is not None else 0, self.mask.data_ptr(),
                self.selected.data_ptr(), self.clip_values.data_ptr(), self.clip_lengths.data_ptr(),
                0 if gd is None else self.result.data_ptr(), self.result[2:].data_ptr(), self.ws.data_ptr(), self.ws_bytes, _s()),
                "postprocess")

    def _status(self, host):
        if host[2:].view(torch.int32)[0].item() != 0:
            raise IndexError(f"a selected clip interval reaches outside the video's {self.full_n} frames "
                             "(utils.py:640 raises IndexError there)")

    def postprocess(self, batch_importances):
        """utils.py:606-643 (full_frames = None). Returns (selected clip indices, summarized_video_frame_indices uint8)."""
        self._launch(batch_importances, with_fscore=False)
        self._status(self.result.cpu())
        return torch.nonzero(self.selected).flatten().tolist(), self.mask.cpu().numpy()

    def __call__(self, batch_predictions) -> Tuple[float, float]:
        """utils.py:586-604: (f_score_avg, f_score_max). One 24-byte read-back."""
        self._launch(batch_predictions, with_fscore=True)
        host = self.result.cpu()
        self._status(host)
        return float(host[0]), float(host[1])


def postprocess(batch_importances, change_points, skip_frames: int, full_n_frames: int, device=None):
    """utils.py:606-643 with the change points as an argument. Returns (selected clip indices, uint8 mask)."""
    return SummaryEvaluator(change_points, full_n_frames, skip_frames, None, device).postprocess(batch_importances)


def postprocess_and_get_fscores(batch_predictions, full_n_batch_frames: int, gd_summarized_video_frame_indices, change_points,
                                skip_frames: int, device=None) -> Tuple[float, float]:
    """utils.py:586-604 with the change points as an argument."""
    return SummaryEvaluator(change_points, full_n_batch_frames, skip_frames, gd_summarized_video_frame_indices, device)(batch_predictions)


def clip_information(ev: SummaryEvaluator) -> Tuple[List[int], List[int]]:
    """clip importances and lengths of the evaluator's last call (get_clip_information, utils.py:445-463)"""
    return ev.clip_values.tolist(), ev.clip_lengths.tolist()
