"""ORACLE — test infrastructure only. Never imported by the product path.

CPU restatement (pure Python / numpy integers) of the step that follows the hot path after every video
(SURVEY.md §8(f)-2): importance rounding, expansion to the raw frame rate, per-clip sums, 0/1 knapsack,
summary mask and the F-score against the annotators. It follows, and cites:

    postprocess                 /root/reference/utils.py:606-643   (HDF5 reads at 617-629 excluded: change points are an input)
    expand_array                /root/reference/utils.py:396-410
    get_clip_information        /root/reference/utils.py:445-463
    knapsack                    /root/reference/utils.py:465-510
    get_fscore                  /root/reference/utils.py:552-580
    postprocess_and_get_fscores /root/reference/utils.py:586-604

Quirks reproduced on purpose: clip importance sums and knapsack weights use the
end-EXCLUSIVE slice [a:b) (utils.py:460), the summary mask uses the end-INCLUSIVE range [a, b] (utils.py:640).

Pinning: tests/golden/make_golden_postproc.py imports the reference's own pure-Python functions
(expand_array, get_clip_information, knapsack, get_fscore) in the build container, runs them on seeded synthetic
inputs and stores inputs + outputs as tests/golden/postproc_*.npz; tests/test_oracle_golden.py checks this file
against them bit for bit. The glue between those functions (round -> int8, capacity = int(0.15 n), mask loop) has no
callable form without h5py and is restated from the cited lines.

One environment dependence is recorded rather than reproduced: `sum(S)` / `sum(G)` at utils.py:572-573 iterate uint8
numpy arrays (the mask of utils.py:637 and the annotator masks of utils.py:118) with Python's sum(). Under numpy >= 2
(NEP 50) the running total stays uint8 and wraps at 256; under the numpy 1.x the reference was written for (report:
Python 3.10 / PyTorch 2.1) it is an exact int64. The restatement and the device kernel compute the exact sums; the
fixtures hold the values the reference returns when handed int64 copies of the same arrays (an input dtype, not a
code change), plus a wrap-free case (every sum < 256) run on the uint8 arrays as they are.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def round_importances(pred: np.ndarray) -> List[int]:
    """utils.py:608-611: (N,1) or (N,) float32 -> torch.round (half to even) -> int8 -> list"""
    p = np.asarray(pred, dtype=np.float32)
    if p.ndim != 1:
        assert p.ndim == 2 and p.shape[-1] == 1, "E: Invalid shape for importance tensor"
        p = p[:, 0]
    return np.rint(p).astype(np.int8).tolist()


def expand_array(arr: Sequence[int], expansion_rate: int, length: int) -> List[int]:
    """utils.py:396-410"""
    if len(arr) == length:
        return list(arr)
    out: List[int] = []
    for el in arr:
        out += [el] * expansion_rate
    if len(out) >= length:
        out = out[:length]
    else:
        out += [out[-1]] * (length - len(out))
    return out


def get_clip_information(clip_intervals, importances: Sequence[int]) -> Tuple[List[int], List[int]]:
    """utils.py:445-463: end-exclusive Python slices (clamped to the array, as slicing does)"""
    vals, lens = [], []
    for a, b in clip_intervals:
        sl = importances[int(a):int(b)]
        vals.append(int(sum(sl)))
        lens.append(len(sl))
    return vals, lens


def knapsack(values: Sequence[int], weights: Sequence[float], capacity, scale_factor=5) -> List[int]:
    """utils.py:465-510, including its back-tracking rule (stop when the remaining value is <= 0)"""
    weights = [int(w * scale_factor) for w in weights]
    capacity = int(capacity * scale_factor)
    n = len(values)
    K = np.zeros((n + 1, capacity + 1), dtype=np.int64)
    for i in range(1, n + 1):
        wt, v = weights[i - 1], values[i - 1]
        K[i, :] = K[i - 1, :]
        if wt <= capacity:
            lo = max(wt, 1)                                       # w == 0 stays 0 (utils.py:486)
            cand = v + K[i - 1, lo - wt:capacity + 1 - wt]
            K[i, lo:] = np.maximum(cand, K[i - 1, lo:])
        K[i, 0] = 0
    res = int(K[n, capacity])
    w = capacity
    selected = []
    for i in range(n, 0, -1):
        if res <= 0:
            break
        if res == int(K[i - 1, w]):
            continue
        selected.append(i - 1)
        res -= values[i - 1]
        w -= weights[i - 1]
    selected.reverse()
    return selected


def summary_mask(clip_intervals, selected: Sequence[int], full_n_frames: int) -> np.ndarray:
    """utils.py:637-641: end-INCLUSIVE; an interval reaching past the video raises IndexError, as numpy does there"""
    mask = np.zeros((full_n_frames,), dtype=np.uint8)
    for c in selected:
        a, b = int(clip_intervals[c][0]), int(clip_intervals[c][1])
        for f in range(a, b + 1):
            mask[f] = 1
    return mask


def get_fscore(gd_summary_indices: np.ndarray, predicted_summary_indices: np.ndarray) -> Tuple[float, float]:
    """utils.py:552-580 with exact integer sums (see the header on sum() over uint8 under numpy >= 2)"""
    n_users = gd_summary_indices.shape[0]
    assert gd_summary_indices.shape[1] == len(predicted_summary_indices)
    S = np.asarray(predicted_summary_indices)
    f_scores = []
    for user in range(n_users):
        G = np.asarray(gd_summary_indices[user])
        ov = int(np.logical_and(S, G).sum())
        s_sum = int(S.astype(np.int64).sum())
        g_sum = G.astype(np.float64).sum() if G.dtype.kind == "f" else int(G.astype(np.int64).sum())
        precision = ov / s_sum if s_sum != 0 else 0
        recall = ov / g_sum if g_sum != 0 else 0
        f_scores.append(2 * precision * recall / (precision + recall) if (precision + recall) != 0 else 0)
    return sum(f_scores) / len(f_scores), max(f_scores)


def postprocess(pred, change_points, skip_frames: int, full_n_frames: int):
    """utils.py:606-643 with `change_points` ([n_clips][2]) as an input. Returns (selected clip indices, mask)."""
    imp = round_importances(pred)
    expanded = expand_array(imp, skip_frames, full_n_frames)
    vals, lens = get_clip_information(change_points, expanded)
    cap = int(0.15 * full_n_frames)                               # utils.py:633
    selected = knapsack(vals, lens, cap)
    return selected, summary_mask(change_points, selected, full_n_frames)


def postprocess_and_get_fscores(pred, change_points, gd_summaries, skip_frames: int, full_n_frames: int):
    """utils.py:586-604"""
    _, mask = postprocess(pred, change_points, skip_frames, full_n_frames)
    return get_fscore(gd_summaries, mask)
