#!/usr/bin/env python3
"""Generate tests/golden/postproc_*.npz from the REFERENCE's own post-processing functions (build container only).

The reference's `postprocess` (/root/reference/utils.py:606-643) cannot be called here: it opens two HDF5 files through
h5py, which the image lacks. Its four pure-Python building blocks can — `expand_array` (utils.py:396), `get_clip_information`
(utils.py:445), `knapsack` (utils.py:465), `get_fscore` (utils.py:552) — and are imported with the recipe of SURVEY.md
Appendix B (empty module objects for the absent I/O libraries; no reference source is edited or copied). This script
chains them exactly as utils.py:608-641 does, with seeded synthetic change points / annotator summaries in place of
the HDF5 reads, and stores every input and every intermediate and final output. The three glue statements between the
calls are restated from the cited lines: round -> int8 (utils.py:611), capacity = int(0.15 n) (utils.py:633) and the
end-inclusive mask loop (utils.py:637-641).

`get_fscore` is called with int64 copies of the mask and of the annotator summaries (exact sums, the numpy 1.x
behaviour the reference was written for: both are uint8 arrays in the reference, utils.py:118, 637, and Python's sum()
over uint8 wraps at 256 under numpy >= 2) and, when every sum stays below 256, also with the uint8 arrays as the
reference builds them — the two must agree (oracle/postproc_ref.py header).

Usage: python tests/golden/make_golden_postproc.py
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import postproc_ref  # noqa: E402


def import_reference():
    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference")
    for name in ("cv2", "librosa", "h5py", "moviepy", "moviepy.editor"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["moviepy.editor"].VideoFileClip = object
    import utils  # the reference
    return utils


def make_case(seed, full_n, skip, n_clips, n_sampled=None, n_users=20, flat=False, gd_frac=0.15, zero_user=False, empty_clip=False):
    """Synthetic stand-ins for what the reference reads from the dataset: model outputs in (1,5), KTS change points as
    TVSum stores them (contiguous [start, end] pairs, end inclusive, last end = n-1) and 0/1 annotator summaries."""
    rng = np.random.default_rng(seed)
    if n_sampled is None:
        n_sampled = (full_n + skip - 1) // skip                     # frames[::skip_frames], utils.py:98
    pred = (1.0 + 4.0 * rng.random(n_sampled)).astype(np.float32)
    if flat:
        pred[:] = 3.0
    pred[: min(4, n_sampled)] = np.array([1.5, 2.5, 3.5, 4.5], dtype=np.float32)[: min(4, n_sampled)]   # exact ties: half to even
    cuts = np.sort(rng.choice(np.arange(1, full_n), size=n_clips - 1, replace=False)) if n_clips > 1 else np.array([], dtype=int)
    starts = np.concatenate([[0], cuts])
    ends = np.concatenate([cuts - 1, [full_n - 1]])
    cps = np.stack([starts, ends], axis=1).astype(np.int32)
    if empty_clip and n_clips > 2:
        cps[1, 1] = cps[1, 0]                                          # a == b: empty slice, weight 0, but one mask frame
    gd = np.zeros((n_users, full_n), dtype=np.uint8)
    for u in range(n_users):
        budget = int(gd_frac * full_n)
        while budget > 0:
            ln = int(min(budget, rng.integers(10, 120)))
            st = int(rng.integers(0, max(1, full_n - ln)))
            gd[u, st:st + ln] = 1
            budget -= ln
    if zero_user:
        gd[3, :] = 0
    return dict(pred=pred.reshape(-1, 1), change_points=cps, gd=gd, skip=skip, full_n=full_n)


CASES = {
    # name: kwargs
    "postproc_typical_n4500": dict(seed=1, full_n=4500, skip=30, n_clips=41),
    "postproc_infer_skip60_n900": dict(seed=2, full_n=900, skip=60, n_clips=12),
    "postproc_noexpand_n300": dict(seed=3, full_n=300, skip=30, n_clips=10, n_sampled=300),
    "postproc_padded_n1000": dict(seed=4, full_n=1000, skip=30, n_clips=17, n_sampled=33),
    "postproc_ties_flat_n1200": dict(seed=5, full_n=1200, skip=30, n_clips=25, flat=True, zero_user=True),
    "postproc_emptyclip_n2000": dict(seed=6, full_n=2000, skip=30, n_clips=30, empty_clip=True),
    "postproc_tiny_n5": dict(seed=7, full_n=5, skip=30, n_clips=2),
    "postproc_long_n20000": dict(seed=8, full_n=20000, skip=30, n_clips=200),
    "postproc_oneclip_n700": dict(seed=9, full_n=700, skip=30, n_clips=1),
    "postproc_wrapfree_n1500": dict(seed=10, full_n=1500, skip=30, n_clips=20),      # every sum < 256: also run on uint8 as is
}


def run_reference(utils, c):
    pred_t = torch.from_numpy(c["pred"])
    batch = pred_t[:, 0]                                               # utils.py:608-610
    imp = torch.round(batch).type(torch.int8).tolist()                 # utils.py:611
    expanded = utils.expand_array(arr=imp, expansion_rate=c["skip"], length=c["full_n"])
    cps = c["change_points"]
    vals, lens, _ = utils.get_clip_information(clip_intervals=cps, importances=expanded)
    cap = int(0.15 * c["full_n"])                                      # utils.py:633
    sel = utils.knapsack(values=vals, weights=lens, capacity=cap)
    mask = np.zeros(shape=(c["full_n"],), dtype=np.uint8)              # utils.py:637-641
    for ci in sel:
        for f in range(cps[ci][0], cps[ci][1] + 1):
            mask[f] = 1
    f_avg, f_max = utils.get_fscore(gd_summary_indices=c["gd"].astype(np.int64), predicted_summary_indices=mask.astype(np.int64))
    if int(mask.sum()) < 256 and all(int(g.sum()) < 256 for g in c["gd"]):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("error")                              # an overflow warning here would mean a wrap
            a2, m2 = utils.get_fscore(gd_summary_indices=c["gd"], predicted_summary_indices=mask)
        assert (a2, m2) == (f_avg, f_max)
        print("   (uint8 arrays as the reference builds them give the same F-scores: no wrap in this case)")
    return dict(importances=np.array(imp, dtype=np.int8), expanded=np.array(expanded, dtype=np.int8),
                clip_values=np.array(vals, dtype=np.int64), clip_lengths=np.array(lens, dtype=np.int64),
                capacity=np.array([cap], dtype=np.int64), selected=np.array(sel, dtype=np.int64), mask=mask,
                fscore=np.array([float(f_avg), float(f_max)], dtype=np.float64))


def main():
    utils = import_reference()
    for name, kw in CASES.items():
        c = make_case(**kw)
        out = run_reference(utils, c)
        # the restatement must agree bit for bit before anything is written
        imp = postproc_ref.round_importances(c["pred"])
        assert imp == out["importances"].tolist(), name
        exp = postproc_ref.expand_array(imp, c["skip"], c["full_n"])
        assert exp == out["expanded"].tolist(), name
        vals, lens = postproc_ref.get_clip_information(c["change_points"], exp)
        assert vals == out["clip_values"].tolist() and lens == out["clip_lengths"].tolist(), name
        sel = postproc_ref.knapsack(vals, lens, int(out["capacity"][0]))
        assert sel == out["selected"].tolist(), name
        sel2, mask = postproc_ref.postprocess(c["pred"], c["change_points"], c["skip"], c["full_n"])
        assert sel2 == sel and np.array_equal(mask, out["mask"]), name
        fa, fm = postproc_ref.get_fscore(c["gd"], mask)
        assert (float(fa), float(fm)) == tuple(out["fscore"].tolist()), (name, fa, fm, out["fscore"])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), pred=c["pred"], change_points=c["change_points"], gd=c["gd"],
                            skip=np.array([c["skip"]]), full_n=np.array([c["full_n"]]), **out)
        print(f"{name}: n_sampled={c['pred'].shape[0]} clips={len(c['change_points'])} cap={int(out['capacity'][0])} "
              f"selected={out['selected'].tolist()[:8]}{'...' if len(out['selected']) > 8 else ''} "
              f"frames={int(out['mask'].sum())} F avg/max = {out['fscore'][0]:.6f} / {out['fscore'][1]:.6f}")


if __name__ == "__main__":
    main()
