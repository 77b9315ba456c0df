"""GPU: post-processing + F-score kernels (reference utils.py:396-410, 445-510, 552-643) — integer
work, so everything here is bit-exact: against the fixtures produced by the reference's own functions
(tests/golden/postproc_*.npz), against the oracle on random cases, and through properties at sizes past the fixtures."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from _golden import POSTPROC_CASES, load_postproc  # noqa: E402
from cvml_goalnet_amd import postprocess as pp  # noqa: E402
from oracle import postproc_ref  # noqa: E402


@pytest.mark.parametrize("case", POSTPROC_CASES)
def test_pipeline_matches_reference_goldens_bit_for_bit(case):
    z = load_postproc(case)
    skip, full_n = int(z["skip"][0]), int(z["full_n"][0])
    ev = pp.SummaryEvaluator(z["change_points"], full_n, skip, z["gd"])
    assert ev.capacity == int(z["capacity"][0])
    pred_gpu = torch.from_numpy(z["pred"]).cuda()                      # (N,1), as the model returns it
    f_avg, f_max = ev(pred_gpu)
    assert [f_avg, f_max] == z["fscore"].tolist()                      # identical doubles
    vals, lens = pp.clip_information(ev)
    assert vals == z["clip_values"].tolist() and lens == z["clip_lengths"].tolist()
    sel, mask = ev.postprocess(pred_gpu)
    assert sel == z["selected"].tolist()
    assert mask.dtype == np.uint8 and np.array_equal(mask, z["mask"])
    # the functional forms, CPU inputs
    sel2, mask2 = pp.postprocess(torch.from_numpy(z["pred"]), z["change_points"], skip, full_n)
    assert sel2 == sel and np.array_equal(mask2, mask)
    assert pp.postprocess_and_get_fscores(torch.from_numpy(z["pred"]), full_n, z["gd"], z["change_points"], skip) == (f_avg, f_max)
    assert pp.get_fscore(z["gd"], z["mask"]) == (f_avg, f_max)


def test_knapsack_random_cases_equal_the_oracle():
    rng = np.random.default_rng(11)
    for trial in range(40):
        n = int(rng.integers(1, 60))
        values = rng.integers(0, 400, size=n).tolist()
        if trial % 4 == 0:
            weights = (rng.integers(0, 90, size=n) / 2.0).tolist()    # fractional durations: int(w * 5) truncates
        else:
            weights = rng.integers(0, 90, size=n).tolist()
        capacity = float(rng.integers(0, 300)) if trial % 5 else 0
        if trial % 7 == 0:
            values = [3] * n                                           # ties everywhere: the back-tracking rule decides
        want = postproc_ref.knapsack(values, weights, capacity)
        assert pp.knapsack(values, weights, capacity) == want, (trial, values, weights, capacity)
    assert pp.knapsack([], [], 10) == []
    assert pp.knapsack([5, 6], [2, 3], 100, scale_factor=1) == postproc_ref.knapsack([5, 6], [2, 3], 100, scale_factor=1)


def test_fscore_random_cases_equal_the_oracle():
    rng = np.random.default_rng(12)
    for trial in range(12):
        n = int(rng.integers(1, 5000))
        users = int(rng.integers(1, 25))
        gd = (rng.random((users, n)) < rng.random()).astype(np.uint8)
        S = (rng.random(n) < rng.random()).astype(np.uint8)
        if trial == 0:
            S[:] = 0
        if trial == 1:
            gd[:] = 0
        want = postproc_ref.get_fscore(gd, S)
        got = pp.get_fscore(gd, S)
        assert got == (float(want[0]), float(want[1])), trial


def test_errors_and_edge_inputs():
    cps = np.array([[0, 4], [5, 12]], dtype=np.int32)                  # second interval reaches frame 12 of a 10-frame video
    pred = torch.tensor([5.0, 5.0, 5.0, 5.0, 5.0, 5.0, 5.0, 5.0, 5.0, 5.0])
    with pytest.raises(IndexError):
        postproc_ref.summary_mask(cps, [1], 10)
    ev = pp.SummaryEvaluator(np.array([[0, 1], [2, 12]], dtype=np.int32), 40, 4, None)
    # capacity int(0.15*40)*5 = 30; clip 1 = frames [2:12) weight 50 -> not selectable; clip 0 weight 5 -> selected
    sel, mask = ev.postprocess(pred)
    assert sel == [0] and mask[:2].tolist() == [1, 1] and mask.sum() == 2
    ev2 = pp.SummaryEvaluator(np.array([[0, 1], [37, 40]], dtype=np.int32), 40, 4, None)     # [37, 40] inclusive leaves the video
    with pytest.raises(IndexError):
        ev2.postprocess(pred)
    with pytest.raises(AssertionError):
        ev.postprocess(torch.zeros(10, 2))
    with pytest.raises(ValueError):
        pp.SummaryEvaluator(np.zeros((0, 2)), 40, 4)
    with pytest.raises(ValueError):
        ev(pred)                                                       # no annotator summaries


def test_large_video_equals_the_oracle_and_respects_the_budget():
    rng = np.random.default_rng(13)
    full_n, skip, n_clips = 60000, 30, 500
    n_sampled = (full_n + skip - 1) // skip
    pred = (1.0 + 4.0 * rng.random(n_sampled)).astype(np.float32)
    cuts = np.sort(rng.choice(np.arange(1, full_n), size=n_clips - 1, replace=False))
    cps = np.stack([np.concatenate([[0], cuts]), np.concatenate([cuts - 1, [full_n - 1]])], axis=1).astype(np.int32)
    gd = (rng.random((20, full_n)) < 0.15).astype(np.uint8)
    ev = pp.SummaryEvaluator(cps, full_n, skip, gd)
    f = ev(torch.from_numpy(pred))
    sel, mask = ev.postprocess(torch.from_numpy(pred))
    vals, lens = pp.clip_information(ev)
    assert sum(lens[c] * 5 for c in sel) <= ev.cap_scaled               # the knapsack budget holds
    want_sel, want_mask = postproc_ref.postprocess(pred, cps, skip, full_n)
    assert sel == want_sel and np.array_equal(mask, want_mask)
    want_f = postproc_ref.get_fscore(gd, want_mask)
    assert f == (float(want_f[0]), float(want_f[1]))
