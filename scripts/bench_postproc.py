"""Latency of the device post-processing + F-score (cvml_goalnet_amd/postprocess.py) per video, on the fixtures'
inputs; the CPU column is the oracle restatement (numpy-vectorised DP — faster than the reference's list-of-lists
Python, which takes 0.13 s / 2.1 s for the same two cases in the build container).

    python scripts/bench_postproc.py
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _golden import load_postproc  # noqa: E402
from cvml_goalnet_amd import postprocess as pp  # noqa: E402
from oracle import postproc_ref  # noqa: E402  (CPU baseline leg of a bench script)

out = {}
for case in ("postproc_typical_n4500", "postproc_long_n20000"):
    z = load_postproc(case)
    skip, full_n = int(z["skip"][0]), int(z["full_n"][0])
    ev = pp.SummaryEvaluator(z["change_points"], full_n, skip, z["gd"])
    pred = torch.from_numpy(z["pred"]).cuda()
    f = ev(pred)
    assert list(f) == z["fscore"].tolist()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 50
    for _ in range(reps):
        f = ev(pred)                                   # includes the 24-byte read-back (one sync per video)
    dt = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    want = postproc_ref.postprocess_and_get_fscores(z["pred"], z["change_points"], z["gd"], skip, full_n)
    cpu = time.perf_counter() - t0
    out[case] = {"device_ms_per_video": dt * 1e3, "oracle_cpu_ms": cpu * 1e3, "clips": int(z["change_points"].shape[0]),
                 "frames": full_n, "dp_cells": (int(z["change_points"].shape[0]) + 1) * (ev.cap_scaled + 1)}
print(json.dumps(out))
