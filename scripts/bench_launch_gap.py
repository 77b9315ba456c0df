"""What one more launch costs at the reference's operating point: a chain of N trivial dependent kernels (goalnet_counter_add on one
counter), eager and as a captured HIP graph. us per launch = the floor under every short kernel of the 10-frame step
(why the step is counted in launches as well as in kernel time).

    python scripts/bench_launch_gap.py [--n 200]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvml_goalnet_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=200)
    a = ap.parse_args()
    ctr = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    for _ in range(10):
        ops.counter_add(ctr, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.n):
        ops.counter_add(ctr, 1)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / a.n * 1e6
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(a.n):
            ops.counter_add(ctr, 1)
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / (reps * a.n) * 1e6
    print(json.dumps({"launches": a.n, "eager_us_per_launch": eager, "graph_us_per_launch": graph}))


if __name__ == "__main__":
    main()
