"""The fp32 engine's GEMM launches of the bench step, one op at a time (for counter passes: scripts/pmc_traffic.sh).

    python scripts/f32_traffic_probe.py [--ops fwd3,dgrad3,wgrad3,fwd2,dgrad2,wgrad2,l5fwd,l5dx,l5dw] [--frames 1024] [--hw 224] [--reps 2]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvml_goalnet_amd import AVM, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ops", default="fwd3,dgrad3,wgrad3,fwd2,dgrad2,wgrad2,l5fwd,l5dx,l5dw")
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--hw", type=int, default=224)
    ap.add_argument("--reps", type=int, default=2)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    n = a.frames
    (_, _), (hp1, wp1), (hp2, wp2), (hp3, wp3) = AVM._sizes(a.hw, a.hw)
    want = set(a.ops.split(","))
    for tag, hh, ww, cin, cout in (("3", hp2, wp2, 256, 512), ("2", hp1, wp1, 64, 256)):
        if not ({"fwd" + tag, "dgrad" + tag, "wgrad" + tag} & want):
            continue
        x = torch.rand(n, hh, ww, cin, device=dev)
        sc = torch.rand(cin, device=dev) + 0.5
        sh = torch.rand(cin, device=dev) - 0.5
        w = (torch.rand(cout * 9 * cin, device=dev) - 0.5) * 0.05
        b = torch.rand(cout, device=dev)
        y = torch.empty(n, hh, ww, cout, device=dev)
        for _ in range(a.reps):
            if "fwd" + tag in want:
                ops.conv3x3_fwd(x, sc, sh, w, b, True, y, n, hh, ww, cin, cout)
            if "dgrad" + tag in want:
                wt = ops.conv3x3_weight_flip(w, torch.empty_like(w), cout, cin)
                ops.conv3x3_fwd(y, None, None, wt, None, False, x, n, hh, ww, cout, cin)
            if "wgrad" + tag in want:
                ops.conv3x3_wgrad(x, sc, sh, y, torch.empty_like(w), n, hh, ww, cin, cout)
        torch.cuda.synchronize()
        del x, y
    if {"l5fwd", "l5dx", "l5dw"} & want:
        k5 = 512 * hp3 * wp3
        p3 = torch.rand(n, k5, device=dev)
        w5 = (torch.rand(512 * k5, device=dev) - 0.5) * 0.01
        sc = torch.rand(512, device=dev) + 0.5
        sh = torch.rand(512, device=dev) - 0.5
        b = torch.rand(512, device=dev)
        y = torch.empty(n, 512, device=dev)
        dz = torch.rand(n, 512, device=dev)
        for _ in range(a.reps):
            if "l5fwd" in want:
                ops.linear_fwd(p3, w5, b, y, relu=True, scale=sc, shift=sh, bnC=512)
            if "l5dx" in want:
                ops.linear_bwd_dx(dz, w5, p3, mult=None)
            if "l5dw" in want:
                ops.linear_bwd_dw(dz, p3, torch.empty_like(w5), scale=sc, shift=sh, bnC=512)
        torch.cuda.synchronize()
    print("ok")


if __name__ == "__main__":
    main()
