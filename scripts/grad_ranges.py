#!/usr/bin/env python3
"""Magnitudes of the activation gradients that a 16-bit backward would have to store (dz of the BatchNorm outputs, dy of the
convolutions, linear5's dz) — to choose the loss scale of precision="fp16". fp32 model, random init and after a few steps.
    python scripts/grad_ranges.py [frames] [hw] [steps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cvml_goalnet_amd import AVM, ops, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 224
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
torch.manual_seed(1234)
m = AVM(True, device=dev, seed=synth.BASE_SEED, precision="fp32")
aud, vis, lab = bench.make_inputs(n, hw, hw, dev, synth.BASE_SEED)
stats = {}


def rec(name, t):
    flat = t.detach().reshape(-1)
    step_ = max(1, flat.numel() // (1 << 24))            # a strided sample of <= 16 M elements
    a = flat[::step_].float().abs()
    nz = a[a > 0]
    stats.setdefault(name, []).append((a.max().item(), nz.median().item() if nz.numel() else 0.0,
                                       (a > 0).float().mean().item(), nz.min().item() if nz.numel() else 0.0))

orig_bwd, orig_dx, orig_conv = ops.bnpool_bwd, ops.linear_bwd_dx, ops.conv3x3_fwd
def bnpool_bwd(dz, p, idx, coef3, dy, dparts, N, Hc, Wc, C):
    rec(f"dbn(C={C})", dz)
    orig_bwd(dz, p, idx, coef3, dy, dparts, N, Hc, Wc, C)
    rec(f"dy(C={C})", dy)
def linear_bwd_dx(dy, w, dx, mult=None):
    if w.numel() > 10_000_000:
        rec("dz5", dy)
    return orig_dx(dy, w, dx, mult=mult)
ops.bnpool_bwd, ops.linear_bwd_dx = bnpool_bwd, linear_bwd_dx
for s in range(steps):
    stats.clear()
    m.train_step(aud, vis, lab)
    torch.cuda.synchronize()
    print(f"--- step {s} (N={n}, {hw}x{hw}): max / median of non-zeros / fraction non-zero / smallest non-zero")
    for k, v in stats.items():
        mx, med, frac, mn = v[0]
        print(f"  {k:12s} max {mx:.3e}  median {med:.3e}  nonzero {frac:.3f}  min {mn:.3e}")
