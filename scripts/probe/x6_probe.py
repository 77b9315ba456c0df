"""bf16x6 convolution (csrc/split3.hip) against the fp32-MFMA convolution and fp64: accuracy on a small shape, time on the bench shapes."""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cvml_goalnet_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(3)


def split_w(w, cout, cin):
    return ops.split_rows(3, w, torch.empty(cout * 9 * 3 * cin, dtype=torch.bfloat16, device=dev), cout * 9, cin)


def run_linear(m, k, j):
    x = torch.relu(torch.randn(m, k, device=dev))
    w = torch.randn(j, k, device=dev) * 0.02
    b = torch.randn(j, device=dev)
    dy = torch.randn(m, j, device=dev)
    sc = torch.rand(512, device=dev) + 0.5
    sh = torch.rand(512, device=dev) - 0.5
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(16)]
    x3 = torch.empty(m * 3 * k, dtype=torch.bfloat16, device=dev)
    w3 = torch.empty(j * 3 * k, dtype=torch.bfloat16, device=dev)
    dy3 = ops.split_rows(3, dy, torch.empty(m * 3 * j, dtype=torch.bfloat16, device=dev), m, j)
    y = torch.empty(m, j, device=dev); dx = torch.empty(m, k, device=dev); dw = torch.empty(j, k, device=dev)
    for rep in range(2):
        ev[0].record(); ops.split_rows(3, x, x3, m, k, scale=sc, shift=sh, bnC=512)
        ev[1].record(); ops.split_rows(3, w, w3, j, k)
        ev[2].record(); ops.linear_fwd_split(3, x3, w3, b, y, m, k, j, relu=True)
        ev[3].record(); ops.linear_bwd_dx_split(3, dy3, w3, dx, m, k, j)
        ev[4].record(); ops.linear_bwd_dw_split(3, dy3, x3, dw, m, k, j)
        ev[5].record()
        y32 = torch.empty(m, j, device=dev); ops.linear_fwd(x, w, b, y32, relu=True, scale=sc, shift=sh, bnC=512)
        ev[6].record(); dx32 = torch.empty(m, k, device=dev); ops.linear_bwd_dx(dy, w, dx32)
        ev[7].record(); dw32 = torch.empty(j, k, device=dev); ops.linear_bwd_dw(dy, x, dw32, scale=sc, shift=sh, bnC=512)
        ev[8].record()
    torch.cuda.synchronize()
    t = lambda i: ev[i].elapsed_time(ev[i + 1])
    print(f"linear5 M={m} K={k} J={j}: split x {t(0):.2f} ms, split w {t(1):.2f} ms | fwd x6 {t(2):.2f} vs fp32 {t(5):.2f} | dX x6 {t(3):.2f} vs {t(6):.2f} | dW x6 {t(4):.2f} vs {t(7):.2f} ms")
    print("   x6 vs fp32-MFMA: fwd", float((y - y32).abs().max() / y32.abs().max()), " dX", float((dx - dx32).abs().max() / dx32.abs().max()),
          " dW", float((dw - dw32).abs().max() / dw32.abs().max()))


def run(n, h, w, cin, cout, check):
    x = torch.relu(torch.randn(n, h, w, cin, device=dev))
    sc = torch.rand(cin, device=dev) + 0.5
    sh = torch.rand(cin, device=dev) - 0.5
    wt = (torch.randn(cout * 9 * cin, device=dev) * 0.05)
    b = torch.randn(cout, device=dev)
    dy = torch.randn(n, h, w, cout, device=dev)
    _, xp3 = ops.padded_bf16_alloc(n, h, w, 3 * cin, dev)
    _, dyp3 = ops.padded_bf16_alloc(n, h, w, 3 * cout, dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(12)]
    ev[0].record(); ops.split_padded(3, x, sc, sh, xp3, n, h, w, cin)
    ev[1].record(); w3 = split_w(wt, cout, cin)
    y6 = torch.empty(n, h, w, cout, device=dev)
    ops.conv3x3_fwd_split(3, xp3, w3, b, True, y6, n, h, w, cin, cout)              # warm-up
    ev[2].record(); ops.conv3x3_fwd_split(3, xp3, w3, b, True, y6, n, h, w, cin, cout)
    ev[3].record()
    y32 = torch.empty(n, h, w, cout, device=dev)
    ops.conv3x3_fwd(x, sc, sh, wt, b, True, y32, n, h, w, cin, cout)
    ev[4].record(); ops.conv3x3_fwd(x, sc, sh, wt, b, True, y32, n, h, w, cin, cout)
    ev[5].record(); ops.split_padded(3, dy, None, None, dyp3, n, h, w, cout)
    ev[6].record()
    dw6 = torch.empty(cout * 9 * cin, device=dev)
    ops.conv3x3_wgrad_split(3, xp3, dyp3, dw6, n, h, w, cin, cout)
    ev[7].record(); ops.conv3x3_wgrad_split(3, xp3, dyp3, dw6, n, h, w, cin, cout)
    ev[8].record()
    dw32 = torch.empty(cout * 9 * cin, device=dev)
    ops.conv3x3_wgrad(x, sc, sh, dy, dw32, n, h, w, cin, cout)
    ev[9].record(); ops.conv3x3_wgrad(x, sc, sh, dy, dw32, n, h, w, cin, cout)
    ev[10].record()
    torch.cuda.synchronize()
    t = lambda i: ev[i].elapsed_time(ev[i + 1])
    print(f"shape n={n} {h}x{w} {cin}->{cout}: split x {t(0):.2f} ms, split dy {t(5):.2f} ms | fwd x6 {t(2):.2f} ms vs fp32 {t(4):.2f} ms | "
          f"wgrad x6 {t(7):.2f} ms vs fp32 {t(9):.2f} ms")
    print("   x6 vs fp32-MFMA: fwd max|d| / max|y| =", float((y6 - y32).abs().max() / y32.abs().max()),
          " wgrad:", float((dw6 - dw32).abs().max() / dw32.abs().max()))
    if check:
        xh = (x.double() * sc.double() + sh.double()).permute(0, 3, 1, 2)
        w4 = wt.view(cout, 3, 3, cin).permute(0, 3, 1, 2).double()
        ref = F.relu(F.conv2d(xh, w4, b.double(), padding=1)).permute(0, 2, 3, 1)
        s = ref.abs().max()
        print("   vs fp64: fwd x6", float((y6 - ref).abs().max() / s), " fp32-MFMA", float((y32 - ref).abs().max() / s))
        refdw = torch.nn.grad.conv2d_weight(xh, (cout, cin, 3, 3), dy.double().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1).reshape(-1)
        s = refdw.abs().max()
        print("   vs fp64: wgrad x6", float((dw6 - refdw).abs().max() / s), " fp32-MFMA", float((dw32 - refdw).abs().max() / s))


run(4, 19, 23, 64, 256, True)
run(3, 24, 24, 256, 512, True)
if len(sys.argv) > 1 and sys.argv[1] == "linear":
    run_linear(1024, 512 * 70 * 70, 512)
    sys.exit(0)
if len(sys.argv) > 1:
    run(1024, 72, 72, 256, 512, False)
    run(1024, 74, 74, 64, 256, False)
