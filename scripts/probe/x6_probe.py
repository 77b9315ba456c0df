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
    return ops.split3_rows(w, torch.empty(cout * 9 * 3 * cin, dtype=torch.bfloat16, device=dev), cout * 9, cin)


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
    ev[0].record(); ops.split3_padded(x, sc, sh, xp3, n, h, w, cin)
    ev[1].record(); w3 = split_w(wt, cout, cin)
    y6 = torch.empty(n, h, w, cout, device=dev)
    ops.conv3x3_fwd_x6(xp3, w3, b, True, y6, n, h, w, cin, cout)              # warm-up
    ev[2].record(); ops.conv3x3_fwd_x6(xp3, w3, b, True, y6, n, h, w, cin, cout)
    ev[3].record()
    y32 = torch.empty(n, h, w, cout, device=dev)
    ops.conv3x3_fwd(x, sc, sh, wt, b, True, y32, n, h, w, cin, cout)
    ev[4].record(); ops.conv3x3_fwd(x, sc, sh, wt, b, True, y32, n, h, w, cin, cout)
    ev[5].record(); ops.split3_padded(dy, None, None, dyp3, n, h, w, cout)
    ev[6].record()
    dw6 = torch.empty(cout * 9 * cin, device=dev)
    ops.conv3x3_wgrad_x6(xp3, dyp3, dw6, n, h, w, cin, cout)
    ev[7].record(); ops.conv3x3_wgrad_x6(xp3, dyp3, dw6, n, h, w, cin, cout)
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
if len(sys.argv) > 1:
    run(1024, 72, 72, 256, 512, False)
    run(1024, 74, 74, 64, 256, False)
