"""Race screen for the phased 256 x 256 kernels: the same launch repeated many times must give bit-identical results
(their reductions have a fixed order), at sizes with whole and ragged tiles, while other work keeps the memory system busy.
Usage: python scripts/race_screen.py [repeats] [bf16|fp16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvml_goalnet_amd import ops

dev = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
H16 = torch.float16 if (len(sys.argv) > 2 and sys.argv[2] == "fp16") else torch.bfloat16      # the 16-bit format under test
g = torch.Generator().manual_seed(5)
bad = 0


def padded(x):
    n, h, w, c = x.shape
    buf, view = ops.padded_bf16_alloc(n, h, w, c, dev, dtype=H16)
    ops.to_bf16_padded(x.to(dev), None, None, view, n, h, w, c)
    return buf, view


def screen(name, fn, out):
    global bad
    fn(); ref = out.clone()
    noise = torch.empty(64 << 20, device=dev)
    diff = 0
    for i in range(reps):
        noise.normal_()                      # unrelated traffic in the same stream's neighbourhood
        out.fill_(float("nan")) if out.dtype == torch.float32 else out.zero_()
        fn()
        if not torch.equal(out, ref):
            diff += 1
    print(f"{name:60s} {reps} runs, {diff} differ")
    bad += diff


for n, h, w, cin, cout in ((49, 72, 72, 256, 512), (25, 74, 74, 64, 256), (13, 70, 66, 128, 320)):
    x = (torch.rand(n, h, w, cin, generator=g) - 0.5)
    dy = (torch.rand(n, h, w, cout, generator=g) - 0.5)
    wt = ((torch.rand(cout, 3, 3, cin, generator=g) - 0.5) * 0.1).to(H16).to(dev)
    bx, xp = padded(x); bd, dyp = padded(dy)
    os.environ["GOALNET_BF16_TILE"] = "256"
    dw = torch.empty(cout, 3, 3, cin, device=dev)
    screen(f"conv wgrad {n}x{h}x{w} {cin}->{cout}", lambda: ops.conv3x3_wgrad_bf16(xp, dyp, dw, n, h, w, cin, cout), dw)
    y = torch.empty(n, h, w, cout, device=dev)
    b = torch.zeros(cout, device=dev)
    screen(f"conv fwd   {n}x{h}x{w} {cin}->{cout}", lambda: ops.conv3x3_fwd_bf16p(xp, wt, b, True, y, n, h, w, cin, cout), y)
    if cout % 8 == 0:
        y16 = torch.empty(n, h, w, cout, dtype=H16, device=dev)
        screen(f"conv fwd (bf16 out) {n}x{h}x{w} {cin}->{cout}", lambda: ops.conv3x3_fwd_bf16p_o16(xp, wt, b, True, y16, n, h, w, cin, cout), y16)
for m, j, k in ((320, 512, (1 << 18) + 264), (1024, 512, 1 << 19), (96, 128, 70000)):
    dyl = (torch.rand(m, j, generator=g) - 0.5).to(H16).to(dev)
    xl = (torch.rand(m, k, generator=g) - 0.5).to(H16).to(dev)
    wl = (torch.rand(j, k, generator=g) - 0.5).to(H16).to(dev)
    dwl = torch.empty(j, k, device=dev)
    screen(f"linear dW  M={m} J={j} K={k}", lambda: ops.linear_bwd_dw_bf16(dyl, xl, dwl), dwl)
    dxl = torch.empty(m, k, device=dev)
    screen(f"linear dX  M={m} J={j} K={k}", lambda: ops.linear_bwd_dx_bf16(dyl, wl, dxl), dxl)
    dx16 = torch.empty(m, k, dtype=H16, device=dev)
    screen(f"linear dX (bf16 out) M={m} J={j} K={k}", lambda: ops.linear_bwd_dx_bf16_o16(dyl, wl, dx16), dx16)
    yl = torch.empty(m, j, device=dev)
    bl = torch.zeros(j, device=dev)
    if k % 64 == 0:
        screen(f"linear fwd M={m} J={j} K={k}", lambda: ops.linear_fwd_bf16(xl, wl, bl, yl, relu=True), yl)
print("RACE SCREEN:", "clean" if bad == 0 else f"{bad} differing runs")
sys.exit(1 if bad else 0)
