"""Micro-benchmark of the GEMM-engine entry points at the bench shapes (fewer frames). Prints TFLOP/s per op."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvml_goalnet_amd import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = "cuda:0"
torch.manual_seed(0)

def timeit(fn, reps=4):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts), sum(ts) / len(ts)

def conv_case(name, h, w, cin, cout):
    x = torch.randn(n, h, w, cin, device=dev); sc = torch.rand(cin, device=dev) + 0.5; sh = torch.randn(cin, device=dev) * 0.1
    wt = torch.randn(cout, 3, 3, cin, device=dev) * 0.05; b = torch.randn(cout, device=dev)
    y = torch.empty(n, h, w, cout, device=dev)
    fl = 2.0 * n * h * w * 9 * cin * cout
    t, ta = timeit(lambda: ops.conv3x3_fwd(x, sc, sh, wt, b, True, y, n, h, w, cin, cout))
    print(f"{name} fwd   : {t:8.3f} ms  {fl / t / 1e9:7.1f} TF/s (avg {fl / ta / 1e9:.1f})")
    dy = torch.randn(n, h, w, cout, device=dev)
    wtf = torch.empty(cout * 9 * cin, device=dev); ops.conv3x3_weight_flip(wt, wtf, cout, cin)
    dx = torch.empty(n, h, w, cin, device=dev)
    t, ta = timeit(lambda: ops.conv3x3_fwd(dy, None, None, wtf, None, False, dx, n, h, w, cout, cin))
    print(f"{name} dgrad : {t:8.3f} ms  {fl / t / 1e9:7.1f} TF/s (avg {fl / ta / 1e9:.1f})")
    dw = torch.empty(cout, 3, 3, cin, device=dev)
    t, ta = timeit(lambda: ops.conv3x3_wgrad(x, sc, sh, dy, dw, n, h, w, cin, cout))
    print(f"{name} wgrad : {t:8.3f} ms  {fl / t / 1e9:7.1f} TF/s (avg {fl / ta / 1e9:.1f})")
    xb = x.to(torch.bfloat16); wb = wt.to(torch.bfloat16)
    t, ta = timeit(lambda: ops.conv3x3_fwd_bf16(xb, wb, b, True, y, n, h, w, cin, cout))
    print(f"{name} fwd bf16   : {t:8.3f} ms  {fl / t / 1e9:7.1f} TF/s (avg {fl / ta / 1e9:.1f})")
    bx, xp = ops.padded_bf16_alloc(n, h, w, cin, dev); ops.to_bf16_padded(x, sc, sh, xp, n, h, w, cin)
    bd, dyp = ops.padded_bf16_alloc(n, h, w, cout, dev); ops.to_bf16_padded(dy, None, None, dyp, n, h, w, cout)
    t, ta = timeit(lambda: ops.conv3x3_fwd_bf16p(xp, wb, b, True, y, n, h, w, cin, cout))
    print(f"{name} fwd bf16p  : {t:8.3f} ms  {fl / t / 1e9:7.1f} TF/s (avg {fl / ta / 1e9:.1f})")
    t, ta = timeit(lambda: ops.conv3x3_wgrad_bf16(xp, dyp, dw, n, h, w, cin, cout))
    print(f"{name} wgrad bf16 : {t:8.3f} ms  {fl / t / 1e9:7.1f} TF/s (avg {fl / ta / 1e9:.1f})")
    t, ta = timeit(lambda: ops.to_bf16_padded(x, sc, sh, xp, n, h, w, cin))
    print(f"{name} to_bf16_padded(x): {t:8.3f} ms  {(x.numel() * 6) / t / 1e6:7.0f} GB/s")
    dyb = dy.to(torch.bfloat16); wfb = wtf.to(torch.bfloat16)
    t, ta = timeit(lambda: ops.conv3x3_fwd_bf16(dyb, wfb, None, False, dx, n, h, w, cout, cin))
    print(f"{name} dgrad bf16 : {t:8.3f} ms  {fl / t / 1e9:7.1f} TF/s (avg {fl / ta / 1e9:.1f})")

def lin5_case(hw3):
    k = 512 * hw3
    x = torch.randn(n, k, device=dev); w = torch.randn(512, k, device=dev) * 0.01; b = torch.randn(512, device=dev)
    sc = torch.rand(512, device=dev) + 0.5; sh = torch.randn(512, device=dev) * 0.1
    y = torch.empty(n, 512, device=dev); fl = 2.0 * n * k * 512
    t, ta = timeit(lambda: ops.linear_fwd(x, w, b, y, relu=True, scale=sc, shift=sh, bnC=512))
    print(f"linear5 fwd   : {t:8.3f} ms  {fl / t / 1e9:7.1f} TF/s   ({w.numel() * 4 / t / 1e6:.0f} GB/s weight stream)")
    xb = x.to(torch.bfloat16); wb = w.to(torch.bfloat16)
    t, ta = timeit(lambda: ops.linear_fwd_bf16(xb, wb, b, y, relu=True))
    print(f"linear5 fwd bf16: {t:8.3f} ms  {fl / t / 1e9:7.1f} TF/s   ({wb.numel() * 2 / t / 1e6:.0f} GB/s weight stream)")
    dy = torch.randn(n, 512, device=dev); dx = torch.empty(n, k, device=dev)
    dyb = dy.to(torch.bfloat16)
    t, ta = timeit(lambda: ops.linear_bwd_dx_bf16(dyb, wb, dx))
    print(f"linear5 dX bf16 : {t:8.3f} ms  {fl / t / 1e9:7.1f} TF/s")
    dwb = torch.empty(512, k, device=dev)
    t, ta = timeit(lambda: ops.linear_bwd_dw_bf16(dyb, xb, dwb))
    print(f"linear5 dW bf16 : {t:8.3f} ms  {fl / t / 1e9:7.1f} TF/s")
    del dwb
    t, ta = timeit(lambda: ops.linear_bwd_dx(dy, w, dx))
    print(f"linear5 dX    : {t:8.3f} ms  {fl / t / 1e9:7.1f} TF/s")
    dw = torch.empty(512, k, device=dev)
    t, ta = timeit(lambda: ops.linear_bwd_dw(dy, x, dw, scale=sc, shift=sh, bnC=512))
    print(f"linear5 dW    : {t:8.3f} ms  {fl / t / 1e9:7.1f} TF/s")

print(f"N = {n} frames; variant = {os.environ.get('GOALNET_GEMM_VARIANT', '0')}")
which = sys.argv[2] if len(sys.argv) > 2 else "gemm"
if which == "gemm":
    conv_case("conv3 72x72 256->512", 72, 72, 256, 512)
    conv_case("conv2 74x74  64->256", 74, 74, 64, 256)
if which in ("gemm", "lin"):
    lin5_case(70 * 70)


def pool_case(name, hc, wc, c):
    """HBM-bound block kernels: max-pool + BN statistics forward, fused BN/pool/ReLU backward"""
    y = torch.relu(torch.randn(n, hc, wc, c, device=dev))
    hp, wp = hc - 2, wc - 2
    p = torch.empty(n, hp, wp, c, device=dev); idx = torch.empty(n, hp, wp, c, dtype=torch.uint8, device=dev)
    parts = ops.stat_parts(8 * n)
    partials = torch.empty(parts * 2 * c, dtype=torch.float64, device=dev)
    t, ta = timeit(lambda: ops.pool_bnstats_fwd(y, p, idx, partials, n, hc, wc, c))
    gb = (y.numel() * 4 + p.numel() * 5) / 1e9
    print(f"{name} pool fwd : {t:8.3f} ms  {gb / t * 1e3:7.0f} GB/s")
    t, ta = timeit(lambda: ops.pool_bnstats_fwd(y, p, None, partials, n, hc, wc, c))
    print(f"{name} pool fwd without the argmax store: {t:8.3f} ms  {(y.numel() * 4 + p.numel() * 4) / 1e9 / t * 1e3:7.0f} GB/s")
    dz = torch.randn(n, hp, wp, c, device=dev); coef3 = torch.randn(3 * c, device=dev)
    dparts = torch.empty(parts * c, dtype=torch.float64, device=dev)
    dy = torch.empty(n, hc, wc, c, device=dev)
    t, ta = timeit(lambda: ops.bnpool_bwd(dz, p, idx, coef3, dy, dparts, n, hc, wc, c))
    gb = (dz.numel() * 4 + p.numel() * 5 + y.numel() * 4) / 1e9
    print(f"{name} bnpool bwd (fp32 dy): {t:8.3f} ms  {gb / t * 1e3:7.0f} GB/s")
    bd, dyp = ops.padded_bf16_alloc(n, hc, wc, c, dev)
    t, ta = timeit(lambda: ops.bnpool_bwd_bf16p(dz, p, idx, coef3, None, dyp, dparts, n, hc, wc, c))
    gb = (dz.numel() * 4 + p.numel() * 5 + y.numel() * 2) / 1e9
    print(f"{name} bnpool bwd (bf16 padded dy): {t:8.3f} ms  {gb / t * 1e3:7.0f} GB/s")
    dzb = dz.to(torch.bfloat16)
    t, ta = timeit(lambda: ops.bnpool_bwd_bf16p(dzb, p, idx, coef3, None, dyp, dparts, n, hc, wc, c))
    gb = (dz.numel() * 2 + p.numel() * 5 + y.numel() * 2) / 1e9
    print(f"{name} bnpool bwd (bf16 dz, bf16 padded dy): {t:8.3f} ms  {gb / t * 1e3:7.0f} GB/s")
    # the 16-bit step's own variants: 16-bit conv output -> 16-bit pooled activation; 16-bit dz and p -> 16-bit padded dy
    yb = y.to(torch.bfloat16); pb = torch.empty(n, hp, wp, c, dtype=torch.bfloat16, device=dev)
    t, ta = timeit(lambda: ops.pool_bnstats_fwd(yb, pb, idx, partials, n, hc, wc, c))
    gb = (y.numel() * 2 + p.numel() * 3) / 1e9
    print(f"{name} pool fwd (bf16 y -> bf16 p): {t:8.3f} ms  {gb / t * 1e3:7.0f} GB/s")
    t, ta = timeit(lambda: ops.bnpool_bwd_bf16p(dzb, pb, idx, coef3, None, dyp, dparts, n, hc, wc, c))
    gb = (dz.numel() * 2 + p.numel() * 3 + y.numel() * 2) / 1e9
    print(f"{name} bnpool bwd (bf16 dz, bf16 p, bf16 padded dy): {t:8.3f} ms  {gb / t * 1e3:7.0f} GB/s")
    mean = torch.randn(c, device=dev); invstd = torch.rand(c, device=dev) + 0.5
    rparts = torch.empty(ops.stat_parts(n * hp * wp // 64) * 2 * c, dtype=torch.float64, device=dev)
    t, ta = timeit(lambda: ops.bn_bwd_reduce(dz, p, mean, invstd, rparts, n * hp * wp, c))
    print(f"{name} bn_bwd_reduce fp32: {t:8.3f} ms  {dz.numel() * 8 / 1e9 / t * 1e3:7.0f} GB/s")
    t, ta = timeit(lambda: ops.bn_bwd_reduce(dzb, pb, mean, invstd, rparts, n * hp * wp, c))
    print(f"{name} bn_bwd_reduce bf16: {t:8.3f} ms  {dz.numel() * 4 / 1e9 / t * 1e3:7.0f} GB/s")


if len(sys.argv) > 2 and sys.argv[2] == "pool":
    pool_case("block3 72x72x512", 72, 72, 512)
    pool_case("block2 74x74x256", 74, 74, 256)


def conv1_case(h, w):
    x = torch.rand(n, 3, h, w, device=dev); wt = torch.randn(64, 3, 3, 3, device=dev) * 0.1; b = torch.randn(64, device=dev)
    ho, wo = (h + 3) // 3 + 1, (w + 3) // 3 + 1
    y = torch.empty(n, ho, wo, 64, device=dev)
    t, ta = timeit(lambda: ops.conv1_fwd(x, wt, b, y, n, h, w))
    print(f"conv1 fwd   {h}x{w}: {t:8.3f} ms  {(x.numel() + y.numel()) * 4 / t / 1e6:7.0f} GB/s")
    dy = torch.randn(n, ho, wo, 64, device=dev); dw = torch.empty(64, 3, 3, 3, device=dev); db = torch.empty(64, device=dev)
    t, ta = timeit(lambda: ops.conv1_wgrad(x, dy, dw, db, n, h, w))
    print(f"conv1 wgrad {h}x{w}: {t:8.3f} ms  {(x.numel() + dy.numel()) * 4 / t / 1e6:7.0f} GB/s")


if len(sys.argv) > 2 and sys.argv[2] == "conv1":
    conv1_case(224, 224)
