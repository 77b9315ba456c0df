"""GPU: every C-ABI kernel against the CPU oracle ops (torch CPU, fp64) on seeded inputs, called through the
C ABI (cvml_goalnet_amd.ops -> ctypes -> libgoalnet_hip.so). Tolerances are stated per test; integer/index
results (argmax positions, generator bits, masks) are bit-exact."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from cvml_goalnet_amd import ops, synth  # noqa: E402
from oracle import avm_ref  # noqa: E402

DEV = "cuda:0"
RT = 2e-6   # fp32 kernels vs fp64 reference, relative to the tensor's max |value|


def rnd(*shape, seed=0, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g, dtype=torch.float64) * (hi - lo) + lo).float()


def close(name, got, want, rtol=RT, atol=0.0):
    got = got.detach().cpu().double()
    want = want.detach().cpu().double()
    assert got.shape == want.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(want.shape)}"
    scale = max(want.abs().max().item(), 1e-30)
    err = (got - want).abs().max().item()
    print(f"[parity] {name}: max|err| = {err:.3e}  scale = {scale:.3e}  rel = {err / scale:.3e}")
    assert err <= rtol * scale + atol, f"{name}: err {err:.3e} > {rtol:.1e} * {scale:.3e}"


def nhwc(t):   # NCHW -> NHWC contiguous
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


# ------------------------------------------------------------------------------------------------
def test_fill_uniform_and_dropout_mask_bit_exact():
    for tid, n, lo, hi in ((3, 1000, -0.5, 0.25), (77, 70001, 0.0, 1.0)):
        d = torch.empty(n, device=DEV)
        ops.fill_uniform(d, synth.BASE_SEED, tid, lo, hi)
        want = synth.uniform(tid, (n,), lo, hi)
        assert np.array_equal(d.cpu().numpy(), want)
    for step in (0, 2):
        masks = synth.make_drop_masks(9, step=step)
        for li, m in enumerate(masks):
            d = torch.empty(m.shape, device=DEV)
            ops.dropout_mask(d, synth.BASE_SEED, synth.TID_DROP + 8 * step + li, synth.DROP_P)
            assert np.array_equal(d.cpu().numpy(), m)


def test_layout_converters_exact():
    x = rnd(5, 37, 70, seed=1)
    d = torch.empty(5 * 37 * 70, device=DEV)
    ops.transpose_inner(x.to(DEV).view(-1), d, 5, 37, 70)
    assert torch.equal(d.cpu().view(5, 70, 37), x.permute(0, 2, 1).contiguous())
    w = rnd(8, 3, 3, 12, seed=2)      # OHWI
    wt = torch.empty(w.numel(), device=DEV)
    ops.conv3x3_weight_flip(w.to(DEV).view(-1), wt, 8, 12)
    want = w.flip(1, 2).permute(3, 1, 2, 0).contiguous()     # [ci][2-kh][2-kw][co]
    assert torch.equal(wt.cpu().view(12, 3, 3, 8), want)


@pytest.mark.parametrize("n,h,w", [(1, 40, 40), (3, 41, 38), (2, 7, 9), (2, 224, 224), (3, 33, 9)])
def test_conv1_fwd_and_wgrad(n, h, w, monkeypatch):
    x = rnd(n, 3, h, w, seed=3, lo=0, hi=1)
    wt = rnd(64, 3, 3, 3, seed=4, lo=-0.2, hi=0.2)       # OIHW
    b = rnd(64, seed=5, lo=-0.2, hi=0.2)
    ref = F.relu(F.conv2d(x.double(), wt.double(), b.double(), stride=3, padding=3))
    ho, wo = ref.shape[2], ref.shape[3]
    y = torch.empty(n, ho, wo, 64, device=DEV)
    w_ohwi = wt.permute(0, 2, 3, 1).contiguous().to(DEV)
    ops.conv1_fwd(x.to(DEV), w_ohwi, b.to(DEV), y, n, h, w)
    close("conv1_fwd", y, nhwc(ref))
    dy = rnd(n, ho, wo, 64, seed=6)
    xd = x.double().requires_grad_(False)
    wd = wt.double().requires_grad_(True)
    bd = b.double().requires_grad_(True)
    out = F.conv2d(xd, wd, bd, stride=3, padding=3)
    out.backward(nchw(dy).double())
    dw = torch.empty(64, 3, 3, 3, device=DEV)
    db = torch.empty(64, device=DEV)
    ops.conv1_wgrad(x.to(DEV), dy.to(DEV), dw, db, n, h, w)
    close("conv1_wgrad.dw", dw, wd.grad.permute(0, 2, 3, 1), rtol=2e-5)
    close("conv1_wgrad.db", db, bd.grad, rtol=2e-5)
    # the row-staged kernels keep the first generation's order of operations: identical bits
    monkeypatch.setenv("GOALNET_CONV1_V1", "1")
    y1 = torch.empty_like(y)
    ops.conv1_fwd(x.to(DEV), w_ohwi, b.to(DEV), y1, n, h, w)
    assert torch.equal(y, y1)
    dw1, db1 = torch.empty_like(dw), torch.empty_like(db)
    ops.conv1_wgrad(x.to(DEV), dy.to(DEV), dw1, db1, n, h, w)
    assert torch.equal(dw, dw1) and torch.equal(db, db1)


@pytest.mark.parametrize("n,h,w,cin,cout,affine,bias,relu", [
    (2, 7, 5, 64, 256, True, True, True),       # M = 70 < one tile
    (3, 13, 13, 64, 256, True, True, True),     # reference conv2 geometry (40x40 frames)
    (2, 11, 11, 256, 512, True, True, True),    # reference conv3 geometry
    (2, 11, 11, 512, 256, False, False, False), # data-gradient form of conv3
    (2, 13, 13, 256, 64, False, False, False),  # data-gradient form of conv2 (N tile mostly empty)
    (16, 11, 11, 512, 256, False, False, False),# M = 1936: 16 M-tiles, ragged last tile
    (16, 11, 11, 256, 512, True, True, True),
    (1, 3, 3, 64, 128, True, False, True),
    (40, 26, 22, 256, 64, False, False, False), # 128 x 64 tile at M = 22 880 pixels (179 M-tiles, ragged last one), no split-K
    (3, 9, 7, 64, 32, True, True, True),        # 128 x 64 tile: 32 of its 64 columns valid, BatchNorm on the load, bias + ReLU
    (2, 5, 5, 128, 60, False, True, False),     # 60 columns (Cout % 4 == 0 is all the entry point asks)
])
def test_conv3x3_fwd(n, h, w, cin, cout, affine, bias, relu):
    x = rnd(n, h, w, cin, seed=7)
    sc = rnd(cin, seed=8, lo=0.5, hi=1.5) if affine else None
    sh = rnd(cin, seed=9, lo=-0.5, hi=0.5) if affine else None
    wt = rnd(cout, 3, 3, cin, seed=10, lo=-0.05, hi=0.05)
    b = rnd(cout, seed=11) if bias else None
    xn = x.double() * sc.double() + sh.double() if affine else x.double()
    ref = F.conv2d(nchw(xn), wt.double().permute(0, 3, 1, 2), None if b is None else b.double(), padding=1)
    if relu:
        ref = F.relu(ref)
    y = torch.full((n, h, w, cout), float("nan"), device=DEV)
    ops.conv3x3_fwd(x.to(DEV), None if sc is None else sc.to(DEV), None if sh is None else sh.to(DEV), wt.to(DEV),
                    None if b is None else b.to(DEV), relu, y, n, h, w, cin, cout)
    close(f"conv3x3_fwd[{n}x{h}x{w}x{cin}->{cout}]", y, nhwc(ref), rtol=5e-6)   # K up to 4608 fp32 accumulations


@pytest.mark.parametrize("path", ["select", "correct"])
@pytest.mark.parametrize("n,h,w,cin,cout,affine", [
    (2, 7, 5, 64, 256, True),
    (3, 13, 13, 64, 256, True),
    (2, 11, 11, 256, 512, True),
    (5, 9, 6, 64, 128, False),
    (16, 11, 11, 256, 512, True),     # M = 1936 -> 3 splits over the reduction
    (16, 13, 13, 64, 256, True),      # M = 2704 -> 5 splits, 576 columns (4.5 N-tiles)
    (40, 13, 13, 64, 256, False),
    (2, 1, 9, 64, 128, True),         # one image row: first and last row coincide (every kh != 1 tap is padding)
    (3, 4, 1, 64, 128, True),         # one image column
    (1, 1, 1, 64, 128, True),
])
def test_conv3x3_wgrad(n, h, w, cin, cout, affine, path, monkeypatch):
    # "select": padding taps zeroed in the loop (few frames); "correct": padding taps load zeros, the BatchNorm shift they pick up
    # is subtracted in the slab reduction from border sums of dy (the bench-size path, forced here onto small shapes)
    monkeypatch.setenv("GOALNET_WGRAD_PATH", path)
    x = rnd(n, h, w, cin, seed=12)
    sc = rnd(cin, seed=13, lo=0.5, hi=1.5) if affine else None
    sh = rnd(cin, seed=14, lo=-0.5, hi=0.5) if affine else None
    dy = rnd(n, h, w, cout, seed=15)
    xn = x.double() * sc.double() + sh.double() if affine else x.double()
    ref = torch.nn.grad.conv2d_weight(nchw(xn), (cout, cin, 3, 3), nchw(dy.double()), padding=1)
    dw = torch.full((cout, 3, 3, cin), float("nan"), device=DEV)
    ops.conv3x3_wgrad(x.to(DEV), None if sc is None else sc.to(DEV), None if sh is None else sh.to(DEV), dy.to(DEV), dw,
                      n, h, w, cin, cout)
    close(f"conv3x3_wgrad[{n}x{h}x{w}x{cin}->{cout}]", dw, ref.permute(0, 2, 3, 1), rtol=5e-6)


@pytest.mark.parametrize("n,hc,wc,c,parts", [(2, 15, 15, 64, 0), (3, 13, 13, 256, 0), (1, 11, 11, 512, 0), (2, 3, 5, 64, 1024),
                                              (16, 11, 11, 512, 5), (20, 13, 13, 256, 1024), (3, 15, 13, 64, 12), (2, 76, 9, 64, 37),
                                              (10, 11, 11, 512, 80)])
def test_pool_bn_forward_and_backward(n, hc, wc, c, parts):
    parts = parts or ops.stat_parts(n)                # partial rows: one per frame / fewer / several row bands per frame
    z = rnd(n, hc, wc, c, seed=16)
    y = F.relu(z)                                     # conv output after ReLU: many exact zeros (ties)
    gamma = rnd(c, seed=17, lo=0.5, hi=1.5)
    beta = rnd(c, seed=18, lo=-0.5, hi=0.5)
    rm0 = rnd(c, seed=19)
    rv0 = rnd(c, seed=20, lo=0.5, hi=2.0)
    # ---- oracle (fp64 autograd): utils.py:175-177
    zd = nchw(z.double()).requires_grad_(True)
    yd = F.relu(zd)
    pd, pidx = F.max_pool2d(yd, 3, 1, 0, return_indices=True)
    rm, rv = rm0.double().clone(), rv0.double().clone()
    gd = gamma.double().requires_grad_(True)
    bd = beta.double().requires_grad_(True)
    od = F.batch_norm(pd, rm, rv, gd, bd, training=True, momentum=0.1, eps=1e-5)
    G = rnd(*od.shape, seed=21).double()
    (od * G).sum().backward()
    # ---- device
    hp, wp = hc - 2, wc - 2
    yg = y.to(DEV)
    p = torch.empty(n, hp, wp, c, device=DEV)
    idx = torch.empty(n, hp, wp, c, dtype=torch.uint8, device=DEV)
    partials = torch.empty(parts * 2 * c, dtype=torch.float64, device=DEV)
    ops.pool_bnstats_fwd(yg, p, idx, partials, n, hc, wc, c)
    st = torch.empty(4, c, device=DEV)
    rmg, rvg = rm0.to(DEV), rv0.to(DEV)
    ops.bn_finalize(partials, gamma.to(DEV), beta.to(DEV), rmg, rvg, 0.1, 1e-5, n * hp * wp, c, st[0], st[1], st[2], st[3])
    close("maxpool", p, nhwc(pd), rtol=0.0)
    # argmax: torch flat index ih*Wc+iw -> tap (ih-ph)*3 + (iw-pw); first-max tie rule must agree (bit-exact)
    ih = pidx // wc
    iw = pidx % wc
    ph = torch.arange(hp).view(1, 1, hp, 1)
    pw = torch.arange(wp).view(1, 1, 1, wp)
    tap = ((ih - ph) * 3 + (iw - pw)).to(torch.uint8)
    assert torch.equal(ops.idx_to_nhwc(idx, n, hp, wp, c).cpu(), nhwc(tap)), "argmax positions differ from ATen's"
    mean = pd.mean(dim=(0, 2, 3))
    var = pd.var(dim=(0, 2, 3), unbiased=False)
    close("bn.mean", st[0], mean, rtol=1e-6)
    close("bn.invstd", st[1], 1.0 / torch.sqrt(var + 1e-5), rtol=1e-6)
    close("bn.running_mean", rmg, rm, rtol=1e-6)
    close("bn.running_var", rvg, rv, rtol=1e-6)
    bn_out = p * st[2] + st[3]
    close("bn.apply(scale,shift)", bn_out, nhwc(od), rtol=2e-6)
    # ---- backward chain
    dbn = nhwc(G.float()).to(DEV)
    ops.bn_bwd_reduce(dbn, p, st[0], st[1], partials, n * hp * wp, c)
    coef3 = torch.empty(3 * c, device=DEV)
    dgamma = torch.empty(c, device=DEV)
    dbeta = torch.empty(c, device=DEV)
    ops.bn_bwd_finalize(partials, gamma.to(DEV), st[0], st[1], n * hp * wp, c, dgamma, dbeta, coef3)
    dy = torch.empty(n, hc, wc, c, device=DEV)
    dparts = torch.empty(parts * c, dtype=torch.float64, device=DEV)
    ops.bnpool_bwd(dbn, p, idx, coef3, dy, dparts, n, hc, wc, c)
    dbias = torch.empty(c, device=DEV)
    ops.partials_sum(dparts, parts, c, c, dbias)
    close("bn.dgamma", dgamma, gd.grad, rtol=5e-6)
    close("bn.dbeta", dbeta, bd.grad, rtol=5e-6)
    close("block.dz (bn+pool+relu bwd)", dy, nhwc(zd.grad), rtol=1e-5)
    # sum(dz) cancels (BatchNorm backward sums to ~0 per channel): tolerance relative to |dz| * sqrt(count)
    close("block.dbias", dbias, zd.grad.sum(dim=(0, 2, 3)), rtol=0.0,
          atol=3e-6 * zd.grad.abs().max().item() * (n * hc * wc) ** 0.5)


@pytest.mark.parametrize("m,k,j,affine,mask,ldextra", [
    (10, 41472, 512, True, True, 128),     # linear5 at the reference's sub-batch (split-K path)
    (37, 640, 512, False, True, 0),        # fusion.0
    (130, 512, 256, False, False, 0),      # ragged M over two tiles
    (1, 1024, 128, False, False, 512),     # audbl.linear3, N = 1
    (16, 1056, 260, False, True, 0),       # weight-streaming path: largest row count, ragged J block, two K slabs
    (7, 41472, 512, True, False, 128),     # a ragged last sub-batch of the reference's loop
    (17, 41472, 512, True, True, 0),       # first row count on the MFMA path
])
def test_linear_fwd(m, k, j, affine, mask, ldextra):
    x = rnd(m, k, seed=22)
    w = rnd(j, k, seed=23, lo=-0.05, hi=0.05)
    b = rnd(j, seed=24)
    bnc = 512 if affine else 0
    sc = rnd(512, seed=25, lo=0.5, hi=1.5) if affine else None
    sh = rnd(512, seed=26, lo=-0.5, hi=0.5) if affine else None
    dm = (torch.rand(m, j, generator=torch.Generator().manual_seed(27)) >= 0.2).float() * 1.25 if mask else None
    xd = x.double()
    if affine:
        ch = torch.arange(k) % 512
        xd = xd * sc.double()[ch] + sh.double()[ch]
    pre = xd @ w.double().t() + b.double()
    ref = F.relu(pre) * (dm.double() if mask else 1.0)
    ybuf = torch.full((m, j + ldextra), float("nan"), device=DEV)
    mbuf = torch.full((m, j + ldextra), float("nan"), device=DEV)
    yv, mv = ybuf[:, ldextra:], mbuf[:, ldextra:]
    ops.linear_fwd(x.to(DEV), w.to(DEV), b.to(DEV), yv, relu=True, scale=None if sc is None else sc.to(DEV),
                   shift=None if sh is None else sh.to(DEV), bnC=bnc, dropmask=None if dm is None else dm.to(DEV), mult_out=mv)
    close(f"linear_fwd[{m}x{k}->{j}]", yv, ref, rtol=3e-6)
    # mult = (pre > 0) * mask; pre-activations within fp32 noise of 0 may legitimately differ
    want_mult = (pre > 0).double() * (dm.double() if mask else 1.0)
    safe = pre.abs() > 1e-4
    assert torch.equal(mv.cpu().double()[safe], want_mult[safe])
    if ldextra:
        assert torch.isnan(ybuf[:, :ldextra]).all(), "wrote outside the column slice"


@pytest.mark.parametrize("m,k,j,use_mult", [(10, 41472, 512, False), (37, 640, 512, True), (130, 512, 256, True), (1, 256, 128, True),
                                            (16, 1000, 96, True), (5, 41472, 512, True), (12, 640, 1056, True)])
def test_linear_bwd_dx(m, k, j, use_mult):
    dy = rnd(m, j, seed=28)
    w = rnd(j, k, seed=29, lo=-0.05, hi=0.05)
    mult = (torch.rand(m, k, generator=torch.Generator().manual_seed(30)) >= 0.5).float() * 1.25 if use_mult else None
    ref = dy.double() @ w.double()
    if use_mult:
        ref = ref * mult.double()
    dx = torch.full((m, k), float("nan"), device=DEV)
    ops.linear_bwd_dx(dy.to(DEV), w.to(DEV), dx, mult=None if mult is None else mult.to(DEV))
    close(f"linear_bwd_dx[{m}x{j}->{k}]", dx, ref, rtol=3e-6)


@pytest.mark.parametrize("m,k,j,affine", [(10, 41472, 512, True), (37, 640, 512, False), (130, 512, 256, False), (1, 1024, 128, False),
                                          (16, 1028, 36, False), (12, 41472, 512, True), (3, 2048, 512, True)])
def test_linear_bwd_dw(m, k, j, affine):
    dy = rnd(m, j, seed=31)
    x = rnd(m, k, seed=32)
    sc = rnd(512, seed=33, lo=0.5, hi=1.5) if affine else None
    sh = rnd(512, seed=34, lo=-0.5, hi=0.5) if affine else None
    xd = x.double()
    if affine:
        ch = torch.arange(k) % 512
        xd = xd * sc.double()[ch] + sh.double()[ch]
    ref = dy.double().t() @ xd
    dw = torch.full((j, k), float("nan"), device=DEV)
    db = torch.full((j,), float("nan"), device=DEV)
    ops.linear_bwd_dw(dy.to(DEV), x.to(DEV), dw, scale=None if sc is None else sc.to(DEV), shift=None if sh is None else sh.to(DEV),
                      bnC=512 if affine else 0, db=db)
    close(f"linear_bwd_dw[{m}: {j}x{k}]", dw, ref, rtol=3e-6)
    close("linear_bwd_dw.db", db, dy.double().sum(0), rtol=1e-6)
    out = torch.empty(j, device=DEV)
    ops.colsum(dy.to(DEV), out)
    close("colsum", out, dy.double().sum(0), rtol=1e-6)


@pytest.mark.parametrize("n,bins", [(1, 30), (10, 30), (33, 17), (70, 30), (257, 21), (515, 30)])      # >= 64 frames: the many-frame gradient kernels; >= 512: frame slices
def test_audbl_conv1d(n, bins):
    x = rnd(n, 30, bins, seed=35, lo=-50, hi=50)
    w1 = rnd(64, 30, 3, seed=36, lo=-0.1, hi=0.1); b1 = rnd(64, seed=37)
    w2 = rnd(128, 64, 3, seed=38, lo=-0.07, hi=0.07); b2 = rnd(128, seed=39)
    xd = x.double()
    w1d, b1d, w2d, b2d = (t.double().requires_grad_(True) for t in (w1, b1, w2, b2))
    a1 = F.relu(F.conv1d(xd, w1d, b1d, stride=2, padding=1))
    a2 = F.relu(F.conv1d(a1, w2d, b2d, stride=2, padding=1))
    G = rnd(*a2.shape, seed=40).double()
    (a2 * G).sum().backward()
    l1, l2 = a1.shape[2], a2.shape[2]
    xg = x.to(DEV)
    g1 = torch.empty(n, 64, l1, device=DEV); g2 = torch.empty(n, 128, l2, device=DEV)
    ops.conv1d_fwd(xg, w1.to(DEV), b1.to(DEV), g1, True, n, 30, bins, 64)
    ops.conv1d_fwd(g1, w2.to(DEV), b2.to(DEV), g2, True, n, 64, l1, 128)
    close("audbl.conv1", g1, a1, rtol=3e-6); close("audbl.conv2", g2, a2, rtol=3e-6)
    dz2 = torch.empty_like(g2)
    ops.relu_bwd(G.float().to(DEV), g2, dz2)
    da1 = torch.empty_like(g1); dw2 = torch.empty(128, 64, 3, device=DEV); db2 = torch.empty(128, device=DEV)
    ops.conv1d_bwd(g1, dz2, w2.to(DEV), da1, dw2, db2, n, 64, l1, 128)
    ops.relu_bwd(da1, g1, da1)
    dw1 = torch.empty(64, 30, 3, device=DEV); db1 = torch.empty(64, device=DEV)
    ops.conv1d_bwd(xg, da1, w1.to(DEV), None, dw1, db1, n, 30, bins, 64)
    close("audbl.dw2", dw2, w2d.grad, rtol=1e-5); close("audbl.db2", db2, b2d.grad, rtol=1e-5)
    close("audbl.dw1", dw1, w1d.grad, rtol=1e-5); close("audbl.db1", db1, b1d.grad, rtol=1e-5)


@pytest.mark.parametrize("n", [1, 10, 300])
def test_head_and_mse(n):
    h = rnd(n, 128, seed=41)
    w = rnd(128, seed=42, lo=-0.1, hi=0.1)
    b = rnd(1, seed=43)
    lab = torch.from_numpy(synth.make_labels(n))
    mult = (torch.rand(n, 128, generator=torch.Generator().manual_seed(44)) >= 0.2).float() * 1.25
    hd = h.double().requires_grad_(True); wd = w.double().requires_grad_(True); bd = b.double().requires_grad_(True)
    z = hd @ wd + bd
    out = 4 * torch.sigmoid(z) + 1
    loss = avm_ref.mse_bcast(out.view(n, 1), lab.double())
    loss.backward()
    logit = torch.empty(n, device=DEV); og = torch.empty(n, device=DEV)
    ops.head_fwd(h.to(DEV), w.to(DEV), b.to(DEV), logit, og)
    close("head.logit", logit, z, rtol=2e-6, atol=2e-7); close("head.out", og, out, rtol=1e-6)
    lg = torch.empty(1, device=DEV); dpred = torch.empty(n, device=DEV)
    ops.mse_bcast(og, lab.to(DEV), lg, dpred)
    close("mse_bcast.loss", lg, loss.detach().view(1), rtol=2e-6)
    dh = torch.empty(n, 128, device=DEV); dw = torch.empty(128, device=DEV); db = torch.empty(1, device=DEV)
    ops.head_bwd(dpred, og, h.to(DEV), w.to(DEV), mult.to(DEV), dh, dw, db)
    close("head.dh", dh, hd.grad * mult.double(), rtol=5e-6)
    close("head.dw", dw, wd.grad, rtol=5e-6); close("head.db", db, bd.grad.view(1), rtol=5e-6)


def test_adam_three_steps_matches_torch_adam():
    n = 100003
    p0 = rnd(n, seed=45)
    p_ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([p_ref], lr=1e-3)
    pg = p0.to(DEV).clone(); m = torch.zeros(n, device=DEV); v = torch.zeros(n, device=DEV)
    # 16-byte aligned arenas, odd length exercises the tail
    for step in range(1, 4):
        g = rnd(n, seed=45 + step, lo=-1e-2, hi=1e-2)
        g[::7] = 0.0
        p_ref.grad = g.clone()
        opt.step()
        ops.adam_step(pg, g.to(DEV), m, v, 1e-3, 0.9, 0.999, 1e-8, step)
        close(f"adam.step{step}", pg, p_ref.detach(), rtol=0.0, atol=2e-7)


# ------------------------------------------------------------------------------------------------
# precision = "bf16" engine. Products of bf16 values are exact in fp32 and accumulation is fp32, so against an
# fp64 reference computed from the SAME bf16-rounded operands the kernels are as tight as the fp32 ones.
# ------------------------------------------------------------------------------------------------
def test_bf16_cast_passes_bit_exact():
    x = rnd(3, 5, 7, 64, seed=50, lo=-3, hi=3)
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=DEV)
    ops.cast_bf16(x.to(DEV), y)
    assert torch.equal(y.cpu(), x.to(torch.bfloat16))                # round-to-nearest-even, as torch
    sc = rnd(64, seed=51, lo=0.5, hi=1.5); sh = rnd(64, seed=52)
    ops.bn_apply_bf16(x.to(DEV), sc.to(DEV), sh.to(DEV), y, 64)
    want = torch.addcmul(sh, x, sc)                                   # fp32 fma, then one rounding to bf16
    got = y.cpu().float()
    ulp = (want.abs() * 2.0 ** -8).clamp_min(1e-30)
    assert ((got - want).abs() <= ulp).all()


@pytest.mark.parametrize("n,h,w,cin,cout,bias,relu", [
    (2, 7, 5, 64, 256, True, True),
    (3, 13, 13, 64, 256, True, True),
    (16, 11, 11, 256, 512, True, True),
    (16, 11, 11, 512, 256, False, False),     # data-gradient form
    (2, 13, 13, 256, 64, False, False),
])
def test_conv3x3_fwd_bf16(n, h, w, cin, cout, bias, relu):
    x = rnd(n, h, w, cin, seed=53).to(torch.bfloat16)
    wt = rnd(cout, 3, 3, cin, seed=54, lo=-0.05, hi=0.05).to(torch.bfloat16)
    b = rnd(cout, seed=55) if bias else None
    ref = F.conv2d(nchw(x.double()), wt.double().permute(0, 3, 1, 2), None if b is None else b.double(), padding=1)
    if relu:
        ref = F.relu(ref)
    y = torch.full((n, h, w, cout), float("nan"), device=DEV)
    ops.conv3x3_fwd_bf16(x.to(DEV), wt.to(DEV), None if b is None else b.to(DEV), relu, y, n, h, w, cin, cout)
    close(f"conv3x3_fwd_bf16[{n}x{h}x{w}x{cin}->{cout}]", y, nhwc(ref), rtol=5e-6)


@pytest.mark.parametrize("tile", ["128", "256"])
@pytest.mark.parametrize("m,k,j", [(10, 41472, 512), (37, 640, 512), (130, 512, 256), (300, 8192, 320)])
def test_linear_fwd_bf16(m, k, j, tile, monkeypatch):
    monkeypatch.setenv("GOALNET_BF16_TILE", tile)
    x = rnd(m, k, seed=56).to(torch.bfloat16)
    w = rnd(j, k, seed=57, lo=-0.05, hi=0.05).to(torch.bfloat16)
    b = rnd(j, seed=58)
    dm = (torch.rand(m, j, generator=torch.Generator().manual_seed(59)) >= 0.2).float() * 1.25
    ref = F.relu(x.double() @ w.double().t() + b.double()) * dm.double()
    y = torch.full((m, j), float("nan"), device=DEV)
    mv = torch.empty(m, j, device=DEV)
    ops.linear_fwd_bf16(x.to(DEV), w.to(DEV), b.to(DEV), y, relu=True, dropmask=dm.to(DEV), mult_out=mv)
    close(f"linear_fwd_bf16[{m}x{k}->{j}]", y, ref, rtol=3e-6)


def _padded(x_nhwc_f32, sc=None, sh=None):
    n, h, w, c = x_nhwc_f32.shape
    buf, view = ops.padded_bf16_alloc(n, h, w, c, DEV)
    ops.to_bf16_padded(x_nhwc_f32.to(DEV), None if sc is None else sc.to(DEV), None if sh is None else sh.to(DEV), view, n, h, w, c)
    return buf, view


def test_to_bf16_padded_layout_exact():
    x = rnd(2, 5, 7, 64, seed=60)
    sc = rnd(64, seed=61, lo=0.5, hi=1.5); sh = rnd(64, seed=62)
    buf, view = _padded(x, sc, sh)
    n, h, w, c = x.shape
    got = view[: n * (h + 2) * (w + 2) * c].view(n, h + 2, w + 2, c).cpu().float()
    want = torch.zeros(n, h + 2, w + 2, c)
    want[:, 1:-1, 1:-1, :] = torch.addcmul(sh, x, sc).to(torch.bfloat16).float()
    assert torch.equal(got, want)
    assert float(buf.float().abs().sum()) == float(want.abs().sum())       # guard bands stay zero


@pytest.mark.parametrize("tile", ["128", "256"])          # the 128 x 128 kernel and the 256 x 256 phased kernel (gemm_bf16_256.hip)
@pytest.mark.parametrize("n,h,w,cin,cout,bias,relu", [
    (2, 7, 5, 64, 256, True, True), (3, 13, 13, 64, 256, True, True), (16, 11, 11, 256, 512, True, True),
    (16, 11, 11, 512, 256, False, False), (2, 13, 13, 256, 64, False, False), (9, 40, 36, 64, 320, True, True),
    (1, 7, 5, 64, 256, True, True), (3, 7, 7, 64, 260, False, False),      # 35 / 147 pixels: the last quad of rows is only partly valid
    (40, 26, 22, 256, 64, False, False), (3, 9, 7, 64, 32, True, True),    # <= 64 output channels: the 128 x 64 tile (tile "128")
])
def test_conv3x3_fwd_bf16_padded_input(n, h, w, cin, cout, bias, relu, tile, monkeypatch):
    monkeypatch.setenv("GOALNET_BF16_TILE", tile)         # read by the entry point at call time; unset = chosen by size
    x = rnd(n, h, w, cin, seed=63)
    wt = rnd(cout, 3, 3, cin, seed=64, lo=-0.05, hi=0.05).to(torch.bfloat16)
    b = rnd(cout, seed=65) if bias else None
    _, xp = _padded(x)
    ref = F.conv2d(nchw(x.to(torch.bfloat16).double()), wt.double().permute(0, 3, 1, 2), None if b is None else b.double(), padding=1)
    if relu:
        ref = F.relu(ref)
    y = torch.full((n, h, w, cout), float("nan"), device=DEV)
    ops.conv3x3_fwd_bf16p(xp, wt.to(DEV), None if b is None else b.to(DEV), relu, y, n, h, w, cin, cout)
    close(f"conv3x3_fwd_bf16p[{n}x{h}x{w}x{cin}->{cout}]", y, nhwc(ref), rtol=5e-6)


@pytest.mark.parametrize("tile", ["128", "256"])
@pytest.mark.parametrize("n,h,w,cin,cout", [(2, 7, 5, 64, 256), (3, 13, 13, 64, 256), (16, 11, 11, 256, 512), (5, 9, 6, 64, 128),
                                            (40, 30, 26, 64, 320)])
def test_conv3x3_wgrad_bf16(n, h, w, cin, cout, tile, monkeypatch):
    monkeypatch.setenv("GOALNET_BF16_TILE", tile)
    x = rnd(n, h, w, cin, seed=66)
    dy = rnd(n, h, w, cout, seed=67)
    _, xp = _padded(x)
    _, dyp = _padded(dy)
    ref = torch.nn.grad.conv2d_weight(nchw(x.to(torch.bfloat16).double()), (cout, cin, 3, 3), nchw(dy.to(torch.bfloat16).double()), padding=1)
    dw = torch.full((cout, 3, 3, cin), float("nan"), device=DEV)
    ops.conv3x3_wgrad_bf16(xp, dyp, dw, n, h, w, cin, cout)
    close(f"conv3x3_wgrad_bf16[{n}x{h}x{w}x{cin}->{cout}]", dw, ref.permute(0, 2, 3, 1), rtol=5e-6)


@pytest.mark.parametrize("tile", ["128", "256"])
@pytest.mark.parametrize("m,k,j,use_mult", [(10, 41472, 512, False), (37, 640, 512, True), (130, 512, 256, True), (300, 4168, 320, False)])
def test_linear_bwd_bf16(m, k, j, use_mult, tile, monkeypatch):
    monkeypatch.setenv("GOALNET_BF16_TILE", tile)
    dy = rnd(m, j, seed=68).to(torch.bfloat16)
    w = rnd(j, k, seed=69, lo=-0.05, hi=0.05).to(torch.bfloat16)
    x = rnd(m, k, seed=70).to(torch.bfloat16)
    mult = (torch.rand(m, k, generator=torch.Generator().manual_seed(71)) >= 0.5).float() * 1.25 if use_mult else None
    ref = dy.double() @ w.double()
    if use_mult:
        ref = ref * mult.double()
    dx = torch.full((m, k), float("nan"), device=DEV)
    ops.linear_bwd_dx_bf16(dy.to(DEV), w.to(DEV), dx, mult=None if mult is None else mult.to(DEV))
    close(f"linear_bwd_dx_bf16[{m}x{j}->{k}]", dx, ref, rtol=3e-6)
    dw = torch.full((j, k), float("nan"), device=DEV)
    ops.linear_bwd_dw_bf16(dy.to(DEV), x.to(DEV), dw)
    close(f"linear_bwd_dw_bf16[{m}: {j}x{k}]", dw, dy.double().t() @ x.double(), rtol=3e-6)


def _conv_rows_fp64(xb, wt, pix):
    """fp64 3x3/s1/p1 convolution of the bf16-rounded NHWC tensor xb with OHWI weights wt at the sampled pixels `pix`
    ((n, h, w) triples): every output channel of those pixels. Products of bf16 values are exact in fp64."""
    n, h, w, cin = xb.shape
    xp = F.pad(xb.double(), (0, 0, 1, 1, 1, 1))                          # zero border, as the reference pads
    out = torch.empty(len(pix), wt.shape[0], dtype=torch.float64)
    w2 = wt.double().reshape(wt.shape[0], -1)                            # [co][(kh, kw, ci)]
    for r, (a, i, j) in enumerate(pix):
        out[r] = w2 @ xp[a, i:i + 3, j:j + 3, :].reshape(-1)
    return out


def _sample_pixels(n, h, w, k, seed):
    g = torch.Generator().manual_seed(seed)
    pix = [(0, 0, 0), (n - 1, h - 1, w - 1), (0, 0, w - 1), (n - 1, h - 1, 0), (n // 2, 0, w // 2), (n // 2, h - 1, 1)]
    pix += [(int(torch.randint(n, (1,), generator=g)), int(torch.randint(h, (1,), generator=g)), int(torch.randint(w, (1,), generator=g)))
            for _ in range(k)]
    return pix


def test_phased_256_kernels_at_the_sizes_that_select_them_vs_fp64(monkeypatch):
    """At bench-like sizes the entry points pick the 256 x 256 phased kernels by themselves (M >= 65 536 pixels for
    forward / data gradient, >= 262 144 padded pixels for the weight gradient). Checked against fp64 on the bf16-rounded
    operands: every output channel of 134 sampled pixels (all four image corners, edges, tile interiors and the ragged last
    tile) for the forward and the data gradient; every tap of 16 x 16 sampled (cout, cin) pairs for the weight gradient,
    whose reduction runs over all 254 016 pixels. The 128 x 128 kernels get the same check on the same operands."""
    monkeypatch.delenv("GOALNET_BF16_TILE", raising=False)
    n, h, w, cin, cout = 49, 72, 72, 256, 512                            # conv3 of the 224 x 224 model, 49 frames
    g = torch.Generator().manual_seed(70)
    x = (torch.rand(n, h, w, cin, generator=g) - 0.5)
    dy = (torch.rand(n, h, w, cout, generator=g) - 0.5)
    wt = ((torch.rand(cout, 3, 3, cin, generator=g) - 0.5) * 0.1).to(torch.bfloat16)
    b = (torch.rand(cout, generator=g) - 0.5)
    wflip = torch.empty(cout * 9 * cin, device=DEV)
    ops.conv3x3_weight_flip(wt.float().contiguous().to(DEV), wflip, cout, cin)
    wflip = wflip.to(torch.bfloat16)
    xb, dyb = x.to(torch.bfloat16), dy.to(torch.bfloat16)
    _, xp = _padded(x)
    _, dyp = _padded(dy)
    pix = _sample_pixels(n, h, w, 128, 71)
    sel = torch.tensor([(a * h + i) * w + j for a, i, j in pix])
    ref_y = F.relu(_conv_rows_fp64(xb, wt, pix) + b.double())
    ref_dx = _conv_rows_fp64(dyb, wflip.cpu().view(cin, 3, 3, cout), pix)
    co_s = torch.tensor([0, 1, 63, 64, 127, 128, 255, 256, 257, 300, 383, 384, 448, 500, 510, 511])
    ci_s = torch.tensor([0, 1, 31, 32, 63, 64, 65, 100, 127, 128, 129, 191, 192, 200, 254, 255])
    ref_dw = torch.nn.grad.conv2d_weight(nchw(xb[..., ci_s].double()), (16, 16, 3, 3), nchw(dyb[..., co_s].double()), padding=1)
    ref_dw = ref_dw.permute(0, 2, 3, 1)                                 # [co_s][kh][kw][ci_s]
    wtd = wt.to(DEV)
    for tile in ("auto", "128"):
        if tile == "128":
            monkeypatch.setenv("GOALNET_BF16_TILE", tile)
        y = torch.full((n, h, w, cout), float("nan"), device=DEV)
        ops.conv3x3_fwd_bf16p(xp, wtd, b.to(DEV), True, y, n, h, w, cin, cout)
        close(f"conv3 forward [{tile}] vs fp64 at {len(pix)} pixels", y.view(-1, cout)[sel.to(DEV)], ref_y, rtol=5e-6)
        assert torch.isfinite(y).all()
        if tile == "auto":
            y_auto = y
        dx = torch.full((n, h, w, cin), float("nan"), device=DEV)
        ops.conv3x3_fwd_bf16p(dyp, wflip, None, False, dx, n, h, w, cout, cin)
        close(f"conv3 data gradient [{tile}] vs fp64 at {len(pix)} pixels", dx.view(-1, cin)[sel.to(DEV)], ref_dx, rtol=5e-6)
        assert torch.isfinite(dx).all()
        dw = torch.full((cout, 3, 3, cin), float("nan"), device=DEV)
        ops.conv3x3_wgrad_bf16(xp, dyp, dw, n, h, w, cin, cout)
        close(f"conv3 weight gradient [{tile}] vs fp64 at 16 x 9 x 16 entries", dw[co_s.to(DEV)][:, :, :, ci_s.to(DEV)], ref_dw, rtol=5e-6)
        assert torch.isfinite(dw).all()
    monkeypatch.delenv("GOALNET_BF16_TILE", raising=False)
    # the bf16-output epilogue of the same launches (what the bench shape stores): fp32 result rounded once
    if ops.conv3x3_fwd_bf16p_o16_ok(n, h, w, cin, cout):
        y16 = torch.empty(n, h, w, cout, dtype=torch.bfloat16, device=DEV)
        ops.conv3x3_fwd_bf16p_o16(xp, wtd, b.to(DEV), True, y16, n, h, w, cin, cout)
        assert torch.equal(y16, y_auto.to(torch.bfloat16))
        close("conv3 forward, bf16 output, vs fp64", y16.view(-1, cout)[sel.to(DEV)].float(), ref_y, rtol=4e-3)
    # linear layers at sizes that select the 256 tile: forward (split-K), dX (bf16 and fp32 outputs), dW — sampled rows / columns in fp64
    m, j, k = 320, 512, (1 << 18) + 264                                 # ragged in M (5 K-tiles) and K (partial last tile)
    dyl = (torch.rand(m, j, generator=g) - 0.5).to(torch.bfloat16)
    xl = (torch.rand(m, k, generator=g) - 0.5).to(torch.bfloat16)
    wl = ((torch.rand(j, k, generator=g) - 0.5) * 0.05).to(torch.bfloat16)
    bl = (torch.rand(j, generator=g) - 0.5)
    rows = torch.tensor([0, 1, 127, 128, 255, 256, 300, 319])
    cols = torch.tensor([0, 1, 255, 256, 257, 1 << 17, (1 << 18) - 1, 1 << 18, (1 << 18) + 255, (1 << 18) + 256, k - 1])
    for tile in ("auto", "128"):
        if tile == "128":
            monkeypatch.setenv("GOALNET_BF16_TILE", tile)
        dwl = ops.linear_bwd_dw_bf16(dyl.to(DEV), xl.to(DEV), torch.full((j, k), float("nan"), device=DEV))
        close(f"linear dW [{tile}] 8 rows vs fp64", dwl[rows.to(DEV)], dyl[:, rows].double().t() @ xl.double(), rtol=3e-6)
        close(f"linear dW [{tile}] 11 columns vs fp64", dwl[:, cols.to(DEV)], dyl.double().t() @ xl[:, cols].double(), rtol=3e-6)
        assert torch.isfinite(dwl).all()
        k2 = k - k % 64                                                  # the forward needs K % 64 == 0
        xf, wf = xl[:, :k2].contiguous(), wl[:, :k2].contiguous()
        yl = torch.full((m, j), float("nan"), device=DEV)
        ops.linear_fwd_bf16(xf.to(DEV), wf.to(DEV), bl.to(DEV), yl, relu=True)
        close(f"linear forward [{tile}] vs fp64", yl, F.relu(xf.double() @ wf.double().t() + bl.double()), rtol=5e-6)
        dxl = ops.linear_bwd_dx_bf16(dyl.to(DEV), wl.to(DEV), torch.full((m, k), float("nan"), device=DEV))
        close(f"linear dX [{tile}] 11 columns vs fp64", dxl[:, cols.to(DEV)], dyl.double() @ wl[:, cols].double(), rtol=3e-6)
        close(f"linear dX [{tile}] 8 rows vs fp64", dxl[rows.to(DEV)], dyl[rows].double() @ wl.double(), rtol=3e-6)
        assert torch.isfinite(dxl).all()


@pytest.mark.parametrize("tile", ["256"])
def test_bf16_gradient_outputs_equal_the_fp32_outputs_rounded_once(tile, monkeypatch):
    """The *_o16 forms of the two data-gradient GEMMs store bf16(accumulator): they must equal, bit for bit, the fp32-output
    forms rounded to nearest-even — on whole tiles and on ragged edges (rows and columns that are not multiples of 256)."""
    monkeypatch.setenv("GOALNET_BF16_TILE", tile)                        # small shapes through the 256 x 256 kernel
    g = torch.Generator().manual_seed(91)
    # linear dX: M frames x K features from J = 128
    m, j, k = 70, 128, 1032
    dy = (torch.rand(m, j, generator=g) - 0.5).to(torch.bfloat16).to(DEV)
    w = (torch.rand(j, k, generator=g) - 0.5).to(torch.bfloat16).to(DEV)
    assert ops.linear_bwd_dx_bf16_o16_ok(m, k, j)
    f32 = ops.linear_bwd_dx_bf16(dy, w, torch.empty(m, k, device=DEV))
    b16 = ops.linear_bwd_dx_bf16_o16(dy, w, torch.full((m, k), float("nan"), dtype=torch.bfloat16, device=DEV))
    assert torch.equal(b16, f32.to(torch.bfloat16))
    close("linear_bwd_dx_bf16_o16 vs fp64", b16.float(), (dy.double() @ w.double()), rtol=4e-3)      # bf16 rounding: 2^-9
    # conv data gradient form: 3x3 conv of a padded bf16 tensor, no bias / ReLU
    n, h, wd, cin, cout = 3, 9, 11, 64, 72
    x = (torch.rand(n, h, wd, cin, generator=g) - 0.5)
    wt = ((torch.rand(cout, 3, 3, cin, generator=g) - 0.5) * 0.1).to(torch.bfloat16).to(DEV)
    _, xp = _padded(x)
    assert ops.conv3x3_fwd_bf16p_o16_ok(n, h, wd, cin, cout)
    f32 = ops.conv3x3_fwd_bf16p(xp, wt, None, False, torch.empty(n, h, wd, cout, device=DEV), n, h, wd, cin, cout)
    b16 = ops.conv3x3_fwd_bf16p_o16(xp, wt, None, False, torch.full((n, h, wd, cout), float("nan"), dtype=torch.bfloat16, device=DEV), n, h, wd, cin, cout)
    assert torch.equal(b16, f32.to(torch.bfloat16))
    # forward form: bias + ReLU, then one rounding
    bias = (torch.rand(cout, generator=g) - 0.5).to(DEV)
    f32 = ops.conv3x3_fwd_bf16p(xp, wt, bias, True, torch.empty(n, h, wd, cout, device=DEV), n, h, wd, cin, cout)
    b16 = ops.conv3x3_fwd_bf16p_o16(xp, wt, bias, True, torch.full((n, h, wd, cout), float("nan"), dtype=torch.bfloat16, device=DEV), n, h, wd, cin, cout)
    assert torch.equal(b16, f32.to(torch.bfloat16)) and (b16 == 0).any() and (b16 > 0).any()
    monkeypatch.setenv("GOALNET_BF16_TILE", "128")                       # the 128 x 128 kernels have no bf16 epilogue: must refuse
    assert not ops.linear_bwd_dx_bf16_o16_ok(m, k, j) and not ops.conv3x3_fwd_bf16p_o16_ok(n, h, wd, cin, cout)
    with pytest.raises(Exception):
        ops.linear_bwd_dx_bf16_o16(dy, w, torch.empty(m, k, dtype=torch.bfloat16, device=DEV))


@pytest.mark.parametrize("n,hc,wc,c", [(3, 9, 11, 64), (2, 13, 13, 256)])
def test_batchnorm_backward_passes_with_bf16_gradient_input(n, hc, wc, c):
    """bn_bwd_reduce and the fused BN/pool/ReLU backward fed a bf16 dz must equal, bit for bit, the same kernels fed
    those values as fp32."""
    g = torch.Generator().manual_seed(92)
    hp, wp = hc - 2, wc - 2
    y = (torch.rand(n, hc, wc, c, generator=g) - 0.3).to(DEV)
    p = torch.empty(n, hp, wp, c, device=DEV)
    idx = torch.empty(n, hp, wp, c, dtype=torch.uint8, device=DEV)
    parts = torch.empty(ops.stat_parts(8 * n) * 2 * c, dtype=torch.float64, device=DEV)
    ops.pool_bnstats_fwd(y, p, idx, parts, n, hc, wc, c)
    gamma = (torch.rand(c, generator=g) + 0.5).to(DEV); beta = torch.zeros(c, device=DEV)
    st = torch.empty(4, c, device=DEV)
    ops.bn_finalize(parts, gamma, beta, None, None, 0.1, 1e-5, n * hp * wp, c, st[0], st[1], st[2], st[3])
    dz16 = (torch.rand(n, hp, wp, c, generator=g) - 0.5).to(torch.bfloat16).to(DEV)
    dz32 = dz16.float()
    npix = n * hp * wp
    out = {}
    for name, dz in (("f32", dz32), ("b16", dz16)):
        red = torch.empty(ops.stat_parts(max(npix // 64, 1)) * 2 * c, dtype=torch.float64, device=DEV)
        ops.bn_bwd_reduce(dz, p, st[0], st[1], red, npix, c)
        coef3 = torch.empty(3 * c, device=DEV); dg = torch.empty(c, device=DEV); db = torch.empty(c, device=DEV)
        ops.bn_bwd_finalize(red, gamma, st[0], st[1], npix, c, dg, db, coef3)
        buf, dyp = ops.padded_bf16_alloc(n, hc, wc, c, DEV)
        buf.zero_()
        dparts = torch.empty(ops.stat_parts(8 * n) * c, dtype=torch.float64, device=DEV)
        dy = torch.empty(n, hc, wc, c, device=DEV)
        ops.bnpool_bwd_bf16p(dz, p, idx, coef3, dy, dyp, dparts, n, hc, wc, c)
        out[name] = (red.clone(), dg, db, coef3, dy, buf.clone(), dparts)
    for a, b in zip(out["f32"], out["b16"]):
        assert torch.equal(a, b)
    assert out["b16"][4].abs().max().item() > 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("n,hc,wc,c,kind", [(3, 13, 13, 256, "relu"), (2, 11, 11, 512, "signed"), (2, 9, 40, 64, "special"),
                                            (1, 76, 74, 64, "relu"), (4, 3, 3, 32, "signed")])
def test_integer_key_pool_is_bit_identical_to_the_compare_and_select_kernel(n, hc, wc, c, kind, dtype, monkeypatch):
    """16-bit conv output -> 16-bit pooled activation: the integer-key kernel (max over (value << 16 | 8 - tap) words; windows
    holding a negative value, -0.0 or a NaN take its exact path) against v2's float compare-and-select on the same input:
    pooled values, argmax bytes and the fp64 partial rows must agree bit for bit."""
    g = torch.Generator().manual_seed(321)
    y = torch.rand(n, hc, wc, c, generator=g) - 0.4
    if kind == "relu":
        y = F.relu(y)                                   # many exact zeros: ties everywhere, the fast path only
    elif kind == "special":
        y = F.relu(y)
        flat = y.view(-1)
        pick = torch.randperm(flat.numel(), generator=g)[:600]
        flat[pick[:150]] = float("nan"); flat[pick[150:300]] = float("inf"); flat[pick[300:450]] = -0.0; flat[pick[450:]] = -float("inf")
    yb = y.to(dtype).to(DEV)
    hp, wp = hc - 2, wc - 2
    parts = ops.stat_parts(8 * n)
    out = {}
    for name, off in (("v2", "1"), ("k16", None)):
        if off:
            monkeypatch.setenv("GOALNET_POOL_K16_OFF", off)
        else:
            monkeypatch.delenv("GOALNET_POOL_K16_OFF", raising=False)
        p = torch.empty(n, hp, wp, c, dtype=dtype, device=DEV); idx = torch.empty(n, hp, wp, c, dtype=torch.uint8, device=DEV)
        st = torch.empty(parts * 2 * c, dtype=torch.float64, device=DEV)
        ops.pool_bnstats_fwd(yb, p, idx, st, n, hc, wc, c)
        out[name] = (p.view(torch.int16).clone(), idx.clone(), st.view(torch.int64).clone())
    assert torch.equal(out["v2"][1], out["k16"][1]), "argmax bytes differ"
    pv, pk = out["v2"][0], out["k16"][0]
    if kind == "special":                               # v2 re-rounds a NaN through float (canonical payload); compare NaN-ness there
        fv, fk = pv.view(dtype).float(), pk.view(dtype).float()
        assert torch.equal(torch.isnan(fv), torch.isnan(fk)) and torch.equal(pv[~torch.isnan(fv)], pk[~torch.isnan(fk)])
    else:
        assert torch.equal(pv, pk), "pooled values differ"
        assert torch.equal(out["v2"][2], out["k16"][2]), "fp64 partial rows differ"


@pytest.mark.parametrize("n,hc,wc,c", [(3, 9, 11, 64), (2, 13, 13, 256), (2, 40, 37, 32)])
def test_bf16_pooled_activation_variants_match_the_fp32_kernels_on_the_stored_values(n, hc, wc, c):
    """precision="bf16" stores the pooled activation p of blocks 2 and 3 as bf16. The p16 pool kernel must store exactly
    bf16(max), keep the argmax, and take its statistics from the stored values; every consumer fed the bf16 tensor must
    equal, bit for bit, its fp32 form fed p.float()."""
    g = torch.Generator().manual_seed(93)
    hp, wp = hc - 2, wc - 2
    y = (torch.rand(n, hc, wc, c, generator=g) - 0.3).to(DEV)
    parts = ops.stat_parts(8 * n)
    p32 = torch.empty(n, hp, wp, c, device=DEV); i32 = torch.empty(n, hp, wp, c, dtype=torch.uint8, device=DEV)
    s32 = torch.empty(parts * 2 * c, dtype=torch.float64, device=DEV)
    ops.pool_bnstats_fwd(y, p32, i32, s32, n, hc, wc, c)
    p16 = torch.empty(n, hp, wp, c, dtype=torch.bfloat16, device=DEV); i16 = torch.empty_like(i32)
    s16 = torch.empty_like(s32)
    ops.pool_bnstats_fwd(y, p16, i16, s16, n, hc, wc, c)
    assert torch.equal(p16, p32.to(torch.bfloat16)) and torch.equal(i16, i32)
    # a bf16 conv output: rounding is monotonic, so p and the statistics are identical; the argmax may move only between
    # window elements that are equal after rounding
    yb = y.to(torch.bfloat16)
    pb = torch.empty_like(p16); ib = torch.empty_like(i32); sb = torch.empty_like(s32)
    ops.pool_bnstats_fwd(yb, pb, ib, sb, n, hc, wc, c)
    assert torch.equal(pb, p16) and torch.equal(sb, s16)
    moved = ops.idx_to_nhwc(ib, n, hp, wp, c) != ops.idx_to_nhwc(i16, n, hp, wp, c)
    if moved.any():
        win = yb.float().unfold(1, 3, 1).unfold(2, 3, 1).permute(0, 1, 2, 4, 5, 3).reshape(n, hp, wp, 9, c)   # [.., tap, c]
        tb = torch.gather(win, 3, ops.idx_to_nhwc(ib, n, hp, wp, c).long().unsqueeze(3)).squeeze(3)
        assert torch.equal(tb, pb.float()), "the argmax of a bf16 conv output must point at an element equal to the maximum"
    pf = p16.float()
    tot = s16.view(parts, 2, c).sum(0).cpu()
    ref_sum, ref_sq = pf.double().sum((0, 1, 2)).cpu(), (pf.double() ** 2).sum((0, 1, 2)).cpu()
    assert torch.allclose(tot[0], ref_sum, rtol=1e-12, atol=1e-9) and torch.allclose(tot[1], ref_sq, rtol=1e-12, atol=1e-9)
    gamma = (torch.rand(c, generator=g) + 0.5).to(DEV); beta = (torch.rand(c, generator=g) - 0.5).to(DEV)
    st = torch.empty(4, c, device=DEV)
    ops.bn_finalize(s16, gamma, beta, None, None, 0.1, 1e-5, n * hp * wp, c, st[0], st[1], st[2], st[3])
    # BatchNorm apply (flat and zero-padded forms)
    a = ops.bn_apply_bf16(p16, st[2], st[3], torch.empty(n, hp, wp, c, dtype=torch.bfloat16, device=DEV), c)
    b = ops.bn_apply_bf16(pf, st[2], st[3], torch.empty(n, hp, wp, c, dtype=torch.bfloat16, device=DEV), c)
    assert torch.equal(a, b)
    ba, va = ops.padded_bf16_alloc(n, hp, wp, c, DEV); bb, vb = ops.padded_bf16_alloc(n, hp, wp, c, DEV)
    ops.to_bf16_padded(p16, st[2], st[3], va, n, hp, wp, c)
    ops.to_bf16_padded(pf, st[2], st[3], vb, n, hp, wp, c)
    assert torch.equal(ba, bb)
    # backward passes, dz fp32 and bf16
    npix = n * hp * wp
    for dzt in (torch.float32, torch.bfloat16):
        dz = (torch.rand(n, hp, wp, c, generator=g) - 0.5).to(torch.bfloat16).to(DEV).to(dzt)
        out = {}
        for name, pp in (("f32", pf), ("b16", p16)):
            red = torch.empty(ops.stat_parts(max(npix // 64, 1)) * 2 * c, dtype=torch.float64, device=DEV)
            ops.bn_bwd_reduce(dz, pp, st[0], st[1], red, npix, c)
            coef3 = torch.empty(3 * c, device=DEV); dg = torch.empty(c, device=DEV); db = torch.empty(c, device=DEV)
            ops.bn_bwd_finalize(red, gamma, st[0], st[1], npix, c, dg, db, coef3)
            buf, dyp = ops.padded_bf16_alloc(n, hc, wc, c, DEV)
            dparts = torch.empty(parts * c, dtype=torch.float64, device=DEV)
            dy = torch.empty(n, hc, wc, c, device=DEV)
            ops.bnpool_bwd_bf16p(dz, pp, i16, coef3, dy, dyp, dparts, n, hc, wc, c)
            out[name] = (red.clone(), coef3, dy, buf.clone(), dparts)
        for u, v in zip(out["f32"], out["b16"]):
            assert torch.equal(u, v)


def test_tile_walk_of_the_256_kernel_equals_one_block_per_tile_bit_for_bit():
    """The 16-bit conv forward / data gradient / linear5 dX launch ONE block per CU that walks the tiles (next tile's operands
    staged under the current epilogue); GOALNET_PERSISTENT=0 launches one block per tile. Same tiles, same K order: every output
    of scripts/probe/swap_hash.py (all six roles of gemm_bf16_256.hip, bf16 and fp16, whole and ragged shapes, 16-bit and fp32
    results) must hash identically. The switch is read once per process, hence two child processes."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for persistent in ("1", "0"):
        env = dict(os.environ, GOALNET_PERSISTENT=persistent)
        env.pop("GOALNET_LIB_PATH", None)
        r = subprocess.run([sys.executable, os.path.join(root, "scripts", "probe", "swap_hash.py")], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout)
    assert outs[0].count("\n") >= 24, outs[0]
    assert outs[0] == outs[1]


# ------------------------------------------------------------------------------------------------
# precision = "bf16x6" / "fp16x3" (csrc/split3.hip): fp32 operands as bf16 triples (six partial products per product) or as fp16 pairs of
# the power-of-two-scaled value (three partial products) on the 16-bit MFMA
def _split_host(x, parts, s=1.0):
    """the parts computed by torch on the host (round to nearest even): bf16 (hi, mid, lo) of x, or fp16 (hi, mid) of x * s"""
    dt = torch.bfloat16 if parts == 3 else torch.float16
    v = x * s
    hi = v.to(dt)
    r1 = v - hi.float()
    mid = r1.to(dt)
    return (hi, mid, (r1 - mid.float()).to(dt)) if parts == 3 else (hi, mid)


def _amax_word(x2d, rows, c, scale=None, shift=None, bnC=0):
    return ops.absmax(x2d, torch.zeros(1, dtype=torch.int32, device=DEV), rows, c, scale=scale, shift=shift, bnC=bnC)


def _scale_of(amax_word):
    """the power of two csrc/split3.hip derives from a magnitude word: the largest scaled magnitude lands in [2^14, 2^15)"""
    a = torch.tensor([int(amax_word.item())], dtype=torch.int32).view(torch.float32).item()
    import math
    return 1.0 if a == 0 else 2.0 ** (14 - math.floor(math.log2(a)))


@pytest.mark.parametrize("parts", [3, 2])
def test_split_is_exact_and_lays_the_parts_out_side_by_side(parts):
    """parts = 3: hi + mid + lo (bf16) is the fp32 value EXACTLY (3 x 8 significand bits). parts = 2: hi + mid (fp16) is the scaled value
    to 22 significant bits, the scale is the power of two that puts the tensor's largest magnitude into [2^14, 2^15), and absmax
    returns that magnitude's bit pattern. Both equal torch's own roundings bit for bit; the BatchNorm affine is applied in fp32 before
    the split (one fmaf); the padded layout keeps its zero border: (N, H+2, W+2, [hi C | mid C | ...])"""
    n, h, w, c = 2, 5, 7, 64
    dt = torch.bfloat16 if parts == 3 else torch.float16
    x = rnd(n, h, w, c, seed=300) * 3.0
    x[0, 0, 0, :8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 2.0 ** -100, 7.5, 1.0 + 2.0 ** -23, -(1.0 + 2.0 ** -12)])
    sc = rnd(c, seed=301, lo=0.5, hi=1.5)
    sh = rnd(c, seed=302)
    for affine in (False, True):
        want = (x.double() * sc.double() + sh.double()).float() if affine else x      # the correctly rounded fma
        am = scale = None
        if parts == 2:
            am = _amax_word(x.to(DEV), n * h * w, c, scale=sc.to(DEV) if affine else None, shift=sh.to(DEV) if affine else None, bnC=c if affine else 0)
            assert int(am.item()) == int(want.abs().max().view(torch.int32).item()), "absmax is not the bit pattern of max |x|"
            scale = _scale_of(am)
            assert 2 ** 14 <= want.abs().max().item() * scale < 2 ** 15
        _, view = ops.padded_bf16_alloc(n, h, w, parts * c, DEV, dtype=dt)
        ops.split_padded(parts, x.to(DEV), sc.to(DEV) if affine else None, sh.to(DEV) if affine else None, view, n, h, w, c, amax=am)
        got = view[: n * (h + 2) * (w + 2) * parts * c].view(n, h + 2, w + 2, parts * c).cpu()
        inner = got[:, 1:-1, 1:-1, :]
        host = _split_host(want, parts, 1.0 if parts == 3 else scale)
        for k, part in enumerate(host):
            assert torch.equal(inner[..., k * c:(k + 1) * c], part), f"part {k} differs from torch's rounding"
        total = sum(inner[..., k * c:(k + 1) * c].double() for k in range(parts))
        if parts == 3:
            assert torch.equal(total, want.double()), "hi + mid + lo is not the fp32 value"
        else:
            err = (total / scale - want.double()).abs()
            assert (err <= want.double().abs() * 2.0 ** -21 + 2.0 ** -25 / scale).all(), "hi + mid is not the scaled value to 22 bits"
        border = got.clone()
        border[:, 1:-1, 1:-1, :] = 0
        assert not border.any(), "the zero border of the padded layout was written"
    wt = rnd(6 * 9, 64, seed=303) * 0.1
    am = _amax_word(wt.to(DEV), 6 * 9, 64) if parts == 2 else None
    w3 = ops.split_rows(parts, wt.to(DEV).view(-1), torch.empty(6 * 9 * parts * 64, dtype=dt, device=DEV), 6 * 9, 64, amax=am).cpu().view(6 * 9, parts * 64)
    for k, part in enumerate(_split_host(wt, parts, 1.0 if parts == 3 else _scale_of(am))):
        assert torch.equal(w3[:, k * 64:(k + 1) * 64], part)


@pytest.mark.parametrize("parts", [3, 2])
@pytest.mark.parametrize("n,h,w,cin,cout,bias,relu", [(2, 9, 11, 64, 256, True, True), (3, 24, 24, 256, 512, True, True),
                                                      (1, 13, 40, 128, 72, False, False), (5, 37, 29, 64, 264, True, False),
                                                      (3, 21, 19, 256, 64, False, False)])      # 64 output channels: the 128 x 64 tile
def test_conv3x3_split_forward_data_gradient_and_weight_gradient_vs_fp64(n, h, w, cin, cout, bias, relu, parts):
    """goalnet_conv3x3_fwd_split / goalnet_conv3x3_wgrad_split against fp64 on the UNROUNDED fp32 operands, whole and ragged tiles, with
    the fp32 engine's kernels beside them: the split-operand results must be fp32-grade — within 6e-6 of the output scale (measured
    5e-7 .. 2.2e-6 for bf16x6; the fp32 MFMA kernels 2.5e-7 .. 4e-7; a single bf16 product would be 4e-3). The gradient operand is
    given a wide dynamic range (x 1e-7 .. 1e-3 per channel) for the scaled fp16 form."""
    dt = torch.bfloat16 if parts == 3 else torch.float16
    x = torch.relu(rnd(n, h, w, cin, seed=310) * 2.0)
    sc = rnd(cin, seed=311, lo=0.5, hi=1.5)
    sh = rnd(cin, seed=312, lo=-0.5, hi=0.5)
    wt = rnd(cout, 3, 3, cin, seed=313) * 0.05
    b = rnd(cout, seed=314) if bias else None
    dy = rnd(n, h, w, cout, seed=315) * (10.0 ** (-7 + 4 * rnd(cout, seed=316, lo=0.0, hi=1.0)))
    xd, scd, shd, wd, dyd = x.to(DEV), sc.to(DEV), sh.to(DEV), wt.to(DEV).view(-1), dy.to(DEV)
    ax = _amax_word(xd, n * h * w, cin, scale=scd, shift=shd, bnC=cin) if parts == 2 else None
    aw = _amax_word(wd, cout * 9, cin) if parts == 2 else None
    osc = (lambda a, b_: ops.split_scales(a, b_)) if parts == 2 else (lambda a, b_: None)
    _, xps = ops.padded_bf16_alloc(n, h, w, parts * cin, DEV, dtype=dt)
    ops.split_padded(parts, xd, scd, shd, xps, n, h, w, cin, amax=ax)
    wsp = ops.split_rows(parts, wd, torch.empty(cout * 9 * parts * cin, dtype=dt, device=DEV), cout * 9, cin, amax=aw)
    y = torch.full((n, h, w, cout), float("nan"), device=DEV)
    ops.conv3x3_fwd_split(parts, xps, wsp, None if b is None else b.to(DEV), relu, y, n, h, w, cin, cout, oscale=osc(ax, aw))
    xh = nchw(x.double() * sc.double() + sh.double())
    ref = F.conv2d(xh, nchw(wt.double()), None if b is None else b.double(), padding=1)
    ref = nhwc(F.relu(ref) if relu else ref)
    close(f"conv3x3_fwd_split[{parts}] vs fp64", y, ref, rtol=6e-6)
    y32 = torch.empty(n, h, w, cout, device=DEV)
    ops.conv3x3_fwd(xd, scd, shd, wd, None if b is None else b.to(DEV), relu, y32, n, h, w, cin, cout)
    close("conv3x3_fwd (fp32 MFMA) vs fp64, same operands", y32, ref)
    # weight gradient
    ady = _amax_word(dyd, n * h * w, cout) if parts == 2 else None
    _, dyps = ops.padded_bf16_alloc(n, h, w, parts * cout, DEV, dtype=dt)
    ops.split_padded(parts, dyd, None, None, dyps, n, h, w, cout, amax=ady)
    dw = torch.full((cout * 9 * cin,), float("nan"), device=DEV)
    ops.conv3x3_wgrad_split(parts, xps, dyps, dw, n, h, w, cin, cout, oscale=osc(ady, ax))
    refdw = torch.nn.grad.conv2d_weight(xh, (cout, cin, 3, 3), nchw(dy.double()), padding=1).permute(0, 2, 3, 1)
    close(f"conv3x3_wgrad_split[{parts}] vs fp64", dw.view(cout, 3, 3, cin), refdw, rtol=6e-6)
    # data gradient: the same forward call on the split gradient and the split flipped weights
    if cout % 64 == 0:
        wflip = torch.empty(cout * 9 * cin, device=DEV)
        ops.conv3x3_weight_flip(wd, wflip, cout, cin)
        awf = _amax_word(wflip, cin * 9, cout) if parts == 2 else None
        wfs = ops.split_rows(parts, wflip, torch.empty(cin * 9 * parts * cout, dtype=dt, device=DEV), cin * 9, cout, amax=awf)
        dx = torch.full((n, h, w, cin), float("nan"), device=DEV)
        ops.conv3x3_fwd_split(parts, dyps, wfs, None, False, dx, n, h, w, cout, cin, oscale=osc(ady, awf))
        refdx = nhwc(torch.nn.grad.conv2d_input((n, cin, h, w), nchw(wt.double()), nchw(dy.double()), padding=1))
        close(f"conv3x3 data gradient (split[{parts}]) vs fp64", dx, refdx, rtol=6e-6)


@pytest.mark.parametrize("parts", [3, 2])
def test_linear5_on_split_operands_forward_dx_dw_vs_fp64(parts):
    """goalnet_linear_fwd_split / _bwd_dx_split / _bwd_dw_split (linear5's three contractions under precision="bf16x6" / "fp16x3") against
    fp64 on the unrounded fp32 operands: ragged in M (320 = 256 + 64) and K (a partial last 256-column tile), BatchNorm affine applied on
    the way into the split, bias + ReLU + dropout mask + saved multiplier in the forward's reduction epilogue. fp32-grade: 6e-6 of the scale."""
    dt = torch.bfloat16 if parts == 3 else torch.float16
    m, k, j, bnc = 320, 66048 + 64, 256, 64
    assert ops.linear_split_ok(parts, m, k, j) and not ops.linear_split_ok(parts, 10, k, j)
    x = rnd(m, k, seed=320)
    sc = rnd(bnc, seed=321, lo=0.5, hi=1.5)
    sh = rnd(bnc, seed=322, lo=-0.5, hi=0.5)
    w = rnd(j, k, seed=323) * 0.02
    b = rnd(j, seed=324)
    dy = rnd(m, j + 128, seed=325)[:, 128:] * 1e-5                       # a column slice of a wider buffer, as dz5 is
    mask = (rnd(m, j, seed=326) > 0).float() * 2.0
    xd, wd = x.to(DEV), w.to(DEV)
    osc = (lambda a, b_: ops.split_scales(a, b_)) if parts == 2 else (lambda a, b_: None)
    ax = _amax_word(xd, m, k, scale=sc.to(DEV), shift=sh.to(DEV), bnC=bnc) if parts == 2 else None
    aw = _amax_word(wd, j, k) if parts == 2 else None
    xs = ops.split_rows(parts, xd, torch.empty(m * parts * k, dtype=dt, device=DEV), m, k, scale=sc.to(DEV), shift=sh.to(DEV), bnC=bnc, amax=ax)
    wsp = ops.split_rows(parts, wd, torch.empty(j * parts * k, dtype=dt, device=DEV), j, k, amax=aw)
    y = torch.full((m, j), float("nan"), device=DEV)
    mult = torch.full((m, j), float("nan"), device=DEV)
    ops.linear_fwd_split(parts, xs, wsp, b.to(DEV), y, m, k, j, relu=True, dropmask=mask.to(DEV), mult_out=mult, oscale=osc(ax, aw))
    xh = x.double() * sc.double().repeat(k // bnc) + sh.double().repeat(k // bnc)
    z = xh @ w.double().t() + b.double()
    close(f"linear_fwd_split[{parts}] vs fp64", y, F.relu(z) * mask.double(), rtol=6e-6)
    assert torch.isfinite(mult).all()
    dyd = torch.zeros(m, j + 128, device=DEV)
    dyd[:, 128:] = dy.to(DEV)
    ady = _amax_word(dyd[:, 128:], m, j) if parts == 2 else None
    dys = ops.split_rows(parts, dyd[:, 128:], torch.empty(m * parts * j, dtype=dt, device=DEV), m, j, amax=ady)
    dx = torch.full((m, k), float("nan"), device=DEV)
    ops.linear_bwd_dx_split(parts, dys, wsp, dx, m, k, j, oscale=osc(ady, aw))
    close(f"linear_bwd_dx_split[{parts}] vs fp64", dx, dy.double() @ w.double(), rtol=6e-6)
    dw = torch.full((j, k), float("nan"), device=DEV)
    ops.linear_bwd_dw_split(parts, dys, xs, dw, m, k, j, oscale=osc(ady, ax))
    close(f"linear_bwd_dw_split[{parts}] vs fp64", dw, dy.double().t() @ xh, rtol=6e-6)


@pytest.mark.parametrize("magnitude", [0.0, 1e-30, 1e-12, 1.0, 3e4, 1e30])
def test_fp16x3_scaling_keeps_any_fp32_magnitude_in_range(magnitude):
    """parts = 2: whatever the magnitude of an operand tensor (zero, 1e-30 .. 1e30), its power-of-two scale puts the largest value into
    [2^14, 2^15) — inside binary16 — and the GEMM's epilogue undoes both scales exactly: the convolution of x * magnitude equals magnitude
    times the convolution of x to fp32-grade accuracy (6e-6 of the output scale), and an all-zero operand gives exact zeros."""
    n, h, w, cin, cout = 2, 9, 11, 64, 256
    x = rnd(n, h, w, cin, seed=400) * magnitude
    wt = rnd(cout, 3, 3, cin, seed=401) * 0.05
    xd, wd = x.to(DEV), wt.to(DEV).view(-1)
    ax, aw = _amax_word(xd, n * h * w, cin), _amax_word(wd, cout * 9, cin)
    if magnitude > 0:
        assert 2 ** 14 <= x.abs().max().item() * _scale_of(ax) < 2 ** 15
    else:
        assert int(ax.item()) == 0
    _, xps = ops.padded_bf16_alloc(n, h, w, 2 * cin, DEV, dtype=torch.float16)
    ops.split_padded(2, xd, None, None, xps, n, h, w, cin, amax=ax)
    assert torch.isfinite(xps.float()).all()
    wsp = ops.split_rows(2, wd, torch.empty(cout * 9 * 2 * cin, dtype=torch.float16, device=DEV), cout * 9, cin, amax=aw)
    y = torch.full((n, h, w, cout), float("nan"), device=DEV)
    ops.conv3x3_fwd_split(2, xps, wsp, None, False, y, n, h, w, cin, cout, oscale=ops.split_scales(ax, aw))
    ref = nhwc(F.conv2d(nchw(x.double()), nchw(wt.double()), None, padding=1))
    if magnitude == 0.0:
        assert not y.any()
    else:
        close(f"conv of x * {magnitude:g} on scaled fp16 pairs vs fp64", y, ref, rtol=6e-6)
