"""GPU: the entry points round 3 added for the reference's operating point — sub-batches of <= 16 frames of 40 x 40
(/root/reference/main.py:44, 177-196) — each against the fp64 CPU oracle ops (torch CPU) on seeded inputs, through the C ABI:

  goalnet_mlp_fwd / goalnet_mlp_bwd        the fusion MLP + Sigmoid + 4y+1 (+ broadcast MSE) and its backward, one launch each
                                           (utils.py:242-258, 269-270; main.py:191-192), every row class (1 .. 16 rows), audio on / off,
                                           dropout masks on / off
  goalnet_pool_bn_fwd_small                MaxPool2d(3, 1) + train-mode BatchNorm statistics + finalise in one launch (utils.py:153-154)
  goalnet_bn_bwd_reduce_small, goalnet_bnpool_bwd_small   their backward
  goalnet_conv1d_bwd_small                 a Conv1d layer's backward with its ReLU backward folded in (utils.py:203-207)
  goalnet_conv3x3_weight_flip2, goalnet_partials_sum2, goalnet_rows_scatter_tick   batched forms: bit-identical to the single ones

Tolerances as in tests/test_gpu_ops.py (fp32 kernels vs fp64: ~2e-6 of the tensor's scale); argmax positions bit-exact.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from cvml_goalnet_amd import ops  # noqa: E402
from test_gpu_ops import DEV, close, nchw, nhwc, rnd  # noqa: E402

WIDTHS = (512, 512, 256, 128)


@pytest.mark.parametrize("n,k0,use_masks", [(1, 640, True), (4, 640, False), (5, 512, True), (8, 640, True), (9, 640, True), (12, 512, False),
                                            (13, 640, True), (16, 640, True), (10, 640, True)])
def test_fused_mlp_forward_and_backward_vs_fp64(n, k0, use_masks):
    g = torch.Generator().manual_seed(100 + n)
    dims = [k0] + list(WIDTHS) + [1]
    ws = [((torch.rand(dims[l + 1], dims[l], generator=g, dtype=torch.float64) - 0.5) * 2 / dims[l] ** 0.5) for l in range(5)]
    bs = [((torch.rand(dims[l + 1], generator=g, dtype=torch.float64) - 0.5) * 0.2) for l in range(5)]
    wide = torch.rand(n, k0 + 64, generator=g, dtype=torch.float64) - 0.3          # cat is a column slice of a wider buffer (row stride)
    masks = [((torch.rand(n, w, generator=g) >= 0.2).double() * 1.25) if use_masks else None for w in WIDTHS]
    labels = torch.randint(1, 6, (n,), generator=g).double()
    # ---- oracle (fp64 autograd)
    wd = [w.clone().requires_grad_(True) for w in ws]
    bd = [b.clone().requires_grad_(True) for b in bs]
    x0 = wide[:, 32:32 + k0].clone().requires_grad_(True)
    x, hs = x0, []
    for l in range(4):
        x = F.relu(F.linear(x, wd[l], bd[l]))
        if masks[l] is not None:
            x = x * masks[l]
        hs.append(x)
    z = F.linear(x, wd[4], bd[4]).view(-1)
    out = 4 * torch.sigmoid(z) + 1
    d = out.view(n, 1) - labels.view(1, n)
    loss = (d * d).mean()                                                          # nn.MSELoss on (n,1) x (n,): (n,n) broadcast, main.py:191
    loss.backward()
    # ---- device
    f = lambda t: t.float().to(DEV)
    wg, bg = [f(w) for w in ws], [f(b) for b in bs]
    cat = f(wide)[:, 32:32 + k0]
    mg = [None if m is None else f(m) for m in masks]
    hg = [torch.empty(n, w, device=DEV) for w in WIDTHS]
    mult = [torch.empty(n, w, device=DEV) for w in WIDTHS]
    logit, og = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    lossg, dout = torch.empty(1, device=DEV), torch.empty(n, device=DEV)
    ops.mlp_fwd(cat, wg, bg, mg, hg, mult, logit, og, f(labels), lossg, dout)
    torch.cuda.synchronize()
    assert not ops.mlp_sync_error(torch.device(DEV), n, k0)
    for l in range(4):
        close(f"mlp.h{l + 1}", hg[l], hs[l], rtol=3e-6)
        want_mult = (hs[l] != 0).double() * (masks[l] if masks[l] is not None else 1.0)
        # a pre-activation within rounding of zero may gate differently: compare where the oracle's value is clearly non-zero or zero by mask
        assert ((mult[l].cpu().double() - want_mult).abs() > 1e-6).sum().item() <= 2, f"saved multipliers of layer {l}"
    close("mlp.logit", logit, z, rtol=3e-6)
    close("mlp.out", og, out, rtol=3e-6)
    close("mlp.loss", lossg, loss.view(1), rtol=3e-6, atol=2e-7)                # (p - y)^2 of an fp32 p: the rounding of p times 2 |p - y|
    close("mlp.dout", dout, (2.0 / n) * (out.detach() - labels.mean()), rtol=3e-6, atol=2e-7)      # dL/dp_i = 2/n (p_i - mean(y))
    # backward from the oracle's dL/dout, through the device's saved tensors
    mcat = torch.ones(n, k0 + 64, device=DEV)[:, 32:32 + k0]                       # no gate in front of `cat` in this test
    dws = [torch.empty_like(w) for w in wg]
    dbs = [torch.empty_like(b) for b in bg]
    dcat = torch.empty(n, k0, device=DEV)
    db5 = torch.empty(512, device=DEV)
    ops.mlp_bwd(dout, og, [cat] + hg, [mcat] + mult, wg, dws, dbs, dcat, db5, k0 - 512)
    torch.cuda.synchronize()
    assert not ops.mlp_sync_error(torch.device(DEV), n, k0)
    for l in range(5):
        close(f"mlp.dw{l}", dws[l], wd[l].grad, rtol=1e-5)
        close(f"mlp.db{l}", dbs[l], bd[l].grad, rtol=1e-5, atol=1e-9)
    close("mlp.dcat", dcat, x0.grad, rtol=1e-5)
    close("mlp.db5 (column sums of dcat[:, voff:])", db5, x0.grad[:, k0 - 512:].sum(0), rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("n,hc,wc,c", [(10, 15, 15, 64), (10, 13, 13, 256), (10, 11, 11, 512), (1, 11, 11, 512), (16, 15, 13, 64), (3, 3, 5, 256),
                                       (2, 40, 37, 64)])
def test_small_pool_batchnorm_forward_and_backward_vs_fp64(n, hc, wc, c):
    z = rnd(n, hc, wc, c, seed=16)
    y = F.relu(z)
    gamma = rnd(c, seed=17, lo=0.5, hi=1.5)
    beta = rnd(c, seed=18, lo=-0.5, hi=0.5)
    rm0 = rnd(c, seed=19)
    rv0 = rnd(c, seed=20, lo=0.5, hi=2.0)
    zd = nchw(z.double()).requires_grad_(True)
    pd, pidx = F.max_pool2d(F.relu(zd), 3, 1, 0, return_indices=True)
    rm, rv = rm0.double().clone(), rv0.double().clone()
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    od = F.batch_norm(pd, rm, rv, gd, bd, training=True, momentum=0.1, eps=1e-5)
    G = rnd(*od.shape, seed=21).double()
    (od * G).sum().backward()
    hp, wp = hc - 2, wc - 2
    p = torch.empty(n, hp, wp, c, device=DEV)
    idx = torch.empty(n, hp, wp, c, dtype=torch.uint8, device=DEV)
    st = torch.empty(4, c, device=DEV)
    rmg, rvg = rm0.to(DEV), rv0.to(DEV)
    ops.pool_bn_fwd_small(y.to(DEV), p, idx, gamma.to(DEV), beta.to(DEV), rmg, rvg, 0.1, 1e-5, st, n, hc, wc, c)
    close("maxpool", p, nhwc(pd), rtol=0.0)
    ih, iw = pidx // wc, pidx % wc
    tap = ((ih - torch.arange(hp).view(1, 1, hp, 1)) * 3 + (iw - torch.arange(wp).view(1, 1, 1, wp))).to(torch.uint8)
    assert torch.equal(ops.idx_to_nhwc(idx, n, hp, wp, c).cpu(), nhwc(tap)), "argmax positions differ from ATen's"
    close("bn.mean", st[0], pd.mean(dim=(0, 2, 3)), rtol=1e-6)
    close("bn.invstd", st[1], 1.0 / torch.sqrt(pd.var(dim=(0, 2, 3), unbiased=False) + 1e-5), rtol=1e-6)
    close("bn.running_mean", rmg, rm, rtol=1e-6)
    close("bn.running_var", rvg, rv, rtol=1e-6)
    close("bn.apply(scale,shift)", p * st[2] + st[3], nhwc(od), rtol=2e-6)
    # the same launch without an argmax output (the no-grad forward)
    p2, st2 = torch.empty_like(p), torch.empty_like(st)
    ops.pool_bn_fwd_small(y.to(DEV), p2, None, gamma.to(DEV), beta.to(DEV), rm0.to(DEV), rv0.to(DEV), 0.1, 1e-5, st2, n, hc, wc, c)
    assert torch.equal(p2, p) and torch.equal(st2, st)
    # ---- backward
    dbn = nhwc(G.float()).to(DEV)
    coef3, dgamma, dbeta = torch.empty(3 * c, device=DEV), torch.empty(c, device=DEV), torch.empty(c, device=DEV)
    ops.bn_bwd_reduce_small(dbn, p, st[0], st[1], gamma.to(DEV), dgamma, dbeta, coef3, n, hc, wc, c)
    dy, dbias = torch.empty(n, hc, wc, c, device=DEV), torch.empty(c, device=DEV)
    ops.bnpool_bwd_small(dbn, p, idx, coef3, dy, dbias, n, hc, wc, c)
    close("bn.dgamma", dgamma, gd.grad, rtol=5e-6)
    close("bn.dbeta", dbeta, bd.grad, rtol=5e-6)
    close("block.dz (bn+pool+relu bwd)", dy, nhwc(zd.grad), rtol=1e-5)
    close("block.dbias", dbias, zd.grad.sum(dim=(0, 2, 3)), rtol=0.0, atol=3e-6 * zd.grad.abs().max().item() * (n * hc * wc) ** 0.5)
    # the ticket counters are back at zero: a second call gives the same bits
    dy2, dbias2 = torch.empty_like(dy), torch.empty_like(dbias)
    ops.bnpool_bwd_small(dbn, p, idx, coef3, dy2, dbias2, n, hc, wc, c)
    assert torch.equal(dy2, dy) and torch.equal(dbias2, dbias)


@pytest.mark.parametrize("n,bins", [(1, 30), (10, 30), (16, 30), (33, 17), (63, 21)])
def test_conv1d_backward_in_one_launch_equals_the_multi_launch_form(n, bins):
    x = rnd(n, 30, bins, seed=35, lo=-50, hi=50).to(DEV)
    w1 = rnd(64, 30, 3, seed=36, lo=-0.1, hi=0.1).to(DEV); b1 = rnd(64, seed=37).to(DEV)
    w2 = rnd(128, 64, 3, seed=38, lo=-0.07, hi=0.07).to(DEV); b2 = rnd(128, seed=39).to(DEV)
    l1 = (bins - 1) // 2 + 1
    l2 = (l1 - 1) // 2 + 1
    a1 = torch.empty(n, 64, l1, device=DEV); a2 = torch.empty(n, 128, l2, device=DEV)
    ops.conv1d_fwd(x, w1, b1, a1, True, n, 30, bins, 64)
    ops.conv1d_fwd(a1, w2, b2, a2, True, n, 64, l1, 128)
    G = rnd(n, 128, l2, seed=40).to(DEV)
    # multi-launch form: relu_bwd, conv1d_bwd (dx, dw, db), relu_bwd, conv1d_bwd
    dz2 = ops.relu_bwd(G, a2, torch.empty_like(a2))
    da1 = torch.empty_like(a1); dw2 = torch.empty(128, 64, 3, device=DEV); db2 = torch.empty(128, device=DEV)
    ops.conv1d_bwd(a1, dz2, w2, da1, dw2, db2, n, 64, l1, 128)
    ops.relu_bwd(da1, a1, da1)
    dw1 = torch.empty(64, 30, 3, device=DEV); db1 = torch.empty(64, device=DEV)
    ops.conv1d_bwd(x, da1, w1, None, dw1, db1, n, 30, bins, 64)
    # one launch per layer, ReLU backward folded into the dz load
    ea1 = torch.empty_like(a1); ew2 = torch.empty_like(dw2); eb2 = torch.empty_like(db2)
    ops.conv1d_bwd_small(a1, G, a2, w2, ea1, ew2, eb2, n, 64, l1, 128)
    ew1 = torch.empty_like(dw1); eb1 = torch.empty_like(db1)
    ops.conv1d_bwd_small(x, ea1, a1, w1, None, ew1, eb1, n, 30, bins, 64)
    assert torch.equal(ops.relu_bwd(ea1, a1, torch.empty_like(ea1)), da1), "data gradient (same 16-lane sums)"
    for name, got, want in (("dw2", ew2, dw2), ("db2", eb2, db2), ("dw1", ew1, dw1), ("db1", eb1, db1)):
        close("conv1d_bwd_small." + name, got, want, rtol=2e-6)


def test_batched_launches_equal_the_single_ones_bit_for_bit():
    # weight flips
    w3 = rnd(512 * 9 * 256, seed=1).to(DEV); w2 = rnd(256 * 9 * 64, seed=2).to(DEV)
    a3, a2 = torch.empty_like(w3), torch.empty_like(w2)
    ops.conv3x3_weight_flip(w3, a3, 512, 256); ops.conv3x3_weight_flip(w2, a2, 256, 64)
    b3, b2 = torch.empty_like(w3), torch.empty_like(w2)
    ops.conv3x3_weight_flip2(w3, b3, 512, 256, w2, b2, 256, 64)
    assert torch.equal(a3, b3) and torch.equal(a2, b2)
    # partial-row sums
    pa = rnd(80, 512, seed=3).double().to(DEV); pb = rnd(37, 256, seed=4).double().to(DEV)
    oa, ob = torch.empty(512, device=DEV), torch.empty(256, device=DEV)
    ops.partials_sum(pa.view(-1), 80, 512, 512, oa); ops.partials_sum(pb.view(-1), 37, 256, 256, ob)
    qa, qb = torch.empty(512, device=DEV), torch.empty(256, device=DEV)
    ops.partials_sum2(pa.view(-1), 512, qa, pb.view(-1), 256, qb)
    assert torch.equal(oa, qa) and torch.equal(ob, qb)
    # scatter + tick == rows_copy_batch + counters_add4 (and the guarded form)
    for guarded in (False, True):
        ctr_a = torch.tensor([3, 5, 20, 2], dtype=torch.int64, device=DEV); ctr_b = ctr_a.clone()
        bad_a = torch.tensor([4 if guarded else 0], dtype=torch.int64, device=DEV); bad_b = bad_a.clone()
        table_a = torch.zeros(64, device=DEV); table_b = torch.zeros(64, device=DEV)
        losses_a = torch.zeros(8, device=DEV); losses_b = torch.zeros(8, device=DEV)
        pred = rnd(10, seed=5).to(DEV); loss = rnd(1, seed=6).to(DEV)
        ops.rows_copy_batch([(table_a, pred, 10, ctr_a[2], 0, False), (losses_a, loss, 1, ctr_a[3], 0, False)])
        if guarded:
            ops.counters_add4_guarded(ctr_a, 1, 1, 10, 1, bad_a[0])
        else:
            ops.counters_add4(ctr_a, 1, 1, 10, 1)
        ops.rows_scatter_tick([(table_b, pred, 10, ctr_b[2], 0, False), (losses_b, loss, 1, ctr_b[3], 0, False)], ctr_b, 1, 1, 10, 1,
                              bad_step=bad_b[0] if guarded else None)
        assert torch.equal(table_a, table_b) and torch.equal(losses_a, losses_b) and torch.equal(ctr_a, ctr_b) and torch.equal(bad_a, bad_b)
        assert ctr_b.tolist() == ([3, 6, 30, 3] if guarded else [4, 6, 30, 3]) and table_b[20:30].ne(0).all()
