"""GPU: precision="fp16" — the 16-bit engine with IEEE binary16 storage instead of bfloat16 (BASELINE.json config 5 names fp16
MFMA; an EXTENSION like the bf16 mode: the reference is fp32 throughout). Same kernels, layouts and entry
points as bf16 (the C ABI's `f16` flag selects the conversions and the MFMA opcode v_mfma_f32_32x32x16_f16).

Kernel level: products of fp16 values are exact in fp32 and accumulation is fp32, so against fp64 on the SAME fp16-rounded
operands the GEMMs are as tight as the fp32 ones (5e-6 of the tensor's max). Conversions are bit-exact against
`tensor.to(torch.float16)` (round to nearest even). Model level: pre-sigmoid logits vs the fp32 CPU oracle — the north
star's 1e-3, and the 11-bit significand should leave a wide margin (asserted: 2.5e-4); weight gradients (loss-scaled in
flight, compared unscaled) relative L2 <= 0.05 under the device's max-pool routing."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from cvml_goalnet_amd import AVM, ops, synth  # noqa: E402
from oracle import avm_ref  # noqa: E402
from test_gpu_avm import _is_reduction_grad, hip_taps  # noqa: E402
from test_gpu_ops import close, nchw, nhwc, rnd  # noqa: E402

DEV = "cuda:0"
H = torch.float16


def _padded(x, sc=None, sh=None):
    n, h, w, c = x.shape
    buf, view = ops.padded_bf16_alloc(n, h, w, c, DEV, dtype=H)
    ops.to_bf16_padded(x.to(DEV), None if sc is None else sc.to(DEV), None if sh is None else sh.to(DEV), view, n, h, w, c)
    return buf, view


def test_fp16_conversions_bit_exact_and_subnormal_operands_survive_the_mfma():
    x = rnd(3, 5, 7, 64, seed=150, lo=-3, hi=3)
    x.view(-1)[:8] = torch.tensor([1e-7, -3e-6, 6.0e-5, 6.2e-5, 65504.0, 65520.0, 1e6, -1e-9])     # subnormals, the largest normal, overflow
    y = torch.empty(x.shape, dtype=H, device=DEV)
    ops.cast_bf16(x.to(DEV), y)
    assert torch.equal(y.cpu(), x.to(H))                                   # round to nearest even, inf on overflow, as torch
    back = torch.empty(x.shape, device=DEV)
    ops.cast_f32(y, back)
    assert torch.equal(back.cpu(), x.to(H).float())
    sc = rnd(64, seed=151, lo=0.5, hi=1.5); sh = rnd(64, seed=152)
    ops.bn_apply_bf16(x.to(DEV), sc.to(DEV), sh.to(DEV), y, 64)
    assert torch.equal(y.cpu(), torch.addcmul(sh, x, sc).to(H))            # one fp32 fma, one rounding
    buf, view = _padded(x[..., :64], sc, sh)
    n, h, w, c = x.shape
    got = view[: n * (h + 2) * (w + 2) * c].view(n, h + 2, w + 2, c).cpu()
    want = torch.zeros(n, h + 2, w + 2, c, dtype=H)
    want[:, 1:-1, 1:-1, :] = torch.addcmul(sh, x, sc).to(H)
    assert torch.equal(got, want)
    # fp16 subnormals as MFMA operands: weights of 2^-20 (a subnormal of binary16) must contribute, not flush to zero
    m, k, j = 64, 256, 64
    a = torch.ones(m, k, dtype=H)
    wsub = torch.full((j, k), 2.0 ** -20, dtype=H)
    out = torch.full((m, j), float("nan"), device=DEV)
    ops.linear_fwd_bf16(a.to(DEV), wsub.to(DEV), torch.zeros(j, device=DEV), out)
    assert torch.equal(out.cpu(), torch.full((m, j), k * 2.0 ** -20)), "fp16 subnormal operands were flushed by the MFMA"


@pytest.mark.parametrize("tile", ["128", "256"])
@pytest.mark.parametrize("n,h,w,cin,cout,bias,relu", [(3, 13, 13, 64, 256, True, True), (16, 11, 11, 512, 256, False, False),
                                                     (9, 40, 36, 64, 320, True, True), (3, 7, 7, 64, 260, False, False),
                                                     (5, 13, 13, 256, 64, False, False)])          # 128 x 64 tile
def test_conv3x3_fp16_forward_and_data_gradient(n, h, w, cin, cout, bias, relu, tile, monkeypatch):
    monkeypatch.setenv("GOALNET_BF16_TILE", tile)
    x = rnd(n, h, w, cin, seed=163)
    wt = rnd(cout, 3, 3, cin, seed=164, lo=-0.05, hi=0.05).to(H)
    b = rnd(cout, seed=165) if bias else None
    _, xp = _padded(x)
    ref = F.conv2d(nchw(x.to(H).double()), wt.double().permute(0, 3, 1, 2), None if b is None else b.double(), padding=1)
    if relu:
        ref = F.relu(ref)
    y = torch.full((n, h, w, cout), float("nan"), device=DEV)
    ops.conv3x3_fwd_bf16p(xp, wt.to(DEV), None if b is None else b.to(DEV), relu, y, n, h, w, cin, cout)
    close(f"conv3x3 fp16 [{tile}] {n}x{h}x{w}x{cin}->{cout}", y, nhwc(ref), rtol=5e-6)
    if ops.conv3x3_fwd_bf16p_o16_ok(n, h, w, cin, cout) and cout % 8 == 0:
        y16 = torch.empty(n, h, w, cout, dtype=H, device=DEV)
        ops.conv3x3_fwd_bf16p_o16(xp, wt.to(DEV), None if b is None else b.to(DEV), relu, y16, n, h, w, cin, cout)
        assert torch.equal(y16, y.to(H)), "the fp16-output epilogue must store the fp32 result rounded once"


@pytest.mark.parametrize("tile", ["128", "256"])
def test_conv3x3_fp16_weight_gradient_and_linear_layers(tile, monkeypatch):
    monkeypatch.setenv("GOALNET_BF16_TILE", tile)
    n, h, w, cin, cout = 5, 9, 6, 64, 128
    x = rnd(n, h, w, cin, seed=166); dy = rnd(n, h, w, cout, seed=167)
    _, xp = _padded(x); _, dyp = _padded(dy)
    ref = torch.nn.grad.conv2d_weight(nchw(x.to(H).double()), (cout, cin, 3, 3), nchw(dy.to(H).double()), padding=1)
    dw = torch.full((cout, 3, 3, cin), float("nan"), device=DEV)
    ops.conv3x3_wgrad_bf16(xp, dyp, dw, n, h, w, cin, cout)
    close(f"conv3x3 wgrad fp16 [{tile}]", dw, ref.permute(0, 2, 3, 1), rtol=5e-6)
    for m, k, j in ((37, 640, 512), (300, 4160, 320)):
        xl = rnd(m, k, seed=168).to(H); wl = rnd(j, k, seed=169, lo=-0.05, hi=0.05).to(H); bl = rnd(j, seed=170)
        dyl = rnd(m, j, seed=171).to(H)
        yl = torch.full((m, j), float("nan"), device=DEV)
        ops.linear_fwd_bf16(xl.to(DEV), wl.to(DEV), bl.to(DEV), yl, relu=True)
        close(f"linear fwd fp16 [{tile}] {m}x{k}->{j}", yl, F.relu(xl.double() @ wl.double().t() + bl.double()), rtol=3e-6)
        dxl = ops.linear_bwd_dx_bf16(dyl.to(DEV), wl.to(DEV), torch.full((m, k), float("nan"), device=DEV))
        close(f"linear dX fp16 [{tile}]", dxl, dyl.double() @ wl.double(), rtol=3e-6)
        dwl = ops.linear_bwd_dw_bf16(dyl.to(DEV), xl.to(DEV), torch.full((j, k), float("nan"), device=DEV))
        close(f"linear dW fp16 [{tile}]", dwl, dyl.double().t() @ xl.double(), rtol=3e-6)
        if ops.linear_bwd_dx_bf16_o16_ok(m, k, j):
            d16 = ops.linear_bwd_dx_bf16_o16(dyl.to(DEV), wl.to(DEV), torch.empty(m, k, dtype=H, device=DEV))
            assert torch.equal(d16, dxl.to(H))


@pytest.mark.parametrize("n,hc,wc,c", [(3, 9, 11, 64), (2, 13, 13, 256)])
def test_fp16_pool_and_batchnorm_passes_match_the_fp32_kernels_on_the_stored_values(n, hc, wc, c):
    g = torch.Generator().manual_seed(193)
    hp, wp = hc - 2, wc - 2
    y = (torch.rand(n, hc, wc, c, generator=g) - 0.3).to(DEV)
    parts = ops.stat_parts(8 * n)
    p32 = torch.empty(n, hp, wp, c, device=DEV); i32 = torch.empty(n, hp, wp, c, dtype=torch.uint8, device=DEV)
    s32 = torch.empty(parts * 2 * c, dtype=torch.float64, device=DEV)
    ops.pool_bnstats_fwd(y, p32, i32, s32, n, hc, wc, c)
    p16 = torch.empty(n, hp, wp, c, dtype=H, device=DEV); i16 = torch.empty_like(i32); s16 = torch.empty_like(s32)
    ops.pool_bnstats_fwd(y, p16, i16, s16, n, hc, wc, c)
    assert torch.equal(p16, p32.to(H)) and torch.equal(i16, i32)
    yb = y.to(H)
    pb = torch.empty_like(p16); ib = torch.empty_like(i32); sb = torch.empty_like(s32)
    ops.pool_bnstats_fwd(yb, pb, ib, sb, n, hc, wc, c)                  # fp16 conv output: same p, same statistics
    assert torch.equal(pb, p16) and torch.equal(sb, s16)
    pf = p16.float()
    tot = s16.view(parts, 2, c).sum(0).cpu()
    assert torch.allclose(tot[0], pf.double().sum((0, 1, 2)).cpu(), rtol=1e-12, atol=1e-9)
    gamma = (torch.rand(c, generator=g) + 0.5).to(DEV); beta = (torch.rand(c, generator=g) - 0.5).to(DEV)
    st = torch.empty(4, c, device=DEV)
    ops.bn_finalize(s16, gamma, beta, None, None, 0.1, 1e-5, n * hp * wp, c, st[0], st[1], st[2], st[3])
    a = ops.bn_apply_bf16(p16, st[2], st[3], torch.empty(n, hp, wp, c, dtype=H, device=DEV), c)
    b = ops.bn_apply_bf16(pf, st[2], st[3], torch.empty(n, hp, wp, c, dtype=H, device=DEV), c)
    assert torch.equal(a, b)
    npix = n * hp * wp
    for dzt in (torch.float32, H):
        dz = (torch.rand(n, hp, wp, c, generator=g) - 0.5).to(H).to(DEV).to(dzt)
        out = {}
        for name, pp in (("f32", pf), ("h16", p16)):
            red = torch.empty(ops.stat_parts(max(npix // 64, 1)) * 2 * c, dtype=torch.float64, device=DEV)
            ops.bn_bwd_reduce(dz, pp, st[0], st[1], red, npix, c)
            coef3 = torch.empty(3 * c, device=DEV); dg = torch.empty(c, device=DEV); db = torch.empty(c, device=DEV)
            ops.bn_bwd_finalize(red, gamma, st[0], st[1], npix, c, dg, db, coef3)
            buf, dyp = ops.padded_bf16_alloc(n, hc, wc, c, DEV, dtype=H)
            dparts = torch.empty(parts * c, dtype=torch.float64, device=DEV)
            dy = torch.empty(n, hc, wc, c, device=DEV)
            ops.bnpool_bwd_bf16p(dz, pp, i16, coef3, dy, dyp, dparts, n, hc, wc, c)
            out[name] = (red.clone(), coef3, dy, buf.clone(), dparts)
            interior = buf[dyp.data_ptr() - buf.data_ptr() >> 1:][: n * (hc + 2) * (wc + 2) * c].view(n, hc + 2, wc + 2, c)[:, 1:-1, 1:-1]
            assert torch.equal(interior, dy.to(H)), "the padded 16-bit dy must be the fp32 dy rounded once to fp16"
        for u, v in zip(out["f32"], out["h16"]):
            assert torch.equal(u, v)


def _fp16_model(h, params):
    m = AVM(audio_included=True, device=DEV, precision="fp16", seed=synth.BASE_SEED)
    sd = {k: torch.from_numpy(v) for k, v in params.items()}
    sd.update(avm_ref.init_buffers())
    m.load_state_dict(sd)
    return m


@pytest.mark.parametrize("n,h", [(16, 224), (32, 40)])
def test_fp16_train_step_vs_oracle(n, h):
    """N = 16 at 224 x 224 (BASELINE.json config 5's per-clip shape) and n = 32 at 40 x 40 (the > 16-row branches)"""
    if h == 224:
        torch.manual_seed(21)
        model = AVM(audio_included=True, device=DEV, precision="fp16", seed=synth.BASE_SEED)
        (_, _), _, _, (hp3, wp3) = model._sizes(h, h)
        model._materialize(hp3 * wp3, 8)
        sd0 = model.state_dict()
        p = {k: v for k, v in sd0.items() if v.is_floating_point() and "running" not in k}
    else:
        params = synth.make_params(h, h, 30, True)
        model = _fp16_model(h, params)
        p = {k: torch.from_numpy(v.copy()) for k, v in params.items()}
    model.keep_ctx = True
    vis = torch.from_numpy(synth.make_visual(n, h, h)); aud = torch.from_numpy(synth.make_audio(n)); lab = torch.from_numpy(synth.make_labels(n))
    masks = [torch.from_numpy(m) for m in synth.make_drop_masks(n, step=0)]
    before = model._arena.clone()
    loss, pred = model.train_step(aud.to(DEV), vis.to(DEV), lab.to(DEV))
    torch.cuda.synchronize()
    assert model._guard.tolist() == [0, 0], "the loss scale overflowed on an ordinary step"
    assert not torch.equal(before, model._arena)
    taps = hip_taps(model.last_ctx)
    inter = {}
    with torch.no_grad():
        avm_ref.forward(p, avm_ref.init_buffers(), aud, vis, masks, True, inter)
    e = (model.last_logit.cpu() - inter["logit"].view(-1)).abs()
    print(f"[parity] fp16 mode {n}x{h}x{h}: logit error vs fp32 CPU reference: mean {e.mean():.2e}, max {e.max():.2e}")
    assert e.max().item() <= 2.5e-4
    o_loss, o_pred, o_g = avm_ref.train_step(p, avm_ref.init_buffers(), {}, aud, vis, lab, masks, True, pool_taps=taps)
    assert (pred.cpu().view(-1, 1) - o_pred).abs().max().item() <= 5e-4
    worst = 0.0
    for k, og in o_g.items():
        if _is_reduction_grad(k):
            continue
        mine = model.grad_of(k).cpu().reshape(og.shape)
        l2 = ((mine - og).norm() / og.norm().clamp_min(1e-30)).item()
        worst = max(worst, l2)
        assert l2 <= 0.05, f"{k}: relative L2 error {l2:.3f}"
    print(f"[parity] fp16 mode {n}x{h}x{h}: worst weight-gradient relative L2 error {worst:.2e} (loss scale {model._arena_grad_scale:g})")


def test_fp16_overflow_guard_skips_the_update_and_counts_it():
    n, h = 20, 40
    params = synth.make_params(h, h, 30, True)
    model = _fp16_model(h, params)
    vis = torch.from_numpy(synth.make_visual(n, h, h)).to(DEV); aud = torch.from_numpy(synth.make_audio(n)).to(DEV)
    lab = torch.from_numpy(synth.make_labels(n)).to(DEV)
    model.train_step(aud, vis, lab)
    good = model._arena.clone(); m1 = model._adam_m.clone()
    assert model._guard.tolist() == [0, 0]
    model.loss_scale = 2.0 ** 60                                          # every 16-bit activation gradient overflows
    model.train_step(aud, vis, lab)
    torch.cuda.synchronize()
    # one update skipped; the stamp (step 2) was consumed by the step's last launch and the step count did NOT advance: a skipped
    # step is not an optimizer step (torch's GradScaler semantics) — the retry runs under the bias corrections of step 2
    assert model._guard[1].item() == 1 and model._guard[0].item() == 0 and model._state[0].item() == 1
    assert model.overflow_skipped_steps() == 1
    assert torch.equal(model._arena, good) and torch.equal(model._adam_m, m1), "a skipped step must leave parameters and moments untouched"
    # the static scale would overflow again and again: the check a caller makes where it synchronises anyway halves it
    assert model.update_loss_scale() and model._loss_scale_for(n) == 2.0 ** 59 and not model.update_loss_scale()
    model.train_step(aud, vis, lab)                                       # still far too large: skipped again, still step 2
    torch.cuda.synchronize()
    assert model._guard[1].item() == 2 and model._state[0].item() == 1 and torch.equal(model._arena, good)
    model.loss_scale = None                                               # back to the automatic scale (backoff cleared)
    assert model._loss_scale_for(n) == 2.0 ** 15
    model.train_step(aud, vis, lab)
    torch.cuda.synchronize()
    assert model._guard[1].item() == 2 and model._state[0].item() == 2
    assert not torch.equal(model._arena, good) and torch.isfinite(model._arena).all()


def test_precision_setter_switches_format_loss_scale_and_cached_operands():
    """`m.precision = "fp16"` on a model that has stepped in bf16 must behave like a model constructed in fp16 —
    the automatic loss scale (bf16 runs with 1.0; binary16 activation gradients would flush to zero with it), no cached
    bfloat16 operand buffers handed to the fp16 kernels, a fresh 16-bit copy of linear5.weight."""
    n, h = 20, 40
    params = synth.make_params(h, h, 30, True)
    vis = torch.from_numpy(synth.make_visual(n, h, h)).to(DEV); aud = torch.from_numpy(synth.make_audio(n)).to(DEV)
    lab = torch.from_numpy(synth.make_labels(n)).to(DEV)
    masks = [torch.from_numpy(m) for m in synth.make_drop_masks(n, step=0)]
    fresh = _fp16_model(h, params)
    fresh.set_dropout_masks(masks)
    fresh.train_step(aud, vis, lab)
    m = AVM(audio_included=True, device=DEV, precision="bf16", seed=synth.BASE_SEED)
    sd = {k: torch.from_numpy(v) for k, v in params.items()}
    sd.update(avm_ref.init_buffers())
    m.load_state_dict(sd)
    m.set_dropout_masks(masks)
    assert m.loss_scale == 1.0
    m.train_step(aud, vis, lab)                                           # a bf16 step: caches bfloat16 operand buffers and the copy
    assert m._padbufs and m._w5b is not None and m._w5b.dtype == torch.bfloat16
    m.load_state_dict(sd)
    m._adam_m.zero_(); m._adam_v.zero_(); m._state.zero_(); m._adam_t = 0
    m.precision = "fp16"
    assert m.loss_scale is None and not m._padbufs and m._w5b is None
    m.train_step(aud, vis, lab)
    torch.cuda.synchronize()
    assert m._arena_grad_scale == fresh._arena_grad_scale == 2.0 ** 15
    assert torch.equal(m._garena, fresh._garena) and torch.equal(m._arena, fresh._arena), "setter-switched model != fresh fp16 model"
    assert m._guard.tolist() == [0, 0]
    # an explicit scale survives a switch; None restores the automatic one
    m.loss_scale = 256.0
    m.precision = "bf16"
    assert m.loss_scale == 256.0
    m.loss_scale = None
    m.precision = "bf16"
    assert m.loss_scale == 1.0
