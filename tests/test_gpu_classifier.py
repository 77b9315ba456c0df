"""GPU: the classifier-head variant (EXTENSION — comment-only in the reference: utils.py:257 nn.Softmax(dim = 1), main.py:69
nn.CrossEntropyLoss(), main.py:96, 189 `(labels - 1).long()`, main.py:97, 190 `argmax + 1`).

"parity unpinned" against the reference (it holds no runnable form of this variant); the checker is the oracle's literal
restatement of those lines on torch CPU ops (oracle/avm_ref.py: head="classifier", ce_loss) — the ATen kernels the
reference would call. fp32 tolerances as for the regression head: scores 2e-5 abs, loss 2e-5 rel, gradients 1e-4 of the
tensor's max; argmax bit-exact."""
import warnings

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from cvml_goalnet_amd import AVM, ops, synth  # noqa: E402
from cvml_goalnet_amd.loop import VideoTrainer  # noqa: E402
from oracle import avm_ref  # noqa: E402
from test_gpu_avm import _is_reduction_grad, _weight_of, hip_taps, routing_disagreements, NEAR_TIE  # noqa: E402

DEV = "cuda:0"
C = 5


def _params(h, audio=True):
    p = synth.make_params(h, h, 30, audio)
    p["fusion.12.weight"] = synth.uniform(900, (C, 128), -1.0 / np.sqrt(128), 1.0 / np.sqrt(128))
    p["fusion.12.bias"] = synth.uniform(901, (C,), -1.0 / np.sqrt(128), 1.0 / np.sqrt(128))
    return p


def _model(h, audio=True, dropout="device"):
    params = _params(h, audio)
    m = AVM(audio_included=audio, device=DEV, seed=synth.BASE_SEED, head="classifier")
    sd = {k: torch.from_numpy(v) for k, v in params.items()}
    sd.update(avm_ref.init_buffers())
    m.load_state_dict(sd)
    m.dropout_mode = dropout
    return m, params


@pytest.mark.parametrize("n", [1, 10, 515])
def test_classifier_head_kernels_vs_torch_fp64(n):
    g = torch.Generator().manual_seed(3 + n)
    k = 128
    h = torch.rand(n, k, generator=g) * 2 - 0.5
    w = (torch.rand(C, k, generator=g) - 0.5) * 0.4
    b = torch.rand(C, generator=g) - 0.5
    mult = (torch.rand(n, k, generator=g) >= 0.2).float() * 1.25
    lab = torch.randint(1, C + 1, (n,), generator=g).float()
    hd = h.double().requires_grad_(True); wd = w.double().requires_grad_(True); bd = b.double().requires_grad_(True)
    z = F.linear(hd, wd, bd)
    s = 4 * torch.softmax(z, dim=1) + 1
    loss = F.cross_entropy(s, (lab - 1).long())
    loss.backward()
    logits = torch.empty(n, C, device=DEV); scores = torch.empty(n, C, device=DEV)
    ops.cls_head_fwd(h.to(DEV), w.to(DEV), b.to(DEV), logits, scores)
    assert (logits.cpu().double() - z.detach()).abs().max().item() < 2e-6
    assert (scores.cpu().double() - s.detach()).abs().max().item() < 2e-6
    lg = torch.empty(1, device=DEV); ds = torch.empty(n, C, device=DEV)
    ops.cross_entropy(scores, lab.to(DEV), lg, ds)
    assert abs(lg.item() - loss.item()) < 2e-6 * max(1.0, abs(loss.item()))
    dh = torch.empty(n, k, device=DEV); dw = torch.empty(C, k, device=DEV); db = torch.empty(C, device=DEV)
    ops.cls_head_bwd(ds, scores, h.to(DEV), w.to(DEV), mult.to(DEV), dh, dw, db)
    for name, got, want in (("dh", dh, hd.grad * mult.double()), ("dw", dw, wd.grad), ("db", db, bd.grad)):
        scale = max(want.abs().max().item(), 1e-30)
        assert (got.cpu().double() - want).abs().max().item() <= 5e-6 * scale, name
    cls = ops.argmax_plus1(scores, torch.empty(n, device=DEV))
    assert torch.equal(cls.cpu(), (torch.argmax(scores.cpu(), dim=1) + 1).float())
    # ties: the first maximal column (torch.argmax)
    t = torch.tensor([[1.0, 3.0, 3.0, 2.0, 3.0], [2.0, 2.0, 2.0, 2.0, 2.0]], device=DEV)
    assert ops.argmax_plus1(t, torch.empty(2, device=DEV)).tolist() == [2.0, 1.0]


@pytest.mark.parametrize("audio", [True, False])
def test_classifier_train_steps_vs_oracle(audio):
    n, h = 10, 40
    model, params = _model(h, audio)
    model.keep_ctx = True
    vis = torch.from_numpy(synth.make_visual(n, h, h))
    aud = torch.from_numpy(synth.make_audio(n)) if audio else None
    lab = torch.from_numpy(synth.make_labels(n))
    p = {k: torch.from_numpy(v.copy()) for k, v in params.items()}
    b = avm_ref.init_buffers()
    state = {}
    for s in range(2):
        loss, scores = model.train_step(aud.to(DEV) if audio else None, vis.to(DEV), lab.to(DEV))
        torch.cuda.synchronize()
        assert scores.shape == (n, C)
        taps = hip_taps(model.last_ctx)
        masks = [torch.from_numpy(m) for m in synth.make_drop_masks(n, step=s)]
        inter = {}
        with torch.no_grad():
            avm_ref.forward(p, {k: v.clone() for k, v in b.items()}, aud, vis, masks, audio, inter, head="classifier")
        nd, worst = routing_disagreements(inter, taps)
        assert worst <= NEAR_TIE
        o_loss, o_pred, o_g = avm_ref.train_step(p, b, state, aud, vis, lab, masks, audio, pool_taps=taps if nd else None,
                                                 head="classifier")
        assert (scores.cpu() - o_pred).abs().max().item() < 2e-5
        assert abs(loss.item() - o_loss.item()) < 2e-5 * max(1.0, abs(o_loss.item()))
        assert torch.equal(model.predict_classes(scores).cpu(), (torch.argmax(o_pred, dim=1) + 1).float())
        for name, og in o_g.items():
            mine = model.grad_of(name).cpu().reshape(og.shape)
            scale = max(og.abs().max().item(), 1e-30)
            floor = 2e-5 * o_g[_weight_of(name)].abs().max().item() if _is_reduction_grad(name) else 0.0
            err = (mine - og).abs().max().item()
            assert err <= 1e-4 * scale + floor, f"step {s} {name}: {err:.3e} vs {scale:.3e}"
        sd = model.state_dict()
        for k in p:                                                      # continue from the device's parameters (test_gpu_avm.py)
            p[k].copy_(sd[k])
        for k in b:
            b[k].copy_(sd[k])


def test_classifier_dropin_surface_with_torch_cross_entropy_and_the_loop():
    """the commented-out lines as they would run: criterion = nn.CrossEntropyLoss(); loss = criterion(pred, (labels-1).long());
    pred = argmax(pred, 1) + 1 — CPU tensors in, stock torch.optim.Adam; then the per-video loop collecting classes"""
    n, h = 10, 40
    model, params = _model(h, True, dropout="off")
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    crit = torch.nn.CrossEntropyLoss()
    vis = torch.from_numpy(synth.make_visual(n, h, h)); aud = torch.from_numpy(synth.make_audio(n)); lab = torch.from_numpy(synth.make_labels(n))
    opt.zero_grad()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = model(aud, vis)
    assert out.shape == (n, C) and out.device.type == "cpu" and out.requires_grad
    loss = crit(out, (lab - 1).long())
    loss.backward()
    opt.step()
    p = {k: torch.from_numpy(v.copy()) for k, v in params.items()}
    o_loss, o_pred, o_g = avm_ref.train_step(p, avm_ref.init_buffers(), {}, aud, vis, lab, None, True, head="classifier")
    assert (out.detach() - o_pred).abs().max().item() < 2e-5 and abs(loss.item() - o_loss.item()) < 2e-5
    gw = model.grad_of("fusion.12.weight").cpu()
    assert (gw - o_g["fusion.12.weight"]).abs().max().item() <= 1e-4 * o_g["fusion.12.weight"].abs().max().item()
    classes = (torch.argmax(out, axis=1) + 1).tolist()
    assert all(1 <= c <= C for c in classes)
    # the loop: predictions collected per sub-batch are the classes (main.py:190, 196)
    m2, _ = _model(h, True)
    tr = VideoTrainer(m2, subbatch_size=10)
    f = 25
    v2 = torch.from_numpy(synth.make_visual(f, h, h)); a2 = torch.from_numpy(synth.make_audio(f)); l2 = torch.from_numpy(synth.make_labels(f))
    losses, preds = tr.train_video(a2, v2, l2)
    torch.cuda.synchronize()
    assert losses.shape == (3,) and preds.shape == (f,) and set(preds.cpu().tolist()) <= {1.0, 2.0, 3.0, 4.0, 5.0}
    el, ep = tr.eval_video(a2, v2, l2)
    assert ep.shape == (f,) and torch.isfinite(el).all()
