"""GPU, two ranks: the data-parallel train step through the real kernels.

Two processes share the box's one MI355X (gloo moves the device tensors; RCCL needs one GPU per rank and is what
bench.py uses under torchrun). Each rank runs `AVM.train_step` on its half of the frames:

  * standard mode (`GradSync`): local BatchNorm statistics, local broadcast MSE -> the exchanged gradient is the SUM of
    the two ranks' stand-alone gradients and Adam applies it with 1/world;
  * global-batch mode (`ddp.enable_global_batch`: `SyncStats` + summed gradients): predictions, loss, BatchNorm running
    statistics, gradients and updated parameters equal ONE process stepping on all the frames.

The single-process side of both comparisons is the same HIP path (its parity with the reference is test_gpu_avm.py's
job), so the tolerances here only cover summation order: 1e-5 of each tensor's scale.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from cvml_goalnet_amd import AVM, synth  # noqa: E402

DEV = "cuda:0"
N, H, WORLD = 8, 40, 2


def _model():
    from oracle import avm_ref
    m = AVM(audio_included=True, device=DEV, seed=synth.BASE_SEED)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_params(H, H, 30, True).items()}
    sd.update(avm_ref.init_buffers())
    m.load_state_dict(sd)
    return m


def _inputs(shard):
    vis = torch.from_numpy(synth.make_visual(N, H, H))[shard].to(DEV)
    aud = torch.from_numpy(synth.make_audio(N))[shard].to(DEV)
    lab = torch.from_numpy(synth.make_labels(N))[shard].to(DEV)
    masks = [torch.from_numpy(m)[shard] for m in synth.make_drop_masks(N, step=0)]
    return aud, vis, lab, masks


def _step(m, shard):
    aud, vis, lab, masks = _inputs(shard)
    m.set_dropout_masks(masks)
    loss, pred = m.train_step(aud, vis, lab)
    torch.cuda.synchronize()
    if m.grad_sync is not None:
        m.grad_sync.wait_weights()                                     # shard_linear5: the all-gather of the updated slices
        torch.cuda.synchronize()
    out = {"loss": loss.cpu(), "pred": pred.cpu().reshape(-1), "grad": m._garena.cpu().clone(), "param": m._arena.cpu().clone()}
    out.update({k: v.cpu().clone() for k, v in m.state_dict().items() if "running" in k or "num_batches" in k})
    return out


def _worker(rank, port, tmp):
    import torch.distributed as dist
    from cvml_goalnet_amd import ddp
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    shard = slice(rank * N // WORLD, (rank + 1) * N // WORLD)
    res = {"alone": _step(_model(), shard)}
    m = _model()
    m.grad_sync = ddp.GradSync()
    res["ddp"] = _step(m, shard)
    m = ddp.enable_global_batch(_model())
    res["global"] = _step(m, shard)
    # ZeRO-1 for linear5.weight: reduce-scatter, Adam on this rank's slice, all-gather of the updated slices
    m = _model()
    m.grad_sync = ddp.GradSync(shard_linear5=True)
    assert m.grad_sync.sharded(m)
    res["sharded"] = _step(m, shard)
    res["sharded_state_elems"] = m._adam_m.numel()
    res["sharded2"] = _step(m, shard)                                  # second step: the gather of step 1 was waited for
    m = _model()
    m.grad_sync = ddp.GradSync()
    _step(m, shard)
    res["ddp2"] = _step(m, shard)
    # the same with precision="bf16" at > 16 rows would gather the bf16 copy; covered on the host side (test_host_logic.py)
    # replicas built from different torch seeds: sync_params makes them one model before the first exchange
    torch.manual_seed(1000 + rank)
    m = AVM(audio_included=True, device=DEV)
    m.grad_sync = ddp.GradSync()
    aud, vis, lab, _ = _inputs(shard)
    m.train_step(aud, vis, lab)
    torch.cuda.synchronize()
    res["seeds"] = {"param": m._arena.cpu().clone(), "dropout_seed": m.dropout_seed,
                    "rm": m.visbl.bnorm2.running_mean.cpu().clone()}
    # bf16-compressed exchange of bucket 1 (an extension, off by default)
    try:
        m = _model()
        m.grad_sync = ddp.GradSync(compress="bf16")
        res["compress"] = _step(m, shard)
    except Exception as e:                                             # this gloo build may not reduce bfloat16
        res["compress"] = repr(e)
    torch.save(res, os.path.join(tmp, f"r{rank}.pt"))
    dist.destroy_process_group()


def _close(a, b, rel, what, floor=0.0):
    scale = max(b.abs().max().item(), 1e-30)
    err = (a - b).abs().max().item()
    assert err <= rel * scale + floor, f"{what}: err {err:.3e} vs scale {scale:.3e}"


def test_two_rank_train_step_standard_and_global_batch(tmp_path):
    import torch.multiprocessing as mp
    port = 23500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    r = [torch.load(tmp_path / f"r{k}.pt") for k in range(WORLD)]
    ref_model = _model()
    specs = ref_model._specs
    whole = _step(ref_model, slice(0, N))

    # ---- standard DDP: exchanged gradient = sum of the stand-alone gradients, identical on both ranks
    want = r[0]["alone"]["grad"] + r[1]["alone"]["grad"]
    assert torch.equal(r[0]["ddp"]["grad"], r[1]["ddp"]["grad"])
    assert torch.equal(r[0]["ddp"]["param"], r[1]["ddp"]["param"])
    for s in specs:
        sl = slice(s.offset, s.offset + s.numel)
        _close(r[0]["ddp"]["grad"][sl], want[sl], 1e-6, f"ddp grad {s.name}")
    for k in range(WORLD):                                             # forward is the stand-alone one (local statistics)
        assert torch.equal(r[k]["ddp"]["pred"], r[k]["alone"]["pred"]) and torch.equal(r[k]["ddp"]["loss"], r[k]["alone"]["loss"])
    assert not torch.equal(r[0]["alone"]["grad"], r[1]["alone"]["grad"])

    # ---- sharded linear5.weight: same update as the replicated Adam, bit for bit, with 1/world of its optimizer state
    for step, ref in (("sharded", "ddp"), ("sharded2", "ddp2")):
        for k in range(WORLD):
            assert torch.equal(r[k][step]["param"], r[0][ref]["param"]), f"{step}: rank {k} parameters differ from the replicated path"
            assert torch.equal(r[k][step]["pred"], r[k][ref]["pred"])
    w5 = ref_model.spec("visbl.linear5.weight")
    assert r[0]["sharded_state_elems"] == ref_model._arena_numel - w5.numel + w5.numel // WORLD
    # ---- replicas from different seeds end up identical; their dropout streams differ
    assert torch.equal(r[0]["seeds"]["param"], r[1]["seeds"]["param"])
    assert r[0]["seeds"]["dropout_seed"] != r[1]["seeds"]["dropout_seed"]
    # ---- bf16-compressed exchange: bucket 1 == the bf16-rounded sum of the bf16-rounded local gradients (2^-8 relative)
    if isinstance(r[0]["compress"], str):
        print("[ddp] compress='bf16' not exercised: " + r[0]["compress"])
    else:
        lo, hi = w5.offset, w5.offset + w5.numel
        got = r[0]["compress"]["grad"][lo:hi]
        assert torch.equal(got, r[1]["compress"]["grad"][lo:hi])
        ssum = want[lo:hi]
        assert ((got - ssum).abs() <= 2.0 ** -7 * (r[0]["alone"]["grad"][lo:hi].abs() + r[1]["alone"]["grad"][lo:hi].abs()) + 1e-30).all()
        for s_ in specs:                                              # the other buckets stay exact fp32 sums
            if s_.name != "visbl.linear5.weight":
                sl = slice(s_.offset, s_.offset + s_.numel)
                assert torch.equal(r[0]["compress"]["grad"][sl], r[0]["ddp"]["grad"][sl]), s_.name

    # ---- global batch: two ranks == one process on all the frames
    g = [x["global"] for x in r]
    assert torch.equal(g[0]["grad"], g[1]["grad"]) and torch.equal(g[0]["param"], g[1]["param"])
    assert torch.equal(g[0]["loss"], g[1]["loss"])
    _close(g[0]["loss"], whole["loss"], 2e-6, "global loss")
    _close(torch.cat([g[0]["pred"], g[1]["pred"]]), whole["pred"], 2e-6, "global predictions")
    for k in whole:
        if "running" in k:
            _close(g[0][k], whole[k], 2e-6, k)
            assert torch.equal(g[0][k], g[1][k])
        elif "num_batches" in k:
            assert int(g[0][k]) == int(whole[k]) == 1
    wmax = {s.name: whole["grad"][s.offset:s.offset + s.numel].abs().max().item() for s in specs}
    for s in specs:
        sl = slice(s.offset, s.offset + s.numel)
        # bias / BatchNorm-affine gradients are cancelling sums (test_gpu_avm._is_reduction_grad): floor from the layer's weight
        floor = 0.0
        if s.name.endswith(".bias") or ".bnorm" in s.name:
            wname = (s.name.replace("bnorm", "conv") if ".bnorm" in s.name else s.name).rsplit(".", 1)[0] + ".weight"
            floor = 2e-5 * wmax[wname]
        _close(g[0]["grad"][sl], whole["grad"][sl], 2e-5, f"global grad {s.name}", floor)
    # the local-statistics run is a different computation: the test would be vacuous if it also matched
    assert (torch.cat([r[0]["alone"]["pred"], r[1]["alone"]["pred"]]) - whole["pred"]).abs().max().item() > 1e-4
