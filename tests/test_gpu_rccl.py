"""GPU: the RCCL code path of cvml_goalnet_amd/ddp.py, executed on the `nccl` (= RCCL) backend.

A one-GPU box cannot hold two RCCL ranks, and every multi-rank test therefore runs on gloo, which takes ddp.py's `else`
branches (all-reduce instead of reduce-scatter, list all-gather of a clone). This test initialises the process group with
backend "nccl", world_size = 1, and sets GOALNET_DDP_FORCE=1 so that `GradSync` issues its collectives with the one rank:

  * `all_reduce(async_op=True)` of the three gradient buckets under backward,
  * `reduce_scatter_tensor` IN PLACE (output = the rank's slice of its own input, ddp.py `on_bucket`),
  * `all_gather_into_tensor` IN PLACE into the live weight arena / the 16-bit GEMM copy, overlapped with the next step's
    convolutions and waited for before linear5 (`after_adam`, `wait_weights`),
  * `gather_master` (16-bit modes: the fp32 master of foreign slices).

With one rank every collective is the identity, so after two train steps the parameters (and the 16-bit copy of
linear5.weight) must equal the run without any exchange bit for bit — for fp32 and for bf16 at > 16 rows (where the 16-bit
copy is what travels), at 40 x 40 and at 224 x 224. What this cannot show is RCCL with more than one rank (no scaling curve
exists: DESIGN.md §7); it shows that the calls, their in-place aliasing and their stream ordering are accepted and correct
on the backend the multi-GPU bench uses.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [("fp32", 40, 10), ("fp32", 40, 32), ("bf16", 40, 32), ("fp32", 224, 16), ("bf16", 224, 32)]


def _two_steps(precision, h, n, sync_kwargs):
    from cvml_goalnet_amd import ddp, synth
    from test_gpu_bench_shapes import _fresh_model
    m = _fresh_model(h, precision, seed=5)
    if sync_kwargs is not None:
        m.grad_sync = ddp.GradSync(**sync_kwargs)
        assert m.grad_sync.active and m.grad_sync._nccl()
        assert m.grad_sync.sharded(m) == bool(sync_kwargs.get("shard_linear5"))
    vis = torch.from_numpy(synth.make_visual(n, h, h)).cuda()
    aud = torch.from_numpy(synth.make_audio(n)).cuda()
    lab = torch.from_numpy(synth.make_labels(n)).cuda()
    losses = []
    for _ in range(2):
        loss, _ = m.train_step(aud, vis, lab)
        losses.append(loss)
    if m.grad_sync is not None:
        m.grad_sync.gather_master(m)            # waits for the weight all-gather; 16-bit + sharded: refreshes the fp32 master
    torch.cuda.synchronize()
    out = {"arena": m._arena.clone(), "loss": torch.cat(losses).cpu(),
           "w5b": None if m._w5b is None else m._w5b.clone(), "adam_m": m._adam_m.clone()}
    del m
    torch.cuda.empty_cache()
    return out


def _worker(rank, port, tmp):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["GOALNET_DDP_FORCE"] = "1"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        for precision, h, n in CASES:
            base = _two_steps(precision, h, n, None)
            for kw in ({}, {"shard_linear5": True}):
                got = _two_steps(precision, h, n, kw)
                tag = f"{precision} {n}x{h}x{h} GradSync({kw})"
                assert torch.equal(got["loss"], base["loss"]), f"{tag}: losses differ from the run without exchange"
                assert torch.equal(got["arena"], base["arena"]), f"{tag}: parameters differ from the run without exchange"
                assert torch.equal(got["adam_m"], base["adam_m"]), f"{tag}: Adam moments differ"
                if base["w5b"] is not None:
                    assert got["w5b"] is not None and torch.equal(got["w5b"], base["w5b"]), f"{tag}: 16-bit copy of linear5.weight differs"
                print(f"[rccl] {tag}: two steps bit-equal to the run without exchange")
        open(os.path.join(tmp, "ok"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_rccl_backend_one_forced_rank_allreduce_and_inplace_reduce_scatter_all_gather(tmp_path):
    import torch.multiprocessing as mp
    port = 26500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(port, str(tmp_path)), nprocs=1, join=True)
    assert (tmp_path / "ok").exists()
