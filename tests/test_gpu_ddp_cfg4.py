"""GPU, two ranks at BASELINE.json config 4's per-GPU shape: batch 64 over 8 GPUs = 8 clips = N = 128 frames of 224 x 224 per rank
(the step being configured is /root/reference/main.py:177-196).

Two processes share the box's one MI355X; gloo carries the device tensors (RCCL needs one GPU per rank — its calls are
exercised by tests/test_gpu_rccl.py with one forced rank). Each rank

  1. steps ALONE on its own 128 frames (8 copies of its own 16 frames) and is held to its own duplicated-batch oracle step —
     predictions, loss, all 30 gradient tensors against an fp64 run of the oracle, updated parameters, running statistics:
     `test_gpu_bench_shapes._run_case`, i.e. exactly what the single-GPU configs are held to;
  2. repeats the step from the same initial model with `ddp.GradSync()` and then with `ddp.GradSync(shard_linear5=True)`:
     the exchanged gradient arena must be the SUM of the two ranks' stand-alone arenas bit for bit (two addends: no
     order dependence), the forward must be the stand-alone one (local BatchNorm, local broadcast-MSE: standard DDP), the
     updated parameters must equal a fused Adam applied to that sum with 1/world, and both ranks must end on identical
     parameters.

This is the N = 128 dispatch of every kernel (split-K slab counts, the weight-gradient border path, linear5's splits) under
the gradient exchange; the 8-GPU aspect of config 4 (RCCL over xGMI, the scaling curve) cannot run on a one-GPU box.
"""
import gc
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from cvml_goalnet_amd import synth  # noqa: E402

H, N_UNIQUE, COPIES, WORLD = 224, 16, 8, 2


def _worker(rank, port, tmp):
    import torch.distributed as dist
    from cvml_goalnet_amd import ddp
    from test_gpu_bench_shapes import _fresh_model, _run_case
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        # 1. stand-alone step, pinned by this rank's own oracle step (raises on any parity failure)
        alone, (aud, vis, lab, masks) = _run_case("fp32", H, N_UNIQUE, COPIES, data_seed=synth.BASE_SEED + 1000 * rank)
        g_alone = alone._garena.clone()
        logit_alone = alone.last_logit.clone()
        del alone
        gc.collect()
        torch.cuda.empty_cache()
        want = g_alone.clone()
        dist.all_reduce(want)                                  # sum of the two stand-alone gradient arenas
        assert not torch.equal(want, g_alone * 2), "the two ranks hold the same frames: the test would be vacuous"
        # what the fused Adam makes of that sum with 1/world, from the same initial model
        ref = _fresh_model(H, "fp32", seed=7)
        ref._ensure_garena()
        ref._garena.copy_(want)
        ref.adam_step(grad_scale=1.0 / WORLD)
        want_param = ref._arena.clone()
        del ref
        torch.cuda.empty_cache()

        for shard in (False, True):
            tag = "shard_linear5" if shard else "all-reduce"
            m = _fresh_model(H, "fp32", seed=7)
            m.grad_sync = ddp.GradSync(shard_linear5=shard)
            m.set_dropout_masks(masks)
            assert m.grad_sync.sharded(m) == shard
            m.train_step(aud, vis, lab)
            m.grad_sync.wait_weights()                         # shard_linear5: the all-gather of the updated slices
            torch.cuda.synchronize()
            assert torch.equal(m.last_logit, logit_alone), f"{tag}: the forward of a rank is not its stand-alone forward"
            lo, hi = ddp.bucket_slices(m._specs, m._arena_numel)[1]
            if shard:
                # gloo has no reduce-scatter: the whole bucket is all-reduced there (ddp.py), RCCL reduces only the rank's slice
                slo, shi = m.grad_sync.shard_range(m)
                assert torch.equal(m._garena[slo:shi], want[slo:shi]), f"{tag}: this rank's slice of linear5.weight's gradient"
                assert m._adam_m.numel() == m._arena_numel - (hi - lo) + (hi - lo) // WORLD
            else:
                assert torch.equal(m._garena[lo:hi], want[lo:hi]), f"{tag}: linear5.weight's exchanged gradient"
            assert torch.equal(m._garena[:lo], want[:lo]) and torch.equal(m._garena[hi:], want[hi:]), f"{tag}: buckets 0 / 2"
            assert torch.equal(m._arena, want_param), f"{tag}: updated parameters != fused Adam on the summed gradient x 1/world"
            other = m._arena.clone()
            dist.broadcast(other, src=0)
            assert torch.equal(other, m._arena), f"{tag}: the two ranks ended on different parameters"
            del m, other
            gc.collect()
            torch.cuda.empty_cache()
        open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")
    except BaseException:
        import traceback
        open(os.path.join(tmp, f"err{rank}.txt"), "w").write(traceback.format_exc())     # the parent prints BOTH ranks' errors
        raise
    finally:
        dist.destroy_process_group()


def test_cfg4_per_gpu_shape_128_frames_of_224_two_ranks_standard_ddp_and_sharded_linear5(tmp_path):
    import torch.multiprocessing as mp
    port = 24500 + (os.getpid() % 2000)
    try:
        mp.spawn(_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    finally:
        for k in range(WORLD):
            if (tmp_path / f"err{k}.txt").exists():
                print(f"---- rank {k} ----\n" + (tmp_path / f"err{k}.txt").read_text())
    assert all((tmp_path / f"ok{k}").exists() for k in range(WORLD))
