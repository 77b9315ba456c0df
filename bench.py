#!/usr/bin/env python3
"""bench.py — training clips/s of the AVM hot path on MI355X (BASELINE.json metric), one JSON line on stdout.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch of synthetic input: forward, broadcast-MSE, backward and the
fused Adam of `AVM` (/root/reference/main.py:187-193) on 64 clips x 16 frames = 1024 frames of 3x224x224 (+ a
30x30 MFCC each) per GPU — the configuration BASELINE.json's metric is quoted on. Inputs are resident in HBM
before the timed region. N > 1 is weak scaling: every rank trains its own 64 clips (local BatchNorm, local
MSE, as an independent reference process would), gradients are summed over RCCL/xGMI in three buckets
overlapped with backward and averaged inside the fused Adam (cvml_goalnet_amd/ddp.py).

Extra objects on the JSON line:
  roofline      the dominant kernel = the fp32-MFMA implicit-GEMM convolution forward (conv2 + conv3, 84 % of
                forward MACs): algorithmic flops of its launches / their duration, measured with HIP events
                inside the timed steps; peak = 157.3 TFLOP/s (fp32 MFMA, MI355X_MICROARCH.md).
  cpu_baseline  the oracle (CPU restatement of the reference, oracle/avm_ref.py) timed on this host's cores on a
                bounded sample (1 clip = 16 frames of 224x224 per step), rank 0, N = 1 only.
  parity        logit MAE / max-abs of the HIP forward vs that CPU reference on the same 16 frames and weights: on the
                random-init weights before the first optimizer step, and again on the weights the timed steps left.
  roofline_fp32_path  (bf16 runs, N = 1 only) the same workload with precision="fp32" — the reference's arithmetic on the
                fp32 matrix cores — so that the line carries the fp32-MFMA roofline of the conv forward next to the bf16 one.
  native_40x40_loop  SURVEY.md §8(d) "report additionally frames/s at the reference-native 40x40": the reference's own
                operating point (main.py:169-198: one video at a time, 10 frames per optimizer step, fp32), run by
                loop.VideoTrainer as one HIP-graph launch per sub-batch; N = 1 only, a few seconds.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md, chip-level parameters
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA (not the 2:1-sparsity marketing figure)
FRAMES_PER_CLIP = 16


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def make_inputs(n, h, w, device, seed):
    from cvml_goalnet_amd import ops, synth
    vis = torch.empty(n, 3, h, w, device=device)
    ops.fill_uniform(vis.view(-1), seed, synth.TID_VISUAL, 0.0, 1.0)
    flat = vis.view(n, -1)
    mn = flat.amin(dim=1, keepdim=True)
    mx = flat.amax(dim=1, keepdim=True)
    flat.sub_(mn).div_(mx - mn)                      # per-frame min-max to [0,1], as utils.py:284 (data prep, untimed)
    aud = torch.from_numpy(synth.make_audio(n, 30, seed)).to(device)
    lab = torch.from_numpy(synth.make_labels(n, seed)).to(device)
    return aud, vis, lab


def logit_parity(model, h, w, seed):
    """Logits of the HIP forward vs the CPU oracle on the same 16 frames, weights and dropout masks (regenerated from the
    seed formula on both sides); BatchNorm in train mode. The probe restores the model's BatchNorm buffers."""
    from cvml_goalnet_amd import synth
    from oracle import avm_ref
    n = FRAMES_PER_CLIP
    aud, vis, lab = make_inputs(n, h, w, model._device, seed + 1)
    if not model._materialized:
        with torch.no_grad():
            model.forward_device(aud, vis, save=False)               # Lazy parameters take their random init here
        for i in (1, 2, 3):                                          # ... and the buffers go back to their initial state
            bn = getattr(model.visbl, f"bnorm{i}")
            bn.running_mean.zero_(); bn.running_var.fill_(1.0); bn.num_batches_tracked.zero_()
    sd = model.state_dict()                                          # torch-native layouts, CPU
    p = {k: v for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    b = {k: v.clone() for k, v in sd.items() if k not in p}
    step = model._drop_step
    masks = [torch.from_numpy(m) for m in synth.make_drop_masks(n, seed=model.dropout_seed, step=step)]
    bn_backup = {k: getattr(*model._module_of(k)).clone() for k in b}
    with torch.no_grad():
        model.forward_device(aud, vis, save=False)
    hip_logit = model.last_logit.cpu()
    for k, v in bn_backup.items():                                   # the parity probe must not disturb the model
        getattr(*model._module_of(k)).copy_(v)
    inter = {}
    with torch.no_grad():
        avm_ref.forward(p, {k: v.clone() for k, v in b.items()}, aud.cpu(), vis.cpu(), masks, model.audio_included, inter)
    ref_logit = inter["logit"].view(-1)
    d = (hip_logit - ref_logit).abs()
    return {"logit_mae_vs_cpu_ref": d.mean().item(), "logit_maxabs_vs_cpu_ref": d.max().item(),
            "max_abs_logit": ref_logit.abs().max().item(), "frames": n}


def cpu_baseline(model, h, w, seed, max_seconds=40.0):
    """Time the oracle's train step on the host cores (bounded sample: one clip per step)."""
    from cvml_goalnet_amd import synth
    from oracle import avm_ref
    n = FRAMES_PER_CLIP
    threads = torch.get_num_threads()
    sd = model.state_dict()
    p = {k: v for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    b = {k: v.clone() for k, v in sd.items() if k not in p}
    aud, vis, lab = make_inputs(n, h, w, model._device, seed + 1)
    a_c, v_c, l_c = aud.cpu(), vis.cpu(), lab.cpu()
    masks = [torch.from_numpy(m) for m in synth.make_drop_masks(n, seed=model.dropout_seed, step=0)]
    # train steps of one clip (the Adam state allocation of the first step is the warm-up)
    state = {}
    times = []
    t_begin = time.time()
    for i in range(3):
        t0 = time.time()
        avm_ref.train_step(p, b, state, a_c, v_c, l_c, masks, model.audio_included)
        times.append(time.time() - t0)
        if time.time() - t_begin > max_seconds:
            break
    t = min(times[1:]) if len(times) > 1 else times[0]
    return {"value": (n / FRAMES_PER_CLIP) / t, "unit": "clips/s", "cores": threads, "kind": "port",
            "sample": f"oracle train step (fwd+MSE+bwd+Adam) on {n} frames (1 clip) of {h}x{w}, fp32, torch {torch.__version__} CPU, "
                      f"{len(times)} steps, best of the non-first: {t:.2f} s/step"}


def fp32_path_roofline(dev, n, h, w, audio, seed):
    """The same step on the reference's own arithmetic (precision="fp32": fp32 MFMA, `v_mfma_f32_32x32x2_f32`): two timed
    steps, conv2 + conv3 forward launches bracketed with HIP events exactly as for the main roofline object."""
    from cvml_goalnet_amd import AVM
    model = AVM(audio_included=audio, device=dev, seed=seed, precision="fp32")
    aud, vis, lab = make_inputs(n, h, w, dev, seed)
    if not audio:
        aud = None
    for _ in range(2):
        model.train_step(aud, vis, lab)
    model.kernel_events = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        model.train_step(aud, vis, lab)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    ev = model.kernel_events["conv_fwd"]
    model.kernel_events = None
    ms = [a.elapsed_time(b) for a, b, _ in ev]
    fl = [f for _, _, f in ev]
    achieved = sum(fl) / (sum(ms) * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP32_MFMA_PEAK_TFLOPS,
            "kernel": "gemm_f32_kernel<ConvALoader<true>, KCLoader<false>> (conv2 + conv3 forward, fp32 MFMA implicit GEMM)",
            "launches": len(ms), "avg_launch_ms": sum(ms) / len(ms), "ms_per_step": 1e3 * dt,
            "clips_per_s": (n / FRAMES_PER_CLIP) / dt, "dtype": "f32"}


def native40_loop(dev, frames=300, videos=3):
    """The reference's loop at its native size: 40x40 frames, sub-batches of 10, fp32, audio on (main.py:44-46, 169-198)."""
    from cvml_goalnet_amd import AVM, synth
    from cvml_goalnet_amd.loop import VideoTrainer
    vis = torch.from_numpy(synth.make_visual(frames, 40, 40)).to(dev)
    aud = torch.from_numpy(synth.make_audio(frames)).to(dev)
    lab = torch.from_numpy(synth.make_labels(frames)).to(dev)
    model = AVM(audio_included=True, device=dev, precision="fp32")
    tr = VideoTrainer(model, subbatch_size=10)
    tr.train_video(aud, vis, lab)                       # warm-up video: first step eager, graph captured at the second
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(videos):
        losses, preds = tr.train_video(aud, vis, lab)
        batch_loss = losses.mean().item()               # the one host read-back per video (main.py:203)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = videos * ((frames + 9) // 10)
    # the same 10-frame train step by the CPU oracle on this host's cores (what the reference's own loop costs per step)
    cpu_us, cpu_threads = None, None
    try:
        from oracle import avm_ref
        sd = model.state_dict()
        p = {k: v for k, v in sd.items() if v.is_floating_point() and "running" not in k}
        b = {k: v.clone() for k, v in sd.items() if k not in p}
        a_c, v_c, l_c = aud[:10].cpu(), vis[:10].cpu(), lab[:10].cpu()
        masks = [torch.from_numpy(m) for m in synth.make_drop_masks(10, step=0)]
        best = None
        all_threads = torch.get_num_threads()
        for th in sorted({8, all_threads}):              # small tensors: more threads are not faster; report the better setting
            torch.set_num_threads(th)
            state, ts = {}, []
            pp = {k: v.clone() for k, v in p.items()}
            bb = {k: v.clone() for k, v in b.items()}
            for _ in range(5):
                t1 = time.perf_counter()
                avm_ref.train_step(pp, bb, state, a_c, v_c, l_c, masks, True)
                ts.append(time.perf_counter() - t1)
            if best is None or min(ts[1:]) < best[0]:
                best = (min(ts[1:]), th)
        torch.set_num_threads(all_threads)
        cpu_us, cpu_threads = 1e6 * best[0], best[1]
    except Exception as e:
        log(f"cpu oracle at 40x40 failed: {e!r}")
    return {"frames_per_s": videos * frames / dt, "cpu_oracle_us_per_step": cpu_us, "cpu_threads": cpu_threads, "us_per_step": 1e6 * dt / steps, "frames_per_step": 10, "h": 40, "w": 40,
            "dtype": "f32", "mode": "one HIP graph launch per sub-batch (loop.VideoTrainer)", "graph_replays": tr.replays,
            "eager_steps": tr.eager_steps, "videos": videos, "frames_per_video": frames, "last_batch_loss": batch_loss}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--clips", type=int, default=64, help="clips of 16 frames per GPU per step")
    ap.add_argument("--hw", type=int, default=224)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-audio", action="store_true")
    ap.add_argument("--no-native40", action="store_true", help="skip the extra 40x40 / 10-frame loop measurement")
    ap.add_argument("--global-batch", action="store_true",
                    help="N > 1: BatchNorm statistics and the broadcast MSE over all ranks' frames, gradients summed "
                         "(ddp.SyncStats: the step equals one reference process on the global batch)")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default=os.environ.get("GOALNET_BENCH_DTYPE", "bf16"),
                    help="bf16 (default) = bf16-MFMA contractions with fp32 accumulation/statistics/master weights, logits within "
                         "the north star's 1e-3 of the fp32 CPU reference; f32 = the reference's arithmetic on the fp32 matrix cores")
    args = ap.parse_args()

    # stdout carries exactly ONE line (the JSON): anything libraries print meanwhile (RCCL prints a version banner to
    # stdout at communicator creation) is diverted to stderr until the result is ready
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"note: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")
    import torch.distributed as dist
    distributed = world > 1 or os.environ.get("GOALNET_DDP_FORCE") == "1"
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
        local = 0
    dev = torch.device("cuda", local)

    from cvml_goalnet_amd import AVM, synth
    from cvml_goalnet_amd.ddp import GradSync

    n = args.clips * FRAMES_PER_CLIP
    h = w = args.hw
    seed = synth.BASE_SEED + rank
    torch.manual_seed(1234)                                   # same initial weights on every rank
    model = AVM(audio_included=not args.no_audio, device=dev, seed=seed, precision="bf16" if args.dtype == "bf16" else "fp32")
    aud, vis, lab = make_inputs(n, h, w, dev, seed)
    if args.no_audio:
        aud = None
    if distributed and args.global_batch:
        from cvml_goalnet_amd.ddp import enable_global_batch
        enable_global_batch(model, compress="bf16" if args.dtype == "bf16" else None)
    elif distributed:
        model.grad_sync = GradSync(compress="bf16" if args.dtype == "bf16" else None)

    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            parity = logit_parity(model, h, w, synth.BASE_SEED)     # on the random-init weights, before any optimizer step
            parity.update({"weights": "random init (before the first optimizer step)", "dropout": "masks from seed formula", "bn": "train"})
        except Exception as e:
            log(f"parity probe failed: {e!r}")
        torch.cuda.empty_cache()            # the probe's 16-frame buffers must not fragment the pool the 1024-frame steps use
    if distributed:
        # RCCL builds its communicator, channels and staging buffers lazily on the first collectives of each kind (seen with one
        # forced rank: +80 ms per step over the first steps): get that out of the way before the W warm-up steps
        for dt_ in (torch.float32, torch.bfloat16):
            t = torch.zeros(64 << 20, dtype=dt_, device=dev)
            for _ in range(2):
                dist.all_reduce(t)
        torch.cuda.synchronize()
        del t
    for _ in range(args.warmup):
        model.train_step(aud, vis, lab)
    model.kernel_events = {}
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, pred = model.train_step(aud, vis, lab)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    dt = time.perf_counter() - t0
    if distributed:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    events = model.kernel_events
    model.kernel_events = None
    assert torch.isfinite(loss).all() and torch.isfinite(pred).all(), "non-finite loss/prediction"

    if rank == 0:
        clips = args.clips * world * args.steps
        res = {
            "metric": "training clips/sec at batch 64, 16-frame 224² clips; logit MAE vs CPU ref",
            "value": clips / dt, "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"AVM train step (forward + broadcast-MSE + backward + fused Adam), {args.clips} clips x 16 frames = "
                                   f"{n} frames of 3x{h}x{w} + 30x30 MFCC per GPU; dropout live (device masks), BatchNorm train mode",
                       "frames_per_gpu": n, "h": h, "w": w, "global_clips_per_step": args.clips * world,
                       "parallelism": f"dp{world}", "ddp_semantics": ("BatchNorm sums + broadcast MSE over all ranks' frames, gradient sum (global batch)"
                                                                 if distributed and args.global_batch else
                                                                 "local BN + local MSE per rank, gradient mean (standard DDP)") +
                       ("; linear5.weight gradient exchanged as bf16" if args.dtype == "bf16" and distributed else ""),
                       "params": int(sum(s.numel for s in model._specs)), "final_loss": float(loss.item())},
        }
        ev = events.get("conv_fwd", [])
        if ev:
            ms = [a.elapsed_time(b) for a, b, _ in ev]
            fl = [f for _, _, f in ev]
            achieved = sum(fl) / (sum(ms) * 1e-3) / 1e12
            traffic = None
            tj = os.path.join(ROOT, "profiles", f"conv_fwd_traffic_{args.dtype}.json")
            if os.path.exists(tj) and args.clips == 64 and h == 224:
                try:
                    traffic = json.load(open(tj)).get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            peak = BF16_MFMA_PEAK_TFLOPS if args.dtype == "bf16" else FP32_MFMA_PEAK_TFLOPS
            kname = ("gemm_bf16_256_kernel<ConvAPadLoader256<64>, KCLoader256<32>, 0> (conv2 + conv3 forward, bf16 MFMA implicit GEMM, 256x256 phased tile, zero-padded bf16 activations)" if args.dtype == "bf16"
                     else "gemm_f32_kernel<ConvALoader<true>, KCLoader<false>> (conv2 + conv3 forward, fp32 MFMA implicit GEMM)")
            res["roofline"] = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                               "frac": achieved / peak, "traffic": traffic,
                               "kernel": kname,
                               "launches": len(ms), "avg_launch_ms": sum(ms) / len(ms),
                               "algorithmic_flops_per_launch": sum(fl) / len(fl)}
            others = {}
            for k, lst in events.items():
                if k == "conv_fwd":
                    continue
                t = sum(a.elapsed_time(b) for a, b, _ in lst) * 1e-3
                others[k] = {"tflops": sum(f for _, _, f in lst) / t / 1e12, "ms_per_step": 1e3 * t / args.steps}
            res["other_kernels"] = others
        if world == 1 and not args.no_cpu_baseline:
            try:
                if parity is not None:
                    after = logit_parity(model, h, w, synth.BASE_SEED)
                    parity[f"after_{args.warmup + args.steps}_adam_steps"] = after      # same probe on the trained weights
                res["parity"] = parity
                res["cpu_baseline"] = cpu_baseline(model, h, w, synth.BASE_SEED)
            except Exception as e:  # the bench line must still be printed
                log(f"cpu_baseline failed: {e!r}")
                res["cpu_baseline"] = None
        if world == 1 and not args.no_native40:
            try:
                del model, aud, vis, lab
                torch.cuda.empty_cache()
                if args.dtype == "bf16":
                    # the reference is fp32: the same workload on the fp32 matrix cores, for the fp32-MFMA roofline
                    res["roofline_fp32_path"] = fp32_path_roofline(dev, n, h, w, not args.no_audio, seed)
                    torch.cuda.empty_cache()
                res["native_40x40_loop"] = native40_loop(dev)
            except Exception as e:
                log(f"native 40x40 loop failed: {e!r}")
                res["native_40x40_loop"] = None
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(res), flush=True)
        os.dup2(2, 1)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
