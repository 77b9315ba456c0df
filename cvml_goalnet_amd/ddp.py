"""Data-parallel gradient exchange for the fused train step: one process per GPU, RCCL over xGMI.

The reference is a single process with no collective of any kind; this is new design.
Semantics = standard DDP: every rank is an independent reference process on its shard of the frames
(local BatchNorm statistics, local broadcast-MSE), gradients are averaged: all-reduce(SUM) here, the 1/world
factor is folded into the fused Adam (`grad_scale`).

The gradient arena is laid out in gradient-readiness order (avm.AVM._param_specs), so the three buckets are
contiguous slices and each all-reduce is issued the moment its slice is complete in backward:

    bucket 0  fusion.* + audbl.* + linear5.bias      ready first (small)
    bucket 1  visbl.linear5.weight                   90 % (40x40) .. 99.8 % (224x224) of all bytes
    bucket 2  bnorm3/conv3 .. bnorm1/conv1           ready at the very end of backward

`torch.distributed.all_reduce(async_op=True)` on the nccl (= RCCL) backend runs on RCCL's own stream after the
kernels already enqueued on the compute stream, so bucket 1 (5.15 GB at 224x224) travels under the conv
data/weight-gradient kernels that follow it; `finish()` makes the compute stream wait for all three.
"""
from __future__ import annotations

import os
from typing import List, Tuple

import torch
import torch.distributed as dist


def bucket_slices(specs, arena_numel: int) -> List[Tuple[int, int]]:
    """[(start, stop)] element ranges of the three buckets in the flat gradient arena."""
    w5 = next(s for s in specs if s.name == "visbl.linear5.weight")
    after = min(s.offset for s in specs if s.offset > w5.offset)
    return [(0, w5.offset), (w5.offset, after), (after, arena_numel)]


class GradSync:
    """Plugged into AVM.grad_sync; called by AVM.train_step between backward and Adam."""

    def __init__(self, process_group=None, compress=None, average=True):
        """compress="bf16": bucket 1 (linear5.weight, 90-99.8 % of the bytes) travels as bf16 (summed in bf16 by RCCL) —
        an extension for precision="bf16" runs, off by default (fp32 exchange, exact mean).
        average=False: the summed gradient is used as it is (global-batch mode, see SyncStats)."""
        if compress not in (None, "bf16"):
            raise ValueError("compress must be None or 'bf16'")
        self.compress = compress
        self.average = average
        self._packed = None
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # GOALNET_DDP_FORCE=1: issue the collectives even with one rank (exercises the RCCL path on a 1-GPU box)
        self.force = os.environ.get("GOALNET_DDP_FORCE") == "1" and dist.is_initialized()
        self._work = []

    def on_bucket(self, model, k: int):
        if self.world == 1 and not self.force:
            return
        lo, hi = bucket_slices(model._specs, model._arena_numel)[k]
        g = model._garena[lo:hi]
        if self.compress == "bf16" and k == 1 and g.is_cuda and (hi - lo) % 8 == 0:
            from . import ops
            self._packed = (ops.cast_bf16(g, torch.empty(hi - lo, dtype=torch.bfloat16, device=g.device)), g)
            g = self._packed[0]
        self._work.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self, model) -> float:
        for w in self._work:
            w.wait()
        self._work = []
        if self._packed is not None:
            from . import ops
            ops.cast_f32(*self._packed)
            self._packed = None
        return 1.0 / self.world if self.average else 1.0


class SyncStats:
    """Opt-in "one process on the global batch" semantics: what couples the frames of a batch in the
    reference is train-mode BatchNorm (`utils.py:154, 159, 164`) and the mean label inside the broadcast MSE
    (`main.py:68, 191`). Plugged into AVM.stat_sync, every rank

      * contributes its per-channel (sum x, sum x^2) of each BatchNorm as ONE row of 2 C doubles
        (`goalnet_partials_sum_f64`) to an all-reduce(SUM) and finalises with the global pixel count — mean, biased
        variance, running statistics (unbiased with the global count) are then those of the whole batch on every rank;
      * does the same with (sum dz, sum dz * xhat) in backward for the dx coefficients, while dgamma / dbeta keep their
        local sums (the gradient all-reduce adds them up like every other parameter gradient);
      * all-gathers the predictions and labels (n floats per rank) and evaluates the (N, N) broadcast MSE on the global
        vectors in rank order: the loss is the global one on every rank, dL/dp carries 2/N_global and the global mean label.

    Gradients are then SUMMED, not averaged (`GradSync(average=False)`), which makes a W-rank step equal to one
    reference process stepping on the concatenation of the W shards (tests/test_gpu_ddp.py). Every rank must hold
    the same number of frames. Six small collectives per step (3 forward, 3 backward) + two all-gathers: latency-bound."""

    def __init__(self, process_group=None):
        if not dist.is_initialized():
            raise RuntimeError("SyncStats needs an initialised torch.distributed process group")
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)

    def all_reduce(self, t: torch.Tensor) -> torch.Tensor:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def gather(self, t: torch.Tensor) -> torch.Tensor:
        """rank-ordered concatenation of a 1-D tensor of equal length on every rank"""
        t = t.reshape(-1).contiguous()
        out = torch.empty(self.world * t.numel(), dtype=t.dtype, device=t.device)
        if dist.get_backend(self.group) == "nccl":
            dist.all_gather_into_tensor(out, t, group=self.group)
        else:                                                           # gloo (tests): per-rank output views
            dist.all_gather(list(out.view(self.world, -1).unbind(0)), t, group=self.group)
        return out


def enable_global_batch(model, process_group=None, compress=None):
    """Switch `model` (an AVM) to global-batch semantics: SyncStats + summed gradients."""
    model.stat_sync = SyncStats(process_group)
    model.grad_sync = GradSync(process_group, compress=compress, average=False)
    return model
