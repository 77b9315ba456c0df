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

    def __init__(self, process_group=None, compress=None):
        """compress="bf16": bucket 1 (linear5.weight, 90-99.8 % of the bytes) travels as bf16 (summed in bf16 by RCCL) —
        an extension for precision="bf16" runs, off by default (fp32 exchange, exact mean)."""
        if compress not in (None, "bf16"):
            raise ValueError("compress must be None or 'bf16'")
        self.compress = compress
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
        return 1.0 / self.world
