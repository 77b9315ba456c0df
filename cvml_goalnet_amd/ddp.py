"""Data-parallel gradient exchange for the fused train step: one process per GPU, RCCL over xGMI.

The reference is a single process with no collective of any kind; this is new design.
Semantics = standard DDP: every rank is an independent reference process on its shard of the frames
(local BatchNorm statistics, local broadcast-MSE), gradients are averaged: SUM here, the 1/world factor is
folded into the fused Adam (`grad_scale`).

The gradient arena is laid out in gradient-readiness order (avm.AVM._param_specs), so the three buckets are
contiguous slices and each collective is issued the moment its slice is complete in backward:

    bucket 0  fusion.* + audbl.* + linear5.bias      ready first (small)                       all-reduce
    bucket 1  visbl.linear5.weight                   90 % (40x40) .. 99.8 % (224x224) of bytes  all-reduce, or (shard_linear5)
                                                                                                reduce-scatter
    bucket 2  bnorm3/conv3 .. bnorm1/conv1           ready at the very end of backward         all-reduce

`async_op=True` collectives on the nccl (= RCCL) backend run on RCCL's own stream after the kernels already enqueued
on the compute stream, so bucket 1 (5.15 GB at 224x224) travels under the conv data/weight-gradient kernels that
follow it; `finish()` makes the compute stream wait for all three.

`shard_linear5=True` (ZeRO-1 for the one tensor that matters): bucket 1 is reduce-scattered, every rank runs Adam on
its 1/world slice of linear5.weight only (its optimizer state exists only for that slice: 28 B/param of HBM traffic
and 8 B/param of memory become 1/world of that), and the updated slices are all-gathered back into every rank's
arena — asynchronously: the gather overlaps the next step's conv forward and is waited for right before linear5
(`wait_weights`). With precision="bf16" the gather moves the bf16 GEMM copy of the weights (half the bytes); the fp32
master of foreign slices is then stale until `consolidate()` — a collective every rank must call before `state_dict()`
(which raises while slices are stale rather than communicating implicitly). EXPERIMENTAL: exercised on RCCL with one
forced rank (tests/test_gpu_rccl.py) and on two gloo ranks, never yet on RCCL with more than one rank.

Replica consistency: `sync_params()` (run automatically before the first synchronised step) broadcasts rank 0's
parameters, BatchNorm buffers, optimizer state and dropout seed, so ranks that were constructed with different torch
seeds still start from one model; each rank then draws its dropout masks from its own counter-based stream
(seed + rank), or — in global-batch mode — from its rows of the global batch's masks.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

_GOLDEN = 0x9E3779B97F4A7C15


def bucket_slices(specs, arena_numel: int) -> List[Tuple[int, int]]:
    """[(start, stop)] element ranges of the three buckets in the flat gradient arena."""
    w5 = next(s for s in specs if s.name == "visbl.linear5.weight")
    after = min(s.offset for s in specs if s.offset > w5.offset)
    return [(0, w5.offset), (w5.offset, after), (after, arena_numel)]


def shard_bounds(lo: int, hi: int, world: int, rank: int) -> Optional[Tuple[int, int]]:
    """rank's slice of [lo, hi) when it splits into `world` equal parts of a multiple of 64 elements (256 B: the fused
    Adam's float4 / bf16x4 accesses stay aligned), else None (the caller falls back to all-reduce + replicated Adam)"""
    n = hi - lo
    if world < 1 or n % (world * 64) or lo % 64:
        return None
    per = n // world
    return lo + rank * per, lo + (rank + 1) * per


class GradSync:
    """Plugged into AVM.grad_sync; called by AVM.train_step between backward and Adam."""

    def __init__(self, process_group=None, compress=None, average=True, shard_linear5=False, broadcast_params=True):
        """compress="bf16": bucket 1 travels as bf16 (summed in bf16 by RCCL) — an extension, off by default
        (fp32 exchange, exact sum); ignored with shard_linear5.
        average=False: the summed gradient is used as it is (global-batch mode, see SyncStats).
        shard_linear5: reduce-scatter + sharded Adam + all-gather for linear5.weight (see the module docstring).
        broadcast_params: make every rank start from rank 0's model (sync_params) before the first exchange."""
        if compress not in (None, "bf16"):
            raise ValueError("compress must be None or 'bf16'")
        self.compress = compress
        self.average = average
        self.shard_linear5 = bool(shard_linear5)
        self.broadcast_params = broadcast_params
        self._packed = None
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        # GOALNET_DDP_FORCE=1: issue the collectives even with one rank (exercises the RCCL path on a 1-GPU box)
        self.force = os.environ.get("GOALNET_DDP_FORCE") == "1" and dist.is_initialized()
        self._work = []
        self._gather_work = None
        self.params_synced = False
        self.master_stale = False          # bf16 + shard_linear5: foreign slices of the fp32 linear5.weight are out of date
        self.timing = None                 # bench: {"exposed_ms": [(event, event), ...]} when not None

    # ---- helpers ------------------------------------------------------------------------------------------------
    @property
    def active(self) -> bool:
        return self.world > 1 or self.force

    def _nccl(self) -> bool:
        return dist.get_backend(self.group) == "nccl"

    def sharded(self, model) -> bool:
        return self.shard_linear5 and self.active and self.shard_range(model) is not None

    def shard_range(self, model) -> Optional[Tuple[int, int]]:
        lo, hi = bucket_slices(model._specs, model._arena_numel)[1]
        return shard_bounds(lo, hi, self.world, self.rank)

    # ---- replica consistency ----------------------------------------------------------------------------------------
    def sync_params(self, model) -> None:
        """Every rank takes rank 0's parameters, BatchNorm buffers, Adam state / step count and dropout seed. A model
        whose Lazy parameters were materialised from each process's own torch RNG (`AVM._materialize`) would otherwise
        apply the averaged gradients to diverging replicas without any error."""
        self.params_synced = True
        if not dist.is_initialized() or self.world == 1:
            return
        src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        # the tensor list below must be the SAME on every rank (a broadcast a rank does not join is a deadlock): "does the model
        # carry optimizer state" is a per-rank fact, so rank 0's answer is broadcast first and every rank follows it
        if model._state is None:
            model._make_state()
        meta = torch.tensor([model.dropout_seed, model._adam_t, model._drop_step] +
                            [int(getattr(model.visbl, f"bnorm{i}").num_batches_tracked) for i in (1, 2, 3)] +
                            [int(model._adam_m is not None and not self.sharded(model))],
                            dtype=torch.int64, device=model._arena.device)
        dist.broadcast(meta, src=src, group=self.group)
        seed0, adam_t, drop_step, *nbt, has_opt = [int(v) for v in meta.tolist()]
        tensors = [model._arena, model._state[0:2]]         # the device counters themselves: applied Adam steps, dropout draws
        for i in (1, 2, 3):
            bn = getattr(model.visbl, f"bnorm{i}")
            tensors += [bn.running_mean, bn.running_var]
        if has_opt:
            model._adam_state()                             # a rank without moments allocates them, then receives rank 0's
            tensors += [model._adam_m, model._adam_v]
        elif model._adam_m is not None and not self.sharded(model):
            model._adam_m = model._adam_v = model._adam_segs = None        # rank 0 starts without optimizer state: so does this rank
        for t in tensors:
            dist.broadcast(t, src=src, group=self.group)
        model._adam_t, model._drop_step = adam_t, drop_step
        for i, v in zip((1, 2, 3), nbt):
            getattr(model.visbl, f"bnorm{i}").num_batches_tracked.fill_(v)
        # dropout: standard DDP = every rank an independent process -> an independent stream per rank, derived from rank 0's
        # seed; global-batch mode keeps ONE seed and draws this rank's rows of the global masks (AVM._masks)
        model.dropout_seed = seed0 if model.stat_sync is not None else (seed0 + _GOLDEN * self.rank) % (1 << 63)
        model._load_count += 1            # the arena was written behind the Parameters' back: bf16 shadow must be re-made

    def ensure_params_synced(self, model) -> None:
        if not self.params_synced:
            if self.broadcast_params and self.active:
                self.sync_params(model)
            self.params_synced = True

    # ---- gradient exchange ---------------------------------------------------------------------------------------------
    def on_bucket(self, model, k: int):
        if not self.active:
            return
        lo, hi = bucket_slices(model._specs, model._arena_numel)[k]
        g = model._garena[lo:hi]
        if k == 1 and self.sharded(model):
            slo, shi = self.shard_range(model)
            out = model._garena[slo:shi]                       # in place: the reduced slice lands where Adam reads it
            if g.is_cuda and self._nccl():
                self._work.append(dist.reduce_scatter_tensor(out, g, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            else:                                              # gloo (CPU tests): no reduce-scatter; same result for the slice
                self._work.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return
        if self.compress == "bf16" and k == 1 and g.is_cuda and (hi - lo) % 8 == 0:
            from . import ops
            self._packed = (ops.cast_bf16(g, torch.empty(hi - lo, dtype=torch.bfloat16, device=g.device)), g)
            g = self._packed[0]
        self._work.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self, model) -> float:
        ev = None
        if self.timing is not None and self._work and torch.cuda.is_available():
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for w in self._work:
            w.wait()
        self._work = []
        if self._packed is not None:
            from . import ops
            ops.cast_f32(*self._packed)
            self._packed = None
        if ev is not None:
            ev[1].record()
            self.timing.setdefault("exposed_ms", []).append(ev)
        return 1.0 / self.world if self.average else 1.0

    # ---- sharded linear5.weight: gather the updated slices ------------------------------------------------------------
    def _all_gather(self, full: torch.Tensor, mine: torch.Tensor):
        if full.is_cuda and self._nccl():
            return dist.all_gather_into_tensor(full, mine, group=self.group, async_op=True)
        parts = list(full.view(self.world, -1).unbind(0))     # gloo: per-rank output views (rank's own view aliases `mine`)
        return dist.all_gather(parts, mine.clone(), group=self.group, async_op=True)

    def after_adam(self, model, shadow: Optional[torch.Tensor]) -> None:
        """Adam has updated this rank's slice of linear5.weight (and, when `shadow` is given, the bf16 copy of that slice):
        start gathering everybody's slices. fp32 precision gathers the master weights; bf16 precision gathers the bf16
        GEMM copy (half the bytes) and leaves the foreign fp32 slices stale."""
        lo, hi = bucket_slices(model._specs, model._arena_numel)[1]
        slo, shi = self.shard_range(model)
        if shadow is not None:
            self._gather_work = self._all_gather(shadow, shadow[slo - lo:shi - lo])
            self.master_stale = True
        else:
            self._gather_work = self._all_gather(model._arena[lo:hi], model._arena[slo:shi])

    def wait_weights(self) -> None:
        """make the current stream wait for the in-flight all-gather of linear5.weight (called right before its first reader)"""
        if self._gather_work is not None:
            self._gather_work.wait()
            self._gather_work = None

    def consolidate(self, model) -> None:
        """Call on EVERY rank before `state_dict()`, before a stock optimizer / in-place edit touches linear5.weight, or before
        switching the exchange mode: waits for the in-flight weight all-gather and (16-bit modes with shard_linear5) all-gathers
        the fp32 master slices. A collective — `AVM.state_dict()` refuses to run while slices are stale instead of issuing it."""
        self.gather_master(model)

    def gather_master(self, model) -> None:
        """bf16 + shard_linear5: bring the fp32 master of linear5.weight up to date on every rank (collective)."""
        self.wait_weights()
        if not self.master_stale:
            return
        lo, hi = bucket_slices(model._specs, model._arena_numel)[1]
        slo, shi = self.shard_range(model)
        stamp_ok = model._w5b is not None and model._w5b_version == model._w5_version()
        self._all_gather(model._arena[lo:hi], model._arena[slo:shi]).wait()
        self.master_stale = False
        if stamp_ok:
            model._w5b_version = model._w5_version()          # the bf16 copy already holds bf16(master) everywhere


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
        vectors in rank order: the loss is the global one on every rank, dL/dp carries 2/N_global and the global mean label;
      * draws rows [rank * n, (rank + 1) * n) of the dropout masks one process would draw for the N_global rows.

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


def enable_global_batch(model, process_group=None, compress=None, shard_linear5=False):
    """Switch `model` (an AVM) to global-batch semantics: SyncStats + summed gradients."""
    model.stat_sync = SyncStats(process_group)
    model.grad_sync = GradSync(process_group, compress=compress, average=False, shard_linear5=shard_linear5)
    return model
