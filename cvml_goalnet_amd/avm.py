"""`AVM` — the drop-in for the reference's hot path, running on one MI355X.

Mirrors `/root/reference/utils.py:229-272` (class AVM) and its callers' contract:

    model = AVM(audio_included)                       # main.py:64, 325; baseline.py:63
    optim.Adam(model.parameters(), lr=1e-3)           # main.py:70 — BEFORE the first forward (Lazy semantics)
    out = model(audio_input, visual_input)            # audio first; (N,30,B) / list of None, (N,3,H,W) -> (N,1)
    nn.MSELoss()(out, labels).backward(); optimizer.step()      # main.py:187-193
    torch.save(model.state_dict(), f) / model.load_state_dict(torch.load(f))   # main.py:66, 263, 282

Same sub-module names (`visbl`, `audbl`, `fusion`) and therefore the same `state_dict` keys, shapes and
torch-native layouts as the reference. The model is always in train mode, as the reference's is (it never
calls `.eval()`): BatchNorm uses batch statistics and updates its running buffers on every forward, also
under `torch.no_grad()`; dropout is live.

All arithmetic runs in libgoalnet_hip.so (hand-written gfx950 kernels) — there is no CPU or eager-PyTorch
fallback. Parameters live in ONE flat fp32 arena in device layouts (conv weights OHWI, linear5 columns in
NHWC-flatten order); each `nn.Parameter` is a strided view of it with the reference's logical shape, so
stock `torch.optim.Adam` works on them, gradients are written straight into a parallel gradient arena,
and the fused Adam / the DDP all-reduce are single passes over flat memory.

Besides the drop-in surface there is a device-resident fast path, `train_step()`, that runs forward,
the broadcast MSE, backward, (optional) gradient all-reduce and the fused Adam without leaving the GPU.
"""
from __future__ import annotations

import math
import os
from typing import List, Optional

import torch
import torch.nn as nn
from torch.nn.parameter import UninitializedParameter

from . import ops
from ._lib import GoalnetError
from .synth import BASE_SEED, DROP_P, TID_DROP

F32 = torch.float32
BN_EPS = 1e-5
BN_MOMENTUM = 0.1
_ALIGN = 64  # arena slots are multiples of 64 floats (256 B)


class _Holder(nn.Module):
    """A sub-module that only owns parameters/buffers (keeps the reference's state_dict key prefixes)."""

    def forward(self, *a, **k):  # pragma: no cover
        raise GoalnetError("sub-modules of the MI355X AVM are parameter holders; call the AVM itself")


def _mk_layer(bias=True, bn_channels=0):
    h = _Holder()
    h.weight = UninitializedParameter()
    if bias:
        h.bias = UninitializedParameter()
    if bn_channels:
        h.register_buffer("running_mean", None)
        h.register_buffer("running_var", None)
        h.register_buffer("num_batches_tracked", None)
    return h


class _Fork:
    """Fork / join onto a second HIP stream for work that is off the critical path of a SMALL step (the reference's 10-frame
    sub-batches, main.py:177-196): there a step is ~80 short kernels in one dependency chain, most of them too small to fill
    256 CUs, and the weight-gradient / bias-gradient / AudBl kernels do not feed that chain. Run on a side stream they execute
    under it; inside a captured HIP graph the fork and join become graph edges (no runtime cost). Results are unchanged (same
    kernels, same order within each dependency chain). Tensors the side stream reads are kept alive until the join, so the
    caching allocator cannot hand their memory to the main stream while the side kernels are still running."""

    def __init__(self, model, enabled):
        self.enabled = enabled
        self.keep = []
        if enabled:
            if model._side_stream is None:
                model._side_stream = torch.cuda.Stream(device=model._device)
            self.side = model._side_stream
            self.forked = False

    def run(self, fn, *tensors):
        """fn() on the side stream, after everything enqueued on the current stream so far"""
        if not self.enabled:
            return fn()
        self.keep.extend(t for t in tensors if t is not None)
        self.side.wait_stream(torch.cuda.current_stream())
        self.forked = True
        with torch.cuda.stream(self.side):
            return fn()

    def run_marked(self, fn, *tensors):
        """run(fn) and return an event recorded right behind it on the side stream (None when not forking): `wait(event)` makes
        the current stream wait for THAT piece of side work only, not for everything the side stream holds"""
        r = self.run(fn, *tensors)
        if not self.enabled:
            return r, None
        ev = torch.cuda.Event()
        ev.record(self.side)
        return r, ev

    @staticmethod
    def wait(ev):
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def join(self):
        if self.enabled and self.forked:
            torch.cuda.current_stream().wait_stream(self.side)
            self.forked = False
        self.keep.clear()


class _Spec:
    __slots__ = ("name", "kind", "shape", "numel", "offset", "fan_in")

    def __init__(self, name, kind, shape, fan_in):
        self.name, self.kind, self.shape, self.fan_in = name, kind, tuple(shape), fan_in
        self.numel = int(math.prod(shape))
        self.offset = 0


class AVM(nn.Module):
    def __init__(self, audio_included, device=None, seed: Optional[int] = None, precision: str = "fp32", head: str = "regression",
                 num_classes: int = 5):
        """`seed`: seed of the counter-based dropout stream. None (default) draws it from the torch RNG at construction, so
        dropout follows `torch.manual_seed` as the reference's does (utils.py:170, 245-254 use the global torch RNG) and two
        model instances / two runs do not replay the same masks; parity tests pass synth.BASE_SEED to regenerate the masks
        from the seed formula on the CPU side."""
        super().__init__()
        self.audio_included = audio_included                      # utils.py:235
        # head="classifier" (EXTENSION): the reference's commented-out variant — Linear(128 -> C), Softmax(dim=1) in place of
        # the Sigmoid (utils.py:257), then 4y+1 (utils.py:270); trained with CrossEntropyLoss on labels-1 (main.py:69, 189),
        # predictions = argmax + 1 (main.py:190). forward returns the (N, C) class scores.
        if head not in ("regression", "classifier"):
            raise ValueError("head must be 'regression' (the reference's live code) or 'classifier' (its commented-out variant)")
        if not 2 <= num_classes <= 8:
            raise ValueError("num_classes must be in 2..8")
        self.head = head
        self.num_classes = num_classes if head == "classifier" else 1
        if precision not in ("fp32", "bf16", "fp16", "bf16x6", "fp16x3"):
            raise ValueError("precision must be 'fp32' (the reference's arithmetic), 'bf16x6' / 'fp16x3' (fp32 operands as bf16 triples / scaled "
                             "fp16 pairs on the 16-bit MFMA: fp32-grade), 'bf16' or 'fp16' (16-bit MFMA contractions)")
        # "bf16x6": everything is stored and computed as under "fp32" except the large 3 x 3 convolutions (conv2 forward; conv3 forward,
        # data gradient and weight gradient at >= 65 536 output pixels), whose fp32 operands are split into bf16 triples hi + mid + lo
        # (exact) and multiplied as six partial products on the 16-bit MFMA with fp32 accumulation (csrc/split3.hip): fp32-grade
        # results (3-4 x the rounding error of the fp32 MFMA's own accumulation) at 1.4-1.65 x its speed.
        # "fp16x3": the same GEMMs on fp16 PAIRS hi + mid of each value scaled by a per-tensor power of two (22 significand bits; the scale
        # comes from a magnitude pass over the tensor, so no loss scale is involved) and three partial products: half the MFMA work.
        # "bf16" / "fp16": the dense contractions (conv2/conv3 forward + data / weight gradient, linear5) run on the 16-bit matrix
        # cores with fp32 accumulation; statistics, master weights, parameter gradients and Adam stay fp32.
        # fp16 keeps 11 significand bits against bf16's 8 (~8 x less rounding noise in the logits) but has 5 exponent bits:
        # the activation gradients it stores need a loss scale (loss_scale, below) and an overflow guard.
        # dL/dpred is multiplied by the loss scale before backward, the fused Adam divides it out again (a power of two: exact).
        # None (fp16 default) = 2^(10 + ceil(log2 n) - backoff) for a step of n frames: dL/dpred is O(1/n) and the 16-bit activation
        # gradients measured with scripts/grad_ranges.py (medians 6e-10 .. 1e-8, maxima 2e-7 .. 8e-6 at n = 1 024; ~1000 x that at
        # n = 10) then sit at 6e-4 .. 8 — inside binary16's normal range [6.1e-5, 65504] with four orders of magnitude of headroom.
        # The precision setter picks the default (None for fp16, 1.0 otherwise) unless the user assigned `loss_scale` explicitly.
        self._loss_scale, self._loss_scale_user = 1.0, False
        self._loss_scale_backoff = 0   # halvings of the automatic scale (update_loss_scale: one per check that found skipped steps)
        self._skipped_seen = 0
        self.precision = precision     # property: also sets _half / _h16 and the default loss scale
        self._guard = None             # fp16: int64[2] device counters — step stamped as overflowed, number of skipped updates
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
        self._device = torch.device(device) if device is not None else None

        self.visbl = _Holder()                                     # utils.py:237
        self.visbl.conv1 = _mk_layer()
        self.visbl.bnorm1 = _mk_layer(bn_channels=64)
        self.visbl.conv2 = _mk_layer()
        self.visbl.bnorm2 = _mk_layer(bn_channels=256)
        self.visbl.conv3 = _mk_layer()
        self.visbl.bnorm3 = _mk_layer(bn_channels=512)
        self.visbl.linear5 = _mk_layer()
        if audio_included:                                         # utils.py:239-240
            self.audbl = _Holder()
            self.audbl.conv1 = _mk_layer()
            self.audbl.conv2 = _mk_layer()
            self.audbl.linear3 = _mk_layer()
        self.fusion = nn.ModuleDict({k: _mk_layer() for k in ("0", "3", "6", "9", "12")})   # utils.py:242-256

        # dropout (utils.py:170, 245-254): "device" = counter-based masks (synth.make_drop_masks formula),
        # "off" = p := 0, "given" = masks supplied through set_dropout_masks() (parity tests)
        self.dropout_mode = "device"
        self.dropout_seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if seed is None else int(seed)
        self.dropout_row_offset = 0    # rows in front of this rank's rows in the global batch (ddp.enable_global_batch)
        self._drop_step = 0
        self._given_masks: Optional[List[torch.Tensor]] = None

        self._specs: List[_Spec] = []
        self._arena = self._garena = self._adam_m = self._adam_v = None
        self._hw3 = self._l2 = None
        self._adam_t = 0
        self._adam_segs = None
        self._arena_grad_scale = 1.0       # the gradient arena currently holds this factor x gradient (fp16 train_step)
        self._w5b, self._w5b_version = None, None      # bf16 shadow of visbl.linear5.weight and the version stamps it matches
        self._load_count = 0                           # bumped by load_state_dict (its layout kernels write the arena directly)
        self._state = None             # int64[4] device counters: adam step, dropout draw, frame cursor, sub-batch index
        self._defer_tick = False       # inside train_step: counters advance once, at the end (one launch)
        self._pending_drop_tick = 0
        self._materialized = False
        self.grad_sync = None          # optional ddp.GradSync: gradient exchange between backward and Adam
        self.stat_sync = None          # optional ddp.SyncStats: BatchNorm sums and the loss over all ranks' frames
        self.grad_bf16 = os.environ.get("GOALNET_DZ16", "1") != "0"   # precision="bf16": bf16 BatchNorm-output gradients (backward_device)
        self.act_bf16 = os.environ.get("GOALNET_P16", "1") != "0"     # precision="bf16": pooled activations of blocks 2, 3 stored as bf16
        self.keep_ctx = False          # tests: keep the last train_step's saved tensors in last_ctx
        self.last_ctx = None
        self.last_used_w5b = False
        self._side_stream = None
        self._adam_stream = None
        self._dbias_pending = {}
        self._fused_loss, self._fused_loss_done = None, False      # train_step -> forward_device: (labels, loss, dout) for the fused MLP launch
        self._fork = _Fork(self, False)
        self.overlap_rows = int(os.environ.get("GOALNET_OVERLAP_ROWS", "64"))   # steps of <= this many frames fork their off-path work
        # Forking at LARGE sizes: off-path work (weight gradients, bias sums, AudBl) on the side stream AND the fused Adam over linear5.weight
        # there as soon as its gradient exists (train_step's "early Adam") — HBM-bound passes under MFMA-bound GEMMs and the other way
        # round. Bit-identical. Measured at 1 024 frames @224 (round 3, alternating runs on one box): bf16 73.7 -> 72.2 ms, fp16x3 182.7 ->
        # 178.6 ms, bf16x6 297.2 -> 295.6 ms, fp32 417.6 -> 428.7 ms (its GEMMs are 90 % of the step and gain nothing from company); at
        # 10 frames @40 845 -> 974 us (the 0.66 GB Adam stream slows every small kernel it runs beside). Hence the automatic rule: on for
        # the 16-bit and split-operand precisions at >= 256 frames, off otherwise. GOALNET_OVERLAP_LARGE=1 forces it at every size and
        # precision, =0 switches it off.
        _ol = os.environ.get("GOALNET_OVERLAP_LARGE")
        self.overlap_large = _ol == "1"
        self.overlap_auto = _ol is None
        self.time_labels = None        # bench: with kernel_events set, time only these labels (None = all, and no forking)
        self.kernel_events = None      # bench: {label: [(start_event, end_event, flops), ...]} when not None

    @property
    def precision(self) -> str:
        return self._precision

    @precision.setter
    def precision(self, value: str):
        if value not in ("fp32", "bf16", "fp16", "bf16x6", "fp16x3"):
            raise ValueError("precision must be 'fp32', 'bf16x6', 'fp16x3', 'bf16' or 'fp16'")
        self._precision = value
        self._half = value in ("bf16", "fp16")
        self._parts = {"bf16x6": 3, "fp16x3": 2}.get(value, 0)                # split-operand GEMMs (csrc/split3.hip); fp32 storage
        self._x6 = self._parts > 0
        self._h16 = torch.float16 if value in ("fp16", "fp16x3") else torch.bfloat16     # the 16-bit storage format of the GEMM operands
        self._w5b, self._w5b_version = None, None                              # a copy in the other format is not reusable
        self._padbufs, self._padgen = {}, {}                                   # nor are the cached padded 16-bit operand buffers
        if not self._loss_scale_user:
            # binary16's 5 exponent bits need the scale (its activation gradients would flush to zero without it, and the overflow
            # guard cannot see an underflow); bf16 / fp32 do not
            self._loss_scale = None if value == "fp16" else 1.0
            self._loss_scale_backoff = 0

    @property
    def loss_scale(self):
        return self._loss_scale

    @loss_scale.setter
    def loss_scale(self, value):
        """an explicit scale (a power of two keeps the unscale exact), or None = the automatic per-size scale of precision="fp16"; an
        explicit value survives later changes of `precision`"""
        self._loss_scale = None if value is None else float(value)
        self._loss_scale_user = value is not None
        self._loss_scale_backoff = 0

    # ------------------------------------------------------------------------------------------
    # parameter arena
    # ------------------------------------------------------------------------------------------
    def _param_specs(self, hw3: int, l2: int) -> List[_Spec]:
        """Arena order = gradient-readiness order in backward: [fusion, audbl, linear5.bias | linear5.weight |
        rest of visbl], so each DDP bucket is one contiguous slice (ddp.py)."""
        f0_in = 640 if self.audio_included else 512
        S = []

        def lin(name, out, inn, kind="plain"):
            S.append(_Spec(name + ".weight", kind, (out, inn), inn))
            S.append(_Spec(name + ".bias", "plain", (out,), inn))

        lin("fusion.12", self.num_classes, 128); lin("fusion.9", 128, 256); lin("fusion.6", 256, 512)
        lin("fusion.3", 512, 512); lin("fusion.0", 512, f0_in)
        if self.audio_included:
            lin("audbl.linear3", 128, 128 * l2)
            S.append(_Spec("audbl.conv2.weight", "plain", (128, 64, 3), 192)); S.append(_Spec("audbl.conv2.bias", "plain", (128,), 192))
            S.append(_Spec("audbl.conv1.weight", "plain", (64, 30, 3), 90)); S.append(_Spec("audbl.conv1.bias", "plain", (64,), 90))
        S.append(_Spec("visbl.linear5.bias", "plain", (512,), 512 * hw3))
        S.append(_Spec("visbl.linear5.weight", "lin5", (512, 512, hw3), 512 * hw3))      # logical (512, C, HW)
        for i, (co, ci) in ((3, (512, 256)), (2, (256, 64)), (1, (64, 3))):
            S.append(_Spec(f"visbl.bnorm{i}.weight", "bn_w", (co,), 0)); S.append(_Spec(f"visbl.bnorm{i}.bias", "bn_b", (co,), 0))
            S.append(_Spec(f"visbl.conv{i}.weight", "ohwi", (co, ci, 3, 3), ci * 9)); S.append(_Spec(f"visbl.conv{i}.bias", "plain", (co,), ci * 9))
        off = 0
        for s in S:
            s.offset = off
            off += (s.numel + _ALIGN - 1) // _ALIGN * _ALIGN
        self._arena_numel = off
        return S

    def _view(self, arena, s: _Spec):
        flat = arena[s.offset:s.offset + s.numel]
        if s.kind == "ohwi":
            o, i = s.shape[0], s.shape[1]
            return flat.view(o, 3, 3, i).permute(0, 3, 1, 2)          # logical OIHW over physical OHWI
        if s.kind == "lin5":
            o, c, hw = s.shape
            return flat.view(o, hw, c).permute(0, 2, 1)               # logical (512, C, HW) over physical (512, HW, C)
        return flat.view(s.shape)

    def _module_of(self, name):
        mod = self
        parts = name.split(".")
        for p in parts[:-1]:
            mod = mod[p] if isinstance(mod, nn.ModuleDict) else getattr(mod, p)
        return mod, parts[-1]

    def _require_device(self):
        if self._device is None or self._device.type != "cuda":
            raise GoalnetError("the MI355X AVM needs a GPU (torch.cuda.is_available() is False); there is no CPU fallback")
        ops.lib()

    def _make_state(self):
        """Device counters read by the graph-capturable kernels (csrc/stepstate.hip); host mirrors: _adam_t, _drop_step."""
        self._require_device()
        self._state = torch.tensor([self._adam_t, self._drop_step, 0, 0], dtype=torch.int64, device=self._device)
        self._guard = torch.zeros(2, dtype=torch.int64, device=self._device)

    def _materialize(self, hw3: int, l2: int, init: bool = True):
        """Fix the Lazy shapes (first forward or load_state_dict) and build the arena. Parameters are
        materialised IN PLACE so an optimizer created earlier keeps valid references (main.py:70)."""
        if self._materialized:
            if hw3 != self._hw3 or (self.audio_included and l2 != self._l2):
                raise RuntimeError(f"input size changed after materialisation: linear5 expects {512 * self._hw3} features "
                                   f"(got {512 * hw3}), audbl.linear3 {128 * (self._l2 or 0)} (got {128 * l2})")
            return
        self._require_device()
        if self._state is None:
            self._make_state()
        self._hw3, self._l2 = hw3, l2
        self._specs = self._param_specs(hw3, l2)
        dev = self._device
        self._arena = torch.zeros(self._arena_numel, dtype=F32, device=dev)
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if init else 0
        for k, s in enumerate(self._specs):
            mod, leaf = self._module_of(s.name)
            prm = getattr(mod, leaf)
            view = self._view(self._arena, s)
            if init:
                flat = self._arena[s.offset:s.offset + s.numel]
                if s.kind == "bn_w":
                    flat.fill_(1.0)
                elif s.kind != "bn_b":
                    bound = 1.0 / math.sqrt(s.fan_in)                 # torch default init, SURVEY.md §8(a) row 1
                    ops.fill_uniform(flat, seed, k, -bound, bound)
            prm.data = view
            if isinstance(prm, UninitializedParameter):
                prm.__class__ = prm.cls_to_become                     # what UninitializedParameter.materialize does
        for i, c in ((1, 64), (2, 256), (3, 512)):
            bn = getattr(self.visbl, f"bnorm{i}")
            bn.running_mean = torch.zeros(c, dtype=F32, device=dev)
            bn.running_var = torch.ones(c, dtype=F32, device=dev)
            bn.num_batches_tracked = torch.zeros((), dtype=torch.int64)   # host counter (no launch per forward)
        self._materialized = True

    def spec(self, name) -> _Spec:
        for s in self._specs:
            if s.name == name:
                return s
        raise KeyError(name)

    def _pflat(self, name):
        s = self.spec(name)
        return self._arena[s.offset:s.offset + s.numel]

    def _gflat(self, name):
        s = self.spec(name)
        return self._garena[s.offset:s.offset + s.numel]

    def _ensure_garena(self):
        if self._garena is None:
            self._garena = torch.zeros(self._arena_numel, dtype=F32, device=self._device)

    # ------------------------------------------------------------------------------------------
    # state_dict interchange in the reference's torch-native layouts (main.py:66, 263, 282, 326)
    # ------------------------------------------------------------------------------------------
    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        if not self._materialized:
            raise RuntimeError("state_dict() before the first forward / load_state_dict(): parameters are uninitialised")
        if self.grad_sync is not None:
            # ddp.GradSync(shard_linear5=True) in a 16-bit mode leaves the fp32 master of the other ranks' slices of linear5.weight
            # stale between steps. Bringing it up to date is a COLLECTIVE: state_dict() does not issue it behind the caller's back
            # (`if rank == 0: torch.save(model.state_dict())` would hang) — every rank calls grad_sync.consolidate(model) first.
            if self.grad_sync.master_stale:
                raise GoalnetError("state_dict(): foreign slices of visbl.linear5.weight's fp32 master are stale (GradSync(shard_linear5=True) "
                                   "in a 16-bit mode); call model.grad_sync.consolidate(model) on EVERY rank first (a collective)")
            self.grad_sync.wait_weights()                # fp32 + sharded: the in-flight all-gather of the updated weights (local wait)
        out = {} if destination is None else destination
        order = self._reference_key_order()
        for name in order:
            if name.endswith(("running_mean", "running_var", "num_batches_tracked")):
                mod, leaf = self._module_of(name)
                out[prefix + name] = getattr(mod, leaf).detach().cpu().clone()
                continue
            s = self.spec(name)
            flat = self._arena[s.offset:s.offset + s.numel]
            if s.kind == "ohwi":
                o, i = s.shape[0], s.shape[1]
                t = torch.empty(s.numel, dtype=F32, device=self._device)
                ops.transpose_inner(flat, t, o, 9, i)                  # [O][9][I] -> [O][I][9]
                t = t.view(o, i, 3, 3)
            elif s.kind == "lin5":
                o, c, hw = s.shape
                t = torch.empty(s.numel, dtype=F32, device=self._device)
                ops.transpose_inner(flat, t, o, hw, c)                 # [512][HW][C] -> [512][C][HW]
                t = t.view(o, c * hw)
            else:
                t = flat.view(s.shape).clone()
            out[prefix + name] = t.cpu()
        return out

    def _reference_key_order(self):
        from .synth import PARAM_ORDER
        keys = []
        for name in PARAM_ORDER:
            if name.startswith("audbl.") and not self.audio_included:
                continue
            keys.append(name)
            if ".bnorm" in name and name.endswith(".bias"):
                base = name.rsplit(".", 1)[0]
                keys += [base + ".running_mean", base + ".running_var", base + ".num_batches_tracked"]
        return keys

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        w5 = state_dict["visbl.linear5.weight"]
        if w5.dim() != 2 or w5.shape[0] != 512 or w5.shape[1] % 512:
            raise RuntimeError(f"visbl.linear5.weight has shape {tuple(w5.shape)}, expected (512, 512*H3*W3)")
        hw3 = w5.shape[1] // 512
        l2 = state_dict["audbl.linear3.weight"].shape[1] // 128 if self.audio_included else 0
        self._load_count += 1
        self._materialize(hw3, l2, init=False)
        if self.grad_sync is not None:
            # every slice of the master is about to be overwritten from the file (load_state_dict runs on all ranks, as in the
            # reference's single process): an in-flight gather must land first, and nothing is stale afterwards
            self.grad_sync.wait_weights()
            self.grad_sync.master_stale = False
        expected = set(self._reference_key_order())
        missing = sorted(expected - set(state_dict.keys()))
        unexpected = sorted(set(state_dict.keys()) - expected)
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing keys {missing}, unexpected keys {unexpected}")
        with torch.no_grad():
            for name in expected & set(state_dict.keys()):
                src = state_dict[name]
                if name.endswith(("running_mean", "running_var", "num_batches_tracked")):
                    mod, leaf = self._module_of(name)
                    getattr(mod, leaf).copy_(src)
                    continue
                s = self.spec(name)
                logical = (s.shape[0], s.shape[1] * s.shape[2]) if s.kind == "lin5" else s.shape
                if tuple(src.shape) != tuple(logical):
                    raise RuntimeError(f"size mismatch for {name}: {tuple(src.shape)} vs {tuple(logical)}")
                flat = self._arena[s.offset:s.offset + s.numel]
                t = src.detach().to(device=self._device, dtype=F32).contiguous().view(-1)
                if s.kind == "ohwi":
                    ops.transpose_inner(t, flat, s.shape[0], s.shape[1], 9)      # [O][I][9] -> [O][9][I]
                elif s.kind == "lin5":
                    ops.transpose_inner(t, flat, s.shape[0], s.shape[1], s.shape[2])  # [512][C][HW] -> [512][HW][C]
                else:
                    flat.copy_(t)
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    # ------------------------------------------------------------------------------------------
    # dropout masks
    # ------------------------------------------------------------------------------------------
    def set_dropout_masks(self, masks: Optional[List[torch.Tensor]]):
        """Parity mode: five (N,width) multiplier tensors [visbl.drop5, fusion.2, .5, .8, .11] used by the next forward."""
        self.dropout_mode = "given" if masks is not None else "off"
        self._given_masks = None if masks is None else [m.to(self._device, F32).contiguous() for m in masks]

    def _masks(self, n: int):
        if self.dropout_mode == "off":
            return [None] * 5
        if self.dropout_mode == "given":
            for m, wdt in zip(self._given_masks, (512, 512, 512, 256, 128)):
                if tuple(m.shape) != (n, wdt):
                    raise RuntimeError(f"dropout mask shape {tuple(m.shape)} != {(n, wdt)}")
            return list(self._given_masks)
        # one launch for the five masks; the draw index is the device counter state[1] (graph-capturable)
        widths = (512, 512, 512, 256, 128)
        buf = torch.empty(n * sum(widths), dtype=F32, device=self._device)
        # global-batch mode: this rank draws rows [rank * n, (rank + 1) * n) of the masks one process would draw for the
        # concatenated batch; standard DDP: an independent stream per rank (ddp.GradSync.sync_params re-seeds)
        row0 = self.stat_sync.rank * n if self.stat_sync is not None else self.dropout_row_offset
        out = ops.dropout_masks_dev(buf, n, widths, self.dropout_seed, TID_DROP, 8, self._state[1], DROP_P, row_offset=row0)
        if self._defer_tick:
            self._pending_drop_tick = 1          # train_step advances all counters in one launch at its end
        else:
            ops.counter_add(self._state[1], 1)
        self._drop_step += 1
        return out

    # ------------------------------------------------------------------------------------------
    # forward / backward on device tensors
    # ------------------------------------------------------------------------------------------
    def _w5_bf16(self, k5):
        """bf16 copy of visbl.linear5.weight. The fused Adam of train_step refreshes it while it updates the fp32 master
        (ops.adam_step_dev_shadow), so the next forward needs no 7.7 GB cast pass; any other writer (a stock torch optimizer,
        load_state_dict, in-place edits of the Parameter or of the arena) changes `_w5_version()` and the copy is re-made."""
        w5 = self._pflat("visbl.linear5.weight")
        if self.grad_sync is not None and self.grad_sync.master_stale and self._w5b_version != self._w5_version():
            # the 16-bit copy is invalid (something wrote linear5.weight outside the fused step) AND the fp32 master of the other
            # ranks' slices is stale: re-casting would bake stale weights in, and the remedy is a collective that this rank-local
            # condition must not trigger on one rank alone (the others would not join it: deadlock)
            raise GoalnetError("visbl.linear5.weight was written outside the fused step while the other ranks' slices of its fp32 master "
                               "are stale (GradSync(shard_linear5=True), 16-bit mode): call model.grad_sync.consolidate(model) on every "
                               "rank BEFORE editing / optimizer steps outside train_step")
        if self._w5b is None or self._w5b.numel() != w5.numel():
            self._w5b, self._w5b_version = torch.empty(w5.numel(), dtype=self._h16, device=self._device), None
        if self._w5b_version != self._w5_version():
            ops.cast_bf16(w5, self._w5b)
            self._w5b_version = self._w5_version()
        return self._w5b

    def _w5_version(self):
        """version stamps of everything a Python-side writer of linear5.weight goes through: the arena tensor and the
        Parameter (whose `.data` alias has its own counter — that is the one a torch optimizer bumps)"""
        return (self._arena._version, self.visbl.linear5.weight._version, self._load_count)

    def _padbuf(self, key, n, h, w, c):
        """Cached zero-padded bf16 activation buffer (borders/guards zeroed once, interior rewritten every step)."""
        k = (key, n, h, w, c)
        if k not in self._padbufs:
            self._padbufs[k] = ops.padded_bf16_alloc(n, h, w, c, self._device, dtype=self._h16)
        self._padgen[key] = self._padgen.get(key, 0) + 1      # backward checks that its saved operand was not overwritten
        return self._padbufs[k][1]

    def _timed(self, label, flops, fn, *args):
        """Run one kernel launch; when bench.py asked for it, bracket it with HIP events on the launching stream."""
        if self.kernel_events is None or (self.time_labels is not None and label not in self.time_labels):
            return fn(*args)
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*args)
        e1.record()
        self.kernel_events.setdefault(label, []).append((e0, e1, flops))
        return r

    @staticmethod
    def _sizes(h, w):
        h1, w1 = (h + 3) // 3 + 1, (w + 3) // 3 + 1
        return (h1, w1), (h1 - 2, w1 - 2), (h1 - 4, w1 - 4), (h1 - 6, w1 - 6)

    def _global_sums(self, partials, width):
        """ddp.SyncStats: this rank's partial rows -> one row of `width` doubles -> all-reduce(SUM) over the ranks"""
        row = torch.empty(width, dtype=torch.float64, device=self._device)
        ops.partials_sum_f64(partials, width, row)
        return self.stat_sync.all_reduce(row)

    def _fork_ok(self, n):
        """side-stream overlap for this step? Not while every kernel is being timed (bench's instrumented steps: an overlapped
        kernel's bracket would measure its neighbours too)."""
        if self.kernel_events is not None and self.time_labels is None:
            return False
        return n <= self.overlap_rows or self._large_overlap(n)

    def _large_overlap(self, n):
        """fork (and run linear5's Adam early) at this size although it is above overlap_rows? (comment at overlap_large)"""
        return self.overlap_large or (self.overlap_auto and self._precision != "fp32" and n >= 256)

    @staticmethod
    def _bwd16_ok(wc):
        """widths goalnet_bnpool_bwd_bf16p(_t) serves: three ring rows (pooled row + 4 guard pixels) of a 32-channel slice (value +
        argmax byte) in 64 KB of LDS, i.e. conv outputs up to 134 pixels wide (frames up to ~404 px); wider blocks take the fp32 kernels"""
        return 3 * (wc + 2) * 32 * 5 <= 65536

    @staticmethod
    def _p16_ok(wc, c):
        """shapes goalnet_pool_bnstats_fwd_p16 serves (32-channel slices, three conv rows in 64 KB of LDS) and whose bf16
        pooled activation the fused backward can read back"""
        return c % 32 == 0 and 3 * wc * 32 * 4 <= 65536 and AVM._bwd16_ok(wc)

    def _x6_conv(self, m, cout):
        """precision="bf16x6": does this convolution (m output pixels, cout output channels of the GEMM) run on split operands?
        Only where the 256 x 256 tile is filled; everything else runs the fp32-MFMA kernels."""
        return self._x6 and m >= 65536 and cout >= 256 and os.environ.get("GOALNET_X6_OFF") != "1"

    # split operands: every helper returns (parts tensor, magnitude word or None); `_osc(a, b)` = the epilogue scale of a GEMM of the two
    def _amax_of(self, x2d, rows, c, scale=None, shift=None, bnC=0):
        if self._parts != 2:
            return None
        return ops.absmax(x2d, torch.zeros(1, dtype=torch.int32, device=x2d.device), rows, c, scale=scale, shift=shift, bnC=bnC)

    def _osc(self, amax_a, amax_b):
        return ops.split_scales(amax_a, amax_b) if self._parts == 2 else None

    def _split_w(self, w, rows, c):
        am = self._amax_of(w, rows, c)
        return ops.split_rows(self._parts, w, torch.empty(rows * self._parts * c, dtype=self._h16, device=w.device), rows, c, amax=am), am

    def _split_act(self, key, x, scale, shift, n, h, w, c):
        am = self._amax_of(x, n * h * w, c, scale=scale, shift=shift, bnC=c)
        return ops.split_padded(self._parts, x, scale, shift, self._padbuf(key, n, h, w, self._parts * c), n, h, w, c, amax=am), am

    def _split_mat(self, x2d, rows, c, scale=None, shift=None, bnC=0):
        am = self._amax_of(x2d, rows, c, scale=scale, shift=shift, bnC=bnC)
        return ops.split_rows(self._parts, x2d, torch.empty(rows * self._parts * c, dtype=self._h16, device=x2d.device), rows, c,
                              scale=scale, shift=shift, bnC=bnC, amax=am), am

    def _x6_linear5(self, n, k5):
        """precision="bf16x6": linear5's three contractions on split operands (>= 256 frames: the 256 x 256 tile)"""
        return self._x6 and ops.linear_split_ok(self._parts, n, k5, 512) and os.environ.get("GOALNET_X6_OFF") != "1" and \
            os.environ.get("GOALNET_X6_LINEAR5", "1") != "0"

    def _mlp_fused(self, n):
        """the one-launch fusion MLP (csrc/mlp.hip): the regression head at the reference's sub-batch sizes"""
        return n <= 16 and self.head == "regression" and os.environ.get("GOALNET_MLP_FUSED", "1") != "0"

    def _small_bn(self, a, b, n, hc, wc, c):
        """the one-launch pool / BatchNorm kernels (csrc/pool_bn.hip, goalnet_*_fused): fp32 tensors, local statistics, few elements"""
        return (a.dtype == F32 and b.dtype == F32 and self.stat_sync is None and c <= 512 and n * hc * wc * c <= ops.SMALL_BN_ELEMS
                and os.environ.get("GOALNET_SMALL_BN", "1") != "0")

    def _bn_block(self, y, n, hc, wc, c, i, save, p16=False):
        """maxpool + BN statistics of block i on conv output y (N,hc,wc,c). Returns (p, idx, mean, invstd, scale, shift).
        p16: the pooled activation is stored as bf16 (precision="bf16", blocks whose every consumer is a bf16 GEMM pass)."""
        dev = self._device
        assert p16 or y.dtype == F32
        p = torch.empty(n, hc - 2, wc - 2, c, dtype=self._h16 if p16 else F32, device=dev)
        idx = torch.empty(n, hc - 2, wc - 2, c, dtype=torch.uint8, device=dev) if save else None
        bn = getattr(self.visbl, f"bnorm{i}")
        st = torch.empty(4, c, dtype=F32, device=dev)
        count = n * (hc - 2) * (wc - 2)
        if self._small_bn(y, p, n, hc, wc, c):
            # the reference's operating point (10 frames of 40 x 40): pool, statistics AND the finalise step in one launch
            ops.pool_bn_fwd_small(y, p, idx, self._pflat(f"visbl.bnorm{i}.weight"), self._pflat(f"visbl.bnorm{i}.bias"),
                                  bn.running_mean, bn.running_var, BN_MOMENTUM, BN_EPS, st, n, hc, wc, c)
            bn.num_batches_tracked += 1
            return p, idx, st
        # one partial row per (frame, row band): up to 8 bands per frame keep the grid full for small sub-batches
        partials = torch.empty(ops.stat_parts(8 * n) * 2 * c, dtype=torch.float64, device=dev)
        ops.pool_bnstats_fwd(y, p, idx, partials, n, hc, wc, c)
        if self.stat_sync is not None:
            partials, count = self._global_sums(partials, 2 * c), count * self.stat_sync.world
        ops.bn_finalize(partials, self._pflat(f"visbl.bnorm{i}.weight"), self._pflat(f"visbl.bnorm{i}.bias"),
                        bn.running_mean, bn.running_var, BN_MOMENTUM, BN_EPS, count, c,
                        st[0], st[1], st[2], st[3])
        bn.num_batches_tracked += 1
        return p, idx, st

    def forward_device(self, audio, visual, save: bool):
        """audio (N,30,B) / None, visual (N,3,H,W): contiguous fp32 GPU tensors. Returns out (N,) [, ctx]."""
        if visual.dim() != 4 or visual.shape[1] != 3:
            raise RuntimeError(f"visual_input must be (N,3,H,W), got {tuple(visual.shape)}")
        n, _, h, w = visual.shape
        (h1, w1), (hp1, wp1), (hp2, wp2), (hp3, wp3) = self._sizes(h, w)
        if hp3 < 1 or wp3 < 1:
            raise RuntimeError(f"frames of {h}x{w} are too small for VisBl")
        bins = l1 = l2 = 0
        if self.audio_included:
            if audio is None or audio.dim() != 3 or audio.shape[0] != n or audio.shape[1] != 30:
                raise RuntimeError("audio_input must be (N,30,B) when audio_included=True")
            bins = audio.shape[2]
            l1 = (bins - 1) // 2 + 1
            l2 = (l1 - 1) // 2 + 1
        self._materialize(hp3 * wp3, l2)
        if self.grad_sync is not None:
            self.grad_sync.ensure_params_synced(self)      # first synchronised step: every rank starts from rank 0's model
        dev = self._device
        P = self._pflat
        masks = self._masks(n)
        ctx = {"n": n, "h": h, "w": w, "bins": bins, "visual": visual, "audio": audio} if save else None

        fw = 640 if self.audio_included else 512
        voff = fw - 512
        cat = torch.empty(n, fw, dtype=F32, device=dev)          # torch.cat((audio, visual), -1), utils.py:266
        mcat = torch.empty(n, fw, dtype=F32, device=dev) if save else None
        a1 = a2 = None
        ffork = _Fork(self, self._fork_ok(n) and self.audio_included)
        if self.audio_included:
            # ---- AudBl, utils.py:214-227: independent of VisBl until the concatenation -> side stream for small steps (_Fork)
            a1 = torch.empty(n, 64, l1, dtype=F32, device=dev)
            a2 = torch.empty(n, 128, l2, dtype=F32, device=dev)

            def audbl_fwd():
                ops.conv1d_fwd(audio, P("audbl.conv1.weight"), P("audbl.conv1.bias"), a1, True, n, 30, bins, 64)
                ops.conv1d_fwd(a1, P("audbl.conv2.weight"), P("audbl.conv2.bias"), a2, True, n, 64, l1, 128)
                ops.linear_fwd(a2.view(n, 128 * l2), P("audbl.linear3.weight"), P("audbl.linear3.bias"), cat[:, :128], relu=True,
                               mult_out=None if mcat is None else mcat[:, :128])
            ffork.run(audbl_fwd, audio, cat, mcat, a1, a2)

        # ---- VisBl, utils.py:172-195
        y1 = torch.empty(n, h1, w1, 64, dtype=F32, device=dev)
        ops.conv1_fwd(visual, P("visbl.conv1.weight"), P("visbl.conv1.bias"), y1, n, h, w)
        # block 1 keeps its pooled activation in fp32 also under precision="bf16": measured with bf16 p1 the forward logit
        # error more than doubles (MAE 4.7e-5 -> 1.1e-4) and after 7 Adam steps it crosses 1e-3 (3.7e-4 -> 1.1e-3) for 0.6 ms
        p1, idx1, st1 = self._bn_block(y1, n, h1, w1, 64, 1, save)
        del y1          # the conv output is only an input of the pool: backward reads the ReLU mask off p (csrc/pool_bn.hip)
        bf = self._half
        BF16 = self._h16                 # the 16-bit storage format of this model (bfloat16 or float16)
        # where p is kept in bf16 AND the 256 x 256 tile computes the convolution, the conv output is stored as bf16 too:
        # rounding is monotonic, so the max-pool of the rounded values is the rounded max-pool (same p, same statistics);
        # only ties in the argmax are broken differently
        p16_2 = bf and self.act_bf16 and self._p16_ok(wp1, 256)
        y16_2 = p16_2 and ops.conv3x3_fwd_bf16p_o16_ok(n, hp1, wp1, 64, 256)
        y2 = torch.empty(n, hp1, wp1, 256, dtype=BF16 if y16_2 else F32, device=dev)
        if bf:
            xh1 = ops.to_bf16_padded(p1, st1[2], st1[3], self._padbuf("x1" if save else "x1e", n, hp1, wp1, 64), n, hp1, wp1, 64)
            w2b = ops.cast_bf16(P("visbl.conv2.weight"), torch.empty(256 * 9 * 64, dtype=BF16, device=dev))
            self._timed("conv_fwd", 2.0 * n * hp1 * wp1 * 576 * 256, ops.conv3x3_fwd_bf16p_o16 if y16_2 else ops.conv3x3_fwd_bf16p,
                        xh1, w2b, P("visbl.conv2.bias"), True, y2, n, hp1, wp1, 64, 256)
        elif self._x6_conv(n * hp1 * wp1, 256):
            x1s, ax = self._split_act("x1s" if save else "x1se", p1, st1[2], st1[3], n, hp1, wp1, 64)
            w2s, aw = self._split_w(P("visbl.conv2.weight"), 256 * 9, 64)
            self._timed("conv_fwd", 2.0 * n * hp1 * wp1 * 576 * 256, ops.conv3x3_fwd_split, self._parts,
                        x1s, w2s, P("visbl.conv2.bias"), True, y2, n, hp1, wp1, 64, 256, self._osc(ax, aw))
            # conv2's weight gradient (3 output tiles of 256 x 256): 12.9 + 2.6 ms (split of dy2) against the fp32 kernel's 13.9 with six
            # segments — no gain; with three (fp16x3) 6.5 + 3.3 against 14-20: taken
            if save and self._parts == 2 and os.environ.get("GOALNET_X3_WGRAD2", "1") != "0":
                ctx.update(x1s=x1s, x1s_amax=ax, x1s_gen=self._padgen["x1s"])
            del x1s, w2s
        else:
            self._timed("conv_fwd", 2.0 * n * hp1 * wp1 * 576 * 256, ops.conv3x3_fwd,
                        p1, st1[2], st1[3], P("visbl.conv2.weight"), P("visbl.conv2.bias"), True, y2, n, hp1, wp1, 64, 256)
        p2, idx2, st2 = self._bn_block(y2, n, hp1, wp1, 256, 2, save, p16=p16_2)
        del y2
        # <= 16 rows: linear5 is a pure weight stream; the fp32 weight-streaming kernels (csrc/skinny.hip) read the
        # arena once, which is cheaper (and exact) compared with casting 4 K J bytes to bf16 first
        bf5 = bf and (n > 16 or os.environ.get("GOALNET_FORCE_BF5") == "1")
        p16_3 = bf5 and self.act_bf16 and self._p16_ok(wp2, 512)
        y16_3 = p16_3 and ops.conv3x3_fwd_bf16p_o16_ok(n, hp2, wp2, 256, 512)
        y3 = torch.empty(n, hp2, wp2, 512, dtype=BF16 if y16_3 else F32, device=dev)
        if bf:
            xh2 = ops.to_bf16_padded(p2, st2[2], st2[3], self._padbuf("x2" if save else "x2e", n, hp2, wp2, 256), n, hp2, wp2, 256)
            w3b = ops.cast_bf16(P("visbl.conv3.weight"), torch.empty(512 * 9 * 256, dtype=BF16, device=dev))
            self._timed("conv_fwd", 2.0 * n * hp2 * wp2 * 2304 * 512, ops.conv3x3_fwd_bf16p_o16 if y16_3 else ops.conv3x3_fwd_bf16p,
                        xh2, w3b, P("visbl.conv3.bias"), True, y3, n, hp2, wp2, 256, 512)
        elif self._x6_conv(n * hp2 * wp2, 256):
            x2s, ax = self._split_act("x2s" if save else "x2se", p2, st2[2], st2[3], n, hp2, wp2, 256)
            w3s, aw = self._split_w(P("visbl.conv3.weight"), 512 * 9, 256)
            self._timed("conv_fwd", 2.0 * n * hp2 * wp2 * 2304 * 512, ops.conv3x3_fwd_split, self._parts,
                        x2s, w3s, P("visbl.conv3.bias"), True, y3, n, hp2, wp2, 256, 512, self._osc(ax, aw))
            if save:
                ctx.update(x2s=x2s, x2s_amax=ax, x2s_gen=self._padgen["x2s"])
            del x2s, w3s
        else:
            self._timed("conv_fwd", 2.0 * n * hp2 * wp2 * 2304 * 512, ops.conv3x3_fwd,
                        p2, st2[2], st2[3], P("visbl.conv3.weight"), P("visbl.conv3.bias"), True, y3, n, hp2, wp2, 256, 512)
        p3, idx3, st3 = self._bn_block(y3, n, hp2, wp2, 512, 3, save, p16=p16_3)
        del y3

        k5 = 512 * hp3 * wp3
        if self.grad_sync is not None:
            # shard_linear5: the all-gather of the updated weights has been running under the convolutions above
            self.grad_sync.wait_weights() if bf5 else self.grad_sync.gather_master(self)
        self.last_used_w5b = bool(bf5)                  # loop.VideoTrainer: does a graph captured from this call read the bf16 copy?
        if bf5:
            xh3 = ops.bn_apply_bf16(p3, st3[2], st3[3], torch.empty(p3.shape, dtype=BF16, device=dev), 512)
            w5b = self._w5_bf16(k5)
            ops.linear_fwd_bf16(xh3.view(n, k5), w5b, P("visbl.linear5.bias"), cat[:, voff:], relu=True,
                                dropmask=masks[0], mult_out=None if mcat is None else mcat[:, voff:])
            if save:
                ctx.update(xh3=xh3, w5b=w5b)
            del xh3, w5b
        elif self._x6_linear5(n, k5):
            # BatchNorm3 is applied in fp32 on the way into the split (one fmaf per value, as the fp32 kernel's load does)
            x3s, ax = self._split_mat(p3.view(n, k5), n, k5, scale=st3[2], shift=st3[3], bnC=512)
            w5s, aw = self._split_mat(P("visbl.linear5.weight").view(512, k5), 512, k5)
            ops.linear_fwd_split(self._parts, x3s, w5s, P("visbl.linear5.bias"), cat[:, voff:], n, k5, 512, relu=True,
                                 dropmask=masks[0], mult_out=None if mcat is None else mcat[:, voff:], oscale=self._osc(ax, aw))
            if save:
                ctx.update(x3s=x3s, w5s=w5s, x3s_amax=ax, w5s_amax=aw)
            del x3s, w5s
        else:
            ops.linear_fwd(p3.view(n, k5), P("visbl.linear5.weight"), P("visbl.linear5.bias"), cat[:, voff:], relu=True,
                           scale=st3[2], shift=st3[3], bnC=512, dropmask=masks[0], mult_out=None if mcat is None else mcat[:, voff:])

        if bf and save:
            ctx.update(xh1=xh1, xh2=xh2, bf5=bf5, padgen=(self._padgen["x1"], self._padgen["x2"]))
        ffork.join()                                # the audio half of `cat` is in place

        # ---- fusion, utils.py:242-258, 269-270
        hs, ms = [cat], [mcat]
        x = cat
        fused_mlp = self._mlp_fused(n)
        if fused_mlp:
            # the reference's sub-batch size: the five layers, the Sigmoid and 4y+1 in ONE launch (csrc/mlp.hip)
            keys = ("0", "3", "6", "9", "12")
            for width in ops.MLP_WIDTHS:
                hs.append(torch.empty(n, width, dtype=F32, device=dev))
                ms.append(torch.empty(n, width, dtype=F32, device=dev) if save else None)
            logit = torch.empty(n, dtype=F32, device=dev)
            out = torch.empty(n, dtype=F32, device=dev)
            fl = self._fused_loss                    # train_step: the broadcast MSE rides in the same launch
            ops.mlp_fwd(cat, [P(f"fusion.{k}.weight") for k in keys], [P(f"fusion.{k}.bias") for k in keys], masks[1:5],
                        hs[1:], ms[1:], logit, out, *(fl if fl is not None else (None, None, None)))
            self._fused_loss_done = fl is not None
            x = hs[4]
        for li, (key, width) in enumerate(() if fused_mlp else (("0", 512), ("3", 512), ("6", 256), ("9", 128))):
            hnext = torch.empty(n, width, dtype=F32, device=dev)
            m = torch.empty(n, width, dtype=F32, device=dev) if save else None
            ops.linear_fwd(x, P(f"fusion.{key}.weight"), P(f"fusion.{key}.bias"), hnext, relu=True,
                           dropmask=masks[1 + li], mult_out=m)
            hs.append(hnext); ms.append(m)
            x = hnext
        if fused_mlp:
            pass
        elif self.head == "classifier":
            logit = torch.empty(n, self.num_classes, dtype=F32, device=dev)
            out = torch.empty(n, self.num_classes, dtype=F32, device=dev)          # class scores 4 softmax(z) + 1
            ops.cls_head_fwd(x, P("fusion.12.weight"), P("fusion.12.bias"), logit, out)
        else:
            logit = torch.empty(n, dtype=F32, device=dev)
            out = torch.empty(n, dtype=F32, device=dev)
            ops.head_fwd(x, P("fusion.12.weight"), P("fusion.12.bias"), logit, out)
        if save:
            ctx.update(p1=p1, idx1=idx1, st1=st1, p2=p2, idx2=idx2, st2=st2, p3=p3, idx3=idx3, st3=st3,
                       a1=a1, a2=a2, hs=hs, ms=ms, logit=logit, out=out, l1=l1, l2=l2)
        self.last_logit = logit
        return out, ctx

    def _block_bwd(self, dbn, ctx, i, n, hc, wc, c):
        """BN backward + max-pool backward + ReLU backward of block i. dbn = grad wrt the BN output (N,hc-2,wc-2,c).
        Returns dy (N,hc,wc,c) = grad wrt the conv's pre-ReLU output; writes dgamma, dbeta, dbias into the grad arena."""
        dev = self._device
        G = self._gflat
        p, idx, st = ctx[f"p{i}"], ctx[f"idx{i}"], ctx[f"st{i}"]
        npix = n * (hc - 2) * (wc - 2)
        small = self._small_bn(dbn, p, n, hc, wc, c) and not (self._half and i > 1)
        if small:
            # the reference's operating point: reduce + finalise in one launch (csrc/pool_bn.hip "small shapes"), then the rolling-row
            # max-pool / ReLU backward (13 us; the one-launch 9-window gather from global memory 35), its bias-gradient rows summed at
            # the end of backward on the side stream
            coef3 = torch.empty(3 * c, dtype=F32, device=dev)
            ops.bn_bwd_reduce_small(dbn, p, st[0], st[1], self._pflat(f"visbl.bnorm{i}.weight"), G(f"visbl.bnorm{i}.weight"),
                                    G(f"visbl.bnorm{i}.bias"), coef3, n, hc, wc, c)
            dy = torch.empty(n, hc, wc, c, dtype=F32, device=dev)
            if os.environ.get("GOALNET_SMALL_BNPOOL", "0") == "1":
                ops.bnpool_bwd_small(dbn, p, idx, coef3, dy, G(f"visbl.conv{i}.bias"), n, hc, wc, c)
                return dy
            dparts = torch.empty(ops.stat_parts(8 * n) * c, dtype=torch.float64, device=dev)
            ops.bnpool_bwd(dbn, p, idx, coef3, dy, dparts, n, hc, wc, c)
            # the bias gradients' row sums feed nothing but Adam: conv3's and conv2's go out together at the end of backward (one
            # launch, side stream); conv1's is written by goalnet_conv1_wgrad from its own sums of dy
            if i > 1:
                self._dbias_pending[i] = dparts
            return dy
        coef3 = torch.empty(3 * c, dtype=F32, device=dev)
        partials = torch.empty(ops.stat_parts(npix // 64) * 2 * c, dtype=torch.float64, device=dev)
        ops.bn_bwd_reduce(dbn, p, st[0], st[1], partials, npix, c)
        ops.bn_bwd_finalize(partials, self._pflat(f"visbl.bnorm{i}.weight"), st[0], st[1], npix, c,
                            G(f"visbl.bnorm{i}.weight"), G(f"visbl.bnorm{i}.bias"), coef3)
        if self.stat_sync is not None:
            # dgamma / dbeta above are this rank's sums (the gradient all-reduce adds the ranks up); the dx coefficients
            # need the sums over every rank's pixels
            scratch = torch.empty(2 * c, dtype=F32, device=dev)
            ops.bn_bwd_finalize(self._global_sums(partials, 2 * c), self._pflat(f"visbl.bnorm{i}.weight"), st[0], st[1],
                                npix * self.stat_sync.world, c, scratch[:c], scratch[c:], coef3)
        dparts = torch.empty(ops.stat_parts(8 * n) * c, dtype=torch.float64, device=dev)        # dbias partials per (frame, row band)
        if self._half and i > 1 and self._bwd16_ok(wc):
            # blocks 2, 3: the only consumers of dy are the bf16 GEMMs -> written once, as bf16, in their padded layout
            dy = self._padbuf(f"dy{i}", n, hc, wc, c)
            ops.bnpool_bwd_bf16p(dbn, p, idx, coef3, None, dy, dparts, n, hc, wc, c)
        elif self._half and i > 1:
            # frames wider than ~416 px: the rolling LDS rows of the fused bf16 kernel do not fit; fp32 kernel (dz and p are
            # fp32 for such widths, see _p16_ok / backward_device) + one cast pass into the padded layout
            dy32 = torch.empty(n, hc, wc, c, dtype=F32, device=dev)
            ops.bnpool_bwd(dbn, p, idx, coef3, dy32, dparts, n, hc, wc, c)
            dy = ops.to_bf16_padded(dy32, None, None, self._padbuf(f"dy{i}", n, hc, wc, c), n, hc, wc, c)
        else:
            dy = torch.empty(n, hc, wc, c, dtype=F32, device=dev)
            ops.bnpool_bwd(dbn, p, idx, coef3, dy, dparts, n, hc, wc, c)
        if i == 1:
            # conv1's bias gradient is written a second time by goalnet_conv1_wgrad (on the main stream, later): keep this
            # one on the main stream too so that the order of the two writers does not depend on the schedule
            ops.partials_sum(dparts, ops.stat_parts(8 * n), c, c, G(f"visbl.conv{i}.bias"))
        else:
            self._fork.run(lambda: ops.partials_sum(dparts, ops.stat_parts(8 * n), c, c, G(f"visbl.conv{i}.bias")), dparts)
        return dy

    def backward_device(self, ctx, dout, on_bucket=None, after_linear5=None):
        """dout (N,) GPU. Fills the gradient arena (every slot is overwritten). `on_bucket(k)` is called when
        bucket k of ddp.bucket_slices() is complete (0: fusion+audbl+linear5.bias, 1: linear5.weight, 2: rest).
        `after_linear5(fork)`: called once linear5's weight gradient (side stream) and data gradient (main stream) are both
        enqueued — from there on nothing reads linear5.weight or its 16-bit copy again in this step (train_step's early Adam)."""
        self._ensure_garena()
        dev = self._device
        n, h, w = ctx["n"], ctx["h"], ctx["w"]
        (h1, w1), (hp1, wp1), (hp2, wp2), (hp3, wp3) = self._sizes(h, w)
        P, G = self._pflat, self._gflat
        hs, ms = ctx["hs"], ctx["ms"]
        # small steps: weight / bias gradients and the AudBl branch run on a side stream under the dX chain (_Fork)
        fork = self._fork = _Fork(self, self._fork_ok(n) and dout.is_cuda)
        self._dbias_pending = {}
        # Large steps (forked by _large_overlap): the side stream's pieces are paired with main-stream work of the OTHER kind — linear5's
        # Adam (HBM-bound) goes out when conv3's data gradient (MFMA-bound) starts, each weight gradient (MFMA-bound) when its layer's
        # data gradient is enqueued, i.e. under the BatchNorm / pool passes of the next block (HBM-bound). Small steps keep the weight
        # gradient FIRST: there the chain is latency-bound and everything off it should start as early as it can. Same kernels, same
        # results either way. Measured (1 024 frames, alternating runs): fp16x3 178.5 -> 176.9 ms, bf16 70.6 -> 70.6 (GOALNET_OVERLAP_PAIR=0
        # restores the weight-gradient-first order).
        pair = fork.enabled and self._large_overlap(n) and os.environ.get("GOALNET_OVERLAP_PAIR", "1") != "0"
        deferred_l5 = []

        def bucket_done(k):
            if on_bucket:
                fork.join()                     # the bucket's gradients may have been written on the side stream
                on_bucket(k)

        # the data gradients of conv3 / conv2 read the weights flipped (csrc/layout.hip); the flips depend on nothing but the
        # weights, so in a small step both go out FIRST, in one launch on the side stream, off the dX chain
        wt3 = torch.empty(512 * 9 * 256, dtype=F32, device=dev)
        wt2 = torch.empty(256 * 9 * 64, dtype=F32, device=dev)
        flips_ev = None
        flips_early = fork.enabled
        if flips_early:
            _, flips_ev = fork.run_marked(lambda: ops.conv3x3_weight_flip2(P("visbl.conv3.weight"), wt3, 512, 256,
                                                                          P("visbl.conv2.weight"), wt2, 256, 64), wt3, wt2)
        # head + fusion MLP (reverse of utils.py:242-258)
        fused_mlp = self._mlp_fused(n) and ms[0] is not None
        dz = torch.empty(n, hs[0].shape[1] if fused_mlp else 128, dtype=F32, device=dev)
        if fused_mlp:
            # one launch: the five layers' dW / db, the dX chain down to the gradient behind `cat`, and linear5's bias gradient
            keys = ("0", "3", "6", "9", "12")
            ops.mlp_bwd(dout, ctx["out"], hs, ms, [P(f"fusion.{k}.weight") for k in keys], [G(f"fusion.{k}.weight") for k in keys],
                        [G(f"fusion.{k}.bias") for k in keys], dz, G("visbl.linear5.bias"), dz.shape[1] - 512)
        elif self.head == "classifier":
            ops.cls_head_bwd(dout.view(n, self.num_classes), ctx["out"], hs[4], P("fusion.12.weight"), ms[4], dz,
                             G("fusion.12.weight"), G("fusion.12.bias"))
        else:
            ops.head_bwd(dout, ctx["out"], hs[4], P("fusion.12.weight"), ms[4], dz, G("fusion.12.weight"), G("fusion.12.bias"))
        for key, li in () if fused_mlp else (("9", 3), ("6", 2), ("3", 1), ("0", 0)):
            x_in, m_in = hs[li], ms[li]
            fork.run(lambda dz=dz, x_in=x_in, key=key: ops.linear_bwd_dw(dz, x_in, G(f"fusion.{key}.weight"), db=G(f"fusion.{key}.bias")), dz)
            dprev = torch.empty(n, x_in.shape[1], dtype=F32, device=dev)
            ops.linear_bwd_dx(dz, P(f"fusion.{key}.weight"), dprev, mult=m_in)
            dz = dprev
        voff = dz.shape[1] - 512
        dz5 = dz[:, voff:]                                     # grad wrt linear5 pre-activation

        if self.audio_included:
            l1, l2, bins = ctx["l1"], ctx["l2"], ctx["bins"]

            def audbl_bwd():                       # the whole AudBl backward hangs off dz and feeds nothing but its own gradients
                dza = dz[:, :128]
                a2f = ctx["a2"].view(n, 128 * l2)
                ops.linear_bwd_dw(dza, a2f, G("audbl.linear3.weight"), db=G("audbl.linear3.bias"))
                da2 = torch.empty(n, 128 * l2, dtype=F32, device=dev)
                ops.linear_bwd_dx(dza, P("audbl.linear3.weight"), da2, mult=None)
                da1 = torch.empty(n, 64, l1, dtype=F32, device=dev)
                if n < 64:
                    # few frames: each Conv1d layer's backward is one launch, its ReLU backward folded into the dz load
                    ops.conv1d_bwd_small(ctx["a1"], da2, ctx["a2"], P("audbl.conv2.weight"), da1, G("audbl.conv2.weight"), G("audbl.conv2.bias"), n, 64, l1, 128)
                    ops.conv1d_bwd_small(ctx["audio"], da1, ctx["a1"], P("audbl.conv1.weight"), None, G("audbl.conv1.weight"), G("audbl.conv1.bias"), n, 30, bins, 64)
                    return
                ops.relu_bwd(da2, a2f, da2)
                ops.conv1d_bwd(ctx["a1"], da2, P("audbl.conv2.weight"), da1, G("audbl.conv2.weight"), G("audbl.conv2.bias"), n, 64, l1, 128)
                ops.relu_bwd(da1, ctx["a1"], da1)
                ops.conv1d_bwd(ctx["audio"], da1, P("audbl.conv1.weight"), None, G("audbl.conv1.weight"), G("audbl.conv1.bias"), n, 30, bins, 64)
            fork.run(audbl_bwd, dz)
        if not fused_mlp:
            fork.run(lambda: ops.colsum(dz5, G("visbl.linear5.bias")), dz)
        bucket_done(0)

        # linear5 (utils.py:191): dW straight into the arena, dX = grad wrt bnorm3's output
        k5 = 512 * hp3 * wp3
        p3f = ctx["p3"].view(n, k5)
        st3 = ctx["st3"]
        bf = self._half
        # precision="bf16" / "fp16": where the 256 x 256 tile serves the data-gradient GEMM, the gradient wrt a BatchNorm output is
        # stored as bf16 (fp32 accumulators rounded once, at the store): its only readers are the two HBM-bound passes of
        # _block_bwd, and the GEMMs behind them consume bf16 anyway
        dz16 = bf and self.grad_bf16
        o16_3 = dz16 and ctx["bf5"] and self._bwd16_ok(wp2) and ops.linear_bwd_dx_bf16_o16_ok(n, k5, 512)
        dbn3 = torch.empty(n, hp3, wp3, 512, dtype=self._h16 if o16_3 else F32, device=dev)
        if bf and ctx["padgen"] != (self._padgen["x1"], self._padgen["x2"]):
            raise RuntimeError("precision='bf16': a second training-mode forward overwrote the saved bf16 operands before "
                               "backward ran; call backward after each forward (as the reference's loop does)")
        if bf and ctx["bf5"]:
            dz5b = ops.cast_bf16(dz5.contiguous(), torch.empty(n, 512, dtype=self._h16, device=dev))
            fork.run(lambda: ops.linear_bwd_dw_bf16(dz5b, ctx["xh3"].view(n, k5), G("visbl.linear5.weight")), dz5b)
            bucket_done(1)
            if o16_3:
                ops.linear_bwd_dx_bf16_o16(dz5b, ctx["w5b"], dbn3.view(n, k5))
            else:
                ops.linear_bwd_dx_bf16(dz5b, ctx["w5b"], dbn3.view(n, k5), mult=None)
            if after_linear5:
                deferred_l5.append(after_linear5) if pair else after_linear5(fork)
        elif "x3s" in ctx:
            dz5s, adz = self._split_mat(dz5, n, 512)
            osc_w = self._osc(adz, ctx["x3s_amax"])
            fork.run(lambda: ops.linear_bwd_dw_split(self._parts, dz5s, ctx["x3s"], G("visbl.linear5.weight"), n, k5, 512, oscale=osc_w), dz5s, osc_w)      # osc_w too: the side stream reads it (kept alive until the join)
            bucket_done(1)
            ops.linear_bwd_dx_split(self._parts, dz5s, ctx["w5s"], dbn3.view(n, k5), n, k5, 512, oscale=self._osc(adz, ctx["w5s_amax"]))
            if after_linear5:
                deferred_l5.append(after_linear5) if pair else after_linear5(fork)
        else:
            fork.run(lambda: ops.linear_bwd_dw(dz5, p3f, G("visbl.linear5.weight"), scale=st3[2], shift=st3[3], bnC=512), dz)
            bucket_done(1)
            ops.linear_bwd_dx(dz5, P("visbl.linear5.weight"), dbn3.view(n, k5), mult=None)
            if after_linear5:
                deferred_l5.append(after_linear5) if pair else after_linear5(fork)

        # block 3 (utils.py:184-187)
        dy3 = self._block_bwd(dbn3, ctx, 3, n, hp2, wp2, 512)
        del dbn3
        for cb in deferred_l5:                       # pair: linear5's Adam starts here, beside conv3's data gradient
            cb(fork)
        st2 = ctx["st2"]
        x6_3 = "x2s" in ctx                          # precision="bf16x6" and the forward ran conv3 on split operands
        if bf:
            dyp3 = dy3
            wg3 = lambda dyp3=dyp3: fork.run(lambda: self._timed("conv_wgrad", 2.0 * n * hp2 * wp2 * 2304 * 512, ops.conv3x3_wgrad_bf16,
                                                                 ctx["xh2"], dyp3, G("visbl.conv3.weight"), n, hp2, wp2, 256, 512), dyp3)
        elif x6_3:
            if ctx["x2s_gen"] != self._padgen["x2s"]:
                raise RuntimeError("precision='bf16x6': a second training-mode forward overwrote the saved split operands before "
                                   "backward ran; call backward after each forward (as the reference's loop does)")
            # the weight gradient and the data gradient read the gradient as 16-bit parts in the padded layout: one split pass
            dys3, ady = self._split_act("dy3s", dy3, None, None, n, hp2, wp2, 512)
            osc_w = self._osc(ady, ctx["x2s_amax"])
            wg3 = lambda dys3=dys3, osc_w=osc_w: fork.run(
                lambda: self._timed("conv_wgrad", 2.0 * n * hp2 * wp2 * 2304 * 512, ops.conv3x3_wgrad_split, self._parts,
                                    ctx["x2s"], dys3, G("visbl.conv3.weight"), n, hp2, wp2, 256, 512, osc_w), dys3, osc_w)
        else:
            wg3 = lambda dy3=dy3: fork.run(lambda: self._timed("conv_wgrad", 2.0 * n * hp2 * wp2 * 2304 * 512, ops.conv3x3_wgrad,
                                                               ctx["p2"], st2[2], st2[3], dy3, G("visbl.conv3.weight"), n, hp2, wp2, 256, 512), dy3)
        if not pair:
            wg3()
        wt = wt3
        if flips_early:
            fork.wait(flips_ev)
        else:
            ops.conv3x3_weight_flip(P("visbl.conv3.weight"), wt, 512, 256)
        o16_2 = dz16 and self._bwd16_ok(wp1) and ops.conv3x3_fwd_bf16p_o16_ok(n, hp2, wp2, 512, 256)
        dbn2 = torch.empty(n, hp2, wp2, 256, dtype=self._h16 if o16_2 else F32, device=dev)
        if self._half:
            wtb = ops.cast_bf16(wt, torch.empty(wt.shape, dtype=self._h16, device=dev))
            if o16_2:
                self._timed("conv_dgrad", 2.0 * n * hp2 * wp2 * 4608 * 256, ops.conv3x3_fwd_bf16p_o16,
                            dyp3, wtb, None, False, dbn2, n, hp2, wp2, 512, 256)
            else:
                self._timed("conv_dgrad", 2.0 * n * hp2 * wp2 * 4608 * 256, ops.conv3x3_fwd_bf16p,
                            dyp3, wtb, None, False, dbn2, n, hp2, wp2, 512, 256)
        elif x6_3:
            wts, awt = self._split_w(wt, 256 * 9, 512)
            self._timed("conv_dgrad", 2.0 * n * hp2 * wp2 * 4608 * 256, ops.conv3x3_fwd_split, self._parts,
                        dys3, wts, None, False, dbn2, n, hp2, wp2, 512, 256, self._osc(ady, awt))
            del dys3, wts
        else:
            self._timed("conv_dgrad", 2.0 * n * hp2 * wp2 * 4608 * 256, ops.conv3x3_fwd,
                        dy3, None, None, wt, None, False, dbn2, n, hp2, wp2, 512, 256)
        if pair:
            wg3()                                    # behind the data gradient: runs under block 2's BatchNorm / pool passes
        del dy3

        # block 2 (utils.py:179-182)
        dy2 = self._block_bwd(dbn2, ctx, 2, n, hp1, wp1, 256)
        del dbn2
        st1 = ctx["st1"]
        if bf:
            dyp2 = dy2
            wg2 = lambda dyp2=dyp2: fork.run(lambda: self._timed("conv_wgrad", 2.0 * n * hp1 * wp1 * 576 * 256, ops.conv3x3_wgrad_bf16,
                                                                 ctx["xh1"], dyp2, G("visbl.conv2.weight"), n, hp1, wp1, 64, 256), dyp2)
        elif "x1s" in ctx:
            if ctx["x1s_gen"] != self._padgen["x1s"]:
                raise RuntimeError("precision='fp16x3': a second training-mode forward overwrote the saved split operands before backward ran")
            dys2, ady2 = self._split_act("dy2s", dy2, None, None, n, hp1, wp1, 256)
            osc_w2 = self._osc(ady2, ctx["x1s_amax"])
            wg2 = lambda dys2=dys2, osc_w2=osc_w2: fork.run(
                lambda: self._timed("conv_wgrad", 2.0 * n * hp1 * wp1 * 576 * 256, ops.conv3x3_wgrad_split, self._parts,
                                    ctx["x1s"], dys2, G("visbl.conv2.weight"), n, hp1, wp1, 64, 256, osc_w2), dys2, osc_w2)
        else:
            wg2 = lambda dy2=dy2: fork.run(lambda: self._timed("conv_wgrad", 2.0 * n * hp1 * wp1 * 576 * 256, ops.conv3x3_wgrad,
                                                               ctx["p1"], st1[2], st1[3], dy2, G("visbl.conv2.weight"), n, hp1, wp1, 64, 256), dy2)
        if not pair:
            wg2()
        wt = wt2
        if not flips_early:
            ops.conv3x3_weight_flip(P("visbl.conv2.weight"), wt, 256, 64)
        dbn1 = torch.empty(n, hp1, wp1, 64, dtype=F32, device=dev)
        if self._half:
            wtb = ops.cast_bf16(wt, torch.empty(wt.shape, dtype=self._h16, device=dev))
            self._timed("conv_dgrad", 2.0 * n * hp1 * wp1 * 2304 * 64, ops.conv3x3_fwd_bf16p,
                        dyp2, wtb, None, False, dbn1, n, hp1, wp1, 256, 64)
        elif "x1s" in ctx and os.environ.get("GOALNET_X3_DGRAD2", "1") != "0":
            # fp16x3: the split gradient is there already (weight gradient above); 128 x 64 tile, three segments
            wts2, awt2 = self._split_w(wt, 64 * 9, 256)
            self._timed("conv_dgrad", 2.0 * n * hp1 * wp1 * 2304 * 64, ops.conv3x3_fwd_split, self._parts,
                        dys2, wts2, None, False, dbn1, n, hp1, wp1, 256, 64, self._osc(ady2, awt2))
            del wts2
        else:
            self._timed("conv_dgrad", 2.0 * n * hp1 * wp1 * 2304 * 64, ops.conv3x3_fwd,
                        dy2, None, None, wt, None, False, dbn1, n, hp1, wp1, 256, 64)
        if pair:
            wg2()
        del dy2

        # block 1 (utils.py:174-177); conv1's input needs no gradient
        dy1 = self._block_bwd(dbn1, ctx, 1, n, h1, w1, 64)
        ops.conv1_wgrad(ctx["visual"], dy1, G("visbl.conv1.weight"), G("visbl.conv1.bias"), n, h, w)
        if len(self._dbias_pending) == 2:
            d3, d2 = self._dbias_pending[3], self._dbias_pending[2]
            fork.run(lambda: ops.partials_sum2(d3, 512, G("visbl.conv3.bias"), d2, 256, G("visbl.conv2.bias")), d3, d2)
        self._dbias_pending = {}
        fork.join()                                 # every gradient is in the arena before anything downstream (Adam, all-reduce) reads it
        self._fork = _Fork(self, False)
        if on_bucket:
            on_bucket(2)

    # ------------------------------------------------------------------------------------------
    # drop-in surface: model(audio_input, visual_input)
    # ------------------------------------------------------------------------------------------
    def _to_device_inputs(self, audio_input, visual_input):
        self._require_device()
        if not torch.is_tensor(visual_input):
            raise TypeError("visual_input must be a tensor (N,3,H,W)")
        src_dev = visual_input.device
        vis = visual_input.detach().to(device=self._device, dtype=F32).contiguous()
        aud = None
        if self.audio_included:
            if not torch.is_tensor(audio_input):
                raise TypeError("audio_input must be a tensor (N,30,B) when audio_included=True")
            aud = audio_input.detach().to(device=self._device, dtype=F32).contiguous()
        return aud, vis, src_dev

    def forward(self, audio_input, visual_input):
        """utils.py:260-272. Accepts the CPU tensors the reference's scripts pass (H2D inside) or GPU tensors; the
        (N,1) result is returned on the inputs' device, attached to autograd when grad mode is on."""
        aud, vis, src_dev = self._to_device_inputs(audio_input, visual_input)
        need_grad = torch.is_grad_enabled()
        if not self._materialized:
            # first forward materialises the Lazy parameters (shapes depend on H, W, B)
            (_, _), _, _, (hp3, wp3) = self._sizes(vis.shape[2], vis.shape[3])
            l2 = 0
            if self.audio_included:
                l2 = (((aud.shape[2] - 1) // 2 + 1) - 1) // 2 + 1
            self._materialize(hp3 * wp3, l2)
        if not need_grad:
            out, _ = self.forward_device(aud, vis, save=False)
            return out.view(-1, self.num_classes).to(src_dev)
        params = [getattr(*self._module_of(s.name)) for s in self._specs]
        return _AVMFunction.apply(self, aud, vis, src_dev, *params)

    # ------------------------------------------------------------------------------------------
    # device-resident fused train step: forward, broadcast MSE, backward, Adam
    # ------------------------------------------------------------------------------------------
    def train_step(self, audio, visual, labels, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, _loop_tick=(0, 0), _scatter=None):
        """main.py:187-193 on GPU tensors. Returns (loss (1,), pred (N,)) as GPU tensors, no host sync.
        `_loop_tick`: (frames, sub-batches) the caller's loop counters advance by (loop.VideoTrainer); `_scatter(loss, pred)` -> row-copy
        segments (ops.rows_copy_batch form, cursors as they stand BEFORE the tick) that the step's last launch writes before it advances
        the counters (the per-video predictions / losses arrays of loop.VideoTrainer)."""
        self._defer_tick, self._pending_drop_tick = True, 0
        n0 = visual.shape[0]
        loss = torch.empty(1, dtype=F32, device=self._device)
        dout = torch.empty(n0 if self.head == "regression" else (n0, self.num_classes), dtype=F32, device=self._device)
        # the regression head's broadcast MSE is evaluated by the fused MLP launch itself where that launch exists (<= 16 rows)
        lab32 = labels if (torch.is_tensor(labels) and labels.dtype == F32 and labels.is_contiguous() and labels.is_cuda) else None
        self._fused_loss = (lab32, loss, dout) if (self.stat_sync is None and lab32 is not None and self._mlp_fused(n0)) else None
        self._fused_loss_done = False
        try:
            out, ctx = self.forward_device(audio, visual, save=True)
        finally:
            self._defer_tick = False
            self._fused_loss = None
        n = out.shape[0]
        if self._fused_loss_done:
            pass
        elif self.head == "classifier":
            if self.stat_sync is not None:
                raise GoalnetError("global-batch mode is defined for the regression head's broadcast MSE only")
            ops.cross_entropy(out, labels.to(F32), loss, dout)          # main.py:69, 189: CrossEntropyLoss(pred, (labels-1).long())
        elif self.stat_sync is not None:
            # the (N, N) broadcast couples every prediction with every label of the batch: evaluate it on the rank-ordered
            # concatenation and keep this rank's slice of dL/dp
            gout = torch.empty(n * self.stat_sync.world, dtype=F32, device=self._device)
            ops.mse_bcast(self.stat_sync.gather(out), self.stat_sync.gather(labels.to(F32)), loss, gout)
            dout = gout[n * self.stat_sync.rank: n * (self.stat_sync.rank + 1)]
        else:
            ops.mse_bcast(out, labels, loss, dout)
        sync = self.grad_sync
        self.last_ctx = ctx if self.keep_ctx else None
        lscale = self._loss_scale_for(n)
        self._arena_grad_scale = lscale
        if lscale != 1.0:
            ops.scale_(dout.view(-1), lscale)               # fp16: every activation gradient downstream carries this factor
        # One GPU, no gradient exchange, no overflow guard to consult: linear5.weight (99.8 % of the parameters at 224 x 224) gets
        # its Adam pass the moment its gradient exists and its last reader of the step (the data gradient) is enqueued — on the
        # side stream, i.e. 36 GB of HBM traffic under the MFMA-bound convolution gradients that follow instead of after them
        early = sync is None and self.precision != "fp16" and self._large_overlap(n) and self._fork_ok(n)
        # Small steps (the reference's 10-frame sub-batches): the same idea as a BACKGROUND pass — the update of linear5.weight (90 % of
        # the step's 0.66 GB of optimizer traffic) on a third stream and on a bounded number of blocks. Built, bit-identical, and
        # measured WITHOUT gain at any width (GOALNET_EARLY_ADAM_BLOCKS=64 / 128 / 256 / 512: 1 226 / 1 042 / 990 / 980 us per step
        # against 836 us without): every kernel of the backward chain is a few dependent memory round trips, and each of them
        # gets slower beside a stream that keeps the memory system busy. Off by default (0).
        bg_blocks = int(os.environ.get("GOALNET_EARLY_ADAM_BLOCKS", "0"))      # measured: 836 us without, 980-1230 us with 64-512 blocks
        early_bg = (not early and sync is None and self.precision == "fp32" and self._w5b is None and n <= self.overlap_rows
                    and self._fork_ok(n) and bg_blocks > 0)
        done_early = []

        def early_adam(fork):
            s5 = self.spec("visbl.linear5.weight")
            if early_bg:
                if self._adam_stream is None:
                    self._adam_stream = torch.cuda.Stream(device=self._device)
                st = self._adam_stream
                st.wait_stream(torch.cuda.current_stream())       # the data gradient (last reader of the weights) is enqueued
                if fork.enabled:
                    st.wait_stream(fork.side)                      # the weight gradient runs on the side stream
                with torch.cuda.stream(st):
                    self._adam_range(s5.offset, s5.offset + s5.numel, lr, betas, eps, 1.0 / lscale, max_blocks=bg_blocks)
            else:
                fork.run(lambda: self._adam_range(s5.offset, s5.offset + s5.numel, lr, betas, eps, 1.0 / lscale))
            done_early.append((s5.offset, s5.offset + s5.numel))
        self.backward_device(ctx, dout, on_bucket=(lambda k: sync.on_bucket(self, k)) if sync is not None else None,
                             after_linear5=early_adam if (early or early_bg) else None)
        scale = 1.0
        if sync is not None:
            scale = sync.finish(self)
        guard = None
        if self.precision == "fp16":
            # an overflow anywhere in the 16-bit chain reaches the gradients computed last (conv1 ... bnorm3) and first (the
            # fusion MLP sees nothing 16-bit): checking the two small buckets (after the exchange: all ranks agree) is enough
            s5 = self.spec("visbl.linear5.weight")
            after = min(s.offset for s in self._specs if s.offset > s5.offset)
            ops.grad_finite_check(self._garena[after:], self._state[0], self._guard[0], self._guard[1])
            guard = self._guard[0]
        self.adam_step(lr, betas, eps, scale / lscale, _tick=False, _guard=guard, _done=done_early)
        if early_bg and done_early:
            torch.cuda.current_stream().wait_stream(self._adam_stream)     # the background pass joins before the step count moves
        if _scatter is not None:
            ops.rows_scatter_tick(_scatter(loss, out), self._state, 1, self._pending_drop_tick, _loop_tick[0], _loop_tick[1], bad_step=guard)
        elif guard is not None:
            # a step whose Adam was skipped is not counted (torch's GradScaler does not count it either): the retry runs under
            # the same step count; `_adam_t` on the host counts ATTEMPTED steps
            ops.counters_add4_guarded(self._state, 1, self._pending_drop_tick, _loop_tick[0], _loop_tick[1], guard)
        else:
            ops.counters_add4(self._state, 1, self._pending_drop_tick, _loop_tick[0], _loop_tick[1])
        return loss, out

    def _adam_segments(self):
        """[(lo, hi)] arena ranges this rank's optimizer owns: everything, or — ddp.GradSync(shard_linear5=True) — everything
        but the foreign slices of visbl.linear5.weight"""
        sync = self.grad_sync
        if sync is None or not sync.sharded(self):
            return [(0, self._arena_numel)]
        s5 = self.spec("visbl.linear5.weight")
        after = min(s.offset for s in self._specs if s.offset > s5.offset)
        slo, shi = sync.shard_range(self)
        return [(0, s5.offset), (slo, shi), (after, self._arena_numel)]

    def _loss_scale_for(self, n: int) -> float:
        if self._loss_scale is not None:
            return float(self._loss_scale) / (1 << self._loss_scale_backoff) if self._loss_scale_backoff else float(self._loss_scale)
        return float(2.0 ** (10 + max(0, math.ceil(math.log2(max(n, 1)))) - self._loss_scale_backoff))

    def overflow_skipped_steps(self) -> int:
        """precision="fp16": optimizer steps skipped so far because a gradient overflowed (one host read-back)"""
        return 0 if self._guard is None else int(self._guard[1].item())

    def update_loss_scale(self) -> bool:
        """precision="fp16": the loss scale is static inside a step (a captured graph bakes it in), so a persistent overflow
        would skip every step silently. Call this where the host synchronises anyway (loop.VideoTrainer does, at its per-video
        read-back): when steps were skipped since the last call the scale is halved (True is returned; graphs keyed on the old
        scale are not replayed again). Skipped steps are not counted as optimizer steps (goalnet_counters_add4_guarded)."""
        k = self.overflow_skipped_steps()
        if k > self._skipped_seen:
            self._skipped_seen = k
            self._loss_scale_backoff += 1
            return True
        return False

    def _adam_state(self):
        segs = self._adam_segments()
        if self._adam_m is not None and self._adam_segs != segs:
            raise GoalnetError("the optimizer state was laid out for another sharding of linear5.weight; attach / detach "
                               "ddp.GradSync(shard_linear5=True) before the first optimizer step, not between steps")
        if self._adam_m is None:
            total = sum(hi - lo for lo, hi in segs)
            self._adam_m = torch.zeros(total, dtype=F32, device=self._device)
            self._adam_v = torch.zeros(total, dtype=F32, device=self._device)
            self._adam_segs = segs
        return segs

    def _adam_range(self, lo, hi, lr, betas, eps, grad_scale, max_blocks=0):
        """the fused Adam on arena[lo:hi] (unsharded optimizer state: moments live at the arena's offsets); refreshes the part of
        the 16-bit copy of linear5.weight that lies inside. The step counter is NOT advanced (train_step does that once)."""
        segs = self._adam_state()
        assert segs == [(0, self._arena_numel)]
        s5 = self.spec("visbl.linear5.weight")
        p, g, m, v = self._arena[lo:hi], self._garena[lo:hi], self._adam_m[lo:hi], self._adam_v[lo:hi]
        a, b = max(lo, s5.offset), min(hi, s5.offset + s5.numel)
        if self._w5b is not None and self._w5b_version == self._w5_version() and a < b:
            ops.adam_step_dev_shadow(p, g, m, v, lr, betas[0], betas[1], eps, self._state[0],
                                     self._w5b[a - s5.offset:b - s5.offset], a - lo, grad_scale, step_bias=1)
        elif max_blocks:
            ops.adam_step_dev_blocks(p, g, m, v, lr, betas[0], betas[1], eps, self._state[0], grad_scale, step_bias=1, max_blocks=max_blocks)
        else:
            ops.adam_step_dev(p, g, m, v, lr, betas[0], betas[1], eps, self._state[0], grad_scale, step_bias=1)

    def adam_step(self, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0, _tick=True, _guard=None, _done=()):
        """torch.optim.Adam defaults over the whole arena in one launch (main.py:70, 193). The step count is the device
        counter state[0] (= completed steps) + 1, so the same captured launch serves every step. With a sharded
        linear5.weight (ddp.py) the pass covers this rank's slice only — three launches — and the optimizer state exists
        only for what the rank owns."""
        segs = self._adam_state()
        self._adam_t += 1
        if _done:
            # ranges train_step already updated under backward (the early Adam on linear5.weight): the rest of the arena now
            assert segs == [(0, self._arena_numel)] and _guard is None
            cur = 0
            for lo, hi in sorted(_done) + [(self._arena_numel, self._arena_numel)]:
                if cur < lo:
                    self._adam_range(cur, lo, lr, betas, eps, grad_scale)
                cur = hi
            if _tick:
                ops.counter_add(self._state[0], 1)
            return
        s5 = self.spec("visbl.linear5.weight")
        shadow_ok = self._w5b is not None and self._w5b_version == self._w5_version()
        off = 0
        for lo, hi in segs:
            cnt = hi - lo
            p, g = self._arena[lo:hi], self._garena[lo:hi]
            m, v = self._adam_m[off:off + cnt], self._adam_v[off:off + cnt]
            off += cnt
            a, b = max(lo, s5.offset), min(hi, s5.offset + s5.numel)       # the part of linear5.weight inside this segment
            if _guard is not None:
                # fp16: the same pass, skipped as a whole when this step's gradients overflowed (goalnet_grad_finite_check)
                sh = self._w5b[a - s5.offset:b - s5.offset] if (shadow_ok and a < b) else None
                ops.adam_step_dev_guarded(p, g, m, v, lr, betas[0], betas[1], eps, self._state[0], _guard, shadow=sh,
                                          shadow_begin=(a - lo) if sh is not None else 0, grad_scale=grad_scale, step_bias=1)
            elif shadow_ok and a < b:
                # bf16 mode at > 16 rows: refresh the shadow of linear5.weight in the same pass (the kernels do not bump versions)
                ops.adam_step_dev_shadow(p, g, m, v, lr, betas[0], betas[1], eps, self._state[0],
                                         self._w5b[a - s5.offset:b - s5.offset], a - lo, grad_scale, step_bias=1)
            else:
                ops.adam_step_dev(p, g, m, v, lr, betas[0], betas[1], eps, self._state[0], grad_scale, step_bias=1)
        if len(segs) > 1:
            self.grad_sync.after_adam(self, self._w5b if shadow_ok else None)
        if _tick:
            ops.counter_add(self._state[0], 1)

    def make_optimizer(self, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        """`optim.Adam(model.parameters(), lr)` (main.py:70) as one fused pass over the arena: cvml_goalnet_amd.optim.Adam"""
        from .optim import Adam
        return Adam(self.parameters(), lr=lr, betas=betas, eps=eps, model=self)

    def predict_classes(self, scores: torch.Tensor) -> torch.Tensor:
        """head="classifier": `torch.argmax(predictions, axis = 1) + 1` (main.py:97, 190) on the device; (N,) float classes 1..C"""
        s = scores.detach().to(device=self._device, dtype=F32).contiguous()
        return ops.argmax_plus1(s, torch.empty(s.shape[0], dtype=F32, device=self._device))

    def grad_of(self, name) -> torch.Tensor:
        """Gradient of a parameter as a strided view with the reference's logical shape (linear5: (512, C*HW) copy)."""
        s = self.spec(name)
        v = self._view(self._garena, s)
        v = v.reshape(s.shape[0], -1) if s.kind == "lin5" else v
        # fp16: train_step leaves loss_scale x gradient in the arena (the fused Adam divides it out); the drop-in path unscales
        return v / self._arena_grad_scale if self._arena_grad_scale != 1.0 else v

    def param_of(self, name) -> torch.Tensor:
        s = self.spec(name)
        v = self._view(self._arena, s)
        return v.reshape(s.shape[0], -1) if s.kind == "lin5" else v


class _AVMFunction(torch.autograd.Function):
    """Couples the HIP forward/backward to autograd so `loss.backward()` fills `.grad` of ordinary
    nn.Parameters (SURVEY.md §8(b) "Train step" (i)). Gradients are strided views of the gradient arena."""

    @staticmethod
    def forward(ctx, model, aud, vis, src_dev, *params):
        out, saved = model.forward_device(aud, vis, save=True)
        ctx.model, ctx.saved, ctx.src_dev = model, saved, src_dev
        return out.view(-1, model.num_classes).to(src_dev)

    @staticmethod
    def backward(ctx, gout):
        model, saved = ctx.model, ctx.saved
        # a previous backward's .grad may alias the arena we are about to overwrite (no zero_grad in between)
        for s in model._specs:
            prm = getattr(*model._module_of(s.name))
            if prm.grad is not None and model._garena is not None and \
                    prm.grad.untyped_storage().data_ptr() == model._garena.untyped_storage().data_ptr():
                prm.grad = prm.grad.clone()
        dout = gout.detach().to(device=model._device, dtype=F32).contiguous().view(-1)
        lscale = model._loss_scale_for(saved["n"])
        if lscale != 1.0:                                 # fp16: scaled through the 16-bit chain, unscaled before torch sees .grad
            dout = ops.scale_(dout.clone(), lscale)
        model.backward_device(saved, dout)
        if lscale != 1.0:
            ops.scale_(model._garena, 1.0 / lscale)
        model._arena_grad_scale = 1.0
        ctx.saved = None
        grads = [model._view(model._garena, s) for s in model._specs]
        return (None, None, None, None, *grads)
