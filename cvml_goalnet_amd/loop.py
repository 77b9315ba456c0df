"""Device-side version of the reference's per-video sub-batch training loop.

The reference trains one video at a time, ten frames per optimizer step, and synchronises the host twice per
step (`subbatch_loss.item()`, `subbatch_predictions.flatten().tolist()`), `/root/reference/main.py:169-198`:

    while subbatch_ceil != len(batch_frames):
        ... frames[a:b], audios[a:b], labels[a:b]                  # main.py:181-184
        optimizer.zero_grad(); pred = model(audios, frames)        # main.py:187-188
        loss = criterion(pred, labels); loss.backward(); optimizer.step()   # main.py:191-193
        batch_loss += loss.item(); batch_predictions += pred.flatten().tolist()   # main.py:195-196
    batch_loss = batch_loss / iterations                            # main.py:203

`VideoTrainer.train_video` is that loop with the video resident in HBM and ONE HIP-graph launch per sub-batch:
the graph gathers rows [cursor, cursor+n) of the video (one launch), runs forward + broadcast MSE + backward + fused
Adam (`AVM.train_step`, whose last launch advances all four counters) and scatters the predictions and the loss into
per-video device arrays (one launch).
Everything that changes between sub-batches (frame cursor, sub-batch index, Adam step count, dropout draw index)
is a device counter (csrc/stepstate.hip), so the same graph is replayed for every full sub-batch; a shorter last
sub-batch uses a second graph captured for its own size. The host reads results back once per video.

The first sub-batch of each size runs eagerly (it is a real step — it also allocates the optimizer state and the
cached operand buffers); the graph is captured at the second occurrence. Capturing executes nothing, so the
sequence of optimizer steps is exactly the reference's.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from . import ops

F32 = torch.float32


class VideoTrainer:
    def __init__(self, model, subbatch_size: int = 10, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 graphs: bool = True):
        if subbatch_size < 1:
            raise ValueError("subbatch_size must be >= 1")
        self.model = model
        self.subbatch_size = subbatch_size          # main.py:44
        self.lr, self.betas, self.eps = lr, betas, eps
        self.graphs = graphs
        self._graphs: Dict[tuple, torch.cuda.CUDAGraph] = {}        # (sub-batch size, loss scale) -> graph
        self._uses_w5b: Dict[tuple, bool] = {}      # that graph reads the bf16 copy of linear5.weight (precision="bf16", n > 16)
        self._seen = set()
        self._pool = None
        self._cap = 0                               # frames the staging tables hold
        self._key = None                            # (H, W, B): shapes the tables / graphs were built for
        self.replays = 0                            # statistics: graph launches vs eager steps
        self.eager_steps = 0

    # ---- staging tables: the whole video in HBM -----------------------------------------------------------------
    def _ensure_tables(self, n_frames: int, hw: Tuple[int, int], bins: int):
        m = self.model
        key = (hw, bins)
        if key == self._key and n_frames <= self._cap:
            return
        dev = m._device
        cap = max(n_frames, 2 * self._cap if key == self._key else 0, 64)
        self._graphs.clear()                        # pointers below are baked into captured graphs
        self._uses_w5b.clear()
        self._pool = None
        self._key, self._cap = key, cap
        sb = self.subbatch_size
        self._vid = torch.empty(cap, 3, hw[0], hw[1], dtype=F32, device=dev)
        self._aud = torch.empty(cap, 30, bins, dtype=F32, device=dev) if m.audio_included else None
        self._lab = torch.empty(cap, dtype=F32, device=dev)
        self._pred = torch.empty(cap, dtype=F32, device=dev)
        self._loss = torch.empty((cap + sb - 1) // sb + 1, dtype=F32, device=dev)

    def _sub_step(self, n: int):
        """gather rows -> train_step -> scatter results -> advance the counters. Pure device work (capturable)."""
        m = self.model
        st = m._state
        dev = m._device
        hw, bins = self._key
        vis = torch.empty(n, 3, hw[0], hw[1], dtype=F32, device=dev)
        lab = torch.empty(n, dtype=F32, device=dev)
        segs = [(self._vid, vis, n, st[2], 0, True), (self._lab, lab, n, st[2], 0, True)]
        aud = None
        if m.audio_included:
            aud = torch.empty(n, 30, bins, dtype=F32, device=dev)
            segs.append((self._aud, aud, n, st[2], 0, True))
        ops.rows_copy_batch(segs)                                       # frames[a:b], labels[a:b], audios[a:b]
        # train_step advances all four counters in its last launch: cursor += n, sub-batch index += 1
        if m.head == "classifier":
            loss, pred = m.train_step(aud, vis, lab, self.lr, self.betas, self.eps, _loop_tick=(n, 1))
            pred = ops.argmax_plus1(pred, torch.empty(n, dtype=F32, device=dev))        # main.py:190: argmax + 1 is what gets collected
            # predictions.extend(...), losses.append(...) at the positions the step started from
            ops.rows_copy_batch([(self._pred, pred, n, st[2], -n, False), (self._loss, loss, 1, st[3], -1, False)])
            return
        # regression head: that same last launch first writes predictions.extend(...), losses.append(...) at the positions the step
        # started from (main.py:195-196), then moves the counters
        m.train_step(aud, vis, lab, self.lr, self.betas, self.eps, _loop_tick=(n, 1),
                     _scatter=lambda loss, pred: [(self._pred, pred, n, st[2], 0, False), (self._loss, loss, 1, st[3], 0, False)])

    def _host_bookkeeping_after_replay(self):
        """What an eager train_step does on the host besides launching kernels."""
        m = self.model
        m._adam_t += 1
        if m.dropout_mode == "device":
            m._drop_step += 1
        for i in (1, 2, 3):
            getattr(m.visbl, f"bnorm{i}").num_batches_tracked += 1

    def _run(self, n: int):
        m = self.model
        if not self.graphs or m.grad_sync is not None or m.dropout_mode == "given" or m.kernel_events is not None:
            self._sub_step(n)                       # collectives / supplied masks / per-kernel timing: eager
            self.eager_steps += 1
            return
        gkey = (n, m._loss_scale_for(n))                # the loss scale is a host scalar baked into the captured launches
        g = self._graphs.get(gkey)
        if g is not None and self._uses_w5b.get(gkey) and m._w5b_version != m._w5_version():
            # the captured graph reads the bf16 copy of linear5.weight and keeps it fresh through its own fused Adam, but holds
            # no cast node: a writer outside the graph since the last step (load_state_dict, a stock optimizer, an in-place
            # edit) has left the copy stale. One eager step re-casts it (AVM._w5_bf16) and re-validates the stamp.
            self._sub_step(n)
            self.eager_steps += 1
            return
        if g is None:
            if gkey not in self._seen:              # first step of this size: eager (allocates Adam state, operand buffers)
                self._seen.add(gkey)
                self._sub_step(n)
                self.eager_steps += 1
                return
            saved = (m._adam_t, m._drop_step, [int(getattr(m.visbl, f"bnorm{i}").num_batches_tracked) for i in (1, 2, 3)],
                     m.keep_ctx)
            m.keep_ctx = False
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, pool=self._pool):
                self._sub_step(n)
            if self._pool is None:
                self._pool = g.pool()
            # capturing launched nothing: undo the host-side counters the captured code advanced
            m._adam_t, m._drop_step, nbt, m.keep_ctx = saved
            for i, v in zip((1, 2, 3), nbt):
                getattr(m.visbl, f"bnorm{i}").num_batches_tracked.fill_(v)
            self._graphs[gkey] = g
            self._uses_w5b[gkey] = m._w5b is not None and m._w5b_version == m._w5_version() and m.last_used_w5b
        g.replay()
        self._host_bookkeeping_after_replay()
        self.replays += 1

    # ---- validation pass ---------------------------------------------------------------------------------------------
    def eval_video(self, val_audios, val_frames, val_labels):
        """main.py:218-226 for one video: the whole video in ONE forward under no_grad (BatchNorm in train mode, running
        statistics updated, dropout live — the reference never calls .eval()), then nn.MSELoss with its (n,1) x (n,)
        broadcast. Returns (loss (1,), predictions (N,)) as GPU tensors; nothing has been synchronised."""
        m = self.model
        m._require_device()
        aud, vis, _ = m._to_device_inputs(val_audios, val_frames)
        lab = torch.as_tensor(val_labels).detach().to(device=m._device, dtype=F32).reshape(-1).contiguous()
        if lab.numel() != vis.shape[0]:
            raise RuntimeError(f"{lab.numel()} labels for {vis.shape[0]} frames")
        with torch.no_grad():
            out, _ = m.forward_device(aud, vis, save=False)
        loss = torch.empty(1, dtype=F32, device=m._device)
        if m.head == "classifier":
            ops.cross_entropy(out, lab, loss, None)                     # main.py:96-97
            return loss, ops.argmax_plus1(out, torch.empty(lab.numel(), dtype=F32, device=m._device))
        ops.mse_bcast(out, lab, loss, None)
        return loss, out

    # ---- the loop ------------------------------------------------------------------------------------------------
    def train_video(self, batch_audios, batch_frames, batch_labels):
        """One video (main.py:169-203). batch_frames (N,3,H,W), batch_audios (N,30,B) or a list of None,
        batch_labels (N,) — CPU or GPU tensors. Returns (losses (S,), predictions (N,)) as GPU tensors with
        S = ceil(N / subbatch_size); nothing has been synchronised. `losses.mean()` is main.py:203's batch_loss."""
        m = self.model
        m._require_device()
        if not torch.is_tensor(batch_frames) or batch_frames.dim() != 4 or batch_frames.shape[1] != 3:
            raise RuntimeError("batch_frames must be a (N,3,H,W) tensor")
        n_frames = batch_frames.shape[0]
        if n_frames < 1:
            raise RuntimeError("empty video")
        if batch_labels.numel() != n_frames:
            raise RuntimeError(f"{batch_labels.numel()} labels for {n_frames} frames")
        bins = 0
        if m.audio_included:
            if not torch.is_tensor(batch_audios) or batch_audios.dim() != 3 or batch_audios.shape[:2] != (n_frames, 30):
                raise RuntimeError("batch_audios must be (N,30,B) when audio_included=True")
            bins = batch_audios.shape[2]
        self._ensure_tables(n_frames, tuple(batch_frames.shape[2:]), bins)
        self._vid[:n_frames].copy_(batch_frames, non_blocking=True)
        self._lab[:n_frames].copy_(batch_labels.reshape(-1), non_blocking=True)
        if m.audio_included:
            self._aud[:n_frames].copy_(batch_audios, non_blocking=True)
        if m._state is None:
            m._make_state()
        m._state[2:4].zero_()                       # subbatch_offset, iterations (main.py:173-175)
        if m.precision == "fp16":
            # the previous video's results have been read back by now (main.py:203): a cheap place to notice skipped steps and
            # halve the loss scale (AVM.update_loss_scale) — a static scale would otherwise stall training silently
            m.update_loss_scale()
        sb = self.subbatch_size
        subs = 0
        for off in range(0, n_frames, sb):
            self._run(min(sb, n_frames - off))
            subs += 1
        return self._loss[:subs].clone(), self._pred[:n_frames].clone()
