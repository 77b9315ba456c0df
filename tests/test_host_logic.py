"""CPU: host-side logic that needs no GPU — the parameter arena layout, the synthetic generator, the DDP
bucket plan and (world_size 2, gloo) the gradient exchange itself."""
import os

import numpy as np
import pytest
import torch

from cvml_goalnet_amd import AVM, synth
from cvml_goalnet_amd.ddp import GradSync, SyncStats, bucket_slices, shard_bounds


def test_synth_is_counter_based_and_stable():
    a = synth.uniform(7, (1000,), -1.0, 1.0)
    b = synth.uniform(7, (10, 100), -1.0, 1.0)
    assert np.array_equal(a, b.reshape(-1))
    assert np.array_equal(synth.unit(9, 50, offset=25), synth.unit(9, 75)[25:])
    # pinned values: changing the generator silently would invalidate the goldens
    assert synth.raw_bits(0, 2).tolist() == synth.raw_bits(0, 2).tolist()
    v = synth.make_visual(2, 40, 40)
    assert v.shape == (2, 3, 40, 40) and v.min() == 0.0 and v.max() == 1.0
    lab = synth.make_labels(1000)
    assert set(np.unique(lab).tolist()) == {1.0, 2.0, 3.0, 4.0, 5.0}
    m = synth.make_drop_masks(64)
    assert [x.shape[1] for x in m] == [512, 512, 512, 256, 128]
    assert set(np.unique(m[0]).tolist()) == {0.0, 1.25}


def test_param_counts_match_survey():
    assert sum(int(np.prod(s)) for s in synth.param_shapes(40, 40, 30, True).values()) == 23_482_433
    assert sum(int(np.prod(s)) for s in synth.param_shapes(40, 40, 30, False).values()) == 23_255_169
    assert sum(int(np.prod(s)) for s in synth.param_shapes(224, 224, 30, True).values()) == 1_286_754_369


@pytest.mark.parametrize("audio", [True, False])
def test_arena_layout(audio):
    m = AVM(audio_included=audio)
    specs = m._param_specs(81, 8)
    names = [s.name for s in specs]
    want = set(synth.param_shapes(40, 40, 30, audio).keys())
    assert set(names) == want and len(names) == len(want)
    end = 0
    for s in specs:
        assert s.offset % 64 == 0 and s.offset >= end        # 256-byte aligned, non-overlapping
        end = s.offset + s.numel
    assert m._arena_numel >= end
    b = bucket_slices(specs, m._arena_numel)
    assert b[0][0] == 0 and b[0][1] == b[1][0] and b[1][1] == b[2][0] and b[2][1] == m._arena_numel
    w5 = next(s for s in specs if s.name == "visbl.linear5.weight")
    assert b[1] == (w5.offset, w5.offset + w5.numel)          # 512*512*81 is already a multiple of 64
    # parameters are Lazy until the first forward / load_state_dict, yet an optimizer can hold them (main.py:70)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    assert len(opt.param_groups[0]["params"]) == len(names)
    keys = m._reference_key_order()
    assert keys[:3] == ["visbl.conv1.weight", "visbl.conv1.bias", "visbl.bnorm1.weight"]
    assert ("audbl.conv1.weight" in keys) == audio


class _FakeModel:
    def __init__(self, specs, numel, garena):
        self._specs, self._arena_numel, self._garena = specs, numel, garena


def _ddp_worker(rank, world, port, tmp):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle import avm_ref
    n, h = 8, 40
    shard = slice(rank * n // world, (rank + 1) * n // world)
    params = synth.make_params(h, h)
    p = {k: torch.from_numpy(v).requires_grad_(True) for k, v in params.items()}
    vis = torch.from_numpy(synth.make_visual(n, h, h))[shard]
    aud = torch.from_numpy(synth.make_audio(n))[shard]
    lab = torch.from_numpy(synth.make_labels(n))[shard]
    loss = avm_ref.mse_bcast(avm_ref.forward(p, avm_ref.init_buffers(), aud, vis, None, True), lab)
    loss.backward()
    m = AVM(audio_included=True)
    specs = m._param_specs(81, 8)
    g = torch.zeros(m._arena_numel)
    for s in specs:
        g[s.offset:s.offset + s.numel] = p[s.name].grad.reshape(-1)
    local = g.clone()
    fake = _FakeModel(specs, m._arena_numel, g)
    sync = GradSync()
    for k in (0, 1, 2):                       # the order backward_device announces the buckets in
        sync.on_bucket(fake, k)
    scale = sync.finish(fake)
    torch.save({"local": local, "reduced": g * scale}, os.path.join(tmp, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_gradient_exchange_world2_gloo(tmp_path):
    """2 ranks, each an independent reference process on its shard (local BN, local MSE): the exchanged
    gradient must be the mean of the two local gradients."""
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_ddp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    mean = (r0["local"] + r1["local"]) / 2
    assert torch.allclose(r0["reduced"], mean, rtol=0, atol=1e-12 + 1e-6 * mean.abs().max().item())
    assert torch.equal(r0["reduced"], r1["reduced"])
    assert not torch.equal(r0["local"], r1["local"])


def _syncstats_worker(rank, world, port, tmp):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    st = SyncStats()
    row = torch.arange(6, dtype=torch.float64) * (rank + 1) + 2.0 ** -40          # bits a float32 exchange would drop
    got = {"rank": st.rank, "world": st.world, "sum": st.all_reduce(row.clone()),
           "cat": st.gather(torch.arange(3, dtype=torch.float32) + 10 * rank), "avg": GradSync(average=False).finish(None)}
    torch.save(got, os.path.join(tmp, f"s{rank}.pt"))
    dist.destroy_process_group()


def test_syncstats_collectives_world2_gloo(tmp_path):
    """global-batch mode's host side: per-channel sums are added in double, predictions / labels are concatenated in
    rank order, and the summed gradient is not divided by the world size (SyncBN)."""
    import torch.multiprocessing as mp
    port = 25500 + (os.getpid() % 2000)
    mp.spawn(_syncstats_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for rank in (0, 1):
        got = torch.load(tmp_path / f"s{rank}.pt")
        assert (got["rank"], got["world"], got["avg"]) == (rank, 2, 1.0)
        assert torch.equal(got["sum"], torch.arange(6, dtype=torch.float64) * 3 + 2.0 ** -39)
        assert got["cat"].tolist() == [0.0, 1.0, 2.0, 10.0, 11.0, 12.0]


class _BN:
    def __init__(self, c, fill):
        self.running_mean = torch.full((c,), float(fill))
        self.running_var = torch.full((c,), 1.0 + fill)
        self.num_batches_tracked = torch.tensor(int(fill))


class _Vis:
    pass


class _FakeReplica:
    """what GradSync touches of an AVM, on CPU tensors"""

    def __init__(self, specs, numel, rank):
        g = torch.Generator().manual_seed(100 + rank)                  # every rank its own "random init"
        self._specs, self._arena_numel = specs, numel
        self._arena = torch.rand(numel, generator=g)
        self._garena = torch.rand(numel, generator=g)
        self._adam_m = self._adam_v = None
        self._adam_t, self._drop_step, self._load_count = 3 * rank, 5 * rank, 0
        self._state = torch.tensor([self._adam_t, self._drop_step, 0, 0])
        self.dropout_seed = 1234 + rank
        self.stat_sync = None
        self.visbl = _Vis()
        for i, c in ((1, 64), (2, 256), (3, 512)):
            setattr(self.visbl, f"bnorm{i}", _BN(c, rank + 1))
        self._w5b, self._w5b_version = None, None

    def _w5_version(self):
        return (self._arena._version, self._load_count)

    def _adam_state(self):
        if self._adam_m is None:
            self._adam_m, self._adam_v = torch.zeros(self._arena_numel), torch.zeros(self._arena_numel)


def _sync_worker(rank, world, port, tmp):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = AVM(audio_included=True)
    specs = m._param_specs(81, 8)
    lo, hi = bucket_slices(specs, m._arena_numel)[1]
    out = {}
    # ---- replicas with different initial state -> one model; dropout streams stay per rank
    fake = _FakeReplica(specs, m._arena_numel, rank)
    sync = GradSync(shard_linear5=True)
    sync.ensure_params_synced(fake)
    out["sync"] = {"arena": fake._arena.clone(), "rm": fake.visbl.bnorm2.running_mean.clone(), "nbt": int(fake.visbl.bnorm3.num_batches_tracked),
                   "adam_t": fake._adam_t, "drop_step": fake._drop_step, "state": fake._state.clone(), "seed": fake.dropout_seed,
                   "load_count": fake._load_count}
    # ---- sharded linear5.weight: reduce(-scatter), "Adam" (p -= g) on the owned slice, all-gather
    assert sync.sharded(fake)
    slo, shi = sync.shard_range(fake)
    local_g = fake._garena.clone()
    for k in (0, 1, 2):
        sync.on_bucket(fake, k)
    scale = sync.finish(fake)
    before = fake._arena.clone()
    for a, b in ((0, lo), (slo, shi), (hi, m._arena_numel)):
        fake._arena[a:b] -= scale * fake._garena[a:b]
    sync.after_adam(fake, None)
    sync.wait_weights()
    out["sharded"] = {"arena": fake._arena.clone(), "before": before, "local_g": local_g, "range": (slo, shi)}
    # ---- precision="bf16": the bf16 copy is gathered, the fp32 master of foreign slices is stale until gather_master()
    fake._w5b = fake._arena[lo:hi].to(torch.bfloat16)
    fake._w5b_version = fake._w5_version()
    fake._arena[slo:shi] += 1.0 + rank                                  # "Adam" on the owned slice + its bf16 copy
    fake._w5b[slo - lo:shi - lo] = fake._arena[slo:shi].to(torch.bfloat16)
    fake._w5b_version = fake._w5_version()
    sync.after_adam(fake, fake._w5b)
    sync.wait_weights()
    stale = fake._arena[lo:hi].clone()
    assert sync.master_stale
    sync.gather_master(fake)
    out["bf16"] = {"shadow": fake._w5b.clone(), "stale": stale, "master": fake._arena[lo:hi].clone(),
                   "stamp_ok": fake._w5b_version == fake._w5_version(), "still_stale": sync.master_stale}
    # ---- optimizer state present on one rank only (a per-rank fact: e.g. one rank resumed): rank 0's answer decides the
    # broadcast list on every rank — a list that differed between ranks would deadlock
    for owner in (0, 1):
        f2 = _FakeReplica(specs, m._arena_numel, rank)
        if rank == owner:
            f2._adam_m, f2._adam_v = torch.full((m._arena_numel,), 0.5), torch.full((m._arena_numel,), 0.25)
        GradSync().sync_params(f2)
        out[f"opt_owner{owner}"] = None if f2._adam_m is None else (float(f2._adam_m[7]), float(f2._adam_v[7]))
    torch.save(out, os.path.join(tmp, f"y{rank}.pt"))
    dist.destroy_process_group()


def test_param_sync_and_sharded_linear5_world2_gloo(tmp_path):
    """ddp.GradSync: (1) ranks that built their model from different torch seeds start from rank 0's parameters, buffers
    and counters, with one dropout stream per rank; (2) shard_linear5 — reduce-scatter, update of the owned slice, all-gather —
    leaves every rank with the parameters the replicated update gives; (3) with a bf16 copy the gather moves the copy and
    gather_master() repairs the fp32 master."""
    import torch.multiprocessing as mp
    port = 27500 + (os.getpid() % 2000)
    mp.spawn(_sync_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [torch.load(tmp_path / f"y{k}.pt") for k in (0, 1)]
    a, b = r[0]["sync"], r[1]["sync"]
    ref = _FakeReplica(AVM(audio_included=True)._param_specs(81, 8), a["arena"].numel(), 0)
    assert torch.equal(a["arena"], ref._arena) and torch.equal(b["arena"], ref._arena)
    assert torch.equal(b["rm"], torch.full((256,), 1.0)) and b["nbt"] == 1
    assert (b["adam_t"], b["drop_step"], b["state"].tolist()) == (0, 0, [0, 0, 0, 0])
    assert a["seed"] == 1234 and b["seed"] != a["seed"] and b["load_count"] == 1
    m = AVM(audio_included=True)
    specs = m._param_specs(81, 8)
    lo, hi = bucket_slices(specs, m._arena_numel)[1]
    assert r[0]["sharded"]["range"] == (lo, lo + (hi - lo) // 2) and r[1]["sharded"]["range"] == (lo + (hi - lo) // 2, hi)
    want = r[0]["sharded"]["before"] - 0.5 * (r[0]["sharded"]["local_g"] + r[1]["sharded"]["local_g"])
    for k in (0, 1):
        assert torch.equal(r[k]["sharded"]["arena"], want), f"rank {k}"
    for k in (0, 1):
        x = r[k]["bf16"]
        assert torch.equal(x["shadow"], r[0]["bf16"]["shadow"]) and torch.equal(x["master"], r[0]["bf16"]["master"])
        assert torch.equal(x["shadow"], x["master"].to(torch.bfloat16)) and x["stamp_ok"] and not x["still_stale"]
        assert not torch.equal(x["stale"], x["master"])                 # the foreign slice really was out of date
    for k in (0, 1):
        assert r[k]["opt_owner0"] == (0.5, 0.25), "rank 0's Adam moments must reach a rank that had none"
        assert r[k]["opt_owner1"] is None, "rank 0 has no optimizer state: a rank that had some starts without it too"
    assert shard_bounds(0, 1000, 8, 3) is None and shard_bounds(64, 64 + 8 * 128, 8, 3) == (64 + 3 * 128, 64 + 4 * 128)


def test_fused_adam_optimizer_can_be_created_before_the_first_forward():
    """cvml_goalnet_amd.optim.Adam mirrors `optim.Adam(model.parameters(), lr)` at main.py:70: it is constructed while the parameters are
    still Lazy (no GPU needed), refuses to step without gradients, and insists on the whole parameter set of its AVM."""
    import pytest
    from cvml_goalnet_amd.optim import Adam
    m = AVM(audio_included=True)
    opt = m.make_optimizer(lr=0.001)
    assert isinstance(opt, torch.optim.Optimizer) and len(opt.param_groups[0]["params"]) == 30
    opt.zero_grad()
    with pytest.raises(RuntimeError):
        opt.step()
    with pytest.raises(RuntimeError):
        Adam(m.parameters(), lr=0.001).step()            # no model given
    with pytest.raises(ValueError):
        Adam(m.parameters(), lr=-1.0, model=m)


def test_the_fp32_gemm_kernels_keep_their_accumulators_out_of_scratch(tmp_path):
    """Build-level guard for the headline kernels (csrc/gemm_f32.hip): the conv forward instantiation uses no scratch at all and no
    gemm_f32_kernel instantiation more than 64 bytes per lane (the one-stage forms are capped at 128 VGPRs for four blocks per CU and
    spill 2-11 registers at their tile boundaries: 8-44 bytes). Round 3 found out why this is worth 20 s: one more conditional load
    in `epi_apply` kept the epilogue's store loop from unrolling, the accumulators were indexed dynamically and moved to scratch (352 B
    per lane), and the fp32 conv forward fell from 0.85 to 0.54 of peak (357 GB of traffic per launch) with every parity test green."""
    import re
    import subprocess
    import __graft_entry__ as ge
    src = os.path.join(ge.CSRC, "gemm_f32.hip")
    r = subprocess.run([ge.HIPCC, *ge.FLAGS, "-I" + os.path.join(ge.ROOT, "include"), "--cuda-device-only", "-c", src, "-o", str(tmp_path / "g.o"),
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    names = re.findall(r"Function Name: (\S+)", r.stderr)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)]
    assert len(names) == len(scratch) and len(names) >= 10
    gemm = [(n, s) for n, s in zip(names, scratch) if "gemm_f32_kernel" in n]
    bad = [(n, s) for n, s in gemm if s > 64]
    assert not bad, f"fp32 GEMM kernels with their accumulators in scratch: {bad}"
    fwd = [s for n, s in gemm if "ConvALoaderILb1EEENS_8KCLoaderILb0EEELb0ELi1" in n]
    assert fwd == [0], f"the fp32 conv forward kernel uses scratch: {fwd}"


def test_split_operand_arithmetic_of_bf16x6_and_fp16x3_in_numpy():
    """The arithmetic behind precision="bf16x6" / "fp16x3" (csrc/split3.hip), restated in numpy with exact (fp64) products and sums:
    (i) hi + mid + lo of three bf16 roundings IS the fp32 value; (ii) the six largest partial products miss a b by less than 2^-23 |a b|;
    (iii) two fp16 parts of the value scaled into [2^14, 2^15) hold it to 2^-22 (relative, for entries within 2^-17 of the tensor's
    largest magnitude) and the three largest products miss a b by less than 2^-20 |a| |b|; (iv) the scale exponent the device derives from a
    magnitude's bit pattern (141 - biased exponent) puts that magnitude into [2^14, 2^15)."""
    rng = np.random.default_rng(5)

    def bf16(x):
        u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
        u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
        return u.astype(np.uint32).view(np.float32)

    a = (rng.standard_normal(4096) * np.exp(rng.standard_normal(4096) * 3)).astype(np.float32)
    b = (rng.standard_normal(4096) * 0.05).astype(np.float32)
    ah = bf16(a); am = bf16(a - ah); al = bf16(a - ah - am)
    bh = bf16(b); bm = bf16(b - bh); bl = bf16(b - bh - bm)
    assert np.array_equal(ah.astype(np.float64) + am + al, a.astype(np.float64))                      # (i)
    exact = a.astype(np.float64) * b.astype(np.float64)
    six = sum(x.astype(np.float64) * y for x, y in ((am, bm), (ah, bl), (al, bh), (ah, bm), (am, bh), (ah, bh)))
    assert np.all(np.abs(six - exact) <= 2.0 ** -23 * np.abs(exact))                                   # (ii)
    for t in (a, b):                                                                                   # (iv)
        bits = int(np.abs(t).max().view(np.uint32))
        k = 141 - (bits >> 23)
        assert 2.0 ** 14 <= float(np.abs(t).max()) * 2.0 ** k < 2.0 ** 15
    sa = 2.0 ** (141 - (int(np.abs(a).max().view(np.uint32)) >> 23)); sb = 2.0 ** (141 - (int(np.abs(b).max().view(np.uint32)) >> 23))
    A, B = (a * np.float32(sa)).astype(np.float32), (b * np.float32(sb)).astype(np.float32)
    Ah = A.astype(np.float16).astype(np.float32); Am = (A - Ah).astype(np.float16).astype(np.float32)
    Bh = B.astype(np.float16).astype(np.float32); Bm = (B - Bh).astype(np.float16).astype(np.float32)
    big = np.abs(A) >= 2.0 ** -3                                                                       # mid is a normal fp16 number there
    assert np.all(np.abs((Ah.astype(np.float64) + Am) - A)[big] <= 2.0 ** -22 * np.abs(A)[big])
    three = (Ah.astype(np.float64) * Bm + Am.astype(np.float64) * Bh + Ah.astype(np.float64) * Bh) / (sa * sb)
    assert np.all(np.abs(three - exact) <= 2.0 ** -20 * np.abs(a.astype(np.float64)) * np.abs(b.astype(np.float64)) + 2.0 ** -24 / (sa * sb) * 2.0 ** 15)   # (iii)
