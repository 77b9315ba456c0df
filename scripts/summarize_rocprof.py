#!/usr/bin/env python3
"""Turn rocprofv3 output of scripts/profile_bench.sh (gpurun_out/prof_*) into the committed summaries:
   profiles/<tag>_kernel_stats.md   per-kernel time table (--kernel-trace --stats pass)
   profiles/<tag>_traffic.md        per-kernel FETCH_SIZE / WRITE_SIZE (separate --pmc passes)
   profiles/conv_fwd_traffic.json   HBM bytes per launch of the dominant kernel, read by bench.py (roofline.traffic)
HBM bytes follow MI355X_MICROARCH.md "HBM": bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 — FETCH_SIZE reports exactly
half of a wide coalesced read stream on gfx950, WRITE_SIZE is exact for 16-B-per-lane streaming stores."""
import csv, glob, json, os, subprocess, sys, collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0] if not name.startswith(("gemm_f32_kernel", "gemm_bf16")) else name.split(">(")[0] + ">"


def one(pattern):
    fs = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", pattern)), key=os.path.getmtime)
    return fs[-1] if fs else None


f = one("prof_stats/*/*_kernel_stats.csv")
bench = None
try:
    bench = json.loads(open(os.path.join(ROOT, "gpurun_out", "prof_stats.json")).read().strip().splitlines()[-1])
except Exception:
    pass
if f:
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(os.path.join(out, f"{tag}_kernel_stats.md"), "w") as o:
        o.write(f"# rocprofv3 --kernel-trace --stats — bench.py ({tag})\n\n")
        o.write("Command (on the MI355X box, scripts/profile_bench.sh): `rocprofv3 --kernel-trace --stats --output-format csv -- "
                "python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-native40 --dtype <see bench line>` (4 train steps in the trace: 2 warm-up + 2 timed).\n\n")
        if bench:
            o.write(f"bench line of the same run: value = {bench['value']:.2f} clips/s, {bench['ms_per_step']:.1f} ms/step; "
                    f"roofline: {json.dumps(bench.get('roofline'))}\n\n")
        o.write(f"Total kernel time {tot / 1e6:.1f} ms.\n\n| kernel | calls | total ms | avg ms | min ms | max ms | % |\n|---|---:|---:|---:|---:|---:|---:|\n")
        for r in rows:
            if float(r["Percentage"]) < 0.01:
                continue
            o.write(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e6:.3f} | "
                    f"{float(r['MinNs']) / 1e6:.3f} | {float(r['MaxNs']) / 1e6:.3f} | {float(r['Percentage']):.2f} |\n")
    print("wrote", f"{tag}_kernel_stats.md")

agg = collections.defaultdict(lambda: {"FETCH_SIZE": [], "WRITE_SIZE": []})
for ctr, pat in (("FETCH_SIZE", "prof_fetch/*/*_counter_collection.csv"), ("WRITE_SIZE", "prof_write/*/*_counter_collection.csv")):
    f = one(pat)
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == ctr:
            agg[short(r["Kernel_Name"])][ctr].append(float(r["Counter_Value"]))
if agg:
    with open(os.path.join(out, f"{tag}_traffic.md"), "w") as o:
        o.write(f"# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) — bench.py ({tag})\n\n")
        o.write("Counter unit: KiB. HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 FETCH_SIZE correction, "
                "MI355X_MICROARCH.md §HBM). Average over the launches of each kernel in 4 train steps.\n\n")
        o.write("| kernel | launches | FETCH_SIZE avg (KiB) | WRITE_SIZE avg (KiB) | HBM GB / launch |\n|---|---:|---:|---:|---:|\n")
        items = []
        for k, v in agg.items():
            fe = sum(v["FETCH_SIZE"]) / max(len(v["FETCH_SIZE"]), 1)
            wr = sum(v["WRITE_SIZE"]) / max(len(v["WRITE_SIZE"]), 1)
            items.append((2 * fe + wr, k, len(v["FETCH_SIZE"]), fe, wr))
        for b, k, n, fe, wr in sorted(items, reverse=True):
            if b * 1024 < 1e6:
                continue
            o.write(f"| `{k}` | {n} | {fe:.0f} | {wr:.0f} | {b * 1024 / 1e9:.3f} |\n")
    for dt, prefix in (("f32", "gemm_f32_kernel<ConvALoader<true>"), ("bf16", "gemm_bf16_256_kernel<ConvAPadLoader256<64>, KCLoader256<32>, 0, false>")):
        key = next((k for k in agg if k.startswith(prefix)), None)
        if not key:
            continue
        v = agg[key]
        fe = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]); wr = sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])
        sys.path.insert(0, ROOT)
        import bench                                   # kernel_sources_sha: bench.py reports the traffic only while it still matches
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
        json.dump({"kernel": key, "launches": len(v["FETCH_SIZE"]), "fetch_size_kib_avg": fe, "write_size_kib_avg": wr,
                   "hbm_bytes_per_launch": (2 * fe + wr) * 1024, "source": f"profiles/{tag}_traffic.md",
                   "sources_sha256": bench.kernel_sources_sha(dt), "git_head": head},
                  open(os.path.join(out, f"conv_fwd_traffic_{dt}.json"), "w"), indent=1)
    print("wrote", f"{tag}_traffic.md")
