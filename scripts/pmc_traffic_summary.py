"""Per-kernel HBM GB / launch ((2 FETCH_SIZE + WRITE_SIZE) x 1024, MI355X_MICROARCH.md "HBM") and average duration from the three passes
of scripts/pmc_traffic.sh."""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
agg = collections.defaultdict(lambda: {"FETCH_SIZE": [], "WRITE_SIZE": []})


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split(">(")[0] + ">" if name.startswith("gemm_") else name.split("(")[0]


for ctr, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    for f in glob.glob(os.path.join(d, sub, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr:
                agg[short(r["Kernel_Name"])][ctr].append(float(r["Counter_Value"]))
dur = {}
for f in glob.glob(os.path.join(d, "stats", "**", "*_kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r["Name"])] = float(r["AverageNs"]) / 1e6
print("| kernel | launches | FETCH_SIZE avg (KiB) | WRITE_SIZE avg (KiB) | HBM GB / launch | avg ms |")
print("|---|---:|---:|---:|---:|---:|")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]["FETCH_SIZE"])):
    if not v["FETCH_SIZE"] or not (k.startswith("gemm_") or "splitk" in k):
        continue
    fe = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"])
    wr = sum(v["WRITE_SIZE"]) / max(len(v["WRITE_SIZE"]), 1)
    print(f"| `{k}` | {len(v['FETCH_SIZE'])} | {fe:.0f} | {wr:.0f} | {(2 * fe + wr) * 1024 / 1e9:.2f} | {dur.get(k, float('nan')):.3f} |")
