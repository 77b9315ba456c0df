#!/usr/bin/env python3
"""Per-kernel means of the counters collected by scripts/pmc_pool.sh (fractions of SQ_WAVE_CYCLES where that makes sense)."""
import csv, glob, os, collections, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmc_a", "pmc_b", "pmc_c"):
    fs = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", d, "*", "*counter_collection.csv")), key=os.path.getmtime)
    if not fs:
        continue
    for r in csv.DictReader(open(fs[-1])):
        n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
        if "pool" not in n and "bn_" not in n and "conv1_" not in n and "adam" not in n:
            continue
        vals[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, cs in sorted(vals.items()):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    wc = m.get("SQ_WAVE_CYCLES", 1.0)
    print(f"{n[:70]:70s} " + "  ".join(f"{c.replace('SQ_', '')}={m[c] / wc * 100:.0f}%" for c in sorted(m) if c != "SQ_WAVE_CYCLES"))
