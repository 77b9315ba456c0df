#!/usr/bin/env python3
"""Per (kernel, grid) durations from the newest rocprofv3 kernel_trace.csv under gpurun_out/<dir>; arg2 = divisor (steps)."""
import csv, glob, collections, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = sys.argv[1] if len(sys.argv) > 1 else "prof_loop"
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", d, "*", "*_kernel_trace.csv")), key=os.path.getmtime)[-1]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
    key = (n[:64], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), r["LDS_Block_Size"], r["VGPR_Count"])
    acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:int(os.environ.get("TOP", 40))]:
    print(f"{sum(v)/div:8.1f} us/unit  n={len(v):5d} avg {sum(v)/len(v):7.1f} us  {k[0]}  grid=({k[1]},{k[2]}) lds={k[3]} vgpr={k[4]}")
