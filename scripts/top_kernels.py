#!/usr/bin/env python3
"""Print the top kernels of the newest rocprofv3 kernel_stats.csv under gpurun_out/<dir> (default prof_loop)."""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = sys.argv[1] if len(sys.argv) > 1 else "prof_loop"
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0     # e.g. number of steps in the trace
fs = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", d, "*", "*_kernel_stats.csv")), key=os.path.getmtime)
rows = list(csv.DictReader(open(fs[-1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total {tot/1e6:.1f} ms, {sum(int(r['Calls']) for r in rows)} launches; per unit: {tot/1e3/div:.1f} us, {sum(int(r['Calls']) for r in rows)/div:.1f} launches")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:int(os.environ.get("TOP", 30))]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f'{float(r["TotalDurationNs"])/tot*100:5.1f}%  {float(r["TotalDurationNs"])/1e3/div:8.1f} us/unit  calls {r["Calls"]:>6}  avg {float(r["AverageNs"])/1e3:8.1f} us  {name[:100]}')
