#!/usr/bin/env python3
"""Summarise gpurun_out/pmc_{a,b,c} (scripts/pmc_gemm.sh): per kernel, mean counter values over its dispatches."""
import csv, glob, os, collections, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = sys.argv[1] if len(sys.argv) > 1 else "gemm_bf16"
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmc_a", "pmc_b", "pmc_c"):
    fs = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", d, "*", "*counter_collection.csv")), key=os.path.getmtime)
    if not fs:
        continue
    for r in csv.DictReader(open(fs[-1])):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        n = re.sub(r"\(.*", "", n)
        if pat not in n:
            continue
        vals[(n, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(vals.items()):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    line = f"{k[0][:70]} grid={k[1]}"
    if "GRBM_GUI_ACTIVE" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        line += f"  MfmaUtil={m['SQ_VALU_MFMA_BUSY_CYCLES'] / (m['GRBM_GUI_ACTIVE'] * 256 * 4) * 100:.1f}% cycles={m['GRBM_GUI_ACTIVE']:.0f}"
    if "SQ_WAVE_CYCLES" in m:
        wc = m["SQ_WAVE_CYCLES"]
        line += f"  wait_any={m.get('SQ_WAIT_ANY', 0) / wc * 100:.0f}% wait_inst={m.get('SQ_WAIT_INST_ANY', 0) / wc * 100:.0f}% active_inst={m.get('SQ_ACTIVE_INST_ANY', 0) / wc * 100:.0f}%"
    if "SQ_ACTIVE_INST_LDS" in m and "SQ_BUSY_CYCLES" in m:
        pass
    for c in ("SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_INST_CYCLES_VMEM"):
        if c in m:
            line += f"  {c}={m[c]:.3g}"
    print(line)
