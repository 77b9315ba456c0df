#!/usr/bin/env python3
"""Summarise gpurun_out/pmc_{a,b,c} (scripts/pmc_gemm.sh): per kernel, mean counter values over its dispatches."""
import csv, glob, os, collections, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = sys.argv[1] if len(sys.argv) > 1 else "gemm_bf16"
md = sys.argv[sys.argv.index("--md") + 1] if "--md" in sys.argv else None
md_rows = []
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
        vals[(n, r["Grid_Size"])]["_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k, cs in sorted(vals.items()):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    line = f"{k[0][:70]} grid={k[1]}"
    if "GRBM_GUI_ACTIVE" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        # this rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs (8 x the kernel's cycle count: the clock it implies with
        # the dispatch timestamps would otherwise be 19 GHz) and SQ_VALU_MFMA_BUSY_CYCLES summed over the 1024 SIMDs
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0
        line += (f"  MfmaUtil={m['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 256 * 4) * 100:.1f}% cycles={cyc:.0f} time={m['_ns'] / 1e6:.3f}ms "
                 f"clock={cyc / m['_ns']:.2f}GHz")
    if "SQ_WAVE_CYCLES" in m:
        wc = m["SQ_WAVE_CYCLES"]
        line += f"  wait_any={m.get('SQ_WAIT_ANY', 0) / wc * 100:.0f}% wait_inst={m.get('SQ_WAIT_INST_ANY', 0) / wc * 100:.0f}% active_inst={m.get('SQ_ACTIVE_INST_ANY', 0) / wc * 100:.0f}%"
    if "SQ_ACTIVE_INST_LDS" in m and "SQ_BUSY_CYCLES" in m:
        pass
    for c in ("SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_INST_CYCLES_VMEM"):
        if c in m:
            line += f"  {c}={m[c]:.3g}"
    print(line)
    if md and "GRBM_GUI_ACTIVE" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0
        wc = m.get("SQ_WAVE_CYCLES", 0.0)
        md_rows.append((k[0], k[1], m["_ns"] / 1e6, cyc / m["_ns"], m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024) * 100,
                        m.get("SQ_WAIT_ANY", 0) / wc * 100 if wc else float("nan"), m.get("SQ_WAIT_INST_ANY", 0) / wc * 100 if wc else float("nan"),
                        m.get("SQ_ACTIVE_INST_LDS", 0) / (cyc * 1024) * 100 if "SQ_ACTIVE_INST_LDS" in m else float("nan"),
                        m.get("SQ_LDS_BANK_CONFLICT", float("nan"))))
if md:
    with open(md, "w") as o:
        o.write("# MFMA-busy / wait / LDS counters of the GEMM kernels (rocprofv3 --pmc, three separate passes)\n\n"
                "Command (MI355X box): `bash scripts/pmc_gemm.sh 128` = `scripts/bench_gemm.py 128` (conv3 72x72 256->512, conv2 74x74 64->256 and linear5 at\n"
                "128 frames, every engine) under `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES`, `--pmc SQ_WAVE_CYCLES SQ_WAIT_ANY\n"
                "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY` and `--pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT ...`; summarised by `scripts/pmc_summary.py gemm --md`.\n\n"
                "* cycles = GRBM_GUI_ACTIVE / 8 (this rocprofv3 sums the counter over the 8 XCDs); clock = cycles / dispatch duration;\n"
                "* MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs); it is a fraction of the cycles AT THE CLOCK THE CHIP HELD, so the\n"
                "  fraction of the nominal (2.4 GHz) peak is MFMA busy x clock / 2.4;\n"
                "* wait any / wait inst = share of wave-cycles waiting on a counter (memory, LDS) / on instruction issue dependencies;\n"
                "* LDS busy = SQ_ACTIVE_INST_LDS / (cycles x 1024).\n\n"
                "| kernel | grid (threads) | ms | clock GHz | MFMA busy % | x clock/2.4 | wait any % | wait inst % | LDS busy % | LDS bank conflicts |\n"
                "|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|\n")
        for r in md_rows:
            o.write(f"| `{r[0]}` | {r[1]} | {r[2]:.3f} | {r[3]:.2f} | {r[4]:.1f} | {r[4] * r[3] / 2.4:.1f} | {r[5]:.0f} | {r[6]:.0f} | {r[7]:.1f} | {r[8]:.0f} |\n")
    print("wrote", md)
