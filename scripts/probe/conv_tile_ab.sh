cd /tmp && export TMPDIR=/tmp
for t in 256 128; do
rm -rf /tmp/kt$t
GOALNET_BF16_TILE=$t rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$t -- python3 $GRAFT_REPO_ROOT/scripts/conv_fwd_probe.py --dtype bf16 --reps 3 > /dev/null 2>&1
echo "== tile $t"; python3 - <<PY
import csv,glob
for f in glob.glob("/tmp/kt$t/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_" in r["Kernel_Name"]:
            print(r["Kernel_Name"][:90], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6, "ms")
PY
done
