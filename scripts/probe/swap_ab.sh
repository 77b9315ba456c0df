cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/scripts/probe/swap_hash.py > $R/gpurun_out/swap_hash_new.txt 2>&1 || { tail -5 $R/gpurun_out/swap_hash_new.txt; exit 1; }
GOALNET_LIB_PATH=$R/cvml_goalnet_amd/csrc/build/libgoalnet_noswap.so python3 $R/scripts/probe/swap_hash.py > $R/gpurun_out/swap_hash_old.txt 2>&1
if cmp -s $R/gpurun_out/swap_hash_new.txt $R/gpurun_out/swap_hash_old.txt; then echo "HASHES IDENTICAL"; else echo "HASHES DIFFER"; diff $R/gpurun_out/swap_hash_new.txt $R/gpurun_out/swap_hash_old.txt | head -20; fi
for v in new old; do
  if [ $v = old ]; then export GOALNET_LIB_PATH=$R/cvml_goalnet_amd/csrc/build/libgoalnet_noswap.so; fi
  rm -rf /tmp/kt$v
  rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$v -- python3 $R/scripts/conv_fwd_probe.py --dtype bf16 --reps 4 > /dev/null 2>&1
  echo "== $v"; python3 - <<PY
import csv,glob
for f in glob.glob("/tmp/kt$v/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_" in r["Kernel_Name"]:
            print(r["Kernel_Name"][:60], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6, "ms")
PY
done
