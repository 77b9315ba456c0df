# per-kernel times of the bf16 bench step: this build vs the build named by OLD_LIB (kernel trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
A="--steps 3 --warmup 2 --no-cpu-baseline --no-native40 --no-second-path --no-live-traffic --dtype bf16"
for v in ${VARIANTS:-new new2}; do   # add "old" with OLD_LIB=<variant .so under csrc/build> for an A/B against another build
  unset GOALNET_LIB_PATH
  if [ $v = old ]; then export GOALNET_LIB_PATH=$R/cvml_goalnet_amd/csrc/build/${OLD_LIB:-libgoalnet_noswap.so}; fi
  rm -rf /tmp/rt$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rt$v -- python3 $R/bench.py $A > /tmp/rt$v.json 2>/dev/null
  echo "== $v"; python3 - <<PY
import csv,glob,collections,json
print(json.loads(open("/tmp/rt$v.json").read().strip().splitlines()[-1])["ms_per_step"])
for f in glob.glob("/tmp/rt$v/**/*_kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in sorted(rows[:14], key=lambda r: r["Name"]):
        if "gemm" in r["Name"] or "adam" in r["Name"]:
            print("%-90s %4s %9.3f" % (r["Name"][29:119].replace("(anonymous namespace)::",""), r["Calls"], float(r["AverageNs"])/1e6))
PY
done
