R=$GRAFT_REPO_ROOT
A="--steps 6 --warmup 2 --no-cpu-baseline --no-native40 --no-second-path --no-live-traffic --dtype bf16"
python3 $R/bench.py $A > $R/gpurun_out/swap_bench_new.json 2> $R/gpurun_out/swap_bench_new.err || exit 1
GOALNET_PERSISTENT=0 python3 $R/bench.py $A > $R/gpurun_out/swap_bench_np.json 2> $R/gpurun_out/swap_bench_np.err || exit 2
GOALNET_LIB_PATH=$R/cvml_goalnet_amd/csrc/build/libgoalnet_noswap.so python3 $R/bench.py $A > $R/gpurun_out/swap_bench_old.json 2> $R/gpurun_out/swap_bench_old.err || exit 2
python3 - <<PY
import json
for t in ("new","np","old"):
    d=json.loads(open("$R/gpurun_out/swap_bench_%s.json"%t).read().strip().splitlines()[-1])
    print(t, d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], {k:(round(v["tflops"]),round(v["ms_per_launch"],3)) for k,v in d.get("other_kernels",{}).items()} if "other_kernels" in d else "")
PY
