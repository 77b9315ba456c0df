set -o pipefail
cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err; echo "bench rc=$?"
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write
BENCH_DTYPE=f32 bash scripts/profile_bench.sh > gpurun_out/r03_prof_f32.log 2>&1; echo "prof f32 rc=$?"
python scripts/summarize_rocprof.py r03 > gpurun_out/r03_sum_f32.log 2>&1
rm -rf gpurun_out/prof_stats
BENCH_DTYPE=bf16 bash scripts/profile_bench.sh > gpurun_out/r03_prof_bf16.log 2>&1; echo "prof bf16 rc=$?"
python scripts/summarize_rocprof.py r03_bf16 > gpurun_out/r03_sum_bf16.log 2>&1
bash scripts/pmc_traffic.sh "--ops fwd3,dgrad3,wgrad3,l5fwd,l5dx,l5dw" c3 > gpurun_out/r03_traffic_c3.md 2>&1
bash scripts/pmc_traffic.sh "--ops fwd2,dgrad2,wgrad2" c2 > gpurun_out/r03_traffic_c2.md 2>&1
rm -rf gpurun_out/prof_loop; bash scripts/profile_loop.sh > gpurun_out/r03_prof_loop.log 2>&1
TOP=60 python scripts/top_kernels.py prof_loop 180 > gpurun_out/r03_loop40_top.txt 2>&1
python scripts/bench_loop.py --videos 5 > gpurun_out/r03_loop40_noprof.json 2>/dev/null
python scripts/bench_launch_gap.py > gpurun_out/r03_launch_gap.json 2>/dev/null
ls profiles | tail -5
