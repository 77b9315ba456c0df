#!/bin/bash
# Profile bench.py on the GPU box with rocprofv3 (run from the repo root through gpurun).
#   pass 1: --kernel-trace --stats            -> per-kernel time
#   pass 2/3: --pmc FETCH_SIZE / WRITE_SIZE   -> HBM traffic (separate passes, MI355X_MICROARCH.md "rocprofv3 PMC slots")
# Output under gpurun_out/prof_*; scripts/summarize_rocprof.py turns it into profiles/.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
ARGS="${BENCH_ARGS:---steps 2 --warmup 2 --no-cpu-baseline --no-native40 --no-second-path --no-live-traffic --dtype ${BENCH_DTYPE:-f32}}"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python3 $ROOT/bench.py $ARGS > $OUT/prof_stats.json 2> $OUT/prof_stats.err || exit 1
if [ "$1" == "pmc" ]; then
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -- python3 $ROOT/bench.py $ARGS > $OUT/prof_fetch.json 2> $OUT/prof_fetch.err || exit 2
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -- python3 $ROOT/bench.py $ARGS > $OUT/prof_write.json 2> $OUT/prof_write.err || exit 3
fi
find $OUT/prof_stats -name "*.csv" | head
