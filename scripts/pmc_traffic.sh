#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and duration of the fp32 GEMM kernels, one op list at a time.
# Usage (through gpurun, from the repo root): [ENV=...] bash scripts/pmc_traffic.sh "<probe args>" <tag>
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_traffic_${2:-x}
ARGS="$1"
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ROOT/scripts/f32_traffic_probe.py $ARGS > /dev/null 2> $OUT.fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ROOT/scripts/f32_traffic_probe.py $ARGS > /dev/null 2> $OUT.write.err || exit 2
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/scripts/f32_traffic_probe.py $ARGS > /dev/null 2> $OUT.stats.err || exit 3
python3 $ROOT/scripts/pmc_traffic_summary.py $OUT
