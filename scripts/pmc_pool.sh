#!/bin/bash
# Counter pass over the pool / BatchNorm kernels (scripts/bench_gemm.py N pool); run through gpurun from the repo root.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
N=${1:-256}
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/pmc_a $OUT/pmc_b $OUT/pmc_c
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_a -- python3 $ROOT/scripts/bench_gemm.py $N pool > /dev/null 2> $OUT/pmc_a.err || exit 1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_b -- python3 $ROOT/scripts/bench_gemm.py $N pool > /dev/null 2> $OUT/pmc_b.err || exit 2
echo done
