#!/bin/bash
# Counter pass over the pool / BatchNorm kernels (scripts/bench_gemm.py N pool); run through gpurun from the repo root.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
N=${1:-256}
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/pmc_a $OUT/pmc_b $OUT/pmc_c
WHICH=${2:-pool}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_a -- python3 $ROOT/scripts/bench_gemm.py $N $WHICH > /dev/null 2> $OUT/pmc_a.err || exit 1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_b -- python3 $ROOT/scripts/bench_gemm.py $N $WHICH > /dev/null 2> $OUT/pmc_b.err || exit 2
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_c -- python3 $ROOT/scripts/bench_gemm.py $N $WHICH > /dev/null 2> $OUT/pmc_c.err || exit 3
echo done
