#!/bin/bash
# Counter passes over scripts/bench_gemm.py (run through gpurun from the repo root): MFMA busy / GPU active, wave waits, LDS.
# Usage: [GOALNET_BF16_TILE=256] bash scripts/pmc_gemm.sh [frames]
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
N=${1:-128}
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/pmc_a $OUT/pmc_b $OUT/pmc_c
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_a -- python3 $ROOT/scripts/bench_gemm.py $N > /dev/null 2> $OUT/pmc_a.err || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_b -- python3 $ROOT/scripts/bench_gemm.py $N > /dev/null 2> $OUT/pmc_b.err || exit 2
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc_c -- python3 $ROOT/scripts/bench_gemm.py $N > /dev/null 2> $OUT/pmc_c.err || exit 3
echo done
