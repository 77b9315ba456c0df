#!/bin/bash
# rocprofv3 kernel stats of the reference-operating-point loop (scripts/bench_loop.py); run through gpurun from the repo root.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_loop
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_loop -- python3 $ROOT/scripts/bench_loop.py --videos 2 ${LOOP_ARGS} > $OUT/prof_loop.json 2> $OUT/prof_loop.err || exit 1
tail -1 $OUT/prof_loop.json
