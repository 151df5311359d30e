#!/bin/bash
# rocprofv3 kernel-trace summary of the default bench command (run on the GPU box via gpurun)
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_bench
rm -rf $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-latency --no-configs > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1
grep '^{' $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log | tail -1 > $GRAFT_REPO_ROOT/gpurun_out/prof_bench_line.json
ls $OUT
