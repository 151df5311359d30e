#!/bin/bash
# HBM traffic counters of the bench command, one counter per pass (MI355X_MICROARCH.md HBM section)
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$c
  rm -rf $OUT
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu --no-latency --no-configs > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1
  ls $OUT | head
done
