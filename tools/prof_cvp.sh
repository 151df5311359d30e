#!/bin/bash
# rocprofv3 kernel-trace summary of the CVP planner on C3 (single plan and batches up to 128).
# MNAV_NO_GRAPH=1: rocprofv3 segfaults inside the hipGraph replays of the step kernels, so the steps are
# launched one by one here; kernel durations are unaffected, launch gaps are not representative.
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_cvp
rm -rf $OUT
MNAV_NO_GRAPH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o cvp -- env PERF_BATCHES=1,128 python $GRAFT_REPO_ROOT/tools/gpu_cvp_perf.py > $GRAFT_REPO_ROOT/gpurun_out/prof_cvp.log 2>&1
head -8 $OUT/cvp_kernel_stats.csv | cut -c1-170
