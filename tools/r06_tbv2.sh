#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
for k in 0 1; do
  MNAV_LIB=$PWD/tools/_variants/libmnav_timing.so MNAV_TB_KERNEL=$k MNAV_TRACE=1 N=1000 B=7168 REPS=1 timeout 300 python tools/gpu_tb_perf.py > $O/tbv_perf_$k.json 2> $O/tbv_perf_$k.err
  grep -h "tile-batch:\|phase cycles" $O/tbv_perf_$k.err | tail -3; tail -1 $O/tbv_perf_$k.json | cut -c1-300
done
