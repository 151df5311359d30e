#!/bin/bash
# Round 5, call 4: where does k_tb_solve_q spend its cycles now (phase timers, -DMNAV_TB_TIMING build), C2 and C4 shapes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
MNAV_LIB=$PWD/tools/_variants/libmnav_timing.so MNAV_TRACE=1 N=1000 B=7168 REPS=2 timeout 300 python tools/gpu_tb_perf.py > gpurun_out/r05/tbperf_c2.json 2> gpurun_out/r05/tbperf_c2.err; echo rc=$?
grep -E "phase cycles|tile-batch:" gpurun_out/r05/tbperf_c2.err | tail -6; cat gpurun_out/r05/tbperf_c2.json
MNAV_LIB=$PWD/tools/_variants/libmnav_timing.so MNAV_TRACE=1 N=3163 B=4096 REPS=1 timeout 400 python tools/gpu_tb_perf.py > gpurun_out/r05/tbperf_c4.json 2> gpurun_out/r05/tbperf_c4.err; echo rc=$?
grep -E "phase cycles|tile-batch:" gpurun_out/r05/tbperf_c4.err | tail -4; cat gpurun_out/r05/tbperf_c4.json
