#!/bin/bash
# Round 6: the register-resident tile-batch solve (k_tbv_solve) -- parity suite of the tile-batch engine with the kernel forced, then A/B against k_tb_solve_q
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
MNAV_TB_KERNEL=1 timeout 900 python -m pytest tests/test_gpu_tile_batch.py tests/test_gpu_bench_paths.py tests/test_gpu_paths_only.py -x -q > $O/tbv_tests.log 2>&1; echo "tests rc=$?"; tail -15 $O/tbv_tests.log
for k in 0 1; do
  MNAV_LIB=$PWD/tools/_variants/libmnav_timing.so MNAV_TB_KERNEL=$k MNAV_TRACE=1 N=1000 B=7168 REPS=1 timeout 300 python tools/gpu_tb_perf.py > $O/tbv_perf_$k.json 2> $O/tbv_perf_$k.err
  grep -h "tile-batch:\|phase cycles" $O/tbv_perf_$k.err | tail -2; tail -1 $O/tbv_perf_$k.json | cut -c1-300
done
timeout 600 python tools/gpu_tb_ab.py mesh_navigation_amd/libmnav.so@MNAV_TB_KERNEL=0 mesh_navigation_amd/libmnav.so@MNAV_TB_KERNEL=1 2>&1 | tail -3
