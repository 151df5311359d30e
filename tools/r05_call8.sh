#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout 600 python tools/gpu_fin_ab.py mesh_navigation_amd/libmnav.so tools/_variants/libmnav_fin_4_3.so 2>&1 | tee gpurun_out/r05/fin_ab2.log
timeout 1200 python -m pytest tests/test_gpu_tile_batch.py tests/test_gpu_bench_paths.py tests/test_gpu_edge_cases.py tests/test_gpu_paths_only.py tests/test_gpu_plugin_dropin.py tests/test_gpu_sharded_processes.py tests/test_gpu_sharded.py -m gpu -x -q > gpurun_out/r05/call8_tests.log 2>&1; echo "tests rc=$?"; tail -12 gpurun_out/r05/call8_tests.log
bash tools/prof_r05.sh c2sq > gpurun_out/r05/prof_c2sq.log 2>&1; tail -3 gpurun_out/r05/prof_c2sq.log
