#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout 900 python tools/gpu_fin_ab.py mesh_navigation_amd/libmnav.so tools/_variants/libmnav_fin_8_3.so tools/_variants/libmnav_fin_4_3.so tools/_variants/libmnav_fin_16_4.so 2>&1 | tee gpurun_out/r05/fin_ab.log
timeout 900 python -m pytest tests/test_gpu_tile_batch.py tests/test_gpu_bench_paths.py tests/test_gpu_sharded.py tests/test_gpu_plugin_dropin.py -m gpu -x -q > gpurun_out/r05/call7_tests.log 2>&1; echo "tests rc=$?"; tail -12 gpurun_out/r05/call7_tests.log
