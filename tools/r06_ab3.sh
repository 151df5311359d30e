#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_tile_batch.py tests/test_gpu_bench_paths.py tests/test_gpu_paths_only.py tests/test_gpu_c4.py -x -q 2>&1 | tail -3
timeout 600 python tools/gpu_tb_ab.py mesh_navigation_amd/libmnav.so mesh_navigation_amd/libmnav.so@MNAV_TB_KERNEL=0 2>&1 | tail -2
for B in 128 2048; do B=$B timeout 300 python tools/gpu_tb_ab.py mesh_navigation_amd/libmnav.so 2>&1 | tail -1; done
N=3163 B=4096 timeout 900 python tools/gpu_tb_ab.py mesh_navigation_amd/libmnav.so 2>&1 | tail -1
