#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_tile_batch.py tests/test_gpu_bench_paths.py tests/test_gpu_paths_only.py -x -q 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python tools/gpu_fin_ab.py mesh_navigation_amd/libmnav.so tools/_variants/libmnav_finpf2.so 2>&1 | tail -2
N=3163 B=4096 timeout 900 python tools/gpu_tb_ab.py mesh_navigation_amd/libmnav.so 2>&1 | tail -1
