#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout 200 python tools/gpu_infl_one.py 7515 > $O/infl_7515.out 2> $O/infl_7515.err; cat $O/infl_7515.out; grep "verification" $O/infl_7515.err | head -12; grep -c "inflation it" $O/infl_7515.err
timeout 600 python tools/gpu_fin_ab.py mesh_navigation_amd/libmnav.so tools/_variants/libmnav_fin1.so tools/_variants/libmnav_fin2.so tools/_variants/libmnav_fin2o4.so tools/_variants/libmnav_fin4o4.so 2>&1 | tail -5
MNAV_LIB=$PWD/tools/_variants/libmnav_timing.so MNAV_TRACE=1 N=3163 B=4096 REPS=1 timeout 600 python tools/gpu_tb_perf.py > $O/tbv_perf_c4.json 2> $O/tbv_perf_c4.err; grep -h "tile-batch:\|phase cycles" $O/tbv_perf_c4.err | tail -2
timeout 600 python -m pytest tests/test_gpu_plugin_dropin.py -x -q 2>&1 | tail -3
