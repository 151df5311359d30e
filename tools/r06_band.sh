#!/bin/bash
# Round 6: band width x solve kernel, C2 (1M, 7168 plans) and C4 (10M, 4096 plans): engine ms per batch
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
echo "C2"; timeout 900 python tools/gpu_tb_ab.py mesh_navigation_amd/libmnav.so@MNAV_TB_KERNEL=1,MNAV_TB_BAND_MULT=2 mesh_navigation_amd/libmnav.so@MNAV_TB_KERNEL=1,MNAV_TB_BAND_MULT=3 mesh_navigation_amd/libmnav.so@MNAV_TB_KERNEL=1,MNAV_TB_BAND_MULT=4 mesh_navigation_amd/libmnav.so@MNAV_TB_KERNEL=1,MNAV_TB_BAND_MULT=1.5 2>&1 | tail -4
echo "C4"; N=3163 B=4096 timeout 1500 python tools/gpu_tb_ab.py mesh_navigation_amd/libmnav.so@MNAV_TB_KERNEL=0 mesh_navigation_amd/libmnav.so@MNAV_TB_KERNEL=1,MNAV_TB_BAND_MULT=2 mesh_navigation_amd/libmnav.so@MNAV_TB_KERNEL=1,MNAV_TB_BAND_MULT=3 mesh_navigation_amd/libmnav.so@MNAV_TB_KERNEL=1,MNAV_TB_BAND_MULT=4 mesh_navigation_amd/libmnav.so@MNAV_TB_KERNEL=1,MNAV_TB_BAND_MULT=6 2>&1 | tail -5
