#!/bin/bash
# how many workgroups the clean-up of the other buffer may have before the small kernels at the start of a batch starve
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O; L=tools/_variants/libmnav_wgs.so
timeout 400 python tools/gpu_tb_modes.py $L@MNAV_TB_FILL_WGS=12 $L@MNAV_TB_FILL_WGS=6 $L@MNAV_TB_FILL_WGS=24 $L@MNAV_TB_FILL_WGS=48 2>&1 | tee $O/fill_wgs.txt
