#!/bin/bash
# Round 6: solve kernel x band at smaller batches on the 1M mesh (what `auto` picks between 97 plans and the headline batch)
cd $GRAFT_REPO_ROOT
for B in 128 512 2048; do echo "B=$B"; B=$B timeout 600 python tools/gpu_tb_ab.py mesh_navigation_amd/libmnav.so@MNAV_TB_KERNEL=0 mesh_navigation_amd/libmnav.so@MNAV_TB_KERNEL=1,MNAV_TB_BAND_MULT=2 mesh_navigation_amd/libmnav.so@MNAV_TB_KERNEL=1,MNAV_TB_BAND_MULT=4 2>&1 | tail -3; done
