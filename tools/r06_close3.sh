#!/bin/bash
# soak of the final build of round 6
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout 330 python tools/gpu_soak.py 24 > $O/soak.json 2> $O/soak.err; tail -c 700 $O/soak.json; echo
