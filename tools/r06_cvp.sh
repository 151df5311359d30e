#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout 2700 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 1200 python bench.py --skip-c4 > $O/cvp_bench.json 2> $O/cvp_bench.err; tail -c 200 $O/cvp_bench.json
