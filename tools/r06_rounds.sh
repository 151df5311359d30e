#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout 2700 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 900 python bench.py --config C4 --steps 10 --warmup 2 > $O/c4_sharded.json 2> $O/c4_sharded.err; tail -c 1200 $O/c4_sharded.json
