#!/bin/bash
# the whole GPU suite, then the default bench line
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee $O/full_tests.log
timeout 1200 python bench.py > $O/full_bench.json 2> $O/full_bench.err; tail -c 600 $O/full_bench.json
