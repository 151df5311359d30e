#!/bin/bash
# the asynchronous engine after a change: its tests, then the single-plan / C5 / C4 legs of the bench
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_async.py tests/test_gpu_edge_cases.py tests/test_gpu_sharded.py tests/test_gpu_plugin_dropin.py tests/test_gpu_adapter.py -x -q 2>&1 | tail -3
timeout 1200 python bench.py > $O/async_bench.json 2> $O/async_bench.err; tail -c 300 $O/async_bench.json
