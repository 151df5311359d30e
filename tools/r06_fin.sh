#!/bin/bash
# finalize pass A/B: parity tests that cover potential / predecessors / vector map of the tile-batch engine, then the headline step
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_tile_batch.py tests/test_gpu_bench_paths.py tests/test_gpu_c4.py -x -q 2>&1 | tail -3
timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu --no-latency --no-configs > $O/fin_bench.json 2> $O/fin_bench.err; tail -c 1500 $O/fin_bench.json
