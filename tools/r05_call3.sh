#!/bin/bash
# Round 5, third GPU call: the software-pipelined DPP sweep; which engine leaves the 10M plan short of its fixed point
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout 600 python -m pytest tests/test_gpu_tile_batch.py tests/test_gpu_bench_paths.py -m gpu -x -q > gpurun_out/r05/call3_first.log 2>&1; echo "tile_batch tests rc=$?"; tail -8 gpurun_out/r05/call3_first.log
timeout 500 python bench.py --steps 5 --warmup 2 --no-cpu --no-latency --no-configs > gpurun_out/r05/bench_call3.json 2> gpurun_out/r05/bench_call3.err; echo "bench rc=$?"; python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r05/bench_call3.json").read().strip().splitlines()[-1])
    print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["roofline"].get("propagation_ms_per_step"))
except Exception as e: print("bench parse failed", e)
PY
timeout 300 python tools/gpu_c4_batch.py 4096 > gpurun_out/r05/c4_call3.json 2> gpurun_out/r05/c4_call3.err; echo "c4 rc=$?"; cut -c1-900 gpurun_out/r05/c4_call3.json
timeout 500 python tools/gpu_async_tune.py 3163 3 tiled,async > gpurun_out/r05/async_tune_3163.json 2> gpurun_out/r05/async_tune_3163.err; echo "tune3163 rc=$?"; cat gpurun_out/r05/async_tune_3163.json; tail -3 gpurun_out/r05/async_tune_3163.err
