#!/bin/bash
# Round 5, call 5: deeper stream prefetch in k_tb_solve_q (8 chunks in flight, two rotations per loop body); the asynchronous engine with bands
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout 600 python -m pytest tests/test_gpu_tile_batch.py tests/test_gpu_async.py -m gpu -x -q > gpurun_out/r05/call5_first.log 2>&1; echo "tile_batch+async tests rc=$?"; tail -8 gpurun_out/r05/call5_first.log
timeout 500 python bench.py --steps 5 --warmup 2 --no-cpu --no-latency --no-configs > gpurun_out/r05/bench_call5.json 2> gpurun_out/r05/bench_call5.err; echo "bench rc=$?"; python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r05/bench_call5.json").read().strip().splitlines()[-1])
    print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["roofline"].get("propagation_ms_per_step"))
except Exception as e: print("bench parse failed", e)
PY
timeout 400 python tools/gpu_async_tune.py 1000 5 > gpurun_out/r05/async_tune_1000b.json 2> gpurun_out/r05/async_tune_1000b.err; echo "tune1000 rc=$?"; cat gpurun_out/r05/async_tune_1000b.json; tail -2 gpurun_out/r05/async_tune_1000b.err
timeout 300 python tools/gpu_c4_batch.py 4096 > gpurun_out/r05/c4_call5.json 2> gpurun_out/r05/c4_call5.err; echo "c4 rc=$?"; cut -c1-700 gpurun_out/r05/c4_call5.json
timeout 500 python tools/gpu_async_tune.py 3163 3 tiled,async,async_band8 > gpurun_out/r05/async_tune_3163b.json 2> gpurun_out/r05/async_tune_3163b.err; echo "tune3163 rc=$?"; cat gpurun_out/r05/async_tune_3163b.json; tail -3 gpurun_out/r05/async_tune_3163b.err
