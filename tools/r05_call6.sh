#!/bin/bash
# Round 5, call 6: the new finalize pass (k_tb_finalize), the asynchronous engine with reserved ticket counts
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_tile_batch.py tests/test_gpu_bench_paths.py tests/test_gpu_edge_cases.py tests/test_gpu_paths_only.py tests/test_gpu_async.py -m gpu -x -q > gpurun_out/r05/call6_first.log 2>&1; echo "tests rc=$?"; tail -12 gpurun_out/r05/call6_first.log
timeout 500 python bench.py --steps 5 --warmup 2 --no-cpu --no-configs > gpurun_out/r05/bench_call6.json 2> gpurun_out/r05/bench_call6.err; echo "bench rc=$?"; python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r05/bench_call6.json").read().strip().splitlines()[-1])
    print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["roofline"].get("propagation_ms_per_step"))
    print({k:v for k,v in d.items() if "latency" in k or "single" in k})
    print(json.dumps(d.get("latency", d.get("single_plan", {})))[:600])
except Exception as e: print("bench parse failed", e)
PY
tail -3 gpurun_out/r05/bench_call6.err
timeout 400 python tools/gpu_async_tune.py 1000 5 tiled,async > gpurun_out/r05/async_tune_1000c.json 2> gpurun_out/r05/async_tune_1000c.err; echo "tune1000 rc=$?"; cat gpurun_out/r05/async_tune_1000c.json; tail -2 gpurun_out/r05/async_tune_1000c.err
