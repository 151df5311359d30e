#!/bin/bash
# Round 5, second GPU call: first hardware run of the DPP sweep (k_tb_solve_q) and of the ticket-queue engine; whole GPU suite; timings.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
export MNAV_VERBOSE=0
timeout 600 python -m pytest tests/test_gpu_tile_batch.py tests/test_gpu_async.py -m gpu -x -q > gpurun_out/r05/call2_first.log 2>&1; echo "tile_batch+async tests rc=$?"; tail -12 gpurun_out/r05/call2_first.log
timeout 400 python tools/gpu_async_tune.py 1000 5 > gpurun_out/r05/async_tune_1000.json 2> gpurun_out/r05/async_tune_1000.err; echo "tune1000 rc=$?"; cat gpurun_out/r05/async_tune_1000.json
timeout 500 python bench.py --steps 5 --warmup 2 --no-cpu > gpurun_out/r05/bench_call2.json 2> gpurun_out/r05/bench_call2.err; echo "bench rc=$?"; python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r05/bench_call2.json").read().strip().splitlines()[-1])
    print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["roofline"].get("propagation_ms_per_step"))
    c=d.get("configs",{})
    for k,v in c.items():
        print(k, {kk:vv for kk,vv in v.items() if isinstance(vv,(int,float))})
except Exception as e: print("bench parse failed", e)
PY
tail -5 gpurun_out/r05/bench_call2.err
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_tile_batch.py --deselect tests/test_gpu_async.py > gpurun_out/r05/call2_suite.log 2>&1; echo "suite rc=$?"; tail -15 gpurun_out/r05/call2_suite.log
timeout 500 python tools/gpu_async_tune.py 3163 3 > gpurun_out/r05/async_tune_3163.json 2> gpurun_out/r05/async_tune_3163.err; echo "tune3163 rc=$?"; cat gpurun_out/r05/async_tune_3163.json
