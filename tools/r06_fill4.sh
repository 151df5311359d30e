#!/bin/bash
# the other buffer cleaned at the END of a call (tb_clean_other): engine / wall per batch, the tile-batch parity tests, the bench line
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout 120 python tools/gpu_tb_modes.py mesh_navigation_amd/libmnav.so 2>&1 | tee $O/fill_end_of_call.txt
timeout 200 python -m pytest tests/test_gpu_tile_batch.py tests/test_gpu_bench_paths.py tests/test_gpu_paths_only.py -x -q 2>&1 | tail -2
MNAV_TRACE=1 timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --skip-c4 > $O/bench_line_fill4.json 2> $O/bench_line_fill4.err; echo "bench rc=$?"; python - <<PY
import json
d = json.load(open("$O/bench_line_fill4.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
p = d["configs"]["C2_paths_only"]; print({k: (round(v["plans_per_s"]), round(v["propagation_ms"], 1)) for k, v in p.items() if isinstance(v, dict)})
PY
