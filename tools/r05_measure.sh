#!/bin/bash
# Round 5: the measurement run behind profiles/r05_* and DESIGN.md section 4 (one gpurun call, ~12 GPU-minutes):
#   gpurun --timeout 3000 -- 'bash tools/r05_measure.sh'
# 1. the whole -m gpu suite  2. one default `python bench.py` (the driver's command)  3. kernel trace + HBM traffic of the headline
# command, of the C4 leg, and the SQ_* passes of both (tools/prof_r05.sh; tools/summarize_r05.py turns them into profiles/r05_*).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05/gputest.log 2>&1; echo "suite rc=$?"; tail -6 gpurun_out/r05/gputest.log
timeout 900 python bench.py > gpurun_out/r05/bench_line.json 2> gpurun_out/r05/bench_line.err; echo "bench rc=$?"; tail -c 600 gpurun_out/r05/bench_line.json | head -c 600; echo
bash tools/prof_r05.sh c2 > gpurun_out/r05/prof_c2.log 2>&1
bash tools/prof_r05.sh c2sq > gpurun_out/r05/prof_c2sq.log 2>&1
bash tools/prof_r05.sh c4 > gpurun_out/r05/prof_c4.log 2>&1
PERF_BATCHES=128 timeout 300 python tools/gpu_cvp_perf.py > gpurun_out/r05/cvp_perf.json 2> gpurun_out/r05/cvp_perf.err; tail -2 gpurun_out/r05/cvp_perf.json | cut -c1-400
