#!/bin/bash
# Round 6: the measurement run behind profiles/r06_* and DESIGN.md section 4 (one gpurun call):
#   gpurun --timeout 4500 -- 'bash tools/r06_measure.sh'
# 1. the whole -m gpu suite  2. `python bench.py --gpus 1 --steps 20 --warmup 5` (the driver's command) with the library's trace on stderr
# 3. kernel trace + HBM traffic of the headline command, of the C4 leg, and the SQ_* passes of both (tools/prof_r06.sh;
#    tools/summarize_r06.py turns them into profiles/r06_*)  4. the soak (tools/gpu_soak.py) and the inflation fuzz
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "suite rc=$?"; tail -4 $O/gputest.log
MNAV_TRACE=1 timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc=$?"; tail -c 300 $O/bench_line.json; echo
grep -h "inflation it" $O/bench_line.err | tail -3 | cut -c1-200
bash tools/prof_r06.sh c2 > $O/prof_c2.log 2>&1
bash tools/prof_r06.sh c2sq > $O/prof_c2sq.log 2>&1
bash tools/prof_r06.sh c4 > $O/prof_c4.log 2>&1
bash tools/prof_r06.sh cvpsq > $O/prof_cvpsq.log 2>&1
timeout 900 python tools/gpu_soak.py 22 > $O/soak.json 2> $O/soak.err; tail -c 600 $O/soak.json; echo
timeout 500 python tools/gpu_infl_fuzz.py 0 300 > $O/infl_fuzz.json 2> $O/infl_fuzz.err; tail -c 400 $O/infl_fuzz.json; echo
PERF_BATCHES=128 timeout 300 python tools/gpu_cvp_perf.py > $O/cvp_perf.json 2> $O/cvp_perf.err; tail -2 $O/cvp_perf.json | cut -c1-400
