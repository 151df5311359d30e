#!/bin/bash
# last run of round 6 (first ticket without an atomic, multiply-high division in the scan, 48-workgroup clean-up of the other
# buffer): whole GPU suite, smoke, the driver's bench command, kernel trace + traffic + SQ passes of the headline command, trace +
# traffic of the C4 leg
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "suite rc=$?"; tail -2 $O/gputest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
MNAV_TRACE=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc=$?"; python - <<PY
import json
d = json.load(open("$O/bench_line.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
c4 = d["configs"]["C4"]; print(c4["plans_per_s_batch"], c4["ms_per_batch"], c4["roofline"]["frac"], c4["ms_per_makeplan_single"])
print(d["configs"]["C5"]["plans_per_s"], d["ms_per_makeplan_single"])
PY
bash tools/prof_r06.sh c2 > $O/prof_c2.log 2>&1; echo "c2 prof done"
bash tools/prof_r06.sh c2sq > $O/prof_c2sq.log 2>&1; echo "c2sq prof done"
PROF_SKIP_SQ=1 bash tools/prof_r06.sh c4 > $O/prof_c4.log 2>&1; echo "c4 prof done"
