#!/bin/bash
# Round 6: the whole -m gpu suite + one default bench.py (the driver's command)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "suite rc=$?"; tail -8 $O/gputest.log
MNAV_TRACE=1 timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc=$?"; python - <<PY
import json
d=json.load(open("$O/bench_line.json"))
c=d["configs"]
print("value", round(d["value"]), "ms_step", round(d["ms_per_step"],1), "frac", round(d["roofline"]["frac"],3), "kernel", d["roofline"]["kernel"][:40], "engine_us", round(d["roofline"]["avg_launch_us"]))
print("paths_only", {k:(round(v["plans_per_s"]), round(v["roofline"]["frac"],3)) for k,v in c["C2_paths_only"].items() if k.startswith("batch")})
print("C5", round(c["C5"]["plans_per_s"]), "C3 batch", round(c["C3"]["plans_per_s_batch"]), "single", round(c["C3"]["ms_per_plan_single"],1), "infl", c["C3"]["cost_stack_used"]["inflation_wave"])
print("C4", round(c["C4"]["plans_per_s_batch"]), c["C4"]["roofline"]["frac"], c["C4"]["roofline"]["kernel"][:30], "single", c["C4"]["ms_per_makeplan_single"], c["C4"]["ms_per_makeplan_single_p95"])
print("single C2", d["ms_per_makeplan_single"], d["ms_per_makeplan_single_p95"])
PY
grep -h "inflation it" $O/bench_line.err | tail -4 | cut -c1-260
