#!/bin/bash
# A/B of the last two changes of round 6 (static first ticket in the solve kernels; pair / nblk by multiply-high in k_tb_scan) against
# the build of 695f56e, then the whole GPU suite, smoke and the driver's bench command on the in-tree library (= both changes)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O; V=tools/_variants
timeout 600 python tools/gpu_tb_ab.py $V/libmnav_base695.so $V/libmnav_ticket.so $V/libmnav_ticket_magic.so $V/libmnav_base695.so mesh_navigation_amd/libmnav.so 2>&1 | tail -5
N=3163 B=4096 timeout 900 python tools/gpu_tb_ab.py $V/libmnav_base695.so mesh_navigation_amd/libmnav.so 2>&1 | tail -2
timeout 900 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "suite rc=$?"; tail -2 $O/gputest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
MNAV_TRACE=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line_ab4.json 2> $O/bench_line_ab4.err; echo "bench rc=$?"; python - <<PY
import json
d = json.load(open("$O/bench_line_ab4.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
c4 = d["configs"]["C4"]; print(c4["plans_per_s_batch"], c4["ms_per_batch"], c4["roofline"]["frac"], c4["ms_per_makeplan_single"])
print(d["configs"]["C5"]["plans_per_s"], d["ms_per_makeplan_single"])
PY
