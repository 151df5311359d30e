#!/bin/bash
# closing run of round 6 after the end-of-launch atomics went (695f56e): the whole GPU suite, smoke, the driver's bench command,
# kernel trace + HBM traffic of the headline command and of the C4 leg, a short soak
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "suite rc=$?"; tail -2 $O/gputest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
MNAV_TRACE=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc=$?"; tail -c 200 $O/bench_line.json; echo
bash tools/prof_r06.sh c2 > $O/prof_c2.log 2>&1; echo "c2 prof done"
PROF_SKIP_SQ=1 bash tools/prof_r06.sh c4 > $O/prof_c4.log 2>&1; echo "c4 prof done"
timeout 240 python tools/gpu_soak.py 22 > $O/soak.json 2> $O/soak.err; tail -c 500 $O/soak.json; echo
