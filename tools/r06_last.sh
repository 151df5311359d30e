#!/bin/bash
# last run of the round: the whole GPU suite, the driver's bench command, the soak
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout 2700 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
MNAV_TRACE=1 timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc=$?"; tail -c 300 $O/bench_line.json; echo
for f in /sys/fs/cgroup/cpu.stat; do [ -r $f ] && grep -h "throttled" $f | tr '\n' ' '; done; echo
timeout 900 python tools/gpu_soak.py 22 > $O/soak.json 2> $O/soak.err; tail -c 400 $O/soak.json; echo
