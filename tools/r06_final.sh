#!/bin/bash
# after the last code change of the round: the driver's bench command (with the library's trace) and the soak
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
MNAV_TRACE=1 timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc=$?"; tail -c 300 $O/bench_line.json; echo
timeout 900 python tools/gpu_soak.py 22 > $O/soak.json 2> $O/soak.err; tail -c 600 $O/soak.json; echo
timeout 500 python tools/gpu_infl_fuzz.py 0 300 > $O/infl_fuzz.json 2> $O/infl_fuzz.err; tail -c 400 $O/infl_fuzz.json; echo
timeout 600 python tools/gpu_async_tune.py > $O/async_tune.json 2> $O/async_tune.err; tail -c 1500 $O/async_tune.json; echo
