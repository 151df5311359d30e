#!/bin/bash
# Round 6, second measurement call: the register-resident sweep routine on its own (parity + cycles per block), and the inflation
# wave after an idle GPU (is the 85 ms mode of the driver's line the clock ramp after bench.py's CPU analysis?)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout 300 python tools/gpu_tbv_micro.py > $O/tbv_micro.json 2> $O/tbv_micro.err; tail -c 1500 $O/tbv_micro.json; tail -3 $O/tbv_micro.err
timeout 300 python tools/gpu_infl_bimodal.py 3 0 0 2.0 > $O/infl_idle.json 2> $O/infl_idle.err; grep repeat $O/infl_idle.err
timeout 300 python tools/gpu_infl_bimodal.py 3 0 0 0.0 > $O/infl_busy.json 2> $O/infl_busy.err; grep repeat $O/infl_busy.err
