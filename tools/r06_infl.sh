#!/bin/bash
# Round 6: the inflation wave with the reset step before a serial band: the layer tests, then the deterministic fuzz
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_layers.py -x -q > $O/infl_tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/infl_tests.log
timeout 700 python tools/gpu_infl_fuzz.py 0 ${FUZZ_S:-420} > $O/infl_fuzz.json 2> $O/infl_fuzz.err; tail -c 1500 $O/infl_fuzz.json; tail -3 $O/infl_fuzz.err
