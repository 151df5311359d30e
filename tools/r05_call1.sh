#!/bin/bash
# Round 5, first GPU call: (1) parity of the CVP batches bench.py times, (2) the two variants written blind at the end of round 4
# (async tile engine, pipelined sweeps), (3) the C4 profile, (4) SQ_* counters of k_tb_solve_q on the headline command.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_c3_bench_config.py -m gpu -x -q > gpurun_out/r05/c3_batch_tests.log 2>&1; echo "c3 tests rc=$?"; tail -15 gpurun_out/r05/c3_batch_tests.log
bash tools/r05_first_run.sh
bash tools/prof_r05.sh c4
bash tools/prof_r05.sh c2sq
