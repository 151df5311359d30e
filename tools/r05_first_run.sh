#!/bin/bash
# First GPU call of round 5: the two variants that were written after round 4's GPU minutes were spent, each under its own
# timeout (their kernels have never run on hardware), results under gpurun_out/r05_first/.  ~6 GPU-minutes.
#   gpurun --timeout 900 -- 'bash tools/r05_first_run.sh'
set -u
out=gpurun_out/r05_first; mkdir -p $out
export MNAV_VERBOSE=1 MNAV_ASYNC_MAX_S=5
echo "== async engine: opt-in parity test" | tee $out/log.txt
MNAV_TEST_ASYNC=1 timeout 120 python -m pytest tests/test_gpu_async.py -m gpu -x -q > $out/async_pytest.log 2>&1; echo "rc=$?" | tee -a $out/log.txt
tail -5 $out/async_pytest.log | tee -a $out/log.txt
echo "== async engine: parity + ms against tiled / tile_batch (300^2, then 1000^2)" | tee -a $out/log.txt
timeout 150 python tools/gpu_async_engine.py 300 64 > $out/async_300.json 2> $out/async_300.err; echo "rc=$?" | tee -a $out/log.txt; cat $out/async_300.json | tee -a $out/log.txt
timeout 240 python tools/gpu_async_engine.py 1000 64 > $out/async_1000.json 2> $out/async_1000.err; echo "rc=$?" | tee -a $out/log.txt; cat $out/async_1000.json | tee -a $out/log.txt
echo "== pipelined sweep of the tile-batch engine: A/B on C2 (1M, 7168 plans) and a 2048-plan batch" | tee -a $out/log.txt
timeout 240 python tools/gpu_tb_pipe_ab.py 1000 7168 3 > $out/pipe_c2.json 2> $out/pipe_c2.err; echo "rc=$?" | tee -a $out/log.txt; cat $out/pipe_c2.json | tee -a $out/log.txt
timeout 120 python tools/gpu_tb_pipe_ab.py 1000 2048 3 > $out/pipe_2048.json 2> $out/pipe_2048.err; echo "rc=$?" | tee -a $out/log.txt; cat $out/pipe_2048.json | tee -a $out/log.txt
tail -3 $out/*.err 2>/dev/null | tail -40
