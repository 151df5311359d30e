#!/bin/bash
# Round-3 profiles of the bench command on the GPU box (run through gpurun): kernel trace + the two HBM traffic passes
# (one counter per pass, MI355X_MICROARCH.md HBM section).  Outputs under gpurun_out/prof_r03/; tools/summarize_r03.py
# turns them into profiles/r03_*.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r03
rm -rf $OUT && mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --no-cpu --no-latency --no-configs"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py $ARGS > $OUT/trace.log 2>&1
grep '^{' $OUT/trace.log | tail -1 > $OUT/trace_line.json
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu --no-latency --no-configs > $OUT/pmc_$c.log 2>&1
done
rm -f $OUT/trace/*kernel_trace.csv                    # tens of MB; the stats file is the summary
find $OUT -name "*.csv" | head -20
du -sh $OUT
