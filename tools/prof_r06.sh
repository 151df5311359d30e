#!/bin/bash
# Round-6 profiles (run through gpurun).  prof_r06.sh c4   : the C4 leg (10M vertices, 4096 plans on the tile-batch engine):
#                                                            kernel trace, FETCH_SIZE / WRITE_SIZE passes, two SQ_* passes
#                                         prof_r06.sh c2sq : SQ_* passes of the headline bench command (k_tb_solve_q at 1M)
#                                         prof_r06.sh c2   : kernel trace + traffic passes of the headline bench command
#                                         prof_r06.sh cvpsq: kernel trace + SQ_* passes of a batch of 128 CVP plans on the C3 configuration (tools/gpu_cvp_perf.py)
# One counter group per pass, never combined with a trace domain other than --kernel-trace (MI355X_MICROARCH.md).
# Outputs under gpurun_out/prof_r06/<what>/; tools/summarize_r06.py turns them into profiles/r06_*.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
what=${1:-c4}
OUT=$R/gpurun_out/prof_r06/$what
rm -rf $OUT && mkdir -p $OUT
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS"
SQ2="SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAIT_INST_ANY"
SQ3="SQ_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD"
SQ4="SQ_WAIT_ANY SQ_IFETCH SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"
case $what in
  c4)   CMD="python $R/tools/gpu_c4_batch.py ${PROF_C4_BATCH:-4096}";;
  c2sq|c2) CMD="python $R/bench.py --steps 1 --warmup 1 --no-cpu --no-latency --no-configs";;
  cvpsq) export PERF_BATCHES=128; CMD="python $R/tools/gpu_cvp_perf.py";;
esac
if [ $what = cvpsq ]; then
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
fi
if [ $what != c2sq ] && [ $what != cvpsq ]; then
  TR="$CMD"; [ $what = c2 ] && TR="python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-latency --no-configs"
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $TR > $OUT/trace.log 2>&1
  grep '^{' $OUT/trace.log | tail -1 > $OUT/trace_line.json
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- $CMD > $OUT/pmc_$c.log 2>&1
  done
fi
if [ $what != c2 ] && [ -z "$PROF_SKIP_SQ" ]; then
  i=0
  for grp in "$SQ1" "$SQ2" "$SQ3" "$SQ4"; do
    i=$((i+1))
    timeout 900 rocprofv3 --pmc $grp --output-format csv -d $OUT/sq_$i -o pmc -- $CMD > $OUT/sq_$i.log 2>&1
  done
fi
rm -f $OUT/*/*kernel_trace.csv $OUT/*/*/*kernel_trace.csv
# per-dispatch counter files are tens of MB: keep sums per (kernel, counter) and the number of dispatches
python - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(float); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"], r["Counter_Name"])
        agg[k] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    with open(f, "w", newline="") as g:
        w = csv.writer(g); w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value", "Dispatch_Id", "Launches"])
        for (kn, cn), v in sorted(agg.items()):
            w.writerow([kn, cn, repr(v), "sum", len(disp[(kn, cn)])])
PY
du -sh $OUT; tail -2 $OUT/*.log | cut -c1-300
