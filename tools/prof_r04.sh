#!/bin/bash
# Round-4 profiles on the GPU box (run through gpurun): kernel trace + the two HBM traffic passes of the bench command (one counter
# per pass, MI355X_MICROARCH.md HBM section), and kernel trace + shader-core counters of the wide CVP step kernel on the benched C3
# configuration (MNAV_NO_GRAPH=1: rocprofv3 does not attribute the kernels of hipGraph replays).  Outputs under gpurun_out/prof_r04/;
# tools/summarize_r04.py turns them into profiles/r04_*.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r04
rm -rf $OUT && mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --no-cpu --no-latency --no-configs"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py $ARGS > $OUT/trace.log 2>&1
grep '^{' $OUT/trace.log | tail -1 > $OUT/trace_line.json
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu --no-latency --no-configs > $OUT/pmc_$c.log 2>&1
done
MNAV_NO_GRAPH=1 PERF_BATCHES=128 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cvp_trace -o cvp -- python $R/tools/gpu_cvp_perf.py > $OUT/cvp_trace.log 2>&1
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  MNAV_NO_GRAPH=1 PERF_BATCHES=128 timeout 600 rocprofv3 --pmc $grp --output-format csv -d $OUT/cvp_pmc_$i -o pmc -- python $R/tools/gpu_cvp_perf.py > $OUT/cvp_pmc_$i.log 2>&1
done
rm -f $OUT/trace/*kernel_trace.csv $OUT/cvp_trace/*kernel_trace.csv $OUT/*/*/*kernel_trace.csv       # tens of MB; the stats files are the summary
# the per-dispatch counter files of the CVP passes are tens of MB (three kernels per step): keep the sums per (kernel, counter)
python - <<'PY'
import csv, glob, os, collections
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_r04"
for f in glob.glob(root + "/*pmc*/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(float); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"], r["Counter_Name"])
        agg[k] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    with open(f, "w", newline="") as g:
        w = csv.writer(g); w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value", "Dispatch_Id", "Launches"])
        for (kn, cn), v in sorted(agg.items()):
            w.writerow([kn, cn, repr(v), "sum", len(disp[(kn, cn)])])
PY
find $OUT -name "*.csv" | head -20
du -sh $OUT
