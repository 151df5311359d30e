#!/bin/bash
# shader-core counters of the CVP step kernel (batch of 128 plans on the device-built C3 costs), one group per pass.
# MNAV_NO_GRAPH=1: rocprofv3 cannot follow the hipGraph replays of the steps.
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_cvp_$i
  rm -rf $OUT
  MNAV_NO_GRAPH=1 PERF_BATCHES=128 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT -o pmc -- python $GRAFT_REPO_ROOT/tools/gpu_cvp_perf.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_cvp_$i.log 2>&1
done
python - <<'PY'
import csv, glob, os, collections
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(set)
for f in glob.glob(root + "/pmc_cvp_*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if int(r.get("Grid_Size_Y", r.get("Grid_Size", 0)) or 0) < 128 and "k_step" in k: continue   # the batch launches only
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for k in agg:
    if "k_step" in k or "verify" in k:
        print(k)
        for c, v in sorted(agg[k].items()): print(f"   {c:24s} total {v:.6g} over {len(n[(k, c)])} launches")
PY
