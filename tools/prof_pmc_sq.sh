#!/bin/bash
# shader-core counters of the bench command (what bounds k_plan_persistent), one group per pass
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "LdsUtil LdsBankConflict" "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_sq_$i
  rm -rf $OUT
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu --no-latency > $GRAFT_REPO_ROOT/gpurun_out/pmc_sq_$i.log 2>&1
done
python - <<'PY'
import csv, glob, os, collections
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/pmc_sq_*/pmc_counter_collection.csv"):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        per[(k, r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (k, c, d), v in per.items():
        agg[k][c].append(v)
for k in agg:
    if "persistent" in k or "finalize" in k:
        print(k)
        for c, vals in sorted(agg[k].items()):
            print(f"   {c:28s} mean per launch {sum(vals)/len(vals):.6g}  (n={len(vals)})")
PY
