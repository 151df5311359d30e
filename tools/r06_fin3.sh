#!/bin/bash
# more shader-core counters for k_tb_finalize (what do its waves wait for?)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06/fin_sq; rm -rf $OUT; mkdir -p $OUT
CMD="python $R/bench.py --steps 1 --warmup 1 --no-cpu --no-latency --no-configs"
i=0
for grp in "SQ_WAIT_ANY SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INST_CYCLES_SALU" "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_WAIT_INST_LDS" "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"; do
  i=$((i+1)); rm -rf /tmp/p$i
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d /tmp/p$i -o pmc -- $CMD > /tmp/p$i.log 2>&1
  python - <<PY
import csv, glob, collections
agg = collections.defaultdict(float)
for f in glob.glob("/tmp/p$i/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        for key in ("finalize", "tbv_solve"):
            if key in kn: agg[(key, r["Counter_Name"])] += float(r["Counter_Value"])
for k, v in sorted(agg.items()): print(k[0], k[1], "%.4g" % v)
PY
done | tee $OUT/counters.txt
