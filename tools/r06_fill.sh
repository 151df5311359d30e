#!/bin/bash
# where the clean-up of the other distance buffer belongs (MNAV_TB_FILL_MODE: +1 = 48 workgroups, +2 = behind the first chunk of
# iterations): engine ms per batch in both modes of bench.py, then a kernel trace of the current placement reduced to the batch starts
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O; L=tools/_variants/libmnav_fill.so
timeout 900 python tools/gpu_tb_modes.py $L $L@MNAV_TB_FILL_MODE=2 $L@MNAV_TB_FILL_MODE=1 $L@MNAV_TB_FILL_MODE=3 $L@MNAV_TB_NO_PREFILL=1 2>&1 | tee $O/fill_modes.txt
cd /tmp && export TMPDIR=/tmp
for M in 0 2; do
  rm -rf /tmp/ftrace; MNAV_LIB=$GRAFT_REPO_ROOT/$L MNAV_TB_FILL_MODE=$M timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ftrace -o t -- python $GRAFT_REPO_ROOT/tools/gpu_tb_modes.py --inline > /tmp/ftrace.log 2>&1
  tail -1 /tmp/ftrace.log | cut -c1-400
  python - <<PY > $GRAFT_REPO_ROOT/$O/fill_trace_mode$M.txt
import csv, glob
f = glob.glob("/tmp/ftrace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
def nm(r): return r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:28]
# batch starts: every k_tb_seed; print the 14 kernels around it and every kernel above 1 ms
seeds = [i for i, r in enumerate(rows) if nm(r).startswith("k_tb_seed")]
for s in seeds:
    print("---- batch")
    for r in rows[max(s - 8, 0): s + 14]:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{(st - t0) / 1e6:10.3f} ms  +{(en - st) / 1e3:9.1f} us  q{r.get('Queue_Id', '?'):>3}  {nm(r)}")
print("---- kernels above 1 ms")
for r in rows:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if en - st > 1_000_000 and not nm(r).startswith(("k_tbv_solve", "k_tb_solve")):
        print(f"{(st - t0) / 1e6:10.3f} ms  +{(en - st) / 1e3:9.1f} us  q{r.get('Queue_Id', '?'):>3}  {nm(r)}")
PY
done
ls -la $GRAFT_REPO_ROOT/$O/fill_trace_mode*.txt
