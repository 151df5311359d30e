#!/bin/bash
# the slow launch at the start of a batch: low-priority stream for the other buffer's clean-up (A/B), then a kernel trace of the
# in-tree build reduced to the batch starts (which kernel, next to which part of the fill)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
timeout 300 python tools/gpu_tb_modes.py mesh_navigation_amd/libmnav.so tools/_variants/libmnav_prio.so 2>&1 | tee $O/fill_prio.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ftrace; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ftrace -o t -- python $GRAFT_REPO_ROOT/tools/gpu_tb_modes.py --inline > /tmp/ftrace.log 2>&1
tail -1 /tmp/ftrace.log | cut -c1-400
python - <<PY > $GRAFT_REPO_ROOT/$O/fill_trace.txt
import csv, glob
f = glob.glob("/tmp/ftrace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
def nm(r): return r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:28]
seeds = [i for i, r in enumerate(rows) if nm(r).startswith("k_tb_seed")]
for s in seeds:
    print("---- batch")
    for r in rows[max(s - 8, 0): s + 16]:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{(st - t0) / 1e6:10.3f} ms  +{(en - st) / 1e3:9.1f} us  q{r.get('Queue_Id', '?'):>3}  {nm(r)}")
print("---- kernels above 1 ms (solve launches left out)")
for r in rows:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if en - st > 1_000_000 and not nm(r).startswith(("k_tbv_solve", "k_tb_solve")):
        print(f"{(st - t0) / 1e6:10.3f} ms  +{(en - st) / 1e3:9.1f} us  q{r.get('Queue_Id', '?'):>3}  {nm(r)}")
PY
wc -l $GRAFT_REPO_ROOT/$O/fill_trace.txt
