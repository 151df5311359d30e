#!/bin/bash
# Round 6, first measurement call: the numbers the round's decisions hang on.
#   gpurun --timeout 1500 -- 'bash tools/r06_probe.sh'
# 1. counters of the tile-batch engine at C2 / C4 (activations, items, sweeps per item, iterations)
# 2. the inflation wave of the C3 cost stack, repeated, under a kernel trace: which launches differ between the 6 ms and the 85 ms mode
# 3. asynchronous engine against the tile rounds on the 10M mesh (auto's choice per mesh size)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
MNAV_TRACE=1 N=1000 B=7168 REPS=1 timeout 300 python tools/gpu_tb_perf.py > $O/tb_c2.json 2> $O/tb_c2.err; grep -h "tile-batch:" $O/tb_c2.err | tail -2; tail -1 $O/tb_c2.json | cut -c1-400
MNAV_TRACE=1 N=3163 B=4096 REPS=1 timeout 600 python tools/gpu_tb_perf.py > $O/tb_c4.json 2> $O/tb_c4.err; grep -h "tile-batch:" $O/tb_c4.err | tail -2; tail -1 $O/tb_c4.json | cut -c1-400
# inflation: plain, then warm (a Dijkstra batch first), then under the profiler
timeout 300 python tools/gpu_infl_bimodal.py 4 0 0 > $O/infl_plain.json 2> $O/infl_plain.err; grep repeat $O/infl_plain.err
timeout 300 python tools/gpu_infl_bimodal.py 3 1 1 > $O/infl_warm.json 2> $O/infl_warm.err; grep repeat $O/infl_warm.err
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/infl_trace -o t -- python $GRAFT_REPO_ROOT/tools/gpu_infl_bimodal.py 4 0 1 > $GRAFT_REPO_ROOT/$O/infl_trace.log 2>&1 )
grep repeat $O/infl_trace.log
python - <<PY
import csv, glob, collections
fs = glob.glob("$O/infl_trace/**/*kernel_trace.csv", recursive=True)
rows = []
for f in fs:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]))
rows.sort()
# split into waves at k_infl_seed; per wave: total, per-kernel sums, the five longest launches with their position
waves, cur = [], None
for s, e, k in rows:
    if k.startswith("k_infl_seed"):
        cur = dict(t0=s, launches=[]); waves.append(cur)
    if cur is not None:
        if k.startswith("k_infl_cost"):
            cur["t1"] = s; cur = None
        else:
            cur["launches"].append((s, e, k))
with open("$O/infl_trace_summary.txt", "w") as g:
    for i, w in enumerate(waves):
        L = w["launches"]
        per = collections.defaultdict(lambda: [0, 0.0])
        for s, e, k in L:
            per[k][0] += 1; per[k][1] += (e - s) / 1e6
        span = (w.get("t1", L[-1][1]) - w["t0"]) / 1e6
        top = sorted(((e - s) / 1e3, j, k) for j, (s, e, k) in enumerate(L))[-6:]
        line = f"wave {i}: span {span:.2f} ms, {len(L)} launches; " + "; ".join(f"{k} x{c} {ms:.2f} ms" for k, (c, ms) in sorted(per.items(), key=lambda x: -x[1][1])[:5]) + " | longest (us, index, kernel): " + ", ".join(f"({a:.0f}, {j}, {k})" for a, j, k in top)
        print(line); g.write(line + "\n")
PY
rm -rf $O/infl_trace
timeout 900 python tools/gpu_async_tune.py 3163 5 tiled,async,async_band2,async_band8,async_wg512 > $O/async_tune_3163.json 2> $O/async_tune_3163.err; tail -1 $O/async_tune_3163.json | cut -c1-1500
