#!/usr/bin/env python
"""gpurun_out/prof_r03 (tools/prof_r03.sh) -> profiles/r03_bench_kernel_stats.{csv,md}, profiles/r03_pmc_traffic.json."""
import csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out", "prof_r03")
P = os.path.join(ROOT, "profiles")
TAG = "r03"
ENGINE = ("k_tb_plan", "k_tb_scan", "k_tb_items", "k_tb_solve")


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def one(pattern):
    fs = glob.glob(os.path.join(G, pattern), recursive=True)
    if not fs:
        raise SystemExit(f"missing {pattern}")
    return fs[0]


line = json.loads(open(os.path.join(G, "trace_line.json")).read())
stats = one("trace/**/*kernel_stats.csv")
shutil.copy(stats, os.path.join(P, f"{TAG}_bench_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))
steps_profiled = line["steps"] + line["warmup"]

pmc = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = one(f"pmc_{c}/**/*counter_collection.csv")
    tot_engine = tot_all = 0.0
    per_kernel = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c:
            continue
        k = short(r["Kernel_Name"])
        v = float(r["Counter_Value"])
        per_kernel[k] = per_kernel.get(k, 0.0) + v
        tot_all += v
        if any(k.startswith(e) for e in ENGINE):
            tot_engine += v
    pmc[c] = dict(engine_kb=tot_engine, all_kb=tot_all, per_kernel_kb={k: v for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1])[:8]})
batches = 2                                                # the PMC passes run --steps 1 --warmup 1
fetch = pmc["FETCH_SIZE"]["engine_kb"] / batches * 1024.0
write = pmc["WRITE_SIZE"]["engine_kb"] / batches * 1024.0
algo = line["roofline"]["algorithmic_bytes_per_step"]
traffic = {
    "command": "tools/prof_r03.sh: rocprofv3 --pmc FETCH_SIZE (then, separately, WRITE_SIZE) -- python bench.py --steps 1 --warmup 1 --no-cpu --no-latency --no-configs",
    "kernel": "k_tb_solve", "engine_kernels": list(ENGINE), "batch": line["config"]["batch_per_gpu"], "grid": 1000,
    "fetch_bytes_per_engine_run_raw": fetch, "write_bytes_per_engine_run_raw": write,
    "note": "FETCH_SIZE / WRITE_SIZE (KB) summed over every launch of the engine's four kernels of one batch (one engine run = a few hundred "
            "iterations).  gfx950: FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) reads (MI355X_MICROARCH.md, HBM); the engine's slice "
            "loads are 16-byte per-lane loads of DIFFERENT lines per lane and its stream loads 4-byte coalesced ones, so the correction "
            "is an upper bound here: traffic = raw fetch + write, traffic_high = 2 x fetch + write.",
    "traffic_bytes_per_launch": fetch + write, "traffic_bytes_per_launch_high": 2 * fetch + write,
    "algorithmic_bytes_per_launch": algo,
    "ratio_traffic_to_algorithmic": (fetch + write) / algo, "ratio_high": (2 * fetch + write) / algo,
    "per_kernel_fetch_kb_both_batches": pmc["FETCH_SIZE"]["per_kernel_kb"], "per_kernel_write_kb_both_batches": pmc["WRITE_SIZE"]["per_kernel_kb"],
}
json.dump(traffic, open(os.path.join(P, f"{TAG}_pmc_traffic.json"), "w"), indent=1)

engine_ms = sum(float(r["TotalDurationNs"]) for r in rows if any(short(r["Name"]).startswith(e) for e in ENGINE)) / 1e6 / steps_profiled
with open(os.path.join(P, f"{TAG}_bench_kernel_stats.md"), "w") as f:
    f.write(f"# profiles/{TAG}_bench_kernel_stats.md — rocprofv3 kernel trace of the bench command\n\n")
    f.write("MI355X (gfx950). Command (tools/prof_r03.sh): `cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv "
            "-d gpurun_out/prof_r03/trace -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu --no-latency --no-configs`.\n")
    f.write(f"Raw CSV: `profiles/{TAG}_bench_kernel_stats.csv`. Workload: {line['config']['workload']}; {steps_profiled} batches (1 warm-up + 3 timed). "
            "One batch = ONE run of the tile-batch engine = a few hundred iterations of k_tb_plan / k_tb_scan / k_tb_items / k_tb_solve replayed "
            "from a hipGraph, then k_tb_unblock + k_dij_finalize + k_vecmap_dijkstra + k_finish for the V-sized outputs.\n\n")
    f.write("| kernel | calls | total ms | avg µs | min µs | max µs | % |\n|---|---|---|---|---|---|---|\n")
    for r in rows[:14]:
        f.write(f"| {short(r['Name'])} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.2f} | "
                f"{float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} | {r['Percentage']} |\n")
    f.write("\nBench line printed by the profiled run:\n```json\n" + json.dumps(line) + "\n```\n\n")
    f.write(f"Agreement check: rocprofv3 total of the engine's four kernels per batch = **{engine_ms:.1f} ms**; `roofline.avg_launch_us` (HIP events on "
            f"the library's stream around the engine run, same process) = **{line['roofline']['avg_launch_us']/1e3:.1f} ms**.\n\n")
    f.write("## HBM traffic (PMC, separate passes)\n\n```json\n" + json.dumps(traffic, indent=1) + "\n```\n")
print("engine ms per batch (rocprof)", engine_ms, "live", line["roofline"]["avg_launch_us"] / 1e3)
print(json.dumps({k: traffic[k] for k in ("traffic_bytes_per_launch", "traffic_bytes_per_launch_high", "algorithmic_bytes_per_launch", "ratio_traffic_to_algorithmic", "ratio_high")}, indent=1))
