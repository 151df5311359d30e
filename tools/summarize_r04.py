#!/usr/bin/env python
"""gpurun_out/prof_r04 (tools/prof_r04.sh) -> profiles/r04_bench_kernel_stats.{csv,md}, profiles/r04_pmc_traffic.json, profiles/r04_cvp_kernel_stats.{csv,md}, profiles/r04_pmc_cvp.md."""
import csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out", "prof_r04")
P = os.path.join(ROOT, "profiles")
TAG = "r04"
ENGINE = ("k_tb_plan", "k_tb_scan", "k_tb_items", "k_tb_solve_q")


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
    "command": "tools/prof_r04.sh: rocprofv3 --pmc FETCH_SIZE (then, separately, WRITE_SIZE) -- python bench.py --steps 1 --warmup 1 --no-cpu --no-latency --no-configs",
    "kernel": "k_tb_solve_q", "engine_kernels": list(ENGINE), "batch": line["config"]["batch_per_gpu"], "grid": 1000,
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
    f.write("MI355X (gfx950). Command (tools/prof_r04.sh): `cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv "
            "-d gpurun_out/prof_r04/trace -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu --no-latency --no-configs`.\n")
    f.write(f"Raw CSV: `profiles/{TAG}_bench_kernel_stats.csv`. Workload: {line['config']['workload']}; {steps_profiled} batches (1 warm-up + 3 timed). "
            "One batch = ONE run of the tile-batch engine = a few hundred iterations of k_tb_plan / k_tb_scan / k_tb_items / k_tb_solve_q replayed "
            "from a hipGraph, then k_dij_finalize<8, true> (potential, predecessors and vector map of every plan) and k_tb_path for the vertex paths.\n\n")
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


# ---- CVP: the wide step kernel on the benched C3 configuration --------------------------------------------------------------------------
cstats = one("cvp_trace/**/*kernel_stats.csv")
shutil.copy(cstats, os.path.join(P, f"{TAG}_cvp_kernel_stats.csv"))
crow = list(csv.DictReader(open(cstats)))
cline = [l for l in open(os.path.join(G, "cvp_trace.log")) if l.startswith("{")]
cline = json.loads(cline[-1]) if cline else {}
with open(os.path.join(P, f"{TAG}_cvp_kernel_stats.md"), "w") as f:
    f.write(f"# profiles/{TAG}_cvp_kernel_stats.md -- rocprofv3 kernel trace of CVP batches (wide step kernel)\n\n")
    f.write("MI355X (gfx950). Command (tools/prof_r04.sh): `MNAV_NO_GRAPH=1 PERF_BATCHES=128 rocprofv3 --kernel-trace --stats --output-format csv -- python "
            "tools/gpu_cvp_perf.py` (two batches of 128 CVP plans on the 1M-vertex seed-3 terrain, device-built Steepness(0.6) + Inflation costs: the benched C3 "
            "configuration; without hipGraphs, so the host launch gaps are part of the wall time printed below, not of the kernel times).\n\n")
    f.write("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
    for r in crow[:10]:
        f.write(f"| {short(r['Name'])} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.2f} | "
                f"{float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} | {r['Percentage']} |\n")
    f.write("\nLine printed by the profiled run (second batch):\n```json\n" + json.dumps(cline) + "\n```\n")

import collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
nl = collections.defaultdict(set)
for fcsv in glob.glob(os.path.join(G, "cvp_pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(fcsv)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r.get("Launches"): nl[(k, r["Counter_Name"])].update(range(int(r["Launches"])))     # (files reduced on the GPU box: sums + launch counts)
        else: nl[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
wide_ns = sum(float(r["TotalDurationNs"]) for r in crow if short(r["Name"]).startswith("k_step_wide"))
with open(os.path.join(P, f"{TAG}_pmc_cvp.md"), "w") as f:
    f.write(f"# profiles/{TAG}_pmc_cvp.md -- shader-core counters of the CVP batch kernels\n\n")
    f.write("MI355X, `tools/prof_r04.sh`: `MNAV_NO_GRAPH=1 PERF_BATCHES=128 rocprofv3 --pmc <group> --output-format csv -- python tools/gpu_cvp_perf.py`, one counter "
            "group per pass (no trace domains combined with `--pmc`).  Workload as in `profiles/r04_cvp_kernel_stats.md` (two batches of 128 plans).\n\n")
    names = sorted({c for k in agg for c in agg[k]})
    ks = [k for k in agg if k.startswith(("k_step_wide", "k_cvp_ctl", "k_step_repair", "k_cvp_verify", "k_step<"))]
    f.write("| counter | " + " | ".join(f"`{k}` ({max(len(nl[(k, c)]) for c in agg[k])} launches)" for k in ks) + " |\n|---|" + "---|" * len(ks) + "\n")
    for c in names:
        f.write(f"| {c} | " + " | ".join((f"{agg[k][c]:.3g}" if c in agg[k] else "-") for k in ks) + " |\n")
    w = agg.get("k_step_wide", {})
    if w and wide_ns > 0:
        cyc = wide_ns * 1e-9 * 2.4e9                                   # kernel time of the traced run in cycles at 2.4 GHz
        f.write(f"\nReading for `k_step_wide` (sum of its kernel durations in the traced run {wide_ns/1e6:.0f} ms = {cyc:.3g} cycles; the batch is stepped in three groups on "
                "their own streams, whose kernels overlap: the sum is an upper bound of the time the kernel was on the machine, so the two figures below are lower bounds; 1024 SIMDs):\n")
        if "SQ_WAVE_CYCLES" in w:
            f.write(f"* resident waves on average: SQ_WAVE_CYCLES x 4 / cycles = **{w['SQ_WAVE_CYCLES'] * 4 / cyc:.0f}** "
                    f"(= {w['SQ_WAVE_CYCLES'] * 4 / cyc / 1024:.2f} per SIMD; 12 KB of LDS and 168 VGPRs allow 12 per CU = 3 072)\n")
        if "SQ_ACTIVE_INST_VALU" in w:
            f.write(f"* VALU busy: SQ_ACTIVE_INST_VALU x 4 / (cycles x 1024 SIMDs) = **{w['SQ_ACTIVE_INST_VALU'] * 4 / (cyc * 1024) * 100:.0f} %**\n")
        if "SQ_INSTS_VALU" in w and "SQ_INSTS_VMEM_RD" in w and w["SQ_INSTS_VMEM_RD"]:
            f.write(f"* VALU instructions per vector memory read: {w['SQ_INSTS_VALU'] / w['SQ_INSTS_VMEM_RD']:.1f}\n")
        ev = cline.get("evals_per_plan", 0) * cline.get("batch", 0) * 2
        if ev and "SQ_INSTS_VALU" in w:
            f.write(f"* wave-level VALU instructions per vertex evaluation: {w['SQ_INSTS_VALU'] / ev:.1f} (two batches, {ev:.3g} evaluations; "
                    f"the 8-lane kernel of round 2: 1.85e11 / 7.4e8 = 250; the 64-entry version of this kernel: 71 -- with 32 entries per round half the lanes idle in the per-vertex phases)\n")
print("cvp:", {k: dict(v) for k, v in agg.items() if k.startswith("k_step_wide")})
