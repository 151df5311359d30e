#!/usr/bin/env python
"""Turn the raw rocprofv3 output of tools/prof_bench.sh / tools/prof_pmc.sh (under gpurun_out/) into the
summaries committed under profiles/ (round tag as first argument, default r01)."""
import csv, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


def last_json_line(path):
    lines = [l for l in open(path) if l.startswith("{")]
    return json.loads(lines[-1]) if lines else None


def pmc_per_launch(counter, kernel):
    path = os.path.join(G, f"pmc_{counter}", "pmc_counter_collection.csv")
    per = {}
    for r in csv.DictReader(open(path)):
        if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
            per.setdefault(r["Dispatch_Id"], 0.0)
            per[r["Dispatch_Id"]] += float(r["Counter_Value"])
    vals = list(per.values())
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


rows = list(csv.DictReader(open(os.path.join(G, "prof_bench", "bench_kernel_stats.csv"))))
shutil.copy(os.path.join(G, "prof_bench", "bench_kernel_stats.csv"), os.path.join(P, f"{tag}_bench_kernel_stats.csv"))
prof_line = last_json_line(os.path.join(G, "prof_bench_line.json"))
bench_path = os.path.join(G, f"bench_{tag}.json")      # the un-profiled default run (made AFTER the PMC passes: it quotes their traffic)
bench_line = last_json_line(bench_path) if os.path.exists(bench_path) else None
if bench_line:
    json.dump(bench_line, open(os.path.join(P, f"{tag}_bench_line.json"), "w"))

kernel = prof_line["roofline"]["kernel"]
fetch, nl = pmc_per_launch("FETCH_SIZE", kernel)
write, _ = pmc_per_launch("WRITE_SIZE", kernel)
pmc_line = last_json_line(os.path.join(G, "pmc_FETCH_SIZE.log"))
traffic = {
    "command": "tools/prof_pmc.sh: rocprofv3 --pmc FETCH_SIZE (then, separately, WRITE_SIZE) --kernel-trace -- python bench.py --steps 1 --warmup 1 --no-cpu --no-latency --no-configs",
    "kernel": kernel, "batch": pmc_line["config"]["batch_per_gpu"], "launches": nl,
    "fetch_kb_per_launch_raw": fetch, "write_kb_per_launch_raw": write,
    "note": "FETCH_SIZE/WRITE_SIZE are in KB. gfx950: FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) coalesced reads "
            "(MI355X_MICROARCH.md, HBM); the tile staging copies (the bulk of the reads) are such reads, the distance gathers are "
            "4-byte loads. traffic_low = raw, traffic_high = 2 x fetch + write.",
    "traffic_bytes_per_launch_low": (fetch + write) * 1024.0,
    "traffic_bytes_per_launch_high": (2 * fetch + write) * 1024.0,
    "algorithmic_bytes_per_launch": pmc_line["roofline"]["algorithmic_bytes_per_step"] / max(pmc_line["roofline"]["launches_per_step"], 1),
}
json.dump(traffic, open(os.path.join(P, f"{tag}_pmc_traffic.json"), "w"), indent=1)

with open(os.path.join(P, f"{tag}_bench_kernel_stats.md"), "w") as f:
    w = prof_line["config"]
    f.write(f"# profiles/{tag}_bench_kernel_stats.md — rocprofv3 kernel trace of the bench command\n\n")
    f.write("MI355X (gfx950). Command (tools/prof_bench.sh): `cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats "
            "--output-format csv -d gpurun_out/prof_bench -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu --no-latency --no-configs`.\n")
    f.write(f"Raw CSV: `profiles/{tag}_bench_kernel_stats.csv`. Workload: {w['workload']}; 4 batches (1 warm-up + 3 timed), one launch of "
            f"`{kernel}` (one workgroup per plan) per batch. Summary written by tools/summarize_profiles.py.\n\n")
    f.write("| kernel | calls | total ms | avg µs | min µs | max µs | % |\n|---|---|---|---|---|---|---|\n")
    avg_dom = None
    for r in rows[:12]:
        n = short(r["Name"])
        if kernel in n and avg_dom is None:
            avg_dom = float(r["AverageNs"]) / 1e3
        f.write(f"| {n} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.2f} | "
                f"{float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} | {r['Percentage']} |\n")
    f.write("\nBench line printed by the profiled run:\n```json\n" + json.dumps(prof_line) + "\n```\n\n")
    f.write(f"Agreement check: rocprofv3 average duration of `{kernel}` = **{avg_dom:.1f} µs**; `roofline.avg_launch_us` measured live in "
            f"the same run with HIP events on the library's stream around the launch = **{prof_line['roofline']['avg_launch_us']:.1f} µs**.\n\n")
    if bench_line:
        f.write(f"Un-profiled bench line of the same build (`python bench.py`, all legs; also `profiles/{tag}_bench_line.json`):\n"
                "```json\n" + json.dumps(bench_line) + "\n```\n\n")
    f.write("## HBM traffic (PMC, separate passes)\n\n```json\n" + json.dumps(traffic, indent=1) + "\n```\n")
print("dominant kernel", kernel, "avg us", avg_dom, "live", prof_line["roofline"]["avg_launch_us"])
print(json.dumps(traffic, indent=1))
