#!/usr/bin/env python
"""gpurun_out/prof_r06/<what> (tools/prof_r06.sh <what>) -> profiles/r06_<tag>_*.  Usage: python tools/summarize_r06.py <what> [tag]

  c4    -> profiles/r06_<tag>_kernel_stats.{csv,md}, profiles/r06_<tag>_pmc.json     (tag default: c4)
  c2sq  -> profiles/r06_<tag>_sq.md                                                  (tag default: c2)
  c2    -> profiles/r06_<tag>_kernel_stats.{csv,md}, profiles/r06_<tag>_pmc_traffic.json
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
what = sys.argv[1] if len(sys.argv) > 1 else "c4"
tag = sys.argv[2] if len(sys.argv) > 2 else {"c4": "c4", "c2sq": "c2", "c2": "bench", "cvpsq": "cvp"}[what]
G = os.path.join(ROOT, "gpurun_out", "prof_r06", what)
P = os.path.join(ROOT, "profiles")
ENGINE = ("k_tb_plan", "k_tb_pairs", "k_tb_scan", "k_tb_items", "k_tb_solve_q", "k_tbv_solve", "k_tb_stats")
CLK_GHZ = 2.4          # MI355X peak engine clock (MI355X_MICROARCH.md); the SQ counters below are ratios, the clock only scales the "busy" lines
N_CU, N_SIMD, N_SE = 256, 1024, 32


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def one(pattern):
    fs = glob.glob(os.path.join(G, pattern), recursive=True)
    if not fs:
        raise SystemExit(f"missing {pattern}")
    return fs[0]


def counters(sub):
    """{kernel: {counter: (sum, launches)}} of one pass directory (the per-dispatch file was reduced to sums on the GPU box)"""
    out = {}
    for f in glob.glob(os.path.join(G, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            out.setdefault(short(r["Kernel_Name"]), {})[r["Counter_Name"]] = (float(r["Counter_Value"]), int(r["Launches"]))
    return out


def sq_table(sq, kernels, wall_ns):
    """derived shader-core figures per kernel.  Units (checked against each other in these files): SQ_BUSY_CYCLES counts per shader
    engine (32 of them: busy / 32 = the kernel's cycles); SQ_WAVE_CYCLES, SQ_ACTIVE_INST_*, SQ_WAIT_INST_ANY count wave-quad-cycles
    (4 clocks: SQ_INSTS_VALU == SQ_ACTIVE_INST_VALU for a kernel of plain 4-cycle VALU instructions)."""
    lines = ["| kernel | launches | cycles (BUSY/32) | waves resident per SIMD | VALU busy of SIMD time | LDS busy of CU time | wave time: issuing / waiting on a counter / waiting for anything | VALU : LDS : SALU : VMEM-rd instructions | LDS bank-conflict share |",
             "|---|---|---|---|---|---|---|---|---|"]
    for k in kernels:
        c = sq.get(k)
        if not c or "SQ_BUSY_CYCLES" not in c:
            continue
        g = lambda n: c.get(n, (0.0, 0))[0]
        cyc = g("SQ_BUSY_CYCLES") / N_SE                               # kernel cycles, summed over its launches
        quads = cyc / 4.0
        res = g("SQ_WAVE_CYCLES") / (quads * N_SIMD) if quads else 0
        valu = g("SQ_ACTIVE_INST_VALU") / (quads * N_SIMD) if quads else 0
        lds = g("SQ_ACTIVE_INST_LDS") / (quads * N_CU) if quads else 0
        wv = g("SQ_WAVE_CYCLES") or 1.0
        lines.append(f"| {k} | {c['SQ_BUSY_CYCLES'][1]} | {cyc:.4g} | {res:.2f} | {100*valu:.1f} % | {100*lds:.1f} % | "
                     f"{100*g('SQ_ACTIVE_INST_ANY')/wv:.0f} % / {100*g('SQ_WAIT_INST_ANY')/wv:.0f} % / {100*g('SQ_WAIT_ANY')/wv:.0f} % | "
                     f"{g('SQ_INSTS_VALU'):.3g} : {g('SQ_INSTS_LDS'):.3g} : {g('SQ_INSTS_SALU'):.3g} : {g('SQ_INSTS_VMEM_RD'):.3g} | "
                     f"{100*g('SQ_LDS_BANK_CONFLICT')/max(g('SQ_ACTIVE_INST_LDS'),1.0):.1f} % |")
    return "\n".join(lines)


def merge_sq():
    sq = {}
    for sub in ("sq_1", "sq_2", "sq_3", "sq_4"):
        for k, v in counters(sub).items():
            sq.setdefault(k, {}).update(v)
    return sq


if what in ("c4", "c2"):
    line = json.loads(open(os.path.join(G, "trace_line.json")).read())
    stats = one("trace/**/*kernel_stats.csv")
    shutil.copy(stats, os.path.join(P, f"r06_{tag}_kernel_stats.csv"))
    rows = list(csv.DictReader(open(stats)))
    fetch, write = counters("pmc_FETCH_SIZE"), counters("pmc_WRITE_SIZE")
    eng = lambda d, c: sum(v[c][0] for k, v in d.items() if k.startswith(ENGINE) and c in v) * 1024.0
    if what == "c4":
        batches = 2                                                    # leg_c4: one warm-up batch + the timed one
        algo = line["roofline"]["algorithmic_bytes"]
        ev_ms = line["roofline"]["avg_launch_us"] / 1e3
        workload = line["workload"]
        cmd = "python tools/gpu_c4_batch.py 4096"
    else:
        batches = 2
        algo = line["roofline"]["algorithmic_bytes_per_step"]
        ev_ms = line["roofline"]["avg_launch_us"] / 1e3
        workload = line["config"]["workload"]
        cmd = "python bench.py --steps 1 --warmup 1 --no-cpu --no-latency --no-configs"
    f_b, w_b = eng(fetch, "FETCH_SIZE") / batches, eng(write, "WRITE_SIZE") / batches
    traffic = {
        "command": f"tools/prof_r06.sh {what}: rocprofv3 --pmc FETCH_SIZE (then, separately, WRITE_SIZE) -- {cmd}",
        "kernel": "k_tbv_solve" if any(short(r["Name"]).startswith("k_tbv_solve") for r in rows) else "k_tb_solve_q", "engine_kernels": list(ENGINE),
        "batch": (line.get("config") or {}).get("batch_per_gpu", line.get("batch")), "grid": 1000 if what == "c2" else 3163,
        "fetch_bytes_per_engine_run_raw": f_b, "write_bytes_per_engine_run_raw": w_b,
        "note": "FETCH_SIZE / WRITE_SIZE (KB) summed over every launch of the engine's kernels, per batch.  gfx950: FETCH_SIZE reports 1/2 of the "
                "bytes of wide (16 B/lane) reads (MI355X_MICROARCH.md, HBM section); the engine's slice loads are 16-byte per-lane loads, its "
                "stream loads too: traffic = raw fetch + write is a lower bound, traffic_high = 2 x fetch + write an upper bound.",
        "traffic_bytes_per_launch": f_b + w_b, "traffic_bytes_per_launch_high": 2 * f_b + w_b, "algorithmic_bytes_per_launch": algo,
        "ratio_traffic_to_algorithmic": (f_b + w_b) / algo, "ratio_high": (2 * f_b + w_b) / algo,
        "hbm_utilisation_of_8TBps": [(f_b + w_b) / (ev_ms * 1e-3) / 8e12, (2 * f_b + w_b) / (ev_ms * 1e-3) / 8e12],
        "per_kernel_fetch_kb": {k: v["FETCH_SIZE"][0] for k, v in sorted(fetch.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", (0,))[0])[:8] if "FETCH_SIZE" in v},
        "per_kernel_write_kb": {k: v["WRITE_SIZE"][0] for k, v in sorted(write.items(), key=lambda kv: -kv[1].get("WRITE_SIZE", (0,))[0])[:8] if "WRITE_SIZE" in v},
    }
    sq = merge_sq() if what == "c4" else {}
    name = f"r06_{tag}_pmc.json" if what == "c4" else f"r06_{tag}_pmc_traffic.json"
    json.dump(traffic, open(os.path.join(P, name), "w"), indent=1)
    trace_batches = (line.get("steps", 1) + line.get("warmup", 1)) if what == "c2" else batches   # the trace pass of c2 runs --steps 3 --warmup 1
    eng_ms = sum(float(r["TotalDurationNs"]) for r in rows if short(r["Name"]).startswith(ENGINE)) / 1e6 / trace_batches
    with open(os.path.join(P, f"r06_{tag}_kernel_stats.md"), "w") as f:
        f.write(f"# profiles/r06_{tag}_kernel_stats.md — rocprofv3 kernel trace, HBM traffic and shader-core counters\n\n")
        f.write(f"MI355X (gfx950). Commands: `tools/prof_r06.sh {what}` = `cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -- {cmd}`, "
                "then one `rocprofv3 --pmc <group>` pass per counter group (FETCH_SIZE; WRITE_SIZE; three SQ_* groups), never combined with a trace domain.\n")
        f.write(f"Raw CSV: `profiles/r06_{tag}_kernel_stats.csv`. Workload: {workload}; {trace_batches} batches in the trace pass, {batches} in each counter pass.\n\n")
        f.write("| kernel | calls | total ms | avg µs | min µs | max µs | % |\n|---|---|---|---|---|---|---|\n")
        for r in rows[:14]:
            f.write(f"| {short(r['Name'])} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.2f} | "
                    f"{float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} | {r['Percentage']} |\n")
        f.write("\nLine printed by the profiled run:\n```json\n" + json.dumps(line) + "\n```\n\n")
        f.write(f"Agreement check: rocprofv3 total of the engine's kernels per batch = **{eng_ms:.1f} ms**; HIP events on the library's stream around the "
                f"engine run in the same process (`roofline.avg_launch_us`) = **{ev_ms:.1f} ms**.\n\n")
        f.write("## HBM traffic (PMC, separate passes)\n\n```json\n" + json.dumps(traffic, indent=1) + "\n```\n")
        if sq:
            f.write("\n## Shader-core counters (three separate passes)\n\n" + sq_table(sq, [k for k in sq if k.startswith(("k_tb_solve_q", "k_tbv_solve", "k_tb_scan", "k_tb_items", "k_tile_round"))], None) + "\n")
    print("engine ms per batch: rocprof", round(eng_ms, 1), "events", round(ev_ms, 1), "| traffic/algorithmic", round(traffic["ratio_traffic_to_algorithmic"], 3), "-", round(traffic["ratio_high"], 3))
elif what == "cvpsq":
    sq = merge_sq()
    ks = [k for k in sq if k.startswith(("k_step_wide", "k_cvp_ctl", "k_step_repair", "k_step<", "k_cvp_verify"))]
    rows = []
    try:
        for r in csv.DictReader(open(one("trace/**/*kernel_stats.csv"))):
            rows.append((short(r["Name"]), int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
    except SystemExit:
        pass
    line = [l for l in open(os.path.join(G, "trace.log")) if l.startswith("{")] if os.path.exists(os.path.join(G, "trace.log")) else []
    with open(os.path.join(P, "r06_pmc_cvp.md"), "w") as f:
        f.write("# profiles/r06_pmc_cvp.md — CVP batch of 128 plans on the C3 configuration: kernel trace and shader-core counters\n\n")
        f.write("MI355X (gfx950). `tools/prof_r06.sh cvpsq`: `rocprofv3 --kernel-trace --stats` and four separate `rocprofv3 --pmc <group>` passes of "
                "`PERF_BATCHES=128 python tools/gpu_cvp_perf.py` (1M-vertex terrain, Steepness + Inflation costs, 128 plans per batch); sums over all launches.\n\n")
        if rows:
            f.write("| kernel | calls | total ms | avg us |\n|---|---|---|---|\n")
            for n, c, t, a in sorted(rows, key=lambda r: -r[2])[:10]:
                f.write(f"| {n} | {c} | {t:.3f} | {a:.2f} |\n")
            f.write("\n")
        if line:
            f.write("Line printed by the traced run:\n```json\n" + line[-1].strip()[:1500] + "\n```\n\n")
        f.write(sq_table(sq, ks, None) + "\n\nRaw sums:\n\n```\n")
        for k in ks:
            for c, (v, n) in sorted(sq[k].items()):
                f.write(f"{k:28s} {c:24s} {v:.6g}  ({n} launches)\n")
        f.write("```\n")
    print(open(os.path.join(P, "r06_pmc_cvp.md")).read()[:3000])
else:
    sq = merge_sq()
    ks = [k for k in sq if k.startswith(("k_tb_solve_q", "k_tbv_solve", "k_dij_finalize", "k_tb_finalize", "k_tb_scan"))]
    with open(os.path.join(P, f"r06_{tag}_sq.md"), "w") as f:
        f.write(f"# profiles/r06_{tag}_sq.md — shader-core counters of the headline bench command\n\n")
        f.write("MI355X (gfx950). `tools/prof_r06.sh c2sq`: three passes of `rocprofv3 --pmc <group> -- python bench.py --steps 1 --warmup 1 --no-cpu --no-latency --no-configs` "
                "(7168 Dijkstra plans on the 1M mesh, two batches), groups `SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS`, "
                "`SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAIT_INST_ANY`, `SQ_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD`; sums over all launches.\n\n")
        f.write(sq_table(sq, ks, None) + "\n\nRaw sums:\n\n```\n")
        for k in ks:
            for c, (v, n) in sorted(sq[k].items()):
                f.write(f"{k:28s} {c:24s} {v:.6g}  ({n} launches)\n")
        f.write("```\n")
    print(open(os.path.join(P, f"r06_{tag}_sq.md")).read()[:3000])
