#!/usr/bin/env python
"""bench.py -- plans/sec of the MI355X wavefront planner on BASELINE config C2.

One "step" = one batch of B (default 7168: 234 GB of the 288 GB with the V-sized outputs of every plan resident) independent Dijkstra (delta-stepping SSSP) plans on the 1M-vertex synthetic
terrain (N=1000, h=0.1 m, seed 2, edge_cost_factor 0, reference default cut-offs goal_dist_offset 0.3 /
cost_limit 1.0), B goal vertices drawn per step, common robot vertex (the concurrent-goals shape of BASELINE config 5).
Every plan does what the reference's dijkstra() does (dijkstra_mesh_planner.cpp:217-398): the wave, the cut-off
semantics + predecessors (finalize pass), computeVectorMap (:189-209, :380) and the vertex path; potential,
predecessors and vector map stay resident in HBM (mnav_set_resident_outputs), the vertex-index paths come back to
the host.  Mesh and costs are resident before the timed region.  For N > 1 every rank (one per GPU) runs its own
batches on its own replica of the mesh: the path shards by plan, there is no data-path collective (weak scaling).

Prints ONE JSON line (see the task contract).  `value` = plans/s of the whole job (C2).  The settled-vertex count that
feeds the algorithmic-bytes figure is instrumentation: it is read after the timed region (the batches are replayed
untimed for the roofline object).  Rank 0 then measures, outside the timed region, the other BASELINE configurations on
the same GPU and reports them under "configs", each with its own roofline and cpu_baseline objects:
  C2_paths_only  the same batches without finalize / vector map (vertex paths only), and with a larger batch
  C5  64 concurrent goals on the C2 mesh (one batch)
  C3  CVP wavefront on the 1M mesh (seed 3) with Steepness + Inflation costs, the whole cost stack built on the device
  C4  Dijkstra on the 10M-vertex mesh (N=3163, seed 4): single plan and a batch (skip with --skip-c4)
plus the adapter-inclusive ms/makePlan (MeshPlanner surface, 1M).
`--config C4 --gpus N` (under torch.distributed.run) times ONE plan range-partitioned over N GPUs instead (strong scaling,
mesh_navigation_amd/sharded.py: RCCL min-allreduce of the interface distances).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# The GPU boxes show 256 CPUs and give the container 16 of them (cgroup cpu.max = 1600000 100000): thread pools sized by the CPU
# count (OpenBLAS / OpenMP behind numpy, scipy, torch) burn the quota within a period and the whole process -- the thread that
# waits for the GPU included -- is frozen for the rest of it, up to 85 ms.  That is the "85 ms inflation wave" of the driver's
# lines of rounds 3-5 (profiles/r06_infl_slow_mode.md: the device's own clock says 4.8 ms, the host waits 80; cpu.stat counts
# two to four throttled periods per run).  Pools of 8 threads stay inside the quota.
_POOL = "8" if int(os.environ.get("WORLD_SIZE", "1")) <= 1 else "2"     # (one process per GPU: the ranks share the quota)
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
    os.environ.setdefault(_v, _POOL)

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 GB/s measured copy


def vertex_normals(mesh) -> np.ndarray:
    """normalised sum of the unit face normals (only the vector maps and the Steepness layer read them)"""
    p = mesh.xyz.astype(np.float64)
    fnrm = np.cross(p[mesh.faces[:, 1]] - p[mesh.faces[:, 0]], p[mesh.faces[:, 2]] - p[mesh.faces[:, 0]])
    fnrm /= np.maximum(np.linalg.norm(fnrm, axis=1, keepdims=True), 1e-30)
    vnrm = np.zeros_like(p)
    for k in range(3):
        for c in range(3):
            vnrm[:, c] += np.bincount(mesh.faces[:, k], weights=fnrm[:, c], minlength=mesh.V)
    vnrm /= np.maximum(np.linalg.norm(vnrm, axis=1, keepdims=True), 1e-30)
    return vnrm.astype(np.float32), fnrm.astype(np.float32)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("MNAV_BENCH_BATCH", "7168")))
    ap.add_argument("--grid", type=int, default=int(os.environ.get("MNAV_BENCH_N", "1000")))
    ap.add_argument("--offset", type=float, default=float(os.environ.get("MNAV_BENCH_OFFSET", "0.3")),
                    help="goal_dist_offset (reference default 0.3; inf = full-field variant of SURVEY.md 8d)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-all-cores-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-latency", action="store_true", help="skip the single-plan latency runs (profiling)")
    ap.add_argument("--config", default="C2", choices=["C2", "C4"], help="C2: the headline batch bench; C4: one sharded plan on the 10M mesh")
    ap.add_argument("--no-configs", action="store_true", help="skip the C3/C4/C5 legs (profiling the headline kernel)")
    ap.add_argument("--skip-c4", action="store_true", help="skip the 10M-vertex leg")
    ap.add_argument("--c4-grid", type=int, default=int(os.environ.get("MNAV_BENCH_C4_N", "3163")))
    ap.add_argument("--c4-batch", type=int, default=int(os.environ.get("MNAV_BENCH_C4_BATCH", "4096")))   # tile-batch engine: 52 MB of blocked distances per plan at 10M = 213 GB (1536 plans on the per-plan engine: 831 plans/s; 4096 here: 1132)
    args = ap.parse_args()
    if args.cpu_all_cores_child:
        return cpu_all_cores_child(args)

    import torch
    torch.set_num_threads(int(_POOL))                                   # (the container's CPU quota: see the top of this file)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    from mesh_navigation_amd import capi, meshgen

    if args.config == "C4":
        return sharded_c4(args, torch, dist, rank, local_rank, world)

    N, B = args.grid, args.batch
    mesh = meshgen.terrain(N, 0.1, 2)
    edge_w = meshgen.edge_lengths(mesh)                  # edge_cost_factor 0 -> weights == edge distances
    costs = np.zeros(mesh.V, np.float32)
    ctx = capi.MnavContext(local_rank)
    vnrm, fnrm = vertex_normals(mesh)
    ctx.upload_mesh(mesh.xyz, mesh.faces, mesh.edges, vnrm.astype(np.float32))
    ctx.upload_costs(costs, edge_w)
    robot = mesh.vertex_at(0.9, 0.9)
    rng = np.random.default_rng(5 + 1000 * rank)

    # the goal sets of all batches are drawn BEFORE the timed region (inputs, like the mesh and the costs)
    n_batches = max(args.warmup, 0) + args.steps + 4
    goal_sets = [rng.choice(mesh.V, size=B, replace=False).astype(np.uint32) for _ in range(n_batches)]
    robots = np.full(B, robot, np.uint32)
    next_set = [0]

    def batch_goals():
        g = goal_sets[next_set[0] % n_batches]
        next_set[0] += 1
        return g, robots

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def first_of(g, t, r):
        # what the legs outside the timed region need of the first batch: its goals and the codes / paths of its first plans
        k = min(B, 512)
        return g, t, {"codes": r["codes"][:k].copy(), "paths": [r["paths"][i] for i in range(k)]}

    # every plan leaves potential, predecessors and vector map behind, like the reference's dijkstra() (:380): resident in HBM
    ctx.set_resident_outputs(True)
    first = None
    for _ in range(max(args.warmup, 0)):
        g, t = batch_goals()
        r = ctx.plan_dijkstra_batch(g, t, goal_dist_offset=args.offset, want_fields=False, path_cap=16384)
        if first is None:
            first = first_of(g, t, r)
    r = None                                                            # (a result keeps its path buffer of the binding's pool busy)
    prop_ms = kern_ms = launches = algo = vec_ms = 0.0
    settled = 0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        g, t = batch_goals()
        r = ctx.plan_dijkstra_batch(g, t, goal_dist_offset=args.offset, want_fields=False, path_cap=16384)
        assert (r["codes"] == 0).all(), r["codes"]
        st = r["stats"]                                                 # event timings + the finalize pass's settled count (already on the host)
        prop_ms += st["ms_propagation"]; kern_ms += st["ms_step_kernels"]; launches += st["launches"]; vec_ms += st["ms_vector_map"]
        algo += st["algorithmic_bytes"]; settled += st["settled"]
        if first is None:
            first = first_of(g, t, r)
    barrier()
    elapsed = time.perf_counter() - t0
    headline_kernel = ctx.last_engine()                                 # which engine / kernel `auto` ran the batches on
    from mesh_navigation_amd import multi
    total_plans, elapsed = multi.aggregate_throughput(B * args.steps, elapsed, dist)   # sum of plans, MAX time over ranks
    ctx.set_resident_outputs(False)

    # single-plan latency (ms/makePlan, device part) -- rank 0 only, outside the timed region
    single_ms = single_p95 = None
    if rank == 0 and not args.no_latency:
        # SURVEY.md §8d protocol: 3 warm-ups, 20 timed plans (different goals), median and p95
        lat = []
        for k in range(23):
            o = ctx.plan_dijkstra(int(first[0][k % B]), robot, goal_dist_offset=args.offset, want_fields=False)
            if k >= 3:
                lat.append(o.stats["ms_total"])
        single_ms = float(np.median(lat))
        single_p95 = float(np.percentile(lat, 95))

    # the second planner on the same mesh (CVP / FMM wavefront, cvp_mesh_planner.cpp:651-918): single-plan
    # latency and a batch of 128 -- reported next to the headline metric, not part of `value`
    cvp = None
    if rank == 0 and not args.no_latency:
        first_face = np.full(mesh.V, -1, np.int64)
        fl = mesh.faces.ravel()
        first_face[fl[::-1]] = (np.arange(fl.size)[::-1] // 3)           # some face of every vertex
        def wave_seed(v):
            f = int(first_face[v])
            return mesh.xyz[mesh.faces[f]].astype(np.float64).mean(axis=0).astype(np.float32), f
        tf = int(first_face[robot])
        lat = []
        for k in range(13):
            sp, sf = wave_seed(int(first[0][k % B]))
            o = ctx.plan_cvp(sp, sf, tf, want_fields=False, want_vecmap=False)
            assert o.code == 0, o.code
            if k >= 3:
                lat.append(o.stats["ms_total"])
        nb = 128
        seeds = [wave_seed(int(v)) for v in first[0][:nb]]
        sps = np.stack([x[0] for x in seeds]); sfs = np.array([x[1] for x in seeds], np.uint32)
        ctx.plan_cvp_batch(sps, sfs, np.full(nb, tf, np.uint32))
        tb = time.perf_counter()
        rb = ctx.plan_cvp_batch(sps, sfs, np.full(nb, tf, np.uint32))
        tb = time.perf_counter() - tb
        cvp = {"ms_per_makeplan_single": float(np.median(lat)), "ms_per_makeplan_single_p95": float(np.percentile(lat, 95)),
               "batch": nb, "plans_per_s_batch": nb / tb, "codes_ok": bool((rb["codes"] == 0).all())}

    out = None
    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        # roofline of the dominant kernel: algorithmic bytes per launch (24 B per settled vertex + 24 B per
        # incident edge, summed over the batch) divided by the average launch duration, measured live with HIP events that
        # the library records on ITS OWN stream around the engine's launches.  Batches of >= 256 plans run on the tile-batch
        # engine: ONE engine run per batch = a few hundred iterations of k_tb_plan / k_tb_scan / k_tb_items / k_tb_solve
        # replayed from a hipGraph (the solve kernel is > 80 % of it, profiles/r06_bench_kernel_stats.md); the events bracket the
        # whole run, so `achieved` prices the scheduling kernels too.  HBM traffic per run from the PMC passes committed
        # under profiles/ (tools/prof_pmc.sh: FETCH_SIZE / WRITE_SIZE summed over the engine's kernels of one batch) -- only
        # quoted when it was measured on this very workload.
        # `traffic` is null in this line: the PMC counters cannot be read inside a timed run (rocprofv3 collects them in its own
        # passes).  The figure of those passes over this very command is quoted next to it, with its source, when the
        # committed profile was taken on the same workload and kernel.
        traffic = None
        traffic_profiled = None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_pmc_traffic.json")))
            if pm.get("kernel", "") in headline_kernel and B == pm.get("batch") and N == pm.get("grid"):
                traffic_profiled = {"bytes_per_launch": pm["traffic_bytes_per_launch"], "bytes_per_launch_high": pm.get("traffic_bytes_per_launch_high"),
                                    "source": "profiles/r06_bench_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, not this run)"}
        except (OSError, ValueError, KeyError):
            pass
        per_launch_bytes = algo / max(launches, 1)
        per_launch_s = kern_ms * 1e-3 / max(launches, 1)
        achieved = per_launch_bytes / per_launch_s / 1e9 if per_launch_s > 0 else 0.0
        out = {
            "metric": "plans/sec (Dijkstra makePlan device path incl. computeVectorMap, 1M-vertex mesh)",
            "value": total_plans / elapsed,
            "unit": "plans/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"C2: delta-stepping SSSP (tile-batch label-correcting) + finalize + computeVectorMap + vertex path, "
                                   f"{N}x{N} terrain = {mesh.V} vertices, uniform edge costs, batch of {B} goals per step per GPU, "
                                   f"common robot vertex, goal_dist_offset {args.offset:g}; potential / predecessors / vector map resident in HBM",
                       "vertices": mesh.V, "edges": mesh.E, "batch_per_gpu": B,
                       "parallelism": f"{world} independent replicas (plans sharded by rank)"},
            "ms_per_makeplan_single": single_ms,
            "ms_per_makeplan_single_p95": single_p95,
            "ms_per_plan_in_batch": ms_step / B,
            "cvp_planner_same_mesh": cvp,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS,
                         # HBM bytes per launch: the counters cannot be read inside a timed run; what is quoted is the figure of the
                         # committed rocprofv3 --pmc passes over this very command, under a key that says so
                         "traffic": ({"profiled": traffic_profiled["bytes_per_launch"], "profiled_high": traffic_profiled["bytes_per_launch_high"],
                                      "source": traffic_profiled["source"]} if traffic_profiled else "not profiled (no PMC passes of this workload and kernel under profiles/)"),
                         "kernel": headline_kernel + " -- the whole engine run: plan / pairs / scan / items / solve iterations",
                         "launches_per_step": launches / args.steps, "vector_map_ms_per_step": vec_ms / args.steps,
                         "algorithmic_bytes_per_step": algo / args.steps,
                         "avg_launch_us": per_launch_s * 1e6, "propagation_ms_per_step": prop_ms / args.steps,
                         "settled_vertices_per_plan": settled / max(args.steps * B, 1)},
        }
        if not args.no_cpu and world >= 1:
            out["cpu_baseline"] = cpu_baseline(mesh, edge_w, costs, first, B, args.offset)
        if not args.no_configs:
            cfgs = {}
            t_cfg = time.perf_counter()
            for name, leg in (("C2_paths_only", lambda: leg_paths_only(ctx, mesh, robot, rng, args)),
                              ("C5", lambda: leg_c5(ctx, mesh, edge_w, costs, robot, args))):
                cfgs[name] = run_leg(leg)
            ctx.close(); ctx = None                                   # free the plan slots before the other meshes
            cfgs["C2_adapter_inclusive"] = run_leg(lambda: leg_adapter(mesh, vnrm, fnrm, edge_w, costs, first, robot))
            cfgs["C3"] = run_leg(lambda: leg_c3(local_rank, args))
            if not args.skip_c4:
                cfgs["C4"] = run_leg(lambda: leg_c4(local_rank, args))
            cfgs["wall_s"] = time.perf_counter() - t_cfg
            out["configs"] = cfgs
        print(json.dumps(out), flush=True)
    if ctx is not None:
        ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()



def run_leg(fn):
    """A failing side leg must not take the headline line with it: its error is reported in its place."""
    t = time.perf_counter()
    try:
        r = fn()
    except Exception as e:                                              # noqa: BLE001 -- reported, not swallowed
        r = {"error": repr(e)[:400]}
    r["leg_wall_s"] = round(time.perf_counter() - t, 2)
    return r


def roofline_of(stats, kernel):
    """achieved = algorithmic bytes per launch / average launch duration (HIP events on the library's stream)"""
    launches = max(int(stats["launches"]), 1)
    per_s = stats["ms_step_kernels"] * 1e-3 / launches
    ach = (stats["algorithmic_bytes"] / launches) / per_s / 1e9 if per_s > 0 else 0.0
    return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": "not profiled",
            "kernel": kernel, "launches": launches, "avg_launch_us": per_s * 1e6,
            "algorithmic_bytes": int(stats["algorithmic_bytes"]), "propagation_ms": stats["ms_propagation"]}


def host_cpu():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return ""


def leg_paths_only(ctx, mesh, robot, rng, args):
    """The headline batches when the caller only wants the vertex paths (getPath): no finalize pass, no vector map,
    predecessors derived along the path (k_tb_path); also with a batch twice as large (better filled waves)."""
    out = {}
    for B in (args.batch, 2 * args.batch):
        robots = np.full(B, robot, np.uint32)
        ts, st = [], None
        for k in range(4):
            g = rng.choice(mesh.V, size=B, replace=False).astype(np.uint32)
            t0 = time.perf_counter()
            r = ctx.plan_dijkstra_batch(g, robots, goal_dist_offset=args.offset, want_fields=False, path_cap=16384, want_stats=False)
            if k:
                ts.append(time.perf_counter() - t0)
            assert (r["codes"] == 0).all()
        st = ctx.stats()                                                # untimed: takes the settled-vertex count of the last batch
        out[f"batch_{B}"] = {"plans_per_s": B / float(np.median(ts)), "ms_per_step": float(np.median(ts)) * 1e3,
                             "propagation_ms": st["ms_propagation"], "roofline": roofline_of(st, ctx.last_engine())}
    out["workload"] = "C2 batches, vertex paths only (no finalize pass / vector map)"
    return out


def leg_c5(ctx, mesh, edge_w, costs, robot, args):
    """BASELINE config 5 on one GPU: 64 goal vertices drawn with rng(5) among the cost-free vertices, common robot
    vertex, ONE batch (with N GPUs the goals are sharded by rank, mesh_navigation_amd/multi.py)."""
    goals = np.random.default_rng(5).choice(mesh.V, size=64, replace=False).astype(np.uint32)
    tg = np.full(64, robot, np.uint32)
    ts = []
    r = None
    for k in range(13):
        t0 = time.perf_counter()
        r = ctx.plan_dijkstra_batch(goals, tg, goal_dist_offset=args.offset, want_fields=False, path_cap=16384)
        if k >= 3:
            ts.append(time.perf_counter() - t0)
    assert (r["codes"] == 0).all()
    med = float(np.median(ts))
    st = r["stats"]
    out = {"workload": "C5: 64 concurrent goals, common robot vertex, 1M-vertex C2 mesh, one batch on one GPU",
           "plans_per_s": 64 / med, "ms_per_batch": med * 1e3, "ms_per_batch_p95": float(np.percentile(ts, 95)) * 1e3,
           "roofline": roofline_of(st, "k_tile_round" if st["launches"] > 1 else "k_plan_async (asynchronous tile engine: one launch for the 64 plans)")}
    if not args.no_cpu:
        from oracle import oracle as O
        om = O.OracleMesh(mesh.xyz, mesh.faces)
        t_sum, ok = 0.0, True
        n = 8
        for k in range(n):
            ref = om.dijkstra(edge_w, costs, int(goals[k]), robot, goal_dist_offset=args.offset)
            t_sum += ref.stats["t_init_ms"] + ref.stats["t_propagation_ms"] + ref.stats["t_backtrack_ms"]
            ok = ok and ref.code == int(r["codes"][k]) and np.array_equal(ref.path, r["paths"][k])
        out["cpu_baseline"] = {"value": n / (t_sum * 1e-3), "unit": "plans/s", "cores": 1, "kind": "port",
                               "sample": f"first {n} of the 64 goals, oracle Dijkstra single thread", "gpu_paths_match_oracle": bool(ok)}
    return out


def leg_adapter(mesh, vnrm, fnrm, edge_w, costs, first, robot):
    """ms per MeshPlanner::makePlan through the adapter (mesh_navigation_amd/csrc/adapter: nearest-vertex lookup, device
    plan, pose assembly; V-sized results stay on the device), 1M mesh, cost arrays unchanged between calls."""
    from mesh_navigation_amd.planner import DijkstraMeshPlanner
    pl = DijkstraMeshPlanner()
    t0 = time.perf_counter()
    ok = pl.initialize("dijkstra_mesh_planner", dict(xyz=mesh.xyz, faces=mesh.faces, edges=mesh.edges, vertex_normals=vnrm, face_normals=fnrm,
                                                      vertex_costs=costs, edge_weights=edge_w, invalid=None))
    t_init = time.perf_counter() - t0
    assert ok
    pl.set_cost_version(1)                                              # the map's change counter: nothing is hashed per call
    def pose(v):
        p = mesh.xyz[int(v)]
        return np.array([p[0], p[1], p[2], 0, 0, 0, 1], np.float64)
    ts, n_poses = [], 0
    for k in range(23):
        g = int(first[0][k])
        t0 = time.perf_counter()
        code, plan, cost, msg = pl.makePlan(pose(robot), pose(g))
        if k >= 3:
            ts.append(time.perf_counter() - t0)
            n_poses += len(plan)
        assert code == 0, (code, msg)
    pl.close()
    return {"workload": "DijkstraMeshPlanner::makePlan through the plugin adapter, 1M-vertex C2 mesh, 20 goals after 3 warm-ups",
            "ms_per_makeplan": float(np.median(ts)) * 1e3, "ms_per_makeplan_p95": float(np.percentile(ts, 95)) * 1e3,
            "poses_per_plan": n_poses / len(ts), "initialize_s": t_init}


def leg_c3(local_rank, args):
    """BASELINE config 3: CVP wavefront on the 1M mesh (N=1000, seed 3) with Steepness + Inflation vertex costs.  The
    cost stack (steepness -> inflation wave -> weighted sum -> edge weights, edge_cost_factor 1) is built ON THE DEVICE."""
    from mesh_navigation_amd import capi, meshgen
    N = args.grid
    mesh = meshgen.terrain(N, 0.1, 3)
    vnrm, _ = vertex_normals(mesh)
    ctx = capi.MnavContext(local_rank)
    try:
        ctx.upload_mesh(mesh.xyz, mesh.faces, mesh.edges, vnrm)
        def stack(threshold):
            # three times in a row, every run reported, the MEDIAN quoted: the first call after a host-side phase (seconds of numpy /
            # scipy with the GPU idle) runs the same 95-step graph up to 16x slower now and then -- same steps, same evaluations, the
            # launch takes 0.07 ms as always and the wait 81 instead of 5 ms (tools/r06_infl_hunt.sh)
            runs = []
            for _ in range(3):
                t0 = time.perf_counter()
                ctx.layer_steepness(0, threshold)
                infl_r = ctx.layer_inflation(1, 0)                      # InflationLayer defaults
                ctx.combine_layers([0, 1], [1.0, 1.0], mode="avg", edge_cost_factor=1.0)
                runs.append(((time.perf_counter() - t0) * 1e3, infl_r))
            order = sorted(range(3), key=lambda k: runs[k][0])
            ms, infl = runs[order[1]]
            infl = dict(infl, ms_wave_of_each_run=[round(r[1]["ms_wave"], 3) for r in runs], stack_ms_of_each_run=[round(r[0], 3) for r in runs])
            vc, w = ctx.download_costs()
            _, lethal = ctx.layer_download(0)
            import scipy.sparse as sp
            import scipy.sparse.csgraph as cg
            fr = vc < 1.0                                               # vertices a wave may run over (cost_limit 1)
            e = mesh.edges
            ok = fr[e[:, 0]] & fr[e[:, 1]]
            g = sp.coo_matrix((np.ones(int(ok.sum()), np.int8), (e[ok, 0], e[ok, 1])), shape=(mesh.V, mesh.V)).tocsr()
            _, lab = cg.connected_components(g, directed=False)
            sizes = np.bincount(lab[fr]) if fr.any() else np.zeros(1, np.int64)
            big = int(np.argmax(sizes))
            info = {"steepness_threshold": threshold, "lethal_vertices": int(lethal.sum()), "traversable_vertices": int(fr.sum()),
                    "largest_traversable_component": int(sizes.max()), "cost_stack_on_device_ms": ms, "inflation_wave": infl}
            return vc, w, lethal, np.nonzero((lab == big) & fr)[0], info
        # As specified (threshold 0.3 rad) the seed-3 terrain is 56 % lethal and, after inflation, falls apart into
        # islands of < 300 vertices: no wave gets anywhere.  The cost stack is timed on it; the PLANS are measured with
        # the threshold at 0.6 rad (3 % lethal, one component of 87 % of the mesh) -- stated in `workload`.
        _, _, _, _, spec = stack(0.3)
        vc, w, lethal, free, used = stack(0.6)
        t_costs = used["cost_stack_on_device_ms"] * 1e-3
        infl = used["inflation_wave"]
        rng = np.random.default_rng(5)
        first_face = np.full(mesh.V, -1, np.int64)
        fl = mesh.faces.ravel()
        first_face[fl[::-1]] = (np.arange(fl.size)[::-1] // 3)
        def wave_seed(v):
            f = int(first_face[v])
            return mesh.xyz[mesh.faces[f]].astype(np.float64).mean(axis=0).astype(np.float32), f
        robot = int(free[np.argmin(np.abs(mesh.xyz[free, 0] - 0.9 * N * 0.1) + np.abs(mesh.xyz[free, 1] - 0.9 * N * 0.1))])
        free = free[vc[free] < 0.5]                                       # goals well inside the traversable component
        tf = int(first_face[robot])
        goals = rng.choice(free, size=160, replace=False)               # (the draw of the earlier rounds: same single plans, same batch of 128)
        goals512 = rng.choice(free, size=512, replace=False)
        lat, codes, st = [], [], None
        for k in range(13):
            sp, sf = wave_seed(int(goals[k]))
            o = ctx.plan_cvp(sp, sf, tf, want_fields=False, want_vecmap=False)
            codes.append(int(o.code))
            if k >= 3:
                lat.append(o.stats["ms_total"])
                st = o.stats
        # with the vector map (what the back-tracking of makePlan reads), left resident
        ctx.set_resident_outputs(True)
        latv = []
        for k in range(8):
            sp, sf = wave_seed(int(goals[k]))
            o = ctx.plan_cvp(sp, sf, tf, want_fields=False, want_vecmap=False)
            if k >= 2:
                latv.append(o.stats["ms_total"])
        ctx.set_resident_outputs(False)
        # batches run the wide step kernel (k_cvp_ctl + k_step_wide + k_step_repair per step); 128 plans as in the earlier rounds,
        # and 512 (a step of 128 plans is ~1.3 rounds of the resident waves: a quarter of the time is the last, half-empty round)
        def batch(nb):
            seeds = [wave_seed(int(v)) for v in (goals[16:16 + nb] if nb <= 128 else goals512[:nb])]
            sps = np.stack([x[0] for x in seeds]); sfs = np.array([x[1] for x in seeds], np.uint32)
            ctx.plan_cvp_batch(sps, sfs, np.full(nb, tf, np.uint32))
            t0 = time.perf_counter()
            r = ctx.plan_cvp_batch(sps, sfs, np.full(nb, tf, np.uint32))
            return r, time.perf_counter() - t0
        nb = 128
        rb, tb = batch(nb)
        rb512, tb512 = batch(512)
        # full-field variant (goal_dist_offset = +inf): the cleanest roofline denominator
        sp0, sf0 = wave_seed(int(goals[0]))
        ctx.plan_cvp(sp0, sf0, tf, goal_dist_offset=float("inf"), want_fields=False, want_vecmap=False)
        of = ctx.plan_cvp(sp0, sf0, tf, goal_dist_offset=float("inf"), want_fields=False, want_vecmap=False)
        out = {"workload": f"C3: CVP wavefront, {N}x{N} terrain seed 3 = {mesh.V} vertices, Steepness + Inflation(defaults) avg-combined on the device, "
                           f"edge_cost_factor 1, cost_limit 1, goal_dist_offset 0.3; steepness threshold 0.6 rad for the plans "
                           f"(the specified 0.3 leaves no traversable component above {spec['largest_traversable_component']} vertices)",
               "cost_stack_as_specified": spec, "cost_stack_used": used,
               "full_field": {"ms_per_plan": of.stats["ms_total"], "code": int(of.code), "roofline": roofline_of(of.stats, "k_step<cvp>")},
               "ms_per_plan_single": float(np.median(lat)), "ms_per_plan_single_p95": float(np.percentile(lat, 95)),
               "ms_per_plan_single_with_vector_map": float(np.median(latv)),
               "codes_single": sorted(set(codes)), "batch": nb, "plans_per_s_batch": nb / tb,
               "codes_batch": sorted(set(int(c) for c in rb["codes"])),
               "batch_512": {"plans_per_s": 512 / tb512, "ms_per_batch": tb512 * 1e3, "codes": sorted(set(int(c) for c in rb512["codes"])),
                             "roofline": roofline_of(rb512["stats"], "k_step_wide (one step = k_cvp_ctl + k_step_wide + k_step_repair)")},
               "roofline": roofline_of(rb["stats"], "k_step_wide (one step = k_cvp_ctl + k_step_wide + k_step_repair)"),
               "roofline_single_plan": roofline_of(st, "k_step<cvp>")}
        if not args.no_cpu:
            from oracle import oracle as O
            om = O.OracleMesh(mesh.xyz, mesh.faces)
            vn = vnrm
            t_sum, n, ok = 0.0, 3, True
            tw = time.perf_counter()
            for k in range(n):
                sp, sf = wave_seed(int(goals[k]))
                ref = om.cvp(w, vc, vn, sp, sf, tf)
                t_sum += ref.stats["t_init_ms"] + ref.stats["t_propagation_ms"]
                ok = ok and ref.code == codes[k]
            out["cpu_baseline"] = {"value": n / (t_sum * 1e-3), "unit": "plans/s", "cores": 1, "kind": "port", "ms_per_plan": t_sum / n,
                                   "sample": f"{n} of the single plans, oracle CVP wavefront single thread (propagation, no back-tracking), {time.perf_counter() - tw:.1f} s wall",
                                   "codes_match_oracle": bool(ok)}
        return out
    finally:
        ctx.close()


def leg_c4(local_rank, args):
    """BASELINE config 4 on ONE GPU: Dijkstra on the 10M-vertex terrain (N=3163, seed 4, C2 settings): single plan and a
    batch.  The range-partitioned version of the same plan over N GPUs is `--config C4 --gpus N`."""
    from mesh_navigation_amd import capi, meshgen
    N = args.c4_grid
    t0 = time.perf_counter()
    mesh = meshgen.terrain(N, 0.1, 4)
    edge_w = meshgen.edge_lengths(mesh)
    costs = np.zeros(mesh.V, np.float32)
    t_gen = time.perf_counter() - t0
    ctx = capi.MnavContext(local_rank)
    try:
        t0 = time.perf_counter()
        ctx.upload_mesh(mesh.xyz, mesh.faces, mesh.edges, None)
        ctx.upload_costs(costs, edge_w)
        t_up = time.perf_counter() - t0
        robot = mesh.vertex_at(0.9, 0.9)
        rng = np.random.default_rng(5)
        B = args.c4_batch
        goals = rng.choice(mesh.V, size=B, replace=False).astype(np.uint32)
        lat, st = [], None
        for k in range(7):
            o = ctx.plan_dijkstra(int(goals[k]), robot, goal_dist_offset=args.offset, want_fields=False)
            assert o.code == 0
            if k >= 2:
                lat.append(o.stats["ms_total"]); st = o.stats
        tg = np.full(B, robot, np.uint32)
        ctx.plan_dijkstra_batch(goals, tg, goal_dist_offset=args.offset, want_fields=False, path_cap=65536, want_stats=False)
        tb = time.perf_counter()
        rb = ctx.plan_dijkstra_batch(goals, tg, goal_dist_offset=args.offset, want_fields=False, path_cap=65536, want_stats=False)
        tb = time.perf_counter() - tb
        assert (rb["codes"] == 0).all()
        batch_kernel = ctx.last_engine()
        sb = ctx.stats()                                                # untimed: the settled-vertex count (k_tb_count, instrumentation for the
                                                                        # algorithmic bytes) is taken here, after the clock -- it was 6 % of ms_per_batch
        out = {"workload": f"C4: delta-stepping SSSP, {N}x{N} terrain seed 4 = {mesh.V} vertices, uniform edge costs, goal_dist_offset {args.offset:g}, one GPU",
               "vertices": mesh.V, "edges": mesh.E, "mesh_generation_s": t_gen, "upload_and_tiling_s": t_up,
               "ms_per_makeplan_single": float(np.median(lat)), "ms_per_makeplan_single_p95": float(np.percentile(lat, 95)),
               "batch": B, "plans_per_s_batch": B / tb, "ms_per_batch": tb * 1e3,
               "roofline": roofline_of(sb, batch_kernel),
               "roofline_single_plan": roofline_of(st, "k_plan_async (asynchronous tile engine: one launch per plan)" if st["launches"] == 1 else "k_tile_round")}
        try:                                                            # HBM bytes of this leg by PMC: a committed profile of the same command, never this run
            pm = json.load(open(os.path.join(ROOT, "profiles", "r06_c4_final_pmc.json")))
            if int(pm.get("batch", 0)) == B and int(pm.get("grid", 0)) == N and str(pm.get("kernel", "")) in str(batch_kernel):
                out["roofline"]["traffic"] = {"profiled": pm["traffic_bytes_per_launch"], "profiled_high": pm.get("traffic_bytes_per_launch_high"),
                                              "source": "profiles/r06_c4_final_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/gpu_c4_batch.py, not this run)"}
        except (OSError, ValueError, KeyError):
            pass
        if not args.no_cpu:
            from oracle import oracle as O
            om = O.OracleMesh(mesh.xyz, mesh.faces)
            t_sum, n, ok = 0.0, 2, True
            for k in range(n):
                ref = om.dijkstra(edge_w, costs, int(goals[k]), robot, goal_dist_offset=args.offset)
                t_sum += ref.stats["t_init_ms"] + ref.stats["t_propagation_ms"] + ref.stats["t_backtrack_ms"]
                ok = ok and ref.code == 0 and np.array_equal(ref.path, rb["paths"][k])
            out["cpu_baseline"] = {"value": n / (t_sum * 1e-3), "unit": "plans/s", "cores": 1, "kind": "port", "ms_per_plan": t_sum / n,
                                   "sample": f"{n} plans of the batch, oracle Dijkstra single thread", "gpu_paths_match_oracle": bool(ok)}
        return out
    finally:
        ctx.close()


def sharded_c4(args, torch, dist, rank, local_rank, world):
    """`--config C4 --gpus N`: ONE Dijkstra plan on the 10M mesh, range-partitioned over the N GPUs (strong scaling).
    A step = one plan; every rank holds the mesh, owns a range of the Morton-ordered tiles and exchanges the interface
    distances with one RCCL min-allreduce per block of local rounds (mesh_navigation_amd/sharded.py)."""
    from mesh_navigation_amd import capi, meshgen, sharded
    N = args.c4_grid
    mesh = meshgen.terrain(N, 0.1, 4)
    edge_w = meshgen.edge_lengths(mesh)
    costs = np.zeros(mesh.V, np.float32)
    ctx = capi.MnavContext(local_rank)
    replicated = os.environ.get("MNAV_SHARD_REPLICATED") is not None
    if replicated:                                                    # every rank holds the whole mesh, ownership of the tiles is partitioned
        ctx.upload_mesh(mesh.xyz, mesh.faces, mesh.edges, None)
        ctx.upload_costs(costs, edge_w)
        eng = sharded.GpuShardEngine(ctx, rank, world)
    else:                                                             # default: the DATA is partitioned -- a rank uploads its part only
        owner = sharded.partition_vertices(mesh.xyz, world)
        part = sharded.extract_part(mesh.xyz, mesh.edges, owner, rank, world)
        sharded.PartitionedShardEngine.upload_part(ctx, part, costs, edge_w)
        eng = sharded.PartitionedShardEngine(ctx, part)
    red = sharded.torch_allreduce_min(dist) if dist is not None else (lambda x: None)
    robot = mesh.vertex_at(0.9, 0.9)
    goals = np.random.default_rng(5).choice(mesh.V, size=args.steps + args.warmup, replace=False)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    dev_loop = os.environ.get("MNAV_SHARD_HOST_LOOP") is None         # default: the exchange loop stays on the device
    res = None
    for k in range(args.warmup):
        res = sharded.run_sharded_plan(eng, red, int(goals[k]), robot, args.offset, max_exchanges=200000, device_loop=dev_loop, gather=replicated)
    barrier()
    t0 = time.perf_counter()
    exch = 0
    for k in range(args.steps):
        res = sharded.run_sharded_plan(eng, red, int(goals[args.warmup + k]), robot, args.offset, max_exchanges=200000, device_loop=dev_loop, gather=replicated)
        assert res.code == 0
        exch += res.exchanges
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        out = {"metric": "plans/sec (one Dijkstra plan range-partitioned over the GPUs, 10M-vertex mesh)", "value": args.steps / elapsed,
               "unit": "plans/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"C4 sharded: {N}x{N} terrain = {mesh.V} vertices, "
                                      + ("mesh replicated, tiles range-partitioned" if replicated else "mesh DATA partitioned (a rank holds its part + 1-ring halo)")
                                      + f" over {world} GPU(s), min-allreduce of {eng.n} interface floats per exchange", "exchanges_per_plan": exch / args.steps,
                          "vertices_on_rank0": int(ctx.V), "device_bytes_rank0": int(ctx.device_bytes()),
                          "path_len": int(len(res.path)),
                          "exchange_loop": "device-resident (termination words read every 8 exchanges)" if dev_loop else "host-checked every exchange"},
               "roofline": None, "cpu_baseline": None}
        print(json.dumps(out), flush=True)
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


_OM = None


def cpu_all_cores_child(args):
    """CPU only (no torch, no HIP): the oracle on every host core, same mesh / goals / robot vertex as the GPU run."""
    import multiprocessing as mp
    from mesh_navigation_amd import meshgen
    from oracle import oracle as O
    global _OM
    mesh = meshgen.terrain(args.grid, 0.1, 2)
    edge_w = meshgen.edge_lengths(mesh)
    costs = np.zeros(mesh.V, np.float32)
    _OM = O.OracleMesh(mesh.xyz, mesh.faces)
    robot = mesh.vertex_at(0.9, 0.9)
    goals = np.random.default_rng(5).choice(mesh.V, size=args.batch, replace=False)
    visible = os.cpu_count() or 1
    quota = cpu_quota()                                                  # CPUs' worth of time the cgroup grants per period (None: no limit)
    cores = max(1, min(visible, int(quota))) if quota else visible       # more workers than that are only frozen in turn
    per = 2
    jobs = [(edge_w, costs, [int(goals[(c * per + i) % args.batch]) for i in range(per)], robot, args.offset) for c in range(cores)]
    with mp.get_context("fork").Pool(cores) as pool:
        pool.map(_oracle_worker, jobs, chunksize=1)                      # warm: page in the forked mesh
        ta = time.perf_counter()
        done = sum(pool.map(_oracle_worker, jobs, chunksize=1))
        ta = time.perf_counter() - ta
    print(json.dumps({"value": done / ta, "unit": "plans/s", "cores": cores, "cpus_visible": visible, "cgroup_cpu_quota": quota,
                      "sample": f"{done} plans, {per} per forked worker, {ta:.2f} s wall"}), flush=True)


def cpu_quota():
    """CPUs' worth of time per period the container's cgroup grants (cgroup v2 cpu.max / v1 cfs quota); None when unlimited."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def _oracle_worker(job):
    """One host core: plans a few goals with the forked, read-only oracle mesh."""
    edge_w, costs, goals, target, offset = job
    for g in goals:
        _OM.dijkstra(edge_w, costs, int(g), int(target), goal_dist_offset=offset)
    return len(goals)


def cpu_baseline(mesh, edge_w, costs, first, B, offset=0.3):
    """The oracle (C restatement of dijkstra_mesh_planner.cpp:217-398, -O3, one thread) timed on this
    box's host cores on the plans of the first batch; also checks the GPU paths of that batch."""
    from oracle import oracle as O
    om = O.OracleMesh(mesh.xyz, mesh.faces)
    g, t, r = first
    n = min(B, 64)
    t_sum = 0.0
    ok = True
    tw = time.perf_counter()
    for k in range(n):
        ref = om.dijkstra(edge_w, costs, int(g[k]), int(t[k]), goal_dist_offset=offset)
        t_sum += ref.stats["t_init_ms"] + ref.stats["t_propagation_ms"] + ref.stats["t_backtrack_ms"]
        ok = ok and ref.code == int(r["codes"][k]) and np.array_equal(ref.path, r["paths"][k])
    wall = time.perf_counter() - tw
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    # the same oracle on ALL host cores: measured by a fresh child process that never loads
    # the GPU runtime and forks one worker per core over the shared, read-only mesh
    all_cores = None
    try:
        import subprocess
        cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-all-cores-child", "--grid", str(mesh.N),
                             "--offset", repr(float(offset)), "--batch", str(B)], capture_output=True, text=True, timeout=120)
        lines = [l for l in cp.stdout.splitlines() if l.startswith("{")]
        all_cores = json.loads(lines[-1]) if lines else {"error": (cp.stderr or "no output")[-300:]}
    except Exception as e:                                               # the single-core figure stands on its own
        all_cores = {"error": repr(e)}
    # the reference's OWN planner code on the same plans (oracle/_ref: dijkstra_mesh_planner.cpp + mesh_map.cpp compiled
    # unmodified against stub lvr2 / ROS headers -- the containers are the stubs', so this is the reference's algorithm and
    # code, not lvr2's memory behaviour): reported next to the port, which is the faster of the two and stays the `value`
    reference_code = None
    try:
        from oracle import ref as R
        if R.available():
            tb = time.perf_counter()
            rm = R.RefMap(mesh.xyz, mesh.faces)
            t_build = time.perf_counter() - tb
            nr, t_ref, same = min(n, 8), 0.0, True
            for k in range(nr):
                t1 = time.perf_counter()
                rr = rm.dijkstra(mesh.xyz[int(g[k])], mesh.xyz[int(t[k])], goal_dist_offset=offset, fields=False)
                t_ref += time.perf_counter() - t1
                same = same and rr.code == int(r["codes"][k]) and np.array_equal(rr.path, r["paths"][k])
            reference_code = {"value": nr / t_ref, "unit": "plans/s", "cores": 1, "ms_per_plan": t_ref / nr * 1e3,
                              "sample": f"first {nr} plans of the first batch through DijkstraMeshPlanner::dijkstra of oracle/_ref (wall clock incl. nearest-vertex lookup)",
                              "map_build_s": t_build, "gpu_paths_match_reference_code": bool(same)}
            if R.gpu_plugins_linked():
                # the REAL plugin (integration/mesh_gpu_planners, loaded by pluginlib lookup name) on the reference's own MeshMap,
                # next to the reference planner's makePlan on the same map: wall time per makePlan, plans compared pose by pose
                def pose(pt):
                    return np.array([pt[0], pt[1], pt[2], 0, 0, 0, 1], np.float64)
                ref_ms, ref_plans = [], []
                for k in range(nr):
                    t1 = time.perf_counter()
                    code, plan, _ = rm.dijkstra_make_plan(pose(mesh.xyz[int(t[k])]), pose(mesh.xyz[int(g[k])]), goal_dist_offset=offset)
                    ref_ms.append((time.perf_counter() - t1) * 1e3)
                    ref_plans.append((code, plan))
                plug, match = {}, True
                for label, params in (("default", {}),
                                      ("reference_side_effects", dict(reference_side_effects=True)),
                                      ("static_costs", dict(static_costs=True))):
                    if not rm.plugin_init("mesh_gpu_planners/GpuDijkstraMeshPlanner", "bench_" + label, goal_dist_offset=float(offset), **params):
                        plug[label] = None
                        continue
                    rm.plugin_make_plan(pose(mesh.xyz[int(t[0])]), pose(mesh.xyz[int(g[0])]))      # tables, graphs
                    ms = []
                    for k in range(nr):
                        t1 = time.perf_counter()
                        code, plan, _, _ = rm.plugin_make_plan(pose(mesh.xyz[int(t[k])]), pose(mesh.xyz[int(g[k])]))
                        ms.append((time.perf_counter() - t1) * 1e3)
                        match = match and code == ref_plans[k][0] and np.array_equal(plan, ref_plans[k][1], equal_nan=True)
                    plug[label] = float(np.median(ms))
                    rm.plugin_release()
                reference_code["make_plan"] = {"reference_planner_ms": float(np.median(ref_ms)), "gpu_plugin_ms": plug,
                                               "plugin_plans_equal_reference_plans": bool(match),
                                               "sample": f"{nr} makePlan calls each, median wall time, same MeshMap object"}
    except Exception as e:                                               # noqa: BLE001 -- the port figure stands on its own
        reference_code = {"error": repr(e)[:200]}
    return {"value": n / (t_sum * 1e-3), "unit": "plans/s", "cores": 1, "kind": "port", "reference_code": reference_code, "all_host_cores": all_cores,
            "sample": f"first {n} plans of the first batch, oracle Dijkstra single thread, {wall:.1f} s wall",
            "ms_per_plan": t_sum / n, "host_cpu": model, "host_cores_available": os.cpu_count(), "cgroup_cpu_quota": cpu_quota(),
            "gpu_paths_match_oracle": bool(ok)}


if __name__ == "__main__":
    main()
