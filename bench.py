#!/usr/bin/env python
"""bench.py -- plans/sec of the MI355X wavefront planner on BASELINE config C2.

One "step" = one batch of B (default 5120 = 4 x the 1280 plans the device runs at once, longest first) independent Dijkstra (delta-stepping SSSP) plans on the 1M-vertex
synthetic terrain (N=1000, h=0.1 m, seed 2, edge_cost_factor 0, reference default
cut-offs goal_dist_offset 0.3 / cost_limit 1.0), B goal vertices drawn per step, common robot
vertex (the concurrent-goals shape of BASELINE config 5).  Mesh and costs are resident in HBM
before the timed region; each plan returns its vertex-index path, the V-sized fields stay on the
device.  For N > 1 every rank (one per GPU) runs its own batches on its own replica of the mesh:
the path shards by plan, there is no data-path collective (weak scaling).

Prints ONE JSON line (see the task contract).  `value` = plans/s of the whole job.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 GB/s measured copy


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("MNAV_BENCH_BATCH", "5120")))
    ap.add_argument("--grid", type=int, default=int(os.environ.get("MNAV_BENCH_N", "1000")))
    ap.add_argument("--offset", type=float, default=float(os.environ.get("MNAV_BENCH_OFFSET", "0.3")),
                    help="goal_dist_offset (reference default 0.3; inf = full-field variant of SURVEY.md 8d)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-all-cores-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-latency", action="store_true", help="skip the single-plan latency runs (profiling)")
    args = ap.parse_args()
    if args.cpu_all_cores_child:
        return cpu_all_cores_child(args)

    import torch

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

    N, B = args.grid, args.batch
    mesh = meshgen.terrain(N, 0.1, 2)
    edge_w = meshgen.edge_lengths(mesh)                  # edge_cost_factor 0 -> weights == edge distances
    costs = np.zeros(mesh.V, np.float32)
    ctx = capi.MnavContext(local_rank)
    # vertex normals (normalised sum of unit face normals) -- only the CVP vector map reads them
    p = mesh.xyz.astype(np.float64)
    fnrm = np.cross(p[mesh.faces[:, 1]] - p[mesh.faces[:, 0]], p[mesh.faces[:, 2]] - p[mesh.faces[:, 0]])
    fnrm /= np.maximum(np.linalg.norm(fnrm, axis=1, keepdims=True), 1e-30)
    vnrm = np.zeros_like(p)
    for k in range(3):
        np.add.at(vnrm, mesh.faces[:, k], fnrm)
    vnrm /= np.maximum(np.linalg.norm(vnrm, axis=1, keepdims=True), 1e-30)
    ctx.upload_mesh(mesh.xyz, mesh.faces, mesh.edges, vnrm.astype(np.float32))
    ctx.upload_costs(costs, edge_w)
    robot = mesh.vertex_at(0.9, 0.9)
    rng = np.random.default_rng(5 + 1000 * rank)

    def batch_goals():
        g = rng.choice(mesh.V, size=B, replace=False).astype(np.uint32)
        return g, np.full(B, robot, np.uint32)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    first = None
    for _ in range(max(args.warmup, 0)):
        g, t = batch_goals()
        r = ctx.plan_dijkstra_batch(g, t, goal_dist_offset=args.offset, want_fields=False, path_cap=16384)
        if first is None:
            first = (g, t, r)
    prop_ms = kern_ms = launches = algo = 0.0
    settled = 0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        g, t = batch_goals()
        r = ctx.plan_dijkstra_batch(g, t, goal_dist_offset=args.offset, want_fields=False, path_cap=16384)
        assert (r["codes"] == 0).all(), r["codes"]
        st = r["stats"]
        prop_ms += st["ms_propagation"]; kern_ms += st["ms_step_kernels"]; launches += st["launches"]; algo += st["algorithmic_bytes"]
        settled += st["settled"]
        if first is None:
            first = (g, t, r)
    barrier()
    elapsed = time.perf_counter() - t0
    from mesh_navigation_amd import multi
    total_plans, elapsed = multi.aggregate_throughput(B * args.steps, elapsed, dist)   # sum of plans, MAX time over ranks

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
        # roofline of the dominant kernel: algorithmic bytes per launch (24 B per
        # settled vertex + 24 B per incident edge, summed over the batch) divided by the average
        # launch duration, measured live with HIP events that the library records on ITS OWN stream
        # around the launch(es): batches of >= 128 plans run as ONE launch of k_plan_persistent (one
        # workgroup per plan); smaller batches as hipGraph replays of 24 k_tile_round launches.
        # HBM traffic per launch from the PMC passes committed under profiles/ (tools/prof_pmc.sh, same
        # command and workload; FETCH_SIZE doubled for the 16-byte staging reads as the microarch guide
        # prescribes, WRITE_SIZE as is) -- only quoted when it was measured on this very workload.
        traffic = None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            if pm.get("kernel") == "k_plan_persistent" and launches <= args.steps and B == pm.get("batch", 1024) and N == 1000:
                traffic = pm["traffic_bytes_per_launch_high"]
        except (OSError, ValueError, KeyError):
            pass
        per_launch_bytes = algo / max(launches, 1)
        per_launch_s = kern_ms * 1e-3 / max(launches, 1)
        achieved = per_launch_bytes / per_launch_s / 1e9 if per_launch_s > 0 else 0.0
        out = {
            "metric": "plans/sec (Dijkstra makePlan device path, 1M-vertex mesh)",
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
            "config": {"workload": f"C2: delta-stepping SSSP (tiled label-correcting), {N}x{N} terrain = {mesh.V} vertices, "
                                   f"uniform edge costs, batch of {B} goals per step per GPU, common robot vertex, goal_dist_offset {args.offset:g}",
                       "vertices": mesh.V, "edges": mesh.E, "batch_per_gpu": B,
                       "parallelism": f"{world} independent replicas (plans sharded by rank)"},
            "ms_per_makeplan_single": single_ms,
            "ms_per_makeplan_single_p95": single_p95,
            "ms_per_plan_in_batch": ms_step / B,
            "cvp_planner_same_mesh": cvp,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": "k_plan_persistent" if launches <= args.steps else "k_tile_round", "launches_per_step": launches / args.steps,
                         "algorithmic_bytes_per_step": algo / args.steps,
                         "avg_launch_us": per_launch_s * 1e6, "propagation_ms_per_step": prop_ms / args.steps,
                         "settled_vertices_per_plan": settled / max(args.steps * B, 1)},
        }
        if not args.no_cpu and world >= 1:
            out["cpu_baseline"] = cpu_baseline(mesh, edge_w, costs, first, B, args.offset)
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
    cores = os.cpu_count() or 1
    per = 2
    jobs = [(edge_w, costs, [int(goals[(c * per + i) % args.batch]) for i in range(per)], robot, args.offset) for c in range(cores)]
    with mp.get_context("fork").Pool(cores) as pool:
        pool.map(_oracle_worker, jobs, chunksize=1)                      # warm: page in the forked mesh
        ta = time.perf_counter()
        done = sum(pool.map(_oracle_worker, jobs, chunksize=1))
        ta = time.perf_counter() - ta
    print(json.dumps({"value": done / ta, "unit": "plans/s", "cores": cores,
                      "sample": f"{done} plans, {per} per forked worker, {ta:.2f} s wall"}), flush=True)


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
    return {"value": n / (t_sum * 1e-3), "unit": "plans/s", "cores": 1, "kind": "port", "all_host_cores": all_cores,
            "sample": f"first {n} plans of the first batch, oracle Dijkstra single thread, {wall:.1f} s wall",
            "ms_per_plan": t_sum / n, "host_cpu": model, "host_cores_available": os.cpu_count(),
            "gpu_paths_match_oracle": bool(ok)}


if __name__ == "__main__":
    main()
