"""Soak of the plan paths against the oracle on the GPU box (round 5): random goals, batch sizes and goal_dist_offsets (negative
ones included) through every Dijkstra engine `auto` uses and both CVP step kernels.  Every device plan carries the library's
own fixed-point check (a mismatch is INTERNAL_ERROR); a sample of each leg is compared with the oracle bit for bit.
    python tools/gpu_soak.py [seconds per leg, default 40]  ->  one JSON line (profiles/r05_soak.json)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402  (HIP runtime order, see tests/conftest.py)

torch.cuda.init()
from mesh_navigation_amd import capi, meshgen  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests.common import Case, layered_costs  # noqa: E402

BUDGET = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
OFFSETS = (0.3, 0.0, 2.5, float("inf"), -0.05, -1.0, float("-inf"))
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "20260925")))
out = {"seed_budget_s": BUDGET, "legs": {}}


def beq(a, b):
    return np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))


def dijkstra_leg(name, case, ctx, batch_sizes, t_budget, oracle_every, limits=(1.0,)):
    m = case.mesh
    t0 = time.perf_counter()
    plans = calls = checked = 0
    bad = []
    while time.perf_counter() - t0 < t_budget:
        n = int(rng.choice(batch_sizes))
        off = float(OFFSETS[int(rng.integers(len(OFFSETS)))])
        seeds = rng.choice(m.V, n, replace=False).astype(np.uint32)
        targets = np.full(n, int(rng.integers(m.V)), np.uint32) if rng.random() < 0.7 else rng.integers(0, m.V, n).astype(np.uint32)
        fields = bool(rng.random() < 0.5) and n <= 512
        lim = float(rng.choice(limits))
        if n == 1:
            o = ctx.plan_dijkstra(int(seeds[0]), int(targets[0]), goal_dist_offset=off, cost_limit=lim, want_fields=fields)
            codes, paths = np.array([o.code]), [o.path]
            dist = [o.dist] if fields else None
            pred = [o.pred] if fields else None
        else:
            b = ctx.plan_dijkstra_batch(seeds, targets, goal_dist_offset=off, cost_limit=lim, want_fields=fields, path_cap=65536)
            codes, paths = b["codes"], b["paths"]
            dist = b["dist"] if fields else None
            pred = b["pred"] if fields else None
        calls += 1
        plans += n
        if not np.isin(codes, (0, 54)).all():
            bad.append(dict(leg=name, n=n, off=off, codes=sorted(set(int(c) for c in codes))))
            continue
        if calls % oracle_every == 0 or calls == 1:
            for k in rng.choice(n, min(n, 2), replace=False):
                ref = case.om.dijkstra(case.weights, case.costs, int(seeds[k]), int(targets[k]), goal_dist_offset=off, cost_limit=lim, invalid=case.invalid)
                ok = int(codes[k]) == ref.code and np.array_equal(paths[k], ref.path)
                if fields:
                    ok = ok and beq(dist[k], ref.dist) and np.array_equal(pred[k], ref.pred)
                checked += 1
                if not ok:
                    bad.append(dict(leg=name, n=n, off=off, lim=lim, k=int(k), seed=int(seeds[k]), target=int(targets[k]), fields=fields))
    out["legs"][name] = dict(calls=calls, plans=plans, compared_with_oracle=checked, mismatches=len(bad), seconds=round(time.perf_counter() - t0, 1))
    return bad


def cvp_leg(name, case, ctx, t_budget):
    m = case.mesh
    free = np.where(case.costs < 0.5)[0]
    offv = np.array([0.02, 0.015, 0.0], np.float32)
    t0 = time.perf_counter()
    plans = checked = 0
    bad = []
    while time.perf_counter() - t0 < t_budget:
        off = float(OFFSETS[int(rng.integers(len(OFFSETS)))])
        s, t = (int(x) for x in rng.choice(free, 2, replace=False))
        sp, tp = m.xyz[s] + offv, m.xyz[t] + offv
        sf, _ = case.om.containing_face(sp)
        tf, _ = case.om.containing_face(tp)
        if sf < 0 or tf < 0 or sf >= m.F or tf >= m.F:
            continue
        ctx.set_option("cvp_wide", int(rng.integers(2)))
        o = ctx.plan_cvp(sp, int(sf), int(tf), goal_dist_offset=off)
        ref = case.om.cvp(case.weights, case.costs, case.vn, sp, int(sf), int(tf), goal_dist_offset=off, invalid=case.invalid)
        plans += 1
        checked += 1
        upd = ref.pred != np.arange(m.V)
        has = ref.has_vec.astype(bool)
        ok = o.code == ref.code and beq(o.dist, ref.dist) and np.array_equal(o.pred, ref.pred) and np.array_equal(o.cutface[upd], ref.cutface[upd]) \
            and beq(o.direction[upd], ref.direction[upd]) and beq(o.vecmap[has], ref.vecmap[has])
        if ok and ref.code == 0 and plans % 3 == 0:                        # the walk along the resident field (cvp :920-951)
            sw = float(rng.choice([0.4, 0.15, 0.05]))
            rc, ppos, pface = case.om.cvp_backtrack(ref.vecmap, ref.has_vec, sp, int(sf), tp, int(tf), step_width=sw, cap=8192)
            st, pos, face = ctx.backtrack_cvp(sp, int(sf), tp, int(tf), step_width=sw, cap=8192)
            ok = (st == 1) == (rc == 0) and np.array_equal(face, pface) and beq(pos, ppos)
        if not ok:
            bad.append(dict(leg=name, off=off, seed=s, target=t, code=int(o.code), ref_code=int(ref.code)))
        if plans % 40 == 0:                                                 # a batch on the wide kernel (plan groups on their own streams)
            nb = int(rng.choice([33, 48, 80]))
            vs = rng.choice(free, nb, replace=False)
            sps = np.stack([m.xyz[v] + offv for v in vs]).astype(np.float32)
            sfs = np.array([case.om.containing_face(q)[0] for q in sps], np.int64)
            keep = (sfs >= 0) & (sfs < m.F)
            sps, sfs = sps[keep], sfs[keep].astype(np.uint32)
            ctx.set_option("cvp_wide", None)
            b = ctx.plan_cvp_batch(sps, sfs, np.full(len(sfs), tf, np.uint32), goal_dist_offset=off, want_fields=True)
            for k in rng.choice(len(sfs), 2, replace=False):
                r2 = case.om.cvp(case.weights, case.costs, case.vn, sps[k], int(sfs[k]), int(tf), goal_dist_offset=off, invalid=case.invalid)
                checked += 1
                if int(b["codes"][k]) != r2.code or not beq(b["dist"][k], r2.dist) or not np.array_equal(b["pred"][k], r2.pred):
                    bad.append(dict(leg=name + "/batch", off=off, n=int(len(sfs)), k=int(k), seed_face=int(sfs[k]), target_face=int(tf)))
            plans += len(sfs)
    ctx.set_option("cvp_wide", None)
    out["legs"][name] = dict(plans=plans, compared_with_oracle=checked, mismatches=len(bad), seconds=round(time.perf_counter() - t0, 1))
    return bad


bad = []
# ---- 1M terrain (C2): single plans and small batches on the asynchronous engine, large ones on the tile-batch engine
case = Case(meshgen.terrain(1000, 0.1, 21))
ctx = capi.MnavContext(0)
case.upload(ctx)
bad += dijkstra_leg("c2_async_1_to_96_plans", case, ctx, [1, 1, 1, 2, 8, 33, 96], BUDGET, 6)
bad += dijkstra_leg("c2_tile_batch_97_to_3000_plans", case, ctx, [97, 128, 500, 1024, 3000], BUDGET, 2)
ctx.close()
# ---- 224 x 224 layered costs (C3 shape: cost-inflated triangles, cascades): CVP on both step kernels
base = Case(meshgen.terrain(224, 0.1, 1))
costs, _ = layered_costs(base, "avg")
c3 = Case(base.mesh, costs, 1.0)
ctx = capi.MnavContext(0)
c3.upload(ctx)
bad += cvp_leg("c3_layered_fragments_cvp_both_kernels", c3, ctx, BUDGET / 2)      # (threshold 0.3: components of <= 266 vertices)
ctx.close()
# ---- 160 x 160 terrain, random per-vertex costs up to 0.9 with edge_cost_factor 1 (one component, non-causal updates, cascades)
m2 = meshgen.terrain(160, 0.1, 5)
adv = Case(m2, np.random.default_rng(2).uniform(0.0, 0.9, m2.V).astype(np.float32), 1.0)
ctx = capi.MnavContext(0)
adv.upload(ctx)
bad += cvp_leg("adversarial_costs_cvp_both_kernels", adv, ctx, BUDGET)
ctx.close()
# ---- 300 x 300 terrain with random costs, 2 % invalid vertices, cost limits that cut parts of the mesh off: every Dijkstra engine
m3 = meshgen.terrain(300, 0.1, 9)
r3 = np.random.default_rng(4)
cm = Case(m3, r3.uniform(0.0, 1.0, m3.V).astype(np.float32), 0.7, (r3.uniform(size=m3.V) < 0.02).astype(np.uint8))
ctx = capi.MnavContext(0)
cm.upload(ctx)
bad += dijkstra_leg("costs_invalid_limits_1_to_96_plans", cm, ctx, [1, 1, 5, 40, 96], BUDGET / 2, 3, limits=(1.0, 0.8, 0.55))
bad += dijkstra_leg("costs_invalid_limits_97_to_700_plans", cm, ctx, [97, 300, 700], BUDGET / 2, 1, limits=(1.0, 0.8, 0.55))
ctx.close()
# ---- the inflation wave (InflationLayer::waveCostInflation on the ordered-wave engine): random lethal sets, radii, invalid vertices
t0 = time.perf_counter()
runs = 0
bl = []
refused = []
while time.perf_counter() - t0 < BUDGET / 2:
    N = int(rng.choice([64, 128, 200]))
    case = Case(meshgen.terrain(N, 0.1, int(rng.integers(1000)), amplitude=float(rng.choice([0.3, 0.8]))))
    m = case.mesh
    lethal = np.zeros(m.V, np.uint8)
    kind = int(rng.integers(3))
    if kind == 0:
        _, lethal = case.om.steepness(case.vn, float(rng.choice([0.3, 0.5])))
    elif kind == 1:
        lethal[m.edges[rng.choice(m.E, max(1, m.E // int(rng.choice([100, 400]))), replace=False)].ravel()] = 1
        lethal[rng.choice(m.V, max(1, m.V // 80), replace=False)] = 1
    else:
        lethal[rng.choice(m.V, max(1, m.V // int(rng.choice([30, 300]))), replace=False)] = 1
    inv = None
    if rng.random() < 0.5:
        inv = np.zeros(m.V, np.uint8)
        inv[rng.choice(m.V, m.V // 40, replace=False)] = 1
    radius = float(rng.choice([0.25, 0.4, 0.9, 1.3]))
    cfg = O.InflationCfg.defaults()
    cfg.inflation_radius = radius
    cost, dist, vec = case.om.inflation(lethal, case.edge_dist, cfg, invalid=inv)
    ctx = capi.MnavContext(0)
    ctx.upload_mesh(m.xyz, m.faces, m.edges, case.vn)
    ctx.layer_upload(0, np.zeros(m.V, np.float32), lethal)
    runs += 1
    try:
        ctx.layer_inflation(1, 0, inflation_radius=radius, invalid=inv)
        c, _, d = ctx.layer_download(1, distances=True)
        if not (beq(d, dist) and beq(c, cost)):
            bl.append(dict(leg="inflation", N=N, kind=kind, radius=radius, invalid=inv is not None, dist_bits_differ=int((np.asarray(d).view(np.uint32) != dist.view(np.uint32)).sum())))
    except RuntimeError as e:                                         # loud failure (INTERNAL_ERROR): tied pop times around isolated lethal vertices, DESIGN.md 3.2
        refused.append(dict(N=N, kind=kind, radius=radius, invalid=inv is not None, error=str(e)[:90]))
    ctx.close()
out["legs"]["inflation_wave_random_sources"] = dict(runs=runs, mismatches=len(bl), refused_with_internal_error=len(refused), seconds=round(time.perf_counter() - t0, 1))
out["inflation_refused"] = refused[:10]
bad += bl
# ---- 10M terrain (C4): single plans on the asynchronous engine
if os.environ.get("SOAK_C4", "1") != "0":
    mesh = meshgen.terrain(3163, 0.1, 4)
    om = O.OracleMesh(mesh.xyz, mesh.faces)
    w = meshgen.edge_lengths(mesh)
    ctx = capi.MnavContext(0)
    ctx.upload_mesh(mesh.xyz, mesh.faces, mesh.edges, None)
    zeros = np.zeros(mesh.V, np.float32)
    ctx.upload_costs(zeros, w)
    t0 = time.perf_counter()
    n = cmp = 0
    b4 = []
    while time.perf_counter() - t0 < BUDGET:
        s, t = int(rng.integers(mesh.V)), int(rng.integers(mesh.V))
        off = float(OFFSETS[int(rng.integers(len(OFFSETS)))])
        o = ctx.plan_dijkstra(s, t, goal_dist_offset=off, want_fields=False)
        n += 1
        if o.code not in (0, 54):
            b4.append(dict(leg="c4", seed=s, target=t, off=off, code=int(o.code)))
        elif n % 12 == 1:
            ref = om.dijkstra(w, zeros, s, t, goal_dist_offset=off)
            cmp += 1
            if o.code != ref.code or not np.array_equal(o.path, ref.path):
                b4.append(dict(leg="c4", seed=s, target=t, off=off))
    out["legs"]["c4_async_single_plans_10M"] = dict(plans=n, compared_with_oracle=cmp, mismatches=len(b4), seconds=round(time.perf_counter() - t0, 1))
    bad += b4
    ctx.close()
out["mismatches"] = bad[:20]
out["ok"] = not bad
print(json.dumps(out))
