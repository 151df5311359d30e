"""GPU parity tests proper: the HIP path, called through the C ABI (ctypes), against the CPU oracle
on the same seeded inputs and against the committed golden fixtures.  Bit-exact for the Dijkstra
potential / predecessor / vertex path; 1e-5 relative for the CVP potential (BASELINE.json north_star)."""
import hashlib
import os
import threading

import numpy as np
import pytest

from mesh_navigation_amd import capi, meshgen
from oracle import oracle as O
from tests.common import Case, layered_costs, terrain_case

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "planner_golden.npz"))
CVP_RTOL = 1e-5


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def assert_dijkstra_equal(out, ref, case=None):
    assert out.code == ref.code
    assert np.array_equal(out.dist.view(np.uint32), ref.dist.view(np.uint32)), "potential must be bit-exact"
    assert np.array_equal(out.pred, ref.pred)
    assert np.array_equal(out.path, ref.path), "vertex-index path must be identical"


def assert_cvp_close(out, ref):
    """The north_star bar for the CVP potential is 1e-5 relative; since the pop order is reproduced exactly
    (cascade-forest pop keys) the device matches the oracle bit for bit -- potential, predecessors, cutting
    faces and directions -- and that is what is asserted."""
    assert out.code == ref.code
    fin = np.isfinite(ref.dist)
    assert np.array_equal(np.isfinite(out.dist), fin), "reached sets differ"
    rel = np.abs(out.dist[fin] - ref.dist[fin]) / np.maximum(ref.dist[fin], 1e-12)
    assert (rel.max() if fin.any() else 0.0) <= CVP_RTOL, f"CVP potential: max rel err {rel.max()}"
    assert np.array_equal(out.dist.view(np.uint32), ref.dist.view(np.uint32)), "CVP potential must be bit-exact"
    assert np.array_equal(out.pred, ref.pred)
    upd = ref.pred != np.arange(len(ref.pred))
    if getattr(out, "cutface", None) is not None and getattr(ref, "cutface", None) is not None:
        assert np.array_equal(out.cutface[upd], ref.cutface[upd])
        assert np.array_equal(out.direction[upd].view(np.uint32), ref.direction[upd].view(np.uint32))
    return float(rel.max()) if fin.any() else 0.0


@pytest.fixture(scope="module")
def c1(gpu_ctx_factory):
    case = terrain_case(224, 1)
    ctx = gpu_ctx_factory()
    case.upload(ctx)
    return case, ctx


@pytest.mark.parametrize("engine", ["tiled", "band", "tile_batch", "async"])
def test_dijkstra_c1_bit_exact_and_golden(c1, engine):
    case, ctx = c1
    ctx.set_dijkstra_engine(engine)
    s, t = (int(x) for x in GOLD["c1_seed_target"])
    ref = case.om.dijkstra(case.weights, case.costs, s, t)
    out = ctx.plan_dijkstra(s, t, want_fields=True, want_vecmap=True)
    assert_dijkstra_equal(out, ref)
    assert out.stats["goal_dist"] == ref.stats["goal_dist"]
    assert np.array_equal(out.path, GOLD["c1_dij_path"])
    assert sha(out.dist) == str(GOLD["c1_dij_dist_sha"]) and sha(out.pred) == str(GOLD["c1_dij_pred_sha"])
    vm = case.om.dijkstra_vector_map(ref.pred)                       # computeVectorMap :189-209
    assert np.array_equal(out.vecmap.view(np.uint32), vm.view(np.uint32))
    ctx.set_dijkstra_engine("auto")


def test_cvp_c1_and_golden(c1):
    case, ctx = c1
    sf, tf = (int(x) for x in GOLD["c1_cvp_faces"])
    sp, tp = GOLD["c1_cvp_seed_pos"], GOLD["c1_cvp_target_pos"]
    ref = case.om.cvp(case.weights, case.costs, case.vn, sp, sf, tf)
    out = ctx.plan_cvp(sp, sf, tf)
    assert_cvp_close(out, ref)
    step = max(1, case.mesh.V // 512)
    gs = GOLD["c1_cvp_dist_sample"]
    fin = np.isfinite(gs)
    assert np.allclose(out.dist[::step][fin], gs[fin], rtol=CVP_RTOL, atol=0)
    # the fixture holds the outputs of the reference's own code (oracle/_ref): same bits on the device
    assert sha(out.dist) == str(GOLD["c1_cvp_dist_sha"]) and sha(out.pred) == str(GOLD["c1_cvp_pred_sha"])
    assert sha(out.cutface) == str(GOLD["c1_cvp_cutface_sha"]) and sha(out.direction) == str(GOLD["c1_cvp_direction_sha"])
    upd = ref.pred != np.arange(case.mesh.V)
    assert (out.cutface[upd] != ref.cutface[upd]).mean() < 1e-3
    assert np.abs(out.direction[upd] - ref.direction[upd]).max() < 1e-4 or \
        (np.abs(out.direction[upd] - ref.direction[upd]) > 1e-4).mean() < 1e-3
    same = (out.pred == ref.pred) & (np.abs(out.direction - ref.direction) < 1e-6)
    assert np.array_equal(out.vecmap[same].view(np.uint32), ref.vecmap[same].view(np.uint32))   # computeVectorMap :204-239, the host libm's sin / cos bits
    # the host back-tracking (cvp :920-951) follows the same path on the device vector field
    hv = (np.abs(out.vecmap).sum(axis=1) > 0).astype(np.uint8)
    code_d, pos_d, face_d = case.om.cvp_backtrack(out.vecmap, hv, sp, sf, tp, tf)
    code_r, pos_r, face_r = case.om.cvp_backtrack(ref.vecmap, ref.has_vec, sp, sf, tp, tf)
    assert code_d == code_r == 0 and np.array_equal(face_d, face_r)
    assert np.array_equal(pos_d.view(np.uint32), pos_r.view(np.uint32))


@pytest.mark.parametrize("engine", ["tiled", "band", "tile_batch", "async"])
@pytest.mark.parametrize("offset", [0.0, 0.01, 0.3, 5.0, float("inf")])
def test_dijkstra_goal_dist_offsets(c1, engine, offset):
    case, ctx = c1
    ctx.set_dijkstra_engine(engine)
    m = case.mesh
    s, t = m.vertex_at(0.5, 0.5), m.vertex_at(0.7, 0.62)
    ref = case.om.dijkstra(case.weights, case.costs, s, t, goal_dist_offset=offset)
    out = ctx.plan_dijkstra(s, t, goal_dist_offset=offset)
    assert_dijkstra_equal(out, ref)
    ctx.set_dijkstra_engine("auto")


def test_return_codes(c1):
    case, ctx = c1
    V = case.mesh.V
    out = ctx.plan_dijkstra(7, 7)                                    # dijkstra :252-255
    assert out.code == capi.SUCCESS and len(out.path) == 0 and np.isinf(out.dist).all()
    assert ctx.plan_dijkstra(V + 5, 3).code == capi.INVALID_START    # stand-in for :240-241
    assert ctx.plan_dijkstra(3, V + 5).code == capi.INVALID_GOAL     # :242-243
    assert ctx.plan_cvp(np.zeros(3, np.float32), case.mesh.F + 1, 0).code == capi.INVALID_START   # cvp :681-685
    assert ctx.plan_cvp(np.zeros(3, np.float32), 0, case.mesh.F + 1).code == capi.INVALID_GOAL    # cvp :686-690


def test_cost_limit_invalid_unreachable(gpu_ctx_factory):
    mesh = meshgen.terrain(96, 0.1, 13)
    rng = np.random.default_rng(3)
    costs = rng.uniform(0, 1.2, mesh.V).astype(np.float32)
    invalid = (rng.uniform(size=mesh.V) < 0.02).astype(np.uint8)
    s, t = mesh.vertex_at(0.1, 0.1), mesh.vertex_at(0.9, 0.9)
    invalid[[s, t]] = 0
    costs[[s, t]] = 0
    case = Case(mesh, costs, 1.0, invalid)
    ctx = gpu_ctx_factory()
    case.upload(ctx)
    for engine in ("tiled", "band", "tile_batch", "async"):
        ctx.set_dijkstra_engine(engine)
        for lim in (1.0, 0.6):
            ref = case.om.dijkstra(case.weights, case.costs, s, t, cost_limit=lim, invalid=case.invalid)
            out = ctx.plan_dijkstra(s, t, cost_limit=lim)
            assert_dijkstra_equal(out, ref)
    ctx.set_dijkstra_engine("auto")
    sp = mesh.xyz[s] + np.array([0.03, 0.02, 0], np.float32)
    tp = mesh.xyz[t] + np.array([0.03, 0.02, 0], np.float32)
    sf, _ = case.om.containing_face(sp)
    tf, _ = case.om.containing_face(tp)
    # CVP on these adversarial weights (random per-vertex costs up to 1.2 inflate single edges by up
    # to 2.2x, so most triangles violate the triangle inequality): updates undercut the pop front in
    # nested cascades, which the pop keys order exactly (mnav_eval.h::PopKey)
    refc = case.om.cvp(case.weights, case.costs, case.vn, sp, sf, tf, invalid=case.invalid)
    assert_cvp_close(ctx.plan_cvp(sp, sf, tf), refc)
    for off in (0.0, float("inf")):
        refc = case.om.cvp(case.weights, case.costs, case.vn, sp, sf, tf, invalid=case.invalid, goal_dist_offset=off)
        assert_cvp_close(ctx.plan_cvp(sp, sf, tf, goal_dist_offset=off), refc)
    # the same geometry with moderate random costs (edges inflated by up to 1.6x) is exact to tolerance
    costs_mod = (case.costs * 0.5).astype(np.float32)
    case_mod = Case(mesh, costs_mod, 1.0, invalid)
    case_mod.upload(ctx)
    refm = case_mod.om.cvp(case_mod.weights, case_mod.costs, case_mod.vn, sp, sf, tf, invalid=case_mod.invalid)
    assert_cvp_close(ctx.plan_cvp(sp, sf, tf), refm)
    case.upload(ctx)
    # a wall of over-limit vertices: NO_PATH_FOUND, wall reached but not expanded (dijkstra :302,:358)
    costs2 = np.zeros(mesh.V, np.float32)
    costs2[48 * 96: 49 * 96] = 5.0
    case2 = Case(mesh, costs2, 0.0)
    case2.upload(ctx)
    ref2 = case2.om.dijkstra(case2.weights, case2.costs, s, t)
    out2 = ctx.plan_dijkstra(s, t)
    assert ref2.code == O.NO_PATH_FOUND
    assert_dijkstra_equal(out2, ref2)
    refc2 = case2.om.cvp(case2.weights, case2.costs, case2.vn, sp, sf, tf)
    outc2 = ctx.plan_cvp(sp, sf, tf)
    assert refc2.code == outc2.code == O.NO_PATH_FOUND               # cvp :912-918


def test_layered_costs_config3_shape(gpu_ctx_factory):
    """BASELINE config 3 at test size: Steepness + Inflation (Avg and Max), edge_cost_factor 1;
    edge weights derived on the device (mesh_map.cpp:517-561) must be bit-identical."""
    base = Case(meshgen.terrain(224, 0.1, 3, amplitude=0.8))
    ctx = gpu_ctx_factory()
    for mode in ("avg", "max"):
        costs, parts = layered_costs(base, mode)
        case = Case(base.mesh, costs, 1.0)
        ctx.upload_mesh(case.mesh.xyz, case.mesh.faces, case.mesh.edges, case.vn)
        w_dev = ctx.compute_edge_weights(case.costs, case.edge_dist, 1.0)
        assert np.array_equal(w_dev.view(np.uint32), case.weights.view(np.uint32))
        # the combination layer itself on the device (combination_layer.cpp:44-85 / :185-248), non-trivial weights too
        for wts in ([1.0, 1.0], [0.7, 1.9]):
            vc_dev, w2 = ctx.combine_costs([parts["steepness"], parts["inflation"]], wts, case.edge_dist, 1.0, mode)
            vc_ref = O.combine([parts["steepness"], parts["inflation"]], wts, mode)
            assert np.array_equal(vc_dev.view(np.uint32), vc_ref.view(np.uint32))
            w_ref = case.om.edge_weights(case.edge_dist, vc_ref, 1.0)
            assert np.array_equal(w2.view(np.uint32), w_ref.view(np.uint32))
        vc_dev, w2 = ctx.combine_costs([parts["steepness"], parts["inflation"]], [1.0, 1.0], case.edge_dist, 1.0, mode)
        assert np.array_equal(vc_dev.view(np.uint32), costs.view(np.uint32)) and np.array_equal(w2.view(np.uint32), case.weights.view(np.uint32))
        m = case.mesh
        free = np.where(costs < 0.5)[0]
        def near(fi, fj):
            v = m.vertex_at(fi, fj)
            return int(free[((m.xyz[free, :2] - m.xyz[v, :2]) ** 2).sum(1).argmin()])
        s, t = near(0.1, 0.1), near(0.9, 0.9)
        ref = case.om.dijkstra(case.weights, case.costs, s, t)
        assert_dijkstra_equal(ctx.plan_dijkstra(s, t), ref)
        sp = m.xyz[s] + np.array([0.02, 0.01, 0], np.float32)
        tp = m.xyz[t] + np.array([0.02, 0.01, 0], np.float32)
        sf, _ = case.om.containing_face(sp)
        tf, _ = case.om.containing_face(tp)
        for off in (0.3, float("inf")):
            refc = case.om.cvp(case.weights, case.costs, case.vn, sp, sf, tf, goal_dist_offset=off)
            assert_cvp_close(ctx.plan_cvp(sp, sf, tf, goal_dist_offset=off), refc)


def test_golden_layered_fixture(gpu_ctx_factory):
    base = Case(meshgen.terrain(40, 0.1, 3, amplitude=0.8))
    case = Case(base.mesh, GOLD["g2_costs"], 1.0)
    assert np.array_equal(case.weights.view(np.uint32), GOLD["g2_weights"].view(np.uint32))
    ctx = gpu_ctx_factory()
    case.upload(ctx)
    s, t = (int(x) for x in GOLD["g2_seed_target"])
    out = ctx.plan_dijkstra(s, t)
    assert np.array_equal(out.dist.view(np.uint32), GOLD["g2_dij_dist"].view(np.uint32))
    assert np.array_equal(out.pred, GOLD["g2_dij_pred"]) and np.array_equal(out.path, GOLD["g2_dij_path"])
    sf, tf = (int(x) for x in GOLD["g2_cvp_faces"])
    outc = ctx.plan_cvp(GOLD["g2_cvp_seed_pos"], sf, tf)
    g = GOLD["g2_cvp_dist"]
    fin = np.isfinite(g)
    assert np.array_equal(np.isfinite(outc.dist), fin)
    assert (np.abs(outc.dist[fin] - g[fin]) / np.maximum(g[fin], 1e-12)).max() <= CVP_RTOL       # the north_star bar ...
    assert np.array_equal(outc.dist.view(np.uint32), g.view(np.uint32))                            # ... and what we hold: the reference's bits
    assert np.array_equal(outc.pred, GOLD["g2_cvp_pred"]) and np.array_equal(outc.cutface, GOLD["g2_cvp_cutface"])
    assert np.array_equal(outc.direction.view(np.uint32), GOLD["g2_cvp_direction"].view(np.uint32))


@pytest.mark.parametrize("which", ["g3", "g4"])
def test_golden_order_sensitive_fixtures(gpu_ctx_factory, which):
    """The device against the committed fixtures of the inputs where the pop order decides (punched terrain,
    adversarial costs) -- no oracle call in this test: the arrays / hashes travel with the repository."""
    from tests.test_golden import RAGGED, ragged_case
    case = ragged_case(which)
    ctx = gpu_ctx_factory()
    case.upload(ctx)
    s, t = (int(x) for x in RAGGED[which + "_seed_target"])
    for engine in ("tiled", "band", "tile_batch", "async"):
        ctx.set_dijkstra_engine(engine)
        out = ctx.plan_dijkstra(s, t)
        assert out.code == int(RAGGED[which + "_dij_code"][0]) and np.array_equal(out.path, RAGGED[which + "_dij_path"])
        assert sha(out.dist) == str(RAGGED[which + "_dij_dist_sha"]) and sha(out.pred) == str(RAGGED[which + "_dij_pred_sha"])
    ctx.set_dijkstra_engine("auto")
    sf, tf = (int(x) for x in RAGGED[which + "_cvp_faces"])
    outc = ctx.plan_cvp(RAGGED[which + "_cvp_seed_pos"], sf, tf, want_vecmap=False)
    assert outc.code == int(RAGGED[which + "_cvp_code"][0])
    assert sha(outc.dist) == str(RAGGED[which + "_cvp_dist_sha"]) and sha(outc.pred) == str(RAGGED[which + "_cvp_pred_sha"])
    assert sha(outc.direction) == str(RAGGED[which + "_cvp_direction_sha"])
    # the reference never clears cutting_faces_: it holds a face exactly where this wave set a vertex
    assert sha(outc.cutface) == str(RAGGED[which + "_cvp_cutface_sha"])
    if which == "g3":
        assert np.array_equal(outc.dist.view(np.uint32), RAGGED["g3_cvp_dist"].view(np.uint32))


def test_batch_equals_single_plans(c1):
    case, ctx = c1
    m = case.mesh
    rng = np.random.default_rng(5)
    goals = rng.choice(m.V, size=12, replace=False).astype(np.uint32)
    goals[3] = goals[0]                                               # duplicate goal
    targets = np.full(12, m.vertex_at(0.9, 0.9), np.uint32)
    goals[5] = targets[5]                                             # seed == target inside a batch
    for engine in ("tiled", "tile_batch", "async"):
        ctx.set_dijkstra_engine(engine)
        b = ctx.plan_dijkstra_batch(goals, targets, want_fields=True)
        check_batch(case, b, goals, targets)
    ctx.set_dijkstra_engine("auto")


def check_batch(case, b, goals, targets):
    for k in range(12):
        ref = case.om.dijkstra(case.weights, case.costs, int(goals[k]), int(targets[k]))
        assert b["codes"][k] == ref.code
        assert np.array_equal(b["paths"][k], ref.path)
        assert np.array_equal(b["dist"][k].view(np.uint32), ref.dist.view(np.uint32))
        assert np.array_equal(b["pred"][k], ref.pred)
    assert b["stats"]["n_plans"] == 11                                # the seed==target plan never reaches the device


def test_full_field_properties(c1):
    """Size-independent properties of the potential (used again at 1M / 10M vertices): it is the
    fixed point of the edge relaxation, zero at the seed, and walking predecessors strictly descends."""
    case, ctx = c1
    m = case.mesh
    s, t = m.vertex_at(0.3, 0.6), m.vertex_at(0.9, 0.1)
    out = ctx.plan_dijkstra(s, t, goal_dist_offset=float("inf"))
    d, e, w = out.dist, m.edges, case.weights
    assert d[s] == 0 and np.isfinite(d).all()
    assert (d[e[:, 0]] <= d[e[:, 1]] + w).all() and (d[e[:, 1]] <= d[e[:, 0]] + w).all()
    v = np.arange(m.V)
    nz = v != s
    assert (d[out.pred[nz]] < d[nz]).all()
    assert out.path[0] == s and out.pred[t] == out.path[-1]


def test_cancel_flag_semantics(c1):
    case, ctx = c1
    m = case.mesh
    s, t = m.vertex_at(0.1, 0.1), m.vertex_at(0.9, 0.9)
    ctx.cancel()                                  # a stale cancel is cleared when a plan starts (dijkstra :238)
    assert ctx.plan_dijkstra(s, t, want_fields=False).code == capi.SUCCESS
    codes = []
    th = threading.Thread(target=lambda: codes.append(ctx.plan_cvp(m.xyz[s] + np.float32(0.02), *(case.om.containing_face(m.xyz[s] + np.float32(0.02))[0],) * 2, goal_dist_offset=float("inf"), want_fields=False, want_vecmap=False).code))
    th.start()
    ctx.cancel()                                  # racing a running plan: CANCELED (51) or, if it already finished, SUCCESS
    th.join(timeout=60)
    assert not th.is_alive() and codes and codes[0] in (capi.SUCCESS, capi.CANCELED)
    assert ctx.plan_dijkstra(s, t, want_fields=False).code == capi.SUCCESS


def test_stats_and_algorithmic_bytes(c1):
    case, ctx = c1
    m = case.mesh
    out = ctx.plan_dijkstra(m.vertex_at(0.1, 0.1), m.vertex_at(0.9, 0.9), goal_dist_offset=float("inf"), want_fields=False)
    st = out.stats
    assert st["settled"] == m.V and st["n_plans"] == 1 and st["steps"] > 0 and st["ms_propagation"] > 0
    assert st["algorithmic_bytes"] == 24 * m.V + 24 * m.E


def test_cvp_batch_equals_single_plans(c1):
    case, ctx = c1
    m = case.mesh
    rng = np.random.default_rng(9)
    verts = rng.choice(m.V, size=5, replace=False)
    off = np.array([0.02, 0.015, 0.0], np.float32)
    sps = np.stack([m.xyz[v] + off for v in verts]).astype(np.float32)
    tp = m.xyz[m.vertex_at(0.9, 0.9)] + off
    sfs = np.array([case.om.containing_face(p)[0] for p in sps], np.uint32)
    tf, _ = case.om.containing_face(tp)
    tfs = np.full(5, tf, np.uint32)
    tfs[2] = m.F + 7                                                    # one invalid goal face inside the batch
    b = ctx.plan_cvp_batch(sps, sfs, tfs, want_fields=True, want_vecmap=True)
    assert b["codes"][2] == capi.INVALID_GOAL
    for k in (0, 1, 3, 4):
        ref = case.om.cvp(case.weights, case.costs, case.vn, sps[k], int(sfs[k]), int(tfs[k]))
        assert b["codes"][k] == ref.code
        fin = np.isfinite(ref.dist)
        assert np.array_equal(np.isfinite(b["dist"][k]), fin)
        rel = np.abs(b["dist"][k][fin] - ref.dist[fin]) / np.maximum(ref.dist[fin], 1e-12)
        assert rel.max() <= CVP_RTOL
        assert np.array_equal(b["dist"][k].view(np.uint32), ref.dist.view(np.uint32))
        single = ctx.plan_cvp(sps[k], int(sfs[k]), int(tfs[k]))
        assert np.array_equal(single.dist.view(np.uint32), b["dist"][k].view(np.uint32))
        assert np.array_equal(single.vecmap.view(np.uint32), b["vecmap"][k].view(np.uint32))
