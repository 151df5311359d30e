"""GPU: CVP parity on the EXACT configuration bench.py's C3 leg times (N=1000 terrain seed 3, amplitude 2 m, Steepness
threshold 0.6 rad, Inflation defaults, avg-combined, edge_cost_factor 1, the whole cost stack built on the device), a
fixed-seed CVP fuzz budget on it (formerly tools/gpu_cvp_fuzz.py), and mnav_vector_at against the oracle's
directionAtPosition (mesh_map.cpp:625-650)."""
import numpy as np
import pytest

from mesh_navigation_amd import meshgen
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def vertex_normals(mesh):
    p = mesh.xyz.astype(np.float64)
    fnrm = np.cross(p[mesh.faces[:, 1]] - p[mesh.faces[:, 0]], p[mesh.faces[:, 2]] - p[mesh.faces[:, 0]])
    fnrm /= np.maximum(np.linalg.norm(fnrm, axis=1, keepdims=True), 1e-30)
    vnrm = np.zeros_like(p)
    for k in range(3):
        for c in range(3):
            vnrm[:, c] += np.bincount(mesh.faces[:, k], weights=fnrm[:, c], minlength=mesh.V)
    vnrm /= np.maximum(np.linalg.norm(vnrm, axis=1, keepdims=True), 1e-30)
    return vnrm.astype(np.float32)


@pytest.fixture(scope="module")
def c3(gpu_ctx_factory):
    mesh = meshgen.terrain(1000, 0.1, 3)
    vn = vertex_normals(mesh)
    ctx = gpu_ctx_factory()
    ctx.upload_mesh(mesh.xyz, mesh.faces, mesh.edges, vn)
    ctx.layer_steepness(0, 0.6)
    ctx.layer_inflation(1, 0)
    ctx.combine_layers([0, 1], [1.0, 1.0], mode="avg", edge_cost_factor=1.0)
    vc, w = ctx.download_costs()
    om = O.OracleMesh(mesh.xyz, mesh.faces)
    first_face = np.full(mesh.V, -1, np.int64)
    fl = mesh.faces.ravel()
    first_face[fl[::-1]] = np.arange(fl.size)[::-1] // 3
    return mesh, vn, ctx, vc, w, om, first_face


def wave_seed(mesh, first_face, v):
    f = int(first_face[v])
    return mesh.xyz[mesh.faces[f]].astype(np.float64).mean(axis=0).astype(np.float32), f


def test_device_cost_stack_equals_the_oracle(c3):
    """layer by layer: Steepness costs / lethal flags, Inflation distances and costs, the avg combination, the edge weights"""
    mesh, vn, ctx, vc, w, om, _ = c3
    ed = om.edge_distances()
    steep, lethal = om.steepness(vn, 0.6)
    d_steep, d_lethal = ctx.layer_download(0)
    assert np.array_equal(d_lethal.astype(bool), lethal.astype(bool))
    assert np.array_equal(d_steep.view(np.uint32), steep.view(np.uint32))
    infl, dist, _ = om.inflation(lethal, ed)
    d_infl, _, d_dist = ctx.layer_download(1, distances=True)
    assert np.array_equal(d_dist.view(np.uint32), dist.view(np.uint32))
    assert np.array_equal(d_infl.view(np.uint32), infl.view(np.uint32))
    comb = O.combine([steep, infl], [1.0, 1.0], "avg")
    assert np.array_equal(vc.view(np.uint32), comb.view(np.uint32))
    assert np.array_equal(w.view(np.uint32), om.edge_weights(ed, comb, 1.0).view(np.uint32))


def test_cvp_plans_of_the_benched_c3_bit_exact(c3):
    """the goals bench.py draws (rng(5) among the traversable vertices), the common robot face: potential, predecessors,
    cutting faces, directions against the sequential oracle; 1e-5 relative is the north_star bar, met bit for bit"""
    mesh, vn, ctx, vc, w, om, first_face = c3
    free = np.nonzero(vc < 0.5)[0]
    N = 1000
    robot = int(free[np.argmin(np.abs(mesh.xyz[free, 0] - 0.9 * N * 0.1) + np.abs(mesh.xyz[free, 1] - 0.9 * N * 0.1))])
    tf = int(first_face[robot])
    goals = np.random.default_rng(5).choice(free, size=160, replace=False)
    for k in (0, 1, 7):
        sp, sf = wave_seed(mesh, first_face, int(goals[k]))
        ref = om.cvp(w, vc, vn, sp, sf, tf)
        out = ctx.plan_cvp(sp, sf, tf)
        assert out.code == ref.code
        fin = np.isfinite(ref.dist)
        assert np.array_equal(np.isfinite(out.dist), fin)
        rel = np.abs(out.dist[fin] - ref.dist[fin]) / np.maximum(ref.dist[fin], 1e-12)
        assert rel.max() <= 1e-5
        assert np.array_equal(out.dist.view(np.uint32), ref.dist.view(np.uint32)) and np.array_equal(out.pred, ref.pred)
        upd = ref.pred != np.arange(mesh.V)
        assert np.array_equal(out.cutface[upd], ref.cutface[upd])
        assert np.array_equal(out.direction[upd].view(np.uint32), ref.direction[upd].view(np.uint32))


def test_cvp_fuzz_fixed_seed_budget(c3):
    """random (goal, robot) pairs and cut-offs: the racy in-place iteration of the device against the sequential oracle"""
    mesh, vn, ctx, vc, w, om, first_face = c3
    free = np.nonzero(vc < 0.5)[0]
    rng = np.random.default_rng(11)
    for k in range(10):
        g, r = rng.choice(free, 2, replace=False)
        sf, tf = int(first_face[g]), int(first_face[r])
        sp = mesh.xyz[mesh.faces[sf]].astype(np.float64).mean(axis=0).astype(np.float32)
        off = float(rng.choice([0.3, 0.3, 2.0, np.inf]))
        o = ctx.plan_cvp(sp, sf, tf, goal_dist_offset=off, want_fields=True, want_vecmap=False)
        ref = om.cvp(w, vc, vn, sp, sf, tf, goal_dist_offset=off)
        assert o.code == ref.code, k
        assert np.array_equal(o.dist.view(np.uint32), ref.dist.view(np.uint32)), k
        assert np.array_equal(o.pred, ref.pred), k


def blend(vecmap, has, vs, bary):
    """directionAtPosition (mesh_map.cpp:625-650): float32 sum of the vertices' vectors scaled by the barycentric weights"""
    acc = np.zeros(3, np.float32)
    any_ = False
    for k in range(3):
        if has[vs[k]]:
            any_ = True
            acc = (acc + vecmap[vs[k]] * np.float32(bary[k])).astype(np.float32)
    return acc if any_ and np.isfinite(acc).all() else None


def test_vector_at_matches_direction_at_position(c3):
    mesh, vn, ctx, vc, w, om, first_face = c3
    free = np.nonzero(vc < 0.5)[0]
    rng = np.random.default_rng(3)
    g, r = rng.choice(free, 2, replace=False)
    ctx.set_resident_outputs(True)
    try:
        # Dijkstra: the device vector map is bit-equal to the oracle's, so is the sample
        out = ctx.plan_dijkstra(int(g), int(r), want_fields=True, want_vecmap=True)
        ref = om.dijkstra(w, vc, int(g), int(r))
        vm = om.dijkstra_vector_map(ref.pred)
        has = ref.pred != np.arange(mesh.V)
        faces = rng.choice(mesh.F, 24, replace=False)
        n_some = 0
        for f in faces:
            vs = mesh.faces[f]
            b = rng.dirichlet([1, 1, 1]).astype(np.float32)
            got = ctx.vector_at(vs, b)
            want = blend(vm, has, vs, b)
            assert (got is None) == (want is None)
            if want is not None:
                n_some += 1
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        # the path runs through reached vertices: samples along it must exist
        for v in out.path[:: max(1, len(out.path) // 8)]:
            f = int(first_face[v])
            assert ctx.vector_at(mesh.faces[f], np.array([0.2, 0.3, 0.5], np.float32)) is not None
        # CVP: the device vector map is within 2e-7 of the oracle's (cosf / sinf), and so is the sample
        sp, sf = wave_seed(mesh, first_face, int(g))
        tf = int(first_face[r])
        refc = om.cvp(w, vc, vn, sp, sf, tf)
        ctx.plan_cvp(sp, sf, tf, want_fields=False, want_vecmap=False)
        for f in faces[:12]:
            vs = mesh.faces[f]
            b = np.array([0.25, 0.25, 0.5], np.float32)
            got = ctx.vector_at(vs, b)
            want = blend(refc.vecmap, refc.has_vec.astype(bool), vs, b)
            assert (got is None) == (want is None)
            if want is not None:
                assert np.abs(got - want).max() <= 4e-7
    finally:
        ctx.set_resident_outputs(False)


# ---- the BATCHES bench.py times on this configuration (configs.C3: 128 plans = 3 plan groups on their own streams, 512 plans =
# ---- 4 groups; k_cvp_ctl + k_step_wide + k_step_repair per step).  The loop they stand for: cvp_mesh_planner.cpp:747-886.

def _bench_draw(c3):
    """bench.py leg_c3's draw: goals among the vertices of the largest traversable component with cost < 0.5, rng(5):
    160 goals (the batch of 128 = goals[16:144]), then 512 more; the common robot face"""
    import scipy.sparse as sp
    import scipy.sparse.csgraph as cg
    mesh, vn, ctx, vc, w, om, first_face = c3
    N = 1000
    fr = vc < 1.0
    e = mesh.edges
    ok = fr[e[:, 0]] & fr[e[:, 1]]
    g = sp.coo_matrix((np.ones(int(ok.sum()), np.int8), (e[ok, 0], e[ok, 1])), shape=(mesh.V, mesh.V)).tocsr()
    _, lab = cg.connected_components(g, directed=False)
    big = int(np.argmax(np.bincount(lab[fr])))
    free = np.nonzero((lab == big) & fr)[0]
    robot = int(free[np.argmin(np.abs(mesh.xyz[free, 0] - 0.9 * N * 0.1) + np.abs(mesh.xyz[free, 1] - 0.9 * N * 0.1))])
    free = free[vc[free] < 0.5]
    rng = np.random.default_rng(5)
    goals = rng.choice(free, size=160, replace=False)
    goals512 = rng.choice(free, size=512, replace=False)
    return goals, goals512, int(first_face[robot])


def _batch_args(c3, verts, tf):
    mesh, first_face = c3[0], c3[6]
    seeds = [wave_seed(mesh, first_face, int(v)) for v in verts]
    return np.stack([s[0] for s in seeds]), np.array([s[1] for s in seeds], np.uint32), np.full(len(seeds), tf, np.uint32)


def _slot_fields(ctx, slot):
    return {k: ctx.download_output(k, slot) for k in ("dist", "pred", "direction", "cutface", "vecmap")}


def _assert_slot_is_the_oracle(c3, ctx, sps, sfs, tf, slot, code):
    mesh, vn, _, vc, w, om, _ = c3
    ref = om.cvp(w, vc, vn, sps[slot], int(sfs[slot]), tf)
    assert code == ref.code, slot
    o = _slot_fields(ctx, slot)
    assert np.array_equal(o["dist"].view(np.uint32), ref.dist.view(np.uint32)), slot
    assert np.array_equal(o["pred"], ref.pred), slot
    upd = ref.pred != np.arange(mesh.V)
    assert np.array_equal(o["cutface"][upd], ref.cutface[upd]), slot
    assert np.array_equal(o["direction"][upd].view(np.uint32), ref.direction[upd].view(np.uint32)), slot
    hv = ref.has_vec.astype(bool)
    assert np.array_equal(o["vecmap"][hv].view(np.uint32), ref.vecmap[hv].view(np.uint32)), slot
    assert not o["vecmap"][~hv].any(), slot


@pytest.mark.parametrize("nb,slots", [(128, (0, 21, 43, 64, 86, 127)), (512, (0, 101, 129, 254, 257, 383, 385, 511))])
def test_cvp_batches_of_the_benched_c3_bit_exact(c3, nb, slots):
    """the two batches of configs.C3 (wide step kernel, multi-group stepping as branches of one captured graph): at least two
    plans of EVERY plan group -- potential, predecessors, cutting faces, directions and vector map against the sequential oracle"""
    ctx = c3[2]
    goals, goals512, tf = _bench_draw(c3)
    sps, sfs, tfs = _batch_args(c3, goals[16:16 + nb] if nb <= 128 else goals512[:nb], tf)
    ctx.set_resident_outputs(True)
    try:
        r = ctx.plan_cvp_batch(sps, sfs, tfs)
        assert r["rc"] == 0 and (r["codes"] == 0).all()                 # (bench.py checks exactly this)
        assert r["stats"]["steps"] > 500
        for s in slots:
            _assert_slot_is_the_oracle(c3, ctx, sps, sfs, tf, s, int(r["codes"][s]))
    finally:
        ctx.set_resident_outputs(False)


def test_cvp_batch_one_group_equals_the_grouped_default(c3):
    """option cvp_groups = 1 (every plan stepped on one stream) against the default three groups: the same bits in every field of
    a sample of slots across the batch"""
    ctx = c3[2]
    goals, _, tf = _bench_draw(c3)
    sps, sfs, tfs = _batch_args(c3, goals[16:144], tf)
    slots = range(0, 128, 9)
    ctx.set_resident_outputs(True)
    try:
        ctx.set_option("cvp_groups", None)
        a = ctx.plan_cvp_batch(sps, sfs, tfs)
        fa = {s: _slot_fields(ctx, s) for s in slots}
        ctx.set_option("cvp_groups", 1)
        b = ctx.plan_cvp_batch(sps, sfs, tfs)
        ctx.set_option("cvp_groups", None)
        assert np.array_equal(a["codes"], b["codes"])
        for s in slots:
            fb = _slot_fields(ctx, s)
            for k, v in fa[s].items():
                assert np.array_equal(v.view(np.uint32), fb[k].view(np.uint32)), (s, k)
    finally:
        ctx.set_resident_outputs(False)
