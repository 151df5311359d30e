"""CPU check of the DEVICE SCHEDULE: the band/gather rules and band controller the HIP kernels run
(mesh_navigation_amd/csrc/mnav_eval.h), executed serially by oracle/schedule_model.cpp, must
reproduce the sequential priority-queue planners of the oracle on the same inputs."""
import numpy as np
import pytest

from mesh_navigation_amd import meshgen
from oracle import oracle as O
from tests.common import Case, layered_costs


def run_dijkstra(case, seed, target, **kw):
    off = kw.pop("offset", 0.3)
    lim = kw.pop("cost_limit", 1.0)
    ref = case.om.dijkstra(case.weights, case.costs, seed, target, goal_dist_offset=off, cost_limit=lim,
                           invalid=case.invalid)
    mod = O.schedule_model(0, case.mesh.faces, case.mesh.edges, case.weights, case.costs, [seed], [0.0], 0, [target],
                           offset=off, cost_limit=lim, invalid=case.invalid, **kw)
    return ref, mod


def run_cvp(case, sp, tp, **kw):
    off = kw.pop("offset", 0.3)
    lim = kw.pop("cost_limit", 1.0)
    sf, _ = case.om.containing_face(sp)
    tf, _ = case.om.containing_face(tp)
    ref = case.om.cvp(case.weights, case.costs, case.vn, sp, sf, tf, goal_dist_offset=off, cost_limit=lim,
                      invalid=case.invalid)
    sv = case.mesh.faces[sf]
    mod = O.schedule_model(1, case.mesh.faces, case.mesh.edges, case.weights, case.costs, sv, ref.dist[sv], sf,
                           case.mesh.faces[tf], offset=off, cost_limit=lim, invalid=case.invalid, **kw)
    if mod["code"] == 0:      # model of k_cvp_verify: the converged state is a fixed point, no walk bound was hit on it
        assert mod["verify_bad"] == 0 and mod["verify_flags"] == 0, (mod["verify_bad"], mod["verify_flags"])
    return ref, mod


@pytest.mark.parametrize("delta", [0.07, 0.3, 2.0])
@pytest.mark.parametrize("order", [0, 1, 2])
def test_dijkstra_schedule_bit_exact(delta, order):
    case = Case(meshgen.terrain(64, 0.1, 11))
    m = case.mesh
    ref, mod = run_dijkstra(case, m.vertex_at(0.2, 0.1), m.vertex_at(0.8, 0.9), delta=delta, order=order)
    assert np.array_equal(mod["dist"].view(np.uint32), ref.dist.view(np.uint32))
    assert np.array_equal(mod["pred"], ref.pred)
    assert mod["goal_dist"] == ref.stats["goal_dist"]


@pytest.mark.parametrize("offset", [0.0, 0.01, 0.3, 5.0, np.inf])
def test_dijkstra_goal_dist_offsets(offset):
    """goal_dist arming + the post-arming repair sweep reproduce the cut-off of dijkstra :293-300."""
    case = Case(meshgen.terrain(48, 0.1, 12))
    m = case.mesh
    ref, mod = run_dijkstra(case, m.vertex_at(0.5, 0.5), m.vertex_at(0.7, 0.6), offset=offset, delta=0.4)
    assert np.array_equal(mod["dist"].view(np.uint32), ref.dist.view(np.uint32))
    assert np.array_equal(mod["pred"], ref.pred)


def test_dijkstra_cost_limit_invalid_and_unreachable():
    mesh = meshgen.terrain(40, 0.1, 13)
    rng = np.random.default_rng(3)
    costs = rng.uniform(0, 1.2, mesh.V).astype(np.float32)          # ~1/6 of the vertices over the limit
    invalid = (rng.uniform(size=mesh.V) < 0.02).astype(np.uint8)
    case = Case(mesh, costs, 1.0, invalid)
    seed, target = mesh.vertex_at(0.1, 0.1), mesh.vertex_at(0.9, 0.9)
    case.invalid[[seed, target]] = 0
    case.costs[[seed, target]] = 0
    ref, mod = run_dijkstra(case, seed, target, delta=0.3)
    assert np.array_equal(mod["dist"].view(np.uint32), ref.dist.view(np.uint32))
    assert np.array_equal(mod["pred"], ref.pred)
    # wall -> unreachable target
    costs2 = np.zeros(mesh.V, np.float32)
    costs2[20 * 40: 21 * 40] = 5.0
    case2 = Case(mesh, costs2, 0.0)
    ref2, mod2 = run_dijkstra(case2, seed, target, delta=0.3)
    assert ref2.code == O.NO_PATH_FOUND and mod2["pred"][target] == target
    assert np.array_equal(mod2["dist"].view(np.uint32), ref2.dist.view(np.uint32))


def test_dijkstra_unit_grid_ties():
    """Tie stress (SURVEY.md §8d C2 variant): un-jittered grid, all weights exactly representable.
    Distances are schedule independent; predecessors follow the documented (dist[u], u) rule: the
    neighbour that pops first under the (value, vertex id) heap order wins (strict '<' at :332)."""
    m = meshgen.flat_grid(24, 1.0)
    case = Case(m)
    ones = np.ones(m.E, np.float32)
    ref = case.om.dijkstra(ones, case.costs, 0, m.V - 1, goal_dist_offset=np.inf)
    mod = O.schedule_model(0, m.faces, m.edges, ones, case.costs, [0], [0.0], 0, [m.V - 1], offset=np.inf, delta=1.5)
    assert np.array_equal(mod["dist"], ref.dist)
    rule = case.om.dijkstra_pred_rule(ones, case.costs, 0, np.inf, 1.0, ref.dist)
    assert np.array_equal(mod["pred"], rule)
    # with the heap tie rule fixed to (value, vertex id) the queue-driven loop picks the same tree
    assert np.array_equal(ref.pred, rule)
    # both predecessor fields are shortest-path trees of the same potential
    for pred in (mod["pred"], ref.pred):
        v = m.V - 1
        hops = 0
        while v != 0:
            assert ref.dist[pred[v]] + 1.0 == ref.dist[v]
            v = pred[v]; hops += 1
        assert hops == int(ref.dist[m.V - 1])


@pytest.mark.parametrize("delta", [0.05, 0.3, 1.5])
@pytest.mark.parametrize("order", [0, 2])
def test_cvp_schedule_matches_oracle(delta, order):
    case = Case(meshgen.terrain(56, 0.1, 14))
    m = case.mesh
    sp = m.xyz[m.vertex_at(0.2, 0.2)] + np.array([0.03, 0.02, 0], np.float32)
    tp = m.xyz[m.vertex_at(0.85, 0.8)] + np.array([0.01, 0.04, 0], np.float32)
    ref, mod = run_cvp(case, sp, tp, delta=delta, order=order)
    assert np.array_equal(mod["dist"].view(np.uint32), ref.dist.view(np.uint32))
    assert np.array_equal(mod["pred"], ref.pred)
    upd = ref.pred != np.arange(m.V)
    assert np.array_equal(mod["cutface"][upd], ref.cutface[upd])
    assert np.array_equal(mod["direction"][upd], ref.direction[upd])
    assert mod["goal_dist"] == ref.stats["goal_dist"]


def test_cvp_layered_costs_config3():
    """BASELINE config 3 shape at test size: Steepness + Inflation costs (Avg combination),
    edge_cost_factor 1 -> cost-inflated, partly obtuse 'weighted' triangles and blocked vertices."""
    base = Case(meshgen.terrain(72, 0.1, 3, amplitude=0.8))
    costs, parts = layered_costs(base, "avg")
    assert 0.0 < parts["lethal"].mean() < 0.2
    case = Case(base.mesh, costs, 1.0)
    m = case.mesh
    free = np.where(costs < 0.5)[0]
    def near(fi, fj):
        v = m.vertex_at(fi, fj)
        return int(free[((m.xyz[free, :2] - m.xyz[v, :2]) ** 2).sum(1).argmin()])
    s, t = near(0.15, 0.15), near(0.85, 0.85)
    sp = m.xyz[s] + np.array([0.02, 0.01, 0], np.float32)
    tp = m.xyz[t] + np.array([0.02, 0.01, 0], np.float32)
    for off in (0.3, np.inf):
        ref, mod = run_cvp(case, sp, tp, offset=off, delta=0.3)
        fin = np.isfinite(ref.dist)
        assert np.array_equal(np.isfinite(mod["dist"]), fin)
        rel = np.abs(mod["dist"][fin] - ref.dist[fin]) / np.maximum(ref.dist[fin], 1e-12)
        assert rel.max() <= 1e-5                                   # north_star tolerance
        assert (mod["pred"] != ref.pred).mean() < 1e-3
    # and Dijkstra on the same cost-inflated weights
    refd, modd = run_dijkstra(case, s, t, delta=0.3)
    assert np.array_equal(modd["dist"].view(np.uint32), refd.dist.view(np.uint32))


def test_cvp_seed_face_equals_target_face_and_blocked_seed():
    case = Case(meshgen.terrain(24, 0.1, 15))
    m = case.mesh
    sp = m.xyz[m.vertex_at(0.5, 0.5)] + np.array([0.03, 0.02, 0], np.float32)
    ref, mod = run_cvp(case, sp, sp.copy(), delta=0.3)               # robot in the goal's face
    assert ref.code == 0
    assert np.array_equal(mod["dist"].view(np.uint32), ref.dist.view(np.uint32))
    # one seed vertex at/above the cost limit: it is fixed but never expands (cvp :757)
    sf, _ = case.om.containing_face(sp)
    costs = np.zeros(m.V, np.float32)
    costs[m.faces[sf][0]] = 1.0
    case2 = Case(m, costs, 0.0)
    tp = m.xyz[m.vertex_at(0.9, 0.1)] + np.array([0.03, 0.02, 0], np.float32)
    ref2, mod2 = run_cvp(case2, sp, tp, delta=0.3)
    assert np.array_equal(mod2["dist"].view(np.uint32), ref2.dist.view(np.uint32))
    assert np.array_equal(mod2["pred"], ref2.pred)


def test_cvp_adversarial_weights_bit_exact():
    """Random per-vertex costs up to 1.2 with edge_cost_factor 1: single edges are inflated by up to
    2.2x, most triangles violate the triangle inequality and updates undercut the pop front in
    nested cascades.  The pop keys order such pops exactly (cascade forest, mnav_eval.h::PopKey), so the
    schedule reproduces the sequential loop bit for bit in every interleaving and for every band width."""
    mesh = meshgen.terrain(96, 0.1, 13)
    rng = np.random.default_rng(3)
    costs = rng.uniform(0, 1.2, mesh.V).astype(np.float32)
    invalid = (rng.uniform(size=mesh.V) < 0.02).astype(np.uint8)
    s, t = mesh.vertex_at(0.1, 0.1), mesh.vertex_at(0.9, 0.9)
    invalid[[s, t]] = 0
    costs[[s, t]] = 0
    case = Case(mesh, costs, 1.0, invalid)
    sp = mesh.xyz[s] + np.array([0.03, 0.02, 0], np.float32)
    tp = mesh.xyz[t] + np.array([0.03, 0.02, 0], np.float32)
    for delta, order in ((0.12, 2), (0.36, 0), (0.36, 3), (1.4, 3)):
        ref, mod = run_cvp(case, sp, tp, delta=delta, order=order, max_steps=200000)
        assert mod["code"] == 0
        assert np.array_equal(mod["dist"].view(np.uint32), ref.dist.view(np.uint32))
        assert np.array_equal(mod["pred"], ref.pred)
        upd = ref.pred != np.arange(mesh.V)
        assert np.array_equal(mod["cutface"][upd], ref.cutface[upd])
        assert np.array_equal(mod["direction"][upd], ref.direction[upd])


@pytest.mark.parametrize("mult", [3, 12, 24])
@pytest.mark.parametrize("order", [0, 3])
def test_cvp_wide_bands_repair_after_arming(mult, order):
    """A band much wider than goal_dist_offset lets vertices beyond goal_dist fire faces before the goal is
    armed; some of the vertices they set get VALUES below goal_dist (non-causal updates) but pop after their
    trigger.  The repair sweeps after arming must take back exactly those (cvp :754: a vertex beyond goal_dist
    is popped but never expands), whatever the band width."""
    mesh = meshgen.terrain(96, 0.1, 13)
    rng = np.random.default_rng(3)
    costs = (rng.uniform(0, 1.2, mesh.V) * 0.5).astype(np.float32)   # edges inflated by up to 1.6x
    invalid = (rng.uniform(size=mesh.V) < 0.02).astype(np.uint8)
    s, t = mesh.vertex_at(0.1, 0.1), mesh.vertex_at(0.9, 0.9)
    invalid[[s, t]] = 0
    costs[[s, t]] = 0
    case = Case(mesh, costs, 1.0, invalid)
    sp = mesh.xyz[s] + np.array([0.03, 0.02, 0], np.float32)
    tp = mesh.xyz[t] + np.array([0.03, 0.02, 0], np.float32)
    mean_w = float(case.weights[np.isfinite(case.weights)].mean())
    ref, mod = run_cvp(case, sp, tp, delta=mult * mean_w, order=order, max_steps=200000)
    assert mod["code"] == 0
    assert np.array_equal(mod["dist"].view(np.uint32), ref.dist.view(np.uint32))
    assert np.array_equal(mod["pred"], ref.pred)
    assert mod["goal_dist"] == ref.stats["goal_dist"]


@pytest.mark.parametrize("which", ["punched", "fan"])
def test_schedule_on_ragged_meshes(which):
    """Holes, two components and face-less vertices; a hub vertex of valence 40."""
    mesh = meshgen.punched(64, 0.1, 5, drop=0.3, cut_column=40) if which == "punched" else meshgen.fan_field(40, 6, 1)
    case = Case(mesh)
    deg = np.bincount(mesh.edges.ravel(), minlength=mesh.V)
    if which == "punched":
        s, t = mesh.vertex_at(0.1, 0.2), mesh.vertex_at(0.5, 0.8)
        while deg[s] == 0: s += 1
        while deg[t] == 0: t += 1
        far = mesh.vertex_at(0.9, 0.5)
        while deg[far] == 0: far += 1
    else:
        s, t, far = 1 + 5 * 40 + 3, 1 + 5 * 40 + 23, 0
    for a, b in ((s, t), (s, far)):
        ref, mod = run_dijkstra(case, a, b, delta=0.3)        # (the model has no path walk: fields only)
        assert np.array_equal(mod["dist"].view(np.uint32), ref.dist.view(np.uint32))
        assert np.array_equal(mod["pred"], ref.pred)
    sf = int(np.where((mesh.faces == s).any(axis=1))[0][0])
    tf = int(np.where((mesh.faces == t).any(axis=1))[0][0])
    sp = mesh.xyz[mesh.faces[sf]].mean(axis=0).astype(np.float32)
    tp = mesh.xyz[mesh.faces[tf]].mean(axis=0).astype(np.float32)
    # (on the punched mesh the wave wraps around the holes and fills their shadows backwards: deep cascades
    # of pops below the main front; a band narrower than the spread of the seed values is covered too)
    for delta, order in ((0.3, 0), (0.3, 3), (0.02, 2), (1.0, 3)):
        ref, mod = run_cvp(case, sp, tp, delta=delta, order=order)
        assert np.array_equal(mod["dist"].view(np.uint32), ref.dist.view(np.uint32))
        assert np.array_equal(mod["pred"], ref.pred)


@pytest.mark.parametrize("order", [0, 2, 3])
def test_cvp_deep_cascades_converge_without_band_shrink(order):
    """200x200 terrain with 20 % of the faces punched out, band of 12 mean edges: cascades nest over a
    thousand levels deep.  Regression for two transient traps of the gather iteration: a vertex adopting its
    own (stale) child as a trigger, and a vertex whose result depends on its own stored key not looking again."""
    mesh = meshgen.punched(200, 0.1, 7, drop=0.2)
    case = Case(mesh)
    deg = np.bincount(mesh.edges.ravel(), minlength=mesh.V)
    s, t = mesh.vertex_at(0.1, 0.1), mesh.vertex_at(0.9, 0.9)
    while deg[s] == 0: s += 1
    while deg[t] == 0: t += 1
    sf = int(np.where((mesh.faces == s).any(axis=1))[0][0])
    tf = int(np.where((mesh.faces == t).any(axis=1))[0][0])
    sp = mesh.xyz[mesh.faces[sf]].astype(np.float64).mean(axis=0).astype(np.float32)
    tp = mesh.xyz[mesh.faces[tf]].astype(np.float64).mean(axis=0).astype(np.float32)
    mean_w = float(case.weights[np.isfinite(case.weights)].mean())
    ref, mod = run_cvp(case, sp, tp, delta=12 * mean_w, order=order, max_steps=100000)
    assert mod["code"] == 0
    assert np.array_equal(mod["dist"].view(np.uint32), ref.dist.view(np.uint32))
    assert np.array_equal(mod["pred"], ref.pred)
    if order != 3:
        assert mod["shrinks"] == 0


@pytest.mark.parametrize("kind", ["layered", "punched"])
def test_cvp_random_pairs_converge_and_match(kind):
    """Random seed/target pairs (list order and Jacobi) on a cost-layered terrain and on a punched one: every run
    must converge without the step cap and reproduce the oracle bit for bit (a 560-run version of this sweep,
    N = 160 and 300, had no failure; this is the part that fits the CPU suite)."""
    if kind == "layered":
        base = Case(meshgen.terrain(96, 0.1, 3, amplitude=0.8))
        costs, _ = layered_costs(base, "avg")
        case = Case(base.mesh, costs, 1.0)
    else:
        case = Case(meshgen.punched(96, 0.1, 11, drop=0.15))
        costs = case.costs
    m = case.mesh
    deg = np.bincount(m.edges.ravel(), minlength=m.V)
    free = np.where((costs < 0.5) & (deg > 0))[0]
    first_face = np.full(m.V, -1, np.int64)
    fl = m.faces.ravel()
    first_face[fl[::-1]] = np.arange(fl.size)[::-1] // 3
    mean_w = float(case.weights[np.isfinite(case.weights)].mean())
    rng = np.random.default_rng(17)
    for _ in range(8):
        s, t = (int(x) for x in rng.choice(free, 2, replace=False))
        sf, tf = int(first_face[s]), int(first_face[t])
        sp = m.xyz[m.faces[sf]].astype(np.float64).mean(axis=0).astype(np.float32)
        tp = m.xyz[m.faces[tf]].astype(np.float64).mean(axis=0).astype(np.float32)
        for order in (0, 3):
            ref, mod = run_cvp(case, sp, tp, delta=12 * mean_w, order=order, max_steps=60000)
            assert mod["code"] == 0
            assert np.array_equal(mod["dist"].view(np.uint32), ref.dist.view(np.uint32))
            assert np.array_equal(mod["pred"], ref.pred)


def test_walk_bound_hit_is_reported_not_silently_reordered(monkeypatch):
    """Deep cascades with the cascade-tree walk bounds cut down to a few links: the comparison falls back to
    another order.  The verification sweep (model of k_cvp_verify) must notice -- a raised walk-limit flag on
    the converged tree or a vertex that is no fixed point -- so the device returns INTERNAL_ERROR instead of a
    potential that may differ from the reference's.  With the default bounds the same input is clean."""
    mesh = meshgen.punched(120, 0.1, 7, drop=0.2)
    case = Case(mesh)
    deg = np.bincount(mesh.edges.ravel(), minlength=mesh.V)
    s, t = mesh.vertex_at(0.1, 0.1), mesh.vertex_at(0.9, 0.9)
    while deg[s] == 0: s += 1
    while deg[t] == 0: t += 1
    sf = int(np.where((mesh.faces == s).any(axis=1))[0][0])
    tf = int(np.where((mesh.faces == t).any(axis=1))[0][0])
    sp = mesh.xyz[mesh.faces[sf]].astype(np.float64).mean(axis=0).astype(np.float32)
    tp = mesh.xyz[mesh.faces[tf]].astype(np.float64).mean(axis=0).astype(np.float32)
    mean_w = float(case.weights[np.isfinite(case.weights)].mean())
    ref = case.om.cvp(case.weights, case.costs, case.vn, sp, sf, tf)
    sv = mesh.faces[sf]

    def run():
        return O.schedule_model(1, mesh.faces, mesh.edges, case.weights, case.costs, sv, ref.dist[sv], sf, mesh.faces[tf],
                                delta=12 * mean_w, max_steps=100000)
    clean = run()
    assert clean["code"] == 0 and clean["verify_bad"] == 0 and clean["verify_flags"] == 0
    assert np.array_equal(clean["dist"].view(np.uint32), ref.dist.view(np.uint32))
    monkeypatch.setenv("MNAV_KEY_WALK_MAX", "3")
    monkeypatch.setenv("MNAV_DESCEND_WALK_MAX", "1")
    cut = run()
    reported = cut["code"] != 0 or cut["verify_flags"] != 0 or cut["verify_bad"] != 0
    assert reported, "a walk bound of 3 links on cascades hundreds of levels deep went unnoticed"


def test_zero_weight_edges_keep_distances_exact_and_predecessors_optimal():
    """Coincident vertices (scanned meshes) give zero-length edges.  A vertex reached over one is queued AT the value
    that is popping, possibly with a smaller id than vertices of that value that have popped already, so the reference's
    pop order is no longer the global (value, id) order the predecessor rule assumes (DESIGN.md "tie rule").  What
    still holds, and is what this pins: the potential is bit-exact; every predecessor is an optimal one
    (dist[pred] + w == dist[v] in float32, the reference's own test at :331-332); the path to the robot vertex has
    the same cost.  Which of several equally good predecessors is chosen may differ from the reference on such
    vertices (16 of 1600 here)."""
    rng = np.random.default_rng(1)
    m0 = meshgen.terrain(40, 0.1, 5)
    xyz = m0.xyz.copy()
    for e in rng.choice(m0.E, 200, replace=False):
        a, b = m0.edges[e]
        xyz[b] = xyz[a]
    m = meshgen.from_faces(xyz, m0.faces, 40, 0.1)
    case = Case(m)
    assert (case.weights == 0).sum() > 100
    s, t = m.vertex_at(0.1, 0.1), m.vertex_at(0.9, 0.9)
    ref = case.om.dijkstra(case.weights, case.costs, s, t)
    wmap = {}
    for e, (a, b) in enumerate(m.edges):
        wmap[(int(a), int(b))] = wmap[(int(b), int(a))] = case.weights[e]
    for order in (0, 3):
        mod = O.schedule_model(0, m.faces, m.edges, case.weights, case.costs, [s], [0.0], O.NONE, [t], order=order)
        assert np.array_equal(mod["dist"].view(np.uint32), ref.dist.view(np.uint32))
        pred = mod["pred"]
        for v in np.nonzero(pred != np.arange(m.V))[0]:
            u = int(pred[v])
            assert np.float32(mod["dist"][u] + wmap[(u, int(v))]) == mod["dist"][v]
        def cost_to_seed(pr):
            c, v, n = 0.0, t, 0
            while v != s and n <= m.V:
                u = int(pr[v]); c += float(wmap[(u, int(v))]); v = u; n += 1
            return c, v
        (c1, e1), (c2, e2) = cost_to_seed(pred), cost_to_seed(ref.pred)
        assert e1 == e2 == s and c1 == pytest.approx(c2, rel=1e-6)


def _assert_cvp_fields_equal(mesh, ref, mod):
    assert mod["code"] == ref.code
    assert np.array_equal(mod["dist"].view(np.uint32), ref.dist.view(np.uint32))
    assert np.array_equal(mod["pred"], ref.pred)
    upd = ref.pred != np.arange(mesh.V)
    assert np.array_equal(mod["cutface"][upd], ref.cutface[upd])
    assert np.array_equal(mod["direction"][upd].view(np.uint32), ref.direction[upd].view(np.uint32))
    assert mod["goal_dist"] == ref.stats["goal_dist"]


@pytest.mark.parametrize("offset", [-0.01, -0.3, -2.0, -1e9])
@pytest.mark.parametrize("delta,order", [(0.1, 0), (0.4, 2), (1.5, 3)])
def test_cvp_negative_goal_dist_offset(offset, delta, order):
    """cvp_mesh_planner.cpp:754 comes BEFORE :765-769 in an iteration: the arming pop itself expands, every pop before it met
    goal_dist = +inf and expanded whatever its value, every later pop lies above goal_dist = (arming value + a negative
    offset) unless its value undercuts it.  The gather rule decides that by POP ORDER against the arming vertex
    (mnav_eval.h::passes_goal_cut), whatever the band width and the order of evaluation."""
    case = Case(meshgen.terrain(56, 0.1, 14))
    m = case.mesh
    sp = m.xyz[m.vertex_at(0.2, 0.2)] + np.array([0.03, 0.02, 0], np.float32)
    tp = m.xyz[m.vertex_at(0.8, 0.7)] + np.array([0.01, 0.04, 0], np.float32)
    ref, mod = run_cvp(case, sp, tp, offset=offset, delta=delta, order=order)
    _assert_cvp_fields_equal(m, ref, mod)
    zero, _ = run_cvp(case, sp, tp, offset=0.0, delta=delta, order=order)
    assert np.isfinite(ref.dist).sum() <= np.isfinite(zero.dist).sum()


@pytest.mark.parametrize("kind", ["adversarial", "punched", "layered"])
def test_cvp_negative_offsets_with_cascades(kind):
    """the same where pops undercut the front (cost-inflated triangles, waves wrapping around holes): values below goal_dist
    that pop after the arming vertex still expand (:754 is a test on the value), values above it that popped before it did too"""
    if kind == "adversarial":
        mesh = meshgen.terrain(72, 0.1, 13)
        rng = np.random.default_rng(3)
        costs = rng.uniform(0, 1.2, mesh.V).astype(np.float32)
        case = Case(mesh, costs, 1.0)
    elif kind == "punched":
        case = Case(meshgen.punched(72, 0.1, 11, drop=0.15))
        costs = case.costs
    else:
        base = Case(meshgen.terrain(72, 0.1, 3, amplitude=0.8))
        costs, _ = layered_costs(base, "avg")
        case = Case(base.mesh, costs, 1.0)
    m = case.mesh
    deg = np.bincount(m.edges.ravel(), minlength=m.V)
    free = np.where((costs < 0.5) & (deg > 0))[0]
    first_face = np.full(m.V, -1, np.int64)
    fl = m.faces.ravel()
    first_face[fl[::-1]] = np.arange(fl.size)[::-1] // 3
    mean_w = float(case.weights[np.isfinite(case.weights)].mean())
    rng = np.random.default_rng(29)
    for k in range(6):
        s, t = (int(x) for x in rng.choice(free, 2, replace=False))
        sf, tf = int(first_face[s]), int(first_face[t])
        sp = m.xyz[m.faces[sf]].astype(np.float64).mean(axis=0).astype(np.float32)
        tp = m.xyz[m.faces[tf]].astype(np.float64).mean(axis=0).astype(np.float32)
        off = (-0.02, -0.5, -3.0)[k % 3]
        ref, mod = run_cvp(case, sp, tp, offset=off, delta=(4, 12, 30)[k % 3] * mean_w, order=(0, 3)[k % 2], max_steps=100000)
        _assert_cvp_fields_equal(m, ref, mod)


def test_cvp_faces_with_a_seed_support_fire_twice():
    """A seed vertex is fixed from the start (cvp :726) but pops like every other vertex, so a face with a seed support is visited at
    the pop of its other support AND again at the seed's own pop.  The second visit offers the same candidate; :411 compares the
    float64 candidate with the stored float32 value, the re-application "succeeds" and takes predecessor / cutting face back from a
    tying face that had them in between.  Found by the round-5 soak on the fragmented layered mesh (0.2 % of random plans, one vertex
    each); the step kernels replay the first event only, seed_ring_fix both for the seeds' ring (mnav_eval.h::corner_fire_second)."""
    base = Case(meshgen.terrain(224, 0.1, 1))
    costs, _ = layered_costs(base, "avg")
    case = Case(base.mesh, costs, 1.0)
    m = case.mesh
    offv = np.array([0.02, 0.015, 0.0], np.float32)
    for s, t, off, order in ((9309, 9983, 0.0, 0), (37731, 10822, 2.5, 2), (9063, 1479, np.inf, 3), (4358, 48191, 2.5, 0), (9758, 31675, -1.0, 1)):
        ref, mod = run_cvp(case, m.xyz[s] + offv, m.xyz[t] + offv, offset=off, delta=0.3, order=order)
        assert np.array_equal(mod["dist"].view(np.uint32), ref.dist.view(np.uint32)) and np.array_equal(mod["pred"], ref.pred)
        upd = ref.pred != np.arange(m.V)
        assert np.array_equal(mod["cutface"][upd], ref.cutface[upd]), (s, t, off)
        assert np.array_equal(mod["direction"][upd].view(np.uint32), ref.direction[upd].view(np.uint32))
