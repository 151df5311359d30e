"""The wide CVP step kernel (k_step_wide: 64 work-list entries per wave and round; phase A one lane per incident face,
phase B one lane per vertex over the prepared items; mnav_eval.h make_cvp_item / eval_cvp_items) against the oracle
(cvp_mesh_planner.cpp:369-556, 651-918) and against the 8-lane replay: batches pick it from 32 plans on, the option cvp_wide forces
either.  Potential and predecessors bit for bit."""
import numpy as np
import pytest

from mesh_navigation_amd import meshgen
from tests.common import Case, layered_costs, terrain_case

pytestmark = pytest.mark.gpu


def _batch(case, n, seed):
    m = case.mesh
    free = np.where(case.costs < 0.5)[0]
    rng = np.random.default_rng(seed)
    verts = rng.choice(free, size=n, replace=False)
    off = np.array([0.02, 0.015, 0.0], np.float32)
    sps = np.stack([m.xyz[v] + off for v in verts]).astype(np.float32)
    sfs = np.array([case.om.containing_face(p)[0] for p in sps], np.uint32)
    t = int(free[((m.xyz[free, :2] - m.xyz[m.vertex_at(0.9, 0.9), :2]) ** 2).sum(1).argmin()])
    tf, _ = case.om.containing_face(m.xyz[t] + off)
    return sps, sfs, np.full(n, tf, np.uint32)


def test_wide_batch_on_layered_costs_matches_oracle_and_the_8_lane_replay(gpu_ctx_factory):
    base = terrain_case(224, 1)
    costs, _ = layered_costs(base, "avg")                             # Steepness + Inflation: cost-inflated triangles, cascades
    case = Case(base.mesh, costs, 1.0)
    ctx = gpu_ctx_factory()
    case.upload(ctx)
    sps, sfs, tfs = _batch(case, 32, 11)
    ctx.set_option("cvp_wide", None)
    wide = ctx.plan_cvp_batch(sps, sfs, tfs, want_fields=True)         # 32 plans: the wide kernel
    ctx.set_option("cvp_wide", 0)
    narrow = ctx.plan_cvp_batch(sps, sfs, tfs, want_fields=True)
    ctx.set_option("cvp_wide", None)
    assert np.array_equal(wide["codes"], narrow["codes"])
    assert np.array_equal(wide["dist"].view(np.uint32), narrow["dist"].view(np.uint32)) and np.array_equal(wide["pred"], narrow["pred"])
    assert wide["stats"]["steps"] > 50
    for k in (0, 7, 19, 31):
        ref = case.om.cvp(case.weights, case.costs, case.vn, sps[k], int(sfs[k]), int(tfs[k]))
        assert wide["codes"][k] == ref.code
        assert np.array_equal(wide["dist"][k].view(np.uint32), ref.dist.view(np.uint32)) and np.array_equal(wide["pred"][k], ref.pred)


def test_wide_kernel_on_irregular_valence_and_adversarial_costs(gpu_ctx_factory):
    """a valence-40 hub (more faces than a vertex gets item slots for: the serial rule inside the wide kernel) and random
    per-vertex costs that break the triangle inequality on most faces (non-causal updates, cascades)"""
    mesh = meshgen.fan_field(40, 6, 1)
    case = Case(mesh)
    ctx = gpu_ctx_factory()
    ctx.set_option("cvp_wide", 1)
    case.upload(ctx)
    for sv, tv in ((1 + 5 * 40 + 3, 1 + 5 * 40 + 23), (0, 1 + 5 * 40 + 23)):
        sf = int(np.where((mesh.faces == sv).any(axis=1))[0][0]); tf = int(np.where((mesh.faces == tv).any(axis=1))[0][0])
        sp = mesh.xyz[mesh.faces[sf]].mean(axis=0).astype(np.float32)
        ref = case.om.cvp(case.weights, case.costs, case.vn, sp, sf, tf, goal_dist_offset=float("inf"))
        out = ctx.plan_cvp(sp, sf, tf, goal_dist_offset=float("inf"))
        assert out.code == ref.code
        assert np.array_equal(out.dist.view(np.uint32), ref.dist.view(np.uint32)) and np.array_equal(out.pred, ref.pred)
    m2 = meshgen.terrain(96, 0.1, 5)
    rng = np.random.default_rng(2)
    case2 = Case(m2, rng.uniform(0.0, 0.9, m2.V).astype(np.float32), 3.0)
    ctx2 = gpu_ctx_factory()
    ctx2.set_option("cvp_wide", 1)
    case2.upload(ctx2)
    sps, sfs, tfs = _batch(Case(m2, np.zeros(m2.V, np.float32)), 6, 3)
    for k in range(6):
        ref = case2.om.cvp(case2.weights, case2.costs, case2.vn, sps[k], int(sfs[k]), int(tfs[k]), goal_dist_offset=float("inf"))
        out = ctx2.plan_cvp(sps[k], int(sfs[k]), int(tfs[k]), goal_dist_offset=float("inf"))
        assert out.code == ref.code
        assert np.array_equal(out.dist.view(np.uint32), ref.dist.view(np.uint32)) and np.array_equal(out.pred, ref.pred)


@pytest.mark.parametrize("kind", ["layered", "adversarial"])
def test_negative_goal_dist_offset_in_cvp(gpu_ctx_factory, kind):
    """`goal_dist_offset` is any double in the reference (cvp_mesh_planner.cpp:157, :769).  :754 is tested BEFORE the arming in
    :765-769, so the arming pop and every pop before it expand whatever their value; below zero everything later is cut off unless
    its value undercuts goal_dist.  Single plans (8-lane replay), a batch on the wide kernel and the forced other kernel each:
    potential, predecessors, cutting faces and directions bit for bit, and the vector map where the reference has an entry."""
    if kind == "layered":
        base = terrain_case(224, 1)
        costs, _ = layered_costs(base, "avg")
        case = Case(base.mesh, costs, 1.0)
    else:
        m2 = meshgen.terrain(128, 0.1, 5)
        case = Case(m2, np.random.default_rng(2).uniform(0.0, 1.1, m2.V).astype(np.float32), 1.0)
    ctx = gpu_ctx_factory()
    case.upload(ctx)
    sps, sfs, tfs = _batch(Case(case.mesh, np.where(case.costs < 0.5, 0.0, 1.0).astype(np.float32)), 34, 7)
    for off in (-0.05, -1.5, float("-inf")):
        ks = [k for k in range(34) if sfs[k] < case.mesh.F]
        ks = ks[:2] + ks[-1:] + [k for k in range(34) if sfs[k] >= case.mesh.F][:1]     # (+ a seed off the mesh: INVALID_START)
        refs = {k: case.om.cvp(case.weights, case.costs, case.vn, sps[k], int(sfs[k]), int(tfs[k]), goal_dist_offset=off) for k in ks}
        for k, ref in refs.items():
            for wide in (0, 1):
                ctx.set_option("cvp_wide", wide)
                out = ctx.plan_cvp(sps[k], int(sfs[k]), int(tfs[k]), goal_dist_offset=off)
                assert out.code == ref.code, (kind, off, k, wide)
                if ref.code == 52:
                    continue
                assert np.array_equal(out.dist.view(np.uint32), ref.dist.view(np.uint32)), (kind, off, k, wide)
                assert np.array_equal(out.pred, ref.pred)
                upd = ref.pred != np.arange(case.mesh.V)
                assert np.array_equal(out.cutface[upd], ref.cutface[upd])
                assert np.array_equal(out.direction[upd].view(np.uint32), ref.direction[upd].view(np.uint32))
                has = ref.has_vec.astype(bool)
                assert np.array_equal(out.vecmap[has].view(np.uint32), ref.vecmap[has].view(np.uint32))
        ctx.set_option("cvp_wide", None)
        b = ctx.plan_cvp_batch(sps, sfs, tfs, goal_dist_offset=off, want_fields=True)      # 34 plans: the wide kernel
        for k, ref in refs.items():
            assert b["codes"][k] == ref.code
            if ref.code != 52:
                assert np.array_equal(b["dist"][k].view(np.uint32), ref.dist.view(np.uint32)) and np.array_equal(b["pred"][k], ref.pred), (kind, off, k)
    with pytest.raises(RuntimeError, match="goal_dist_offset"):
        ctx.plan_cvp(sps[0], int(sfs[0]), int(tfs[0]), goal_dist_offset=float("nan"))


def test_faces_with_a_seed_support_fire_twice(gpu_ctx_factory):
    """cvp :726 fixes the seeds from the start, but they pop like everybody else: a face with a seed support is visited twice, and the
    second visit re-applies its candidate through :411's float64-against-float32 comparison (tests/test_schedule_model.py, same
    name).  The cases the round-5 soak found, on both step kernels: cutting faces and directions around the seed face bit for bit."""
    base = terrain_case(224, 1)
    costs, _ = layered_costs(base, "avg")
    case = Case(base.mesh, costs, 1.0)
    m = case.mesh
    ctx = gpu_ctx_factory()
    case.upload(ctx)
    offv = np.array([0.02, 0.015, 0.0], np.float32)
    for s, t, off in ((9309, 9983, 0.0), (37731, 10822, 2.5), (9063, 1479, float("inf")), (4358, 48191, 2.5), (9758, 31675, -1.0)):
        sp, tp = m.xyz[s] + offv, m.xyz[t] + offv
        sf, _ = case.om.containing_face(sp)
        tf, _ = case.om.containing_face(tp)
        ref = case.om.cvp(case.weights, case.costs, case.vn, sp, int(sf), int(tf), goal_dist_offset=off)
        upd = ref.pred != np.arange(m.V)
        for wide in (0, 1):
            ctx.set_option("cvp_wide", wide)
            out = ctx.plan_cvp(sp, int(sf), int(tf), goal_dist_offset=off)
            assert out.code == ref.code
            assert np.array_equal(out.dist.view(np.uint32), ref.dist.view(np.uint32)) and np.array_equal(out.pred, ref.pred)
            assert np.array_equal(out.cutface[upd], ref.cutface[upd]), (s, t, off, wide)
            assert np.array_equal(out.direction[upd].view(np.uint32), ref.direction[upd].view(np.uint32))
            has = ref.has_vec.astype(bool)
            assert np.array_equal(out.vecmap[has].view(np.uint32), ref.vecmap[has].view(np.uint32))
    ctx.set_option("cvp_wide", None)
