"""GPU, row f3 of SURVEY.md section 8: CVPMeshPlanner's back-tracking (cvp_mesh_planner.cpp:920-951, MeshMap::meshAhead
mesh_map.cpp:1070-1108) ON THE DEVICE, over the vector map the CVP call left in HBM (mnav_backtrack_cvp, include/mnav.h).
Expected: the oracle's restatement of the reference loop on the oracle's own field -- same faces, same float32 positions,
bit for bit (the device field is the oracle's bit for bit since the rotation uses the host libm's sinf / cosf bits)."""
from __future__ import annotations

import numpy as np
import pytest

from mesh_navigation_amd import capi, meshgen
from oracle import oracle as O
from tests.common import Case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def world():
    case = Case(meshgen.terrain(160, 0.1, 21))
    ctx = capi.MnavContext(0)
    ctx.upload_mesh(case.mesh.xyz, case.mesh.faces, case.mesh.edges, case.vn)
    ctx.upload_costs(case.costs, case.weights)
    ctx.set_resident_outputs(True)
    yield case, ctx
    ctx.close()


def ends(case, a, b):
    m = case.mesh
    robot = m.xyz[m.vertex_at(*a)] + np.array([0.031, 0.017, 0.0], np.float32)
    goal = m.xyz[m.vertex_at(*b)] + np.array([0.023, 0.011, 0.0], np.float32)
    sf, _ = case.om.containing_face(goal)
    tf, _ = case.om.containing_face(robot)
    return goal, int(sf), robot, int(tf)


def test_vector_map_and_walk_are_the_oracles_bits(world):
    case, ctx = world
    goal, sf, robot, tf = ends(case, (0.85, 0.8), (0.12, 0.2))
    ref = case.om.cvp(case.weights, case.costs, case.vn, goal, sf, tf)
    out = ctx.plan_cvp(goal, sf, tf, want_fields=False, want_vecmap=True)
    assert out.code == ref.code == 0
    exp = ref.vecmap * ref.has_vec[:, None]                         # the device writes zeros where the reference's map has no entry
    assert np.array_equal(out.vecmap.view(np.uint32), exp.astype(np.float32).view(np.uint32))
    for sw in (0.4, 0.25, 0.1, 0.03):
        rc, ppos, pface = case.om.cvp_backtrack(ref.vecmap, ref.has_vec, goal, sf, robot, tf, step_width=sw)
        st, pos, face = ctx.backtrack_cvp(goal, sf, robot, tf, step_width=sw, cap=8192)
        assert (st == 1) == (rc == 0), sw
        assert np.array_equal(face, pface) and np.array_equal(pos.view(np.uint32), ppos.view(np.uint32)), sw
    # nothing V-sized has to cross PCIe for it: a plan without host outputs leaves the field resident
    out2 = ctx.plan_cvp(goal, sf, tf, want_fields=False, want_vecmap=False)
    assert out2.code == 0
    rc, ppos, pface = case.om.cvp_backtrack(ref.vecmap, ref.has_vec, goal, sf, robot, tf, step_width=0.1)
    st, pos, face = ctx.backtrack_cvp(goal, sf, robot, tf, step_width=0.1)
    assert st == 1 and np.array_equal(face, pface) and np.array_equal(pos.view(np.uint32), ppos.view(np.uint32))
    # capacity: the walk stops with "no path" when the row is full, like the oracle's guard
    st, pos, face = ctx.backtrack_cvp(goal, sf, robot, tf, step_width=0.1, cap=16)
    rc, ppos, pface = case.om.cvp_backtrack(ref.vecmap, ref.has_vec, goal, sf, robot, tf, step_width=0.1, cap=16)
    assert st == 0 and rc != 0 and len(face) == 16 == len(pface)


def test_batch_of_walks(world):
    case, ctx = world
    pairs = [((0.85, 0.8), (0.12, 0.2)), ((0.2, 0.9), (0.8, 0.15)), ((0.5, 0.5), (0.52, 0.5)), ((0.1, 0.1), (0.9, 0.9)), ((0.9, 0.1), (0.3, 0.6))]
    E = [ends(case, a, b) for a, b in pairs]
    sps = np.array([e[0] for e in E], np.float32)
    sfs = np.array([e[1] for e in E], np.uint32)
    tps = np.array([e[2] for e in E], np.float32)
    tfs = np.array([e[3] for e in E], np.uint32)
    ctx.plan_cvp_batch(sps, sfs, tfs)
    got = ctx.backtrack_cvp_batch(sps, sfs, tps, tfs, step_width=0.15)
    assert len(got) == len(E)
    for (goal, sf, robot, tf), (st, pos, face) in zip(E, got):
        ref = case.om.cvp(case.weights, case.costs, case.vn, goal, sf, tf)
        rc, ppos, pface = case.om.cvp_backtrack(ref.vecmap, ref.has_vec, goal, sf, robot, tf, step_width=0.15)
        assert (st == 1) == (rc == 0)
        assert np.array_equal(face, pface) and np.array_equal(pos.view(np.uint32), ppos.view(np.uint32))


def test_walk_needs_a_cvp_field(world):
    case, ctx = world
    goal, sf, robot, tf = ends(case, (0.85, 0.8), (0.12, 0.2))
    ctx.plan_dijkstra(3, 900, want_fields=False, want_vecmap=False)
    with pytest.raises(RuntimeError):
        ctx.backtrack_cvp(goal, sf, robot, tf)
    ctx.plan_cvp(goal, sf, tf, want_fields=False, want_vecmap=False)
    with pytest.raises(RuntimeError):
        ctx.backtrack_cvp(goal, case.mesh.F + 5, robot, tf)
    with pytest.raises(RuntimeError):
        ctx.backtrack_cvp(goal, sf, robot, tf, step_width=0.0)


def test_walk_with_the_device_built_inflation_layer():
    """meshAhead adds the layers' vectorAt (mesh_map.cpp:1099-1102): the inflation layer computed on the device
    (mnav_layer_inflation) feeds the device walk; expected = the oracle's walk with the oracle's layer fields."""
    mesh = meshgen.terrain(44, 0.1, 12, amplitude=0.3)
    N = mesh.N
    lethal = np.zeros(mesh.V, np.uint8)
    i, j = np.meshgrid(np.arange(N), np.arange(N))
    lethal[(((j == 18) | (j == 25)) & (i > 3) & (i < N - 4)).ravel()] = 1            # a corridor between two lethal walls
    case = Case(mesh)
    cfg = O.InflationCfg.defaults()
    icost, idist, ivec = case.om.inflation(lethal, case.edge_dist, cfg)
    goal = mesh.xyz[21 * N + 6] + np.array([0.02, 0.03, 0.0], np.float32)
    robot = mesh.xyz[22 * N + N - 8] + np.array([0.03, 0.01, 0.0], np.float32)
    sf, _ = case.om.containing_face(goal)
    tf, _ = case.om.containing_face(robot)
    with capi.MnavContext(0) as ctx:
        ctx.upload_mesh(mesh.xyz, mesh.faces, mesh.edges, case.vn)
        ctx.layer_upload(0, np.zeros(mesh.V, np.float32), lethal)
        ctx.layer_inflation(1, 0)
        ctx.combine_layers([1], [1.0], mode="max", edge_cost_factor=1.0)
        vc, w = ctx.download_costs()
        ctx.set_resident_outputs(True)
        out = ctx.plan_cvp(goal, sf, tf, want_fields=False, want_vecmap=False)
        st, pos, face = ctx.backtrack_cvp(goal, sf, robot, tf, step_width=0.2, inflation_layer=1)
        st0, pos0, face0 = ctx.backtrack_cvp(goal, sf, robot, tf, step_width=0.2)
        with pytest.raises(RuntimeError):
            ctx.backtrack_cvp(goal, sf, robot, tf, step_width=0.2, inflation_layer=0)     # not an inflation layer
    ref = case.om.cvp(w, vc, case.vn, goal, sf, tf)
    assert out.code == ref.code == 0
    field = (np.where(np.isfinite(idist), idist, 0).astype(np.float32), ivec, cfg, True)
    rc, ppos, pface = case.om.cvp_backtrack(ref.vecmap, ref.has_vec, goal, sf, robot, tf, step_width=0.2, inflation_field=field)
    assert rc == 0 and st == 1 and len(pface) > 10
    assert np.array_equal(face, pface) and np.array_equal(pos.view(np.uint32), ppos.view(np.uint32))
    rc0, ppos0, pface0 = case.om.cvp_backtrack(ref.vecmap, ref.has_vec, goal, sf, robot, tf, step_width=0.2)
    assert st0 == 1 and np.array_equal(face0, pface0) and np.array_equal(pos0.view(np.uint32), ppos0.view(np.uint32))
    assert not np.array_equal(pos0, pos) if len(pos0) == len(pos) else True


def test_walk_over_a_high_valence_hub_and_holes():
    """a valence-40 hub: the rows of getFacesOfVertex exceed the 32 candidate slots a lane gathers per listed face, the wave
    search then reads the rows from memory -- same list order as the sequential search; holes: the search runs dry alike"""
    for mesh in (meshgen.fan_field(spokes=40, rings=6, seed=1), meshgen.punched(72, 0.1, 4, drop=0.12)):
        case = Case(mesh)
        m = case.mesh
        deg = np.bincount(m.faces.ravel(), minlength=m.V)
        okv = np.flatnonzero(deg > 0)
        rng = np.random.default_rng(7)
        with capi.MnavContext(0) as ctx:
            ctx.upload_mesh(m.xyz, m.faces, m.edges, case.vn)
            ctx.upload_costs(case.costs, case.weights)
            ctx.set_resident_outputs(True)
            done = 0
            for _ in range(10):
                a, b = rng.choice(okv, 2, replace=False)
                goal = m.xyz[a] + np.array([0.011, 0.007, 0.0], np.float32)
                robot = m.xyz[b] + np.array([0.009, 0.013, 0.0], np.float32)
                sf, _ = case.om.containing_face(goal)
                tf, _ = case.om.containing_face(robot)
                if not (0 <= sf < m.F and 0 <= tf < m.F):
                    continue
                ref = case.om.cvp(case.weights, case.costs, case.vn, goal, int(sf), int(tf))
                out = ctx.plan_cvp(goal, int(sf), int(tf), want_fields=False, want_vecmap=False)
                assert out.code == ref.code
                if ref.code not in (0, 54):
                    continue
                for sw in (0.3, 0.08):
                    rc, ppos, pface = case.om.cvp_backtrack(ref.vecmap, ref.has_vec, goal, int(sf), robot, int(tf), step_width=sw)
                    st, pos, face = ctx.backtrack_cvp(goal, int(sf), robot, int(tf), step_width=sw, cap=8192)
                    assert (st == 1) == (rc == 0)
                    assert np.array_equal(face, pface) and np.array_equal(pos.view(np.uint32), ppos.view(np.uint32))
                    done += 1
            assert done >= 6
