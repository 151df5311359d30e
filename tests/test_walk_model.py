"""CPU: the arithmetic of the device back-tracking kernel (mesh_navigation_amd/csrc/mnav_walk.h, compiled for the host in
oracle/libmnav_model.so) against the oracle's restatement of the reference loop (cvp_mesh_planner.cpp:920-951 with
MeshMap::meshAhead, mesh_map.cpp:1070-1108): same positions and faces, bit for bit."""
from __future__ import annotations

import numpy as np
import pytest

from mesh_navigation_amd import meshgen
from oracle import oracle as O
from tests.common import Case


def walk_both(case, vm, hv, goal, sf, robot, tf, sw, field=None):
    m = case.mesh
    ptr, vf = case.om.vertex_faces()
    rc, ppos, pface = case.om.cvp_backtrack(vm, hv, goal, sf, robot, tf, step_width=sw, inflation_field=field)
    st, p2, f2 = O.product_backtrack(m.xyz, m.faces, ptr, vf, vm * hv[:, None], goal, sf, robot, tf, step_width=sw, inflation_field=field)
    assert (st == 1) == (rc == 0)
    assert np.array_equal(pface, f2) and np.array_equal(ppos.view(np.uint32), p2.view(np.uint32))
    return rc, len(pface)


@pytest.mark.parametrize("seed", [21, 5])
def test_walk_equals_the_oracle_on_terrain(seed):
    case = Case(meshgen.terrain(128, 0.1, seed))
    m = case.mesh
    robot = m.xyz[m.vertex_at(0.85, 0.8)] + np.array([0.031, 0.017, 0.0], np.float32)
    goal = m.xyz[m.vertex_at(0.12, 0.2)] + np.array([0.023, 0.011, 0.0], np.float32)
    sf, _ = case.om.containing_face(goal)
    tf, _ = case.om.containing_face(robot)
    ref = case.om.cvp(case.weights, case.costs, case.vn, goal, sf, tf)
    vm, hv = np.ascontiguousarray(ref.vecmap), np.ascontiguousarray(ref.has_vec)
    lens = [walk_both(case, vm, hv, goal, sf, robot, tf, sw) for sw in (0.4, 0.25, 0.1, 0.03)]
    assert any(rc == 0 and n > 100 for rc, n in lens)
    # a field cut short (goal_dist_offset 0: the wave stops at the robot's face): the walk still finds its way, or fails alike
    ref0 = case.om.cvp(case.weights, case.costs, case.vn, goal, sf, tf, goal_dist_offset=0.0)
    walk_both(case, np.ascontiguousarray(ref0.vecmap), np.ascontiguousarray(ref0.has_vec), goal, sf, robot, tf, 0.1)


def test_walk_with_the_inflation_layers_repulsive_field():
    """meshAhead adds every layer's vectorAt (mesh_map.cpp:1099-1102); InflationLayer's (inflation_layer.cpp:493-521) takes
    cos of a float -- the device's cosf_ref is the host's."""
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
    costs = np.where(np.isfinite(icost), icost, 0).astype(np.float32)
    case2 = Case(mesh, costs, edge_cost_factor=1.0)
    ref = case2.om.cvp(case2.weights, case2.costs, case2.vn, goal, sf, tf)
    field = (np.where(np.isfinite(idist), idist, 0).astype(np.float32), ivec, cfg, True)
    vm, hv = np.ascontiguousarray(ref.vecmap), np.ascontiguousarray(ref.has_vec)
    rc, n = walk_both(case2, vm, hv, goal, sf, robot, tf, 0.2, field=field)
    assert rc == 0 and n > 10
    rc0, ppos0, _ = case2.om.cvp_backtrack(vm, hv, goal, sf, robot, tf, step_width=0.2)
    rc1, ppos1, _ = case2.om.cvp_backtrack(vm, hv, goal, sf, robot, tf, step_width=0.2, inflation_field=field)
    assert len(ppos0) != len(ppos1) or not np.array_equal(ppos0, ppos1)               # the layer's field does bend the path


def test_walk_on_a_mesh_with_holes_and_a_high_valence_hub():
    """holes: searchNeighbourFaces runs out of faces where the reference's does; a valence-40 hub: vertex rows longer than
    the device's 32 candidate slots per listed face (the wave search falls back to the rows in memory -- same list order)"""
    for mesh in (meshgen.punched(72, 0.1, 4, drop=0.12), meshgen.fan_field(spokes=40, rings=6, seed=1)):
        case = Case(mesh)
        m = case.mesh
        deg = np.bincount(m.faces.ravel(), minlength=m.V)
        ok = np.flatnonzero(deg > 0)
        rng = np.random.default_rng(7)
        done = 0
        for _ in range(12):
            a, b = rng.choice(ok, 2, replace=False)
            goal = m.xyz[a] + np.array([0.011, 0.007, 0.0], np.float32)
            robot = m.xyz[b] + np.array([0.009, 0.013, 0.0], np.float32)
            sf, _ = case.om.containing_face(goal)
            tf, _ = case.om.containing_face(robot)
            if sf >= m.F or tf >= m.F or sf < 0 or tf < 0:
                continue
            ref = case.om.cvp(case.weights, case.costs, case.vn, goal, int(sf), int(tf))
            vm, hv = np.ascontiguousarray(ref.vecmap), np.ascontiguousarray(ref.has_vec)
            for sw in (0.3, 0.08):
                walk_both(case, vm, hv, goal, int(sf), robot, int(tf), sw)
                done += 1
        assert done >= 8
