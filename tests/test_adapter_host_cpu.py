"""CPU checks of the adapter's host-side MeshMap stand-in (nearest vertex, containing face,
vector-field back-tracking: the parts of makePlan that stay on the CPU by design) against the oracle."""
import ctypes as C

import numpy as np

from mesh_navigation_amd import build, meshgen
from tests.common import Case


def lib():
    L = C.CDLL(build.build_adapter())
    vp, u32 = C.c_void_p, C.c_uint32
    L.mnav_adapter_host_backtrack.restype = u32
    L.mnav_adapter_host_backtrack.argtypes = [u32, u32, vp, vp, vp, vp, vp, u32, vp, u32, C.c_double, u32, vp, vp, C.POINTER(u32)]
    L.mnav_adapter_host_nearest_vertex.restype = u32
    L.mnav_adapter_host_nearest_vertex.argtypes = [u32, u32, vp, vp, vp]
    L.mnav_adapter_host_containing_face.restype = u32
    L.mnav_adapter_host_containing_face.argtypes = [u32, u32, vp, vp, vp]
    return L


def P(a):
    return a.ctypes.data_as(C.c_void_p)


def test_seed_lookup_matches_oracle():
    L = lib()
    case = Case(meshgen.terrain(64, 0.1, 22))
    m = case.mesh
    rng = np.random.default_rng(0)
    for _ in range(200):
        p = np.array([rng.uniform(-0.3, 6.6), rng.uniform(-0.3, 6.6), rng.uniform(-1, 1)], np.float32)
        assert L.mnav_adapter_host_nearest_vertex(m.V, m.F, P(m.xyz), P(m.faces), P(p)) == case.om.nearest_vertex(p)
        f, _ = case.om.containing_face(p)
        assert L.mnav_adapter_host_containing_face(m.V, m.F, P(m.xyz), P(m.faces), P(p)) == (f & 0xFFFFFFFF)


def test_backtracking_matches_oracle():
    L = lib()
    case = Case(meshgen.terrain(128, 0.1, 21))
    m = case.mesh
    robot = m.xyz[m.vertex_at(0.85, 0.8)] + np.array([0.031, 0.017, 0.0], np.float32)
    goal = m.xyz[m.vertex_at(0.12, 0.2)] + np.array([0.023, 0.011, 0.0], np.float32)
    sf, _ = case.om.containing_face(goal)
    tf, _ = case.om.containing_face(robot)
    ref = case.om.cvp(case.weights, case.costs, case.vn, goal, sf, tf)
    vm, hv = np.ascontiguousarray(ref.vecmap), np.ascontiguousarray(ref.has_vec)
    for sw in (0.4, 0.25, 0.1):
        rc, ppos, pface = case.om.cvp_backtrack(vm, hv, goal, sf, robot, tf, step_width=sw)
        pp = np.empty((5000, 3), np.float32)
        pf = np.empty(5000, np.uint32)
        n = C.c_uint32(0)
        code = L.mnav_adapter_host_backtrack(m.V, m.F, P(m.xyz), P(m.faces), P(vm), P(hv), P(goal), sf, P(robot), tf, sw, 5000,
                                             P(pp), P(pf), C.byref(n))
        assert code == rc and n.value == len(pface)
        assert np.array_equal(pf[: n.value], pface) and np.array_equal(pp[: n.value], ppos)


def test_nearest_vertex_for_queries_far_off_the_mesh():
    """the reference's kd-tree always returns the nearest vertex; the grid search must too (clamped start cell)"""
    L = lib()
    case = Case(meshgen.terrain(40, 0.1, 8))
    m = case.mesh
    for p in ([-25.0, 1.0, 0.0], [30.0, 40.0, 3.0], [2.0, -90.0, 0.0], [-7.0, -7.0, -7.0], [1.0, 1.0, 50.0]):
        p = np.array(p, np.float32)
        assert L.mnav_adapter_host_nearest_vertex(m.V, m.F, P(m.xyz), P(m.faces), P(p)) == case.om.nearest_vertex(p)


def corridor_case():
    """a corridor between two walls of lethal vertices 0.7 m apart: every corridor vertex lies inside the inflation
    radius of a wall, so the Inflation layer's repulsive field exists along the whole path"""
    mesh = meshgen.terrain(44, 0.1, 12, amplitude=0.3)
    N = mesh.N
    lethal = np.zeros(mesh.V, np.uint8)
    i, j = np.meshgrid(np.arange(N), np.arange(N))
    lethal[(((j == 18) | (j == 25)) & (i > 3) & (i < N - 4)).ravel()] = 1
    return mesh, lethal


def test_backtracking_with_the_inflation_repulsive_field_matches_the_reference():
    """meshAhead adds every layer's vectorAt (mesh_map.cpp:1099-1102, inflation_layer.cpp:493-521): the adapter's
    stand-in against the REFERENCE's own MeshMap + InflationLayer + CVP planner (oracle/_ref)."""
    import pytest
    from oracle import ref as R
    if not R.available():
        pytest.skip("oracle/_ref not available")
    L = lib()
    u32, vp = C.c_uint32, C.c_void_p
    L.mnav_adapter_host_backtrack_layer.restype = u32
    L.mnav_adapter_host_backtrack_layer.argtypes = [u32, u32, vp, vp, vp, vp, vp, u32, vp, u32, C.c_double, vp, vp, vp, vp, C.c_double, C.c_double,
                                                    C.c_double, C.c_double, C.c_int, u32, vp, vp, C.POINTER(u32), C.POINTER(C.c_int)]
    mesh, lethal = corridor_case()
    rm = R.RefMap(mesh.xyz, mesh.faces, layers="array+inflation", lethal=lethal, edge_cost_factor=1.0)
    d, vec = rm.inflation_fields()
    has_d = np.isfinite(d).astype(np.uint8)
    has_v = (np.abs(vec).sum(1) > 0).astype(np.uint8)
    N = mesh.N
    goal = mesh.xyz[21 * N + 6] + np.array([0.02, 0.03, 0.0], np.float32)       # inside the corridor, left end
    robot = mesh.xyz[22 * N + N - 8] + np.array([0.03, 0.01, 0.0], np.float32)  # right end
    rc = rm.cvp(goal, robot, step_width=0.2)
    assert rc.code == 0, rc.message                                             # repulsive_field = true (default) works here
    sf, tf = rm.containing_face(goal)[0], rm.containing_face(robot)[0]
    pp = np.empty((5000, 3), np.float32)
    pf = np.empty(5000, np.uint32)
    n, panicked = u32(0), C.c_int(0)
    vm, hv = np.ascontiguousarray(rc.vecmap), np.ascontiguousarray(rc.has_vec)
    dd = np.where(np.isfinite(d), d, 0).astype(np.float32)
    code = L.mnav_adapter_host_backtrack_layer(mesh.V, mesh.F, P(mesh.xyz), P(mesh.faces), P(vm), P(hv), P(goal), sf, P(robot), tf, 0.2,
                                               P(dd), P(has_d), P(np.ascontiguousarray(vec)), P(has_v), 0.25, 0.4, 1.0, 0.99, 1, 5000,
                                               P(pp), P(pf), C.byref(n), C.byref(panicked))
    assert code == 0 and panicked.value == 0 and n.value == len(rc.path_face)
    assert np.array_equal(pf[: n.value], rc.path_face)
    assert np.array_equal(pp[: n.value].view(np.uint32), rc.path_pos.view(np.uint32))
    # without the layer term the path is a different one: the field really acts
    code0 = L.mnav_adapter_host_backtrack(mesh.V, mesh.F, P(mesh.xyz), P(mesh.faces), P(vm), P(hv), P(goal), sf, P(robot), tf, 0.2, 5000,
                                          P(pp), P(pf), C.byref(n))
    assert code0 == 0 and not (n.value == len(rc.path_face) and np.array_equal(pp[: n.value], rc.path_pos))
    # a robot outside the corridor: the reference panics on the first face the inflation wave did not reach -- so do we
    far = mesh.xyz[38 * N + 22] + np.array([0.02, 0.02, 0.0], np.float32)
    rc2 = rm.cvp(goal, far, step_width=0.2)
    tf2 = rm.containing_face(far)[0]
    if rc2.message.endswith("HalfEdgeMesh panicked!"):
        vm2, hv2 = np.ascontiguousarray(rc2.vecmap), np.ascontiguousarray(rc2.has_vec)
        code2 = L.mnav_adapter_host_backtrack_layer(mesh.V, mesh.F, P(mesh.xyz), P(mesh.faces), P(vm2), P(hv2), P(goal), sf, P(far), tf2, 0.2,
                                                    P(dd), P(has_d), P(np.ascontiguousarray(vec)), P(has_v), 0.25, 0.4, 1.0, 0.99, 1, 5000,
                                                    P(pp), P(pf), C.byref(n), C.byref(panicked))
        assert code2 == 54 and panicked.value == 1
