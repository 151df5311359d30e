"""The MeshPlanner-shaped adapter (initialize / makePlan / cancel), driven like mbf_mesh_nav drives a
planner plugin, against the oracle's restatement of both makePlan bodies
(dijkstra_mesh_planner.cpp:55-134, cvp_mesh_planner.cpp:62-140)."""
import numpy as np
import pytest

from mesh_navigation_amd import meshgen
from mesh_navigation_amd.planner import CVPMeshPlanner, DijkstraMeshPlanner
from tests.common import Case

pytestmark = pytest.mark.gpu


def mesh_map_of(case):
    return dict(xyz=case.mesh.xyz, faces=case.mesh.faces, edges=case.mesh.edges, vertex_normals=case.vn,
                face_normals=case.fn, vertex_costs=case.costs, edge_weights=case.weights, invalid=case.invalid)


def pose(p):
    return np.array([p[0], p[1], p[2], 0, 0, 0, 1], np.float64)


@pytest.fixture(scope="module")
def case():
    return Case(meshgen.terrain(128, 0.1, 21))


def test_dijkstra_make_plan_matches_reference_restatement(case):
    m = case.mesh
    robot = m.xyz[m.vertex_at(0.85, 0.8)] + np.array([0.031, 0.017, 0.05], np.float32)
    goal = m.xyz[m.vertex_at(0.12, 0.2)] + np.array([0.023, 0.011, 0.02], np.float32)
    pl = DijkstraMeshPlanner()
    assert pl.initialize("dijkstra_mesh_planner", mesh_map_of(case))
    code, plan, cost, msg = pl.makePlan(pose(robot), pose(goal))
    # oracle: dijkstra(goal, start) then the pose loop of makePlan
    seed, target = case.om.nearest_vertex(goal), case.om.nearest_vertex(robot)
    ref = case.om.dijkstra(case.weights, case.costs, seed, target)
    poses, rcost = case.om.dijkstra_poses(case.vn, ref.path, robot, goal)
    assert code == ref.code == 0
    assert plan.shape == poses.shape and len(plan) == len(ref.path) + 1
    assert np.array_equal(plan[:, :3], poses[:, :3]) and np.allclose(plan[:, 3:], poses[:, 3:], atol=1e-12)
    assert cost == pytest.approx(rcost, rel=1e-12)
    # same vertex -> SUCCESS and an empty plan (dijkstra :252-255, makePlan :90)
    code2, plan2, cost2, _ = pl.makePlan(pose(robot), pose(robot))
    assert code2 == 0 and len(plan2) == 0 and cost2 == 0
    pl.close()


def test_cvp_make_plan_matches_reference_restatement(case):
    m = case.mesh
    robot = m.xyz[m.vertex_at(0.85, 0.8)] + np.array([0.031, 0.017, 0.0], np.float32)
    goal = m.xyz[m.vertex_at(0.12, 0.2)] + np.array([0.023, 0.011, 0.0], np.float32)
    pl = CVPMeshPlanner()
    # step_width 0.25: with the default 0.4 (four edge lengths on this terrain) the reference's own
    # meshAhead/searchNeighbourFaces loses the surface after three steps -- checked below as well
    assert pl.initialize("cvp_mesh_planner", mesh_map_of(case), dict(step_width=0.25))
    gpose = pose(goal)
    gpose[3:] = [0, 0, np.sin(0.3), np.cos(0.3)]
    code, plan, cost, msg = pl.makePlan(pose(robot), gpose)
    sf, _ = case.om.containing_face(goal)
    tf, _ = case.om.containing_face(robot)
    ref = case.om.cvp(case.weights, case.costs, case.vn, goal, sf, tf)
    rcode, ppos, pface = case.om.cvp_backtrack(ref.vecmap, ref.has_vec, goal, sf, robot, tf, step_width=0.25)
    poses, rcost = case.om.cvp_poses(case.fn, ppos, pface, gpose)
    assert code == ref.code == rcode == 0, msg
    assert len(plan) == len(poses)
    assert np.abs(plan[:, :3] - poses[:, :3]).max() < 2e-3          # back-tracking on a field within 1e-5 of the oracle's
    assert np.allclose(plan[-1], gpose)                              # goal pose verbatim (cvp :119-123)
    assert cost == pytest.approx(rcost, rel=1e-3)
    pl.close()
    # default step width: same failure mode and message as the reference restatement (cvp :937-942)
    pl2 = CVPMeshPlanner()
    assert pl2.initialize("cvp_mesh_planner", mesh_map_of(case))
    code2, plan2, _, msg2 = pl2.makePlan(pose(robot), gpose)
    rcode2, ppos2, pface2 = case.om.cvp_backtrack(ref.vecmap, ref.has_vec, goal, sf, robot, tf, step_width=0.4)
    assert code2 == rcode2
    if code2 == 54:
        # like the reference, makePlan still turns the partial path into poses (cvp :101-124)
        poses2, _ = case.om.cvp_poses(case.fn, ppos2, pface2, gpose)
        assert "back-tracking" in msg2 and len(plan2) == len(poses2)
    pl2.close()


def test_adapter_picks_up_cost_changes_and_cancel(case):
    m = case.mesh
    pl = DijkstraMeshPlanner()
    assert pl.initialize("dijkstra_mesh_planner", mesh_map_of(case))
    robot, goal = m.xyz[m.vertex_at(0.9, 0.5)], m.xyz[m.vertex_at(0.1, 0.5)]
    code, plan, cost, _ = pl.makePlan(pose(robot), pose(goal))
    assert code == 0
    # a wall of over-limit vertices appears (layer update, mesh_map.cpp:454-493): no path any more
    costs = case.costs.copy()
    costs[np.abs(m.xyz[:, 0] - m.xyz[:, 0].mean()) < 0.15] = 5.0
    pl.set_costs(costs, case.weights)
    code2, plan2, _, _ = pl.makePlan(pose(robot), pose(goal))
    assert code2 == 54 and len(plan2) == 0                           # NO_PATH_FOUND (dijkstra :358-362)
    pl.set_costs(case.costs, case.weights)
    assert pl.cancel()                                               # a stale cancel is cleared at the next plan (:238)
    code3, plan3, cost3, _ = pl.makePlan(pose(robot), pose(goal))
    # (the last orientation is NaN here: the goal sits exactly on a vertex, zero direction, as in the reference)
    assert code3 == 0 and np.array_equal(plan3, plan, equal_nan=True) and cost3 == cost
    pl.close()
