"""Regenerates tests/golden/*.npz from the REFERENCE's own code (python tests/golden/make_golden.py).

Every array in the fixtures is an output of oracle/_ref/libmnav_ref.so: the reference's planner, mesh_map and
mesh_layers translation units compiled unmodified by oracle/ref_build/build.sh (against stub lvr2 / rclcpp / tf2
headers) and driven through MeshMap::readMap(), DijkstraMeshPlanner::dijkstra and
CVPMeshPlanner::waveFrontPropagation.  Needs /root/reference (or a prebuilt oracle/_ref); the fixtures themselves
travel with the repository, so neither the CPU oracle tests nor the GPU tests need the reference at run time.
tests/test_golden.py checks the C restatement (oracle/mnav_oracle.c) against them, tests/test_gpu_planners.py the
device."""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from mesh_navigation_amd import meshgen  # noqa: E402
from oracle import ref as R  # noqa: E402
from tests.common import Case  # noqa: E402


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def seed_target(case, fs=(0.1, 0.1), ft=(0.9, 0.9), free=None):
    m = case.mesh
    if free is None:
        return m.vertex_at(*fs), m.vertex_at(*ft)
    def near(f):
        v = m.vertex_at(*f)
        return int(free[((m.xyz[free, :2] - m.xyz[v, :2]) ** 2).sum(1).argmin()])
    return near(fs), near(ft)


def propagation_code(c):
    """waveFrontPropagation returns one code for propagation + back-tracking; a back-tracking failure
    (cvp_mesh_planner.cpp:939,946) means the propagation itself had succeeded (:912-918 passed)."""
    return 0 if c.message.startswith("Could not find a valid path, while back-tracking") else c.code


def check_lookup(case, rm, pos, face):
    """the reference's own getContainingFace must land on the face the fixture means"""
    f, _ = rm.containing_face(pos)
    assert f == face, (f, face)
    of, _ = case.om.containing_face(pos)
    assert of == face, (of, face)


def plan_record(case, rm, seed, target, prefix, out):
    m = case.mesh
    r = rm.dijkstra(m.xyz[seed], m.xyz[target])
    off = np.array([0.03, 0.02, 0.0], np.float32)
    sp, tp = m.xyz[seed] + off, m.xyz[target] + off
    sf, _ = rm.containing_face(sp)
    tf, _ = rm.containing_face(tp)
    check_lookup(case, rm, sp, sf); check_lookup(case, rm, tp, tf)
    c = rm.cvp(sp, tp)                      # propagation + back-tracking, step_width 0.4 (the reference default)
    goal_d = lambda res, tv: np.float32(np.float64(res.dist[tv]) + 0.3)
    out.update({
        prefix + "seed_target": np.array([seed, target], np.uint32),
        prefix + "dij_code": np.array([r.code], np.uint32),
        prefix + "dij_path": r.path, prefix + "dij_dist_sha": np.array(sha(r.dist)), prefix + "dij_pred_sha": np.array(sha(r.pred)),
        prefix + "dij_dist_sample": r.dist[:: max(1, m.V // 512)].copy(),
        prefix + "cvp_faces": np.array([sf, tf], np.uint32), prefix + "cvp_seed_pos": sp, prefix + "cvp_target_pos": tp,
        prefix + "cvp_code": np.array([propagation_code(c)], np.uint32), prefix + "cvp_message": np.array(c.message),
        prefix + "cvp_dist_sha": np.array(sha(c.dist)), prefix + "cvp_pred_sha": np.array(sha(c.pred)),
        prefix + "cvp_cutface_sha": np.array(sha(c.cutface)), prefix + "cvp_direction_sha": np.array(sha(c.direction)),
        prefix + "cvp_dist_sample": c.dist[:: max(1, m.V // 512)].copy(),
        prefix + "cvp_path_code": np.array([c.code], np.uint32), prefix + "cvp_path_pos": c.path_pos, prefix + "cvp_path_face": c.path_face,
    })
    return r, c


def main():
    out = {}
    # G1: BASELINE config C1 (224x224 terrain, seed 1, zero costs)
    c1 = Case(meshgen.terrain(224, 0.1, 1))
    rm1 = R.RefMap(c1.mesh.xyz, c1.mesh.faces)
    plan_record(c1, rm1, *seed_target(c1), "c1_", out)
    # G2: small terrain with the config-3 cost stack (the reference's own Steepness, Inflation and
    # AvgCombination layers through its LayerManager), edge_cost_factor 1, full arrays.  repulsive_field is
    # off: with it the reference's meshAhead panics on faces the inflation wave did not reach.
    base = Case(meshgen.terrain(40, 0.1, 3, amplitude=0.8))
    rm2 = R.RefMap(base.mesh.xyz, base.mesh.faces, layers="c3", edge_cost_factor=1.0,
                   extra_params={"mesh_map.inflation.repulsive_field": False})
    costs = rm2.vertex_costs()
    steep, lethal = rm2.layer_costs("steepness")
    infl, _ = rm2.layer_costs("inflation")
    g2 = Case(base.mesh, costs, 1.0)
    free = np.where(costs < 0.5)[0]
    s, t = seed_target(g2, (0.15, 0.15), (0.85, 0.85), free)
    r, c = plan_record(g2, rm2, s, t, "g2_", out)
    out.update({"g2_costs": costs, "g2_weights": rm2.edge_weights(), "g2_dij_dist": r.dist, "g2_dij_pred": r.pred,
                "g2_cvp_dist": c.dist, "g2_cvp_pred": c.pred, "g2_cvp_direction": c.direction, "g2_cvp_cutface": c.cutface,
                "g2_steepness": steep, "g2_lethal": lethal, "g2_inflation": infl,
                "g2_vertex_normals": rm2.vertex_normals(), "g2_edge_distances": rm2.edge_distances()})
    np.savez_compressed(os.path.join(HERE, "planner_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "planner_golden.npz"), os.path.getsize(os.path.join(HERE, "planner_golden.npz")), "bytes")
    ragged()


def face_plan(case, rm, s, t, prefix, out, full):
    """Wave seeded at the centroid of a face of s, robot at the centroid of a face of t."""
    m = case.mesh
    r = rm.dijkstra(m.xyz[s], m.xyz[t])
    first_face = np.full(m.V, -1, np.int64)
    fl = m.faces.ravel()
    first_face[fl[::-1]] = np.arange(fl.size)[::-1] // 3
    sf, tf = int(first_face[s]), int(first_face[t])
    sp = m.xyz[m.faces[sf]].astype(np.float64).mean(axis=0).astype(np.float32)
    tp = m.xyz[m.faces[tf]].astype(np.float64).mean(axis=0).astype(np.float32)
    check_lookup(case, rm, sp, sf); check_lookup(case, rm, tp, tf)
    c = rm.cvp(sp, tp)
    out.update({
        prefix + "seed_target": np.array([s, t], np.uint32), prefix + "cvp_faces": np.array([sf, tf], np.uint32),
        prefix + "cvp_seed_pos": sp, prefix + "cvp_target_pos": tp,
        prefix + "dij_code": np.array([r.code], np.uint32), prefix + "dij_path": r.path,
        prefix + "dij_dist_sha": np.array(sha(r.dist)), prefix + "dij_pred_sha": np.array(sha(r.pred)),
        prefix + "cvp_code": np.array([propagation_code(c)], np.uint32), prefix + "cvp_path_code": np.array([c.code], np.uint32),
        prefix + "cvp_message": np.array(c.message),
        prefix + "cvp_dist_sha": np.array(sha(c.dist)), prefix + "cvp_pred_sha": np.array(sha(c.pred)),
        prefix + "cvp_cutface_sha": np.array(sha(c.cutface)), prefix + "cvp_direction_sha": np.array(sha(c.direction)),
        prefix + "cvp_path_pos": c.path_pos, prefix + "cvp_path_face": c.path_face,
    })
    if full:
        out.update({prefix + "dij_dist": r.dist, prefix + "dij_pred": r.pred, prefix + "cvp_dist": c.dist, prefix + "cvp_pred": c.pred})


def ragged():
    """Second fixture file: the inputs on which pop ORDER decides the result (cascades below the main front)."""
    out = {}
    # G3: terrain with 30 % of the faces punched out and a cut column (holes, two components, face-less vertices)
    mesh = meshgen.punched(64, 0.1, 5, drop=0.3, cut_column=40)
    g3 = Case(mesh)
    deg = np.bincount(mesh.edges.ravel(), minlength=mesh.V)
    s, t = mesh.vertex_at(0.1, 0.2), mesh.vertex_at(0.5, 0.8)
    while deg[s] == 0: s += 1
    while deg[t] == 0: t += 1
    rm3 = R.RefMap(mesh.xyz, mesh.faces)
    face_plan(g3, rm3, s, t, "g3_", out, full=True)
    # G4: random per-vertex costs up to 1.2, edge_cost_factor 1 (most triangles violate the triangle inequality), 2 % invalid
    mesh = meshgen.terrain(96, 0.1, 13)
    rng = np.random.default_rng(3)
    costs = rng.uniform(0, 1.2, mesh.V).astype(np.float32)
    invalid = (rng.uniform(size=mesh.V) < 0.02).astype(np.uint8)
    s, t = mesh.vertex_at(0.1, 0.1), mesh.vertex_at(0.9, 0.9)
    invalid[[s, t]] = 0
    costs[[s, t]] = 0
    g4 = Case(mesh, costs, 1.0, invalid)
    rm4 = R.RefMap(mesh.xyz, mesh.faces, vertex_costs=costs, edge_cost_factor=1.0)
    rm4.set_invalid(invalid)
    assert np.array_equal(rm4.edge_weights().view(np.uint32), g4.weights.view(np.uint32))
    face_plan(g4, rm4, s, t, "g4_", out, full=False)
    out["g4_costs_sha"] = np.array(sha(costs)); out["g4_invalid_sha"] = np.array(sha(invalid)); out["g4_weights_sha"] = np.array(sha(g4.weights))
    path = os.path.join(HERE, "planner_golden_ragged.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
