/*
 * mnav_oracle.h -- CPU restatement of the mesh_navigation wavefront planners.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library, and there only as the checker / reported CPU
 * baseline -- never as the thing measured or shipped.
 *
 * PARITY PINNED AGAINST THE REFERENCE'S OWN CODE: oracle/ref_build/build.sh
 * compiles the reference's planner, mesh_map and mesh_layers translation
 * units unmodified (against stub lvr2/rclcpp/tf2 headers) into
 * oracle/_ref/libmnav_ref.so, and tests/test_ref_pins_oracle.py asserts that
 * this restatement returns the same bits (potential, predecessors, path,
 * cutting faces, directions, layer costs, edge weights) on the seeded
 * configurations; tests/golden/ fixtures are generated from that library.
 * What stays a CONVENTION is what lives inside the un-vendored lvr2 / pmp
 * (modelled in oracle/ref_build/stubs/lvr2_stub.hpp): heap behaviour among
 * EQUAL keys, BaseVector::rotated.  The reference's own known-answer test
 * (mesh_layers/test/inflation_layer_test.cpp:38-100) is run unmodified by the
 * same build and replayed in tests/test_oracle_kat.py.
 *
 * Every function cites the reference file:line it restates (paths relative
 * to the reference checkout).  Conventions the reference delegates to the
 * un-vendored dependency lvr2@main (source_dependencies.yaml:4-7) are marked
 * CONVENTION and documented in DESIGN.md.
 */
#ifndef MNAV_ORACLE_H
#define MNAV_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* MBF GetPath result codes (dijkstra_mesh_planner.h:72-85) */
enum {
  MO_SUCCESS = 0,
  MO_CANCELED = 51,
  MO_INVALID_START = 52,
  MO_INVALID_GOAL = 53,
  MO_NO_PATH_FOUND = 54,
};

typedef struct mo_mesh mo_mesh;

/* Build topology from an indexed triangle list.
 * CONVENTION (lvr2 absent): vertex ids and face ids/vertex order are the
 * caller's; undirected edge ids are assigned in order of first appearance
 * while iterating faces 0..F-1 and, inside a face, the sides (v0,v1), (v1,v2),
 * (v2,v0) -- which is what pmp::SurfaceMesh::add_face does; edges/faces around a
 * vertex are enumerated in the half-edge circulator order of lvr2::PMPMesh
 * (counter-clockwise from the vertex's outgoing half-edge after adding the
 * faces 0..F-1 in order; he_add_triangle in the .c file). */
mo_mesh* mo_mesh_create(uint32_t V, uint32_t F, const float* xyz, const uint32_t* faces);
void mo_mesh_destroy(mo_mesh* m);
uint32_t mo_mesh_num_vertices(const mo_mesh* m);
uint32_t mo_mesh_num_faces(const mo_mesh* m);
uint32_t mo_mesh_num_edges(const mo_mesh* m);
/* 0 if pmp would reject a face of the list (complex vertex / edge): the reference's loader then discards
 * faces and re-indexes (mesh_map.cpp:276-300); incidence rows stay in ascending id order */
int mo_mesh_is_manifold(const mo_mesh* m);
/* getFacesOfVertex / getEdgesOfVertex orders as CSR (V+1 row pointers, 3F / 2E entries) */
void mo_mesh_vertex_faces(const mo_mesh* m, uint32_t* vf_ptr, uint32_t* vf);
void mo_mesh_vertex_edges(const mo_mesh* m, uint32_t* ve_ptr, uint32_t* ve);
/* out: E*2 vertex ids per undirected edge */
void mo_mesh_edges(const mo_mesh* m, uint32_t* edge_vtx);
/* out: F*3, face_edges[f*3+k] = edge between face vertex k and (k+1)%3 */
void mo_mesh_face_edges(const mo_mesh* m, uint32_t* face_edges);

/* lvr2::calcVertexDistances stand-in (mesh_map.cpp:414): float32 Euclidean
 * length of every edge. */
void mo_edge_distances(const mo_mesh* m, float* edge_dist);
/* lvr2::calcFaceNormals / calcVertexNormals stand-ins (mesh_map.cpp:351,374).
 * CONVENTION: face normal = normalize((b-a)x(c-a)); vertex normal =
 * normalize(sum of unit normals of incident faces). */
void mo_face_normals(const mo_mesh* m, float* fn);
void mo_vertex_normals(const mo_mesh* m, const float* fn, float* vn);

/* MeshMap::computeEdgeWeights, mesh_map/src/mesh_map.cpp:517-561 */
void mo_compute_edge_weights(const mo_mesh* m, const float* edge_dist, const float* vertex_costs,
                             double edge_cost_factor, float* edge_weights);

/* SteepnessLayer::computeLayer, mesh_layers/src/steepness_layer.cpp:157-166,
 * computeLethals :82-93.  lethal[v]=1 where steepness > threshold. */
void mo_steepness(const mo_mesh* m, const float* vertex_normals, double threshold, float* steepness,
                  uint8_t* lethal);

typedef struct {
  double inscribed_radius;   /* 0.25 */
  double inflation_radius;   /* 0.4  */
  double lethal_value;       /* 1.0  */
  double inscribed_value;    /* 0.99 */
  double cost_scaling_factor;/* 1.0  */
} mo_inflation_cfg;          /* inflation_layer.h:240-248 */

/* InflationLayer::computeUpdateSethianMethod, inflation_layer.cpp:181-234 */
float mo_inflation_sethian(float d1, float d2, float a, float b, float dot, float F);
/* InflationLayer::fading, inflation_layer.cpp:315-339 */
float mo_inflation_fading(const mo_inflation_cfg* cfg, float distance);
/* InflationLayer::waveFrontUpdate on a single face (inflation_layer.cpp:236-313);
 * dist has V entries, +inf == "no value".  Returns the bool of the reference. */
int mo_inflation_wavefront_update(const mo_mesh* m, float* dist, float* vecmap /*V*3 or NULL*/,
                                  float max_distance, const float* edge_weights, uint32_t v1,
                                  uint32_t v2, uint32_t v3);
/* InflationLayer::waveCostInflation, inflation_layer.cpp:341-491.
 * lethal: V flags.  Outputs: riskiness cost (V; vertices never reached hold the
 * layer default 0, inflation_layer.h:74-77), distances (V, +inf where unset),
 * vecmap (V*3, zero where unset). */
void mo_inflation(const mo_mesh* m, const mo_inflation_cfg* cfg, const uint8_t* lethal,
                  const uint8_t* invalid, const float* edge_dist, float* cost_out, float* dist_out,
                  float* vec_out);

/* MaxCombinationLayer::computeLayer combination_layer.cpp:44-85 (mode 0) and
 * AvgCombinationLayer::computeLayer :185-248 (mode 1; weights w[i]). */
void mo_combine(uint32_t V, int mode, int n_layers, const float* const* layers, const float* weights,
                float* out);

/* lvr2::Meap<VertexHandle,float> emulation hooks (CONVENTION, see .c) exposed
 * for unit tests. */
typedef struct mo_meap mo_meap;
/* tie rule of the heap: 1 = (value, vertex id) [default, the device path's rule], 0 = plain lvr2-style array heap */
void mo_set_heap_ties_by_id(int on);
mo_meap* mo_meap_create(uint32_t capacity);
void mo_meap_destroy(mo_meap* h);
void mo_meap_insert(mo_meap* h, uint32_t key, float value);
int mo_meap_empty(const mo_meap* h);
uint32_t mo_meap_pop_min(mo_meap* h, float* value);

typedef struct {
  uint64_t fixed_set_cnt; /* popped vertices (dijkstra_mesh_planner.cpp:285,291) */
  uint64_t expanded;      /* popped vertices that passed the cut-offs */
  uint64_t relaxations;   /* successful strict-< updates */
  uint64_t edge_visits;   /* directed edges / face corners looked at */
  float goal_dist;        /* armed value or +inf */
  double t_init_ms, t_propagation_ms, t_backtrack_ms;
} mo_stats;

/* DijkstraMeshPlanner::dijkstra (7-arg), dijkstra_mesh_planner.cpp:217-398.
 * seed_vertex = wave seed (nav goal), target_vertex = robot vertex.
 * dist/pred: V each.  path: capacity V (reference path list order, i.e. before
 * makePlan's reverse: seed ... pred[target]); path_len out.
 * cancel: optional pointer polled each pop (may be NULL). */
uint32_t mo_dijkstra(const mo_mesh* m, const float* edge_weights, const float* vertex_costs,
                     uint8_t* invalid, uint32_t seed_vertex, uint32_t target_vertex,
                     double goal_dist_offset, double cost_limit, float* dist, uint32_t* pred,
                     uint32_t* path, uint32_t* path_len, const volatile int* cancel,
                     mo_stats* stats);

/* Predecessors by the documented deterministic rule (DESIGN.md "tie rule"):
 * pred[v] = argmin over expanded neighbours u with dist[u]+w(u,v)==dist[v] of
 * (dist[u], u).  Used to count where the emulated-Meap order differs. */
void mo_dijkstra_pred_rule(const mo_mesh* m, const float* edge_weights, const float* vertex_costs,
                           const uint8_t* invalid, uint32_t seed_vertex, float goal_dist,
                           double cost_limit, const float* dist, uint32_t* pred);

/* DijkstraMeshPlanner::computeVectorMap, dijkstra_mesh_planner.cpp:189-209.
 * vecmap V*3, rows of vertices with pred==self are left untouched. */
void mo_dijkstra_vector_map(const mo_mesh* m, const uint32_t* pred, float* vecmap);

/* CVPMeshPlanner::waveFrontUpdate, cvp_mesh_planner.cpp:369-556, on plain
 * numbers (for KATs).  in: u1,u2,u3 and side weights a,b,c.  out: new u3
 * (float), which (1: pred=v1, 2: pred=v2), direction.  returns reference bool. */
int mo_cvp_update_scalar(float u1, float u2, float u3, float a, float b, float c, float* u3_out,
                         int* pred_sel, float* direction);

/* CVPMeshPlanner::waveFrontPropagation (8-arg) cvp_mesh_planner.cpp:651-918,
 * i.e. everything up to (not including) the vector-field back-tracking.
 * seed_pos = exact wave seed position (nav goal), seed_face / target_face the
 * containing faces.  direction / cutface persist across calls in the reference
 * (cvp_mesh_planner.cpp:179,698-705): caller passes the arrays in and out
 * (cutface: 0xFFFFFFFF = none).  vecmap V*3 (cleared to 0 here, like
 * vector_map_.clear()), has_vec V flags. */
uint32_t mo_cvp_propagate(const mo_mesh* m, const float* edge_weights, const float* vertex_costs,
                          uint8_t* invalid, const float* vertex_normals, const float seed_pos[3],
                          uint32_t seed_face, uint32_t target_face, double goal_dist_offset,
                          double cost_limit, float* dist, uint32_t* pred, float* direction,
                          uint32_t* cutface, float* vecmap, uint8_t* has_vec,
                          const volatile int* cancel, mo_stats* stats);

/* Optional per-layer vector field for MeshMap::meshAhead (mesh_map.cpp:1096-1102):
 * InflationLayer::vectorAt(handles, bary), inflation_layer.cpp:493-521. */
typedef struct {
  const float* distances; /* V, +inf unset */
  const float* vecmap;    /* V*3 */
  mo_inflation_cfg cfg;
  int repulsive_field;
} mo_inflation_field;

/* InflationLayer::vectorAt(handles, barycentric coords), inflation_layer.cpp:493-521 */
void mo_inflation_vector_at(const mo_inflation_field* L, const uint32_t vs[3], const float bary[3], float out[3]);

/* CVP vector-field back-tracking, cvp_mesh_planner.cpp:920-951 with
 * MeshMap::meshAhead mesh_map.cpp:1070-1108.  path_pos: cap*3, path_face: cap
 * (reference list order before makePlan's reverse: seed first, target last).
 * returns MBF code; path_len out. */
uint32_t mo_cvp_backtrack(const mo_mesh* m, const float* vecmap, const uint8_t* has_vec,
                          const mo_inflation_field* infl /*NULL ok*/, const float seed_pos[3],
                          uint32_t seed_face, const float target_pos[3], uint32_t target_face,
                          double step_width, uint32_t cap, float* path_pos, uint32_t* path_face,
                          uint32_t* path_len);

/* MeshMap::getNearestVertexHandle mesh_map.cpp:1161-1174 (brute force stand-in
 * for nanoflann 1-NN; ties -> lowest id).  returns 0xFFFFFFFF if V==0. */
uint32_t mo_nearest_vertex(const mo_mesh* m, const float p[3]);
/* MeshMap::searchContainingFace mesh_map.cpp:1120-1159; returns face or
 * 0xFFFFFFFF; bary out (3). */
uint32_t mo_containing_face(const mo_mesh* m, const float p[3], float bary[3]);
/* mesh_map::projectedBarycentricCoords util.cpp:320-347 */
int mo_projected_barycentric(const float p[3], const float a[3], const float b[3], const float c[3],
                             float bary[3], float* dist);

/* mesh_map::calculatePoseFromPosition util.cpp:292-298 + calculatePoseFromDirection
 * :267-284.  pose out: x y z qx qy qz qw (doubles).  returns dir length. */
float mo_pose_from_position(const float current[3], const float next[3], const float normal[3],
                            double pose[7]);

/* DijkstraMeshPlanner::makePlan pose assembly, dijkstra_mesh_planner.cpp:83-116.
 * path = dijkstra() order (not yet reversed).  poses: (path_len+1)*7. returns
 * number of poses, cost out. */
uint32_t mo_dijkstra_poses(const mo_mesh* m, const float* vertex_normals, const uint32_t* path,
                           uint32_t path_len, const float robot_pos[3], const float goal_pos[3],
                           double* poses, double* cost);
/* CVPMeshPlanner::makePlan pose assembly, cvp_mesh_planner.cpp:93-124 */
uint32_t mo_cvp_poses(const mo_mesh* m, const float* face_normals, const float* path_pos,
                      const uint32_t* path_face, uint32_t path_len, const double goal_pose[7],
                      double* poses, double* cost);

#ifdef __cplusplus
}
#endif
#endif
