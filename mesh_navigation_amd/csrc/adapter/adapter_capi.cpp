// adapter_capi.cpp -- tiny C harness around the C++ planner adapter so that the (Python) parity tests
// can drive makePlan()/cancel()/initialize() exactly like mbf_mesh_nav does
// (mbf_mesh_nav/src/mesh_navigation_server.cpp:185-212, mesh_planner_execution.cpp:55-66).
#include <cstring>
#include <memory>
#include <string>

#include "gpu_mesh_planners.h"

struct mnav_adapter_planner {
  std::shared_ptr<mesh_map::MeshMap> map;
  rclcpp::Node::SharedPtr node;
  std::shared_ptr<mbf_mesh_core::MeshPlanner> planner;
  bool is_cvp = false;
  std::string message;
};

extern "C" {

// kind: 0 = "dijkstra_mesh_planner/DijkstraMeshPlanner", 1 = "cvp_mesh_planner/CVPMeshPlanner"
mnav_adapter_planner* mnav_adapter_create(int kind, uint32_t V, uint32_t F, uint32_t E, const float* xyz, const uint32_t* faces,
                                          const uint32_t* edges, const float* vertex_normals, const float* face_normals,
                                          const float* vertex_costs, const float* edge_weights, const uint8_t* invalid,
                                          double goal_dist_offset, double cost_limit, double step_width)
{
  auto* a = new mnav_adapter_planner();
  a->map = std::make_shared<mesh_map::MeshMap>();
  mesh_map::MeshMap& m = *a->map;
  m.positions.assign(xyz, xyz + 3 * (size_t)V);
  m.faces.assign(faces, faces + 3 * (size_t)F);
  m.edges.assign(edges, edges + 2 * (size_t)E);
  m.vertex_normals.assign(vertex_normals, vertex_normals + 3 * (size_t)V);
  m.face_normals.assign(face_normals, face_normals + 3 * (size_t)F);
  m.vertex_costs.assign(vertex_costs, vertex_costs + V);
  m.edge_weights.assign(edge_weights, edge_weights + E);
  if (invalid) m.invalid.assign(invalid, invalid + V);
  m.finalize();
  a->node = std::make_shared<rclcpp::Node>();
  const std::string name = kind ? "cvp_mesh_planner" : "dijkstra_mesh_planner";
  a->node->set_override(name + ".goal_dist_offset", goal_dist_offset);
  a->node->set_override(name + ".cost_limit", cost_limit);
  a->node->set_override(name + ".step_width", step_width);
  a->is_cvp = kind != 0;
  if (kind) a->planner = std::make_shared<cvp_mesh_planner::CVPMeshPlanner>();
  else a->planner = std::make_shared<dijkstra_mesh_planner::DijkstraMeshPlanner>();
  if (!a->planner->initialize(name, a->map, a->node)) { delete a; return nullptr; }
  return a;
}

void mnav_adapter_destroy(mnav_adapter_planner* a) { delete a; }

// poses: x y z qx qy qz qw per pose.  Returns the MBF code; *n_poses = number of poses produced.
uint32_t mnav_adapter_make_plan(mnav_adapter_planner* a, const double start_pose[7], const double goal_pose[7], double* poses,
                                uint32_t cap, uint32_t* n_poses, double* cost, char* message, uint32_t message_cap)
{
  geometry_msgs::msg::PoseStamped s, g;
  auto fill = [](geometry_msgs::msg::PoseStamped& p, const double* v) {
    p.header.frame_id = "map";
    p.pose.position.x = v[0]; p.pose.position.y = v[1]; p.pose.position.z = v[2];
    p.pose.orientation.x = v[3]; p.pose.orientation.y = v[4]; p.pose.orientation.z = v[5]; p.pose.orientation.w = v[6];
  };
  fill(s, start_pose); fill(g, goal_pose);
  std::vector<geometry_msgs::msg::PoseStamped> plan;
  double c = 0;
  std::string msg;
  const uint32_t code = a->planner->makePlan(s, g, 0.0, plan, c, msg);
  *n_poses = (uint32_t)plan.size();
  for (uint32_t i = 0; i < plan.size() && i < cap; ++i) {
    const auto& p = plan[i].pose;
    double* o = poses + 7 * (size_t)i;
    o[0] = p.position.x; o[1] = p.position.y; o[2] = p.position.z;
    o[3] = p.orientation.x; o[4] = p.orientation.y; o[5] = p.orientation.z; o[6] = p.orientation.w;
  }
  *cost = c;
  if (message && message_cap) { std::strncpy(message, msg.c_str(), message_cap - 1); message[message_cap - 1] = 0; }
  return code;
}

int mnav_adapter_cancel(mnav_adapter_planner* a) { return a->planner->cancel() ? 1 : 0; }

// change counter of the map's cost arrays (0 = unknown: the device mirror hashes them on every plan)
void mnav_adapter_set_cost_version(mnav_adapter_planner* a, uint64_t version) { a->map->cost_version = version; }

// V-sized results of the last plan, fetched from the device on demand; what: 0 potential (float V), 1 predecessors
// (uint32 V; Dijkstra), 4 vector map (float V*3; Dijkstra)
int mnav_adapter_fetch(mnav_adapter_planner* a, int what, void* out)
{
  const uint32_t V = a->map->V;
  if (a->is_cvp) {
    auto* p = static_cast<cvp_mesh_planner::CVPMeshPlanner*>(a->planner.get());
    if (what != 0) return -1;
    std::memcpy(out, p->potential().data(), 4 * (size_t)V);
    return 0;
  }
  auto* p = static_cast<dijkstra_mesh_planner::DijkstraMeshPlanner*>(a->planner.get());
  if (what == 0) std::memcpy(out, p->potential().data(), 4 * (size_t)V);
  else if (what == 1) std::memcpy(out, p->predecessors().data(), 4 * (size_t)V);
  else if (what == 4) std::memcpy(out, p->getVectorMap().data(), 12 * (size_t)V);
  else return -1;
  return 0;
}

static void fill_layer_field(mesh_map::MeshMap::LayerVectorField& L, uint32_t V, const float* distances, const uint8_t* has_distance,
                             const float* vectors, const uint8_t* has_vector, double inscribed_radius, double inflation_radius,
                             double lethal_value, double inscribed_value, int repulsive_field)
{
  L.distances.assign(distances, distances + V); L.has_distance.assign(has_distance, has_distance + V);
  L.vectors.assign(vectors, vectors + 3 * (size_t)V); L.has_vector.assign(has_vector, has_vector + V);
  L.inscribed_radius = inscribed_radius; L.inflation_radius = inflation_radius; L.lethal_value = lethal_value;
  L.inscribed_value = inscribed_value; L.repulsive_field = repulsive_field != 0;
}

// a layer's repulsive vector field (InflationLayer distances_ / vector_map_) for MeshMap::meshAhead (mesh_map.cpp:1099-1102)
void mnav_adapter_add_layer_field(mnav_adapter_planner* a, const float* distances, const uint8_t* has_distance, const float* vectors,
                                  const uint8_t* has_vector, double inscribed_radius, double inflation_radius, double lethal_value,
                                  double inscribed_value, int repulsive_field)
{
  a->map->layer_fields.emplace_back();
  fill_layer_field(a->map->layer_fields.back(), a->map->V, distances, has_distance, vectors, has_vector, inscribed_radius, inflation_radius,
                   lethal_value, inscribed_value, repulsive_field);
}

// update the map's cost arrays in place (what a layer change does, mesh_map.cpp:454-493)
void mnav_adapter_set_costs(mnav_adapter_planner* a, const float* vertex_costs, const float* edge_weights)
{
  a->map->vertex_costs.assign(vertex_costs, vertex_costs + a->map->V);
  a->map->edge_weights.assign(edge_weights, edge_weights + a->map->E);
}

// Host-only geometry entry points (no device involved) so the CPU test-suite can check the adapter's
// MeshMap stand-in (nearest vertex, containing face, vector-field back-tracking) on its own.
uint32_t mnav_adapter_host_nearest_vertex(uint32_t V, uint32_t F, const float* xyz, const uint32_t* faces, const float p[3])
{
  mesh_map::MeshMap m;
  m.positions.assign(xyz, xyz + 3 * (size_t)V); m.faces.assign(faces, faces + 3 * (size_t)F);
  m.finalize();
  return m.getNearestVertexHandle(mesh_map::Vector(p[0], p[1], p[2]));
}

uint32_t mnav_adapter_host_containing_face(uint32_t V, uint32_t F, const float* xyz, const uint32_t* faces, const float p[3])
{
  mesh_map::MeshMap m;
  m.positions.assign(xyz, xyz + 3 * (size_t)V); m.faces.assign(faces, faces + 3 * (size_t)F);
  m.finalize();
  return m.getContainingFace(mesh_map::Vector(p[0], p[1], p[2]), 0.4f);
}

// cvp_mesh_planner.cpp:920-951 on a given vector field; path in reference list order (seed first)
static uint32_t host_backtrack(mesh_map::MeshMap& m, uint32_t V, const float* vecmap, const uint8_t* has_vec, const float seed_pos[3],
                               uint32_t seed_face, const float target_pos[3], uint32_t target_face, double step_width, uint32_t cap,
                               float* path_pos, uint32_t* path_face, uint32_t* path_len);

// the same with one layer vector field (Inflation) added in meshAhead; code 54 + *panicked = 1 where the reference's
// attribute-map lookup panics
uint32_t mnav_adapter_host_backtrack_layer(uint32_t V, uint32_t F, const float* xyz, const uint32_t* faces, const float* vecmap,
                                           const uint8_t* has_vec, const float seed_pos[3], uint32_t seed_face, const float target_pos[3],
                                           uint32_t target_face, double step_width, const float* distances, const uint8_t* has_distance,
                                           const float* vectors, const uint8_t* has_vector, double inscribed_radius, double inflation_radius,
                                           double lethal_value, double inscribed_value, int repulsive_field, uint32_t cap, float* path_pos,
                                           uint32_t* path_face, uint32_t* path_len, int* panicked)
{
  mesh_map::MeshMap m;
  m.positions.assign(xyz, xyz + 3 * (size_t)V); m.faces.assign(faces, faces + 3 * (size_t)F);
  m.finalize();
  m.layer_fields.emplace_back();
  fill_layer_field(m.layer_fields.back(), V, distances, has_distance, vectors, has_vector, inscribed_radius, inflation_radius, lethal_value,
                   inscribed_value, repulsive_field);
  *panicked = 0;
  try {
    return host_backtrack(m, V, vecmap, has_vec, seed_pos, seed_face, target_pos, target_face, step_width, cap, path_pos, path_face, path_len);
  } catch (const mesh_map::MeshMap::MapPanic&) { *panicked = 1; *path_len = 0; return 54; }
}

uint32_t mnav_adapter_host_backtrack(uint32_t V, uint32_t F, const float* xyz, const uint32_t* faces, const float* vecmap,
                                     const uint8_t* has_vec, const float seed_pos[3], uint32_t seed_face, const float target_pos[3],
                                     uint32_t target_face, double step_width, uint32_t cap, float* path_pos, uint32_t* path_face,
                                     uint32_t* path_len)
{
  mesh_map::MeshMap m;
  m.positions.assign(xyz, xyz + 3 * (size_t)V); m.faces.assign(faces, faces + 3 * (size_t)F);
  m.finalize();
  return host_backtrack(m, V, vecmap, has_vec, seed_pos, seed_face, target_pos, target_face, step_width, cap, path_pos, path_face, path_len);
}

static uint32_t host_backtrack(mesh_map::MeshMap& m, uint32_t V, const float* vecmap, const uint8_t* has_vec, const float seed_pos[3],
                               uint32_t seed_face, const float target_pos[3], uint32_t target_face, double step_width, uint32_t cap,
                               float* path_pos, uint32_t* path_face, uint32_t* path_len)
{
  m.setVectorMap(std::vector<float>(vecmap, vecmap + 3 * (size_t)V), std::vector<uint8_t>(has_vec, has_vec + V));
  const mesh_map::Vector start(seed_pos[0], seed_pos[1], seed_pos[2]);
  mesh_map::Vector pos(target_pos[0], target_pos[1], target_pos[2]);
  uint32_t face = target_face;
  std::vector<std::pair<mesh_map::Vector, uint32_t>> rev;
  rev.push_back({ pos, face });
  uint32_t code = 0;
  while (pos.distance2(start) > step_width) {
    if (m.meshAhead(pos, face, (float)step_width)) rev.push_back({ pos, face });
    else { code = 54; break; }
    if (rev.size() >= cap) { code = 54; break; }
  }
  if (code == 0) rev.push_back({ start, seed_face });
  uint32_t n = 0;
  for (size_t i = rev.size(); i-- > 0 && n < cap; ++n) {
    path_pos[3 * n] = rev[i].first.x; path_pos[3 * n + 1] = rev[i].first.y; path_pos[3 * n + 2] = rev[i].first.z;
    path_face[n] = rev[i].second;
  }
  *path_len = n;
  return code;
}

}  // extern "C"
