// gpu_mesh_planners.cpp -- see gpu_mesh_planners.h.  Reference line numbers in the comments.
// The host half of makePlan (poses along the path, the vector-field back-tracking, the device call) is
// include/mnav_planner_host.hpp, shared with the ROS 2 plugin package; this file binds it to the ROS-free host map.
#include "gpu_mesh_planners.h"

#include "mnav_planner_host.hpp"

#include <cmath>
#include <cstring>

using geometry_msgs::msg::PoseStamped;
typedef mbf_msgs::action::GetPath::Result Result;

namespace mnav_adapter {

// 64-bit content hash over 8-byte words (multiply-xorshift mixing): ~2 ms for the 17 MB of a 1M-vertex map, against
// the 15-20 ms a byte-serial hash took -- and not needed at all when the map carries a change counter
static uint64_t hash_words(const void* p, size_t n, uint64_t h)
{
  const unsigned char* b = static_cast<const unsigned char*>(p);
  size_t i = 0;
  for (; i + 8 <= n; i += 8) {
    uint64_t w; std::memcpy(&w, b + i, 8);
    h = (h ^ w) * 0x9E3779B97F4A7C15ull;
    h ^= h >> 32;
  }
  uint64_t tail = 0;
  if (i < n) std::memcpy(&tail, b + i, n - i);
  h = (h ^ tail ^ (uint64_t)n) * 0x9E3779B97F4A7C15ull;
  return h ^ (h >> 29);
}

MeshMapDevice::MeshMapDevice(int device) { ctx_ = mnav_create(device); }
MeshMapDevice::~MeshMapDevice() { if (ctx_) mnav_destroy(ctx_); }

bool MeshMapDevice::sync(const mesh_map::MeshMap& map, std::string& err)
{
  if (!ctx_) { err = "no MI355X device context (mnav_create failed)"; return false; }
  if (uploaded_ != &map) {
    if (mnav_upload_mesh(ctx_, map.V, map.F, map.E, map.positions.data(), map.faces.data(), map.edges.data(),
                         map.vertex_normals.empty() ? nullptr : map.vertex_normals.data()) != 0) { err = mnav_last_error(ctx_); return false; }
    uploaded_ = &map; cost_hash_ = 0; have_costs_ = false;
    (void)mnav_set_resident_outputs(ctx_, 1);                        // V-sized outputs stay in HBM until somebody asks for them
  }
  // the reference re-reads vertex_costs / edge_weights on every plan; re-upload only when they changed: by the
  // map's change counter when the integration maintains one, else by content hash
  uint64_t h;
  if (map.cost_version != 0) h = map.cost_version | (1ull << 63);
  else {
    h = hash_words(map.vertex_costs.data(), map.vertex_costs.size() * 4, 0x243F6A8885A308D3ull);
    h = hash_words(map.edge_weights.data(), map.edge_weights.size() * 4, h);
    h = hash_words(map.invalid.data(), map.invalid.size(), h);
    h &= ~(1ull << 63);
  }
  if (h != cost_hash_ || !have_costs_) {
    if (mnav_upload_costs(ctx_, map.vertex_costs.data(), map.edge_weights.data(), map.invalid.data()) != 0) { err = mnav_last_error(ctx_); return false; }
    cost_hash_ = h; have_costs_ = true;
  }
  return true;
}
}  // namespace mnav_adapter

// =============================================================================================
namespace dijkstra_mesh_planner {

// dijkstra_mesh_planner.cpp:55-134
uint32_t DijkstraMeshPlanner::makePlan(const PoseStamped& start, const PoseStamped& goal, double /*tolerance*/,
                                       std::vector<PoseStamped>& plan, double& cost, std::string& /*message*/)
{
  const PoseStamped start_in_map = mesh_map_->transformToMapFrame(start);     // :63
  const PoseStamped goal_in_map = mesh_map_->transformToMapFrame(goal);       // :64
  mesh_map::Vector start_vec = mesh_map::toVector(start_in_map.pose.position);   // :72
  const mesh_map::Vector goal_vec = mesh_map::toVector(goal_in_map.pose.position);   // :73
  std::list<uint32_t> path;
  const uint32_t outcome = dijkstra(goal_vec, start_vec, path);               // :81 the wave starts at the goal pose
  path.reverse();                                                             // :83
  std_msgs::msg::Header header;
  header.stamp = node_ ? node_->now() : builtin_interfaces::msg::Time();
  header.frame_id = mesh_map_->mapFrame();                                    // :87
  PoseStamped stamped;
  stamped.header = header;
  mnav_host::vertex_path_poses(path, start_vec, goal_vec, stamped,           // :89-116
                               [&](uint32_t vH) { return mesh_map_->vertex(vH); }, [&](uint32_t vH) { return mesh_map_->vertexNormal(vH); },
                               [](const mesh_map::Vector& from, const mesh_map::Vector& to, const mesh_map::Normal& up, float& len) {
                                 return mesh_map::calculatePoseFromPosition(from, to, up, len);
                               },
                               plan, cost);
  // :118-131 publishing (path, "Potential" vertex costs, vector field) is ROS I/O and out of scope here
  return outcome;
}

bool DijkstraMeshPlanner::cancel()                                            // :136-140
{
  cancel_planning_ = true;
  if (dev_ && dev_->ok()) mnav_cancel(dev_->ctx());
  return true;
}

bool DijkstraMeshPlanner::initialize(const std::string& plugin_name, const std::shared_ptr<mesh_map::MeshMap>& mesh_map_ptr,
                                     const rclcpp::Node::SharedPtr& node)      // :142-169
{
  mesh_map_ = mesh_map_ptr;
  name_ = plugin_name;
  map_frame_ = mesh_map_->mapFrame();
  node_ = node;
  if (node_) {
    config_.publish_vector_field = node_->declare_parameter(name_ + ".publish_vector_field", config_.publish_vector_field);   // :149
    config_.publish_face_vectors = node_->declare_parameter(name_ + ".publish_face_vectors", config_.publish_face_vectors);   // :150
    config_.goal_dist_offset = node_->declare_parameter(name_ + ".goal_dist_offset", config_.goal_dist_offset);               // :151
    config_.cost_limit = node_->declare_parameter(name_ + ".cost_limit", config_.cost_limit);                                 // :159
  }
  dev_.reset(new mnav_adapter::MeshMapDevice(0));     // CSR + float32 arrays are built and uploaded once
  std::string err;
  return dev_->ok() && dev_->sync(*mesh_map_, err);
}

// dijkstra_mesh_planner.cpp:217-398 with the loop :287-348 on the device
uint32_t DijkstraMeshPlanner::dijkstra(const mesh_map::Vector& original_start, const mesh_map::Vector& original_goal,
                                       std::list<uint32_t>& path)
{
  const uint32_t start_vertex = mesh_map_->getNearestVertexHandle(original_start);   // :235
  const uint32_t goal_vertex = mesh_map_->getNearestVertexHandle(original_goal);     // :236
  cancel_planning_ = false;                                                   // :238
  if (start_vertex == mesh_map::kNoHandle) return Result::INVALID_START;      // :240
  if (goal_vertex == mesh_map::kNoHandle) return Result::INVALID_GOAL;        // :242
  path.clear();
  std::string err;
  if (!dev_ || !dev_->sync(*mesh_map_, err)) return Result::INTERNAL_ERROR;
  fields_on_host_ = false;
  // potential, predecessors and the vector map (computeVectorMap :380) are computed and kept on the device; only
  // the vertex path comes back.  They are fetched when somebody reads them (potential(), the controller's
  // MeshMap::getVectorMap, publishing) -- fetchFields().
  std::vector<uint32_t> ids;
  const uint32_t code = mnav_host::dijkstra_vertex_path(dev_->ctx(), start_vertex, goal_vertex, config_.goal_dist_offset, config_.cost_limit,
                                                       mesh_map_->V, ids);
  if (code != Result::SUCCESS) return code;
  path.assign(ids.begin(), ids.end());                                        // :367-373 list order: seed first
  if (config_.publish_vector_field || eager_fields_) fetchFields();          // :126-129
  return Result::SUCCESS;
}

// V-sized results of the last plan, on demand: 20 MB over PCIe at 1M vertices, O(V) host work
void DijkstraMeshPlanner::fetchFields()
{
  if (fields_on_host_ || !dev_ || !dev_->ok()) return;
  const uint32_t V = mesh_map_->V;
  potential_.assign(V, 0.f); predecessors_.assign(V, 0u); vector_map_.assign((size_t)V * 3, 0.f);
  if (mnav_download_output(dev_->ctx(), 0, 0, potential_.data()) != 0 || mnav_download_output(dev_->ctx(), 0, 1, predecessors_.data()) != 0 ||
      mnav_download_output(dev_->ctx(), 0, 4, vector_map_.data()) != 0) return;
  std::vector<uint8_t> set(V, 0);
  for (uint32_t v = 0; v < V; ++v) set[v] = predecessors_[v] != v;            // :197
  mesh_map_->setVectorMap(vector_map_, set);                                  // :208
  fields_on_host_ = true;
}
}  // namespace dijkstra_mesh_planner

// =============================================================================================
namespace cvp_mesh_planner {

// cvp_mesh_planner.cpp:62-140
uint32_t CVPMeshPlanner::makePlan(const PoseStamped& start, const PoseStamped& goal, double /*tolerance*/,
                                  std::vector<PoseStamped>& plan, double& cost, std::string& message)
{
  const PoseStamped start_in_map = mesh_map_->transformToMapFrame(start);     // :72
  const PoseStamped goal_in_map = mesh_map_->transformToMapFrame(goal);       // :73
  const mesh_map::Vector start_vec = mesh_map::toVector(start_in_map.pose.position);   // :82
  const mesh_map::Vector goal_vec = mesh_map::toVector(goal_in_map.pose.position);     // :83
  std::list<std::pair<mesh_map::Vector, uint32_t>> path;
  const uint32_t outcome = waveFrontPropagation(goal_vec, start_vec, path, message);   // :89
  path.reverse();                                                             // :93
  std_msgs::msg::Header header;
  header.stamp = node_ ? node_->now() : builtin_interfaces::msg::Time();
  header.frame_id = mesh_map_->mapFrame();
  PoseStamped stamped;
  stamped.header = header;
  mnav_host::face_path_poses(path, cancel_planning_, goal_in_map.pose, stamped,   // :99-124
                             [&](uint32_t fH) { return mesh_map_->faceNormal(fH); },
                             [](const mesh_map::Vector& from, const mesh_map::Vector& to, const mesh_map::Normal& up, float& len) {
                               return mesh_map::calculatePoseFromPosition(from, to, up, len);
                             },
                             plan, cost);
  return outcome;
}

bool CVPMeshPlanner::cancel()                                                 // :142-146
{
  cancel_planning_ = true;
  if (dev_ && dev_->ok()) mnav_cancel(dev_->ctx());
  return true;
}

bool CVPMeshPlanner::initialize(const std::string& plugin_name, const std::shared_ptr<mesh_map::MeshMap>& mesh_map_ptr,
                                const rclcpp::Node::SharedPtr& node)           // :148-186
{
  mesh_map_ = mesh_map_ptr;
  name_ = plugin_name;
  map_frame_ = mesh_map_->mapFrame();
  node_ = node;
  if (node_) {
    config_.publish_vector_field = node_->declare_parameter(name_ + ".publish_vector_field", config_.publish_vector_field);   // :155
    config_.publish_face_vectors = node_->declare_parameter(name_ + ".publish_face_vectors", config_.publish_face_vectors);   // :156
    config_.goal_dist_offset = node_->declare_parameter(name_ + ".goal_dist_offset", config_.goal_dist_offset);               // :157
    config_.cost_limit = node_->declare_parameter(name_ + ".cost_limit", config_.cost_limit);                                 // :165
    config_.step_width = node_->declare_parameter(name_ + ".step_width", config_.step_width);                                 // :175
  }
  direction_.assign(mesh_map_->V, 0.f);                                       // :179
  dev_.reset(new mnav_adapter::MeshMapDevice(0));
  std::string err;
  return dev_->ok() && dev_->sync(*mesh_map_, err);
}

// cvp_mesh_planner.cpp:651-970 with the loop :747-886 and computeVectorMap :897 on the device
uint32_t CVPMeshPlanner::waveFrontPropagation(const mesh_map::Vector& original_start, const mesh_map::Vector& original_goal,
                                              std::list<std::pair<mesh_map::Vector, uint32_t>>& path, std::string& message)
{
  const mesh_map::Vector start = original_start, goal = original_goal;
  const uint32_t start_face = mesh_map_->getContainingFace(start, 0.4f);      // :673
  const uint32_t goal_face = mesh_map_->getContainingFace(goal, 0.4f);        // :674
  cancel_planning_ = false;                                                   // :679
  if (start_face == mesh_map::kNoHandle) { message = "Could not find a face close enough to the given start pose"; return Result::INVALID_START; }   // :681-685
  if (goal_face == mesh_map::kNoHandle) { message = "Could not find a face close enough to the given goal pose"; return Result::INVALID_GOAL; }      // :686-690
  path.clear();
  std::string err;
  if (!dev_ || !dev_->sync(*mesh_map_, err)) { message = err; return Result::INTERNAL_ERROR; }
  const uint32_t V = mesh_map_->V;
  fields_on_host_ = false;
  vector_map_.assign((size_t)V * 3, 0.f);
  const float seed_pos[3] = { start.x, start.y, start.z };
  // the host back-tracking below reads the vector map: that one field comes back (12 MB at 1M vertices); potential,
  // predecessors, directions and cutting faces stay on the device until fetchFields()
  const uint32_t code = mnav_plan_cvp(dev_->ctx(), seed_pos, start_face, goal_face, config_.goal_dist_offset, config_.cost_limit,
                                      nullptr, nullptr, nullptr, nullptr, vector_map_.data());
  if (code == Result::CANCELED) return code;                                  // :888-892
  if (code == Result::INTERNAL_ERROR) { message = mnav_last_error(dev_->ctx()); return code; }
  // MeshMap::setVectorMap (:238): seeds hold their raw offset vector (:722-724), updated vertices the rotated direction;
  // everything else is the all-zero entry the device writes for "no value"
  std::vector<uint8_t> set(V, 0);
  const uint32_t* seeds = &mesh_map_->faces[3 * (size_t)start_face];
  for (uint32_t v = 0; v < V; ++v) set[v] = mnav_host::cvp_field_is_set(vector_map_.data(), v, seeds);
  mesh_map_->setVectorMap(vector_map_, set);
  if (config_.publish_vector_field || eager_fields_) fetchFields();
  if (code == Result::NO_PATH_FOUND) { message = "Predecessor of the goal is not set! No path found!"; return code; }   // :912-918
  // vector field back-tracking :920-966 (sequential, ~path_length / step_width iterations, host)
  return mnav_host::backtrack_on_host(start, start_face, goal, goal_face, config_.step_width, [&] { return cancel_planning_.load(); },
                                      [&](mesh_map::Vector& pos, uint32_t& face, double width) {
                                        try { return mesh_map_->meshAhead(pos, face, (float)width) ? 1 : 0; }
                                        catch (const mesh_map::MeshMap::MapPanic&) { return -1; }   // lvr2::PanicException
                                      },
                                      10u * (size_t)V + 1000u, path, message);
}

void CVPMeshPlanner::fetchFields()
{
  if (fields_on_host_ || !dev_ || !dev_->ok()) return;
  const uint32_t V = mesh_map_->V;
  potential_.assign(V, 0.f); predecessors_.assign(V, 0u); cutting_faces_.assign(V, mesh_map::kNoHandle); direction_.assign(V, 0.f);
  if (mnav_download_output(dev_->ctx(), 0, 0, potential_.data()) != 0 || mnav_download_output(dev_->ctx(), 0, 1, predecessors_.data()) != 0 ||
      mnav_download_output(dev_->ctx(), 0, 2, direction_.data()) != 0 || mnav_download_output(dev_->ctx(), 0, 3, cutting_faces_.data()) != 0) return;
  fields_on_host_ = true;
}
}  // namespace cvp_mesh_planner
