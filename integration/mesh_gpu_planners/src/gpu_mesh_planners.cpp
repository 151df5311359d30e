// mesh_gpu_planners -- see include/mesh_gpu_planners/gpu_mesh_planners.h.
// Reference line numbers (dijkstra_mesh_planner.cpp / cvp_mesh_planner.cpp / mesh_map.cpp) mark the step of the
// reference each block stands in for.
#include <mesh_gpu_planners/gpu_mesh_planners.h>

#include <mnav_planner_host.hpp>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <limits>

#include <mesh_map/util.h>
#include <pluginlib/class_list_macros.hpp>

PLUGINLIB_EXPORT_CLASS(mesh_gpu_planners::GpuDijkstraMeshPlanner, mbf_mesh_core::MeshPlanner)
PLUGINLIB_EXPORT_CLASS(mesh_gpu_planners::GpuCVPMeshPlanner, mbf_mesh_core::MeshPlanner)

namespace mesh_gpu_planners
{
using Result = mbf_msgs::action::GetPath::Result;
using geometry_msgs::msg::PoseStamped;

// ------------------------------------------------------------------------------------------------------------------
namespace
{
std::atomic<uint64_t> g_full_uploads{ 0 }, g_incremental_updates{ 0 }, g_signing_passes{ 0 };
}
extern "C" void mesh_gpu_planners_cost_sync_counts(uint64_t* full_uploads, uint64_t* incremental_updates, uint64_t* signing_passes)
{
  if (full_uploads) *full_uploads = g_full_uploads.load();
  if (incremental_updates) *incremental_updates = g_incremental_updates.load();
  if (signing_passes) *signing_passes = g_signing_passes.load();
}

DeviceMap::DeviceMap(int device) : ctx_(mnav_create(device)) {}
DeviceMap::~DeviceMap()
{
  if (log_ && log_id_ >= 0) log_->unsubscribe(log_id_);
  if (ctx_) mnav_destroy(ctx_);
}

// The half-edge mesh as flat arrays, in the reference's own ids (handle indices), once per map.
bool DeviceMap::uploadMesh(const std::shared_ptr<mesh_map::MeshMap>& map_ptr, std::string& err)
{
  if (!ctx_) { err = "no usable GPU (mnav_create failed)"; return false; }
  if (!map_ptr) { err = "no map"; return false; }
  mesh_map::MeshMap& map = *map_ptr;
  const auto mesh = map.mesh();
  if (!mesh) { err = "the map holds no mesh"; return false; }
  V_ = (uint32_t)mesh->nextVertexIndex(); F_ = (uint32_t)mesh->nextFaceIndex(); E_ = (uint32_t)mesh->nextEdgeIndex();
  if (mesh->numVertices() != V_ || mesh->numFaces() != F_ || mesh->numEdges() != E_) {
    err = "meshes with deleted elements (gaps in the handle indices) are not supported";
    return false;
  }
  std::vector<float> xyz((size_t)V_ * 3), nrm((size_t)V_ * 3, 0.f);
  std::vector<uint32_t> faces((size_t)F_ * 3), edges((size_t)E_ * 2), vf_ptr, vf;
  const auto& normals = map.vertexNormals();
  vf_ptr.reserve((size_t)V_ + 1); vf.reserve((size_t)F_ * 3);
  vf_ptr.push_back(0);
  for (uint32_t v = 0; v < V_; ++v) {
    const lvr2::VertexHandle vH(v);
    const auto p = mesh->getVertexPosition(vH);
    xyz[3 * (size_t)v] = p.x; xyz[3 * (size_t)v + 1] = p.y; xyz[3 * (size_t)v + 2] = p.z;
    const auto n = std::as_const(normals).get(vH);
    if (n) { nrm[3 * (size_t)v] = n->x; nrm[3 * (size_t)v + 1] = n->y; nrm[3 * (size_t)v + 2] = n->z; }
    for (const auto fH : mesh->getFacesOfVertex(vH)) vf.push_back(fH.idx());   // the circulator ORDER the CVP loop depends on (:775-778)
    vf_ptr.push_back((uint32_t)vf.size());
  }
  for (uint32_t f = 0; f < F_; ++f) {
    const auto vs = mesh->getVerticesOfFace(lvr2::FaceHandle(f));
    for (int k = 0; k < 3; ++k) faces[3 * (size_t)f + k] = vs[k].idx();
  }
  for (uint32_t e = 0; e < E_; ++e) {
    const auto vs = mesh->getVerticesOfEdge(lvr2::EdgeHandle(e));
    edges[2 * (size_t)e] = vs[0].idx(); edges[2 * (size_t)e + 1] = vs[1].idx();
  }
  if (mnav_set_face_circulation(ctx_, V_, F_, vf_ptr.data(), vf.data()) != 0 ||
      mnav_upload_mesh(ctx_, V_, F_, E_, xyz.data(), faces.data(), edges.data(), nrm.data()) != 0) {
    err = mnav_last_error(ctx_);
    return false;
  }
  mnav_set_resident_outputs(ctx_, 1);            // potential / predecessors / vector map stay on the device until asked for
  have_costs_ = false;
  if (log_ && log_id_ >= 0) log_->unsubscribe(log_id_);
  log_ = CostChangeLog::of(map_ptr);             // from now on every change an observer layer files is kept for this mirror
  log_id_ = log_->subscribe();
  return true;
}

// vertexCosts() / edgeWeights() / invalid, re-read like the reference does on every plan; uploaded when they changed.
// The reference re-reads the map's cost arrays by const reference on every plan (dijkstra_mesh_planner.cpp:214-215); the
// device keeps a copy, so each makePlan has to find out whether the map's layers changed it.  MeshMap offers no change
// counter, so the arrays are signed in ONE pass straight out of the map (no staging copy); only a changed signature
// stages and uploads.  `static_costs` (the planners' parameter) skips even that pass: the copy taken at initialize
// stays until the parameter `<name>.reload_costs` is set.
bool DeviceMap::syncCosts(mesh_map::MeshMap& map, std::string& err, bool force)
{
  if (static_costs_ && have_costs_ && !force) return true;
  const auto& vc = map.vertexCosts();
  const auto& ew = map.edgeWeights();
  if (have_costs_ && !force && log_ && log_->attached()) {
    // The change signal (cost_observer_layer.h): MeshMap::layerChanged (mesh_map.cpp:454-493) has updated vertex_costs and, through
    // updateEdgeWeights(changed) (:563-618), the weights of the changed vertices' edges; exactly those go to the device.
    // What the signal does NOT cover -- `mesh_map.edge_cost_factor` reconfigured (mesh_map.cpp:1379-1397 recomputes every edge
    // weight with no layer notification), `invalid` flipped by a planner that tripped over a broken vertex (dijkstra :306-321) --
    // is caught by a BACKSTOP: every plan compares a rotating window of kProbe vertices (cost, invalid) and kProbe edges
    // (weight) of the map with the mirror of what the device holds; any difference takes the full copy.  A change of every
    // edge weight is seen by the very next plan, a sparse unsignalled change within ceil(V / kProbe) plans (or at once with
    // `reload_costs`).
    constexpr uint32_t kProbe = 4096;
    auto probe_differs = [&]() -> bool {
      bool diff = false;
      const uint32_t nv = std::min(kProbe, V_), ne = std::min(kProbe, E_);
      for (uint32_t i = 0; i < nv && !diff; ++i) {
        const uint32_t v = (probe_v_ + i) % V_;
        const lvr2::VertexHandle vH(v);
        const auto c = std::as_const(vc).get(vH);
        const float f = c ? *c : 0.f;
        diff = std::memcmp(&f, &costs_[v], 4) != 0 || (map.invalid[vH] ? 1 : 0) != invalid_[v];
      }
      for (uint32_t i = 0; i < ne && !diff; ++i) {
        const uint32_t e = (probe_e_ + i) % E_;
        const auto w = std::as_const(ew).get(lvr2::EdgeHandle((size_t)e));
        const float f = w ? *w : std::numeric_limits<float>::infinity();
        diff = std::memcmp(&f, &weights_[e], 4) != 0;
      }
      if (V_) probe_v_ = (probe_v_ + nv) % V_;
      if (E_) probe_e_ = (probe_e_ + ne) % E_;
      return diff;
    };
    const std::vector<uint32_t> ids = log_->take(log_id_);
    if (ids.empty()) {
      if (costs_.size() == V_ && weights_.size() == E_ && probe_differs()) return syncCosts(map, err, true);
      return true;
    }
    const auto mesh = map.mesh();
    std::vector<float> vals(ids.size());
    std::vector<uint32_t> eids;
    std::vector<lvr2::EdgeHandle> edges;
    bool broken = false;
    for (size_t i = 0; i < ids.size() && !broken; ++i) {
      const lvr2::VertexHandle vH(ids[i]);
      const auto c = std::as_const(vc).get(vH);
      vals[i] = c ? *c : 0.f;
      edges.clear();
      try { mesh->getEdgesOfVertex(vH, edges); }
      catch (...) { broken = true; }                                   // a broken vertex (lvr2 panics, cf. dijkstra :312-321): take the full copy instead
      for (const auto eH : edges) eids.push_back((uint32_t)eH.idx());
    }
    if (broken) return syncCosts(map, err, true);
    std::sort(eids.begin(), eids.end());
    eids.erase(std::unique(eids.begin(), eids.end()), eids.end());
    std::vector<float> evals(eids.size());
    for (size_t i = 0; i < eids.size(); ++i) {
      const auto w = std::as_const(ew).get(lvr2::EdgeHandle((size_t)eids[i]));
      evals[i] = w ? *w : std::numeric_limits<float>::infinity();
    }
    if (mnav_update_costs(ctx_, (uint32_t)ids.size(), ids.data(), vals.data()) != 0 ||
        mnav_update_edge_weights(ctx_, (uint32_t)eids.size(), eids.data(), evals.data()) != 0) { err = mnav_last_error(ctx_); return false; }
    ++g_incremental_updates;
    if (costs_.size() == V_ && weights_.size() == E_) {               // the mirror follows (the backstop compares against it)
      for (size_t i = 0; i < ids.size(); ++i) costs_[ids[i]] = vals[i];
      for (size_t i = 0; i < eids.size(); ++i) weights_[eids[i]] = evals[i];
      if (probe_differs()) return syncCosts(map, err, true);
    }
    return true;
  }
  ++g_signing_passes;
  if (log_) (void)log_->take(log_id_);            // what was filed up to here is part of the full copy taken below; later changes stay pending
  uint64_t h = 0xCBF29CE484222325ull;
  auto mix = [&h](uint32_t w) { h = (h ^ w) * 0x9E3779B97F4A7C15ull; h ^= h >> 29; };
  for (uint32_t v = 0; v < V_; ++v) {
    const lvr2::VertexHandle vH(v);
    const auto c = std::as_const(vc).get(vH);
    const float f = c ? *c : 0.f;
    uint32_t w;
    std::memcpy(&w, &f, 4);
    mix(w); mix(map.invalid[vH] ? 1u : 0u);                            // the flag as its own word: no cost bit pattern can mask a flip
  }
  for (uint32_t e = 0; e < E_; ++e) {
    const auto wgt = std::as_const(ew).get(lvr2::EdgeHandle(e));
    const float f = wgt ? *wgt : std::numeric_limits<float>::infinity();
    uint32_t w;
    std::memcpy(&w, &f, 4);
    mix(w);
  }
  if (have_costs_ && h == cost_hash_) return true;
  costs_.resize(V_); weights_.resize(E_); invalid_.resize(V_);
  for (uint32_t v = 0; v < V_; ++v) {
    const lvr2::VertexHandle vH(v);
    const auto c = std::as_const(vc).get(vH);
    costs_[v] = c ? *c : 0.f;
    invalid_[v] = map.invalid[vH] ? 1 : 0;
  }
  for (uint32_t e = 0; e < E_; ++e) {
    const auto w = std::as_const(ew).get(lvr2::EdgeHandle(e));
    weights_[e] = w ? *w : std::numeric_limits<float>::infinity();
  }
  if (mnav_upload_costs(ctx_, costs_.data(), weights_.data(), invalid_.data()) != 0) { err = mnav_last_error(ctx_); return false; }
  ++g_full_uploads;
  cost_hash_ = h; have_costs_ = true;
  return true;
}

// ------------------------------------------------------------------------------------------------------------------
// Dijkstra
// ------------------------------------------------------------------------------------------------------------------
bool GpuDijkstraMeshPlanner::initialize(const std::string& plugin_name, const std::shared_ptr<mesh_map::MeshMap>& mesh_map_ptr,
                                        const rclcpp::Node::SharedPtr& node)                     // dijkstra :142-169
{
  mesh_map_ = mesh_map_ptr; name_ = plugin_name; node_ = node;
  map_frame_ = mesh_map_->mapFrame();
  config_.publish_vector_field = node_->declare_parameter(name_ + ".publish_vector_field", config_.publish_vector_field);
  // The reference leaves two V-sized host structures behind every plan: the map's vector field (setVectorMap, :208) and the
  // "Potential" vertex-cost message (:124).  Here they live on the device; bringing them to the host after EVERY plan is three
  // quarters of a default makePlan at 1M vertices (lvr2 maps of V entries: 18.8 ms against 4.6 ms).  `reference_side_effects`
  // (default false) turns both on together -- what a deployment with the reference's MeshController (which copies the map's
  // field in setPlan, mesh_controller.cpp:182) or a "Potential" display sets; each can also be set on its own.
  const bool ref_fx = node_->declare_parameter(name_ + ".reference_side_effects", false);
  config_.sync_vector_map = node_->declare_parameter(name_ + ".sync_vector_map", ref_fx);
  config_.publish_face_vectors = node_->declare_parameter(name_ + ".publish_face_vectors", config_.publish_face_vectors);
  config_.goal_dist_offset = node_->declare_parameter(name_ + ".goal_dist_offset", config_.goal_dist_offset);
  config_.cost_limit = node_->declare_parameter(name_ + ".cost_limit", config_.cost_limit);
  config_.publish_potential = node_->declare_parameter(name_ + ".publish_potential", ref_fx);
  const int device = (int)node_->declare_parameter(name_ + ".gpu_device", 0);
  const bool static_costs = node_->declare_parameter(name_ + ".static_costs", false);
  node_->declare_parameter(name_ + ".reload_costs", false);
  dev_ = std::make_unique<DeviceMap>(device);
  dev_->setStaticCosts(static_costs);
  std::string err;
  if (!dev_->uploadMesh(mesh_map_, err) || !dev_->syncCosts(*mesh_map_, err)) {
    RCLCPP_ERROR_STREAM(node_->get_logger(), name_ << ": " << err);
    return false;
  }
  path_pub_ = node_->create_publisher<nav_msgs::msg::Path>("~/path", rclcpp::QoS(1).transient_local());     // :158
  reconfiguration_callback_handle_ = node_->add_on_set_parameters_callback(                                    // :161-162
      std::bind(&GpuDijkstraMeshPlanner::reconfigureCallback, this, std::placeholders::_1));
  return true;
}

// dynamic reconfigure (:172-187): the cost limit; it is an argument of every device plan, nothing is re-uploaded
rcl_interfaces::msg::SetParametersResult GpuDijkstraMeshPlanner::reconfigureCallback(std::vector<rclcpp::Parameter> parameters)
{
  rcl_interfaces::msg::SetParametersResult result;
  for (const auto& parameter : parameters)
    if (parameter.get_name() == name_ + ".reload_costs") { if (parameter.as_bool()) reload_costs_ = true; }
    else if (parameter.get_name() == name_ + ".cost_limit") config_.cost_limit = parameter.as_double();
  result.successful = true;
  return result;
}

bool GpuDijkstraMeshPlanner::cancel()                                                              // :136-140
{
  cancel_planning_ = true;
  if (dev_ && dev_->ok()) mnav_cancel(dev_->ctx());
  return true;
}

// dijkstra(start, goal, path) :217-398: the wave starts in the vertex next to `wave_seed` and runs until the vertex
// next to `wave_target` is settled; `path` = vertices from the seed side to the predecessor of the target (:358-373)
uint32_t GpuDijkstraMeshPlanner::plan(const mesh_map::Vector& wave_seed, const mesh_map::Vector& wave_target, std::list<lvr2::VertexHandle>& path)
{
  const auto seed_opt = mesh_map_->getNearestVertexHandle(wave_seed);                             // :235
  const auto target_opt = mesh_map_->getNearestVertexHandle(wave_target);                         // :236
  cancel_planning_ = false;                                                                        // :238
  if (!seed_opt) return Result::INVALID_START;                                                     // :240
  if (!target_opt) return Result::INVALID_GOAL;                                                    // :242
  path.clear();
  std::string err;
  if (!dev_->syncCosts(*mesh_map_, err, reload_costs_.exchange(false))) { RCLCPP_ERROR_STREAM(node_->get_logger(), name_ << ": " << err); return Result::INTERNAL_ERROR; }
  std::vector<uint32_t> ids;
  const uint32_t code = mnav_host::dijkstra_vertex_path(dev_->ctx(), seed_opt.unwrap().idx(), target_opt.unwrap().idx(), config_.goal_dist_offset,
                                                       config_.cost_limit, dev_->numVertices(), ids);
  if (code != Result::SUCCESS) return code;
  for (const uint32_t id : ids) path.push_back(lvr2::VertexHandle(id));
  // computeVectorMap ends with mesh_map_->setVectorMap(vector_map_) (:208): the controller copies the map's field in
  // setPlan (mesh_controller.cpp:182), so a drop-in has to leave it there after every successful plan.  16 bytes per
  // vertex cross PCIe for it; a deployment whose controller samples the resident field instead (mnav_vector_at) turns
  // `sync_vector_map` off.
  if (config_.sync_vector_map || config_.publish_vector_field) exportVectorMap();
  return Result::SUCCESS;
}

// computeVectorMap's side effect, MeshMap::setVectorMap (:189-209)
void GpuDijkstraMeshPlanner::exportVectorMap()
{
  const uint32_t V = dev_->numVertices();
  std::vector<float> vm((size_t)V * 3);
  std::vector<uint32_t> pred(V);
  if (mnav_download_output(dev_->ctx(), 0, 4, vm.data()) != 0 || mnav_download_output(dev_->ctx(), 0, 1, pred.data()) != 0) return;
  vector_map_.clear();
  for (uint32_t v = 0; v < V; ++v)
    if (pred[v] != v) vector_map_.insert(lvr2::VertexHandle(v), mesh_map::Vector(vm[3 * (size_t)v], vm[3 * (size_t)v + 1], vm[3 * (size_t)v + 2]));   // :197
  mesh_map_->setVectorMap(vector_map_);                                                            // :208
}

bool GpuDijkstraMeshPlanner::potential(std::vector<float>& out)
{
  out.assign(dev_->numVertices(), 0.f);
  return mnav_download_output(dev_->ctx(), 0, 0, out.data()) == 0;
}

uint32_t GpuDijkstraMeshPlanner::makePlan(const PoseStamped& start, const PoseStamped& goal, double /*tolerance*/,
                                          std::vector<PoseStamped>& plan_out, double& cost, std::string& /*message*/)   // :55-134
{
  PoseStamped start_in_map, goal_in_map;
  try {
    start_in_map = mesh_map_->transformToMapFrame(start);
    goal_in_map = mesh_map_->transformToMapFrame(goal);
  } catch (tf2::TransformException& ex) {
    RCLCPP_ERROR_STREAM(node_->get_logger(), name_ << ": cannot transform start or goal into '" << map_frame_ << "': " << ex.what());
    return Result::TF_ERROR;
  }
  mesh_map::Vector robot = mesh_map::toVector(start_in_map.pose.position);
  const mesh_map::Vector target = mesh_map::toVector(goal_in_map.pose.position);
  std::list<lvr2::VertexHandle> path;
  const uint32_t outcome = plan(target, robot, path);              // the wave runs from the goal towards the robot (:84)
  path.reverse();                                                  // robot side first
  std_msgs::msg::Header header;
  header.stamp = node_->now();
  header.frame_id = mesh_map_->mapFrame();
  {                                                                // one pose per path vertex, looking at the next one (:89-116)
    const auto mesh = mesh_map_->mesh();
    const auto& normals = mesh_map_->vertexNormals();
    PoseStamped stamped;
    stamped.header = header;
    mnav_host::vertex_path_poses(path, robot, target, stamped,
                                 [&](lvr2::VertexHandle vH) -> mesh_map::Vector { return mesh->getVertexPosition(vH); },
                                 [&](lvr2::VertexHandle vH) -> mesh_map::Normal { return normals[vH]; },
                                 [](const mesh_map::Vector& from, const mesh_map::Vector& to, const mesh_map::Normal& up, float& len) {
                                   return mesh_map::calculatePoseFromPosition(from, to, up, len);
                                 },
                                 plan_out, cost);
  }
  // :119-131: the path, the potential as a vertex-cost layer, the vector field on request
  nav_msgs::msg::Path path_msg;
  path_msg.poses = plan_out;
  path_msg.header = header;
  path_pub_->publish(path_msg);
  if (config_.publish_potential) {                                 // 4 bytes per vertex cross PCIe for it; off = O(path) host work per plan
    std::vector<float> pot;
    // like the reference (:124) after EVERY plan the wave ran for, NO_PATH_FOUND included; a plan that never reached the
    // device (invalid start / goal, error, cancel) left no potential behind and publishes nothing
    if (potential(pot)) {
      lvr2::DenseVertexMap<float> potential_map;
      for (uint32_t v = 0; v < pot.size(); ++v) potential_map.insert(lvr2::VertexHandle(v), pot[v]);
      mesh_map_->publishVertexCosts(potential_map, "Potential", node_->now());
    }
  }
  if (config_.publish_vector_field && outcome == Result::SUCCESS)
    mesh_map_->publishVectorField("vector_field", vector_map_, config_.publish_face_vectors);
  return outcome;
}

// ------------------------------------------------------------------------------------------------------------------
// CVP
// ------------------------------------------------------------------------------------------------------------------
bool GpuCVPMeshPlanner::initialize(const std::string& plugin_name, const std::shared_ptr<mesh_map::MeshMap>& mesh_map_ptr,
                                   const rclcpp::Node::SharedPtr& node)                            // cvp :148-186
{
  mesh_map_ = mesh_map_ptr; name_ = plugin_name; node_ = node;
  map_frame_ = mesh_map_->mapFrame();
  config_.publish_vector_field = node_->declare_parameter(name_ + ".publish_vector_field", config_.publish_vector_field);
  config_.publish_face_vectors = node_->declare_parameter(name_ + ".publish_face_vectors", config_.publish_face_vectors);
  config_.goal_dist_offset = node_->declare_parameter(name_ + ".goal_dist_offset", config_.goal_dist_offset);
  config_.cost_limit = node_->declare_parameter(name_ + ".cost_limit", config_.cost_limit);
  config_.step_width = node_->declare_parameter(name_ + ".step_width", config_.step_width);
  const bool ref_fx = node_->declare_parameter(name_ + ".reference_side_effects", false);   // (see GpuDijkstraMeshPlanner::initialize)
  config_.publish_potential = node_->declare_parameter(name_ + ".publish_potential", ref_fx);
  config_.sync_vector_map = node_->declare_parameter(name_ + ".sync_vector_map", ref_fx);
  config_.device_backtracking = node_->declare_parameter(name_ + ".device_backtracking", config_.device_backtracking);
  config_.device_inflation_layer = (int)node_->declare_parameter(name_ + ".device_inflation_layer", config_.device_inflation_layer);
  const int device = (int)node_->declare_parameter(name_ + ".gpu_device", 0);
  const bool static_costs = node_->declare_parameter(name_ + ".static_costs", false);
  node_->declare_parameter(name_ + ".reload_costs", false);
  dev_ = std::make_unique<DeviceMap>(device);
  dev_->setStaticCosts(static_costs);
  std::string err;
  if (!dev_->uploadMesh(mesh_map_, err) || !dev_->syncCosts(*mesh_map_, err)) {
    RCLCPP_ERROR_STREAM(node_->get_logger(), name_ << ": " << err);
    return false;
  }
  if (config_.device_backtracking) mnav_set_resident_outputs(dev_->ctx(), 1);   // the field stays in HBM for mnav_backtrack_cvp
  path_pub_ = node_->create_publisher<nav_msgs::msg::Path>("~/path", rclcpp::QoS(1).transient_local());     // cvp :176
  reconfiguration_callback_handle_ = node_->add_on_set_parameters_callback(                                    // :181-182
      std::bind(&GpuCVPMeshPlanner::reconfigureCallback, this, std::placeholders::_1));
  return true;
}

// dynamic reconfigure (cvp :187-202): cost limit and back-tracking step width
rcl_interfaces::msg::SetParametersResult GpuCVPMeshPlanner::reconfigureCallback(std::vector<rclcpp::Parameter> parameters)
{
  rcl_interfaces::msg::SetParametersResult result;
  for (const auto& parameter : parameters) {
    if (parameter.get_name() == name_ + ".reload_costs") { if (parameter.as_bool()) reload_costs_ = true; }
    else if (parameter.get_name() == name_ + ".cost_limit") config_.cost_limit = parameter.as_double();
    else if (parameter.get_name() == name_ + ".step_width") config_.step_width = parameter.as_double();
  }
  result.successful = true;
  return result;
}

bool GpuCVPMeshPlanner::cancel()                                                                   // :142-146
{
  cancel_planning_ = true;
  if (dev_ && dev_->ok()) mnav_cancel(dev_->ctx());
  return true;
}

bool GpuCVPMeshPlanner::potential(std::vector<float>& out)
{
  out.assign(dev_->numVertices(), 0.f);
  return mnav_download_output(dev_->ctx(), 0, 0, out.data()) == 0;
}

// waveFrontPropagation(start, goal, path, message) :651-970: wave from the face under `wave_seed` until the face under
// `wave_target` is settled (GPU), then the walk along the vector field from the target back to the seed (host, the
// map's own meshAhead)
uint32_t GpuCVPMeshPlanner::plan(const mesh_map::Vector& wave_seed, const mesh_map::Vector& wave_target,
                                 std::list<std::pair<mesh_map::Vector, lvr2::FaceHandle>>& path, std::string& message)
{
  mesh_map::Vector seed = wave_seed, target = wave_target;        // getContainingFace projects its argument (:673-674)
  const lvr2::OptionalFaceHandle seed_opt = mesh_map_->getContainingFace(seed, 0.4);
  const lvr2::OptionalFaceHandle target_opt = mesh_map_->getContainingFace(target, 0.4);
  cancel_planning_ = false;                                                                        // :679
  if (!seed_opt) { message = "Could not find a face close enough to the given start pose"; return Result::INVALID_START; }   // :681
  if (!target_opt) { message = "Could not find a face close enough to the given goal pose"; return Result::INVALID_GOAL; }   // :686
  const lvr2::FaceHandle seed_face = seed_opt.unwrap(), target_face = target_opt.unwrap();
  path.clear();
  std::string err;
  if (!dev_->syncCosts(*mesh_map_, err, reload_costs_.exchange(false))) { message = err; return Result::INTERNAL_ERROR; }
  const uint32_t V = dev_->numVertices();
  // the V-sized field crosses PCIe only when someone on the host reads it: the map (setVectorMap, the controller's
  // directionAtPosition), the vector-field publisher, or the host back-tracking below
  const bool field_to_host = config_.sync_vector_map || config_.publish_vector_field || !config_.device_backtracking;
  std::vector<float> vm(field_to_host ? (size_t)V * 3 : 0);
  const float seed_pos[3] = { seed.x, seed.y, seed.z };
  const uint32_t code = mnav_plan_cvp(dev_->ctx(), seed_pos, seed_face.idx(), target_face.idx(), config_.goal_dist_offset, config_.cost_limit,
                                      nullptr, nullptr, nullptr, nullptr, field_to_host ? vm.data() : nullptr);   // only the vector map comes back
  if (code == Result::CANCELED) return code;
  if (code == Result::INTERNAL_ERROR) { message = mnav_last_error(dev_->ctx()); return code; }
  if (field_to_host) {
    // MeshMap::setVectorMap (:238): the field the map's meshAhead walks on.  Present for the three seed vertices (their
    // offset from the seed position, :722-724) and for every vertex the wave updated; the device writes zeros elsewhere.
    lvr2::DenseVertexMap<mesh_map::Vector>& field = vector_map_;
    field.clear();
    const auto seed_vertices = mesh_map_->mesh()->getVerticesOfFace(seed_face);
    const uint32_t seeds[3] = { seed_vertices[0].idx(), seed_vertices[1].idx(), seed_vertices[2].idx() };
    for (uint32_t v = 0; v < V; ++v)
      if (mnav_host::cvp_field_is_set(vm.data(), v, seeds)) field.insert(lvr2::VertexHandle(v), mesh_map::Vector(vm[3 * (size_t)v], vm[3 * (size_t)v + 1], vm[3 * (size_t)v + 2]));
    mesh_map_->setVectorMap(field);
  }
  if (code == Result::NO_PATH_FOUND) { message = "Predecessor of the goal is not set! No path found!"; return code; }   // :912-918
  if (config_.device_backtracking) {
    // :920-951 on the device (mnav_backtrack_cvp): the same float32 walk over the field resident in HBM, O(path) bytes
    // back.  The layers' vectorAt (mesh_map.cpp:1099-1102) is the device's own inflation layer (device_inflation_layer,
    // computed by mnav_layer_inflation) or none: a host-side layer plugin's private field is not reachable from here.
    const uint32_t cap = 1u << 16;
    std::vector<float> pp((size_t)cap * 3);
    std::vector<uint32_t> pf(cap);
    uint32_t n = 0;
    const float target_pos[3] = { target.x, target.y, target.z };
    const int st = mnav_backtrack_cvp(dev_->ctx(), seed_pos, seed_face.idx(), target_pos, target_face.idx(), config_.step_width,
                                      config_.device_inflation_layer, cap, pp.data(), pf.data(), &n);
    const char* derr = mnav_last_error(dev_->ctx());
    if (st < 0 && derr && *derr) { message = derr; return Result::INTERNAL_ERROR; }
    for (uint32_t i = n; i-- > 0;)                                    // rows come seed first; push_front from the robot's end
      path.push_front(std::make_pair(mesh_map::Vector(pp[3 * (size_t)i], pp[3 * (size_t)i + 1], pp[3 * (size_t)i + 2]), lvr2::FaceHandle(pf[i])));
    if (st == -1) { message = "Could not find a valid path, while back-tracking from the goal: HalfEdgeMesh panicked!"; return Result::NO_PATH_FOUND; }
    if (st != 1) { message = "Could not find a valid path, while back-tracking from the goal"; return Result::NO_PATH_FOUND; }
    if (cancel_planning_) return Result::CANCELED;
    return Result::SUCCESS;
  }
  // :920-966 on the host: the map's own meshAhead over the field just handed to it
  return mnav_host::backtrack_on_host(seed, seed_face, target, target_face, config_.step_width, [&] { return cancel_planning_.load(); },
                                      [&](mesh_map::Vector& pos, lvr2::FaceHandle& face, double width) {
                                        try { return mesh_map_->meshAhead(pos, face, width) ? 1 : 0; }
                                        catch (lvr2::PanicException&) { return -1; }
                                      },
                                      0, path, message);
}

uint32_t GpuCVPMeshPlanner::makePlan(const PoseStamped& start, const PoseStamped& goal, double /*tolerance*/,
                                     std::vector<PoseStamped>& plan_out, double& cost, std::string& message)   // :62-140
{
  PoseStamped start_in_map, goal_in_map;
  try {
    start_in_map = mesh_map_->transformToMapFrame(start);
    goal_in_map = mesh_map_->transformToMapFrame(goal);
  } catch (tf2::TransformException& ex) {
    RCLCPP_ERROR_STREAM(node_->get_logger(), name_ << ": cannot transform start or goal into '" << map_frame_ << "': " << ex.what());
    return Result::TF_ERROR;
  }
  const mesh_map::Vector robot = mesh_map::toVector(start_in_map.pose.position);
  const mesh_map::Vector target = mesh_map::toVector(goal_in_map.pose.position);
  std::list<std::pair<mesh_map::Vector, lvr2::FaceHandle>> path;
  const uint32_t outcome = plan(target, robot, path, message);     // wave from the goal (:89)
  path.reverse();
  std_msgs::msg::Header header;
  header.stamp = node_->now();
  header.frame_id = mesh_map_->mapFrame();
  {                                                                // :99-124
    const auto& face_normals = mesh_map_->faceNormals();
    PoseStamped stamped;
    stamped.header = header;
    mnav_host::face_path_poses(path, cancel_planning_, goal_in_map.pose, stamped,
                               [&](lvr2::FaceHandle fH) -> mesh_map::Normal { return face_normals[fH]; },
                               [](const mesh_map::Vector& from, const mesh_map::Vector& to, const mesh_map::Normal& up, float& len) {
                                 return mesh_map::calculatePoseFromPosition(from, to, up, len);
                               },
                               plan_out, cost);
  }
  // :125-137: the path, the potential as a vertex-cost layer, the vector field on request
  nav_msgs::msg::Path path_msg;
  path_msg.poses = plan_out;
  path_msg.header = header;
  path_pub_->publish(path_msg);
  if (config_.publish_potential) {
    std::vector<float> pot;
    if ((outcome == Result::SUCCESS || outcome == Result::NO_PATH_FOUND) && potential(pot)) {
      lvr2::DenseVertexMap<float> potential_map;
      for (uint32_t v = 0; v < pot.size(); ++v) potential_map.insert(lvr2::VertexHandle(v), pot[v]);
      mesh_map_->publishVertexCosts(potential_map, "Potential", header.stamp);
    }
  }
  if (config_.publish_vector_field) mesh_map_->publishVectorField("vector_field", vector_map_, config_.publish_face_vectors);
  return outcome;
}
}  // namespace mesh_gpu_planners
