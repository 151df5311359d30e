// mesh_gpu_planners -- the MI355X planners as REAL mbf_mesh_core::MeshPlanner plugins.
//
// This is the ROS 2 package a robot builds: it includes the reference's own headers (mbf_mesh_core, mesh_map, lvr2)
// and binds the C ABI of libmnav.so (include/mnav.h).  Nothing of mesh_navigation is modified; the two classes are
// loaded by mbf_mesh_nav through pluginlib like the reference's planners (mesh_gpu_planners.xml) and are used through
// MeshPlanner::initialize / makePlan / cancel only (mbf_mesh_core/include/mbf_mesh_core/mesh_planner.h:50-92).
//
//   mesh_gpu_planners/GpuDijkstraMeshPlanner  replaces  dijkstra_mesh_planner/DijkstraMeshPlanner
//   mesh_gpu_planners/GpuCVPMeshPlanner       replaces  cvp_mesh_planner/CVPMeshPlanner
//
// Same ROS parameters (<name>.goal_dist_offset 0.3, <name>.cost_limit 1.0, <name>.step_width 0.4, publish_*), same
// result codes and messages, same side effect (MeshMap::setVectorMap).  What runs where: the wavefront loop and
// computeVectorMap on the GPU (one mnav_plan_* call); frame transform, nearest vertex / containing face, the
// back-tracking over the vector field (the map's own MeshMap::meshAhead) and the pose assembly on the host, through the
// map's own functions.  In this repository the package is compiled against the stub dependency headers of
// oracle/ref_build (ROS 2 and lvr2 are absent) together with the reference's unmodified mesh_map sources, and run
// against the reference's planners on the same MeshMap object (tests/test_gpu_plugin_dropin.py).
#pragma once
#include <atomic>
#include <list>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include <mbf_mesh_core/mesh_planner.h>
#include <mbf_msgs/action/get_path.hpp>
#include <mesh_map/mesh_map.h>
#include <rclcpp/rclcpp.hpp>

#include "mnav.h"
#include "mesh_gpu_planners/cost_observer_layer.h"

namespace mesh_gpu_planners
{
// Device mirror of one mesh_map::MeshMap: owns the mnav context; the mesh goes up once, the cost arrays whenever
// their content changed (the reference re-reads them by const-ref on every plan, dijkstra_mesh_planner.cpp:214).
class DeviceMap
{
public:
  explicit DeviceMap(int device);
  ~DeviceMap();
  DeviceMap(const DeviceMap&) = delete;
  DeviceMap& operator=(const DeviceMap&) = delete;
  mnav_ctx* ctx() const { return ctx_; }
  bool ok() const { return ctx_ != nullptr; }
  uint32_t numVertices() const { return V_; }
  bool uploadMesh(const std::shared_ptr<mesh_map::MeshMap>& map_ptr, std::string& err);
  // true: the device copy is current.  With a CostObserverLayer in the map's layer graph: the vertices it filed since the
  // last call and their edges are updated (O(changed)); without one: one signing pass over the map's arrays per call, an
  // upload only when they changed; with setStaticCosts(true) not even that unless `force`
  bool syncCosts(mesh_map::MeshMap& map, std::string& err, bool force = false);
  void setStaticCosts(bool on) { static_costs_ = on; }
private:
  bool static_costs_ = false;
  mnav_ctx* ctx_ = nullptr;
  uint32_t V_ = 0, F_ = 0, E_ = 0;
  uint64_t cost_hash_ = 0;
  uint32_t probe_v_ = 0, probe_e_ = 0;   // rotating windows of the backstop check behind the change signal (syncCosts)
  bool have_costs_ = false;
  std::vector<float> costs_, weights_;
  std::vector<uint8_t> invalid_;
  std::shared_ptr<CostChangeLog> log_;       // the map's change log (cost_observer_layer.h) and this mirror's place in it
  int log_id_ = -1;
};

class GpuDijkstraMeshPlanner : public mbf_mesh_core::MeshPlanner
{
public:
  typedef std::shared_ptr<GpuDijkstraMeshPlanner> Ptr;
  GpuDijkstraMeshPlanner() = default;
  ~GpuDijkstraMeshPlanner() override = default;
  uint32_t makePlan(const geometry_msgs::msg::PoseStamped& start, const geometry_msgs::msg::PoseStamped& goal, double tolerance,
                    std::vector<geometry_msgs::msg::PoseStamped>& plan, double& cost, std::string& message) override;
  bool cancel() override;
  bool initialize(const std::string& name, const std::shared_ptr<mesh_map::MeshMap>& mesh_map_ptr,
                  const rclcpp::Node::SharedPtr& node) override;
  // potential of the last plan (downloaded on demand; what the reference publishes as "Potential")
  bool potential(std::vector<float>& out);
private:
  uint32_t plan(const mesh_map::Vector& wave_seed, const mesh_map::Vector& wave_target, std::list<lvr2::VertexHandle>& path);
  void exportVectorMap();
  rcl_interfaces::msg::SetParametersResult reconfigureCallback(std::vector<rclcpp::Parameter> parameters);   // dijkstra_mesh_planner.h:154
  rclcpp::Publisher<nav_msgs::msg::Path>::SharedPtr path_pub_;                                                  // :166
  rclcpp::node_interfaces::OnSetParametersCallbackHandle::SharedPtr reconfiguration_callback_handle_;          // :176
  lvr2::DenseVertexMap<mesh_map::Vector> vector_map_;                                                          // :174 (host copy of the last field)
  std::shared_ptr<mesh_map::MeshMap> mesh_map_;
  std::string name_, map_frame_;
  rclcpp::Node::SharedPtr node_;
  std::atomic_bool cancel_planning_{ false }, reload_costs_{ false };   // reload_costs_: parameter `<name>.reload_costs` was set (static_costs maps)
  struct { bool publish_vector_field = false; bool publish_face_vectors = false; double goal_dist_offset = 0.3; double cost_limit = 1.0;
           bool sync_vector_map = false; bool publish_potential = false; } config_;
  // sync_vector_map: MeshMap::setVectorMap after every plan, like the reference (:208); publish_potential: the "Potential"
  // cost layer after every plan (:124).  Both cross PCIe with V-sized arrays; switched off, a plan costs the host O(path).
  std::unique_ptr<DeviceMap> dev_;
};

class GpuCVPMeshPlanner : public mbf_mesh_core::MeshPlanner
{
public:
  typedef std::shared_ptr<GpuCVPMeshPlanner> Ptr;
  GpuCVPMeshPlanner() = default;
  ~GpuCVPMeshPlanner() override = default;
  uint32_t makePlan(const geometry_msgs::msg::PoseStamped& start, const geometry_msgs::msg::PoseStamped& goal, double tolerance,
                    std::vector<geometry_msgs::msg::PoseStamped>& plan, double& cost, std::string& message) override;
  bool cancel() override;
  bool initialize(const std::string& name, const std::shared_ptr<mesh_map::MeshMap>& mesh_map_ptr,
                  const rclcpp::Node::SharedPtr& node) override;
  bool potential(std::vector<float>& out);
private:
  uint32_t plan(const mesh_map::Vector& wave_seed, const mesh_map::Vector& wave_target,
                std::list<std::pair<mesh_map::Vector, lvr2::FaceHandle>>& path, std::string& message);
  rcl_interfaces::msg::SetParametersResult reconfigureCallback(std::vector<rclcpp::Parameter> parameters);   // cvp_mesh_planner.h:184
  rclcpp::Publisher<nav_msgs::msg::Path>::SharedPtr path_pub_;
  rclcpp::node_interfaces::OnSetParametersCallbackHandle::SharedPtr reconfiguration_callback_handle_;
  lvr2::DenseVertexMap<mesh_map::Vector> vector_map_;
  std::shared_ptr<mesh_map::MeshMap> mesh_map_;
  std::string name_, map_frame_;
  rclcpp::Node::SharedPtr node_;
  std::atomic_bool cancel_planning_{ false }, reload_costs_{ false };
  struct { bool publish_vector_field = false; bool publish_face_vectors = false; double goal_dist_offset = 0.3; double cost_limit = 1.0; double step_width = 0.4;
           bool publish_potential = false; bool sync_vector_map = false; bool device_backtracking = false; int device_inflation_layer = -1; } config_;
  // device_backtracking: the walk over the vector field (cvp :920-951) runs on the device (mnav_backtrack_cvp); with
  // sync_vector_map and publish_vector_field off as well, nothing V-sized crosses PCIe per plan.
  std::unique_ptr<DeviceMap> dev_;
};
}  // namespace mesh_gpu_planners
