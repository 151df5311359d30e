// gpu_mesh_planners.h -- drop-in replacements of the two reference planner plugins on top of the
// C ABI (include/mnav.h).  Class names, namespaces, parameters and return codes are the reference's:
//   dijkstra_mesh_planner::DijkstraMeshPlanner  (dijkstra_mesh_planner/include/.../dijkstra_mesh_planner.h:51-198)
//   cvp_mesh_planner::CVPMeshPlanner            (cvp_mesh_planner/include/.../cvp_mesh_planner.h:55-230)
// Everything the reference does around the wavefront loop stays on the host, line for line the
// same steps: frame transform, seed/goal lookup, path -> poses.  The loop itself is one mnav_plan_* call.
#pragma once
#include <atomic>
#include <list>
#include <utility>

#include "../../../include/mnav.h"
#include "mesh_planner.h"

namespace mnav_adapter {
// Device mirror of one MeshMap: owns the mnav context, re-uploads costs when they changed.
class MeshMapDevice {
public:
  explicit MeshMapDevice(int device = 0);
  ~MeshMapDevice();
  bool ok() const { return ctx_ != nullptr; }
  mnav_ctx* ctx() const { return ctx_; }
  // uploads the mesh on first use and the cost arrays whenever their content changed (the reference
  // re-reads vertex_costs / edge_weights by const-ref on every plan, dijkstra_mesh_planner.cpp:214)
  bool sync(const mesh_map::MeshMap& map, std::string& err);
private:
  mnav_ctx* ctx_ = nullptr;
  const mesh_map::MeshMap* uploaded_ = nullptr;
  uint64_t cost_hash_ = 0;
  bool have_costs_ = false;
};
}  // namespace mnav_adapter

namespace dijkstra_mesh_planner {
class DijkstraMeshPlanner : public mbf_mesh_core::MeshPlanner {
public:
  typedef std::shared_ptr<DijkstraMeshPlanner> Ptr;
  DijkstraMeshPlanner() {}
  virtual ~DijkstraMeshPlanner() {}
  uint32_t makePlan(const geometry_msgs::msg::PoseStamped& start, const geometry_msgs::msg::PoseStamped& goal, double tolerance,
                    std::vector<geometry_msgs::msg::PoseStamped>& plan, double& cost, std::string& message) override;
  bool cancel() override;
  bool initialize(const std::string& name, const std::shared_ptr<mesh_map::MeshMap>& mesh_map_ptr,
                  const rclcpp::Node::SharedPtr& node) override;
  // V-sized results live on the device; reading them fetches them (20 MB over PCIe at 1M vertices)
  const std::vector<float>& potential() { fetchFields(); return potential_; }
  const std::vector<uint32_t>& predecessors() { fetchFields(); return predecessors_; }
  const std::vector<float>& getVectorMap() { fetchFields(); return vector_map_; }   // dijkstra_mesh_planner.h:126
  void fetchFields();
  void setEagerFields(bool on) { eager_fields_ = on; }                // download after every plan, like the reference's host maps
protected:
  // dijkstra_mesh_planner.cpp:211-215 (the 3-arg overload): wave from `start` towards `goal`
  uint32_t dijkstra(const mesh_map::Vector& start, const mesh_map::Vector& goal, std::list<uint32_t>& path);
private:
  std::shared_ptr<mesh_map::MeshMap> mesh_map_;
  std::string name_, map_frame_;
  rclcpp::Node::SharedPtr node_;
  std::atomic_bool cancel_planning_{ false };
  struct { bool publish_vector_field = false; bool publish_face_vectors = false; double goal_dist_offset = 0.3; double cost_limit = 1.0; } config_;   // dijkstra_mesh_planner.h:178-187
  std::vector<uint32_t> predecessors_;
  std::vector<float> vector_map_, potential_;
  bool fields_on_host_ = false, eager_fields_ = false;
  std::unique_ptr<mnav_adapter::MeshMapDevice> dev_;
};
}  // namespace dijkstra_mesh_planner

namespace cvp_mesh_planner {
class CVPMeshPlanner : public mbf_mesh_core::MeshPlanner {
public:
  typedef std::shared_ptr<CVPMeshPlanner> Ptr;
  CVPMeshPlanner() {}
  virtual ~CVPMeshPlanner() {}
  uint32_t makePlan(const geometry_msgs::msg::PoseStamped& start, const geometry_msgs::msg::PoseStamped& goal, double tolerance,
                    std::vector<geometry_msgs::msg::PoseStamped>& plan, double& cost, std::string& message) override;
  bool cancel() override;
  bool initialize(const std::string& name, const std::shared_ptr<mesh_map::MeshMap>& mesh_map_ptr,
                  const rclcpp::Node::SharedPtr& node) override;
  const std::vector<float>& potential() { fetchFields(); return potential_; }
  void fetchFields();
  void setEagerFields(bool on) { eager_fields_ = on; }
protected:
  // cvp_mesh_planner.cpp:241-247 (the 4-arg overload)
  uint32_t waveFrontPropagation(const mesh_map::Vector& start, const mesh_map::Vector& goal,
                                std::list<std::pair<mesh_map::Vector, uint32_t>>& path, std::string& message);
private:
  std::shared_ptr<mesh_map::MeshMap> mesh_map_;
  std::string name_, map_frame_;
  rclcpp::Node::SharedPtr node_;
  std::atomic_bool cancel_planning_{ false };
  struct { bool publish_vector_field = false; bool publish_face_vectors = false; double goal_dist_offset = 0.3; double cost_limit = 1.0; double step_width = 0.4; } config_;   // cvp_mesh_planner.h:201-212
  std::vector<float> direction_, vector_map_, potential_;
  std::vector<uint32_t> predecessors_, cutting_faces_;
  bool fields_on_host_ = false, eager_fields_ = false;
  std::unique_ptr<mnav_adapter::MeshMapDevice> dev_;
};
}  // namespace cvp_mesh_planner
