// mbf_mesh_core::MeshPlanner -- the plugin interface the navigation server calls, declared exactly
// as in the reference (mbf_mesh_core/include/mbf_mesh_core/mesh_planner.h:50-92).  On a robot the
// real header is used; this copy of the three signatures only exists so the adapter builds here.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "mesh_map_host.h"
#include "ros_stubs.h"

namespace mbf_mesh_core {
class MeshPlanner {
public:
  typedef std::shared_ptr<MeshPlanner> Ptr;
  virtual ~MeshPlanner() {}
  // mesh_planner.h:71-73
  virtual uint32_t makePlan(const geometry_msgs::msg::PoseStamped& start, const geometry_msgs::msg::PoseStamped& goal,
                            double tolerance, std::vector<geometry_msgs::msg::PoseStamped>& plan, double& cost,
                            std::string& message) = 0;
  virtual bool cancel() = 0;                                                                     // :80
  virtual bool initialize(const std::string& name, const std::shared_ptr<mesh_map::MeshMap>& mesh_map_ptr,
                          const rclcpp::Node::SharedPtr& node) = 0;                              // :88
protected:
  MeshPlanner() {}                                                                               // :91
};
}  // namespace mbf_mesh_core
