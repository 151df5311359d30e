// ros_stubs.h -- the handful of ROS 2 / Move Base Flex / geometry types the planner plugins touch,
// reduced to what compiles without ROS.  On a robot this header is replaced by the real ones
// (geometry_msgs, rclcpp, mbf_msgs); the planner sources do not change.  See INTEGRATION.md.
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace builtin_interfaces { namespace msg { struct Time { int32_t sec = 0; uint32_t nanosec = 0; }; } }
namespace std_msgs { namespace msg { struct Header { builtin_interfaces::msg::Time stamp; std::string frame_id; }; } }
namespace geometry_msgs { namespace msg {
struct Point { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Pose { Point position; Quaternion orientation; };
struct PoseStamped { std_msgs::msg::Header header; Pose pose; };
} }

// mbf_msgs::action::GetPath::Result codes used by the planners (dijkstra_mesh_planner.h:72-85)
namespace mbf_msgs { namespace action { struct GetPath { struct Result { enum : uint32_t {
  SUCCESS = 0, CANCELED = 51, INVALID_START = 52, INVALID_GOAL = 53, NO_PATH_FOUND = 54, TF_ERROR = 57, NOT_INITIALIZED = 58, INVALID_PLUGIN = 59, INTERNAL_ERROR = 60 }; }; }; } }

namespace rclcpp {
// parameter store with the declare_parameter() contract the planners rely on
class Node {
public:
  using SharedPtr = std::shared_ptr<Node>;
  template <class T> T declare_parameter(const std::string& name, const T& def)
  {
    auto it = overrides_.find(name);
    const double v = (it == overrides_.end()) ? (double)def : it->second;
    params_[name] = v;
    return (T)v;
  }
  void set_override(const std::string& name, double v) { overrides_[name] = v; }
  builtin_interfaces::msg::Time now() const { return builtin_interfaces::msg::Time(); }
private:
  std::map<std::string, double> overrides_, params_;
};
}  // namespace rclcpp
