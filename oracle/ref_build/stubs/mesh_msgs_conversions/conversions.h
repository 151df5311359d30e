// STUB: message conversions used only for publishing (dropped by the stub publishers).
#pragma once
#include "../ros_stub.hpp"
#include "../lvr2_stub.hpp"
namespace mesh_msgs_conversions
{
template <typename CoordT, typename MeshPtrT, typename NormalsT>
mesh_msgs::msg::MeshGeometryStamped toMeshGeometryStamped(const MeshPtrT&, const std::string& frame, const std::string& uuid,
                                                          const NormalsT&, const rclcpp::Time& stamp)
{
  mesh_msgs::msg::MeshGeometryStamped m; m.header.frame_id = frame; m.header.stamp = stamp; m.uuid = uuid; return m;
}
inline mesh_msgs::msg::MeshVertexCostsStamped toVertexCostsStamped(const lvr2::VertexMap<float>&, std::size_t, float, const std::string& name,
                                                                   const std::string& frame, const std::string& uuid, const rclcpp::Time& stamp)
{
  mesh_msgs::msg::MeshVertexCostsStamped m; m.header.frame_id = frame; m.header.stamp = stamp; m.uuid = uuid; m.type = name; return m;
}
inline mesh_msgs::msg::MeshVertexCostsSparseStamped toVertexCostsSparseStamped(const lvr2::VertexMap<float>&, float, const std::string& name,
                                                                               const std::string& frame, const std::string& uuid, const rclcpp::Time& stamp)
{
  mesh_msgs::msg::MeshVertexCostsSparseStamped m; m.header.frame_id = frame; m.header.stamp = stamp; m.uuid = uuid; m.type = name; return m;
}
}  // namespace mesh_msgs_conversions
