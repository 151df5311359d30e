// STUB: message conversions used only for publishing (the stub publishers keep the last message for the tests).
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
// as mesh_msgs_conversions::toVertexCostsStamped: num_values entries of default_value, overwritten where the map has a value
inline mesh_msgs::msg::MeshVertexCostsStamped toVertexCostsStamped(const lvr2::VertexMap<float>& costs, std::size_t num_values, float default_value,
                                                                   const std::string& name, const std::string& frame, const std::string& uuid,
                                                                   const rclcpp::Time& stamp)
{
  mesh_msgs::msg::MeshVertexCostsStamped m; m.header.frame_id = frame; m.header.stamp = stamp; m.uuid = uuid; m.type = name;
  m.mesh_vertex_costs.costs.assign(num_values, default_value);
  for (auto vH : costs) if (vH.idx() < num_values) m.mesh_vertex_costs.costs[vH.idx()] = costs[vH];
  return m;
}
inline mesh_msgs::msg::MeshVertexCostsSparseStamped toVertexCostsSparseStamped(const lvr2::VertexMap<float>&, float, const std::string& name,
                                                                               const std::string& frame, const std::string& uuid, const rclcpp::Time& stamp)
{
  mesh_msgs::msg::MeshVertexCostsSparseStamped m; m.header.frame_id = frame; m.header.stamp = stamp; m.uuid = uuid; m.type = name; return m;
}
}  // namespace mesh_msgs_conversions
