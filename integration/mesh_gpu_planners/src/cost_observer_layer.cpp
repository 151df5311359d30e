// mesh_gpu_planners/cost_observer_layer.cpp -- see cost_observer_layer.h
#include "mesh_gpu_planners/cost_observer_layer.h"

#include <algorithm>

#include <pluginlib/class_list_macros.hpp>

namespace mesh_gpu_planners
{
namespace
{
std::mutex g_logs_mutex;
struct MapLog { std::weak_ptr<mesh_map::MeshMap> map; std::shared_ptr<CostChangeLog> log; };
std::map<const mesh_map::MeshMap*, MapLog> g_logs;
}  // namespace

std::shared_ptr<CostChangeLog> CostChangeLog::of(const std::shared_ptr<mesh_map::MeshMap>& map)
{
  std::lock_guard<std::mutex> l(g_logs_mutex);
  for (auto it = g_logs.begin(); it != g_logs.end();) it = it->second.map.expired() ? g_logs.erase(it) : std::next(it);   // maps that are gone
  MapLog& e = g_logs[map.get()];
  if (!e.log) { e.map = map; e.log = std::make_shared<CostChangeLog>(); }
  return e.log;
}

int CostChangeLog::subscribe()
{
  std::lock_guard<std::mutex> l(m_);
  pending_[next_id_];
  return next_id_++;
}

void CostChangeLog::unsubscribe(int id)
{
  std::lock_guard<std::mutex> l(m_);
  pending_.erase(id);
}

void CostChangeLog::add(const std::set<lvr2::VertexHandle>& changed)
{
  std::lock_guard<std::mutex> l(m_);
  for (auto& kv : pending_)
    for (const auto vH : changed) kv.second.insert((uint32_t)vH.idx());
}

std::vector<uint32_t> CostChangeLog::take(int id)
{
  std::lock_guard<std::mutex> l(m_);
  std::vector<uint32_t> out;
  auto it = pending_.find(id);
  if (it != pending_.end()) { out.assign(it->second.begin(), it->second.end()); it->second.clear(); }
  return out;
}

bool CostObserverLayer::initialize()
{
  const auto map = map_ptr_.lock();
  if (!map) return false;
  // The layer only hears about changes of its INPUTS (layer_manager.cpp:229-261): without the map's default layer among them it
  // would report "attached" and never file a change.  Refuse that configuration loudly; the planners then keep signing the arrays.
  if (node_) {
    std::string default_layer;
    std::vector<std::string> inputs;
    if (node_->get_parameter(mesh_map::MeshMap::MESH_MAP_NAMESPACE + ".default_layer", default_layer) &&
        node_->get_parameter(mesh_map::MeshMap::MESH_MAP_NAMESPACE + "." + layer_name_ + ".inputs", inputs) &&
        std::find(inputs.begin(), inputs.end(), default_layer) == inputs.end()) {
      RCLCPP_WARN_STREAM(node_->get_logger(), "CostObserverLayer '" << layer_name_ << "': the map's default layer '" << default_layer
                                               << "' is not among its inputs -- it would never see a cost change; not attached");
      return true;
    }
  }
  log_ = CostChangeLog::of(map);
  log_->attach();
  return true;
}

CostObserverLayer::~CostObserverLayer()
{
  if (log_) log_->detach();
}

void CostObserverLayer::onInputChanged(const rclcpp::Time&, const std::set<lvr2::VertexHandle>& changed)
{
  // (the layer manager calls the dependents after MeshMap::layerChanged updated vertex_costs and the edge weights,
  //  layer_manager.cpp:229-261: what the planners read on their next plan is current)
  if (log_) log_->add(changed);
}
}  // namespace mesh_gpu_planners

PLUGINLIB_EXPORT_CLASS(mesh_gpu_planners::CostObserverLayer, mesh_map::AbstractLayer)
