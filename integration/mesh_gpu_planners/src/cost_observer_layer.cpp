// mesh_gpu_planners/cost_observer_layer.cpp -- see cost_observer_layer.h
#include "mesh_gpu_planners/cost_observer_layer.h"

#include <pluginlib/class_list_macros.hpp>

namespace mesh_gpu_planners
{
namespace
{
std::mutex g_logs_mutex;
std::map<const mesh_map::MeshMap*, std::weak_ptr<CostChangeLog>> g_logs;
}  // namespace

std::shared_ptr<CostChangeLog> CostChangeLog::of(const mesh_map::MeshMap* map)
{
  std::lock_guard<std::mutex> l(g_logs_mutex);
  for (auto it = g_logs.begin(); it != g_logs.end();) it = it->second.expired() ? g_logs.erase(it) : std::next(it);   // maps that are gone
  auto& w = g_logs[map];
  auto log = w.lock();
  if (!log) { log = std::make_shared<CostChangeLog>(); w = log; }
  return log;
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
  log_ = CostChangeLog::of(map.get());
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
