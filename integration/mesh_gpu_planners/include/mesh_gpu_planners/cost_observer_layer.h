// mesh_gpu_planners/cost_observer_layer.h -- the change signal of the map's costs for the GPU planners.
//
// The reference planners re-read MeshMap::vertexCosts() / edgeWeights() by const reference on every plan
// (dijkstra_mesh_planner.cpp:214-215); the GPU planners keep a device copy and have to learn when it is stale.
// MeshMap::layerChanged (mesh_map.cpp:454-493) is where the map learns about a change, but a plugin cannot observe it: a
// layer's notify function is fixed by the layer manager (abstract_layer.h:53,162,220).  What the reference's plugin API does
// offer is AbstractLayer::onInputChanged (abstract_layer.h:108-120): the layer manager calls it, with the set of changed
// vertices, on every layer that lists the changed layer among its inputs (layer_manager.cpp:229-261).
//
// CostObserverLayer is such a layer and nothing else: no costs of its own, no lethals.  Put into the map's layer list
// with the map's DEFAULT layer as its input
//     mesh_map.layers: [..., gpu_cost_observer]
//     mesh_map.gpu_cost_observer.type: mesh_gpu_planners/CostObserverLayer
//     mesh_map.gpu_cost_observer.inputs: [<default layer>]
// it files the changed vertices in the map's CostChangeLog, from which every GPU planner on that map takes what came in
// since its last plan: mnav_update_costs + mnav_update_edge_weights with the changed vertices and their edges -- O(changed)
// per makePlan instead of a signing pass over all vertices and edges.  Without the layer the planners fall back to that pass.
#pragma once

#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include <mesh_map/abstract_layer.h>
#include <mesh_map/mesh_map.h>

namespace mesh_gpu_planners
{
// Changed vertices per consumer (a DeviceMap), per map.  Thread safe: layers notify from their own threads.
class CostChangeLog
{
public:
  // the map's log (created on first use).  Logs live and die with THEIR map: a map allocated later at the same address starts with a
  // fresh one, whatever a layer that was never destroyed (below) still holds
  static std::shared_ptr<CostChangeLog> of(const std::shared_ptr<mesh_map::MeshMap>& map);
  void attach() { std::lock_guard<std::mutex> l(m_); ++observers_; }
  void detach() { std::lock_guard<std::mutex> l(m_); if (observers_) --observers_; }
  bool attached() { std::lock_guard<std::mutex> l(m_); return observers_ > 0; }
  int subscribe();                                                     // a consumer id; its pending set starts empty
  void unsubscribe(int id);
  void add(const std::set<lvr2::VertexHandle>& changed);               // from the observer layer
  std::vector<uint32_t> take(int id);                                  // the consumer's pending vertices, ascending; clears them
private:
  std::mutex m_;
  int observers_ = 0, next_id_ = 0;
  std::map<int, std::set<uint32_t>> pending_;
};

class CostObserverLayer : public mesh_map::AbstractLayer
{
public:
  // AbstractLayer has no virtual destructor and pluginlib deletes layers through the base pointer: this destructor does NOT run
  // then, and nothing relies on it -- a log is dropped with its map, and the planners probe the map's arrays on every plan
  // (DeviceMap::syncCosts) instead of trusting attached() blindly
  ~CostObserverLayer();
  bool readLayer() override { return true; }                           // nothing to read, nothing to compute
  bool writeLayer() override { return true; }
  float defaultValue() override { return 0.0f; }
  float threshold() override { return std::numeric_limits<float>::infinity(); }
  bool computeLayer() override { return true; }
  const lvr2::VertexMap<float>& costs() override { return costs_; }
  const std::set<lvr2::VertexHandle>& lethals() override { return lethals_; }
  void onInputChanged(const rclcpp::Time& timestamp, const std::set<lvr2::VertexHandle>& changed) override;
protected:
  bool initialize() override;
private:
  lvr2::DenseVertexMap<float> costs_;
  std::set<lvr2::VertexHandle> lethals_;
  std::shared_ptr<CostChangeLog> log_;
};

// process-wide counters of the planners' cost synchronisation (instrumentation, read by the tests)
extern "C" void mesh_gpu_planners_cost_sync_counts(uint64_t* full_uploads, uint64_t* incremental_updates, uint64_t* signing_passes);
}  // namespace mesh_gpu_planners
