// C-ABI harness around the REFERENCE's own translation units (compiled unmodified from /root/reference by
// oracle/ref_build/build.sh against the stub headers in oracle/ref_build/stubs).  It drives
//   mesh_map::MeshMap::readMap()  (layers, vertex costs, computeEdgeWeights, kd-tree)
//   dijkstra_mesh_planner::DijkstraMeshPlanner::{dijkstra, makePlan}
//   cvp_mesh_planner::CVPMeshPlanner::{waveFrontPropagation, makePlan}
//   mesh_layers::{SteepnessLayer, InflationLayer, Max/AvgCombinationLayer} through the real LayerManager
// and copies their results into plain arrays.  This file contains no planner arithmetic of its own; it is
// compiled with -fno-access-control so it can read the planners' private result maps.
// Test infrastructure only (the checker that pins oracle/mnav_oracle.c); never linked into the product.
#include <cstdint>
#include <cstring>
#include <list>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include <mbf_mesh_core/mesh_planner.h>
#include <mesh_map/mesh_map.h>
#include <mesh_map/util.h>
#include <dijkstra_mesh_planner/dijkstra_mesh_planner.h>
#include <cvp_mesh_planner/cvp_mesh_planner.h>
#include <mesh_layers/inflation_layer.h>
#include <pluginlib/class_list_macros.hpp>

namespace ref_harness
{
struct ArrayLayerData { std::vector<float> costs; std::vector<uint8_t> lethal; };
static std::map<std::string, ArrayLayerData>& array_layer_table()
{
  static std::map<std::string, ArrayLayerData> t;
  return t;
}

// A layer plugin that serves per-vertex costs handed over by the harness (the reference's plugin API for
// custom layers, mesh_map/abstract_layer.h).  Lets the tests put arbitrary cost vectors under the planners.
class ArrayLayer : public mesh_map::AbstractLayer
{
public:
  bool readLayer() override { return false; }
  bool writeLayer() override { return true; }
  float defaultValue() override { return 0.0f; }
  float threshold() override { return 1.0f; }
  bool computeLayer() override
  {
    const auto& d = array_layer_table()[layer_name_];
    costs_.clear();
    lethals_.clear();
    for (std::size_t i = 0; i < d.costs.size(); ++i) costs_.insert(lvr2::VertexHandle(i), d.costs[i]);
    for (std::size_t i = 0; i < d.lethal.size(); ++i) if (d.lethal[i]) lethals_.insert(lvr2::VertexHandle(i));
    return true;
  }
  const lvr2::VertexMap<float>& costs() override { return costs_; }
  const std::set<lvr2::VertexHandle>& lethals() override { return lethals_; }
  bool initialize() override { return true; }
  void update(const std::vector<uint32_t>& ids, const std::vector<float>& values, const std::vector<uint8_t>& lethal)
  {
    std::set<lvr2::VertexHandle> changed;
    {
      const auto lock = writeLock();
      for (std::size_t i = 0; i < ids.size(); ++i) {
        const lvr2::VertexHandle v(ids[i]);
        costs_.insert(v, values[i]);
        if (!lethal.empty()) { if (lethal[i]) lethals_.insert(v); else lethals_.erase(v); }
        changed.insert(v);
      }
    }
    notifyChange(node_->get_clock()->now(), changed);
  }
private:
  lvr2::DenseVertexMap<float> costs_;
  std::set<lvr2::VertexHandle> lethals_;
};
}  // namespace ref_harness
PLUGINLIB_EXPORT_CLASS(ref_harness::ArrayLayer, mesh_map::AbstractLayer)

namespace
{
struct Ref
{
  std::string file;
  rclcpp::Node::SharedPtr node;
  tf2_ros::Buffer tf;
  std::shared_ptr<mesh_map::MeshMap> map;
  std::shared_ptr<dijkstra_mesh_planner::DijkstraMeshPlanner> dij;
  std::shared_ptr<cvp_mesh_planner::CVPMeshPlanner> cvp;
  std::shared_ptr<mbf_mesh_core::MeshPlanner> plugin;     // any planner loaded by lookup name (ref_plugin_*)
  std::string message;
};
int g_counter = 0;

geometry_msgs::msg::PoseStamped pose_from(const double* p7, const std::string& frame)
{
  geometry_msgs::msg::PoseStamped p;
  p.header.frame_id = frame;
  p.pose.position.x = p7[0]; p.pose.position.y = p7[1]; p.pose.position.z = p7[2];
  p.pose.orientation.x = p7[3]; p.pose.orientation.y = p7[4]; p.pose.orientation.z = p7[5]; p.pose.orientation.w = p7[6];
  return p;
}
void pose_to(const geometry_msgs::msg::Pose& p, double* o)
{
  o[0] = p.position.x; o[1] = p.position.y; o[2] = p.position.z;
  o[3] = p.orientation.x; o[4] = p.orientation.y; o[5] = p.orientation.z; o[6] = p.orientation.w;
}
}  // namespace

extern "C" {

void* ref_new()
{
  auto* r = new Ref();
  r->file = "ref_mem_" + std::to_string(++g_counter) + ".h5";
  r->node = std::make_shared<rclcpp::Node>("ref");
  r->node->stub_set_override("mesh_map.mesh_file", rclcpp::ParameterValue(r->file));
  r->node->stub_set_override("mesh_map.mesh_part", rclcpp::ParameterValue(std::string("mesh")));
  return r;
}
void ref_free(void* h)
{
  auto* r = static_cast<Ref*>(h);
  lvr2::ref_store_registry().erase(r->file);
  delete r;
}

// ---- parameters (before ref_read_map / planner init: overrides; afterwards: `ros2 param set`) ----
void ref_param_double(void* h, const char* name, double v) { static_cast<Ref*>(h)->node->stub_set_override(name, rclcpp::ParameterValue(v)); }
void ref_param_bool(void* h, const char* name, int v) { static_cast<Ref*>(h)->node->stub_set_override(name, rclcpp::ParameterValue(v != 0)); }
void ref_param_int(void* h, const char* name, int64_t v) { static_cast<Ref*>(h)->node->stub_set_override(name, rclcpp::ParameterValue(v)); }
void ref_param_string(void* h, const char* name, const char* v) { static_cast<Ref*>(h)->node->stub_set_override(name, rclcpp::ParameterValue(std::string(v))); }
// comma separated list
void ref_param_string_array(void* h, const char* name, const char* csv)
{
  std::vector<std::string> out;
  std::string s(csv), tok;
  std::size_t pos = 0;
  while (!s.empty()) {
    pos = s.find(',');
    tok = s.substr(0, pos);
    if (!tok.empty()) out.push_back(tok);
    if (pos == std::string::npos) break;
    s.erase(0, pos + 1);
  }
  static_cast<Ref*>(h)->node->stub_set_override(name, rclcpp::ParameterValue(out));
}
int ref_set_param_bool(void* h, const char* name, int v) { return static_cast<Ref*>(h)->node->stub_set_parameter(name, rclcpp::ParameterValue(v != 0)) ? 1 : 0; }
int ref_set_param_double(void* h, const char* name, double v) { return static_cast<Ref*>(h)->node->stub_set_parameter(name, rclcpp::ParameterValue(v)) ? 1 : 0; }

// ---- the "map file" ----
void ref_set_mesh(void* h, uint32_t V, uint32_t F, const float* xyz, const uint32_t* faces)
{
  auto* r = static_cast<Ref*>(h);
  auto buf = std::make_shared<lvr2::MeshBuffer>();
  buf->setVertices(xyz, V);
  buf->setFaceIndices(faces, F);
  lvr2::ref_store_open(r->file)->meshes["mesh"] = buf;
}
// optional attributes a real map file may carry (readMap uses them instead of recomputing, mesh_map.cpp:351-432)
void ref_set_attr_face_normals(void* h, uint32_t F, const float* fn)
{
  lvr2::DenseFaceMap<mesh_map::Normal> m;
  for (uint32_t f = 0; f < F; ++f) m.insert(lvr2::FaceHandle(std::size_t(f)), mesh_map::Normal::raw(fn[3 * f], fn[3 * f + 1], fn[3 * f + 2]));
  lvr2::ref_store_open(static_cast<Ref*>(h)->file)->attributes["mesh/face_normals"] = m;
}
void ref_set_attr_vertex_normals(void* h, uint32_t V, const float* vn)
{
  lvr2::DenseVertexMap<mesh_map::Normal> m;
  for (uint32_t v = 0; v < V; ++v) m.insert(lvr2::VertexHandle(v), mesh_map::Normal::raw(vn[3 * v], vn[3 * v + 1], vn[3 * v + 2]));
  lvr2::ref_store_open(static_cast<Ref*>(h)->file)->attributes["mesh/vertex_normals"] = m;
}
void ref_set_attr_edge_distances(void* h, uint32_t E, const float* ed)
{
  lvr2::DenseEdgeMap<float> m;
  for (uint32_t e = 0; e < E; ++e) m.insert(lvr2::EdgeHandle(std::size_t(e)), ed[e]);
  lvr2::ref_store_open(static_cast<Ref*>(h)->file)->attributes["mesh/edge_distances"] = m;
}
void ref_set_array_layer(void* h, const char* layer_name, uint32_t V, const float* costs, const uint8_t* lethal)
{
  (void)h;
  auto& d = ref_harness::array_layer_table()[layer_name];
  d.costs.assign(costs, costs + V);
  if (lethal) d.lethal.assign(lethal, lethal + V); else d.lethal.clear();
}

int ref_read_map(void* h)
{
  auto* r = static_cast<Ref*>(h);
  try {
    r->map = std::make_shared<mesh_map::MeshMap>(r->tf, r->node);
    return r->map->readMap() ? 1 : 0;
  } catch (const std::exception& e) {
    r->message = e.what();
    std::fprintf(stderr, "ref_read_map: %s\n", e.what());
    return 0;
  }
}
const char* ref_message(void* h) { return static_cast<Ref*>(h)->message.c_str(); }
long ref_logged_errors() { return rclcpp::LogState::errors().load(); }

// ---- map accessors ----
uint32_t ref_num_vertices(void* h) { return static_cast<Ref*>(h)->map->mesh()->numVertices(); }
uint32_t ref_num_faces(void* h) { return static_cast<Ref*>(h)->map->mesh()->numFaces(); }
uint32_t ref_num_edges(void* h) { return static_cast<Ref*>(h)->map->mesh()->numEdges(); }
void ref_edges(void* h, uint32_t* ev)
{
  const auto mesh = static_cast<Ref*>(h)->map->mesh();
  for (auto e : mesh->edges()) { const auto v = mesh->getVerticesOfEdge(e); ev[2 * e.idx()] = v[0].idx(); ev[2 * e.idx() + 1] = v[1].idx(); }
}
void ref_face_vertices(void* h, uint32_t* fv)
{
  const auto mesh = static_cast<Ref*>(h)->map->mesh();
  for (auto f : mesh->faces()) { const auto v = mesh->getVerticesOfFace(f); for (int k = 0; k < 3; ++k) fv[3 * f.idx() + k] = v[k].idx(); }
}
// incident edges / faces of one vertex in the circulator order the planners iterate them
uint32_t ref_edges_of_vertex(void* h, uint32_t v, uint32_t* out, uint32_t cap)
{
  std::vector<lvr2::EdgeHandle> es;
  static_cast<Ref*>(h)->map->mesh()->getEdgesOfVertex(lvr2::VertexHandle(v), es);
  for (uint32_t i = 0; i < es.size() && i < cap; ++i) out[i] = es[i].idx();
  return es.size();
}
uint32_t ref_faces_of_vertex(void* h, uint32_t v, uint32_t* out, uint32_t cap)
{
  std::vector<lvr2::FaceHandle> fs;
  static_cast<Ref*>(h)->map->mesh()->getFacesOfVertex(lvr2::VertexHandle(v), fs);
  for (uint32_t i = 0; i < fs.size() && i < cap; ++i) out[i] = fs[i].idx();
  return fs.size();
}
void ref_vertex_costs(void* h, float* out) { const auto& m = static_cast<Ref*>(h)->map->vertexCosts(); for (auto v : m) out[v.idx()] = m[v]; }
void ref_edge_weights(void* h, float* out) { const auto& m = static_cast<Ref*>(h)->map->edgeWeights(); for (auto e : m) out[e.idx()] = m[e]; }
void ref_edge_distances(void* h, float* out) { const auto& m = static_cast<Ref*>(h)->map->edgeDistances(); for (auto e : m) out[e.idx()] = m[e]; }
void ref_face_normals(void* h, float* out)
{
  const auto& m = static_cast<Ref*>(h)->map->faceNormals();
  for (auto f : m) { const auto& n = m[f]; out[3 * f.idx()] = n.x; out[3 * f.idx() + 1] = n.y; out[3 * f.idx() + 2] = n.z; }
}
void ref_vertex_normals(void* h, float* out)
{
  const auto& m = static_cast<Ref*>(h)->map->vertexNormals();
  for (auto v : m) { const auto& n = m[v]; out[3 * v.idx()] = n.x; out[3 * v.idx() + 1] = n.y; out[3 * v.idx() + 2] = n.z; }
}
void ref_set_invalid(void* h, uint32_t V, const uint8_t* invalid)
{
  auto& inv = static_cast<Ref*>(h)->map->invalid;
  for (uint32_t v = 0; v < V; ++v) inv.insert(lvr2::VertexHandle(v), invalid[v] != 0);
}
void ref_get_invalid(void* h, uint32_t V, uint8_t* invalid)
{
  auto& inv = static_cast<Ref*>(h)->map->invalid;
  for (uint32_t v = 0; v < V; ++v) invalid[v] = inv[lvr2::VertexHandle(v)] ? 1 : 0;
}
// costs of one layer; vertices without a value get the layer's defaultValue() (what consumers do)
int ref_layer_costs(void* h, const char* name, float* out, uint8_t* lethal)
{
  auto* r = static_cast<Ref*>(h);
  const auto layer = r->map->layer(name);
  if (!layer) return 0;
  const uint32_t V = r->map->mesh()->numVertices();
  const auto& cm = layer->costs();
  const float def = layer->defaultValue();
  for (uint32_t v = 0; v < V; ++v) out[v] = cm.get(lvr2::VertexHandle(v)).value_or(def);
  if (lethal) { std::memset(lethal, 0, V); for (auto v : layer->lethals()) lethal[v.idx()] = 1; }
  return 1;
}
// Inflation internals: distance field and repulsive vector field (mesh_layers/inflation_layer.h members)
int ref_layer_vector_at(void* h, const char* name, const uint32_t vs[3], const float bary[3], float out[3])
{
  const auto layer = static_cast<Ref*>(h)->map->layer(name);
  if (!layer) return 0;
  try {
    const auto v = layer->vectorAt({ lvr2::VertexHandle(vs[0]), lvr2::VertexHandle(vs[1]), lvr2::VertexHandle(vs[2]) }, { bary[0], bary[1], bary[2] });
    out[0] = v.x; out[1] = v.y; out[2] = v.z;
  } catch (const lvr2::PanicException&) {
    return -1;   // an attribute map without a value for one of the vertices (what meshAhead's caller catches, cvp :944)
  }
  return 1;
}
// InflationLayer members: distance field (inf where the wave never came) and the repulsive vector field
int ref_inflation_fields(void* h, const char* name, float* dist, float* vec)
{
  auto* r = static_cast<Ref*>(h);
  const auto layer = std::dynamic_pointer_cast<mesh_layers::InflationLayer>(r->map->layer(name));
  if (!layer) return 0;
  const uint32_t V = r->map->mesh()->numVertices();
  for (uint32_t v = 0; v < V; ++v) {
    const lvr2::VertexHandle vh(v);
    const auto d = std::as_const(layer->distances_).get(vh);
    dist[v] = d ? *d : std::numeric_limits<float>::infinity();
    const auto m = std::as_const(layer->vector_map_).get(vh);
    vec[3 * v] = m ? m->x : 0; vec[3 * v + 1] = m ? m->y : 0; vec[3 * v + 2] = m ? m->z : 0;
  }
  return 1;
}
// cost change at run time through the reference's notification chain: layer -> LayerManager::layer_changed
// -> MeshMap::layerChanged -> updateEdgeWeights (incremental) and dependents' onInputChanged
int ref_update_array_layer(void* h, const char* name, uint32_t n, const uint32_t* ids, const float* values, const uint8_t* lethal)
{
  const auto layer = std::dynamic_pointer_cast<ref_harness::ArrayLayer>(static_cast<Ref*>(h)->map->layer(name));
  if (!layer) return 0;
  layer->update(std::vector<uint32_t>(ids, ids + n), std::vector<float>(values, values + n),
                lethal ? std::vector<uint8_t>(lethal, lethal + n) : std::vector<uint8_t>());
  return 1;
}

// ---- seed / goal resolution (mesh_map.cpp:1110-1174) ----
uint32_t ref_nearest_vertex(void* h, const float p[3])
{
  const auto o = static_cast<Ref*>(h)->map->getNearestVertexHandle(mesh_map::Vector(p[0], p[1], p[2]));
  return o ? o.unwrap().idx() : 0xFFFFFFFFu;
}
uint32_t ref_containing_face(void* h, const float p[3], float max_dist, float bary[3])
{
  mesh_map::Vector q(p[0], p[1], p[2]);
  const auto res = static_cast<Ref*>(h)->map->searchContainingFace(q, max_dist);
  if (!res) return 0xFFFFFFFFu;
  if (bary) for (int k = 0; k < 3; ++k) bary[k] = std::get<2>(*res)[k];
  return std::get<0>(*res).idx();
}
int ref_mesh_ahead(void* h, float pos[3], uint32_t* face, float step)
{
  mesh_map::Vector p(pos[0], pos[1], pos[2]);
  lvr2::FaceHandle f{ std::size_t(*face) };
  const bool ok = static_cast<Ref*>(h)->map->meshAhead(p, f, step);
  pos[0] = p.x; pos[1] = p.y; pos[2] = p.z; *face = f.idx();
  return ok ? 1 : 0;
}

// ---- Dijkstra planner ----
int ref_dijkstra_init(void* h, const char* name)
{
  auto* r = static_cast<Ref*>(h);
  r->dij = std::make_shared<dijkstra_mesh_planner::DijkstraMeshPlanner>();
  return r->dij->initialize(name, r->map, r->node) ? 1 : 0;
}
// DijkstraMeshPlanner::dijkstra(start = wave seed, goal = wave target, path) (dijkstra_mesh_planner.cpp:211-215)
uint32_t ref_dijkstra(void* h, const float seed[3], const float target[3], uint32_t* path, uint32_t cap, uint32_t* path_len)
{
  auto* r = static_cast<Ref*>(h);
  std::list<lvr2::VertexHandle> p;
  const uint32_t code = r->dij->dijkstra(mesh_map::Vector(seed[0], seed[1], seed[2]), mesh_map::Vector(target[0], target[1], target[2]), p);
  uint32_t n = 0;
  for (auto v : p) { if (n < cap) path[n] = v.idx(); ++n; }
  *path_len = n;
  return code;
}
// results of the last dijkstra()/makePlan: potential_, predecessors_, vector_map_ (0 / has flag)
void ref_dijkstra_fields(void* h, float* dist, uint32_t* pred, float* vecmap, uint8_t* has_vec)
{
  auto* r = static_cast<Ref*>(h);
  const uint32_t V = r->map->mesh()->numVertices();
  for (uint32_t v = 0; v < V; ++v) {
    const lvr2::VertexHandle vh(v);
    if (dist) { const auto d = std::as_const(r->dij->potential_).get(vh); dist[v] = d ? *d : std::numeric_limits<float>::quiet_NaN(); }
    if (pred) { const auto p = std::as_const(r->dij->predecessors_).get(vh); pred[v] = p ? p->idx() : 0xFFFFFFFFu; }
    if (vecmap) {
      const auto m = std::as_const(r->dij->vector_map_).get(vh);
      if (has_vec) has_vec[v] = m ? 1 : 0;
      vecmap[3 * v] = m ? m->x : 0; vecmap[3 * v + 1] = m ? m->y : 0; vecmap[3 * v + 2] = m ? m->z : 0;
    }
  }
}
// the field the MAP holds (MeshMap::getVectorMap, mesh_map.h:268): what a planner's setVectorMap left for the controller
void ref_map_vector_map(void* h, float* vecmap, uint8_t* has_vec)
{
  auto* r = static_cast<Ref*>(h);
  const uint32_t V = r->map->mesh()->numVertices();
  const auto& vm = r->map->getVectorMap();
  for (uint32_t v = 0; v < V; ++v) {
    const auto m = vm.get(lvr2::VertexHandle(v));
    has_vec[v] = m ? 1 : 0;
    vecmap[3 * v] = m ? m->x : 0; vecmap[3 * v + 1] = m ? m->y : 0; vecmap[3 * v + 2] = m ? m->z : 0;
  }
}
uint32_t ref_dijkstra_make_plan(void* h, const double start7[7], const double goal7[7], double* poses, uint32_t cap, uint32_t* n_poses, double* cost)
{
  auto* r = static_cast<Ref*>(h);
  std::vector<geometry_msgs::msg::PoseStamped> plan;
  r->message.clear();
  const uint32_t code = r->dij->makePlan(pose_from(start7, r->map->mapFrame()), pose_from(goal7, r->map->mapFrame()), 0.0, plan, *cost, r->message);
  *n_poses = plan.size();
  for (uint32_t i = 0; i < plan.size() && i < cap; ++i) pose_to(plan[i].pose, poses + 7 * i);
  return code;
}
void ref_dijkstra_cancel(void* h) { static_cast<Ref*>(h)->dij->cancel(); }

// ---- any MeshPlanner plugin, loaded the way mbf_mesh_nav loads planners (mesh_navigation_server.cpp:74-124): by
// lookup name through pluginlib, then used through the MeshPlanner interface only.  With the GPU build of this
// library (build.sh: libmnav_ref_gpu.so) "mesh_gpu_planners/GpuDijkstraMeshPlanner" and ".../GpuCVPMeshPlanner"
// are registered next to the reference's own planners, on the same MeshMap object.
int ref_plugin_init(void* h, const char* lookup_name, const char* name)
{
  auto* r = static_cast<Ref*>(h);
  try {
    pluginlib::ClassLoader<mbf_mesh_core::MeshPlanner> loader("mbf_mesh_core", "mbf_mesh_core::MeshPlanner");
    r->plugin = loader.createSharedInstance(lookup_name);
  } catch (const std::exception& e) { r->message = e.what(); r->plugin.reset(); return 0; }
  return r->plugin->initialize(name, r->map, r->node) ? 1 : 0;
}
uint32_t ref_plugin_make_plan(void* h, const double start7[7], const double goal7[7], double* poses, uint32_t cap, uint32_t* n_poses, double* cost)
{
  auto* r = static_cast<Ref*>(h);
  std::vector<geometry_msgs::msg::PoseStamped> plan;
  r->message.clear();
  const uint32_t code = r->plugin->makePlan(pose_from(start7, r->map->mapFrame()), pose_from(goal7, r->map->mapFrame()), 0.0, plan, *cost, r->message);
  *n_poses = plan.size();
  for (uint32_t i = 0; i < plan.size() && i < cap; ++i) pose_to(plan[i].pose, poses + 7 * i);
  return code;
}
// ---- what was published (the stub publishers keep the last message per topic) ----------------------------------
long ref_pub_count(const char* topic)
{
  const auto* p = rclcpp::stub_find_publisher(topic);
  return p ? (long)p->count_ : -1;
}
// last nav_msgs::Path on `topic`: poses as 7 doubles each
int ref_pub_last_path(const char* topic, double* poses, uint32_t cap, uint32_t* n)
{
  auto* p = dynamic_cast<rclcpp::Publisher<nav_msgs::msg::Path>*>(rclcpp::stub_find_publisher(topic));
  if (!p || p->count_ == 0) return 0;
  const auto& msg = p->last();
  *n = (uint32_t)msg.poses.size();
  for (uint32_t i = 0; i < msg.poses.size() && i < cap; ++i) {
    const auto& q = msg.poses[i].pose;
    double* o = poses + 7 * (size_t)i;
    o[0] = q.position.x; o[1] = q.position.y; o[2] = q.position.z;
    o[3] = q.orientation.x; o[4] = q.orientation.y; o[5] = q.orientation.z; o[6] = q.orientation.w;
  }
  return 1;
}
// last mesh_msgs::MeshVertexCostsStamped of ANY topic whose layer name (`type`) is `name`: the values per vertex
int ref_pub_last_costs(const char* name, float* values, uint32_t cap, uint32_t* n)
{
  rclcpp::Publisher<mesh_msgs::msg::MeshVertexCostsStamped>* best = nullptr;
  for (auto* b : rclcpp::stub_publishers()) {
    auto* p = dynamic_cast<rclcpp::Publisher<mesh_msgs::msg::MeshVertexCostsStamped>*>(b);
    if (!p || p->count_ == 0 || p->last().type != name) continue;
    if (!best || p->seq_ > best->seq_) best = p;
  }
  if (!best) return 0;
  const auto& c = best->last().mesh_vertex_costs.costs;
  *n = (uint32_t)c.size();
  for (uint32_t i = 0; i < c.size() && i < cap; ++i) values[i] = c[i];
  return 1;
}

void ref_plugin_cancel(void* h) { auto* r = static_cast<Ref*>(h); if (r->plugin) r->plugin->cancel(); }
void ref_plugin_release(void* h) { static_cast<Ref*>(h)->plugin.reset(); }

// ---- CVP planner ----
int ref_cvp_init(void* h, const char* name)
{
  auto* r = static_cast<Ref*>(h);
  r->cvp = std::make_shared<cvp_mesh_planner::CVPMeshPlanner>();
  return r->cvp->initialize(name, r->map, r->node) ? 1 : 0;
}
// CVPMeshPlanner::waveFrontPropagation(start = wave seed position, goal = wave target position, path, message)
uint32_t ref_cvp(void* h, const float seed[3], const float target[3], float* path_pos, uint32_t* path_face, uint32_t cap, uint32_t* path_len)
{
  auto* r = static_cast<Ref*>(h);
  std::list<std::pair<mesh_map::Vector, lvr2::FaceHandle>> p;
  r->message.clear();
  const uint32_t code = r->cvp->waveFrontPropagation(mesh_map::Vector(seed[0], seed[1], seed[2]), mesh_map::Vector(target[0], target[1], target[2]), p, r->message);
  uint32_t n = 0;
  for (const auto& e : p) {
    if (n < cap) { path_pos[3 * n] = e.first.x; path_pos[3 * n + 1] = e.first.y; path_pos[3 * n + 2] = e.first.z; path_face[n] = e.second.idx(); }
    ++n;
  }
  *path_len = n;
  return code;
}
void ref_cvp_fields(void* h, float* dist, uint32_t* pred, float* direction, uint32_t* cutface, float* vecmap, uint8_t* has_vec)
{
  auto* r = static_cast<Ref*>(h);
  const uint32_t V = r->map->mesh()->numVertices();
  for (uint32_t v = 0; v < V; ++v) {
    const lvr2::VertexHandle vh(v);
    if (dist) { const auto d = std::as_const(r->cvp->potential_).get(vh); dist[v] = d ? *d : std::numeric_limits<float>::quiet_NaN(); }
    if (pred) { const auto p = std::as_const(r->cvp->predecessors_).get(vh); pred[v] = p ? p->idx() : 0xFFFFFFFFu; }
    if (direction) { const auto d = std::as_const(r->cvp->direction_).get(vh); direction[v] = d ? *d : 0.0f; }
    if (cutface) { const auto c = std::as_const(r->cvp->cutting_faces_).get(vh); cutface[v] = c ? c->idx() : 0xFFFFFFFFu; }
    if (vecmap) {
      const auto m = std::as_const(r->cvp->vector_map_).get(vh);
      if (has_vec) has_vec[v] = m ? 1 : 0;
      vecmap[3 * v] = m ? m->x : 0; vecmap[3 * v + 1] = m ? m->y : 0; vecmap[3 * v + 2] = m ? m->z : 0;
    }
  }
}
uint32_t ref_cvp_make_plan(void* h, const double start7[7], const double goal7[7], double* poses, uint32_t cap, uint32_t* n_poses, double* cost)
{
  auto* r = static_cast<Ref*>(h);
  std::vector<geometry_msgs::msg::PoseStamped> plan;
  r->message.clear();
  const uint32_t code = r->cvp->makePlan(pose_from(start7, r->map->mapFrame()), pose_from(goal7, r->map->mapFrame()), 0.0, plan, *cost, r->message);
  *n_poses = plan.size();
  for (uint32_t i = 0; i < plan.size() && i < cap; ++i) pose_to(plan[i].pose, poses + 7 * i);
  return code;
}
void ref_cvp_cancel(void* h) { static_cast<Ref*>(h)->cvp->cancel(); }

}  // extern "C"
