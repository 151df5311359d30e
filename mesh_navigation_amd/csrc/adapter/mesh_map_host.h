// mesh_map_host.h -- host-side stand-in for mesh_map::MeshMap, limited to what the planner plugins
// call (mesh_map/include/mesh_map/mesh_map.h): the data accessors (:276-353, :447) and the geometry
// queries that stay on the CPU by design: getNearestVertexHandle
// (mesh_map.cpp:1161-1174), getContainingFace (:1110-1159), meshAhead (:1070-1108),
// searchNeighbourFaces (:999-1068), directionAtPosition (:625-650), and the util helpers
// projectedBarycentricCoords (util.cpp:320-347), calculatePoseFromPosition (:267-298).
// On a robot the real MeshMap provides all of this; only MeshMapDevice (below) is new.
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "ros_stubs.h"

namespace mesh_map {

struct Vector {
  float x = 0, y = 0, z = 0;
  Vector() = default;
  Vector(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
  Vector operator+(const Vector& o) const { return Vector(x + o.x, y + o.y, z + o.z); }
  Vector operator-(const Vector& o) const { return Vector(x - o.x, y - o.y, z - o.z); }
  Vector operator*(float s) const { return Vector(x * s, y * s, z * s); }
  Vector operator/(float s) const { return Vector(x / s, y / s, z / s); }
  float dot(const Vector& o) const { return x * o.x + y * o.y + z * o.z; }
  Vector cross(const Vector& o) const { return Vector(y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x); }
  float length2() const { return x * x + y * y + z * z; }
  float length() const;
  Vector normalized() const { return *this / length(); }
  float distance2(const Vector& o) const { return (*this - o).length2(); }
  float distance(const Vector& o) const { return (*this - o).length(); }
};
using Normal = Vector;

inline Vector toVector(const geometry_msgs::msg::Point& p) { return Vector((float)p.x, (float)p.y, (float)p.z); }

bool projectedBarycentricCoords(const Vector& p, const std::array<Vector, 3>& vertices, std::array<float, 3>& bary, float& dist);
geometry_msgs::msg::Pose calculatePoseFromPosition(const Vector& current, const Vector& next, const Normal& normal, float& cost);

constexpr uint32_t kNoHandle = 0xFFFFFFFFu;

class MeshMap {
public:
  using Ptr = std::shared_ptr<MeshMap>;
  // flat mesh description (what lvr2::PMPMesh + the attribute maps hold)
  uint32_t V = 0, F = 0, E = 0;
  std::vector<float> positions;       // V*3
  std::vector<uint32_t> faces;        // F*3
  std::vector<uint32_t> edges;        // E*2, reference edge ids
  std::vector<float> vertex_normals;  // V*3
  std::vector<float> face_normals;    // F*3
  std::vector<float> vertex_costs;    // V   (MeshMap::vertexCosts)
  std::vector<float> edge_weights;    // E   (MeshMap::edgeWeights)
  std::vector<uint8_t> invalid;       // V   (MeshMap::invalid)
  // optional: rows of PMPMesh::getFacesOfVertex (circulator order) as CSR; empty -> derived from `faces`
  std::vector<uint32_t> face_circulation_ptr, face_circulation;
  // Change counter of vertex_costs / edge_weights / invalid.  The real MeshMap has none; the
  // integration bumps it from MeshMap::layerChanged / computeEdgeWeights.  0 = unknown: the device
  // mirror then falls back to hashing the arrays (8-byte words) on every plan.
  uint64_t cost_version = 0;
  // Repulsive vector fields of layers (AbstractLayer::vectorAt, mesh_map.cpp:1099-1102): the Inflation layer's
  // distance + vector maps with their has-value flags and the parameters vectorAt reads (inflation_layer.cpp:493-521)
  struct LayerVectorField {
    std::vector<float> distances; std::vector<uint8_t> has_distance;    // V   (distances_)
    std::vector<float> vectors; std::vector<uint8_t> has_vector;        // V*3 (vector_map_)
    double inscribed_radius = 0.25, inflation_radius = 0.4, lethal_value = 1.0, inscribed_value = 0.99;
    bool repulsive_field = true;
  };
  std::vector<LayerVectorField> layer_fields;
  std::string map_frame = "map";
  // vector map set by the planners for the controller (MeshMap::setVectorMap, mesh_map.cpp:620-623)
  std::vector<float> vector_map;      // V*3
  std::vector<uint8_t> vector_map_set;// V

  void finalize();                    // builds vertex->face adjacency and the nearest-vertex grid
  const std::string& mapFrame() const { return map_frame; }
  geometry_msgs::msg::PoseStamped transformToMapFrame(const geometry_msgs::msg::PoseStamped& p) const { return p; }  // :1312-1328, already in map frame

  Vector vertex(uint32_t v) const { return Vector(positions[3 * (size_t)v], positions[3 * (size_t)v + 1], positions[3 * (size_t)v + 2]); }
  Vector vertexNormal(uint32_t v) const { return Vector(vertex_normals[3 * (size_t)v], vertex_normals[3 * (size_t)v + 1], vertex_normals[3 * (size_t)v + 2]); }
  Vector faceNormal(uint32_t f) const { return Vector(face_normals[3 * (size_t)f], face_normals[3 * (size_t)f + 1], face_normals[3 * (size_t)f + 2]); }
  std::array<Vector, 3> facePositions(uint32_t f) const { return { vertex(faces[3 * (size_t)f]), vertex(faces[3 * (size_t)f + 1]), vertex(faces[3 * (size_t)f + 2]) }; }

  uint32_t getNearestVertexHandle(const Vector& pos) const;                         // :1161-1174
  uint32_t getContainingFace(const Vector& position, float max_dist) const;         // :1110-1159
  // throws MapPanic where the reference's attribute-map lookup would panic (a layer without a value for a vertex)
  bool meshAhead(Vector& pos, uint32_t& face, float step_size) const;               // :1070-1108
  struct MapPanic { };                                                              // lvr2::PanicException stand-in
  void setVectorMap(const std::vector<float>& vm, const std::vector<uint8_t>& set) { vector_map = vm; vector_map_set = set; }

private:
  bool searchNeighbourFaces(const Vector& pos, uint32_t face, float max_radius, float max_dist, uint32_t& found,
                            std::array<float, 3>& bary) const;                      // :999-1068
  Vector layerVectorAt(const LayerVectorField& L, const uint32_t vs[3], const std::array<float, 3>& bary) const;
  std::vector<uint32_t> vf_ptr_, vf_;   // vertex -> faces, half-edge circulator order (getFacesOfVertex)
  mutable std::vector<uint32_t> seen_;  // searchNeighbourFaces: per-face stamp of the current search (no O(F) clear per step)
  mutable uint32_t seen_gen_ = 0;
  // uniform grid over xy for the 1-NN query (stands in for the nanoflann kd-tree, :307-309)
  float gx0_ = 0, gy0_ = 0, gcell_ = 1;
  uint32_t gnx_ = 1, gny_ = 1;
  std::vector<uint32_t> gptr_, gidx_;
};

}  // namespace mesh_map
