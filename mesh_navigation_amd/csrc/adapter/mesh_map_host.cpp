// mesh_map_host.cpp -- see mesh_map_host.h.  Float arithmetic follows the reference expressions
// (lvr2::BaseVector<float> component ops, CONVENTION: lvr2 is not vendored).
#include "mesh_map_host.h"
#include "../mnav_build.h"

#include <algorithm>
#include <cfloat>
#include <cmath>

namespace mesh_map {

float Vector::length() const { return std::sqrt(length2()); }

// mesh_map/src/util.cpp:320-347
bool projectedBarycentricCoords(const Vector& p, const std::array<Vector, 3>& vertices, std::array<float, 3>& bary, float& dist)
{
  const Vector& a = vertices[0];
  const Vector& b = vertices[1];
  const Vector& c = vertices[2];
  const Vector u = b - a, v = c - a, w = p - a;
  const Vector n = u.cross(v);
  const float oneOver4ASquared = (float)(1.0 / (double)n.dot(n));   // :333
  const float gamma = u.cross(w).dot(n) * oneOver4ASquared;         // :335
  const float beta = w.cross(v).dot(n) * oneOver4ASquared;          // :337
  const float alpha = 1 - gamma - beta;                             // :338
  bary = { alpha, beta, gamma };
  dist = n.dot(w) / n.length();                                     // :341
  const float EPSILON = 0.01f;                                      // :343
  return ((0 - EPSILON <= alpha) && (alpha <= 1 + EPSILON) && (0 - EPSILON <= beta) && (beta <= 1 + EPSILON) &&
          (0 - EPSILON <= gamma) && (gamma <= 1 + EPSILON));
}

// tf2::Matrix3x3::getRotation + normalize (tf2 is not vendored: the published Bullet algorithm)
static geometry_msgs::msg::Quaternion basisToQuaternion(const double m[3][3])
{
  double t[4];
  const double trace = m[0][0] + m[1][1] + m[2][2];
  if (trace > 0.0) {
    double s = std::sqrt(trace + 1.0);
    t[3] = s * 0.5; s = 0.5 / s;
    t[0] = (m[2][1] - m[1][2]) * s; t[1] = (m[0][2] - m[2][0]) * s; t[2] = (m[1][0] - m[0][1]) * s;
  } else {
    const int i = m[0][0] < m[1][1] ? (m[1][1] < m[2][2] ? 2 : 1) : (m[0][0] < m[2][2] ? 2 : 0);
    const int j = (i + 1) % 3, k = (i + 2) % 3;
    double s = std::sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0);
    t[i] = s * 0.5; s = 0.5 / s;
    t[3] = (m[k][j] - m[j][k]) * s; t[j] = (m[j][i] + m[i][j]) * s; t[k] = (m[k][i] + m[i][k]) * s;
  }
  const double len = std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2] + t[3] * t[3]);
  geometry_msgs::msg::Quaternion q;
  q.x = t[0] / len; q.y = t[1] / len; q.z = t[2] / len; q.w = t[3] / len;
  return q;
}

// mesh_map/src/util.cpp:267-298
geometry_msgs::msg::Pose calculatePoseFromPosition(const Vector& current, const Vector& next, const Normal& normal, float& cost)
{
  const Vector direction = next - current;                          // :295
  cost = direction.length();                                        // :296
  const Normal ez = normal.normalized();                            // :269
  const Normal ey = normal.cross(direction).normalized();           // :270
  const Normal ex = ey.cross(normal).normalized();                  // :271
  const double basis[3][3] = { { ex.x, ey.x, ez.x }, { ex.y, ey.y, ez.y }, { ex.z, ey.z, ez.z } };   // :273
  geometry_msgs::msg::Pose pose;
  pose.orientation = basisToQuaternion(basis);                      // :278-279
  pose.position.x = current.x; pose.position.y = current.y; pose.position.z = current.z;   // :275
  return pose;
}

void MeshMap::finalize()
{
  V = (uint32_t)(positions.size() / 3); F = (uint32_t)(faces.size() / 3); E = (uint32_t)(edges.size() / 2);
  if (invalid.size() != V) invalid.assign(V, 0);
  // getFacesOfVertex rows: the caller's (the real lvr2 mesh) if given, else the half-edge replay of the face list
  if (face_circulation_ptr.size() == (size_t)V + 1 && face_circulation.size() == faces.size()) {
    vf_ptr_ = face_circulation_ptr; vf_ = face_circulation;
  } else {
    mnav::FaceCirculation c = mnav::build_face_circulation(V, F, faces.data());
    vf_ptr_ = std::move(c.ptr); vf_ = std::move(c.faces);
  }
  // uniform grid for nearest-vertex queries
  float x1 = -FLT_MAX, y1 = -FLT_MAX; gx0_ = FLT_MAX; gy0_ = FLT_MAX;
  for (uint32_t v = 0; v < V; ++v) {
    gx0_ = std::min(gx0_, positions[3 * (size_t)v]); x1 = std::max(x1, positions[3 * (size_t)v]);
    gy0_ = std::min(gy0_, positions[3 * (size_t)v + 1]); y1 = std::max(y1, positions[3 * (size_t)v + 1]);
  }
  if (V == 0) { gx0_ = gy0_ = 0; x1 = y1 = 1; }
  const double area = std::max(1e-12, (double)(x1 - gx0_) * (double)(y1 - gy0_));
  gcell_ = (float)std::max(1e-6, std::sqrt(area / std::max<uint32_t>(V, 1)) * 2.0);
  gnx_ = (uint32_t)std::min(4096.0, std::floor((double)(x1 - gx0_) / gcell_) + 1.0);
  gny_ = (uint32_t)std::min(4096.0, std::floor((double)(y1 - gy0_) / gcell_) + 1.0);
  auto cell = [&](uint32_t v) {
    const uint32_t cx = std::min(gnx_ - 1, (uint32_t)std::max(0.0f, (positions[3 * (size_t)v] - gx0_) / gcell_));
    const uint32_t cy = std::min(gny_ - 1, (uint32_t)std::max(0.0f, (positions[3 * (size_t)v + 1] - gy0_) / gcell_));
    return cy * gnx_ + cx;
  };
  gptr_.assign((size_t)gnx_ * gny_ + 1, 0);
  for (uint32_t v = 0; v < V; ++v) gptr_[cell(v) + 1]++;
  for (size_t c = 0; c < (size_t)gnx_ * gny_; ++c) gptr_[c + 1] += gptr_[c];
  gidx_.resize(V);
  std::vector<uint32_t> gf((size_t)gnx_ * gny_, 0);
  for (uint32_t v = 0; v < V; ++v) { const uint32_t c = cell(v); gidx_[gptr_[c] + gf[c]++] = v; }   // ascending id per cell
}

// mesh_map.cpp:1161-1174: 1-NN over vertex positions (L2, float accumulation); ties -> lowest id
uint32_t MeshMap::getNearestVertexHandle(const Vector& pos) const
{
  if (V == 0) return kNoHandle;
  // the reference's kd-tree always returns the nearest vertex, also for queries off the mesh: start at the cell the
  // query clamps into; a ring r of cells around it lies at least (r-1) cells plus the clamping distance away
  const int qx = (int)std::floor((pos.x - gx0_) / gcell_), qy = (int)std::floor((pos.y - gy0_) / gcell_);
  const int cx = std::min((int)gnx_ - 1, std::max(0, qx)), cy = std::min((int)gny_ - 1, std::max(0, qy));
  const float off_x = (float)std::abs(qx - cx) * gcell_, off_y = (float)std::abs(qy - cy) * gcell_;
  const float off = std::max(0.0f, std::max(off_x, off_y) - gcell_);
  uint32_t best = kNoHandle; float bd = FLT_MAX;
  const int rmax = (int)std::max(gnx_, gny_) + 1;                    // covers the whole grid from any cell
  for (int r = 0; r <= rmax; ++r) {
    // once a candidate exists, a ring further out than its distance cannot improve it
    if (best != kNoHandle) { const float ring = (float)(r - 1) * gcell_ + off; if (ring > 0 && ring * ring > bd) break; }
    for (int y = cy - r; y <= cy + r; ++y)
      for (int x = cx - r; x <= cx + r; ++x) {
        if (std::max(std::abs(x - cx), std::abs(y - cy)) != r) continue;
        if (x < 0 || y < 0 || x >= (int)gnx_ || y >= (int)gny_) continue;
        const uint32_t c = (uint32_t)y * gnx_ + (uint32_t)x;
        for (uint32_t i = gptr_[c]; i < gptr_[c + 1]; ++i) {
          const uint32_t v = gidx_[i];
          const float dx = pos.x - positions[3 * (size_t)v], dy = pos.y - positions[3 * (size_t)v + 1], dz = pos.z - positions[3 * (size_t)v + 2];
          const float d = dx * dx + dy * dy + dz * dz;
          if (d < bd || (d == bd && v < best)) { bd = d; best = v; }
        }
      }
  }
  return best;
}

// mesh_map.cpp:1110-1159 (max_dist is not used by the reference's search either)
uint32_t MeshMap::getContainingFace(const Vector& position, float /*max_dist*/) const
{
  const uint32_t vH = getNearestVertexHandle(position);
  if (vH == kNoHandle) return kNoHandle;
  float lowest = FLT_MAX;                                           // :1131
  uint32_t best = kNoHandle;
  for (uint32_t i = vf_ptr_[vH]; i < vf_ptr_[vH + 1]; ++i) {        // :1135
    const uint32_t f = vf_[i];
    std::array<float, 3> bary; float dist = 0;
    if (projectedBarycentricCoords(position, facePositions(f), bary, dist) && dist < lowest) { lowest = dist; best = f; }   // :1140-1147
  }
  return best;
}

// mesh_map.cpp:999-1068
bool MeshMap::searchNeighbourFaces(const Vector& pos, uint32_t face, float max_radius, float max_dist, uint32_t& found,
                                   std::array<float, 3>& bary) const
{
  std::vector<uint32_t> possible{ face };
  if (seen_.size() != F) { seen_.assign(F, 0u); seen_gen_ = 0; }
  if (++seen_gen_ == 0u) { std::fill(seen_.begin(), seen_.end(), 0u); seen_gen_ = 1; }   // SparseFaceMap<bool> in_list_map (:1026)
  const uint32_t gen = seen_gen_;
  seen_[face] = gen;
  Vector center(0, 0, 0);
  const auto start = facePositions(face);
  for (const auto& v : start) center = center + v;                  // :1010-1013
  center = center / 3;                                              // :1014
  float vertex_center_max = 0;
  for (const auto& v : start) vertex_center_max = std::max(vertex_center_max, v.distance(center));   // :1017-1020
  const float ext_radius = max_radius + vertex_center_max;          // :1022
  const float max_radius_sq = ext_radius * ext_radius;              // :1023
  for (size_t it = 0; it < possible.size(); ++it) {                 // :1031
    const uint32_t f = possible[it];
    float dist;
    if (projectedBarycentricCoords(pos, facePositions(f), bary, dist) && std::fabs(dist) < max_dist) { found = f; return true; }   // :1035
    for (int k = 0; k < 3; ++k) {                                   // :1042
      const uint32_t vertex = faces[3 * (size_t)f + k];
      if (center.distance2(this->vertex(vertex)) < max_radius_sq)   // :1044
        for (uint32_t i = vf_ptr_[vertex]; i < vf_ptr_[vertex + 1]; ++i) {   // :1048-1049
          const uint32_t nn = vf_[i];
          if (seen_[nn] != gen) { possible.push_back(nn); seen_[nn] = gen; }   // :1051-1055
        }
    }
  }
  return false;
}

// InflationLayer::vectorAt(handles, barycentric coords), inflation_layer.cpp:493-521.  lvr2 attribute maps panic on
// a key without a value (AttributeMap::operator[]): distances_ / vector_map_ only hold the vertices the inflation
// wave reached, so on faces beyond it the reference's back-tracking ends with "HalfEdgeMesh panicked!" (cvp :944).
Vector MeshMap::layerVectorAt(const LayerVectorField& L, const uint32_t vs[3], const std::array<float, 3>& bary) const
{
  if (!L.repulsive_field) return Vector();                           // :496
  for (int k = 0; k < 3; ++k) if (!L.has_distance[vs[k]]) throw MapPanic();
  const float distance = L.distances[vs[0]] * bary[0] + L.distances[vs[1]] * bary[1] + L.distances[vs[2]] * bary[2];   // :499
  if ((double)distance > L.inflation_radius) return Vector();       // :501
  for (int k = 0; k < 3; ++k) if (!L.has_vector[vs[k]]) throw MapPanic();
  auto vec = [&](int k) { return Vector(L.vectors[3 * (size_t)vs[k]], L.vectors[3 * (size_t)vs[k] + 1], L.vectors[3 * (size_t)vs[k] + 2]); };
  const Vector comb = vec(0) * bary[0] + vec(1) * bary[1] + vec(2) * bary[2];
  if ((double)distance > L.inscribed_radius) {                       // :505
    const float alpha = (float)(((double)std::sqrt(distance) - L.inscribed_radius) / (L.inflation_radius - L.inscribed_radius) * M_PI);   // :507-508
    return comb * (float)L.inscribed_value * (std::cos(alpha) + 1) / 2.0f;   // :509-510, three float vector operations
  }
  if (distance > 0) return comb * (float)L.inscribed_value;          // :514-517
  return comb * (float)L.lethal_value;                               // :520
}

// mesh_map.cpp:1070-1108 with directionAtPosition :625-650 and the layers' vectorAt (:1099-1102).
bool MeshMap::meshAhead(Vector& pos, uint32_t& face, float step_size) const
{
  std::array<float, 3> bary; float dist;
  if (projectedBarycentricCoords(pos, facePositions(face), bary, dist)) {   // :1075
  } else {
    uint32_t nf;
    if (!searchNeighbourFaces(pos, face, step_size, 0.4f, nf, bary)) return false;   // :1079,:1090-1093
    face = nf;
    const auto p = facePositions(nf);
    pos = p[0] * bary[0] + p[1] * bary[1] + p[2] * bary[2];         // :1087
  }
  const uint32_t* vs = &faces[3 * (size_t)face];
  if (vector_map_set.size() != V) return false;
  if (!(vector_map_set[vs[0]] || vector_map_set[vs[1]] || vector_map_set[vs[2]])) return false;   // :633
  Vector vec(0, 0, 0);
  for (int k = 0; k < 3; ++k)
    if (vector_map_set[vs[k]]) vec = vec + Vector(vector_map[3 * (size_t)vs[k]], vector_map[3 * (size_t)vs[k] + 1], vector_map[3 * (size_t)vs[k] + 2]) * bary[k];   // :636-638
  if (!(std::isfinite(vec.x) && std::isfinite(vec.y) && std::isfinite(vec.z))) return false;      // :639
  Vector dir = vec.normalized();                                    // :1096
  for (const auto& L : layer_fields) dir = dir + layerVectorAt(L, vs, bary);   // :1099-1102 every layer's vectorAt
  dir = dir.normalized();                                           // :1103
  pos = pos + dir * step_size;                                      // :1104
  return true;
}

}  // namespace mesh_map
