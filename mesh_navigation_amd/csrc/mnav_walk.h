// mnav_walk.h -- the consumer side of the vector field, shared by the device kernel (k_backtrack, mnav.hip) and the
// host-side mirror the CPU tests run (oracle/schedule_model.cpp): CVPMeshPlanner's back-tracking loop
// (cvp_mesh_planner.cpp:920-951) = repeated MeshMap::meshAhead (mesh_map.cpp:1070-1108) with
// projectedBarycentricCoords (util.cpp:320-347), searchNeighbourFaces (mesh_map.cpp:999-1068), directionAtPosition
// (:625-650) and InflationLayer::vectorAt (inflation_layer.cpp:493-521).  float32 arithmetic in the reference's operation
// order (lvr2::BaseVector<float> component operations; no contraction; correctly rounded sqrt / division), so the walked
// positions are the host's bit for bit.
#pragma once
#include <cmath>
#include <cstdint>

#include "mnav_eval.h"

namespace mnav {

struct W3 { float x, y, z; };
MNAV_HD W3 w3(float x, float y, float z) { W3 r; r.x = x; r.y = y; r.z = z; return r; }
MNAV_HD W3 w3_load(const float* p) { return w3(p[0], p[1], p[2]); }
MNAV_HD W3 w3_add(W3 a, W3 b) { return w3(a.x + b.x, a.y + b.y, a.z + b.z); }
MNAV_HD W3 w3_sub(W3 a, W3 b) { return w3(a.x - b.x, a.y - b.y, a.z - b.z); }
MNAV_HD W3 w3_scale(W3 a, float s) { return w3(a.x * s, a.y * s, a.z * s); }
MNAV_HD W3 w3_div(W3 a, float s) { return w3(a.x / s, a.y / s, a.z / s); }
MNAV_HD float w3_dot(W3 a, W3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
MNAV_HD W3 w3_cross(W3 a, W3 b) { return w3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
MNAV_HD float w3_length2(W3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
MNAV_HD float w3_length(W3 a) { return sqrtf(w3_length2(a)); }
MNAV_HD W3 w3_normalized(W3 a) { return w3_div(a, w3_length(a)); }
MNAV_HD float w3_distance2(W3 a, W3 b) { return w3_length2(w3_sub(a, b)); }

// what the walk reads (device or host pointers)
struct WalkMesh {
  const float* xyz; const uint32_t* faces; const uint32_t* vf_ptr; const uint32_t* vf;   // positions, face vertex triples, getFacesOfVertex rows
  uint32_t V, F;
};
// the vector field of a plan: a vertex "has a vector" when its entry is not the all-zero row the vector-map kernels write
// for vertices the wave never set, or when it is one of the three seed-face vertices (their entries always exist, cvp :722-724)
struct WalkField {
  const float* vecmap; uint32_t seed_vs[3];
};
// InflationLayer's repulsive field (optional): distances / vectors per vertex + flags, configuration
struct WalkInflation {
  const float* distances; const float* vectors;   // null: no such layer.  A vertex without a distance entry holds +inf
  const uint8_t* has_vector;                      // 1 = the layer's vector map holds an entry (null: all do)
  double inflation_radius, inscribed_radius, inscribed_value, lethal_value; int repulsive_field;
};

constexpr int kWalkListCap = 1024;   // faces searchNeighbourFaces may collect around one position (a few dozen on real meshes)
enum : int { kWalkOk = 1, kWalkLost = 0, kWalkPanic = -1, kWalkListFull = -2 };

// mesh_map/src/util.cpp:320-347
MNAV_HD bool walk_bary(const WalkMesh& M, W3 p, uint32_t f, float bary[3], float* dist)
{
  const W3 a = w3_load(M.xyz + 3 * (size_t)M.faces[3 * (size_t)f]), b = w3_load(M.xyz + 3 * (size_t)M.faces[3 * (size_t)f + 1]),
           c = w3_load(M.xyz + 3 * (size_t)M.faces[3 * (size_t)f + 2]);
  const W3 u = w3_sub(b, a), v = w3_sub(c, a), w = w3_sub(p, a);
  const W3 n = w3_cross(u, v);
  const float oneOver4ASquared = (float)(1.0 / (double)w3_dot(n, n));   // :333
  const float gamma = w3_dot(w3_cross(u, w), n) * oneOver4ASquared;     // :335
  const float beta = w3_dot(w3_cross(w, v), n) * oneOver4ASquared;      // :337
  const float alpha = 1 - gamma - beta;                                 // :338
  bary[0] = alpha; bary[1] = beta; bary[2] = gamma;
  *dist = w3_dot(n, w) / w3_length(n);                                  // :341
  const float EPSILON = 0.01f;                                          // :343
  return (0 - EPSILON <= alpha) && (alpha <= 1 + EPSILON) && (0 - EPSILON <= beta) && (beta <= 1 + EPSILON) &&
         (0 - EPSILON <= gamma) && (gamma <= 1 + EPSILON);
}

// mesh_map.cpp:999-1068: breadth-first over the faces around `face` within max_radius (+ the face's own extent); the
// reference's SparseFaceMap "already listed" test is a linear scan of the (short) list here.  kNone: nothing found;
// *status = kWalkListFull when the list capacity ended the search.
MNAV_HD void walk_search_setup(const WalkMesh& M, uint32_t face, float max_radius, W3* center_out, float* max_radius_sq)
{
  W3 center = w3(0, 0, 0);
  for (int k = 0; k < 3; ++k) center = w3_add(center, w3_load(M.xyz + 3 * (size_t)M.faces[3 * (size_t)face + k]));   // :1010-1013
  center = w3_div(center, 3);                                           // :1014
  float vertex_center_max = 0;
  for (int k = 0; k < 3; ++k)
    vertex_center_max = fmaxf(vertex_center_max, sqrtf(w3_distance2(w3_load(M.xyz + 3 * (size_t)M.faces[3 * (size_t)face + k]), center)));   // :1017-1020
  const float ext_radius = max_radius + vertex_center_max;              // :1022
  *max_radius_sq = ext_radius * ext_radius;                             // :1023
  *center_out = center;
}

#if defined(__HIP_DEVICE_COMPILE__)
constexpr int kWalkCand = 32;                                           // candidate slots per listed face (3 vertex rows; valence 6 -> 18)
constexpr int kWalkScratchWords = kWalkListCap + 64 * kWalkCand + 64;   // list | candidates of the 64 faces of a round | their counts

// "is nn in list[0..n)": the 64 lanes of the wave divide the scan
__device__ __forceinline__ bool walk_listed(const uint32_t* list, int n, uint32_t nn)
{
  bool hit = false;
  for (int q = (int)(threadIdx.x & 63u); q < n; q += 64) hit |= list[q] == nn;
  return __any(hit);
}

// The wave version of the search (k_backtrack: 64 lanes in lockstep on one plan).  The sequential loop tests list[it],
// expands it, tests list[it+1], ...: the LIST ORDER does not depend on the tests, only where the loop stops does.  So a
// round takes the next <= 64 listed faces, one per lane: every lane tests its face and gathers the faces around its
// three vertices (the dependent global loads of 64 faces in flight at once); the first lane whose test passes is the
// sequential loop's answer.  If none passes, the gathered candidates are appended in the sequential order (face by face,
// vertex by vertex, row order), each after a lane-parallel "already listed" scan.
__device__ inline uint32_t walk_search_faces(const WalkMesh& M, W3 pos, uint32_t face, float max_radius, float max_dist, float bary_out[3],
                                             uint32_t* list, int* status)
{
  const int lane = (int)(threadIdx.x & 63u);
  uint32_t* cand = list + kWalkListCap;
  uint32_t* cnt = cand + 64 * kWalkCand;
  int n = 0, it = 0;
  list[n++] = face;
  W3 center; float max_radius_sq;
  walk_search_setup(M, face, max_radius, &center, &max_radius_sq);
  while (it < n) {
    const int m = n - it < 64 ? n - it : 64;
    float bary[3] = { 0.f, 0.f, 0.f }, dist = 0.f;
    bool pass = false;
    if (lane < m) {
      const uint32_t f = list[it + lane];
      pass = walk_bary(M, pos, f, bary, &dist) && fabsf(dist) < max_dist;   // :1035
      uint32_t c = 0; bool over = false;
      for (int k = 0; k < 3; ++k) {                                     // :1042
        const uint32_t vertex = M.faces[3 * (size_t)f + k];
        if (w3_distance2(center, w3_load(M.xyz + 3 * (size_t)vertex)) < max_radius_sq)   // :1044
          for (uint32_t i = M.vf_ptr[vertex]; i < M.vf_ptr[vertex + 1]; ++i) {           // :1048-1049
            if (c < (uint32_t)kWalkCand) cand[lane * kWalkCand + c++] = M.vf[i]; else over = true;
          }
      }
      cnt[lane] = over ? 0xFFFFFFFFu : c;
    }
    __syncthreads();                                                    // (one wave per workgroup: orders the LDS traffic of the round)
    const unsigned long long mask = __ballot(pass);
    if (mask) {
      const int first = __ffsll((long long)mask) - 1;
      for (int k = 0; k < 3; ++k) bary_out[k] = __shfl(bary[k], first);
      return list[it + first];
    }
    for (int j = 0; j < m; ++j) {                                       // append in the sequential order
      const uint32_t c = cnt[j];
      if (c != 0xFFFFFFFFu) {
        for (uint32_t t = 0; t < c; ++t) {
          const uint32_t nn = cand[j * kWalkCand + t];
          if (!walk_listed(list, n, nn)) {                              // :1051-1055
            if (n >= kWalkListCap) { *status = kWalkListFull; return kNone; }
            list[n++] = nn;
          }
        }
      } else {                                                          // a vertex row longer than the slots: straight from memory
        const uint32_t f = list[it + j];
        for (int k = 0; k < 3; ++k) {
          const uint32_t vertex = M.faces[3 * (size_t)f + k];
          if (w3_distance2(center, w3_load(M.xyz + 3 * (size_t)vertex)) < max_radius_sq)
            for (uint32_t i = M.vf_ptr[vertex]; i < M.vf_ptr[vertex + 1]; ++i) {
              const uint32_t nn = M.vf[i];
              if (!walk_listed(list, n, nn)) {
                if (n >= kWalkListCap) { *status = kWalkListFull; return kNone; }
                list[n++] = nn;
              }
            }
        }
      }
    }
    __syncthreads();
    it += m;                                                            // :1063
  }
  return kNone;
}
#else
constexpr int kWalkScratchWords = kWalkListCap;
inline uint32_t walk_search_faces(const WalkMesh& M, W3 pos, uint32_t face, float max_radius, float max_dist, float bary_out[3],
                                  uint32_t* list, int* status)
{
  int n = 0, it = 0;
  list[n++] = face;
  W3 center; float max_radius_sq;
  walk_search_setup(M, face, max_radius, &center, &max_radius_sq);
  while (it < n) {                                                      // :1031
    const uint32_t f = list[it];
    float bary[3], dist;
    if (walk_bary(M, pos, f, bary, &dist) && fabsf(dist) < max_dist) {  // :1035
      bary_out[0] = bary[0]; bary_out[1] = bary[1]; bary_out[2] = bary[2];
      return f;
    }
    for (int k = 0; k < 3; ++k) {                                       // :1042
      const uint32_t vertex = M.faces[3 * (size_t)f + k];
      if (w3_distance2(center, w3_load(M.xyz + 3 * (size_t)vertex)) < max_radius_sq) {   // :1044
        for (uint32_t i = M.vf_ptr[vertex]; i < M.vf_ptr[vertex + 1]; ++i) {             // :1048-1049
          const uint32_t nn = M.vf[i];
          bool seen = false;
          for (int q = 0; q < n; ++q) if (list[q] == nn) { seen = true; break; }
          if (!seen) {                                                   // :1051-1055
            if (n >= kWalkListCap) { *status = kWalkListFull; return kNone; }
            list[n++] = nn;
          }
        }
      }
    }
    ++it;                                                               // :1063
  }
  return kNone;
}
#endif

// inflation_layer.cpp:493-521; *panic where lvr2's attribute maps would throw (a vertex without an entry, :499 / :503)
MNAV_HD W3 walk_inflation_vector(const WalkInflation& L, const uint32_t vs[3], const float bary[3], bool* panic)
{
  if (!L.distances || !L.repulsive_field) return w3(0, 0, 0);          // :496
  for (int k = 0; k < 3; ++k) if (!std::isfinite(L.distances[vs[k]])) { *panic = true; return w3(0, 0, 0); }
  const float distance = L.distances[vs[0]] * bary[0] + L.distances[vs[1]] * bary[1] + L.distances[vs[2]] * bary[2];   // :499
  if ((double)distance > L.inflation_radius) return w3(0, 0, 0);        // :501
  for (int k = 0; k < 3; ++k) if (L.has_vector && L.has_vector[vs[k]] != 1) { *panic = true; return w3(0, 0, 0); }
  const W3 comb = w3_add(w3_add(w3_scale(w3_load(L.vectors + 3 * (size_t)vs[0]), bary[0]), w3_scale(w3_load(L.vectors + 3 * (size_t)vs[1]), bary[1])),
                         w3_scale(w3_load(L.vectors + 3 * (size_t)vs[2]), bary[2]));
  if ((double)distance > L.inscribed_radius) {                          // :505
    const float alpha = (float)(((double)sqrtf(distance) - L.inscribed_radius) / (L.inflation_radius - L.inscribed_radius) * 3.14159265358979323846);   // :507-508
    return w3_div(w3_scale(w3_scale(comb, (float)L.inscribed_value), cosf_ref(alpha) + 1), 2.0f);   // :509-510, three float vector operations; the host libm's cos bits (mnav_eval.h)
  }
  if (distance > 0) return w3_scale(comb, (float)L.inscribed_value);    // :514-517
  return w3_scale(comb, (float)L.lethal_value);                         // :520
}

MNAV_HD bool walk_has_vector(const WalkField& Fd, uint32_t v)
{
  if (v == Fd.seed_vs[0] || v == Fd.seed_vs[1] || v == Fd.seed_vs[2]) return true;
  const float* q = Fd.vecmap + 3 * (size_t)v;
  return q[0] != 0.f || q[1] != 0.f || q[2] != 0.f;
}

// mesh_map.cpp:1070-1108 with directionAtPosition :625-650 and the layer's vectorAt (:1099-1102)
MNAV_HD int walk_mesh_ahead(const WalkMesh& M, const WalkField& Fd, const WalkInflation& L, W3* pos, uint32_t* face, float step_size, uint32_t* list)
{
  float bary[3], dist;
  if (!walk_bary(M, *pos, *face, bary, &dist)) {                        // :1075
    int status = kWalkLost;
    const uint32_t nf = walk_search_faces(M, *pos, *face, step_size, 0.4f, bary, list, &status);   // :1079
    if (nf == kNone) return status;                                     // :1090-1093
    *face = nf;
    const uint32_t* fv = M.faces + 3 * (size_t)nf;
    *pos = w3_add(w3_add(w3_scale(w3_load(M.xyz + 3 * (size_t)fv[0]), bary[0]), w3_scale(w3_load(M.xyz + 3 * (size_t)fv[1]), bary[1])),
                  w3_scale(w3_load(M.xyz + 3 * (size_t)fv[2]), bary[2]));   // :1087
  }
  const uint32_t* vs = M.faces + 3 * (size_t)*face;
  const bool h0 = walk_has_vector(Fd, vs[0]), h1 = walk_has_vector(Fd, vs[1]), h2 = walk_has_vector(Fd, vs[2]);
  if (!(h0 || h1 || h2)) return kWalkLost;                              // :633
  W3 vec = w3(0, 0, 0);
  if (h0) vec = w3_add(vec, w3_scale(w3_load(Fd.vecmap + 3 * (size_t)vs[0]), bary[0]));   // :636-638
  if (h1) vec = w3_add(vec, w3_scale(w3_load(Fd.vecmap + 3 * (size_t)vs[1]), bary[1]));
  if (h2) vec = w3_add(vec, w3_scale(w3_load(Fd.vecmap + 3 * (size_t)vs[2]), bary[2]));
  if (!(std::isfinite(vec.x) && std::isfinite(vec.y) && std::isfinite(vec.z))) return kWalkLost;   // :639
  W3 dir = w3_normalized(vec);                                          // :1096
  if (L.distances) {
    bool panic = false;
    dir = w3_add(dir, walk_inflation_vector(L, vs, bary, &panic));      // :1099-1102
    if (panic) return kWalkPanic;
  }
  dir = w3_normalized(dir);                                             // :1103
  *pos = w3_add(*pos, w3_scale(dir, step_size));                        // :1104
  return kWalkOk;
}

// cvp_mesh_planner.cpp:920-951: from the target position back to the wave seed.  Writes the visited (position, face)
// pairs in WALK order (target first), the caller reverses; *n_out counts them.  Returns kWalkOk when the walk came within
// step_width of the seed (the seed pair itself is appended as the last entry), else the failing status.
MNAV_HD int walk_backtrack(const WalkMesh& M, const WalkField& Fd, const WalkInflation& L, W3 seed, uint32_t seed_face, W3 target, uint32_t target_face,
                           double step_width, uint32_t cap, float* pos_out, uint32_t* face_out, uint32_t* n_out, uint32_t* list)
{
  uint32_t face = target_face;                                          // :922
  W3 pos = target;                                                      // :923
  uint32_t n = 0;
  int status = kWalkOk;
  if (n < cap) { pos_out[3 * (size_t)n] = pos.x; pos_out[3 * (size_t)n + 1] = pos.y; pos_out[3 * (size_t)n + 2] = pos.z; face_out[n] = face; }
  ++n;                                                                  // :924
  while ((double)w3_distance2(pos, seed) > step_width) {                // :927 (squared distance against the width, as is)
    status = walk_mesh_ahead(M, Fd, L, &pos, &face, (float)step_width, list);   // :933
    if (status != kWalkOk) break;                                       // :937-942
    if (n < cap) { pos_out[3 * (size_t)n] = pos.x; pos_out[3 * (size_t)n + 1] = pos.y; pos_out[3 * (size_t)n + 2] = pos.z; face_out[n] = face; }
    ++n;                                                                // :935
    if (n >= cap) { status = kWalkLost; break; }                        // guard against endless loops (the field circles)
  }
  if (status == kWalkOk) {
    if (n < cap) { pos_out[3 * (size_t)n] = seed.x; pos_out[3 * (size_t)n + 1] = seed.y; pos_out[3 * (size_t)n + 2] = seed.z; face_out[n] = seed_face; }
    ++n;                                                                // :951
  }
  *n_out = n < cap ? n : cap;
  return status;
}

}  // namespace mnav
