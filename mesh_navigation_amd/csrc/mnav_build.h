// mnav_build.h -- host-side construction of the device index arrays from the reference's mesh
// description (vertex ids, face vertex triples, undirected edge ids).  Host-only C++17.
//
// This is what DijkstraMeshPlanner::initialize / CVPMeshPlanner::initialize would run once
// (dijkstra_mesh_planner.cpp:142-169, cvp_mesh_planner.cpp:148-186): the half-edge circulators the
// reference walks on every pop (getEdgesOfVertex :305-308, getFacesOfVertex cvp :775-776,
// getEdgeBetween cvp :380-390) become flat CSR arrays.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "mnav_eval.h"

namespace mnav {

struct HostTopology {
  uint32_t V = 0, F = 0, E = 0;
  // Dijkstra gather CSR: row v lists its incident undirected edges in ascending edge id
  std::vector<uint32_t> row_ptr;   // V+1
  std::vector<uint32_t> nbr_u;     // 2E other endpoint
  std::vector<uint32_t> nbr_e;     // 2E undirected edge id (index into edge_weights)
  // CVP corners: row v lists its incident faces in ascending face id
  std::vector<uint32_t> crn_ptr;   // V+1
  std::vector<uint32_t> crn_v1, crn_v2;          // 3F
  std::vector<uint32_t> crn_ea, crn_eb, crn_ec;  // 3F edge ids of sides a=(v2,v3) b=(v1,v3) c=(v1,v2)
  std::vector<uint32_t> crn_face;  // 3F
};

namespace detail {
inline uint64_t mix64(uint64_t x)
{
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}
inline uint64_t ekey(uint32_t a, uint32_t b)
{
  return a < b ? (uint64_t(a) << 32) | b : (uint64_t(b) << 32) | a;
}
// open-addressing (vertex pair) -> edge id
struct EdgeIndex {
  std::vector<uint64_t> keys;
  std::vector<uint32_t> vals;
  uint64_t mask = 0;
  void build(uint32_t E, const uint32_t* edge_vtx)
  {
    uint64_t cap = 16;
    while (cap < uint64_t(E) * 2 + 16) cap <<= 1;
    mask = cap - 1;
    keys.assign(cap, 0);
    vals.assign(cap, kNone);
    for (uint32_t e = 0; e < E; ++e) {
      const uint64_t k = ekey(edge_vtx[2 * size_t(e)], edge_vtx[2 * size_t(e) + 1]);
      uint64_t h = mix64(k) & mask;
      while (vals[h] != kNone && keys[h] != k) h = (h + 1) & mask;
      if (vals[h] == kNone) { keys[h] = k; vals[h] = e; }   // first id wins on duplicates
    }
  }
  uint32_t find(uint32_t a, uint32_t b) const
  {
    const uint64_t k = ekey(a, b);
    uint64_t h = mix64(k) & mask;
    while (vals[h] != kNone) {
      if (keys[h] == k) return vals[h];
      h = (h + 1) & mask;
    }
    return kNone;
  }
};
}  // namespace detail

// Throws std::invalid_argument on out-of-range ids or a face side that is not a listed edge.
inline HostTopology build_topology(uint32_t V, uint32_t F, uint32_t E, const uint32_t* face_vtx,
                                   const uint32_t* edge_vtx)
{
  HostTopology t;
  t.V = V; t.F = F; t.E = E;
  for (size_t i = 0; i < size_t(E) * 2; ++i)
    if (edge_vtx[i] >= V) throw std::invalid_argument("edge vertex id out of range");
  for (size_t i = 0; i < size_t(F) * 3; ++i)
    if (face_vtx[i] >= V) throw std::invalid_argument("face vertex id out of range");

  t.row_ptr.assign(size_t(V) + 1, 0);
  for (uint32_t e = 0; e < E; ++e) { t.row_ptr[edge_vtx[2 * size_t(e)] + 1]++; t.row_ptr[edge_vtx[2 * size_t(e) + 1] + 1]++; }
  for (uint32_t v = 0; v < V; ++v) t.row_ptr[v + 1] += t.row_ptr[v];
  t.nbr_u.resize(size_t(E) * 2);
  t.nbr_e.resize(size_t(E) * 2);
  {
    std::vector<uint32_t> fill(size_t(V) + 1, 0);
    for (uint32_t e = 0; e < E; ++e) {                       // ascending edge id per row
      const uint32_t a = edge_vtx[2 * size_t(e)], b = edge_vtx[2 * size_t(e) + 1];
      size_t ia = size_t(t.row_ptr[a]) + fill[a]++;
      t.nbr_u[ia] = b; t.nbr_e[ia] = e;
      size_t ib = size_t(t.row_ptr[b]) + fill[b]++;
      t.nbr_u[ib] = a; t.nbr_e[ib] = e;
    }
  }

  detail::EdgeIndex idx;
  idx.build(E, edge_vtx);
  t.crn_ptr.assign(size_t(V) + 1, 0);
  for (size_t i = 0; i < size_t(F) * 3; ++i) t.crn_ptr[face_vtx[i] + 1]++;
  for (uint32_t v = 0; v < V; ++v) t.crn_ptr[v + 1] += t.crn_ptr[v];
  const size_t C = size_t(F) * 3;
  t.crn_v1.resize(C); t.crn_v2.resize(C); t.crn_ea.resize(C); t.crn_eb.resize(C); t.crn_ec.resize(C);
  t.crn_face.resize(C);
  {
    std::vector<uint32_t> fill(size_t(V) + 1, 0);
    for (uint32_t f = 0; f < F; ++f) {                       // ascending face id per row
      const uint32_t fv[3] = { face_vtx[3 * size_t(f)], face_vtx[3 * size_t(f) + 1], face_vtx[3 * size_t(f) + 2] };
      uint32_t fe[3];                                        // fe[k] joins fv[k], fv[(k+1)%3]
      for (int k = 0; k < 3; ++k) {
        fe[k] = idx.find(fv[k], fv[(k + 1) % 3]);
        if (fe[k] == kNone) throw std::invalid_argument("face side is not a listed edge");
      }
      for (int k3 = 0; k3 < 3; ++k3) {
        const int k1 = (k3 + 1) % 3, k2 = (k3 + 2) % 3;
        const uint32_t v3 = fv[k3];
        const size_t i = size_t(t.crn_ptr[v3]) + fill[v3]++;
        t.crn_v1[i] = fv[k1]; t.crn_v2[i] = fv[k2];
        t.crn_ec[i] = fe[k1];   // (v1,v2)
        t.crn_eb[i] = fe[k3];   // (v3,v1)
        t.crn_ea[i] = fe[k2];   // (v2,v3)
        t.crn_face[i] = f;
      }
    }
  }
  return t;
}

// Host restatement of the device materialisation kernels (mnav_kernels.hip: k_build_nbr,
// k_build_corners); used by the CPU schedule model only.
inline void materialize_host(const HostTopology& t, const float* edge_weights, const float* vertex_costs,
                             const uint8_t* invalid, double cost_limit, std::vector<Nbr>& nbr,
                             std::vector<Corner>& crn, std::vector<uint8_t>& blocked)
{
  nbr.resize(t.nbr_u.size());
  for (uint32_t v = 0; v < t.V; ++v)
    for (uint32_t i = t.row_ptr[v]; i < t.row_ptr[v + 1]; ++i) {
      const uint32_t u = t.nbr_u[i];
      float w = edge_weights[t.nbr_e[i]];
      if ((invalid && invalid[v]) || (double)vertex_costs[u] > cost_limit) w = inf_f();
      nbr[i].u = u; nbr[i].w = w;
    }
  crn.resize(t.crn_v1.size());
  for (uint32_t v = 0; v < t.V; ++v)
    for (uint32_t i = t.crn_ptr[v]; i < t.crn_ptr[v + 1]; ++i) {
      Corner k;
      k.v1 = t.crn_v1[i]; k.v2 = t.crn_v2[i];
      k.a = edge_weights[t.crn_ea[i]]; k.b = edge_weights[t.crn_eb[i]]; k.c = edge_weights[t.crn_ec[i]];
      k.face = t.crn_face[i];
      if (invalid && (invalid[v] || invalid[k.v1] || invalid[k.v2])) k.v1 = kNone;
      crn[i] = k;
    }
  blocked.resize(t.V);
  for (uint32_t v = 0; v < t.V; ++v)
    blocked[v] = ((double)vertex_costs[v] >= cost_limit || (invalid && invalid[v])) ? 1 : 0;
}

}  // namespace mnav
