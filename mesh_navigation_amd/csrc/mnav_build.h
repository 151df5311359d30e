// mnav_build.h -- host-side construction of the device index arrays from the reference's mesh
// description (vertex ids, face vertex triples, undirected edge ids).  Host-only C++17.
//
// This is what DijkstraMeshPlanner::initialize / CVPMeshPlanner::initialize would run once
// (dijkstra_mesh_planner.cpp:142-169, cvp_mesh_planner.cpp:148-186): the half-edge circulators the
// reference walks on every pop (getEdgesOfVertex :305-308, getFacesOfVertex cvp :775-776,
// getEdgeBetween cvp :380-390) become flat CSR arrays.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <numeric>
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
  // CVP corners: row v lists its incident faces in ascending face id; crn_face carries the order flags
  std::vector<uint32_t> crn_ptr;   // V+1
  std::vector<uint32_t> crn_v1, crn_v2;          // 3F
  std::vector<uint32_t> crn_ea, crn_eb, crn_ec;  // 3F edge ids of sides a=(v2,v3) b=(v1,v3) c=(v1,v2)
  std::vector<uint32_t> crn_face;  // 3F  face id | kCornerFirst1/2
  // 3F  where the face is visited in the inflation wave's walk around each of its three vertices (inflation_layer.cpp
  // :423-427: for every neighbour of the popped vertex the face left of cur->nh, then the one left of nh->cur, so a
  // face comes up twice): 6 x 5 bits {v1: first, second; v2: first, second; the row's own vertex: first, second};
  // kWalkUnknown when a vertex has too many neighbours for 5 bits (the repulsive field is then not computed there)
  std::vector<uint32_t> crn_walk;
  // getFacesOfVertex rows the corner flags were derived from (the caller's or the half-edge replay): the order
  // MeshMap::searchNeighbourFaces expands faces in (mesh_map.cpp:1048-1049), read by the device back-tracking
  std::vector<uint32_t> vf_ptr, vf;   // V+1, 3F
};
constexpr uint32_t kWalkUnknown = 0xFFFFFFFFu;
inline uint32_t walk_pos(uint32_t walk, int slot) { return (walk >> (5 * slot)) & 31u; }   // slot 0..5

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

// ---------------------------------------------------------------------------------------------
// Face circulation: the order in which lvr2::PMPMesh::getFacesOfVertex (a pmp::SurfaceMesh circulator)
// lists the faces around a vertex -- counter-clockwise, starting at the vertex's stored outgoing
// half-edge, which depends on the order in which the faces were added.  The reference builds its mesh by
// adding the faces of the map file in index order (mesh_map.cpp:273), so the same incremental half-edge
// construction (the published pmp / OpenMesh add_face algorithm) is replayed here on the face list.
// A caller that holds the real lvr2 mesh can pass its own getFacesOfVertex rows instead
// (mnav_upload_face_circulation).  ok == false: the face list is not manifold in pmp's sense (complex
// vertex / edge); rows are then left in ascending face id.
// ---------------------------------------------------------------------------------------------
struct FaceCirculation {
  std::vector<uint32_t> ptr;   // V+1
  std::vector<uint32_t> faces; // 3F
  bool ok = false;
};

namespace detail {
class HalfEdgeBuild {
public:
  HalfEdgeBuild(uint32_t V, uint32_t F) : out_(V, kNone) { to_.reserve(size_t(F) * 3 + 8); }
  // add triangle f = (v[0], v[1], v[2]); false on a topological error
  bool add(uint32_t f, const uint32_t* v)
  {
    uint32_t h[3]; bool fresh[3], adjust[3] = { false, false, false };
    relink_.clear();
    for (int i = 0; i < 3; ++i) {
      if (!open_vertex(v[i])) return false;                      // complex vertex
      h[i] = find(v[i], v[(i + 1) % 3]);
      fresh[i] = h[i] == kNone;
      if (!fresh[i] && left_[h[i]] != kNone) return false;       // complex edge
    }
    for (int i = 0; i < 3; ++i) {                                // two existing boundary edges that are not yet
      const int n = (i + 1) % 3;                                 // consecutive: move the patch between them
      if (fresh[i] || fresh[n] || next_[h[i]] == h[n]) continue;
      uint32_t gap = h[n] ^ 1u;
      do { gap = next_[gap] ^ 1u; } while (left_[gap] != kNone || gap == h[i]);
      const uint32_t gap_next = next_[gap];
      if (gap_next == h[n]) return false;                        // patch re-linking failed
      relink_.push_back({ gap, next_[h[i]] });
      relink_.push_back({ prev_[h[n]], gap_next });
      relink_.push_back({ h[i], h[n] });
    }
    for (int i = 0; i < 3; ++i)
      if (fresh[i]) {                                            // pair 2e: v[i] -> v[i+1], 2e+1 back
        h[i] = uint32_t(to_.size());
        to_.push_back(v[(i + 1) % 3]); to_.push_back(v[i]);
        left_.push_back(kNone); left_.push_back(kNone);
        next_.push_back(kNone); next_.push_back(kNone);
        prev_.push_back(kNone); prev_.push_back(kNone);
      }
    for (int i = 0; i < 3; ++i) {
      const int n = (i + 1) % 3;
      const uint32_t c = v[n], in = h[i], on = h[n];             // corner vertex, incoming and outgoing inner half-edge
      if (fresh[i] || fresh[n]) {
        const uint32_t outer_in = on ^ 1u, outer_out = in ^ 1u;  // boundary half-edges through c after the insert
        if (!fresh[n]) {                                         // incoming side is new
          relink_.push_back({ prev_[on], outer_out });
          out_[c] = outer_out;
        } else if (!fresh[i]) {                                  // outgoing side is new
          relink_.push_back({ outer_in, next_[in] });
          out_[c] = next_[in];
        } else if (out_[c] == kNone) {                           // isolated vertex
          out_[c] = outer_out;
          relink_.push_back({ outer_in, outer_out });
        } else {                                                 // both new at a vertex that already has a fan
          const uint32_t b = out_[c];
          relink_.push_back({ prev_[b], outer_out });
          relink_.push_back({ outer_in, b });
        }
        relink_.push_back({ in, on });
      } else {
        adjust[n] = out_[c] == on;
      }
      left_[in] = f;
    }
    for (const auto& r : relink_) { next_[r.first] = r.second; prev_[r.second] = r.first; }
    for (int i = 0; i < 3; ++i)
      if (adjust[i]) {                                           // keep a boundary half-edge as the stored one
        uint32_t x = out_[v[i]];
        const uint32_t x0 = x;
        do {
          if (left_[x] == kNone) { out_[v[i]] = x; break; }
          x = next_[x ^ 1u];
        } while (x != x0);
      }
    return true;
  }
  // faces around v, counter-clockwise from the stored half-edge
  template <class Fn> void faces_around(uint32_t v, Fn&& fn) const
  {
    uint32_t x = out_[v];
    const uint32_t x0 = x;
    if (x == kNone) return;
    size_t guard = 0;
    do {
      if (left_[x] != kNone) fn(left_[x]);
      x = prev_[x] ^ 1u;
    } while (x != x0 && ++guard <= to_.size());
  }
private:
  bool open_vertex(uint32_t v) const { const uint32_t x = out_[v]; return x == kNone || left_[x] == kNone; }
  uint32_t find(uint32_t a, uint32_t b) const
  {
    uint32_t x = out_[a];
    const uint32_t x0 = x;
    if (x == kNone) return kNone;
    do {
      if (to_[x] == b) return x;
      x = next_[x ^ 1u];
    } while (x != x0);
    return kNone;
  }
  std::vector<uint32_t> out_, to_, left_, next_, prev_;
  std::vector<std::pair<uint32_t, uint32_t>> relink_;
};
}  // namespace detail

inline FaceCirculation build_face_circulation(uint32_t V, uint32_t F, const uint32_t* face_vtx)
{
  FaceCirculation c;
  c.ptr.assign(size_t(V) + 1, 0);
  for (size_t i = 0; i < size_t(F) * 3; ++i) c.ptr[face_vtx[i] + 1]++;
  for (uint32_t v = 0; v < V; ++v) c.ptr[v + 1] += c.ptr[v];
  c.faces.assign(size_t(F) * 3, kNone);
  detail::HalfEdgeBuild he(V, F);
  bool ok = true;
  for (uint32_t f = 0; f < F && ok; ++f) ok = he.add(f, face_vtx + 3 * size_t(f));
  if (ok) {
    for (uint32_t v = 0; v < V && ok; ++v) {
      uint32_t n = 0;
      const uint32_t cap = c.ptr[v + 1] - c.ptr[v];
      he.faces_around(v, [&](uint32_t f) { if (n < cap) c.faces[size_t(c.ptr[v]) + n] = f; ++n; });
      ok = n == cap;
    }
  }
  if (!ok) {                                                     // not manifold: ascending face id
    std::vector<uint32_t> fill(size_t(V) + 1, 0);
    for (uint32_t f = 0; f < F; ++f)
      for (int k = 0; k < 3; ++k) { const uint32_t v = face_vtx[3 * size_t(f) + k]; c.faces[size_t(c.ptr[v]) + fill[v]++] = f; }
  }
  c.ok = ok;
  return c;
}

// Throws std::invalid_argument on out-of-range ids or a face side that is not a listed edge.
// The inflation wave's walk around vertex tv (inflation_layer.cpp:423-427): for every outgoing halfedge h of the vertex
// circulator the face left of h, then the face left of its opposite.  `rows` = tv's faces in getFacesOfVertex order.
// The faces around tv form fans (runs of edge-adjacent faces); an interior vertex has one closed fan.  pmp's circulators
// start at halfedge(tv), which for a boundary vertex is the boundary halfedge that ENDS a fan (no face on its left):
// that fan's faces come last in `rows`, and its closing visit [-, last face] opens the walk.  Per fan, in row order:
// [F_0], [F_1, F_0], ..., [F_(k-1), F_(k-2)] and, except for that last fan, its own closing [F_(k-1)].
inline void inflation_walk(const uint32_t* rows, uint32_t m, const uint32_t* face_vtx, uint32_t tv, uint32_t degree, std::vector<uint32_t>& walk)
{
  walk.clear();
  if (m == 0) return;
  auto adjacent = [&](uint32_t f, uint32_t g) {                  // share an edge at tv
    for (int a = 0; a < 3; ++a) {
      const uint32_t x = face_vtx[3 * size_t(f) + a];
      if (x == tv) continue;
      for (int b = 0; b < 3; ++b) if (face_vtx[3 * size_t(g) + b] == x) return true;
    }
    return false;
  };
  bool closed = (m == degree) && (m == 1 || adjacent(rows[m - 1], rows[0]));
  for (uint32_t i = 0; closed && i + 1 < m; ++i) closed = adjacent(rows[i], rows[i + 1]);
  if (closed) {                                                   // interior vertex: pair i = [R_i, R_(i-1)]
    for (uint32_t i = 0; i < m; ++i) { walk.push_back(rows[i]); walk.push_back(rows[(i + m - 1) % m]); }
    return;
  }
  walk.push_back(rows[m - 1]);                                    // closing visit of the fan that halfedge(tv) ends
  uint32_t beg = 0;
  while (beg < m) {
    uint32_t end = beg + 1;
    while (end < m && adjacent(rows[end - 1], rows[end])) ++end;
    for (uint32_t i = beg; i < end; ++i) { walk.push_back(rows[i]); if (i > beg) walk.push_back(rows[i - 1]); }
    if (end < m) walk.push_back(rows[end - 1]);                   // this fan's closing visit (not the last fan's: it came first)
    beg = end;
  }
}

// `circ`: getFacesOfVertex rows (nullptr: derived from the face list, build_face_circulation).
inline HostTopology build_topology(uint32_t V, uint32_t F, uint32_t E, const uint32_t* face_vtx,
                                   const uint32_t* edge_vtx, const FaceCirculation* circ = nullptr)
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
  t.crn_face.resize(C); t.crn_walk.assign(C, 0u);
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
  // order flags (Corner, mnav_eval.h): walk every vertex t's faces in circulator order; of the (at most two)
  // faces that contain edge (t, v) the one met first gets the flag on v's corner for that face
  if (F >= (1u << 28)) throw std::invalid_argument("face ids must fit 28 bits");
  FaceCirculation own;
  if (!circ) { own = build_face_circulation(V, F, face_vtx); circ = &own; }
  {
    std::vector<uint32_t> seen, order, walk_tmp;                 // the v's already met around t; the walk order
    for (int pass = 0; pass < 2; ++pass) {                       // 0: getFacesOfVertex order (CVP), 1: inflation wave order
      const uint32_t flag1 = pass == 0 ? kCornerFirst1 : kCornerInfl1, flag2 = pass == 0 ? kCornerFirst2 : kCornerInfl2;
      for (uint32_t tv = 0; tv < V; ++tv) {
        const uint32_t r0 = circ->ptr[tv], m = circ->ptr[tv + 1] - r0;
        order.clear();
        for (uint32_t r = 0; r < m; ++r) order.push_back(circ->faces[r0 + r]);
        if (pass == 1) {                                            // first visits of the walk, in order
          inflation_walk(circ->faces.data() + r0, m, face_vtx, tv, t.row_ptr[tv + 1] - t.row_ptr[tv], walk_tmp);
          order.clear();
          for (uint32_t f : walk_tmp) if (std::find(order.begin(), order.end(), f) == order.end()) order.push_back(f);
        }
        seen.clear();
        for (uint32_t f : order) {
          if (f >= F) throw std::invalid_argument("face circulation row names an unknown face");
          for (int k = 0; k < 3; ++k) {
            const uint32_t v = face_vtx[3 * size_t(f) + k];
            if (v == tv) continue;
            const bool first = std::find(seen.begin(), seen.end(), v) == seen.end();
            if (!first) continue;
            seen.push_back(v);
            for (uint32_t i = t.crn_ptr[v]; i < t.crn_ptr[v + 1]; ++i)
              if ((t.crn_face[i] & kCornerFaceMask) == f) {
                t.crn_face[i] |= (t.crn_v1[i] == tv) ? flag1 : flag2;
                break;
              }
          }
        }
      }
    }
  }
  // walk positions (crn_walk): where each face comes up (at most twice) in inflation_walk(tv)
  {
    std::vector<uint32_t> walk;
    for (uint32_t tv = 0; tv < V; ++tv) {
      const uint32_t r0 = circ->ptr[tv], m = circ->ptr[tv + 1] - r0;
      if (m == 0) continue;
      inflation_walk(circ->faces.data() + r0, m, face_vtx, tv, t.row_ptr[tv + 1] - t.row_ptr[tv], walk);
      const bool fits = walk.size() <= 32;
      for (uint32_t r = 0; r < m; ++r) {
        const uint32_t f = circ->faces[r0 + r];
        uint32_t pa = 0, pb = 0; int seen = 0;
        for (uint32_t q = 0; q < walk.size(); ++q) if (walk[q] == f) { if (seen == 0) pa = q; else pb = q; ++seen; }
        if (seen == 1) pb = pa;                                     // boundary: the first / last face comes up once
        for (int k = 0; k < 3; ++k) {                               // the corner of every vertex of f gets tv's two positions
          const uint32_t u = face_vtx[3 * size_t(f) + k];
          for (uint32_t i = t.crn_ptr[u]; i < t.crn_ptr[u + 1]; ++i)
            if ((t.crn_face[i] & kCornerFaceMask) == f) {
              const int slot = (t.crn_v1[i] == tv) ? 0 : (t.crn_v2[i] == tv) ? 2 : 4;
              if (!fits || t.crn_walk[i] == kWalkUnknown) { t.crn_walk[i] = kWalkUnknown; break; }
              t.crn_walk[i] |= (pa << (5 * slot)) | (pb << (5 * (slot + 1)));
              break;
            }
        }
      }
    }
  }
  if (circ == &own) { t.vf_ptr = std::move(own.ptr); t.vf = std::move(own.faces); }
  else { t.vf_ptr = circ->ptr; t.vf = circ->faces; }
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

// ---------------------------------------------------------------------------------------------
// LDS tiles for the label-correcting SSSP engine (mnav_tiles.h: k_tile_round).  Vertices are sorted
// along a Morton curve of their quantised positions and cut into chunks of <= tile_size
// vertices.  A tile's local graph is PUSH oriented: local vertex x (owned vertices first, then
// the halo = neighbours owned by other tiles) lists the local vertices y it can relax, with the
// weight w(x -> y) taken from row y of the global gather CSR (entry `src`).  Owned vertices list
// all their neighbours (owned and halo), halo vertices only their owned neighbours.
// ---------------------------------------------------------------------------------------------
struct HostTiles {
  uint32_t ntiles = 0, tile_size = 0;
  uint32_t max_nv = 0, max_nh = 0, max_ne = 0;
  std::vector<uint32_t> vptr;        // ntiles+1 -> verts
  std::vector<uint32_t> verts;       // V   vertex ids grouped by tile
  std::vector<uint32_t> hptr;        // ntiles+1 -> halo_verts / halo_tile
  std::vector<uint32_t> halo_verts;  // vertex ids of the halo
  std::vector<uint32_t> halo_tile;   // owning tile of each halo vertex
  std::vector<uint32_t> eptr;        // ntiles+1 -> col / src (each tile padded to a multiple of 8 entries)
  std::vector<uint32_t> rptr;        // ntiles+1 -> rowptr (nv+nh+1 entries per tile, padded to a multiple of 8)
  std::vector<uint16_t> rowptr;      // local row pointers
  std::vector<uint16_t> col;         // tile-local target index: < nv owned, else nv + halo index
  std::vector<uint32_t> src;         // position of w(x -> y) in the global gather CSR; kNone on padding
  std::vector<uint32_t> vert_tile;   // V   tile of a vertex
};

namespace detail {
inline uint64_t spread3(uint64_t x)   // 21 bits -> every third bit
{
  x &= 0x1fffffULL;
  x = (x | x << 32) & 0x1f00000000ffffULL;
  x = (x | x << 16) & 0x1f0000ff0000ffULL;
  x = (x | x << 8) & 0x100f00f00f00f00fULL;
  x = (x | x << 4) & 0x10c30c30c30c30c3ULL;
  x = (x | x << 2) & 0x1249249249249249ULL;
  return x;
}
}  // namespace detail

inline HostTiles build_tiles(const HostTopology& t, const float* xyz, uint32_t tile_size)
{
  HostTiles T;
  const uint32_t V = t.V;
  T.tile_size = tile_size;
  // Morton order of quantised positions (isotropic scale: the largest extent maps to 2^21)
  float lo[3] = { 0, 0, 0 }, hi[3] = { 0, 0, 0 };
  for (uint32_t v = 0; v < V; ++v)
    for (int k = 0; k < 3; ++k) {
      const float x = xyz[3 * size_t(v) + k];
      if (v == 0 || x < lo[k]) lo[k] = x;
      if (v == 0 || x > hi[k]) hi[k] = x;
    }
  float ext = 0;
  for (int k = 0; k < 3; ++k) ext = std::max(ext, hi[k] - lo[k]);
  const double scale = ext > 0 ? 2097151.0 / ext : 0.0;
  std::vector<uint64_t> key(V);
  for (uint32_t v = 0; v < V; ++v) {
    uint64_t q[3];
    for (int k = 0; k < 3; ++k) {
      double f = (double(xyz[3 * size_t(v) + k]) - lo[k]) * scale;
      if (!(f >= 0)) f = 0;
      if (f > 2097151.0) f = 2097151.0;
      q[k] = (uint64_t)f;
    }
    key[v] = detail::spread3(q[0]) | (detail::spread3(q[1]) << 1) | (detail::spread3(q[2]) << 2);
  }
  T.verts.resize(V);
  std::iota(T.verts.begin(), T.verts.end(), 0u);
  std::sort(T.verts.begin(), T.verts.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b] || (key[a] == key[b] && a < b); });

  // chunk: <= tile_size vertices and < 60000 directed entries per tile (16-bit local row pointers)
  T.vert_tile.assign(V, 0);
  T.vptr.push_back(0);
  {
    uint32_t nv = 0; uint64_t ne = 0;
    for (uint32_t i = 0; i < V; ++i) {
      const uint32_t v = T.verts[i];
      const uint32_t deg = t.row_ptr[v + 1] - t.row_ptr[v];
      if (nv > 0 && (nv >= tile_size || ne + deg >= 60000)) { T.vptr.push_back(i); nv = 0; ne = 0; }
      T.vert_tile[v] = uint32_t(T.vptr.size() - 1);
      ++nv; ne += deg;
    }
    T.vptr.push_back(V);
  }
  T.ntiles = uint32_t(T.vptr.size() - 1);
  if (V == 0) { T.ntiles = 0; T.vptr.assign(1, 0); }
  // Inside a tile the vertices are listed in ascending id (the halo likewise, below): whatever walks a tile with consecutive
  // lanes -- staging, write-back, the finalize pass's V-sized outputs -- then touches runs of consecutive addresses of the
  // caller's vertex-order arrays (a scan row of the patch) instead of the Z-curve's pairs.  The local numbering is free: every
  // engine computes the order-independent fixed point.
  for (uint32_t tl = 0; tl < T.ntiles; ++tl) std::sort(T.verts.begin() + T.vptr[tl], T.verts.begin() + T.vptr[tl + 1]);

  // position of source x inside row y of the gather CSR
  auto pos_in_row = [&](uint32_t y, uint32_t x) -> uint32_t {
    for (uint32_t k = t.row_ptr[y]; k < t.row_ptr[y + 1]; ++k) if (t.nbr_u[k] == x) return k;
    return kNone;
  };
  std::vector<uint32_t> local(V, kNone);     // vertex -> local index inside the tile being built
  T.hptr.assign(1, 0); T.eptr.assign(1, 0); T.rptr.assign(1, 0);
  std::vector<uint32_t> halo;
  for (uint32_t tl = 0; tl < T.ntiles; ++tl) {
    const uint32_t b = T.vptr[tl], e = T.vptr[tl + 1], nv = e - b;
    for (uint32_t i = b; i < e; ++i) local[T.verts[i]] = i - b;
    halo.clear();
    for (uint32_t i = b; i < e; ++i) {
      const uint32_t v = T.verts[i];
      for (uint32_t k = t.row_ptr[v]; k < t.row_ptr[v + 1]; ++k) {
        const uint32_t u = t.nbr_u[k];
        if (local[u] == kNone) { local[u] = nv; halo.push_back(u); }   // (marked; numbered below)
      }
    }
    std::sort(halo.begin(), halo.end());
    for (uint32_t h = 0; h < uint32_t(halo.size()); ++h) local[halo[h]] = nv + h;
    const uint32_t nh = uint32_t(halo.size());
    if (nv + nh > 0xFFFFu) throw std::invalid_argument("tile too large for 16-bit local indices");
    uint32_t ne = 0;
    for (uint32_t x = 0; x < nv + nh; ++x) {
      const uint32_t gx = x < nv ? T.verts[b + x] : halo[x - nv];
      T.rowptr.push_back(uint16_t(ne));
      for (uint32_t k = t.row_ptr[gx]; k < t.row_ptr[gx + 1]; ++k) {
        const uint32_t gy = t.nbr_u[k];
        const uint32_t ly = local[gy];
        if (ly == kNone) continue;                 // halo vertex: neighbour outside the tile
        if (x >= nv && ly >= nv) continue;         // halo -> halo is not this tile's business
        T.col.push_back(uint16_t(ly));
        T.src.push_back(pos_in_row(gy, gx));
        ++ne;
      }
    }
    if (ne > 0xFFFFu) throw std::invalid_argument("tile too large for 16-bit local row pointers");
    T.rowptr.push_back(uint16_t(ne));
    while (T.rowptr.size() % 8) T.rowptr.push_back(uint16_t(ne));
    while (T.col.size() % 8) { T.col.push_back(0); T.src.push_back(kNone); }
    for (uint32_t h = 0; h < nh; ++h) { T.halo_verts.push_back(halo[h]); T.halo_tile.push_back(T.vert_tile[halo[h]]); local[halo[h]] = kNone; }
    for (uint32_t i = b; i < e; ++i) local[T.verts[i]] = kNone;
    T.hptr.push_back(uint32_t(T.halo_verts.size()));
    T.eptr.push_back(uint32_t(T.col.size()));
    T.rptr.push_back(uint32_t(T.rowptr.size()));
    T.max_nv = std::max(T.max_nv, nv);
    T.max_nh = std::max(T.max_nh, nh);
    T.max_ne = std::max(T.max_ne, uint32_t(T.col.size()) - T.eptr[tl]);   // padded entry count (staged as is)
  }
  return T;
}

}  // namespace mnav
