// mnav_tb_build.h -- host-side construction of the TILE-BATCH SSSP engine's arrays (k_tb_solve in mnav_tb.h).
// Host-only C++17, shared by the library (mnav.hip) and the CPU model of the schedule (oracle/tb_model.cpp).
//
// The engine runs many Dijkstra wavefronts (dijkstra_mesh_planner.cpp:287-348, one per plan) at once, TILE-major and
// plan-vectorised: the mesh is cut into compact tiles of <= T vertices; a 64-lane wave takes ONE tile and up to 64
// plans that have work on it (one plan per lane), keeps the plans' distances of the tile's vertices in LDS as
// [vertex][lane] and relaxes them to the tile-local fixed point with Gauss-Seidel sweeps.  Every lane executes the
// same edge sequence, so the tile's graph is SCALAR data: it is stored as flat record streams that the wave reads
// with scalar loads (constant address space), one stream per sweep direction.
//
// Per plan the distances live in per-tile SLICES: [T owned slots | ghost slots] -- the ghosts are copies of the
// neighbouring tiles' boundary vertices, written by their owners whenever they change (halo exchange), so an
// activation reads one contiguous slice and nothing else.  All plans' slices of one tile are adjacent in memory
// (D[tile][plan][slot]): the 64 lanes of a wave touch one few-MB region.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <stdexcept>
#include <vector>

#include "mnav_build.h"

namespace mnav {

struct TbRec { uint32_t a, b; };           // 8 bytes; meaning depends on the stream, see below
struct TbExp { uint32_t u, soff, sl, off; };   // boundary vertex `u` (LDS byte offset of its row) is a ghost of another tile:
                                               // its value goes to D[soff * NP + plan * sl + off]

// One 64-byte header per tile (read with scalar loads).
struct TbTile {
  uint32_t soff;          // start of the tile's region, in words PER PLAN (prefix sum of sl)
  uint32_t sl;            // slice length in words: T owned slots + ghosts, padded to a multiple of 4
  uint32_t nv;            // owned vertices (<= T; the slots nv..T-1 stay +inf)
  uint32_t nh;            // ghosts
  uint32_t sweep_off;     // first record of sweep order 0; order k starts at sweep_off + k * sweep_blocks * 8
  uint32_t sweep_blocks;  // 8-record blocks per order
  uint32_t pre_off, pre_blocks;     // 4-record blocks, ghost -> owned edges, grouped by ghost
  uint32_t post_off, post_blocks;   // 4-record blocks, owned -> ghost edges, grouped by ghost
  uint32_t exp_off, exp_n;          // TbExp records
  uint32_t v0;            // position of the first owned vertex in `verts`
  uint32_t pad[3];
};
static_assert(sizeof(TbTile) == 64, "TbTile is read as one 64-byte scalar load");

// Record streams.  A "row" is a local vertex; records carry the row's LDS byte offset (row * 256: 64 lanes x 4 B).
//   sweep block (8 records):  [0] = {target row offset, edge count}   [1..7] = {source row offset, weight bits}
//                             unused slots: {target row offset, +inf}
//   pre block (4 records):    [0] = {flags | j, edge count}           [1..3] = {target row offset, weight bits}
//                             ghost -> owned edges of ghost 4 * group + j
//   post block (4 records):   [0] = {flags | j | count << 8, owner tile}   [1..3] = {source row offset, weight bits}
//                             owned -> ghost edges of ghost 4 * group + j
// flags of pre / post headers:
constexpr uint32_t kTbGhostEnd = 1u << 4;   // last block of this ghost
constexpr uint32_t kTbGroupEnd = 1u << 5;   // last block of this group of 4 ghosts (the next block needs the next 16-byte load)
constexpr uint32_t kTbTileEnd = 1u << 6;    // last ghost owned by this neighbour tile (post: emit the wake-up)
constexpr uint32_t kTbInfBits = 0x7f800000u;
constexpr uint32_t kTbDirty = 0x80000000u;  // sign bit of an LDS value: lowered during this activation

struct HostTb {
  uint32_t T = 0, ntiles = 0, V = 0;
  uint64_t S = 0;                     // words per plan (sum of the slice lengths)
  uint32_t max_nh = 0, max_sl = 0;
  std::vector<TbTile> tiles;
  std::vector<TbRec> recs;
  std::vector<uint32_t> wsrc;         // per record: index into the gather CSR (Nbr) its weight comes from, kNone otherwise
  std::vector<TbExp> exps;
  std::vector<uint32_t> verts;        // tile order -> vertex id
  std::vector<uint32_t> vert_tile, vert_local;   // V
};

namespace detail {
// recursive coordinate bisection: compact, balanced tiles of <= T vertices in an order that keeps neighbours close
inline void tb_bisect(std::vector<uint32_t>& ids, size_t lo, size_t hi, const float* xyz, uint32_t T, std::vector<uint32_t>& cuts)
{
  const size_t n = hi - lo;
  if (n <= T) { cuts.push_back((uint32_t)hi); return; }
  float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
  for (size_t i = lo; i < hi; ++i)
    for (int k = 0; k < 3; ++k) { const float x = xyz[3 * (size_t)ids[i] + k]; mn[k] = std::min(mn[k], x); mx[k] = std::max(mx[k], x); }
  int ax = 0;
  for (int k = 1; k < 3; ++k) if (mx[k] - mn[k] > mx[ax] - mn[ax]) ax = k;
  // leaves as full as possible: the left half gets a multiple of T when that keeps the halves balanced
  const size_t leaves = (n + T - 1) / T;
  const size_t left = std::min(n - 1, std::max<size_t>(1, (leaves / 2) * (size_t)T));
  std::nth_element(ids.begin() + lo, ids.begin() + lo + left, ids.begin() + hi, [&](uint32_t a, uint32_t b) {
    const float xa = xyz[3 * (size_t)a + ax], xb = xyz[3 * (size_t)b + ax];
    return xa < xb || (xa == xb && a < b);
  });
  tb_bisect(ids, lo, lo + left, xyz, T, cuts);
  tb_bisect(ids, lo + left, hi, xyz, T, cuts);
}
}  // namespace detail

inline HostTb build_tb(const HostTopology& t, const float* xyz, uint32_t T)
{
  if (T == 0 || T > 255 || (T & 3)) throw std::invalid_argument("tile-batch engine: T must be a multiple of 4 below 256");
  HostTb H;
  const uint32_t V = t.V;
  H.T = T; H.V = V;
  H.verts.resize(V);
  std::iota(H.verts.begin(), H.verts.end(), 0u);
  std::vector<uint32_t> cuts;
  if (V) detail::tb_bisect(H.verts, 0, V, xyz, T, cuts);
  H.ntiles = (uint32_t)cuts.size();
  H.vert_tile.assign(V, 0); H.vert_local.assign(V, 0);
  H.tiles.assign(H.ntiles, TbTile{});
  {
    uint32_t b = 0;
    for (uint32_t tl = 0; tl < H.ntiles; ++tl) {
      // owned vertices in ascending id inside a tile (deterministic, and the order the stream builder assumes nothing about)
      std::sort(H.verts.begin() + b, H.verts.begin() + cuts[tl]);
      H.tiles[tl].v0 = b; H.tiles[tl].nv = cuts[tl] - b;
      for (uint32_t i = b; i < cuts[tl]; ++i) { H.vert_tile[H.verts[i]] = tl; H.vert_local[H.verts[i]] = i - b; }
      b = cuts[tl];
    }
  }
  // ghosts per tile, sorted by (owner tile, local index)
  std::vector<std::vector<uint32_t>> ghosts(H.ntiles);
  auto gkey = [&](uint32_t v) { return ((uint64_t)H.vert_tile[v] << 32) | H.vert_local[v]; };
  for (uint32_t tl = 0; tl < H.ntiles; ++tl) {
    auto& g = ghosts[tl];
    const TbTile& L = H.tiles[tl];
    for (uint32_t i = 0; i < L.nv; ++i) {
      const uint32_t v = H.verts[L.v0 + i];
      for (uint32_t k = t.row_ptr[v]; k < t.row_ptr[v + 1]; ++k) if (H.vert_tile[t.nbr_u[k]] != tl) g.push_back(t.nbr_u[k]);
    }
    std::sort(g.begin(), g.end(), [&](uint32_t a, uint32_t b) { return gkey(a) < gkey(b); });
    g.erase(std::unique(g.begin(), g.end()), g.end());
    TbTile& W = H.tiles[tl];
    W.nh = (uint32_t)g.size();
    W.sl = T + ((W.nh + 3u) & ~3u);
    W.soff = (uint32_t)H.S;
    if (H.S + W.sl > 0xFFFFFFFFull) throw std::invalid_argument("tile-batch engine: mesh too large for 32-bit slice offsets");
    H.S += W.sl;
    H.max_nh = std::max(H.max_nh, W.nh); H.max_sl = std::max(H.max_sl, W.sl);
  }
  auto ghost_slot = [&](uint32_t tl, uint32_t v) -> uint32_t {
    const auto& g = ghosts[tl];
    const auto it = std::lower_bound(g.begin(), g.end(), v, [&](uint32_t a, uint32_t b) { return gkey(a) < gkey(b); });
    return (uint32_t)(it - g.begin());
  };
  const uint32_t kRow = 256;   // bytes per LDS row
  auto push = [&](uint32_t a, uint32_t b, uint32_t src) { H.recs.push_back(TbRec{ a, b }); H.wsrc.push_back(src); };
  std::vector<uint16_t> order;
  std::vector<uint32_t> ts;
  for (uint32_t tl = 0; tl < H.ntiles; ++tl) {
    TbTile& W = H.tiles[tl];
    const auto& g = ghosts[tl];
    // --- sweep streams: four orders, sorted along the diagonals of the tile's two widest axes
    float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (uint32_t i = 0; i < W.nv; ++i)
      for (int k = 0; k < 3; ++k) { const float x = xyz[3 * (size_t)H.verts[W.v0 + i] + k]; mn[k] = std::min(mn[k], x); mx[k] = std::max(mx[k], x); }
    int a0 = 0, a1 = 1;
    {
      int idx[3] = { 0, 1, 2 };
      std::sort(idx, idx + 3, [&](int x, int y) { return (mx[x] - mn[x]) > (mx[y] - mn[y]) || ((mx[x] - mn[x]) == (mx[y] - mn[y]) && x < y); });
      a0 = std::min(idx[0], idx[1]); a1 = std::max(idx[0], idx[1]);
    }
    W.sweep_off = (uint32_t)H.recs.size();
    static const float dirs[4][2] = { { 1, 1 }, { -1, 1 }, { -1, -1 }, { 1, -1 } };
    for (int o = 0; o < 4; ++o) {
      order.resize(W.nv);
      for (uint32_t i = 0; i < W.nv; ++i) order[i] = (uint16_t)i;
      std::stable_sort(order.begin(), order.end(), [&](uint16_t x, uint16_t y) {
        const float* px = &xyz[3 * (size_t)H.verts[W.v0 + x]]; const float* py = &xyz[3 * (size_t)H.verts[W.v0 + y]];
        return dirs[o][0] * px[a0] + dirs[o][1] * px[a1] < dirs[o][0] * py[a0] + dirs[o][1] * py[a1];
      });
      uint32_t blocks = 0;
      for (uint32_t i = 0; i < W.nv; ++i) {
        const uint32_t y = order[i], v = H.verts[W.v0 + y];
        uint32_t in_block = 0, hdr = 0;
        for (uint32_t k = t.row_ptr[v]; k < t.row_ptr[v + 1]; ++k) {
          const uint32_t u = t.nbr_u[k];
          if (H.vert_tile[u] != tl) continue;
          if (in_block == 0) { hdr = (uint32_t)H.recs.size(); push(y * kRow, 0, kNone); ++blocks; }
          push(H.vert_local[u] * kRow, kTbInfBits, k);
          if (++in_block == 7) { H.recs[hdr].b = 7; in_block = 0; }
        }
        if (in_block) { H.recs[hdr].b = in_block; for (; in_block < 7; ++in_block) push(y * kRow, kTbInfBits, kNone); }
      }
      if (o == 0) W.sweep_blocks = blocks;
      else if (blocks != W.sweep_blocks) throw std::logic_error("tile-batch engine: sweep orders differ in size");
    }
    // --- pre stream: ghost -> owned
    auto emit_ghost_stream = [&](bool post) {
      const uint32_t off = (uint32_t)H.recs.size();
      uint32_t blocks = 0;
      for (uint32_t h = 0; h < W.nh; ++h) {
        const uint32_t gv = g[h];
        const uint32_t owner = H.vert_tile[gv];
        // edges between ghost gv and the owned vertices of this tile
        struct E { uint32_t row, k; };
        std::vector<E> es;
        if (!post) {                                 // gv -> owned y: weight in row y
          for (uint32_t k2 = t.row_ptr[gv]; k2 < t.row_ptr[gv + 1]; ++k2) {
            const uint32_t y = t.nbr_u[k2];
            if (H.vert_tile[y] != tl) continue;
            for (uint32_t k = t.row_ptr[y]; k < t.row_ptr[y + 1]; ++k) if (t.nbr_u[k] == gv) es.push_back(E{ H.vert_local[y], k });
          }
        } else {                                     // owned u -> gv: weight in row gv
          for (uint32_t k = t.row_ptr[gv]; k < t.row_ptr[gv + 1]; ++k) if (H.vert_tile[t.nbr_u[k]] == tl) es.push_back(E{ H.vert_local[t.nbr_u[k]], k });
        }
        const bool tile_end = (h + 1 == W.nh) || H.vert_tile[g[h + 1]] != owner;
        const bool group_end = (h + 1 == W.nh) || ((h & 3u) == 3u);
        size_t i = 0;
        do {
          const uint32_t n = (uint32_t)std::min<size_t>(3, es.size() - i);
          const bool last = i + n >= es.size();
          uint32_t fl = h & 3u;
          if (last) fl |= kTbGhostEnd | (group_end ? kTbGroupEnd : 0u) | (tile_end ? kTbTileEnd : 0u);
          if (!post) push(fl, n, kNone); else push(fl | (n << 8), owner, kNone);
          for (uint32_t q = 0; q < 3; ++q) {
            if (q < n) push(es[i + q].row * kRow, kTbInfBits, es[i + q].k);
            else push(0, kTbInfBits, kNone);
          }
          ++blocks; i += n;
        } while (i < es.size());
      }
      if (!post) { W.pre_off = off; W.pre_blocks = blocks; } else { W.post_off = off; W.post_blocks = blocks; }
    };
    emit_ghost_stream(false);
    emit_ghost_stream(true);
    // --- exports: owned boundary vertices -> ghost slots of the neighbouring tiles
    W.exp_off = (uint32_t)H.exps.size();
    for (uint32_t i = 0; i < W.nv; ++i) {
      const uint32_t v = H.verts[W.v0 + i];
      ts.clear();
      for (uint32_t k = t.row_ptr[v]; k < t.row_ptr[v + 1]; ++k) if (H.vert_tile[t.nbr_u[k]] != tl) ts.push_back(H.vert_tile[t.nbr_u[k]]);
      std::sort(ts.begin(), ts.end()); ts.erase(std::unique(ts.begin(), ts.end()), ts.end());
      for (uint32_t t2 : ts) H.exps.push_back(TbExp{ i * kRow, H.tiles[t2].soff, H.tiles[t2].sl, T + ghost_slot(t2, v) });
    }
    W.exp_n = (uint32_t)H.exps.size() - W.exp_off;
    while (H.exps.size() % 4) H.exps.push_back(TbExp{ 0, 0, 0, 0 });   // groups of 4 records = one 64-byte scalar load
    while (H.recs.size() % 8) push(0, kTbInfBits, kNone);
  }
  for (int k = 0; k < 16; ++k) push(0, kTbInfBits, kNone);            // tail slack for the block prefetch
  for (int k = 0; k < 4; ++k) H.exps.push_back(TbExp{ 0, 0, 0, 0 });
  return H;
}

}  // namespace mnav
