// mnav_tb_build.h -- host-side construction of the TILE-BATCH SSSP engine's arrays (k_tb_solve_q in mnav_tb.h).
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

struct TbExp { uint32_t u, soff, sl, off; };   // boundary vertex `u` (LDS byte offset of its row) is a ghost of another tile:
                                               // its value goes to D[soff * NP + plan * sl + off]

// One 64-byte header per tile (read with one scalar load).
struct TbTile {
  uint32_t soff;          // start of the tile's region, in words PER PLAN (prefix sum of sl)
  uint32_t sl;            // slice length in words: T owned slots + ghosts, padded to a multiple of 4
  uint32_t nv;            // owned vertices (<= T; the slots nv..T-1 stay +inf)
  uint32_t nh;            // ghosts
  uint32_t sweep_off;     // first chunk of sweep order 0; order k starts at chunk sweep_off + k * sweep_chunks
  uint32_t sweep_chunks;  // chunks per order
  uint32_t pre_off, pre_chunks;     // ghost -> owned edges
  uint32_t post_off, post_chunks;   // owned -> ghost edges
  uint32_t exp_off, exp_n;          // TbExp records
  uint32_t v0;            // position of the first owned vertex in `verts`
  uint32_t goff;          // position of the tile's first ghost in `ghost_gid` (ghosts in slice order)
  uint32_t ovf_off, ovf_n;          // TbFinOvf records: sources beyond the kTbFinSlots a vertex has in the finalize tables
};
static_assert(sizeof(TbTile) == 64, "TbTile is read as one 64-byte scalar load");

// Streams.  The graph of a tile is stored as CHUNKS of 64 dwords (256 bytes).  A wave reads a chunk with ONE coalesced
// vector load (lane l holds dword l; the next chunks are prefetched behind the vector-memory counter), parks it in a
// 256-byte LDS staging buffer, and every lane then reads the dwords it needs back with uniform-address ds_read_b128:
// an LDS broadcast delivers a wave-uniform value to all lanes at a quarter of the cost of a v_readlane -> SGPR hop
// (measured: ~30 cycles per readlane'd value against ~8 per broadcast dword), and scalar loads would share the LDS's
// wait counter and miss the scalar cache on every block (each order's stream is read once per sweep).
// A chunk holds 4 BLOCKS of 16 dwords.  A "row" is a local vertex; offsets are the row's LDS byte offset (row * 256:
// 64 lanes x 4 B).
//   sweep block:  d0 = target offset, d1..d7 = source offsets, d8..d14 = weight bits of the sources (unused slots:
//                 source = target, weight = +inf), d15 = 0.  A block that FOLLOWS another one inside its chunk (blocks 1..3)
//                 never has that block's target, and its slot 1 (d1, d8) is reserved for the edge from that block's target row
//                 (weight +inf when there is no such edge); no other slot names that row.  Read in order the stream is a plain
//                 Gauss-Seidel sweep; the kernel uses the rule to have the LDS reads of block j + 1 in flight while block j is
//                 computed: the one value block j + 1 needs from block j comes from a register (k_tb_solve_q).
//   pre block:    d0 = flags | j | n << 8 (n = edges, 0 = empty block), d1..d5 = target offsets, d6..d10 = weight bits:
//                 the ghost -> owned edges of ghost 4 * group + j;  in block 0 of a chunk d12 = the chunk's ghost group
//                 (which 16-byte quad of the slice's ghost part), d13 = the NEXT chunk's group (prefetch)
//   post block:   as pre with source offsets (owned -> ghost edges), d11 = owner tile of the ghost
constexpr uint32_t kTbChunk = 64;           // dwords per chunk
constexpr uint32_t kTbBlock = 16;           // dwords per block
constexpr uint32_t kTbBlocksPerChunk = 4;
constexpr uint32_t kTbGhostEdges = 5;       // edges per pre / post block
constexpr uint32_t kTbGhostEnd = 1u << 4;   // last block of this ghost (post: compare the candidate with the ghost value)
constexpr uint32_t kTbTileEnd = 1u << 6;    // last ghost owned by this neighbour tile (post: emit the wake-up)
constexpr uint32_t kTbOrderShift = 12;      // pre blocks, 2 bits: the sweep order that runs with a wave entering through this ghost
constexpr uint32_t kTbInfBits = 0x7f800000u;
constexpr uint32_t kTbDirty = 0x80000000u;  // sign bit of an LDS value: lowered during this activation

// Finalize tables (k_tb_finalize: potential with the reference's cut-off semantics, predecessors and vector map straight from a
// tile's slice): per tile and owned vertex y the SOURCES of y -- the pull form of the tile's graph --, kTbFinSlots per vertex,
// slot-major: entry (tile, k, y) at (tile * kTbFinSlots + k) * T + y.  fin_src: the source's index in the tile's SLICE (owned:
// its local index, ghost: T + its ghost slot; kTbFinNone: no source), fin_wsrc: the entry of the gather CSR its weight
// w(source -> y) comes from.  The few vertices of higher valence continue in TbFinOvf records.
constexpr uint32_t kTbFinSlots = 8;
constexpr uint16_t kTbFinNone = 0xFFFFu;
struct TbFinOvf { uint32_t y, src, wsrc, pad; };

// Export RUN of the V layout: up to four boundary rows of a tile whose ghost copies in ONE neighbour's slices are adjacent slots
// (the ghosts of a slice are sorted by owner tile and the owner's local index): rows = four window rows, one per byte (a run of
// fewer repeats its first row), values to D[soff * NP + plan * sl + off .. + n - 1] -- one 16-byte store instead of four scattered
// 4-byte ones (the kernel is bound by the NUMBER of store requests there: 64 per instruction, one per lane)
struct TbvExp { uint32_t rows, soff, sl, off, n, pad[3]; };
static_assert(sizeof(TbvExp) == 32, "read as two 16-byte loads");

struct HostTb {
  std::vector<uint16_t> fin_src; std::vector<uint32_t> fin_wsrc; std::vector<TbFinOvf> fin_ovf;
  std::vector<uint32_t> ghost_gid;    // vertex ids of the tiles' ghosts, tile after tile in slice order (TbTile.goff)
  uint32_t T = 0, ntiles = 0, V = 0;
  uint64_t S = 0;                     // words per plan (sum of the slice lengths)
  uint32_t max_nh = 0, max_sl = 0;
  std::vector<TbTile> tiles;
  std::vector<uint32_t> stream;       // chunks
  std::vector<uint32_t> wsrc;         // per stream dword: index into the gather CSR (Nbr) its weight comes from, kNone otherwise
  // the sweep streams once more in the V layout (k_tbv_solve, mnav_tbv.h: the distances of a tile in registers, rows addressed
  // through the VGPR index mode): chunks, their weight sources, and per tile {first chunk, chunks per order}
  std::vector<uint32_t> vstream, vwsrc;
  std::vector<uint32_t> vtile;        // kTbvTileWords x ntiles: {pre first chunk, pre chunks, sweep first chunk, chunks per order, post first chunk, post chunks, first group, groups}
  std::vector<TbvExp> vexps;          // export runs, tile after tile (vtile words 8, 9)
  std::vector<uint32_t> vgroups;      // owner tiles of a tile's ghosts, one entry per run of ghosts with the same owner (= per wake-up), in slice order
  std::vector<TbExp> exps;
  std::vector<uint32_t> verts;        // tile order -> vertex id
  std::vector<uint32_t> vert_tile, vert_local;   // V
};

namespace detail {
// Recursive coordinate bisection of a vertex set into `k` compact tiles of <= T vertices, in an order that keeps
// neighbours close.  The cut is placed in the LARGEST GAP of the coordinate near the proportional split position: on a
// scanned / gridded surface a cut through the middle of a row of vertices would deal that row to the two sides at
// random (a zigzag boundary that wavefronts cross back and forth: twice the tile activations on the 1M terrain);
// cuts between rows give straight boundaries.  The leaf budget leaves ~10 % slack so that such a cut usually exists
// within the positions that keep both sides within their budgets.
inline void tb_bisect(std::vector<uint32_t>& ids, size_t lo, size_t hi, size_t k, const float* xyz, uint32_t T, std::vector<uint32_t>& cuts)
{
  const size_t n = hi - lo;
  if (k <= 1) { cuts.push_back((uint32_t)hi); return; }
  float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
  for (size_t i = lo; i < hi; ++i)
    for (int q = 0; q < 3; ++q) { const float x = xyz[3 * (size_t)ids[i] + q]; mn[q] = std::min(mn[q], x); mx[q] = std::max(mx[q], x); }
  int ax = 0;
  for (int q = 1; q < 3; ++q) if (mx[q] - mn[q] > mx[ax] - mn[ax]) ax = q;
  auto less = [&](uint32_t a, uint32_t b) {
    const float xa = xyz[3 * (size_t)a + ax], xb = xyz[3 * (size_t)b + ax];
    return xa < xb || (xa == xb && a < b);
  };
  int ax2 = (ax + 1) % 3;                                            // second widest axis
  { const int o = (ax + 2) % 3; if (mx[o] - mn[o] > mx[ax2] - mn[ax2]) ax2 = o; }
  const size_t kl = k / 2, kr = k - kl;
  const size_t target = (size_t)((double)n * (double)kl / (double)k);
  // positions c (vertices on the left) that keep both sides within their budgets
  const size_t f_lo = std::max<size_t>(n > kr * (size_t)T ? n - kr * (size_t)T : 1, 1), f_hi = std::min(kl * (size_t)T, n - 1);
  // sorted window of order statistics around the target: about two rows of a square patch of n vertices on each side
  const size_t w = (size_t)(2.0 * std::sqrt((double)n)) + 8;
  const size_t a = target > w ? target - w : 0, b = std::min(target + w, n - 1);   // positions a..b are sorted
  if (a > 0) std::nth_element(ids.begin() + lo, ids.begin() + lo + a, ids.begin() + hi, less);
  if (b + 1 < n) std::nth_element(ids.begin() + lo + a, ids.begin() + lo + b, ids.begin() + hi, less);
  std::sort(ids.begin() + lo + a, ids.begin() + lo + b + 1, less);
  auto gap_at = [&](size_t c) { return xyz[3 * (size_t)ids[lo + c] + ax] - xyz[3 * (size_t)ids[lo + c - 1] + ax]; };   // between c-1 and c
  float maxgap = 0.f;
  for (size_t c = a + 1; c <= b; ++c) maxgap = std::max(maxgap, gap_at(c));
  const size_t c_lo = std::min(std::max(f_lo, a + 1), b), c_hi = std::max(std::min(f_hi, b), c_lo);
  size_t cut = std::min(std::max(target, c_lo), c_hi); float best = -1.f;
  for (size_t c = c_lo; c <= c_hi; ++c) {
    const float gap = gap_at(c);
    const size_t dc = c > target ? c - target : target - c, db = cut > target ? cut - target : target - cut;
    if (gap > best * 1.0001f + 1e-12f || (gap >= best && dc < db)) { best = gap; cut = c; }
  }
  if (best < 0.5f * maxgap) {
    // no cut between two rows fits the budgets: cut through a row, but deal its vertices to the two sides in the order of the
    // second axis (an L-shaped, contiguous boundary) instead of by their jitter along the first one
    cut = std::min(std::max(target, c_lo), c_hi);
    size_t rl = a, rr = b + 1;
    for (size_t c = cut; c > a; --c) if (gap_at(c) >= 0.5f * maxgap) { rl = c; break; }
    for (size_t c = cut + 1; c <= b; ++c) if (gap_at(c) >= 0.5f * maxgap) { rr = c; break; }
    std::sort(ids.begin() + lo + rl, ids.begin() + lo + rr, [&](uint32_t p, uint32_t q) {
      const float xp = xyz[3 * (size_t)p + ax2], xq = xyz[3 * (size_t)q + ax2];
      return xp < xq || (xp == xq && p < q);
    });
  }
  tb_bisect(ids, lo, lo + cut, kl, xyz, T, cuts);
  tb_bisect(ids, lo + cut, hi, kr, xyz, T, cuts);
}
}  // namespace detail

// SWEEP chunks are stored TRANSPOSED: dword q of block j of a chunk sits at chunk dword 4 * q + j.  Lane l of a 16-lane
// quarter loads the 16 bytes l of its quarter's chunk -- one VGPR per block, holding dword l of that block in lane l -- and
// the sweep consumes a descriptor dword as a DPP operand (row_newbcast:q broadcasts lane q of every 16-lane row to the row):
// the block descriptors never pass through the LDS (they were two thirds of the kernel's LDS traffic: four broadcast
// ds_read_b128 per block against eight ds_read_b32 of data).  The ghost and export streams keep the plain layout (block j =
// dwords 16 j .. 16 j + 15; they are parked in the LDS staging area).
// n / d for n < 2^31 as one multiply-high and a shift (k_tb_scan divides a listed pair by the blocks per tile once per row):
// s = floor(log2 d), M = ceil(2^(32+s) / d) -- M d - 2^(32+s) < d, so the quotient is exact while n d <= 2^(32+s), which n < 2^31
// guarantees.  Powers of two take s - 1 (M = 2^31); d == 1 has no such pair and is tested for.
inline void tb_div_magic(uint32_t d, uint32_t* magic, uint32_t* shift)
{
  if (d <= 1u) { *magic = 0u; *shift = 0u; return; }
  uint32_t s = 31u - (uint32_t)__builtin_clz(d);
  if ((d & (d - 1u)) == 0u) s -= 1u;
  *magic = (uint32_t)(((1ull << (32u + s)) + d - 1ull) / d); *shift = s;
}
inline uint32_t tb_div_by_magic(uint32_t n, uint32_t d, uint32_t magic, uint32_t shift)
{
  return d <= 1u ? n : ((uint32_t)(((uint64_t)n * magic) >> 32) >> shift);
}

inline uint32_t tb_sweep_index(uint32_t j, uint32_t q) { return 4u * q + j; }

// V layout of the streams (k_tbv_solve, mnav_tbv.h).  The kernel keeps a tile's distances in a window of VGPRs -- rows 0 ..
// kTbvGhostRows - 1: the ghost slots of the slice, rows kTbvGhostRows + r: owned vertex r -- and addresses a row through the VGPR
// index mode: row indices are stored as ready-made values of M0 (index in bits 7:0, the operands it applies to in bits 15:12).
// One block format for the three phases:
//   d0 = (mode | target) | (0x2000 | source0) << 16      d1 = (0x2000 | source1) | (0x2000 | source2) << 16
//   d2 = (0x2000 | source3) | (0x2000 | source4) << 16   d3 = (0x2000 | source5) | flags << 16
//   d8 .. d13 = the six weights (unused slot: source = target, weight +inf); chunks of 4 blocks, transposed like the Q sweep chunks
//   sweep (four orders): target = an owned row (mode 0xA000: read and written), sources = its neighbours inside the tile
//   pre  : target = an owned row with neighbours outside (0xA000), sources = those ghosts; flags bits 0-1 = the sweep order that
//          runs with a wave entering at this row
//   post : "target" = a ghost (0x2000: read only), sources = its owned neighbours; flags bit 0 = last block of this ghost,
//          bit 1 = last ghost of its owner tile (one entry of vgroups per such run: the tile that is woken)
// A vertex with more than six sources continues in a second block of the same target.  No forwarding rule, no separator blocks.
constexpr uint32_t kTbvSources = 6;
constexpr uint32_t kTbvGhostRows = 64;      // a tile with more ghosts than this keeps the engine on k_tb_solve_q
constexpr uint32_t kTbvSrc = 0x2000u, kTbvDst = 0xA000u;
constexpr uint32_t kTbvTileWords = 12;     // + {first export run, export runs, 0, 0}
constexpr uint32_t kTbvGhostEnd = 1u, kTbvTileEnd = 2u;
inline void tbv_set_source(uint32_t* blk, uint32_t k, uint32_t row)   // blk: the 16 dwords of a block, plain order
{
  static const uint32_t dw[6] = { 0, 1, 1, 2, 2, 3 }, sh[6] = { 16, 0, 16, 0, 16, 0 };
  blk[dw[k]] = (blk[dw[k]] & ~(0xFFFFu << sh[k])) | ((kTbvSrc | row) << sh[k]);
}
inline void tbv_init_block(uint32_t* blk, uint32_t row, uint32_t mode = kTbvDst)
{
  for (uint32_t q = 0; q < kTbBlock; ++q) blk[q] = 0u;
  blk[0] = mode | row;
  for (uint32_t k = 0; k < kTbvSources; ++k) { tbv_set_source(blk, k, row); blk[8 + k] = 0x7f800000u; }
}
inline uint32_t tbv_target(const uint32_t* blk) { return blk[0] & 0xFFu; }
inline uint32_t tbv_flags(const uint32_t* blk) { return blk[3] >> 16; }
inline uint32_t tbv_source(const uint32_t* blk, uint32_t k)
{
  static const uint32_t dw[6] = { 0, 1, 1, 2, 2, 3 }, sh[6] = { 16, 0, 16, 0, 16, 0 };
  return (blk[dw[k]] >> sh[k]) & 0xFFu;
}
// blocks in plain order (16 dwords each) -> chunks appended to `stream` (padded with do-nothing blocks on row `pad_row`, transposed);
// returns the first chunk, *chunks = their number (padded up to `min_chunks`)
inline uint32_t tbv_append(std::vector<uint32_t>& stream, std::vector<uint32_t>& wsrc, std::vector<uint32_t>& blocks, std::vector<uint32_t>& bw,
                           uint32_t pad_row, uint32_t pad_mode, uint32_t min_chunks, uint32_t* chunks)
{
  const uint32_t first = (uint32_t)(stream.size() / kTbChunk);
  uint32_t n = (uint32_t)((blocks.size() / kTbBlock + kTbBlocksPerChunk - 1) / kTbBlocksPerChunk);
  n = std::max(n, min_chunks);
  while (blocks.size() < (size_t)n * kTbChunk) { const size_t at = blocks.size(); blocks.resize(at + kTbBlock); bw.resize(at + kTbBlock, 0xFFFFFFFFu); tbv_init_block(&blocks[at], pad_row, pad_mode); }
  for (uint32_t c = 0; c < n; ++c) {
    const size_t at = stream.size();
    stream.resize(at + kTbChunk); wsrc.resize(at + kTbChunk);
    for (uint32_t j = 0; j < kTbBlocksPerChunk; ++j)
      for (uint32_t q = 0; q < kTbBlock; ++q) {
        stream[at + tb_sweep_index(j, q)] = blocks[(size_t)c * kTbChunk + kTbBlock * j + q];
        wsrc[at + tb_sweep_index(j, q)] = bw[(size_t)c * kTbChunk + kTbBlock * j + q];
      }
  }
  *chunks = n;
  return first;
}

inline HostTb build_tb(const HostTopology& t, const float* xyz, uint32_t T)
{
  if (T == 0 || T > 255 || (T & 3)) throw std::invalid_argument("tile-batch engine: T must be a multiple of 4 below 256");
  HostTb H;
  const uint32_t V = t.V;
  H.T = T; H.V = V;
  H.verts.resize(V);
  std::iota(H.verts.begin(), H.verts.end(), 0u);
  std::vector<uint32_t> cuts;
  if (V) detail::tb_bisect(H.verts, 0, V, std::max<size_t>(1, (size_t)std::ceil((double)V / (0.9 * T))), xyz, T, cuts);
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
  // finalize tables
  H.fin_src.assign((size_t)H.ntiles * kTbFinSlots * T, kTbFinNone);
  H.fin_wsrc.assign((size_t)H.ntiles * kTbFinSlots * T, kNone);
  for (uint32_t tl = 0; tl < H.ntiles; ++tl) {
    TbTile& W = H.tiles[tl];
    W.goff = (uint32_t)H.ghost_gid.size();
    H.ghost_gid.insert(H.ghost_gid.end(), ghosts[tl].begin(), ghosts[tl].end());
    W.ovf_off = (uint32_t)H.fin_ovf.size();
    for (uint32_t y = 0; y < W.nv; ++y) {
      const uint32_t v = H.verts[W.v0 + y];
      uint32_t n = 0;
      for (uint32_t k = t.row_ptr[v]; k < t.row_ptr[v + 1]; ++k, ++n) {
        const uint32_t u = t.nbr_u[k];
        const uint32_t si = (H.vert_tile[u] == tl) ? H.vert_local[u] : T + ghost_slot(tl, u);
        if (si >= kTbFinNone) throw std::invalid_argument("tile-batch engine: a slice is too long for the finalize tables");
        if (n < kTbFinSlots) { const size_t at = ((size_t)tl * kTbFinSlots + n) * T + y; H.fin_src[at] = (uint16_t)si; H.fin_wsrc[at] = k; }
        else H.fin_ovf.push_back(TbFinOvf{ y, si, k, 0u });
      }
    }
    W.ovf_n = (uint32_t)H.fin_ovf.size() - W.ovf_off;
  }
  const uint32_t kRow = 256;   // bytes per LDS row
  // chunk writer: blocks are appended to the open chunk, a closed chunk is padded with empty blocks
  uint32_t in_chunk = 0;
  // Every sweep block rewrites its target row (with the bits it read when nothing improved); two ADJACENT blocks of a chunk
  // never have the same target, so that a kernel may have the reads of block j+1 in flight before block j's result is
  // written (tried and dropped: +24 % sweeps).  `last_target` = target row of the previous block of the open chunk.
  uint32_t last_target = kNone;
  auto open_block = [&]() -> size_t {                                // returns the dword index of the new block
    if (in_chunk == kTbBlocksPerChunk) { in_chunk = 0; last_target = kNone; }
    const size_t at = H.stream.size();
    H.stream.resize(at + kTbBlock, 0u); H.wsrc.resize(at + kTbBlock, kNone);
    ++in_chunk;
    return at;
  };
  // the row the block opened NEXT follows inside its chunk (kNone: it will be block 0 of a chunk)
  auto chunk_prev = [&]() -> uint32_t { return (in_chunk == 0 || in_chunk == kTbBlocksPerChunk) ? kNone : last_target; };
  auto init_sweep_block = [&](size_t at, uint32_t row, uint32_t prev) {   // every slot "target, +inf" until filled; slot 1 of a follower: "prev, +inf"
    for (int q = 0; q <= 7; ++q) H.stream[at + q] = row * kRow;
    if (prev != kNone) H.stream[at + 1] = prev * kRow;
    for (int q = 8; q <= 14; ++q) H.stream[at + q] = kTbInfBits;
  };
  auto noop_sweep_block = [&]() {                                    // all weights +inf, on a row the previous block did not target
    const uint32_t row = (last_target == 0u) ? 1u : 0u;
    const uint32_t prev = chunk_prev();
    const size_t at = open_block();
    init_sweep_block(at, row, prev);
    last_target = row;
  };
  auto close_chunk = [&](bool sweep) {                               // pad the open chunk
    if (in_chunk == 0) return;
    while (in_chunk < kTbBlocksPerChunk) { if (sweep) noop_sweep_block(); else open_block(); }
    in_chunk = 0; last_target = kNone;
  };
  std::vector<uint16_t> order;
  std::vector<std::pair<uint32_t, uint32_t>> srcs;
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
    W.sweep_off = (uint32_t)(H.stream.size() / kTbChunk);
    static const float dirs[4][2] = { { 1, 1 }, { -1, 1 }, { -1, -1 }, { 1, -1 } };
    uint32_t order_chunks[4] = { 0, 0, 0, 0 };
    std::vector<uint32_t> vs[4], vw[4];                               // the V layout of the four orders (plain block order until transposed below)
    for (int o = 0; o < 4; ++o) {
      order.resize(W.nv);
      for (uint32_t i = 0; i < W.nv; ++i) order[i] = (uint16_t)i;
      std::stable_sort(order.begin(), order.end(), [&](uint16_t x, uint16_t y) {
        const float* px = &xyz[3 * (size_t)H.verts[W.v0 + x]]; const float* py = &xyz[3 * (size_t)H.verts[W.v0 + y]];
        return dirs[o][0] * px[a0] + dirs[o][1] * px[a1] < dirs[o][0] * py[a0] + dirs[o][1] * py[a1];
      });
      for (uint32_t i = 0; i < W.nv; ++i) {                          // V layout: blocks of up to six sources, nothing else
        const uint32_t y = order[i], v = H.verts[W.v0 + y];
        uint32_t k6 = kTbvSources; size_t at = 0;
        for (uint32_t k = t.row_ptr[v]; k < t.row_ptr[v + 1]; ++k) {
          if (H.vert_tile[t.nbr_u[k]] != tl) continue;
          if (k6 == kTbvSources) { at = vs[o].size(); vs[o].resize(at + kTbBlock); vw[o].resize(at + kTbBlock, kNone); tbv_init_block(&vs[o][at], kTbvGhostRows + y); k6 = 0; }
          tbv_set_source(&vs[o][at], k6, kTbvGhostRows + H.vert_local[t.nbr_u[k]]); vw[o][at + 8 + k6] = k; ++k6;
        }
      }
      const size_t first = H.stream.size() / kTbChunk;
      for (uint32_t i = 0; i < W.nv; ++i) {
        const uint32_t y = order[i], v = H.verts[W.v0 + y];
        srcs.clear();                                                // the row's sources inside the tile: {local row, entry of the gather CSR}
        for (uint32_t k = t.row_ptr[v]; k < t.row_ptr[v + 1]; ++k)
          if (H.vert_tile[t.nbr_u[k]] == tl) srcs.push_back({ H.vert_local[t.nbr_u[k]], k });
        while (!srcs.empty()) {
          if (chunk_prev() == y) noop_sweep_block();                 // continuation of a high-valence vertex: never the same target twice in a row
          const uint32_t prev = chunk_prev();
          const size_t at = open_block();
          init_sweep_block(at, y, prev);
          last_target = y;
          uint32_t slot = 1;
          if (prev != kNone) {                                       // a follower: slot 1 belongs to the edge from the row the block before it writes
            for (size_t e = 0; e < srcs.size(); ++e)
              if (srcs[e].first == prev) { H.wsrc[at + 8] = srcs[e].second; srcs.erase(srcs.begin() + e); break; }
            slot = 2;
          }
          while (slot <= 7 && !srcs.empty()) {
            H.stream[at + slot] = srcs.front().first * kRow; H.wsrc[at + 7 + slot] = srcs.front().second;
            srcs.erase(srcs.begin()); ++slot;
          }
        }
      }
      close_chunk(true);
      order_chunks[o] = (uint32_t)(H.stream.size() / kTbChunk - first);
    }
    // the four orders hold the same edges but may differ by a few separator blocks: all are padded to the longest one
    // (chunks of do-nothing blocks), so that order k starts at sweep_off + k * sweep_chunks
    W.sweep_chunks = std::max(std::max(order_chunks[0], order_chunks[1]), std::max(order_chunks[2], order_chunks[3]));
    {
      std::vector<uint32_t> st2, ws2;
      size_t src = (size_t)W.sweep_off * kTbChunk;
      for (int o = 0; o < 4; ++o) {
        const size_t len = (size_t)order_chunks[o] * kTbChunk;
        st2.insert(st2.end(), H.stream.begin() + src, H.stream.begin() + src + len);
        ws2.insert(ws2.end(), H.wsrc.begin() + src, H.wsrc.begin() + src + len);
        src += len;
        for (uint32_t c = order_chunks[o]; c < W.sweep_chunks; ++c)
          for (uint32_t j = 0; j < kTbBlocksPerChunk; ++j) {
            const size_t at = st2.size();
            st2.resize(at + kTbBlock, 0u); ws2.resize(at + kTbBlock, kNone);
            for (int q = 0; q <= 7; ++q) st2[at + q] = (j & 1u) * kRow;
            if (j) st2[at + 1] = ((j - 1u) & 1u) * kRow;             // (a follower's slot 1 names the row the block before it writes)
            for (int q = 8; q <= 14; ++q) st2[at + q] = kTbInfBits;
          }
      }
      H.stream.resize((size_t)W.sweep_off * kTbChunk); H.wsrc.resize((size_t)W.sweep_off * kTbChunk);
      H.stream.insert(H.stream.end(), st2.begin(), st2.end()); H.wsrc.insert(H.wsrc.end(), ws2.begin(), ws2.end());
    }
    // what the software-pipelined sweep of the kernel relies on (see the stream description above), checked on every chunk
    for (size_t c = (size_t)W.sweep_off; c < (size_t)W.sweep_off + 4u * W.sweep_chunks; ++c)
      for (uint32_t j = 1; j < kTbBlocksPerChunk; ++j) {
        const uint32_t* K = &H.stream[c * kTbChunk + kTbBlock * j];
        const uint32_t* Ws = &H.wsrc[c * kTbChunk + kTbBlock * j];
        const uint32_t prev = K[-(int)kTbBlock];
        bool ok = K[0] != prev && K[1] == prev;
        for (int q = 2; q <= 7; ++q) ok = ok && !(K[q] == prev && Ws[7 + q] != kNone);
        if (!ok) throw std::logic_error("tile-batch streams: a follower block breaks the forwarding rules");
      }
    // transpose the tile's sweep chunks (tb_sweep_index): built block by block above, read lane by lane by the kernel
    for (size_t c = (size_t)W.sweep_off; c < (size_t)W.sweep_off + 4u * W.sweep_chunks; ++c) {
      uint32_t tmp[kTbChunk], tmpw[kTbChunk];
      std::copy(H.stream.begin() + c * kTbChunk, H.stream.begin() + (c + 1) * kTbChunk, tmp);
      std::copy(H.wsrc.begin() + c * kTbChunk, H.wsrc.begin() + (c + 1) * kTbChunk, tmpw);
      for (uint32_t j = 0; j < kTbBlocksPerChunk; ++j)
        for (uint32_t q = 0; q < kTbBlock; ++q) {
          H.stream[c * kTbChunk + tb_sweep_index(j, q)] = tmp[kTbBlock * j + q];
          H.wsrc[c * kTbChunk + tb_sweep_index(j, q)] = tmpw[kTbBlock * j + q];
        }
    }
    // the V layout of the same four orders: padded to the longest one with do-nothing blocks (the first owned row from itself,
    // weights +inf); order k at chunk (sweep first chunk) + k * (chunks per order)
    uint32_t v_sweep_off = 0, v_sweep_chunks = 0;
    {
      size_t vblocks = 0;
      for (int o = 0; o < 4; ++o) vblocks = std::max(vblocks, vs[o].size() / kTbBlock);
      const uint32_t vchunks = (uint32_t)((vblocks + kTbBlocksPerChunk - 1) / kTbBlocksPerChunk);
      for (int o = 0; o < 4; ++o) {
        uint32_t n = 0;
        const uint32_t first = tbv_append(H.vstream, H.vwsrc, vs[o], vw[o], kTbvGhostRows, kTbvDst, vchunks, &n);
        if (o == 0) { v_sweep_off = first; v_sweep_chunks = n; }
      }
    }
    // which of the four sweep orders runs WITH a wave that enters through ghost gv: the one whose direction has the largest
    // component along (tile centroid - ghost position).  The solve starts its sweeps with the order most lanes ask for.
    float cen[2] = { 0.f, 0.f };
    for (uint32_t i = 0; i < W.nv; ++i) { const float* q = &xyz[3 * (size_t)H.verts[W.v0 + i]]; cen[0] += q[a0]; cen[1] += q[a1]; }
    if (W.nv) { cen[0] /= (float)W.nv; cen[1] /= (float)W.nv; }
    auto ghost_order = [&](uint32_t gv) -> uint32_t {
      const float d0 = cen[0] - xyz[3 * (size_t)gv + a0], d1 = cen[1] - xyz[3 * (size_t)gv + a1];
      uint32_t best = 0; float bv = -INFINITY;
      for (uint32_t o = 0; o < 4; ++o) { const float sc = dirs[o][0] * d0 + dirs[o][1] * d1; if (sc > bv) { bv = sc; best = o; } }
      return best;
    };
    // --- pre / post streams: one chunk per group of 4 ghosts (more when a group needs more than 5 blocks)
    auto emit_ghost_stream = [&](bool post) {
      const size_t first = H.stream.size() / kTbChunk;
      std::vector<size_t> chunk_at;                                  // dword index of every chunk of this stream
      std::vector<uint32_t> chunk_group;
      auto begin_chunk = [&](uint32_t group) {
        close_chunk(false);
        chunk_at.push_back(H.stream.size()); chunk_group.push_back(group);
      };
      for (uint32_t h = 0; h < W.nh; ++h) {
        const uint32_t gv = g[h];
        const uint32_t owner = H.vert_tile[gv];
        if ((h & 3u) == 0u) begin_chunk(h / 4);
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
        size_t i = 0;
        do {
          if (in_chunk == kTbBlocksPerChunk) begin_chunk(h / 4);     // continuation chunk of the same group
          const uint32_t n = (uint32_t)std::min<size_t>(kTbGhostEdges, es.size() - i);
          const bool last = i + n >= es.size();
          const size_t at = open_block();
          uint32_t fl = (h & 3u) | (n << 8);
          if (last) fl |= kTbGhostEnd | (tile_end ? kTbTileEnd : 0u);
          if (!post) fl |= ghost_order(gv) << kTbOrderShift;
          H.stream[at] = fl;
          for (uint32_t q = 0; q < n; ++q) {
            H.stream[at + 1 + q] = es[i + q].row * kRow;
            H.stream[at + 6 + q] = kTbInfBits; H.wsrc[at + 6 + q] = es[i + q].k;
          }
          if (post) H.stream[at + 11] = owner;
          i += n;
        } while (i < es.size());
      }
      close_chunk(false);
      for (size_t c = 0; c < chunk_at.size(); ++c) {
        H.stream[chunk_at[c] + 12] = chunk_group[c];
        H.stream[chunk_at[c] + 13] = c + 1 < chunk_at.size() ? chunk_group[c + 1] : 0u;
      }
      const uint32_t chunks = (uint32_t)(H.stream.size() / kTbChunk - first);
      if (!post) { W.pre_off = (uint32_t)first; W.pre_chunks = chunks; } else { W.post_off = (uint32_t)first; W.post_chunks = chunks; }
    };
    emit_ghost_stream(false);
    emit_ghost_stream(true);
    // --- the ghost phases in the V layout (pull form, the block format of the sweeps)
    {
      std::vector<uint32_t> pb, pw, qb, qw;
      for (uint32_t y = 0; y < W.nv; ++y) {                            // pre: owned rows with neighbours outside the tile <- those ghosts (weight in row y)
        const uint32_t v = H.verts[W.v0 + y];
        uint32_t k6 = kTbvSources; size_t at = 0;
        for (uint32_t k = t.row_ptr[v]; k < t.row_ptr[v + 1]; ++k) {
          const uint32_t u = t.nbr_u[k];
          if (H.vert_tile[u] == tl) continue;
          if (k6 == kTbvSources) {
            at = pb.size(); pb.resize(at + kTbBlock); pw.resize(at + kTbBlock, kNone);
            tbv_init_block(&pb[at], kTbvGhostRows + y); pb[at + 3] |= ghost_order(v) << 16; k6 = 0;
          }
          tbv_set_source(&pb[at], k6, ghost_slot(tl, u)); pw[at + 8 + k6] = k; ++k6;
        }
      }
      const uint32_t grp_off = (uint32_t)H.vgroups.size();
      for (uint32_t h = 0; h < W.nh; ++h) {                            // post: ghost h <- its owned neighbours (weight in row gv)
        const uint32_t gv = g[h], owner = H.vert_tile[gv];
        const bool tile_end = (h + 1 == W.nh) || H.vert_tile[g[h + 1]] != owner;
        uint32_t k6 = kTbvSources; size_t at = 0;
        for (uint32_t k = t.row_ptr[gv]; k < t.row_ptr[gv + 1]; ++k) {
          const uint32_t u = t.nbr_u[k];
          if (H.vert_tile[u] != tl) continue;
          if (k6 == kTbvSources) { at = qb.size(); qb.resize(at + kTbBlock); qw.resize(at + kTbBlock, kNone); tbv_init_block(&qb[at], h, kTbvSrc); k6 = 0; }
          tbv_set_source(&qb[at], k6, kTbvGhostRows + H.vert_local[u]); qw[at + 8 + k6] = k; ++k6;
        }
        qb[at + 3] |= (kTbvGhostEnd | (tile_end ? kTbvTileEnd : 0u)) << 16;   // (a ghost has at least one neighbour in the tile: `at` is its last block)
        if (tile_end) H.vgroups.push_back(owner);
      }
      uint32_t pre_chunks = 0, post_chunks = 0;
      const uint32_t pre_off = tbv_append(H.vstream, H.vwsrc, pb, pw, kTbvGhostRows, kTbvDst, 0u, &pre_chunks);
      const uint32_t post_off = tbv_append(H.vstream, H.vwsrc, qb, qw, 0u, kTbvSrc, 0u, &post_chunks);
      const uint32_t words[kTbvTileWords] = { pre_off, pre_chunks, v_sweep_off, v_sweep_chunks, post_off, post_chunks, grp_off, (uint32_t)H.vgroups.size() - grp_off, 0u, 0u, 0u, 0u };
      H.vtile.insert(H.vtile.end(), words, words + kTbvTileWords);
    }
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
    {
      // the same records as runs of adjacent ghost slots of one neighbour (V layout)
      std::vector<TbExp> rs(H.exps.begin() + W.exp_off, H.exps.end());
      std::sort(rs.begin(), rs.end(), [](const TbExp& a, const TbExp& b) { return a.soff != b.soff ? a.soff < b.soff : a.off < b.off; });
      const uint32_t first = (uint32_t)H.vexps.size();
      for (size_t i = 0; i < rs.size();) {
        TbvExp r{}; r.soff = rs[i].soff; r.sl = rs[i].sl; r.off = rs[i].off;
        const uint32_t row0 = kTbvGhostRows + rs[i].u / kRow;
        r.rows = row0 * 0x01010101u;
        uint32_t n = 1;
        while (n < 4 && i + n < rs.size() && rs[i + n].soff == r.soff && rs[i + n].off == r.off + n) { r.rows = (r.rows & ~(0xFFu << (8 * n))) | ((kTbvGhostRows + rs[i + n].u / kRow) << (8 * n)); ++n; }
        r.n = n;
        H.vexps.push_back(r);
        i += n;
      }
      H.vtile[(size_t)kTbvTileWords * tl + 8] = first; H.vtile[(size_t)kTbvTileWords * tl + 9] = (uint32_t)H.vexps.size() - first;
    }
    while (H.exps.size() % 4) H.exps.push_back(TbExp{ 0, 0, 0, 0 });   // groups of 4 records = one 64-byte scalar load
  }
  H.stream.resize(H.stream.size() + 4 * kTbChunk, 0u); H.wsrc.resize(H.wsrc.size() + 4 * kTbChunk, kNone);   // tail slack for the chunk prefetch
  H.vstream.resize(H.vstream.size() + 4 * kTbChunk, 0u); H.vwsrc.resize(H.vwsrc.size() + 4 * kTbChunk, kNone);
  for (int k = 0; k < 64; ++k) H.vexps.push_back(TbvExp{});          // tail slack: a wave fetches 64 runs at a time
  for (int k = 0; k < 48; ++k) H.exps.push_back(TbExp{ 0, 0, 0, 0 });   // tail slack: the quarter-wave solve reads the records in chunks of 16, two chunks ahead
  if (H.stream.size() / kTbChunk > 0xFFFFFFF0ull) throw std::invalid_argument("tile-batch engine: stream too large");
  return H;
}

}  // namespace mnav
