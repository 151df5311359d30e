// mnav_tiles.h -- Dijkstra on LDS tiles: the tile plan record, the LDS image and the tile solve (stage_tile_graph, tile_sweeps)
// and the kernels built on them -- k_tile_round (one launch per round), k_plan_async
// (mnav_async.h, included here) --, k_tile_init, k_tile_weights.  Included by mnav.hip inside its anonymous namespace; not a
// stand-alone header.
#pragma once

// ---------------------------------------------------------------------------------------------
// Tiled label-correcting SSSP (Dijkstra planner).  The final float32 distances of the reference
// loop (dijkstra :287-348) are the unique fixed point of d[v] = min_u fl(d[u] + w(u,v)) over
// expanding sources u, so any relaxation schedule reproduces them bit for bit.  Schedule used
// here: the mesh is cut into Morton tiles of <= tile_size vertices (mnav_build.h).  One
// workgroup stages a tile's push graph, its distances and its halo in LDS and relaxes to the
// local fixed point with LDS-only sweeps over an active queue (ds_min on the float bits, 8 lanes
// per active vertex), restricted to sources below the current band threshold `thr`; it then
// writes the owned distances back and leaves a wake-up value (the smallest source value still
// to be propagated) for itself and for the tiles owning halo vertices it undercut.  One launch
// = one round over all tiles whose wake-up value lies below thr; thr advances by `band` when
// nothing below it is left.  Sources above the running bound dist[target] + offset are never
// relaxed (goal_dist cut-off, dijkstra :293-300); the exact cut-off semantics and the
// predecessors are then produced by one gather pass (k_dij_finalize).
// ---------------------------------------------------------------------------------------------
struct TCtl { int32_t it; uint32_t done; float thr; float thr_prev; uint32_t acts; uint32_t sweeps; uint32_t pad[2]; };
struct TCnt { uint32_t minpend; uint32_t acts; uint32_t sweeps; uint32_t pad; };

struct TilePlan {
  uint32_t V, ntiles;
  // (GPtr, mnav_eval.h: element access typed global in the device pass -- a flat_load through a generic pointer may alias the LDS
  // as far as the compiler knows, so it was kept in order with every LDS store around it: the seven words of a tile header were
  // seven round trips)
  GPtr<const uint32_t> vptr, verts, hptr, halo_verts, halo_tile, eptr, rptr;
  GPtr<const uint16_t> rowptr;
  GPtr<const uint16_t> col;     // per local edge: local target            } split arrays: 6 B per edge in LDS;
  GPtr<const float> tw;         // per local edge: push weight (+inf on padding) } tiles padded to 8 entries
  GPtr<float> dist;
  GPtr<uint32_t> pend[2];       // per-tile wake-up value (float bits), ping-pong by round parity
  GPtr<float> tlast;            // per-tile threshold of its last solve
  GPtr<TCtl> ctl;               // [2]
  GPtr<TCnt> cnt;               // [3]
  uint32_t seed, target;
  double offset;
  float band;
  uint32_t max_rounds;
  uint32_t max_nv, max_nh, max_ne;
  GPtr<const uint32_t> cancel;  // device word set by mnav_cancel (polled by k_plan_async), may be null
  uint32_t t_lo, t_hi;     // tiles this process owns (sharded single plan, mnav_shard_*); t_hi == 0: all tiles
  GPtr<uint32_t> parked;        // asynchronous engine (mnav_async.h): the two parked lists of the plan, 2 x kParkedLists x ntiles tile ids
  GPtr<const uint8_t> owned;    // partitioned mesh (mnav_shard_setup_partition): 1 = this process owns the vertex, 0 = halo copy whose
                           // neighbourhood is incomplete here (its value arrives through the exchange); null: every vertex is owned
  uint32_t pend1_is_state; // asynchronous engine: pend[1] holds the tiles' state words (0 at the end), not wake-up values -- the
                           // finalize pass then tells an untouched tile by tlast and pend[0] alone
};

constexpr int kTileBlock = 256;
constexpr int kTileVpt = 8;              // owned vertices per thread: tile_size <= 2048
constexpr int kTileTodo = 64;            // tiles one workgroup takes per round
constexpr uint32_t kInfBits = 0x7f800000u;

__host__ __device__ inline uint32_t pad_to(uint32_t x, uint32_t a) { return (x + a - 1) / a * a; }

// LDS image of one tile (dynamic shared memory), shared by k_tile_round and k_plan_async
struct TileLds {
  float* lw;        // push weights                      4 B x ne
  uint16_t* lcol;   // push targets (local ids)          2 B x ne
  uint32_t* ldu;    // distances as float bits           owned, then halo
  uint32_t* lh0;    // halo distances as loaded
  uint32_t* mask;   // 3 rotating "already queued" bitmasks over the owned vertices
  uint16_t* lrow;   // local row pointers
  uint16_t *q0, *q1;
  uint32_t mw;      // words per bitmask
};
__host__ __device__ inline uint32_t tile_mask_words(uint32_t max_nv) { return (max_nv + 31) / 32; }
__host__ __device__ inline size_t tile_lds_bytes(uint32_t max_nv, uint32_t max_nh, uint32_t max_ne)
{
  const uint32_t nl = max_nv + max_nh;
  return 6 * (size_t)pad_to(max_ne, 8) + 4 * (size_t)pad_to(nl, 4) + 4 * (size_t)pad_to(max_nh, 4) +
         4 * (size_t)pad_to(3 * tile_mask_words(max_nv), 4) + 2 * (size_t)pad_to(nl + 1, 8) + 2 * 2 * (size_t)pad_to(nl, 8);
}
__device__ __forceinline__ TileLds tile_lds_layout(char* smem, uint32_t max_nv, uint32_t max_nh, uint32_t max_ne)
{
  const uint32_t nl = max_nv + max_nh;
  TileLds L;
  L.lw = reinterpret_cast<float*>(smem);
  L.lcol = reinterpret_cast<uint16_t*>(L.lw + pad_to(max_ne, 8));
  L.ldu = reinterpret_cast<uint32_t*>(L.lcol + pad_to(max_ne, 8));
  L.lh0 = L.ldu + pad_to(nl, 4);
  L.mask = L.lh0 + pad_to(max_nh, 4);
  L.mw = tile_mask_words(max_nv);
  L.lrow = reinterpret_cast<uint16_t*>(L.mask + pad_to(3 * L.mw, 4));
  L.q0 = L.lrow + pad_to(nl + 1, 8);
  L.q1 = L.q0 + pad_to(nl, 8);
  return L;
}

#ifdef MNAV_TILE_TIMING
__device__ unsigned long long g_tile_timing[4096 * 8];
__device__ unsigned int g_tile_timing_n;
#define TT_STAMP(k) do { if (tid == 0) tt[k] = clock64(); } while (0)
#else
#define TT_STAMP(k) do { } while (0)
#endif

// Stage a tile's push graph (weights, targets, row pointers) into LDS: every 16-byte global load of
// a thread is issued before the first LDS store, so one memory round trip covers the whole copy
// for tiles of up to 4 x 256 x 4 edges (larger tiles loop).  The arrays have a 64-byte tail slack.
__device__ __forceinline__ void stage_tile_graph(const TilePlan& P, const TileLds& L, uint32_t e0, uint32_t ne, uint32_t r0,
                                                 uint32_t nl, int tid)
{
  MNAV_GLOBAL const u32x4* sw = (MNAV_GLOBAL const u32x4*)(P.tw + e0);          // e0, ne multiples of 8 entries
  MNAV_GLOBAL const u32x4* sc = (MNAV_GLOBAL const u32x4*)(P.col + e0);
  MNAV_GLOBAL const u32x4* sr = (MNAV_GLOBAL const u32x4*)(P.rowptr + r0);      // r0 multiple of 8 entries
  u32x4* dw = reinterpret_cast<u32x4*>(L.lw);
  u32x4* dc = reinterpret_cast<u32x4*>(L.lcol);
  u32x4* dr = reinterpret_cast<u32x4*>(L.lrow);
  const uint32_t nw16 = ne / 4, nc16 = ne / 8, nr16 = (nl + 1 + 7) / 8;
  u32x4 aw[4], ac[2], ar;
#pragma unroll
  for (int u = 0; u < 4; ++u) { const uint32_t i = tid + u * kTileBlock; aw[u] = sw[i < nw16 ? i : 0]; }
#pragma unroll
  for (int u = 0; u < 2; ++u) { const uint32_t i = tid + u * kTileBlock; ac[u] = sc[i < nc16 ? i : 0]; }
  ar = sr[(uint32_t)tid < nr16 ? tid : 0];
#pragma unroll
  for (int u = 0; u < 4; ++u) { const uint32_t i = tid + u * kTileBlock; if (i < nw16) dw[i] = aw[u]; }
#pragma unroll
  for (int u = 0; u < 2; ++u) { const uint32_t i = tid + u * kTileBlock; if (i < nc16) dc[i] = ac[u]; }
  if ((uint32_t)tid < nr16) dr[tid] = ar;
  for (uint32_t i = tid + 4 * kTileBlock; i < nw16; i += kTileBlock) dw[i] = sw[i];
  for (uint32_t i = tid + 2 * kTileBlock; i < nc16; i += kTileBlock) dc[i] = sc[i];
  for (uint32_t i = tid + kTileBlock; i < nr16; i += kTileBlock) dr[i] = sr[i];
  for (uint32_t i = tid; i < 3 * L.mw; i += kTileBlock) L.mask[i] = 0u;
}

// Sweeps over the active queue of the staged tile until it runs dry: 8 lanes per active vertex push
// along its row with ds_min on the float bits (the float add is dijkstra :331); improved owned
// targets enter the next queue once (ds_or on a rotating bitmask).  s_nq[3] rotates like the masks:
// [sweep % 3] is consumed, [(sweep+1) % 3] filled, [(sweep+2) % 3] cleared.  Returns the sweep count.
__device__ __forceinline__ uint32_t tile_sweeps(const TileLds& L, uint32_t nv, float thr, float bound, uint32_t* s_nq, int tid)
{
  const int sub = tid & (kGroup - 1);
  uint32_t sweep = 0;
  for (;;) {
    const uint32_t nq = s_nq[sweep % 3];
    if (nq == 0) break;
    if (tid == 0) s_nq[(sweep + 2) % 3] = 0;
    if ((uint32_t)tid < L.mw) L.mask[((sweep + 2) % 3) * L.mw + tid] = 0u;
    for (uint32_t i = tid + kTileBlock; i < L.mw; i += kTileBlock) L.mask[((sweep + 2) % 3) * L.mw + i] = 0u;
    const uint16_t* qa = (sweep & 1) ? L.q1 : L.q0;
    uint16_t* qb = (sweep & 1) ? L.q0 : L.q1;
    uint32_t* nqb = &s_nq[(sweep + 1) % 3];
    uint32_t* mk = L.mask + ((sweep + 1) % 3) * L.mw;
    for (uint32_t idx = (uint32_t)tid >> 3; idx < nq; idx += kTileBlock / kGroup) {
      const uint32_t x = qa[idx];
      const uint32_t dib = L.ldu[x];
      const uint32_t eb = L.lrow[x], ee = L.lrow[x + 1];          // issued together with ldu[x]
      const float di = u2f(dib);
      if (!(di < thr) || !(di <= bound)) continue;
      for (uint32_t e = eb + sub; e < ee; e += kGroup) {
        const uint32_t c = L.lcol[e];
        const uint32_t ndb = f2u(di + L.lw[e]);
        const uint32_t old = atomicMin(&L.ldu[c], ndb);
        if (ndb < old && c < nv) {
          const uint32_t bit = 1u << (c & 31);
          if (!(atomicOr(&mk[c >> 5], bit) & bit)) qb[atomicAdd(nqb, 1u)] = (uint16_t)c;
        }
      }
    }
    ++sweep;
    __syncthreads();
  }
  return sweep;
}

__global__ __launch_bounds__(kTileBlock) void k_tile_round(const TilePlan* __restrict__ plans, int j)
{
#ifdef MNAV_TILE_TIMING
  unsigned long long tt[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#endif
  const TilePlan& P = plans[blockIdx.y];
  const int tid = threadIdx.x;
  TT_STAMP(0);
  __shared__ TCtl s_ctl;
  __shared__ float s_bound;
  __shared__ uint32_t s_todo[kTileTodo];
  __shared__ uint32_t s_ntodo;
  __shared__ uint32_t s_hdr[kTileTodo][8];
  __shared__ uint32_t s_nq[3];
  __shared__ uint32_t s_left;
  if (tid == 0) {
    const TCtl prev = P.ctl[(j + 1) & 1];
    const TCnt cprev = P.cnt[(j + 2) % 3];
    TCtl cur = prev;
    cur.it = prev.it + 1;
    cur.acts = prev.acts + cprev.acts;
    cur.sweeps = prev.sweeps + cprev.sweeps;
    const float m = u2f(cprev.minpend);
    const float dt = P.dist[P.target];
    const float bound = (float)((double)dt + fmax(P.offset, 0.0)); // >= the final goal_dist (dijkstra :296); a negative offset is applied in the finalize pass (goal_cut)
    cur.done = (prev.done || !(m < inf_f()) || m > bound || (uint32_t)cur.it >= P.max_rounds) ? 1u : 0u;
    if (!cur.done && !(m < prev.thr)) {                             // band exhausted: advance
      cur.thr_prev = prev.thr;
      float thr = m + P.band;
      if (!(thr > m)) thr = next_up(m);
      cur.thr = thr;
    }
    s_ctl = cur; s_bound = bound; s_ntodo = 0;
    if (blockIdx.x == 0) {
      P.ctl[j & 1] = cur;
      TCnt z; z.minpend = kInfBits; z.acts = 0; z.sweeps = 0; z.pad = 0;
      P.cnt[(j + 1) % 3] = z;
    }
  }
  __syncthreads();
  const TCtl cur = s_ctl;
  if (cur.done) return;
  TT_STAMP(1);
  const float bound = s_bound, thr = cur.thr;
  TCnt* cnt = &P.cnt[j % 3];
  MNAV_GLOBAL uint32_t* pc = as_global(P.pend[cur.it & 1]);
  MNAV_GLOBAL uint32_t* pn = as_global(P.pend[(cur.it + 1) & 1]);
  MNAV_GLOBAL const uint32_t* g_vptr = as_global(P.vptr);
  MNAV_GLOBAL const uint32_t* g_hptr = as_global(P.hptr);
  MNAV_GLOBAL const uint32_t* g_eptr = as_global(P.eptr);
  MNAV_GLOBAL const uint32_t* g_rptr = as_global(P.rptr);
  MNAV_GLOBAL const uint32_t* g_verts = as_global(P.verts);
  MNAV_GLOBAL const uint32_t* g_halo_verts = as_global(P.halo_verts);
  MNAV_GLOBAL const uint32_t* g_halo_tile = as_global(P.halo_tile);
  MNAV_GLOBAL float* g_dist = as_global(P.dist);
  MNAV_GLOBAL float* g_tlast = as_global(P.tlast);

  // every tile is looked at by exactly one thread of one workgroup per round
  uint32_t carry_min = kInfBits;
  const uint32_t t_end_owned = P.t_hi ? P.t_hi : P.ntiles;
  for (uint32_t t = P.t_lo + blockIdx.x + (uint32_t)tid * gridDim.x; t < t_end_owned; t += gridDim.x * kTileBlock) {
    const uint32_t pb = pc[t];
    if (pb == kInfBits) continue;
    pc[t] = kInfBits;
    const float p = u2f(pb);
    if (!(p <= bound)) {                                            // can never propagate any more; the finalize pass
      if (!(g_tlast[t] > -inf_f())) g_tlast[t] = -3.0e38f;          // still has to visit the tile (finite mark, filters nothing)
      continue;
    }
    bool take = false;
    if (p < thr) {
      const uint32_t k = atomicAdd(&s_ntodo, 1u);
      if (k < (uint32_t)kTileTodo) {
        s_todo[k] = t; take = true;                                 // ... and fetches the tile header
        s_hdr[k][0] = g_vptr[t]; s_hdr[k][1] = g_vptr[t + 1]; s_hdr[k][2] = g_hptr[t]; s_hdr[k][3] = g_hptr[t + 1];
        s_hdr[k][4] = g_eptr[t]; s_hdr[k][5] = g_eptr[t + 1]; s_hdr[k][6] = g_rptr[t]; s_hdr[k][7] = f2u(g_tlast[t]);
      }
    }
    if (!take) { atomicMin((uint32_t*)&pn[t], pb); carry_min = min(carry_min, pb); }   // carry the wake-up over
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) carry_min = min(carry_min, (uint32_t)__shfl_xor((int)carry_min, o));
  if ((tid & 63) == 0 && carry_min != kInfBits) atomicMin(&cnt->minpend, carry_min);
  __syncthreads();
  const uint32_t ntodo = min(s_ntodo, (uint32_t)kTileTodo);
  if (ntodo == 0) return;
  TT_STAMP(2);
  TT_STAMP(3);

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const TileLds L = tile_lds_layout(smem, P.max_nv, P.max_nh, P.max_ne);
  uint32_t* const ldu = L.ldu; uint32_t* const lh0 = L.lh0; uint16_t* const q0 = L.q0;

  for (uint32_t ti = 0; ti < ntodo; ++ti) {
    const uint32_t t = s_todo[ti];
    const uint32_t v0 = s_hdr[ti][0], nv = s_hdr[ti][1] - v0;
    const uint32_t h0 = s_hdr[ti][2], nh = s_hdr[ti][3] - h0;
    const uint32_t e0 = s_hdr[ti][4], ne = s_hdr[ti][5] - e0;
    const uint32_t r0 = s_hdr[ti][6];
    const uint32_t nl = nv + nh;
    const float tl = u2f(s_hdr[ti][7]);
    if (tid == 0) { s_nq[0] = 0; s_nq[1] = 0; s_nq[2] = 0; s_left = kInfBits; }
    __syncthreads();
    // stage: all index / bulk loads in flight at once (16-byte vectors), then the distance gathers
    uint32_t gi[kTileVpt];
#pragma unroll
    for (int k = 0; k < kTileVpt; ++k) { const uint32_t i = tid + k * kTileBlock; gi[k] = (i < nv) ? g_verts[v0 + i] : 0u; }
    uint32_t hi[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) { const uint32_t i = tid + k * kTileBlock; hi[k] = (i < nh) ? g_halo_verts[h0 + i] : 0u; }
    stage_tile_graph(P, L, e0, ne, r0, nl, tid);
    uint32_t orig[kTileVpt];
#pragma unroll
    for (int k = 0; k < kTileVpt; ++k) {
      const uint32_t i = tid + k * kTileBlock;
      orig[k] = 0u;
      if (i < nv) {
        const float d = g_dist[gi[k]];
        orig[k] = f2u(d); ldu[i] = orig[k];
        if (d < thr && d <= bound && !(d < tl)) q0[atomicAdd(&s_nq[0], 1u)] = (uint16_t)i;   // owned sources in [tlast, thr)
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const uint32_t i = tid + k * kTileBlock;
      if (i < nh) { const float d = g_dist[hi[k]]; ldu[nv + i] = f2u(d); lh0[i] = f2u(d); if (d < thr && d <= bound) q0[atomicAdd(&s_nq[0], 1u)] = (uint16_t)(nv + i); }
    }
    for (uint32_t i = tid + 2 * kTileBlock; i < nh; i += kTileBlock) {
      const float d = g_dist[g_halo_verts[h0 + i]];
      ldu[nv + i] = f2u(d); lh0[i] = f2u(d); if (d < thr && d <= bound) q0[atomicAdd(&s_nq[0], 1u)] = (uint16_t)(nv + i);
    }
    __syncthreads();
    TT_STAMP(4);
    const uint32_t sweep = tile_sweeps(L, nv, thr, bound, s_nq, tid);
    TT_STAMP(5);
    // wake the owners of the halo vertices we undercut (value = the candidate we found for them)
    uint32_t left = kInfBits, own_left = kInfBits;
    for (uint32_t i = tid; i < nh; i += kTileBlock) {
      const uint32_t b = ldu[nv + i];
      if (b < lh0[i]) { atomicMin((uint32_t*)&pn[g_halo_tile[h0 + i]], b); left = min(left, b); }
    }
    // write back what moved; remember the smallest owned value that still has to propagate
#pragma unroll
    for (int k = 0; k < kTileVpt; ++k) {
      const uint32_t i = tid + k * kTileBlock;
      if (i < nv) {
        const uint32_t db = ldu[i];
        if (db != orig[k]) g_dist[gi[k]] = u2f(db);
        const float d = u2f(db);
        if (!(d < thr) && d <= bound) { if (db < left) left = db; own_left = min(own_left, db); }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      left = min(left, (uint32_t)__shfl_xor((int)left, o));
      own_left = min(own_left, (uint32_t)__shfl_xor((int)own_left, o));
    }
    if ((tid & 63) == 0) {
      if (left != kInfBits) atomicMin(&s_left, left);
      if (own_left != kInfBits) atomicMin((uint32_t*)&pn[t], own_left);
    }
    __syncthreads();
    if (tid == 0) {
      g_tlast[t] = thr;
      const uint32_t l = s_left;                                   // own left-overs and halo wake-ups
      if (l != kInfBits) atomicMin(&cnt->minpend, l);
      atomicAdd(&cnt->acts, 1u); atomicAdd(&cnt->sweeps, sweep);
    }
    __syncthreads();
#ifdef MNAV_TILE_TIMING
    if (tid == 0 && ti == 0) {
      tt[6] = clock64(); tt[7] = ((unsigned long long)cur.it << 32) | sweep;
      const unsigned int k = atomicAdd(&g_tile_timing_n, 1u);
      if (k < 4096) for (int q = 0; q < 8; ++q) g_tile_timing[k * 8 + q] = tt[q];
    }
#endif
  }
}

#include "mnav_async.h"   // k_plan_async: the tiles without rounds (engine 6)

__global__ __launch_bounds__(kBlock) void k_tile_init(const TilePlan* __restrict__ plans, const uint32_t* __restrict__ vert_tile, float tlast0)
{
  const TilePlan& P = plans[blockIdx.y];
  const uint32_t stride = gridDim.x * kBlock;
  const uint32_t st = vert_tile[P.seed];
  for (uint32_t t = blockIdx.x * kBlock + threadIdx.x; t < P.ntiles; t += stride) {
    if (P.pend[1] != P.pend[0]) P.pend[1][t] = kInfBits;            // (the per-plan engines use a single buffer)
    P.pend[0][t] = (t == st) ? 0u : kInfBits;                       // the seed's tile wakes at 0
    P.tlast[t] = tlast0;                                            // -inf: never solved (the finalize pass skips the tile unless it was woken)
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    P.dist[P.seed] = 0.0f;                                          // dijkstra :276
    TCtl c0; memset(&c0, 0, sizeof(c0));
    c0.it = -1; c0.done = 0; c0.thr = -inf_f(); c0.thr_prev = -inf_f();
    P.ctl[0] = c0; P.ctl[1] = c0;
    TCnt ci; ci.minpend = 0u; ci.acts = 0; ci.sweeps = 0; ci.pad = 0;
    P.cnt[2] = ci;
    TCnt z; z.minpend = kInfBits; z.acts = 0; z.sweeps = 0; z.pad = 0;
    P.cnt[0] = z; P.cnt[1] = z;
  }
}

// per-tile weights from the (cost-limit folded) gather CSR
__global__ __launch_bounds__(kBlock) void k_tile_weights(uint32_t n, const uint32_t* __restrict__ src, const uint16_t* __restrict__ col,
                                                         const Nbr* __restrict__ nbr, float* __restrict__ tw)
{
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) tw[i] = (src[i] == kNone) ? inf_f() : nbr[src[i]].w;       // padding never relaxes anything
  (void)col;
}
