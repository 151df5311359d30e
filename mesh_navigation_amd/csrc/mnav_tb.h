// mnav_tb.h -- TILE-BATCH SSSP engine: many Dijkstra wavefronts (dijkstra_mesh_planner.cpp:287-348, one per plan of a
// batch) advanced together, tile-major and plan-vectorised.  Included by mnav.hip (device kernels + host driver).
//
// Why: with one workgroup per plan (k_plan_persistent) every tile activation re-stages the tile's graph and runs a
// latency-bound chain of LDS queue sweeps for ONE plan.  Here the lanes of a wave are PLANS that have work on the same tile:
// 16 plans per quarter of a wave, the four quarters on the same tile or on four different ones (k_tb_solve_q).  The lanes of a
// quarter execute the same edge sequence, so the tile's graph is uniform data per quarter: flat record streams, parked chunk
// by chunk in a per-quarter LDS staging area and read back with broadcast ds_read_b128; the distances of the tile's vertices
// live in LDS as [vertex][lane] (a lane only ever touches its own column: no bank conflicts, no barriers, no atomics), and the
// relaxation is a plain Gauss-Seidel sweep over the tile's vertices in one of four diagonal orders until a sweep
// changes nothing in any lane -- the tile-local fixed point of  d[v] = min_u fl(d[u] + w(u,v)),  which is the
// reference's float32 relaxation (dijkstra :331) and has a unique fixed point, so any schedule reproduces it bit
// for bit.
//
// Schedule (level-synchronous over all plans, a few launches per iteration, replayed from a hipGraph):
//   k_tb_plan    per plan: band threshold thr = (smallest pending wake-up) + band, bound = dist[target] + offset
//   k_tb_scan    every pending value pend[tile][plan]: < thr -> the tile's bucket; > bound -> dropped; else carried (counted)
//   k_tb_items   per tile: the bucket is cut into work items of <= 16 plans
//   k_tb_solve_q per item: load the plans' slices, fold the ghost values in, sweep, write back what changed, wake the
//                neighbouring tiles whose vertices were undercut, export changed boundary values to their ghost slots
// Data written during an iteration is consumed in the next one (kernel boundary), so there is no intra-kernel
// producer/consumer protocol; wake-ups use atomicMin on the pair's wake-up value, the first waker counts the pair.
#pragma once

#include "mnav_tb_build.h"

namespace {

namespace tb {

// Tile headers and export records are read through the constant address space (uniform addresses there are always
// scalar loads); the edge streams come in 256-byte chunks through the vector memory path, one dword per lane, are parked
// in an LDS staging buffer and read back with uniform-address (broadcast) ds_read_b128 (mnav_tb_build.h).
#define MNAV_CONST __attribute__((address_space(4)))
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));   // 4 export records / a tile header
typedef const MNAV_CONST u32x16* cblk8_t;
typedef __attribute__((address_space(3))) uint32_t* lds_u32_t;
// word indices of a TbTile read as u32x16
enum { kTwSoff = 0, kTwSl = 1, kTwNv = 2, kTwNh = 3, kTwSweepOff = 4, kTwSweepChunks = 5, kTwPreOff = 6, kTwPreChunks = 7,
       kTwPostOff = 8, kTwPostChunks = 9, kTwExpOff = 10, kTwExpN = 11 };
static_assert(offsetof(TbTile, exp_n) == 4 * kTwExpN && offsetof(TbTile, sweep_off) == 4 * kTwSweepOff, "TbTile layout");

struct Ctl {
  uint32_t n_cand[2];        // 1: (tile, plan) pairs are pending after / during an iteration, by iteration parity (0 at the end: converged)
  uint32_t n_items, next_item;
  uint32_t err;              // 1: sweep cap hit
  uint32_t iters;
  unsigned long long acts;   // (tile, plan) activations
  unsigned long long sweeps; // wave sweeps
  unsigned long long items;  // work items (waves of <= 64 plans)
  unsigned long long wakes;
  uint32_t n_pairs, pad_;    // (tile, block of 64 plans) pairs whose pending flag is set this iteration (k_tb_pairs -> k_tb_scan)
};

struct Args {
  const TbTile* tiles; const uint32_t* stream; const TbExp* exps;
  float* D; uint32_t* pend; uint32_t NP, ntiles;
  uint16_t* bucket; uint32_t* bcnt; uint2* items; Ctl* ctl;
  uint32_t* marr[2];
  float* thr; float* bnd;
  const uint32_t* seed; const uint32_t* target;                        // per plan: wave source / robot vertex
  const uint2* vaddr; const uint32_t* vert_tile;                      // per vertex: {soff, sl << 8 | local}, tile
  double offset; float band;
  uint8_t* pflag; uint32_t nblk;                                     // per (tile, block of 64 plans): 1 = some pend[tile][plan] of the block may be set
  uint32_t* pairs; uint32_t n_flag16;                                // the flagged pairs of the iteration (tile * nblk + block); 16-byte units of the flag matrix
  uint32_t nblk_magic, nblk_shift;                                   // pair / nblk as one multiply-high and a shift (tb::div_magic; k_tb_scan divides once per listed row)
  unsigned long long* wstat; uint32_t wstat_slots;                  // statistics per persistent wave {items, activations, sweeps, wakes}: k_tb_stats sums them
  uint32_t item_plans;                                               // plans per work item: 16 (a quarter of a wave, k_tb_solve_q) or 64 (a wave, k_tbv_solve)
};

__device__ __forceinline__ size_t slot_addr(const uint2 va, uint32_t NP, uint32_t p)
{
  return (size_t)va.x * NP + (size_t)p * (va.y >> 8) + (va.y & 255u);
}

__device__ __forceinline__ uint32_t ldsr(uint32_t off) { return *(lds_u32_t)(uintptr_t)off; }
__device__ __forceinline__ void ldsw(uint32_t off, uint32_t v) { *(lds_u32_t)(uintptr_t)off = v; }
typedef __attribute__((address_space(3))) u32x4* lds_u32x4_t;
__device__ __forceinline__ u32x4 ldsr4(uint32_t off) { return *(lds_u32x4_t)(uintptr_t)off; }   // uniform address: broadcast

__device__ __forceinline__ uint32_t rfl(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }

// n / d for n < 2^31 as one multiply-high and a shift (mnav::tb_div_magic, mnav_tb_build.h, makes the pair on the host)
__device__ __forceinline__ uint32_t div_by_magic(uint32_t n, uint32_t d, uint32_t magic, uint32_t shift)
{
  return d <= 1u ? n : (__umulhi(n, magic) >> shift);                  // (= mnav::tb_div_by_magic, the host's copy the CPU test runs)
}

}  // namespace tb

__global__ __launch_bounds__(kBlock) void k_tb_weights(size_t n, const uint32_t* __restrict__ wsrc, const Nbr* __restrict__ nbr, uint32_t* __restrict__ stream)
{
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const uint32_t s = wsrc[i];
    if (s != kNone) stream[i] = f2u(nbr[s].w);
  }
}

__global__ __launch_bounds__(kBlock) void k_tb_fill(u32x4* __restrict__ p, size_t n16, uint32_t v)
{
  const size_t stride = (size_t)gridDim.x * kBlock;
  const u32x4 x = { v, v, v, v };
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n16; i += stride) __builtin_nontemporal_store(x, p + i);
}

// seeds: distance 0 at the wave source (dijkstra :272-277), exported to the ghost slots that mirror it, its tile pending
__global__ __launch_bounds__(kBlock) void k_tb_seed(tb::Args A)
{
  const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= A.NP) return;
  const uint32_t s = A.seed[p];
  const uint2 va = A.vaddr[s];
  A.D[tb::slot_addr(va, A.NP, p)] = 0.0f;
  const uint32_t t = A.vert_tile[s];
  const TbTile W = A.tiles[t];
  const uint32_t row = (va.y & 255u) * 256u;
  for (uint32_t k = 0; k < W.exp_n; ++k) {
    const TbExp e = A.exps[W.exp_off + k];
    if (e.u == row) A.D[(size_t)e.soff * A.NP + (size_t)p * e.sl + e.off] = 0.0f;
  }
  A.pend[(size_t)t * A.NP + p] = 0u;
  A.pflag[(size_t)t * A.nblk + (p >> 6)] = 1;
  A.marr[0][p] = 0u;
  A.ctl->n_cand[0] = 1u;                                              // (a flag: see k_tb_scan)
}

// per plan and iteration: band threshold and goal bound
__global__ __launch_bounds__(kBlock) void k_tb_plan(tb::Args A, int par)
{
  const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
  if (p == 0) { A.ctl->n_cand[par ^ 1] = 0u; A.ctl->next_item = 0u; A.ctl->n_items = 0u; A.ctl->n_pairs = 0u; A.ctl->iters += 1u; }
  if (p >= A.NP) return;
  const uint32_t mb = A.marr[par][p];
  A.marr[par ^ 1][p] = kTbInfBits;
  const float m = u2f(mb);
  const float dt = A.D[tb::slot_addr(A.vaddr[A.target[p]], A.NP, p)];
  const float bound = (float)((double)dt + fmax(A.offset, 0.0));    // >= the final goal_dist (dijkstra :296); a negative offset is applied after the rounds (goal_cut)
  const bool done = !(m < inf_f()) || m > bound;
  float thr = m + A.band;
  if (!(thr > m)) thr = next_up(m);
  A.thr[p] = done ? 0.0f : thr;                                     // wake-up values are >= 0: nothing is below 0
  A.bnd[p] = bound;
}

// The pending values pend[tile][plan] (+inf: none) once per iteration: an entry below its plan's band threshold is READY (cleared,
// the plan goes into the tile's bucket), one beyond the goal bound is dropped (it can never propagate any more: the bound only
// shrinks), the rest is CARRIED (counted, and its smallest value per plan is next iteration's band start).  The matrix is sparse --
// a plan's pending tiles are the ring around its front --: one byte per (tile, block of 64 plans), set by whoever writes a pending
// value, says whether a 256-byte row of it may hold anything.  Rounds 3-4 walked ALL rows (a thread per plan down a column of 16
// tiles, skipping rows by their flag): 92 600 tiles x 64 blocks of flags per iteration on the 10M mesh, 21 % of the engine run
// (profiles/r05_c4_kernel_stats.md).  Now k_tb_pairs compacts the set flags into a list (the 5.9 MB flag matrix read as 16-byte
// units: microseconds) and k_tb_scan gives one wave to each listed row.
constexpr uint32_t kTbPairUnits = 4;                                 // 16-byte units of the flag matrix per thread of k_tb_pairs
// (one atomic per WORKGROUP of 256 threads x 4 units = 16 384 flags: the list's counter is one address, and one atomic per wave of 64
// units was 5 800 atomics in a row at the L2 on the 10M mesh -- 52 us for a pass that reads 6 MB)
__global__ __launch_bounds__(kBlock) void k_tb_pairs(tb::Args A)
{
  __shared__ uint32_t s_wave[kBlock / 64], s_base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t i0 = (blockIdx.x * kBlock + threadIdx.x) * kTbPairUnits;
  u32x4 f[kTbPairUnits];
  uint32_t c = 0;
#pragma unroll
  for (uint32_t u = 0; u < kTbPairUnits; ++u) {
    f[u] = u32x4{ 0u, 0u, 0u, 0u };
    if (i0 + u < A.n_flag16) f[u] = ((MNAV_GLOBAL const u32x4*)as_global(A.pflag))[i0 + u];
    const uint32_t w[4] = { f[u].x, f[u].y, f[u].z, f[u].w };
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int b = 0; b < 4; ++b) c += ((w[k] >> (8 * b)) & 0xFFu) ? 1u : 0u;
  }
  uint32_t incl = c;                                                 // inclusive scan over the wave, then over the workgroup's waves
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const uint32_t x = __shfl_up(incl, o); if (lane >= o) incl += x; }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tot = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) { const uint32_t x = s_wave[w]; s_wave[w] = tot; tot += x; }
    s_base = tot ? atomicAdd(&A.ctl->n_pairs, tot) : 0u;
  }
  __syncthreads();
  uint32_t base = s_base + s_wave[wave] + incl - c;
  if (c) {
#pragma unroll
    for (uint32_t u = 0; u < kTbPairUnits; ++u) {
      const uint32_t w[4] = { f[u].x, f[u].y, f[u].z, f[u].w };
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          if ((w[k] >> (8 * b)) & 0xFFu) A.pairs[base++] = 16u * (i0 + u) + 4u * k + b;
    }
  }
}

constexpr uint32_t kTbStatSlots = 16384;                             // >= the persistent waves of a solve launch (k_tb_stats)
constexpr uint32_t kTbScanWaves = 8192;                              // persistent waves of k_tb_scan (a row of the list after the other)
// One listed row per wave and turn.  A row is a chain of dependent round trips -- list entry -> the row of pending values and the
// plans' thresholds -> the bucket's counter -> the bucket entries.  What a turn issues, in this order: (B) the current row is
// classified (its loads were issued a turn ago) and lane 0 asks the bucket's counter for slots; (A) the NEXT row's loads go out --
// pending values, thresholds, bounds, the plans' carried minima, and the list entry after that one --, every one of them
// unconditional at a clamped address; (C) the bucket entries are stored once the counter has answered.  The memory counter counts
// in order, so (C) waits for the atomic with the loads of (A) still in flight (s_waitcnt vmcnt(5) in the ISA), and (B) of the next
// turn finds them arrived.  Measured: no faster than the unpipelined loop (102 us per launch at 7168 plans either way) -- the pass
// is not waiting for memory, it is issuing: ~200 000 listed rows per iteration x 130 instructions, 8 waves per SIMD, for 6-16 %
// of the lanes holding a value.  (A list of the VALUES instead of the rows was tried too)
__global__ __launch_bounds__(kBlock) void k_tb_scan(tb::Args A, int par)
{
  const int lane = threadIdx.x & 63;
  const uint32_t wave = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), nwaves = gridDim.x * (kBlock / 64);
  const uint32_t n_pairs = A.ctl->n_pairs;
  if (wave >= n_pairs) return;
  MNAV_GLOBAL const uint32_t* const pairs = as_global(A.pairs);
  MNAV_GLOBAL const float* const g_thr = as_global(A.thr);
  MNAV_GLOBAL const float* const g_bnd = as_global(A.bnd);
  MNAV_GLOBAL uint32_t* const g_pend = as_global(A.pend);
  MNAV_GLOBAL uint32_t* const g_min = as_global(A.marr[par ^ 1]);
  uint32_t carried = 0;
  struct Row { uint32_t pr, t, p, pb, pm; float thr, bnd; bool live; };
  auto fetch = [&](uint32_t pr, bool have) {                          // the loads of one row, all issued together and none of them under a branch
    Row r; r.pr = pr;
    const uint32_t t = tb::div_by_magic(pr, A.nblk, A.nblk_magic, A.nblk_shift), p = (pr - t * A.nblk) * 64u + (uint32_t)lane;
    r.live = have && t < A.ntiles && p < A.NP;                         // (padding bytes of the flag matrix are never set)
    r.t = r.live ? t : 0u; r.p = r.live ? p : 0u;
    r.pb = g_pend[(size_t)r.t * A.NP + r.p]; r.thr = g_thr[r.p]; r.bnd = g_bnd[r.p]; r.pm = g_min[r.p];
    return r;
  };
  uint32_t k = wave;
  const uint32_t last = n_pairs - 1u;
  uint32_t pr_next = pairs[min(k + nwaves, last)];
  Row cur = fetch(pairs[k], true);
  for (; k < n_pairs; k += nwaves) {
    // ---- (B) the current row
    asm volatile("" :: "v"(cur.pb), "v"(cur.thr), "v"(cur.bnd), "v"(cur.pm));   // (all four loads waited for HERE: a value only read under a
                                                                       // branch stays "in flight" for the compiler on the other path, and it then
                                                                       // makes room for the next loads by waiting for everything issued since)
    const uint32_t t = cur.t, p = cur.p, pb = cur.pb;
    MNAV_GLOBAL uint32_t* const pe = g_pend + ((size_t)t * A.NP + p);
    bool ready = false, keep = false;
    if (cur.live && pb != kTbInfBits) {
      const float pv = u2f(pb);
      if (pv > cur.bnd) *pe = kTbInfBits;
      else if (pv < cur.thr) { *pe = kTbInfBits; ready = true; }
      else {
        keep = true; ++carried;
        if (pb < cur.pm) atomicMin((uint32_t*)(g_min + p), pb);       // (cur.pm is a turn old: the minimum only falls, an older value only lets more through)
      }
    }
    if (!__any(keep) && lane == 0) A.pflag[cur.pr] = 0;
    const unsigned long long m = __ballot(ready);
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&A.bcnt[t], (uint32_t)__popcll(m));   // (+0 when nothing is ready: the atomic is not worth a branch)
    // ---- (A) the next row's loads, and the list entry after it
    const bool have_next = k + nwaves < n_pairs;
    const uint32_t pr_after = pairs[min(k + 2u * nwaves, last)];
    const Row nxt = fetch(pr_next, have_next);
    pr_next = pr_after;
    __builtin_amdgcn_sched_barrier(0);                                 // (the scheduler would pull the read of `base`, and with it the wait, above the loads)
    // ---- (C) the bucket entries
    base = tb::rfl(base);
    if (ready) A.bucket[(size_t)t * A.NP + base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)p;
    cur = nxt;
  }
  carried = wave_sum(carried);
  // "something is still pending" is all anybody asks of this word (the host's loop, the finalize pass): a plain store of 1, not
  // a count -- 8192 waves adding to ONE address were 8192 atomics in a row at the L2 at the end of every launch
  if (lane == 0 && carried) A.ctl->n_cand[par ^ 1] = 1u;
}

constexpr uint32_t kTbItemPlans = 16;     // plans per work item = lanes per quarter of a wave (k_tb_solve_q)
// per tile: cut the bucket into items of <= kTbItemPlans plans.  One thread per tile, one atomic per wave for the item slots (the
// items of a tile stay adjacent; their order among the tiles is whatever the atomics make it -- the fixed point does not care).
// Rounds 3-4 ran ONE workgroup over all tiles for a deterministic order: 122 us per iteration at 92 600 tiles, 8 % of the C4 run.
__global__ __launch_bounds__(kBlock) void k_tb_items(tb::Args A)
{
  const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const uint32_t gran = A.item_plans;
  uint32_t c = 0;
  if (t < A.ntiles) { c = A.bcnt[t]; if (c) A.bcnt[t] = 0u; }
  const uint32_t k = (c + gran - 1u) / gran;
  uint32_t incl = k;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const uint32_t x = __shfl_up(incl, o); if (lane >= o) incl += x; }
  __shared__ uint32_t s_wave[kBlock / 64], s_base;                    // (one atomic per workgroup: see k_tb_pairs)
  const int wave = threadIdx.x >> 6;
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tot = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) { const uint32_t x = s_wave[w]; s_wave[w] = tot; tot += x; }
    s_base = tot ? atomicAdd(&A.ctl->n_items, tot) : 0u;
  }
  __syncthreads();
  const uint32_t base = s_base + s_wave[wave] + incl - k;
  for (uint32_t q = 0; q < k; ++q) A.items[base + q] = make_uint2(t, (q * gran) | (min(gran, c - q * gran) << 16));
}

// The statistics of the run: every persistent wave of the solve kernels keeps its own four counters (a slot per wave, plain
// adds), one block sums them into the control record at the end of a chunk of iterations.  (Until round 6 every wave ended with four
// atomicAdds on the control record: 8192 atomics on ONE cache line in a row at the L2, at the end of every launch.)
__global__ __launch_bounds__(1024) void k_tb_stats(tb::Args A)
{
  __shared__ unsigned long long s_part[16][4];
  unsigned long long acc[4] = { 0ull, 0ull, 0ull, 0ull };
  for (uint32_t w = threadIdx.x; w < A.wstat_slots; w += 1024u)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] += A.wstat[4u * w + k];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc[k] += __shfl_xor(acc[k], o);
  if ((threadIdx.x & 63u) == 0u)
#pragma unroll
    for (int k = 0; k < 4; ++k) s_part[threadIdx.x >> 6][k] = acc[k];
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t[4] = { 0ull, 0ull, 0ull, 0ull };
    for (int w = 0; w < 16; ++w)
      for (int k = 0; k < 4; ++k) t[k] += s_part[w][k];
    A.ctl->items = t[0]; A.ctl->acts = t[1]; A.ctl->sweeps = t[2]; A.ctl->wakes = t[3];
  }
}

// One block of a Gauss-Seidel sweep: the target row is relaxed from up to 7 source rows (dijkstra :331).  No branch: the row is
// rewritten unconditionally (old bits when nothing improved).
struct TbBlk { uint32_t ya, raw, v[7]; };

#ifdef MNAV_TB_TIMING                      // debugging aid: cycles per phase of k_tb_solve_q, summed over all waves
__device__ unsigned long long g_tb_timing[8];
#define TB_STAMP(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); tt[k] += now_ - t_last; t_last = now_; } while (0)
#else
#define TB_STAMP(k) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------
// The quarter-wave solve: one wave per FOUR work items of <= 16 plans, one item per 16-lane quarter.
// ---------------------------------------------------------------------------------------------
// One wave per (tile, <= 64 plans) -- round 3's solve -- fills a wave with the plans that have work on ONE tile in one iteration:
// 42-50 of 64 lanes on the 1M mesh, 15 of 64 on the 10M mesh, where the batch cannot grow any further (the blocked distances
// fill the HBM), 1-2 in small batches.  Here the buckets are cut into items of <= 16 plans and a wave takes four of them -- of
// the same tile or of four different ones (measured against the old kernel: C4 batch 3555 -> 2184 ms, C2 234 -> 220 ms once the
// export records were read as a stream too, 2048 plans 136 -> 110 ms; the old kernel is gone).  Everything that was wave-uniform becomes uniform per QUARTER: the tile header sits in VGPRs, each quarter
// parks its own tile's stream chunk in its own 256-byte staging area (lane l loads the 16 bytes (l & 15) of its
// quarter's chunk: still one vector load per lane and chunk) and reads the descriptors back with ds_read_b128 at a
// per-quarter address (the 16 lanes of a quarter read one address: a broadcast per 8-lane pass, like the uniform read).
// Streams of different length run to the longest one: a quarter past the end of its sweep stream re-runs its last
// chunk (a relaxation applied twice changes nothing), past the end of a ghost stream it idles.  The staging is
// single-buffered: all descriptor reads of a chunk are issued before the next chunk is written over it (the LDS
// executes in order), which keeps a wave at 31.8 KB of LDS -- five waves per CU as before.
constexpr uint32_t kTbQStride = 272;      // bytes between the quarters' staging areas: 256 + 16, so that they start in different banks

namespace tb {
__device__ __forceinline__ void ldsw4(uint32_t off, u32x4 v) { *(lds_u32x4_t)(uintptr_t)off = v; }
__device__ __forceinline__ uint32_t wmax4(uint32_t x)                // max over the four quarters (x uniform per quarter), wave-uniform
{
  x = max(x, (uint32_t)__shfl_xor((int)x, 16));
  x = max(x, (uint32_t)__shfl_xor((int)x, 32));
  return rfl(x);
}
// Per-quarter stream reader.  The staging area holds chunk c; chunk c + 1 waits in a register, chunk c + 2 is in flight.
struct QStream {
  MNAV_GLOBAL const u32x4* st;   // this lane's 16 bytes of chunk 0
  uint32_t last, wr, c; u32x4 c1;
  // `first`: the 16-byte unit 0 of chunk 0
  __device__ __forceinline__ void begin_at(MNAV_GLOBAL const u32x4* first, uint32_t nch, uint32_t stage_q, uint32_t l16)
  {
    st = first + l16;
    last = nch ? nch - 1u : 0u; wr = stage_q + 16u * l16; c = 0;
    ldsw4(wr, st[0]);
    c1 = st[(size_t)min(1u, last) * 16u];
  }
  __device__ __forceinline__ void begin(MNAV_GLOBAL const uint32_t* stream, uint32_t chunk_off, uint32_t nch, uint32_t stage_q, uint32_t l16)
  {
    begin_at((MNAV_GLOBAL const u32x4*)(stream + (size_t)chunk_off * kTbChunk), nch, stage_q, l16);
  }
  // call when every LDS read of chunk c has been issued
  __device__ __forceinline__ void advance()
  {
    ldsw4(wr, c1);
    c1 = st[(size_t)min(c + 2u, last) * 16u];
    ++c;
  }
};
}  // namespace tb

// One Gauss-Seidel sweep of a quarter's tile.  The sweep chunks are stored transposed (mnav_tb_build.h, tb_sweep_index): lane l
// of a quarter loads the 16 bytes l of the quarter's chunk -- register k of the load holds dword l of block k in lane l -- and
// every descriptor dword is consumed as a DPP operand of the instruction that needs it (row_newbcast:q = lane q of each 16-lane
// row, i.e. of each quarter, broadcast to the row): v_add_u32_dpp for the eight LDS addresses, v_add_f32_dpp with |.| on the
// other operand for the seven relaxations.  No LDS staging, no descriptor reads: the LDS carries the eight data reads and the
// one write of a block and nothing else (before: four broadcast ds_read_b128 per block on top, two thirds of the kernel's LDS
// cycles -- profiles/r05_c2_sq.md).  The two chunks after the current one are in registers or in flight.
namespace tb {
template <int Q> __device__ __forceinline__ uint32_t bc(uint32_t d)   // dword Q of the block whose descriptor register is d
{
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)d, 0x150 + Q, 0xf, 0xf, true);   // row_newbcast:Q, folded into the consumer
}
}  // namespace tb

// One block of the sweep in two halves.  tb_issue: the LDS reads (FIRST: block 0 of a chunk reads all seven sources; a follower
// reads six -- its slot 1 is the row the block before it writes, mnav_tb_build.h).  tb_retire: the relaxation (dijkstra :331) and
// the unconditional rewrite of the target row (old bits when nothing improved; the sign bit marks "lowered in this activation");
// `fwd` = what the block before wrote, a follower's slot-1 source.  Returns the bits written.
template <bool FIRST>
__device__ __forceinline__ void tb_issue(TbBlk& B, uint32_t D, uint32_t lane4)
{
  B.ya = tb::bc<0>(D) + lane4;
  B.raw = tb::ldsr(B.ya);
  if (FIRST) B.v[0] = tb::ldsr(tb::bc<1>(D) + lane4);
  B.v[1] = tb::ldsr(tb::bc<2>(D) + lane4); B.v[2] = tb::ldsr(tb::bc<3>(D) + lane4);
  B.v[3] = tb::ldsr(tb::bc<4>(D) + lane4); B.v[4] = tb::ldsr(tb::bc<5>(D) + lane4); B.v[5] = tb::ldsr(tb::bc<6>(D) + lane4);
  B.v[6] = tb::ldsr(tb::bc<7>(D) + lane4);
}
template <bool FIRST>
__device__ __forceinline__ uint32_t tb_retire(const TbBlk& B, uint32_t D, uint32_t fwd, bool& ch)
{
  const uint32_t acc0 = B.raw & 0x7fffffffu;
  const uint32_t s0 = FIRST ? B.v[0] : fwd;
  const uint32_t t0 = f2u(fabsf(u2f(s0)) + u2f(tb::bc<8>(D))), t1 = f2u(fabsf(u2f(B.v[1])) + u2f(tb::bc<9>(D)));
  const uint32_t t2 = f2u(fabsf(u2f(B.v[2])) + u2f(tb::bc<10>(D))), t3 = f2u(fabsf(u2f(B.v[3])) + u2f(tb::bc<11>(D)));
  const uint32_t t4 = f2u(fabsf(u2f(B.v[4])) + u2f(tb::bc<12>(D))), t5 = f2u(fabsf(u2f(B.v[5])) + u2f(tb::bc<13>(D)));
  const uint32_t t6 = f2u(fabsf(u2f(B.v[6])) + u2f(tb::bc<14>(D)));
  uint32_t acc = min(min(acc0, t0), t1);
  acc = min(min(acc, t2), t3); acc = min(min(acc, t4), t5); acc = min(acc, t6);
  ch = acc < acc0;
  const uint32_t w = ch ? (acc | kTbDirty) : B.raw;
  tb::ldsw(B.ya, w);
  return w;
}
// A chunk = four blocks, software-pipelined: the reads of block j + 1 are issued before block j is computed and written.  That
// is the plain Gauss-Seidel sweep bit for bit: block j + 1 never has block j's target, its only source that IS block j's target
// sits in slot 1 and is taken from block j's result register, and everything older was written before the reads were issued (the
// LDS executes in order).  Drained at the end of the chunk (block 0 of the next chunk reads everything from the LDS).
__device__ __forceinline__ bool tb_chunk(const u32x4& ch4, uint32_t lane4)
{
  TbBlk b0, b1, b2, b3;
  bool c0, c1, c2, c3;
  tb_issue<true>(b0, ch4.x, lane4);
  tb_issue<false>(b1, ch4.y, lane4);
  const uint32_t w0 = tb_retire<true>(b0, ch4.x, 0u, c0);
  tb_issue<false>(b2, ch4.z, lane4);
  const uint32_t w1 = tb_retire<false>(b1, ch4.y, w0, c1);
  tb_issue<false>(b3, ch4.w, lane4);
  const uint32_t w2 = tb_retire<false>(b2, ch4.z, w1, c2);
  (void)tb_retire<false>(b3, ch4.w, w2, c3);
  return c0 | c1 | c2 | c3;
}

// All sweeps of an activation as ONE stream of chunks: three chunk registers in rotation, the load cursor three chunks ahead of the
// compute cursor and running on into the next sweep's order while the current sweep finishes (speculatively: when the sweep turns
// out to have changed nothing, the three loads in flight are dropped) -- no stall at a sweep boundary.  Each load is followed by
// an empty memory-clobbering asm statement: without it the compiler sinks the load to its use three chunks later (a full memory
// latency per three chunks, seen in the ISA); with it the load is issued where it is written and waited for where it is used.
// A quarter past the end of its own stream re-runs its last chunk.  Returns the number of sweeps (the last one changed nothing in
// any lane), or 0 with `overrun` set when the cap was hit.
#ifndef MNAV_TB_DEPTH
#define MNAV_TB_DEPTH 8
#endif
constexpr int kTbDepth = MNAV_TB_DEPTH;   // stream chunks in registers or in flight per wave
#ifndef MNAV_TB_ROT
#define MNAV_TB_ROT 2
#endif
constexpr int kTbRot = MNAV_TB_ROT;
template <int T>
__device__ __forceinline__ uint32_t tbq_sweeps(MNAV_GLOBAL const uint32_t* stream, uint32_t sweep_off, uint32_t nch, uint32_t max_nch, uint32_t first_order,
                                               uint32_t l16, uint32_t lane4, bool& overrun)
{
  MNAV_GLOBAL const u32x4* const st0 = (MNAV_GLOBAL const u32x4*)(stream + (size_t)sweep_off * kTbChunk) + l16;   // this lane's 16 bytes of chunk 0 of order 0
  const uint32_t last = nch ? nch - 1u : 0u;
  uint32_t s_ld = 0, c_ld = 0;                                       // load cursor: sweep, chunk (wave-uniform)
  auto load = [&](u32x4& R) {
    MNAV_GLOBAL const u32x4* const p = st0 + ((size_t)((s_ld + first_order) & 3u) * nch + min(c_ld, last)) * 16u;
    R = *p;
    asm volatile("" ::: "memory");                                    // pins the load here: written plainly, the compiler sinks it to its use three chunks later
    if (++c_ld >= max_nch) { c_ld = 0; ++s_ld; }
  };
  unsigned long long any = 0ull;
  uint32_t sweep = 0, c = 0;
  overrun = false;
  bool done = false;
  auto step = [&](u32x4& R) {
    any |= __ballot(tb_chunk(R, lane4));
    load(R);
    if (++c >= max_nch) {
      c = 0; ++sweep;
      if (any == 0ull) done = true;
      else if (sweep >= 16u * T) { overrun = true; done = true; }
      any = 0ull;
    }
  };
  // kTbDepth chunk registers in rotation (the loop body is one full rotation: a load lands in the register it is consumed from)
  u32x4 R[kTbDepth];
#pragma unroll
  for (int k = 0; k < kTbDepth; ++k) load(R[k]);
  for (;;) {
    // (the compiler waits for EVERY load in flight at the head of a loop, whatever the head needs -- seen in the ISA: vmcnt(0)
    //  there, vmcnt(kTbDepth - 1) at the other steps --, so the body holds several rotations: one drain per kTbRot * kTbDepth chunks)
#pragma unroll
    for (int k = 0; k < kTbRot * kTbDepth; ++k) {
      step(R[k % kTbDepth]);
      if (done) break;
    }
    if (done) break;
  }
  return sweep;
}

template <int T>
__global__ __launch_bounds__(64) void k_tb_solve_q(tb::Args A, int par)
{
#ifdef MNAV_TB_TIMING
  unsigned long long tt[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, t_last = __builtin_readcyclecounter();
#endif
  __shared__ __attribute__((aligned(16))) uint32_t lds[T * 64 + kTbQStride];   // [row][lane] + four staging areas of 272 bytes
  const int lane = threadIdx.x;
  const uint32_t q = (uint32_t)lane >> 4, l16 = (uint32_t)lane & 15u;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(tb::lds_u32_t)lds;
  const uint32_t lane4 = lds0 + 4u * lane;
  const uint32_t stage_q = lds0 + 4u * (T * 64) + q * kTbQStride;
  const unsigned long long qmask = 0xFFFFull << (16u * q);
  const uint32_t NP = A.NP;
  const uint32_t n_items = A.ctl->n_items;
  MNAV_GLOBAL const uint32_t* const stream = as_global(A.stream);
  uint32_t my_items = 0, my_acts = 0, my_sweeps = 0, my_wakes = 0, my_first = 0;
  // The first ticket of a wave is its own index (no atomic: a launch of 2048 waves began with 2048 atomics in a row on one word,
  // 13 us of every iteration); the counter hands out the tickets behind the grid's.
  bool first_ticket = true;
  for (;;) {
    uint32_t it0 = 4u * blockIdx.x;
    if (!first_ticket) {
      if (lane == 0) it0 = 4u * gridDim.x + atomicAdd(&A.ctl->next_item, 4u);
      it0 = tb::rfl(it0);
    }
    first_ticket = false;
    if (it0 >= n_items) break;
    // a quarter beyond the last item shadows it and stores nothing
    const u32x2 item = ((MNAV_GLOBAL const u32x2*)as_global(A.items))[min(it0 + q, n_items - 1u)];
    const uint32_t t = item.x, start = item.y & 0xFFFFu, count = (it0 + q < n_items) ? (item.y >> 16) : 0u;
    TbTile W;
    {
      MNAV_GLOBAL const u32x4* hp = (MNAV_GLOBAL const u32x4*)as_global(A.tiles) + 4u * (size_t)t;
      const u32x4 h0 = hp[0], h1 = hp[1], h2 = hp[2];
      W.soff = h0.x; W.sl = h0.y; W.nv = h0.z; W.nh = h0.w;
      W.sweep_off = h1.x; W.sweep_chunks = h1.y; W.pre_off = h1.z; W.pre_chunks = h1.w;
      W.post_off = h2.x; W.post_chunks = h2.y; W.exp_off = h2.z; W.exp_n = h2.w;
    }
    const uint32_t max_sweep = tb::wmax4(W.sweep_chunks), max_pre = tb::wmax4(W.pre_chunks), max_post = tb::wmax4(W.post_chunks),
                   max_exp = tb::wmax4(W.exp_n);
    ++my_items; my_acts += (l16 == 0u) ? count : 0u;
    TB_STAMP(0);
    {
      // Every lane runs the whole item (the stream chunks are loaded 16 bytes per lane).  Lanes beyond `count` shadow the
      // last plan of their quarter's item and store nothing.
      const bool active = l16 < count;
      const uint32_t p = A.bucket[(size_t)t * NP + start + min(l16, max(count, 1u) - 1u)];
      MNAV_GLOBAL float* sl = as_global(A.D) + ((size_t)W.soff * NP + (size_t)p * W.sl);
      // ---- load the owned slots: LDS[row][lane]
      {
        MNAV_GLOBAL const u32x4* s4 = (MNAV_GLOBAL const u32x4*)sl;
        u32x4 v[T / 4];
#pragma unroll
        for (int c = 0; c < T / 4; ++c) v[c] = s4[c];
#pragma unroll
        for (int c = 0; c < T / 4; ++c) {
          tb::ldsw(lane4 + (4 * c + 0) * 256, v[c].x); tb::ldsw(lane4 + (4 * c + 1) * 256, v[c].y);
          tb::ldsw(lane4 + (4 * c + 2) * 256, v[c].z); tb::ldsw(lane4 + (4 * c + 3) * 256, v[c].w);
        }
      }
      MNAV_GLOBAL const u32x4* g4p = (MNAV_GLOBAL const u32x4*)(sl + T);
      uint32_t first_order = 0;                                       // sweep order of the first sweep (uniform per quarter)
      TB_STAMP(1);
      // ---- ghosts -> owned (the ghosts are constant during the activation)
      if (max_pre) {
        tb::QStream S; S.begin(stream, W.pre_off, W.pre_chunks, stage_q, l16);
        u32x4 G = { 0u, 0u, 0u, 0u };
        float gmin = inf_f();                                         // smallest ghost value that lowered one of this lane's vertices ...
        uint32_t gord = 0;                                            // ... and the sweep order that runs with a wave entering there
        for (uint32_t c = 0; c < max_pre; ++c) {
          const bool live = c < W.pre_chunks;
          u32x4 h[kTbBlocksPerChunk], o1[kTbBlocksPerChunk], w2[kTbBlocksPerChunk];
#pragma unroll
          for (int j = 0; j < (int)kTbBlocksPerChunk; ++j) {
            h[j] = tb::ldsr4(stage_q + 64 * j); o1[j] = tb::ldsr4(stage_q + 64 * j + 16); w2[j] = tb::ldsr4(stage_q + 64 * j + 32);
          }
          const u32x4 q3 = tb::ldsr4(stage_q + 48);                   // block 0: d12 = the chunk's ghost group, d13 = the next chunk's
          S.advance();
          if (c == 0) G = g4p[live ? q3.x : 0u];
          const u32x4 Gn = g4p[(c + 1u < W.pre_chunks) ? q3.y : 0u];  // the next chunk's ghost values
#pragma unroll
          for (int j = 0; j < (int)kTbBlocksPerChunk; ++j) {
            const uint32_t n = live ? ((h[j].x >> 8) & 7u) : 0u;
            if (n) {
              const uint32_t jj = h[j].x & 3u;
              const float g = u2f(jj == 0 ? G.x : jj == 1 ? G.y : jj == 2 ? G.z : G.w);
              const uint32_t offs[5] = { h[j].y, h[j].z, h[j].w, o1[j].x, o1[j].y };
              const uint32_t ws[5] = { o1[j].z, o1[j].w, w2[j].x, w2[j].y, w2[j].z };
              bool lowered = false;
#pragma unroll
              for (int k = 0; k < (int)kTbGhostEdges; ++k) {
                if ((uint32_t)k < n) {
                  const uint32_t a = offs[k] + lane4;
                  const uint32_t nd = f2u(g + u2f(ws[k]));
                  const uint32_t raw = tb::ldsr(a);
                  const bool low = nd < (raw & 0x7fffffffu);
                  lowered |= low;
                  tb::ldsw(a, low ? (nd | kTbDirty) : raw);
                }
              }
              if (lowered && g < gmin) { gmin = g; gord = (h[j].x >> kTbOrderShift) & 3u; }
            }
          }
          G = Gn;
        }
        // the order most lanes of the quarter ask for (lanes whose ghosts lowered nothing do not vote; no votes: order 0)
        const bool votes = active && gmin < inf_f();
        uint32_t bestc = 0;
#pragma unroll
        for (uint32_t o = 0; o < 4; ++o) {
          const uint32_t cn = (uint32_t)__popcll(__ballot(votes && gord == o) & qmask);
          if (cn > bestc) { bestc = cn; first_order = o; }
        }
      }
      TB_STAMP(2);
      // ---- Gauss-Seidel sweeps to the tile-local fixed point of every quarter (a converged quarter changes nothing any more)
      bool overrun;
      const uint32_t sweep = max_sweep ? tbq_sweeps<T>(stream, W.sweep_off, W.sweep_chunks, max_sweep, first_order, l16, lane4, overrun) : 1u;
      if (max_sweep && overrun && lane == 0) A.ctl->err = 1u;
      my_sweeps += sweep;
      TB_STAMP(3);
      // ---- write back the 16-byte chunks that hold a lowered value
      {
        MNAV_GLOBAL u32x4* s4 = (MNAV_GLOBAL u32x4*)sl;
#pragma unroll
        for (int c = 0; c < T / 4; ++c) {
          u32x4 x;
          x.x = tb::ldsr(lane4 + (4 * c + 0) * 256); x.y = tb::ldsr(lane4 + (4 * c + 1) * 256);
          x.z = tb::ldsr(lane4 + (4 * c + 2) * 256); x.w = tb::ldsr(lane4 + (4 * c + 3) * 256);
          if (active && ((x.x | x.y | x.z | x.w) & kTbDirty)) {
            x.x &= 0x7fffffffu; x.y &= 0x7fffffffu; x.z &= 0x7fffffffu; x.w &= 0x7fffffffu;
            s4[c] = x;
          }
        }
      }
      TB_STAMP(4);
      // ---- owned -> ghosts: a neighbour tile is woken when a candidate undercuts what we know of its vertex.  A wake-up is
      // three dependent memory operations (look at the pending value, atomicMin it, learn from the old value whether this is the
      // pair's first wake-up): they are pipelined over the neighbour tiles -- at tile end k the look for tile k is issued, the
      // atomic for tile k-1 (whose look has arrived), and the old value of tile k-2 is consumed -- and the first wake-ups of the
      // whole item are counted with ONE atomic at the end.  Every stage is per lane: the neighbour tile differs between the quarters.
      if (max_post) {
        tb::QStream S; S.begin(stream, W.post_off, W.post_chunks, stage_q, l16);
        u32x4 G = { 0u, 0u, 0u, 0u };
        uint32_t cand = kTbInfBits, best = kTbInfBits;
        MNAV_GLOBAL uint32_t* const pend_p = as_global(A.pend) + p;
        MNAV_GLOBAL uint8_t* const pflag_p = as_global(A.pflag) + (p >> 6);
        MNAV_GLOBAL uint32_t* const pm = as_global(A.marr[par ^ 1]) + p;
        uint32_t t2_1 = 0, best_1 = kTbInfBits, cur_1 = 0, best_2 = kTbInfBits, old_2 = 0;
        bool want_1 = false, did_2 = false;
        uint32_t n_first = 0;
        auto advance = [&](uint32_t t2_new, uint32_t best_new, bool want_new) {
          bool first = false;
          if (did_2) {
            first = old_2 == kTbInfBits;
            if (best_2 < old_2) atomicMin((uint32_t*)pm, best_2);
            ++my_wakes;
          }
          n_first += first ? 1u : 0u;
          did_2 = want_1 && best_1 < cur_1;
          best_2 = best_1;
          if (did_2) { old_2 = atomicMin((uint32_t*)(pend_p + (size_t)t2_1 * NP), best_1); pflag_p[(size_t)t2_1 * A.nblk] = 1; }
          want_1 = want_new; t2_1 = t2_new; best_1 = best_new;
          if (want_new) cur_1 = pend_p[(size_t)t2_new * NP];
        };
        for (uint32_t c = 0; c < max_post; ++c) {
          const bool live = c < W.post_chunks;
          u32x4 h[kTbBlocksPerChunk], o1[kTbBlocksPerChunk], w2[kTbBlocksPerChunk];
#pragma unroll
          for (int j = 0; j < (int)kTbBlocksPerChunk; ++j) {
            h[j] = tb::ldsr4(stage_q + 64 * j); o1[j] = tb::ldsr4(stage_q + 64 * j + 16); w2[j] = tb::ldsr4(stage_q + 64 * j + 32);
          }
          const u32x4 q3 = tb::ldsr4(stage_q + 48);
          S.advance();
          if (c == 0) G = g4p[live ? q3.x : 0u];
          const u32x4 Gn = g4p[(c + 1u < W.post_chunks) ? q3.y : 0u];
#pragma unroll
          for (int j = 0; j < (int)kTbBlocksPerChunk; ++j) {
            const uint32_t hd = live ? h[j].x : 0u, n = (hd >> 8) & 7u;
            if (n) {
              const uint32_t offs[5] = { h[j].y, h[j].z, h[j].w, o1[j].x, o1[j].y };
              const uint32_t ws[5] = { o1[j].z, o1[j].w, w2[j].x, w2[j].y, w2[j].z };
#pragma unroll
              for (int k = 0; k < (int)kTbGhostEdges; ++k)
                if ((uint32_t)k < n) cand = min(cand, f2u(fabsf(u2f(tb::ldsr(offs[k] + lane4))) + u2f(ws[k])));
              if (hd & kTbGhostEnd) {
                const uint32_t jj = hd & 3u;
                const uint32_t g = jj == 0 ? G.x : jj == 1 ? G.y : jj == 2 ? G.z : G.w;
                if (cand < g) best = min(best, cand);
                cand = kTbInfBits;
              }
              if (hd & kTbTileEnd) {
                advance(w2[j].w, best, active && best != kTbInfBits);   // d11: owner tile of the ghosts just closed
                best = kTbInfBits;
              }
            }
          }
          G = Gn;
        }
        advance(0u, kTbInfBits, false);                                // drain the two stages in flight
        advance(0u, kTbInfBits, false);
        my_first |= n_first;                                           // (per lane; one store per wave at the end, mnav_tbv.h)
      }
      TB_STAMP(5);
      // ---- export the lowered boundary values to the ghost slots that mirror them.  The records {row, soff, sl, off} are read
      // like a stream: 16 of them are one 256-byte chunk (a per-lane load of the records one after the other was a chain of
      // dependent global loads: +70 % on this phase)
      if (max_exp) {
        tb::QStream S; S.begin_at((MNAV_GLOBAL const u32x4*)as_global(A.exps) + W.exp_off, (W.exp_n + 15u) >> 4, stage_q, l16);
        for (uint32_t k0 = 0; k0 < max_exp; k0 += 16u) {
          u32x4 x[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) x[r] = tb::ldsr4(stage_q + 16 * r);
          S.advance();
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (k0 + r < W.exp_n) {
              const uint32_t v = tb::ldsr(x[r].x + lane4);
              if (active && (v & kTbDirty)) as_global(A.D)[(size_t)x[r].y * NP + ((size_t)p * x[r].z + x[r].w)] = u2f(v & 0x7fffffffu);
            }
          }
        }
      }
      TB_STAMP(6);
    }
  }
#ifdef MNAV_TB_TIMING
  if (lane == 0) for (int k = 0; k < 8; ++k) if (tt[k]) atomicAdd(&g_tb_timing[k], tt[k]);
#endif
  my_wakes = wave_sum(my_wakes); my_acts = wave_sum(my_acts);
  if (__any(my_first != 0u) && lane == 0) A.ctl->n_cand[par ^ 1] = 1u;   // "pairs are pending" (k_tb_scan)
  if (lane == 0 && my_items && blockIdx.x < A.wstat_slots) {         // (its own slot, a plain read-modify-write: see k_tb_stats)
    unsigned long long* const ws = A.wstat + 4u * blockIdx.x;
    ws[0] += (unsigned long long)my_items; ws[1] += (unsigned long long)my_acts; ws[2] += (unsigned long long)my_sweeps; ws[3] += (unsigned long long)my_wakes;
  }
}

// vertex path of a plan from the blocked distances: the walk of k_path_lazy (predecessor = argmin (dist[u] + w, dist[u], u)
// over the expanded neighbours, the minimum must BE the vertex's distance), dijkstra :358-373
__global__ __launch_bounds__(kWave) void k_tb_path(tb::Args A, const uint32_t* __restrict__ row_ptr, const Nbr* __restrict__ nbr, uint32_t V,
                                                   PlanResult* __restrict__ res, PathRows rows, uint32_t* __restrict__ mismatch)
{
  const uint32_t p = blockIdx.x;
  if (rows.skip(p)) return;
  const int lane = threadIdx.x;
  PlanResult& R = res[p];
  const uint32_t seed = A.seed[p], target = A.target[p];
  const float dt = A.D[tb::slot_addr(A.vaddr[target], A.NP, p)];
  const GoalCut gcut = goal_cut(dt, A.offset, target);
  const float goal_dist = gcut.goal;
  uint32_t code = kSuccess, n = 0, bad = 0;
  if (A.ctl->err || A.ctl->n_cand[0]) code = kInternalError;          // sweep cap hit / pairs still pending (the host stops after an odd iteration: counter 0 is its count)
  else if (!(dt < inf_f())) code = kNoPathFound;                      // the target was never reached (dijkstra :358)
  else {
    uint32_t* path = rows.row(p);                                     // written target-side first
    const uint32_t cap = rows.capacity(p);
    uint32_t v = target;
    float dv = dt;
    while (v != seed && n <= V) {
      float best_s = inf_f(), best_du = inf_f();
      uint32_t best_u = v;
      const uint32_t beg = row_ptr[v], end = row_ptr[v + 1];
      for (uint32_t i = beg + lane; i < end; i += kWave) {
        const Nbr nb = nbr[i];
        const float du = A.D[tb::slot_addr(A.vaddr[nb.u], A.NP, p)];
        if (!expanded_source(gcut, du, nb.u)) continue;               // never expanded (dijkstra :299)
        const float sm = du + nb.w;                                   // :331
        if (sm < best_s || (sm == best_s && sm < inf_f() && (du < best_du || (du == best_du && nb.u < best_u)))) { best_s = sm; best_du = du; best_u = nb.u; }
      }
#pragma unroll
      for (int o = 1; o < kWave; o <<= 1) {
        const float os = __shfl_xor(best_s, o), odu = __shfl_xor(best_du, o);
        const uint32_t ou = __shfl_xor(best_u, o);
        if (os < best_s || (os == best_s && os < inf_f() && (odu < best_du || (odu == best_du && ou < best_u)))) { best_s = os; best_du = odu; best_u = ou; }
      }
      if (f2u(best_s) != f2u(dv) || best_u == v) { bad = 1; break; }  // not a fixed point here: reported, never returned
      v = best_u; dv = best_du;
      if (lane == 0 && n < cap) path[n] = v;
      ++n;
    }
    if (bad || v != seed) code = kInternalError;
    else if (n > cap) code = kPathOverflow;                           // the host walks this plan again into a row of n ids
  }
  if (lane == 0) {
    R.code = code; R.path_len = (code == kSuccess || code == kPathOverflow) ? n : 0;
    R.steps = A.ctl->iters; R.bands = 0; R.armed = (dt < inf_f()) ? 1u : 0u; R.overflow = A.ctl->err;
    R.goal_dist = goal_dist; R.evals = 0; R.shrinks = 0;
    if (bad) atomicAdd(mismatch, 1u);
  }
}

// The vector-map entries (dijkstra :189-209) of three vertices of ONE plan, straight from the blocked distances: the predecessor
// is derived like along the path (k_tb_path) -- the first-popped expanded neighbour attaining the smallest sum, which for a vertex
// beyond goal_dist is the tentative value's predecessor of the reference -- and the vector is k_vecmap_dijkstra's arithmetic.
// What mnav_vector_at samples after a paths-only batch: no finalize pass, no V-sized output.  One wave per vertex.
__global__ __launch_bounds__(kWave) void k_tb_vector3(tb::Args A, const uint32_t* __restrict__ row_ptr, const Nbr* __restrict__ nbr,
                                                      const float* __restrict__ xyz, uint32_t p, uint3 vs, float* __restrict__ out)
{
  const int lane = threadIdx.x;
  const uint32_t v = blockIdx.x == 0 ? vs.x : blockIdx.x == 1 ? vs.y : vs.z;
  const float dt = A.D[tb::slot_addr(A.vaddr[A.target[p]], A.NP, p)];
  const GoalCut gcut = goal_cut(dt, A.offset, A.target[p]);
  float best_s = inf_f(), best_du = inf_f();
  uint32_t best_u = v;
  if (v != A.seed[p]) {
    const uint32_t beg = row_ptr[v], end = row_ptr[v + 1];
    for (uint32_t i = beg + lane; i < end; i += kWave) {
      const Nbr nb = nbr[i];
      const float du = A.D[tb::slot_addr(A.vaddr[nb.u], A.NP, p)];
      if (!expanded_source(gcut, du, nb.u)) continue;                 // never expanded (dijkstra :299)
      const float sm = du + nb.w;                                     // :331
      if (sm < best_s || (sm == best_s && sm < inf_f() && (du < best_du || (du == best_du && nb.u < best_u)))) { best_s = sm; best_du = du; best_u = nb.u; }
    }
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
      const float os = __shfl_xor(best_s, o), odu = __shfl_xor(best_du, o);
      const uint32_t ou = __shfl_xor(best_u, o);
      if (os < best_s || (os == best_s && os < inf_f() && (odu < best_du || (odu == best_du && ou < best_u)))) { best_s = os; best_du = odu; best_u = ou; }
    }
    if (!(best_s < inf_f())) best_u = v;
  }
  if (lane == 0) {
    float x = 0.f, y = 0.f, z = 0.f;
    if (best_u != v) {                                                // :197
      x = xyz[3 * (size_t)best_u] - xyz[3 * (size_t)v];               // :204
      y = xyz[3 * (size_t)best_u + 1] - xyz[3 * (size_t)v + 1];
      z = xyz[3 * (size_t)best_u + 2] - xyz[3 * (size_t)v + 2];
      const float len = sqrtf(x * x + y * y + z * z);                 // normalized(), :206
      x = x / len; y = y / len; z = z / len;
    }
    out[3 * blockIdx.x] = x; out[3 * blockIdx.x + 1] = y; out[3 * blockIdx.x + 2] = z;
  }
}

// settled vertices per plan (the popped ones: dist <= goal_dist), for the algorithmic-bytes figure: one wave per slice
__global__ __launch_bounds__(kBlock) void k_tb_count(tb::Args A, uint32_t T, PlanResult* __restrict__ res)
{
  const uint32_t t = blockIdx.x;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const TbTile W = A.tiles[t];
  for (uint32_t p = blockIdx.y * (kBlock / 64) + wid; p < A.NP; p += gridDim.y * (kBlock / 64)) {
    const float dt = A.D[tb::slot_addr(A.vaddr[A.target[p]], A.NP, p)];
    const float goal_dist = goal_cut(dt, A.offset, A.target[p]).cut;
    const float* sl = A.D + ((size_t)W.soff * A.NP + (size_t)p * W.sl);
    uint32_t c = 0;
    for (uint32_t i = lane; i < W.nv; i += 64) { const float d = sl[i]; c += (d < inf_f() && d <= goal_dist) ? 1u : 0u; }
    c = wave_sum(c);
    if (lane == 0 && c) atomicAdd(&res[p].settled, (unsigned long long)c);
  }
}

// popped potential of one plan in vertex order: the reference's value wherever it popped the vertex (dist <= goal_dist),
// +inf elsewhere (mnav_download_output what = 5)
__global__ __launch_bounds__(kBlock) void k_tb_popped(tb::Args A, uint32_t p, uint32_t V, float* __restrict__ out)
{
  const float dt = A.D[tb::slot_addr(A.vaddr[A.target[p]], A.NP, p)];
  const float goal_dist = goal_cut(dt, A.offset, A.target[p]).cut;    // (a negative offset: final wherever d <= dist[target])
  const uint32_t stride = gridDim.x * kBlock;
  for (uint32_t v = blockIdx.x * kBlock + threadIdx.x; v < V; v += stride) {
    const float d = A.D[tb::slot_addr(A.vaddr[v], A.NP, p)];
    out[v] = (d <= goal_dist) ? d : inf_f();
  }
}

__global__ __launch_bounds__(kBlock) void k_popped(const float* __restrict__ dist, uint32_t target, double offset, uint32_t V, float* __restrict__ out)
{
  const float dt = dist[target];
  const float goal_dist = goal_cut(dt, offset, target).cut;
  const uint32_t stride = gridDim.x * kBlock;
  for (uint32_t v = blockIdx.x * kBlock + threadIdx.x; v < V; v += stride) { const float d = dist[v]; out[v] = (d <= goal_dist) ? d : inf_f(); }
}

// host-side state of the engine (device arrays of the mesh-dependent streams, and of the running batch)
struct TbState {
  bool built = false, w_valid = false;
  uint32_t T = 120, ntiles = 0, max_nh = 0;   // 120 rows x 256 B + staging = 31 232 B of LDS: five waves per CU (128 rows: four)
  uint64_t S = 0;                       // words per plan
  size_t nrec = 0, nexp = 0;
  std::vector<uint32_t> vert_tile;      // host copy: plans are ordered by the tile of their wave source
  uint32_t* d_verts = nullptr;          // tile order -> vertex id
  TbTile* d_tiles = nullptr; uint32_t* d_stream = nullptr; uint32_t* d_wsrc = nullptr; TbExp* d_exps = nullptr;
  uint32_t *d_vstream = nullptr, *d_vwsrc = nullptr, *d_vtile = nullptr, *d_vgroups = nullptr; TbvExp* d_vexps = nullptr; size_t nvrec = 0;   // the streams in the V layout (k_tbv_solve, mnav_tbv.h)
  int kernel = 0;                       // solve kernel of the running batch: 0 = k_tb_solve_q (quarters, distances in LDS), 1 = k_tbv_solve (waves, distances in registers)
  uint2* d_vaddr = nullptr; uint32_t* d_vert_tile = nullptr;
  // finalize tables (mnav_tb_finalize.h)
  uint16_t* d_fin_src = nullptr; uint32_t* d_fin_wsrc = nullptr; float* d_fin_w = nullptr; TbFinOvf* d_fin_ovf = nullptr; uint32_t* d_fin_ovf_wsrc = nullptr; float* d_fin_ovf_w = nullptr;
  uint32_t* d_ghost_gid = nullptr; uint32_t* d_fin_order = nullptr; unsigned long long* wstat = nullptr; struct FinRec* d_recs = nullptr; size_t fin_n = 0, fin_novf = 0; bool fin_w_valid = false; uint32_t max_sl = 0;
  // batch state, sized for cap_np plans
  uint32_t cap_np = 0;
  float* D = nullptr; uint32_t* pend = nullptr; uint8_t* pflag = nullptr; uint32_t* pairs = nullptr; uint16_t* bucket = nullptr; uint32_t* bcnt = nullptr; uint2* items = nullptr;
  tb::Ctl* ctl = nullptr; tb::Ctl* h_ctl = nullptr;
  uint32_t* marr[2] = { nullptr, nullptr };
  float *thr = nullptr, *bnd = nullptr; uint32_t *seed = nullptr, *target = nullptr;
  uint32_t min_batch = 48;              // auto engine: batches of at least this many plans (and of tiles / 1000: mnav.hip dijkstra_impl)
  float band_mult = 2.0f;               // band = band_mult * mean edge weight * sqrt(T)  (measured on C2: 1 -> 236 ms, 2 -> 218 ms per 5120 plans)
  int iters_per_replay = 16, waves_per_cu = 0;
  hipGraphExec_t graph[2] = { nullptr, nullptr }; tb::Args graph_args[2]{};   // one per distance buffer
  // second distance buffer, filled with +inf on its own stream behind the previous call (the fill of 6 B x slots x plans is
  // otherwise 2 % of a batch); only when both fit comfortably
  float* D2 = nullptr; bool d2_clean = false; uint32_t d2_clean_np = 0, d2_wanted_np = 0; hipStream_t fill_stream = nullptr; hipEvent_t fill_done = nullptr;
  tb::Ctl last{};                       // counters of the last batch
  bool count_pending = false;           // the settled-vertex count of the last (paths-only) batch has not been taken yet
};

}  // namespace

#ifdef MNAV_TB_TIMING
extern "C" int mnav_debug_tb_timing(unsigned long long* out)
{
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tb_timing), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
  unsigned long long z[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
  return hipMemcpyToSymbol(HIP_SYMBOL(g_tb_timing), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
