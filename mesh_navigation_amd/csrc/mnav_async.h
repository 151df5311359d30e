// mnav_async.h -- Dijkstra on the LDS tiles WITHOUT rounds: an asynchronous label-correcting engine for single plans
// and small batches (engine 6, "async"; opt-in).  Included by mnav.hip after the tile kernels.
//
// k_tile_round pays one dependent launch per round (260-670 rounds of 22-100 us for one plan on the 1M mesh, whatever
// the batch), k_plan_persistent gives a plan ONE workgroup.  Here a set of resident workgroups shares the plans of a call:
// a workgroup scans a plan's wake-up values, claims one tile of the current band [m, m + band) -- any of them, the
// label-correcting fixed point does not depend on the order (dijkstra :287-348 reformulated) -- solves it
// in LDS exactly like k_tile_round (same staging, same sweeps), publishes the lowered distances and wakes the neighbour
// tiles it undercut.  No round, no barrier between workgroups, no host round trip: the critical path is the chain of
// tile solves along the shortest path.
//
// Protocol (every word another workgroup may read or write is accessed with relaxed AGENT-scope atomics = sc1 loads /
// stores / RMWs, which bypass the per-CU L1 and the per-XCD L2: MI355X_MICROARCH.md, visibility):
//   pend[t]   wake-up value of tile t (float bits, inf = none).  Wakers atomicMin it; the solver takes it with an exchange.
//   lock[t]   (the second pend buffer of the slot) inf = free, 0 = a workgroup is solving t.  One solver per tile: two
//             concurrent solves would race on tlast[t] (the "owned sources below it have been propagated" mark).
//   work      per plan: number of tiles that are pending or being solved, never below the true count: a waker adds 1
//             BEFORE its atomicMin and takes it back when the tile was pending already; the solver's 1 is the pending 1
//             it took over and is given back after its own wake-ups.  work == 0 <=> the plan is at its fixed point.
//   order     a solver's distance stores are drained (s_waitcnt vmcnt(0) in every storing wave, then a barrier) before
//             the first wake-up is posted; a later solver reads distances only after its claim returned.
// Tiles whose wake-up value lies beyond the running bound dist[target] + offset can never propagate (dijkstra :293-300):
// whoever sees one takes it out of the count.
// Every wait is bounded: workgroups never wait FOR each other (an idle one re-scans, sleeps, and gives up after
// `limit_ticks` of the 100 MHz wall clock with abort = 2); a grid larger than what is resident only starts later.
#pragma once

constexpr uint32_t kAsyncWake = 32;        // distinct neighbour tiles one solve can wake through the LDS table (more: slow path)
struct AsyncCtl { uint32_t abort; uint32_t done_plans; uint32_t claim_fails; uint32_t idle_passes; };

namespace aq {
typedef MNAV_GLOBAL uint32_t* gptr;    // every shared word is accessed through a GLOBAL pointer (global_* instructions, vmcnt only), never flat
__device__ __forceinline__ uint32_t ld(const uint32_t* p) { return __hip_atomic_load((MNAV_GLOBAL const uint32_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st(uint32_t* p, uint32_t v) { __hip_atomic_store((gptr)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t add(uint32_t* p, uint32_t v) { return __hip_atomic_fetch_add((gptr)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t sub(uint32_t* p, uint32_t v) { return __hip_atomic_fetch_sub((gptr)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t amin(uint32_t* p, uint32_t v) { return __hip_atomic_fetch_min((gptr)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t xchg(uint32_t* p, uint32_t v) { return __hip_atomic_exchange((gptr)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool cas(uint32_t* p, uint32_t expect, uint32_t v)
{
  return __hip_atomic_compare_exchange_strong((gptr)p, &expect, v, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// the per-plan words live in the slot's three TCnt records (unused by this engine otherwise)
__device__ __forceinline__ uint32_t* work_of(const TilePlan& P) { return &P.cnt[0].minpend; }
__device__ __forceinline__ uint32_t* acts_of(const TilePlan& P) { return &P.cnt[0].acts; }
__device__ __forceinline__ uint32_t* sweeps_of(const TilePlan& P) { return &P.cnt[0].sweeps; }
__device__ __forceinline__ uint32_t* done_of(const TilePlan& P) { return &P.cnt[0].pad; }

// the plan reached its fixed point: publish the control record the finalize pass / the path walk read (after the kernel)
__device__ __forceinline__ void plan_finish(const TilePlan& P, AsyncCtl* actl)
{
  TCtl c; memset(&c, 0, sizeof(c));
  c.acts = ld(acts_of(P)); c.sweeps = ld(sweeps_of(P));
  c.it = (int32_t)c.acts; c.done = 1u; c.thr = inf_f(); c.thr_prev = inf_f();
  P.ctl[0] = c; P.ctl[1] = c;
  st(done_of(P), 1u);
  add(&actl->done_plans, 1u);
}
// one pending-or-in-solve tile less; `fin`: this thread took the count to zero and owes the plan its plan_finish (the callers
// settle that at a few places of the loop instead of inlining the record stores at every decrement)
__device__ __forceinline__ void work_dec(const TilePlan& P, bool& fin)
{
  if (sub(work_of(P), 1u) == 1u) fin = true;
}
// wake tile t2 with value v (float bits)
__device__ __forceinline__ void wake(const TilePlan& P, bool& fin, uint32_t t2, uint32_t v)
{
  add(work_of(P), 1u);
  if (amin(P.pend[0] + t2, v) != kInfBits) work_dec(P, fin);        // (never the last one: the caller's own count is still held)
}
}  // namespace aq

__global__ void k_async_init(const TilePlan* __restrict__ plans, uint32_t n)
{
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const TilePlan& P = plans[p];
  TCnt z; z.minpend = 1u; z.acts = 0u; z.sweeps = 0u; z.pad = 0u;   // work = 1: the seed's tile is pending (k_tile_init)
  P.cnt[0] = z;
  z.minpend = 0u; P.cnt[1] = z; P.cnt[2] = z;
}

template <int VPT>   // owned vertices per thread: tile_size <= VPT * 256
__global__ __launch_bounds__(kTileBlock) void k_plan_async(const TilePlan* __restrict__ plans, uint32_t n, AsyncCtl* __restrict__ actl,
                                                           unsigned long long limit_ticks)
{
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  __shared__ unsigned long long s_red[kTileBlock / 64];
  __shared__ uint32_t s_hdr[8];
  __shared__ uint32_t s_nq[3];
  __shared__ uint32_t s_pick, s_flag, s_bound_bits, s_thr_bits, s_wover;
  __shared__ unsigned long long s_live;
  __shared__ uint32_t s_wtile[kAsyncWake], s_wval[kAsyncWake];
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const TileLds L = tile_lds_layout(smem, plans[0].max_nv, plans[0].max_nh, plans[0].max_ne);   // one mesh: the same for every plan
  uint32_t* const ldu = L.ldu; uint32_t* const lh0 = L.lh0; uint16_t* const q0 = L.q0;

  const unsigned long long t_begin = wall_clock64();
  uint32_t home = blockIdx.x % n, idle = 0, iter = 0, passes = 0;
  uint32_t my_fails = 0, my_idle = 0;
  for (;;) {
    // ---- leave?  (thread 0 decides, the workgroup follows) and which plans of the window [home, home + 64) still run
    if (wid == 0) {
      uint32_t live = 0;
      if ((uint32_t)lane < min(n, 64u)) { const uint32_t pi = (home + lane) % n; live = aq::ld(aq::done_of(plans[pi])) ? 0u : 1u; }
      const unsigned long long mask = __ballot(live != 0u);
      if (lane == 0) {
        uint32_t f = 0;
        if (aq::ld(&actl->abort)) f = 1u;
        else if (aq::ld(&actl->done_plans) >= n) f = 1u;
        else if (plans[0].cancel && aq::ld(plans[0].cancel)) { aq::st(&actl->abort, 3u); f = 1u; }          // mnav_cancel, dijkstra :287
        else if (wall_clock64() - t_begin > limit_ticks) { aq::st(&actl->abort, 2u); f = 1u; }
        else if (++passes > 20000000u) { aq::st(&actl->abort, 4u); f = 1u; }                       // second guard, should the clock not tick
        s_flag = f; s_live = mask;
      }
    }
    __syncthreads();
    if (s_flag) break;
    unsigned long long live = s_live;
    bool did = false;
    while (live && !did) {
      const uint32_t k = (uint32_t)__ffsll((long long)live) - 1u;
      live &= live - 1ull;
      const TilePlan& P = plans[(home + k) % n];
      uint32_t* const pend = P.pend[0];
      uint32_t* const lock = P.pend[1];
      uint32_t* const dbits = reinterpret_cast<uint32_t*>(P.dist);
      ++iter;
      bool fin = false;                                              // this thread owes P its plan_finish
      // ---- scan 1: the smallest wake-up value of the plan
      uint32_t mn = kInfBits;
      for (uint32_t t0 = tid; t0 < P.ntiles; t0 += 8 * kTileBlock) {
        uint32_t pv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const uint32_t t = t0 + u * kTileBlock; pv[u] = (t < P.ntiles) ? aq::ld(pend + t) : kInfBits; }
#pragma unroll
        for (int u = 0; u < 8; ++u) mn = min(mn, pv[u]);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mn = min(mn, (uint32_t)__shfl_xor((int)mn, o));
      __syncthreads();                                               // (s_red / s_pick of the previous attempt are dead)
      if (lane == 0) s_red[wid] = mn;
      if (tid == 0) {
        const float dt = u2f(aq::ld(dbits + P.target));
        s_bound_bits = f2u((float)((double)dt + fmax(P.offset, 0.0)));   // >= the final goal_dist (dijkstra :296); negative offsets: goal_cut
        s_pick = kNone; s_wover = 0u;
      }
      if (tid < (int)kAsyncWake) { s_wtile[tid] = kNone; s_wval[tid] = kInfBits; }
      __syncthreads();
      mn = min(min((uint32_t)s_red[0], (uint32_t)s_red[1]), min((uint32_t)s_red[2], (uint32_t)s_red[3]));
      if (mn == kInfBits) continue;                                  // nothing pending here (tiles in solve may still wake some)
      const float bound = u2f(s_bound_bits);
      const float m = u2f(mn);
      float thr = m + P.band;
      if (!(thr > m)) thr = next_up(m);
      // ---- scan 2: a tile of the band, a different one for every workgroup; tiles beyond the bound leave the count
      unsigned long long key = ~0ull;
      const uint32_t salt = (blockIdx.x + 1u) * 0x9E3779B9u + iter * 0x85EBCA6Bu;
      for (uint32_t t0 = tid; t0 < P.ntiles; t0 += 8 * kTileBlock) {
        uint32_t pv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const uint32_t t = t0 + u * kTileBlock; pv[u] = (t < P.ntiles) ? aq::ld(pend + t) : kInfBits; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (pv[u] == kInfBits) continue;
          const uint32_t t = t0 + u * kTileBlock;
          const float p = u2f(pv[u]);
          if (p > bound) {                                           // can never propagate any more (k_tile_round does the same)
            const uint32_t v = aq::xchg(pend + t, kInfBits);
            if (v == kInfBits) continue;                             // somebody else took it
            if (u2f(v) > bound) {
              uint32_t* const tl = reinterpret_cast<uint32_t*>(P.tlast) + t;
              if (!(u2f(aq::ld(tl)) > -inf_f())) aq::st(tl, f2u(-3.0e38f));   // the finalize pass still has to visit the tile
              aq::work_dec(P, fin);
            } else if (aq::amin(pend + t, v) != kInfBits) aq::work_dec(P, fin);   // lowered in between: put it back (merged with a newer wake-up)
            continue;
          }
          if (p < thr) {
            const unsigned long long h = ((unsigned long long)((t ^ salt) * 0x9E3779B1u) << 32) | t;
            key = h < key ? h : key;
          }
        }
      }
      if (fin) { aq::plan_finish(P, actl); fin = false; }           // (the last pending tile lay beyond the bound)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { const unsigned long long ok = __shfl_xor(key, o); key = ok < key ? ok : key; }
      __syncthreads();
      if (lane == 0) s_red[wid] = key;
      __syncthreads();
      if (tid == 0) {
        unsigned long long best = s_red[0];
        for (int w = 1; w < kTileBlock / 64; ++w) best = s_red[w] < best ? s_red[w] : best;
        if (best != ~0ull) {
          const uint32_t t = (uint32_t)best;
          // ---- claim: the lock first (one solver per tile), then the wake-up value
          if (aq::cas(lock + t, kInfBits, 0u)) {
            const uint32_t v = aq::xchg(pend + t, kInfBits);
            bool ok = v != kInfBits;
            if (ok && u2f(v) > bound) {                              // (re-woken with a value beyond the bound since the scan)
              uint32_t* const tl = reinterpret_cast<uint32_t*>(P.tlast) + t;
              if (!(u2f(aq::ld(tl)) > -inf_f())) aq::st(tl, f2u(-3.0e38f));
              aq::drain();
              aq::st(lock + t, kInfBits);
              aq::work_dec(P, fin);
              ok = false;
            } else if (!ok) aq::st(lock + t, kInfBits);
            if (ok) {
              const float pv = u2f(v);
              if (!(pv < thr)) { thr = pv + P.band; if (!(thr > pv)) thr = next_up(pv); }   // a wake-up is only consumed by a solve whose band holds it
              s_thr_bits = f2u(thr);
              s_hdr[0] = P.vptr[t]; s_hdr[1] = P.vptr[t + 1]; s_hdr[2] = P.hptr[t]; s_hdr[3] = P.hptr[t + 1];
              s_hdr[4] = P.eptr[t]; s_hdr[5] = P.eptr[t + 1]; s_hdr[6] = P.rptr[t];
              s_hdr[7] = aq::ld(reinterpret_cast<uint32_t*>(P.tlast) + t);
              s_nq[0] = 0; s_nq[1] = 0; s_nq[2] = 0;
              s_pick = t;
            }
          }
          if (s_pick == kNone) ++my_fails;
        }
        if (fin) { aq::plan_finish(P, actl); fin = false; }
      }
      __syncthreads();
      const uint32_t t = s_pick;
      if (t == kNone) continue;
      did = true;
      thr = u2f(s_thr_bits);
      // ---- solve tile t (k_tile_round's solve; distances through agent-scope loads / stores)
      const uint32_t v0 = s_hdr[0], nv = s_hdr[1] - v0;
      const uint32_t h0 = s_hdr[2], nh = s_hdr[3] - h0;
      const uint32_t e0 = s_hdr[4], ne = s_hdr[5] - e0;
      const uint32_t r0 = s_hdr[6];
      const uint32_t nl = nv + nh;
      const float tl = u2f(s_hdr[7]);
      MNAV_GLOBAL const uint32_t* g_verts = as_global(P.verts);
      MNAV_GLOBAL const uint32_t* g_halo_verts = as_global(P.halo_verts);
      MNAV_GLOBAL const uint32_t* g_halo_tile = as_global(P.halo_tile);
      uint32_t gi[VPT];
#pragma unroll
      for (int k2 = 0; k2 < VPT; ++k2) { const uint32_t i = tid + k2 * kTileBlock; gi[k2] = (i < nv) ? g_verts[v0 + i] : 0u; }
      uint32_t hi[2];
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) { const uint32_t i = tid + k2 * kTileBlock; hi[k2] = (i < nh) ? g_halo_verts[h0 + i] : 0u; }
      stage_tile_graph(P, L, e0, ne, r0, nl, tid);
      uint32_t orig[VPT];
#pragma unroll
      for (int k2 = 0; k2 < VPT; ++k2) {
        const uint32_t i = tid + k2 * kTileBlock;
        orig[k2] = 0u;
        if (i < nv) {
          orig[k2] = aq::ld(dbits + gi[k2]);
          const float d = u2f(orig[k2]);
          ldu[i] = orig[k2];
          if (d < thr && d <= bound && !(d < tl)) q0[atomicAdd(&s_nq[0], 1u)] = (uint16_t)i;       // owned sources in [tlast, thr)
        }
      }
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const uint32_t i = tid + k2 * kTileBlock;
        if (i < nh) {
          const uint32_t b = aq::ld(dbits + hi[k2]); const float d = u2f(b);
          ldu[nv + i] = b; lh0[i] = b;
          if (d < thr && d <= bound) q0[atomicAdd(&s_nq[0], 1u)] = (uint16_t)(nv + i);
        }
      }
      for (uint32_t i = tid + 2 * kTileBlock; i < nh; i += kTileBlock) {
        const uint32_t b = aq::ld(dbits + g_halo_verts[h0 + i]); const float d = u2f(b);
        ldu[nv + i] = b; lh0[i] = b;
        if (d < thr && d <= bound) q0[atomicAdd(&s_nq[0], 1u)] = (uint16_t)(nv + i);
      }
      __syncthreads();
      const uint32_t sweep = tile_sweeps(L, nv, thr, bound, s_nq, tid);
      // ---- publish: the lowered owned distances first ...
      uint32_t own_left = kInfBits;
#pragma unroll
      for (int k2 = 0; k2 < VPT; ++k2) {
        const uint32_t i = tid + k2 * kTileBlock;
        if (i < nv) {
          const uint32_t db = ldu[i];
          if (db != orig[k2]) aq::st(dbits + gi[k2], db);
          const float d = u2f(db);
          if (!(d < thr) && d <= bound) own_left = min(own_left, db);   // owned values that still have to propagate: the tile's own wake-up
        }
      }
      // ... the wake-ups collected per neighbour tile in LDS (a tile has a handful of neighbours, a hundred halo vertices)
      auto collect = [&](uint32_t t2, uint32_t v) {
        uint32_t slot = (t2 * 0x9E3779B1u) >> 27;
        for (uint32_t probe = 0; probe < kAsyncWake; ++probe, slot = (slot + 1u) & (kAsyncWake - 1u)) {
          const uint32_t prev = atomicCAS(&s_wtile[slot], kNone, t2);
          if (prev == kNone || prev == t2) { atomicMin(&s_wval[slot], v); return; }
        }
        s_wover = 1u;
      };
      for (uint32_t i = tid; i < nh; i += kTileBlock) {
        const uint32_t b = ldu[nv + i];
        if (b < lh0[i]) collect(g_halo_tile[h0 + i], b);            // we undercut a neighbour's vertex: it has to re-derive it
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) own_left = min(own_left, (uint32_t)__shfl_xor((int)own_left, o));
      if (lane == 0 && own_left != kInfBits) collect(t, own_left);
      aq::drain();                                                   // every storing wave: its distance stores have left the CU ...
      __syncthreads();                                               // ... before the first wake-up can be seen
      if (tid < (int)kAsyncWake && s_wtile[tid] != kNone) aq::wake(P, fin, s_wtile[tid], s_wval[tid]);
      if (s_wover) {                                                 // more neighbour tiles than table slots: one wake-up per halo vertex
        for (uint32_t i = tid; i < nh; i += kTileBlock) {
          const uint32_t b = ldu[nv + i];
          if (b < lh0[i]) aq::wake(P, fin, g_halo_tile[h0 + i], b);
        }
        if (lane == 0 && own_left != kInfBits) aq::wake(P, fin, t, own_left);   // (own_left: this wave's minimum)
      }
      __syncthreads();                                               // every wake-up has returned (its old value was looked at)
      if (tid == 0) {
        aq::st(reinterpret_cast<uint32_t*>(P.tlast) + t, f2u(thr));
        aq::add(aq::acts_of(P), 1u); aq::add(aq::sweeps_of(P), sweep);
        aq::drain();
        aq::st(lock + t, kInfBits);                                  // the next solver of t reads tlast after its claim returned
        aq::work_dec(P, fin);                                        // the count this solve held
      }
      if (fin) { aq::plan_finish(P, actl); fin = false; }            // (wake-ups never take the count to zero: only thread 0 can get here)
    }
    if (did) { idle = 0; continue; }
    // nothing to do in this window: look at the next one, back off a little (255 pollers cost the chip a third of its bandwidth)
    ++idle; ++my_idle;
    if (n > 64u) home = (home + 64u) % n;
    __builtin_amdgcn_s_sleep(32);
    if (idle > 4u) __builtin_amdgcn_s_sleep(127);
  }
  if (tid == 0 && (my_fails | my_idle)) { atomicAdd(&actl->claim_fails, my_fails); atomicAdd(&actl->idle_passes, my_idle); }
}
