// mnav_async.h -- Dijkstra on the LDS tiles WITHOUT rounds and WITHOUT scans: an asynchronous label-correcting engine for
// single plans and small batches (engine 6, "async").  Included by mnav.hip after the tile kernels.
//
// k_tile_round pays one dependent launch per round (260-670 rounds of 22-100 us for ONE plan, whatever the batch); a real
// makePlan call is one plan (mbf_mesh_nav/src/mesh_planner_execution.cpp:55-66).  Here a set of resident workgroups serves a
// TICKET QUEUE of woken tiles: a workgroup takes the next ticket (plan, tile), solves that tile in LDS exactly like
// k_tile_round (same staging, same sweeps; dijkstra :287-348 reformulated as the label-correcting fixed point, DESIGN.md
// 3.1), publishes the lowered distances, wakes the neighbour tiles it undercut -- the first waker of a tile files its
// ticket -- and takes the next ticket.  No round, no barrier between workgroups, no host round trip, no scan over the
// tiles: the critical path is the chain of tile solves along the shortest path.  The fixed point does not depend on the
// order, so the results are the bits of every other engine.
//
// Round 4's first version of this engine found its work by scanning all wake-up values twice per attempt (48 workgroups
// per plan, 4 500 failed claims per plan: 6.1 ms for a 1M-vertex plan against 7.4 ms for the rounds, first hardware run of
// round 5); the queue replaces the scans, the claims and the band.
//
// Bands.  A queue alone is not enough: served first-in-first-out with every solve run to its tile's local fixed point the wave
// races ahead on values that are corrected later, and every correction is propagated again -- the first hardware run of the
// queue took 128 workgroups 9.7 ms for ONE 1M-vertex plan (26 ms with 32 workgroups: it is the amount of work, not the critical
// path), 124 ms for 47 plans against 49 ms on the rounds.  So the plans advance in BANDS like the rounds do (delta-stepping):
// a plan has a threshold thr; a tile woken with a value below thr gets a ticket at once (ACTIVE), one woken at or beyond thr is
// PARKED on the plan's list; a solve relaxes sources below thr only (the tile parks itself for the rest).  Inside a band
// everything is asynchronous -- a tile woken by a neighbour's solve is solved as soon as a workgroup is free, not a round later
// --; when a plan's last active ticket is retired, the workgroup that retired it advances the band: thr = smallest parked
// wake-up value + band, the parked tiles below it get their tickets.  Only that one workgroup touches the plan then (nobody
// else holds a ticket of it); other plans are not involved.
//
// Protocol.  Every word another workgroup may touch is accessed with relaxed AGENT-scope atomics (sc1: past the per-CU L1
// and the per-XCD L2, MI355X_MICROARCH.md "visibility"), through GLOBAL pointers (vmcnt only).
//   pend[t]    wake-up value of tile t (float bits, inf = none): wakers atomicMin it, the solver takes it with an exchange.
//   state[t]   (the slot's second pend buffer) 0 = idle, 1 / 2 = parked on the list of band parity 0 / 1, 3 = ACTIVE: a ticket
//              is in the queue or the tile is being solved.  WHOEVER RAISES IT TO 3 FILES THE TICKET (at most one ticket per
//              tile, hence one solver per tile: two would race on tlast[t] and on the owned distances); whoever raises it from
//              below the current parked value appends the tile to the current parked list.
//              waker:   atomicMin(pend[t], v);  v < thr ?  atomicMax(state[t], 3) < 3 -> push(t)
//                                                       :  atomicMax(state[t], parked) < parked -> park(t)
//              solver:  ... solve, publish ...;  state[t] = 0;  v = pend[t];  v != inf -> as a waker, without the atomicMin
//              Both sides do "write mine, THEN look at yours", every operation awaited before the next one is issued (Dekker):
//              a wake-up that arrives while the tile is in solve is seen by the solver's look or files / parks for itself.
//   ring[p][i] ticket i of plan p = a tile id, 0xffffffff = not filed yet.  push: i = tail[p]++, ring[p][i] = tile;
//              pop: i = head[p]++, then the workgroup polls ring[p][i] (one word, with s_sleep).  A ticket index belongs to exactly
//              one popper, a slot is written exactly once per call: no reuse, no ABA.  A ring holds every ticket a plan can file
//              in practice (host: 16 per tile); a plan that runs out of slots -- or of room on a parked list -- sets abort = 5
//              and the host re-runs the call on the tile rounds.  ONE ring PER PLAN, and a workgroup serves one plan (block id
//              mod plans) until that plan is finished, then the next unfinished one: with one ring for the whole call its head
//              and tail words took every atomic of every plan -- 47 plans ran 8x slower per plan than one (round 5, first runs).
//   work[p]    per plan: tickets filed and not yet retired, counted BEFORE the ticket becomes visible and given back after
//              the solver's own wake-ups.  The workgroup that takes it to 0 advances the band (holding a count of its own
//              while it files the next band's tickets); no parked tile left <=> the plan is at its fixed point: that workgroup
//              publishes the plan record (plan_finish) and counts the plan in done_plans; done_plans == n ends the kernel.
//   order      a solver's distance stores are drained (s_waitcnt vmcnt(0) in every storing wave, then a barrier) before its
//              first wake-up; a later solver reads distances only after its ticket arrived.
// Tiles whose wake-up value lies beyond the running bound dist[target] + offset can never propagate (dijkstra :293-300):
// their ticket is retired without a solve (the band advance hands such parked tiles a ticket for exactly that).  No workgroup
// waits FOR another one -- a popper whose ticket is never filed leaves with done_plans == n --, every ticket and every poll
// looks at the 100 MHz wall clock (abort = 2) and at mnav_cancel's word (abort = 3), so residency is not a correctness
// condition: a grid larger than what fits only starts later.
#pragma once

constexpr uint32_t kAsyncWake = 32;        // distinct neighbour tiles one solve can wake through the LDS table (more: slow path)
constexpr uint32_t kTicketNone = 0xFFFFFFFFu;
constexpr uint32_t kTicketExit = 0xFFFFFFFEu;
// words 4.. of the context's control line (word 0: mnav_cancel)
struct AsyncCtl { uint32_t abort; uint32_t done_plans; uint32_t tickets; uint32_t switches; uint32_t polls; uint32_t dropped; uint32_t ring_cap; uint32_t pad; };   // ring_cap: slots per plan

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
__device__ __forceinline__ uint32_t amax(uint32_t* p, uint32_t v) { return __hip_atomic_fetch_max((gptr)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// the per-plan words live in the slot's three TCnt records (48 bytes, unused by this engine otherwise)
struct PlanWords { uint32_t work, acts, sweeps, epochs, thr, par, nparked[2], head, tail, done, pad; };
static_assert(sizeof(PlanWords) == 3 * sizeof(TCnt), "PlanWords overlays TilePlan.cnt[3]");
__device__ __forceinline__ PlanWords* words_of(const TilePlan& P) { return reinterpret_cast<PlanWords*>(P.cnt.p); }
constexpr uint32_t kActive = 3u;
constexpr uint32_t kParkedLists = 4u;    // capacity of a parked list in tiles-of-the-mesh (a tile can be parked, promoted, solved and parked again within a band)

// the plan reached its fixed point: publish the control record the finalize pass / the path walk read (after the kernel)
__device__ __forceinline__ void plan_finish(const TilePlan& P, AsyncCtl* actl)
{
  PlanWords* const W = words_of(P);
  TCtl c; memset(&c, 0, sizeof(c));
  c.acts = ld(&W->acts); c.sweeps = ld(&W->sweeps);
  c.it = (int32_t)c.acts; c.done = 1u; c.thr = inf_f(); c.thr_prev = inf_f();
  P.ctl[0] = c; P.ctl[1] = c;
  drain();
  st(&W->done, 1u);                                                   // (its workgroups move on to another plan)
  add(&actl->tickets, ld(&W->tail));
  add(&actl->done_plans, 1u);
}
// file the ticket of tile t of plan p (the caller has just raised state[t] to ACTIVE)
// `filed`: the caller holds a RESERVATION in work[p] (it added more than it can file before its first wake-up and gives back what it
// did not use: two dependent memory round trips less on the path from a solve to the next one) and counts its tickets here;
// null: the ticket is counted on the spot, before it can be seen.
__device__ __forceinline__ void push(const TilePlan& P, uint32_t p, uint32_t t, uint32_t* ring, AsyncCtl* actl, uint32_t* filed)
{
  if (filed) atomicAdd(filed, 1u);                                    // (LDS)
  else { add(&words_of(P)->work, 1u); drain(); }
  const uint32_t i = add(&words_of(P)->tail, 1u);
  if (i < actl->ring_cap) st(ring + (size_t)p * actl->ring_cap + i, t);
  else st(&actl->abort, 5u);                                          // out of slots: the host re-runs the call on the tile rounds
}
// put tile t on the parked list of band parity `par` (the caller has just raised state[t] to that list's parked value)
__device__ __forceinline__ void park(const TilePlan& P, uint32_t t, uint32_t par, AsyncCtl* actl)
{
  const uint32_t cap = kParkedLists * P.ntiles;
  const uint32_t i = add(&words_of(P)->nparked[par], 1u);
  if (i < cap) st(P.parked + (size_t)par * cap + i, t);
  else st(&actl->abort, 5u);
}
// tile t2 has the pending value v (already merged into pend[t2]): ticket or parked list, unless somebody else saw to it
__device__ __forceinline__ void route(const TilePlan& P, uint32_t p, uint32_t t2, uint32_t v, float thr, uint32_t par, uint32_t* ring, AsyncCtl* actl, uint32_t* filed)
{
  if (u2f(v) < thr) { if (amax(P.pend[1] + t2, kActive) < kActive) push(P, p, t2, ring, actl, filed); }
  else {
    // (a tile still in the OTHER parity's parked state is on the list the band advance is working through: with pk above that value
    //  this call takes it over -- the advance then finds the state changed and skips it --, with pk below it the advance moves it)
    const uint32_t pk = 1u + par;
    if (amax(P.pend[1] + t2, pk) < pk) park(P, t2, par, actl);
  }
}
// wake tile t2 with value v (float bits)
__device__ __forceinline__ void wake(const TilePlan& P, uint32_t p, uint32_t t2, uint32_t v, float thr, uint32_t par, uint32_t* ring, AsyncCtl* actl, uint32_t* filed)
{
  (void)amin(P.pend[0] + t2, v);
  drain();                                                            // merged BEFORE the state word is touched (Dekker, see the header)
  route(P, p, t2, v, thr, par, ring, actl, filed);
}
}  // namespace aq

// work = 1, the seed's tile ACTIVE (ticket p), every other tile idle, first band [0, band); the control words; grid (tiles / 256, n)
__global__ __launch_bounds__(kBlock) void k_async_init(const TilePlan* __restrict__ plans, uint32_t* __restrict__ ring, const uint32_t* __restrict__ vert_tile,
                                                       AsyncCtl* __restrict__ actl, uint32_t n, uint32_t ring_cap)
{
  const TilePlan& P = plans[blockIdx.y];
  const uint32_t st = vert_tile[P.seed];
  const uint32_t stride = gridDim.x * kBlock;
  for (uint32_t t = blockIdx.x * kBlock + threadIdx.x; t < P.ntiles; t += stride) P.pend[1][t] = (t == st) ? aq::kActive : 0u;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    aq::PlanWords W; memset(&W, 0, sizeof(W));
    W.work = 1u; W.tail = 1u;
    W.thr = f2u((P.band > 0.f && P.band < inf_f()) ? P.band : inf_f());
    *aq::words_of(P) = W;
    ring[(size_t)blockIdx.y * ring_cap] = st;
    if (blockIdx.y == 0) {
      AsyncCtl c; c.abort = 0u; c.done_plans = 0u; c.tickets = 0u; c.switches = 0u; c.polls = 0u; c.dropped = 0u; c.ring_cap = ring_cap; c.pad = 0u;
      *actl = c;
    }
  }
}

template <int VPT>   // owned vertices per thread: tile_size <= VPT * 256
__global__ __launch_bounds__(kTileBlock) void k_plan_async(const TilePlan* __restrict__ plans, uint32_t n, AsyncCtl* __restrict__ actl,
                                                           uint32_t* __restrict__ ring, unsigned long long limit_ticks)
{
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  __shared__ uint32_t s_hdr[8];
  __shared__ uint32_t s_nq[3];
  __shared__ uint32_t s_ticket, s_plan, s_bound_bits, s_thr_bits, s_par, s_wover, s_solve, s_advance, s_filed;
  __shared__ uint32_t s_wtile[kAsyncWake], s_wval[kAsyncWake];
  __shared__ uint32_t s_min[kTileBlock / 64], s_bey[kTileBlock / 64];
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const TileLds L = tile_lds_layout(smem, plans[0].max_nv, plans[0].max_nh, plans[0].max_ne);   // one mesh: the same for every plan
  uint32_t* const ldu = L.ldu; uint32_t* const lh0 = L.lh0; uint16_t* const q0 = L.q0;
  const unsigned long long t_begin = wall_clock64();
  uint32_t my_polls = 0, my_dropped = 0, my_switches = 0;
  uint32_t p_cur = blockIdx.x % n;                                    // the plan this workgroup serves (thread 0's copy decides)
  for (;;) {
    // ---- the next ticket of the plan this workgroup serves (thread 0 takes and awaits it, the workgroup sits in the barrier);
    // when that plan is finished: the next unfinished one
    if (tid == 0) {
      uint32_t e = kTicketExit;
      for (bool again = true; again;) {
        again = false;
        aq::PlanWords* const Wc = aq::words_of(plans[p_cur]);
        uint32_t* const ring_p = ring + (size_t)p_cur * actl->ring_cap;
        // (three independent operations in flight together: a workgroup that always finds its ticket filed never enters the poll
        //  loop below, and must still see an abort or mnav_cancel)
        const uint32_t ab = aq::ld(&actl->abort);
        const uint32_t cn = plans[0].cancel ? aq::ld(plans[0].cancel) : 0u;
        const uint32_t i = aq::add(&Wc->head, 1u);
        if (ab) { }
        else if (cn) aq::st(&actl->abort, 3u);                        // mnav_cancel, dijkstra :287
        else if (wall_clock64() - t_begin > limit_ticks) aq::st(&actl->abort, 2u);
        else if (i < actl->ring_cap) {
          for (uint32_t spins = 0;; ++spins) {
            e = aq::ld(ring_p + i);
            if (e != kTicketNone) break;
            if ((spins & 7u) == 7u) {                                 // leave?  (every 8th look)
              e = kTicketExit;
              if (aq::ld(&actl->abort) || aq::ld(&actl->done_plans) >= n) break;
              if (plans[0].cancel && aq::ld(plans[0].cancel)) { aq::st(&actl->abort, 3u); break; }   // mnav_cancel, dijkstra :287
              if (wall_clock64() - t_begin > limit_ticks) { aq::st(&actl->abort, 2u); break; }
              if (spins > 400000000u) { aq::st(&actl->abort, 4u); break; }                            // second guard, should the clock not tick
              if (aq::ld(&Wc->done)) {                                // this plan is finished: serve the next unfinished one
                uint32_t q = p_cur;
                for (uint32_t k = 1; k < n; ++k) { const uint32_t c2 = (p_cur + k) % n; if (!aq::ld(&aq::words_of(plans[c2])->done)) { q = c2; break; } }
                if (q != p_cur) { p_cur = q; ++my_switches; again = true; }
                break;                                                // (no unfinished plan left: done_plans reaches n in a moment; e = exit)
              }
              e = kTicketNone;
            }
            ++my_polls;
            if (spins < 64u) __builtin_amdgcn_s_sleep(2); else __builtin_amdgcn_s_sleep(16);
          }
        } else aq::st(&actl->abort, 5u);
      }
      s_ticket = e; s_plan = p_cur;
    }
    __syncthreads();
    const uint32_t ticket = s_ticket;
    if (ticket == kTicketExit) break;
    const uint32_t p = s_plan, t = ticket;
    const TilePlan& P = plans[p];
    aq::PlanWords* const W = aq::words_of(P);
    uint32_t* const pend = P.pend[0];
    uint32_t* const state = P.pend[1];
    uint32_t* const dbits = reinterpret_cast<uint32_t*>(P.dist.p);
    if (tid == 0) {
      // Everything the decision and the solve's header need goes out together -- ONE round trip, then the wake-up value is taken.
      // (Until round 6: the exchange, the target's distance, thr, par and the seven header words one after the other, each waited
      // for before the next was issued -- the header words through generic pointers, which the compiler keeps in order with the LDS
      // stores between them: a dozen round trips per ticket on a path that is nothing but round trips.)
      uint32_t* const tl = reinterpret_cast<uint32_t*>(P.tlast.p) + t;
      const uint32_t dtb = aq::ld(dbits + P.target);
      const uint32_t thr_b = aq::ld(&W->thr), par_b = aq::ld(&W->par);   // (constant while anybody holds a ticket of the plan)
      const uint32_t tl_b = aq::ld(tl);                                  // (written by the tile's previous solver before it cleared the state)
      const uint32_t h0w = P.vptr[t], h1w = P.vptr[t + 1], h2w = P.hptr[t], h3w = P.hptr[t + 1], h4w = P.eptr[t], h5w = P.eptr[t + 1], h6w = P.rptr[t];
      const uint32_t v = aq::xchg(pend + t, kInfBits);
      const float bound = (float)((double)u2f(dtb) + fmax(P.offset, 0.0));   // >= the final goal_dist (dijkstra :296); negative offsets: goal_cut
      s_bound_bits = f2u(bound);
      s_thr_bits = thr_b; s_par = par_b;
      uint32_t solve = 0u;
      if (v != kInfBits) {
        if (u2f(v) > bound) {                                         // can never propagate any more (k_tile_round does the same)
          if (!(u2f(tl_b) > -inf_f())) aq::st(tl, f2u(-3.0e38f));         // the finalize pass still has to visit the tile
          ++my_dropped;
        } else {
          solve = 1u;
          s_hdr[0] = h0w; s_hdr[1] = h1w; s_hdr[2] = h2w; s_hdr[3] = h3w; s_hdr[4] = h4w; s_hdr[5] = h5w; s_hdr[6] = h6w;
          s_hdr[7] = tl_b;
          s_nq[0] = 0; s_nq[1] = 0; s_nq[2] = 0;
        }
      }
      s_solve = solve; s_wover = 0u; s_advance = 0u; s_filed = 0u;
      if (solve) (void)aq::add(&W->work, kAsyncWake + 1u);             // reservation for the tickets this solve may file (given back at the retire); nobody waits for it
    }
    if (tid < (int)kAsyncWake) { s_wtile[tid] = kNone; s_wval[tid] = kInfBits; }
    __syncthreads();
    uint32_t sweep = 0;
    const float thr = u2f(s_thr_bits);
    const uint32_t par = s_par;
    if (s_solve) {
      // ---- solve tile t (k_tile_round's solve; distances through agent-scope loads / stores)
      const float bound = u2f(s_bound_bits);
      const uint32_t v0 = s_hdr[0], nv = s_hdr[1] - v0;
      const uint32_t h0 = s_hdr[2], nh = s_hdr[3] - h0;
      const uint32_t e0 = s_hdr[4], ne = s_hdr[5] - e0;
      const uint32_t r0 = s_hdr[6];
      const uint32_t nl = nv + nh;
      const float tl = u2f(s_hdr[7]);
      MNAV_GLOBAL const uint32_t* g_verts = as_global(P.verts);
      MNAV_GLOBAL const uint32_t* g_halo_verts = as_global(P.halo_verts);
      MNAV_GLOBAL const uint32_t* g_halo_tile = as_global(P.halo_tile);
      // The ids, then the distances: every load unconditional at a clamped index, so that all of a level are in flight together
      // (a load under `if (i < nv)` is waited for before the next branch: six ids and six distances were twelve round trips).
      const uint32_t nvm = nv ? nv - 1u : 0u, nhm = nh ? nh - 1u : 0u;   // (an empty halo reads the slack behind the array: dev_upload)
      uint32_t gi[VPT];
#pragma unroll
      for (int k2 = 0; k2 < VPT; ++k2) gi[k2] = g_verts[v0 + min((uint32_t)tid + k2 * kTileBlock, nvm)];
      uint32_t hi[2];
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) hi[k2] = g_halo_verts[h0 + min((uint32_t)tid + k2 * kTileBlock, nhm)];
      stage_tile_graph(P, L, e0, ne, r0, nl, tid);
      uint32_t orig[VPT], hb[2];
#pragma unroll
      for (int k2 = 0; k2 < VPT; ++k2) orig[k2] = aq::ld(dbits + gi[k2]);
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) hb[k2] = aq::ld(dbits + (nh ? hi[k2] : gi[0]));
#pragma unroll
      for (int k2 = 0; k2 < VPT; ++k2) {
        const uint32_t i = tid + k2 * kTileBlock;
        if (i < nv) {
          const float d = u2f(orig[k2]);
          ldu[i] = orig[k2];
          if (d < thr && d <= bound && !(d < tl)) q0[atomicAdd(&s_nq[0], 1u)] = (uint16_t)i;       // owned sources in [tlast, thr)
        }
      }
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const uint32_t i = tid + k2 * kTileBlock;
        if (i < nh) {
          const uint32_t b = hb[k2]; const float d = u2f(b);
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
      sweep = tile_sweeps(L, nv, thr, bound, s_nq, tid);
      // ---- publish: the lowered owned distances first ...
      uint32_t own_left = kInfBits;
#pragma unroll
      for (int k2 = 0; k2 < VPT; ++k2) {
        const uint32_t i = tid + k2 * kTileBlock;
        if (i < nv) {
          const uint32_t db = ldu[i];
          if (db != orig[k2]) aq::st(dbits + gi[k2], db);
          const float d = u2f(db);
          if (!(d < thr) && d <= bound) own_left = min(own_left, db);   // owned values that still have to propagate: the tile parks itself for them
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
        if (b < lh0[i]) collect(g_halo_tile[h0 + i], b);              // we undercut a neighbour's vertex: it has to re-derive it
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) own_left = min(own_left, (uint32_t)__shfl_xor((int)own_left, o));
      if (lane == 0 && own_left != kInfBits) collect(t, own_left);
      aq::drain();                                                     // every storing wave: its distance stores have left the CU ...
      __syncthreads();                                                 // ... before the first wake-up can be seen
      if (tid < (int)kAsyncWake && s_wtile[tid] != kNone) aq::wake(P, p, s_wtile[tid], s_wval[tid], thr, par, ring, actl, &s_filed);
      if (s_wover) {                                                   // more neighbour tiles than table slots: one wake-up per halo vertex
        for (uint32_t i = tid; i < nh; i += kTileBlock) {
          const uint32_t b = ldu[nv + i];
          if (b < lh0[i]) aq::wake(P, p, g_halo_tile[h0 + i], b, thr, par, ring, actl, nullptr);   // (beyond the reservation: counted one by one)
        }
        if (lane == 0 && own_left != kInfBits) aq::wake(P, p, t, own_left, thr, par, ring, actl, nullptr);   // (own_left: this wave's minimum)
      }
      aq::drain();
      __syncthreads();                                                 // every wake-up has returned
    }
    // ---- retire the ticket: the tile is ours until state[t] is cleared; a wake-up that came in meanwhile is routed here
    if (tid == 0) {
      if (s_solve) {
        aq::st(reinterpret_cast<uint32_t*>(P.tlast.p) + t, f2u(thr));
        aq::add(&W->acts, 1u); aq::add(&W->sweeps, sweep);
        aq::drain();                                                   // the next solver of t reads tlast after its ticket arrived
      }
      (void)aq::xchg(state + t, 0u);
      aq::drain();                                                     // cleared BEFORE the look (Dekker, see the header)
      const uint32_t v2 = aq::ld(pend + t);
      if (v2 != kInfBits) aq::route(P, p, t, v2, thr, par, ring, actl, s_solve ? &s_filed : nullptr);
      aq::drain();
      // give back this ticket's count and what is left of the reservation; taking the count to zero: the plan's last ticket, nobody
      // else touches the plan now
      const uint32_t back = 1u + (s_solve ? kAsyncWake + 1u - s_filed : 0u);
      if (aq::sub(&W->work, back) == back) s_advance = 1u;
    }
    __syncthreads();
    // ---- advance the band (the whole workgroup; exclusive until its first ticket is filed: no ticket of the plan is out)
    uint32_t par_c = par;
    while (s_advance) {
      const uint32_t cap = aq::kParkedLists * P.ntiles;
      const uint32_t pk_old = 1u + par_c, par2 = par_c ^ 1u, pk_new = 1u + par2;
      uint32_t* const list = P.parked + (size_t)par_c * cap;
      const uint32_t cnt = min(aq::ld(&W->nparked[par_c]), cap);
      const float bound = (float)((double)u2f(aq::ld(dbits + P.target)) + fmax(P.offset, 0.0));
      // pass 1: the smallest parked wake-up value that can still propagate
      uint32_t mn = kInfBits, beyond = 0u;
      for (uint32_t i = tid; i < cnt; i += kTileBlock) {
        const uint32_t t2 = aq::ld(list + i);
        if (aq::ld(state + t2) != pk_old) continue;                    // promoted since it was listed (a tile parked again has a second entry)
        const uint32_t v = aq::ld(pend + t2);
        if (u2f(v) > bound) beyond = 1u; else mn = min(mn, v);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { mn = min(mn, (uint32_t)__shfl_xor((int)mn, o)); beyond |= (uint32_t)__shfl_xor((int)beyond, o); }
      if (lane == 0) { s_min[wid] = mn; s_bey[wid] = beyond; }
      __syncthreads();
      mn = min(min(s_min[0], s_min[1]), min(s_min[2], s_min[3]));
      beyond = s_bey[0] | s_bey[1] | s_bey[2] | s_bey[3];
      __syncthreads();
      if (mn == kInfBits && !beyond) {                                 // nothing parked: the plan is at its fixed point
        if (tid == 0) { aq::plan_finish(P, actl); s_advance = 0u; }
        __syncthreads();
        break;
      }
      // the next band [m, m + band); parked tiles beyond the bound get a ticket too (it is retired without a solve, tlast marked)
      float thr2 = inf_f();
      if (mn != kInfBits && P.band > 0.f && P.band < inf_f()) { const float m = u2f(mn); thr2 = m + P.band; if (!(thr2 > m)) thr2 = next_up(m); }
      if (tid == 0) {
        s_filed = 0u;
        aq::st(&W->work, 1u + cnt);                                    // this workgroup's own hold + a reservation for every ticket the pass may file
        aq::st(&W->thr, f2u(thr2)); aq::st(&W->par, par2); aq::st(&W->nparked[par2], 0u);
        aq::add(&W->epochs, 1u);
        aq::drain();
      }
      __syncthreads();
      // pass 2: tickets for the parked tiles of the new band, the others move to the new list.  The first ticket filed ends the
      // exclusivity: wakers of the new band route tiles concurrently (also tiles still in the old parked state: see route()).
      for (uint32_t i = tid; i < cnt; i += kTileBlock) {
        const uint32_t t2 = aq::ld(list + i);
        if (aq::ld(state + t2) != pk_old) continue;                    // (also: the second entry of a tile this pass has handled already)
        const uint32_t v = aq::ld(pend + t2);
        if (u2f(v) < thr2 || u2f(v) > bound) { if (aq::amax(state + t2, aq::kActive) < aq::kActive) aq::push(P, p, t2, ring, actl, &s_filed); }
        else if (aq::cas(state + t2, pk_old, pk_new)) aq::park(P, t2, par2, actl);   // (failed: a waker has routed it meanwhile)
      }
      aq::drain();
      __syncthreads();
      if (tid == 0) { const uint32_t back = 1u + cnt - s_filed; s_advance = (aq::sub(&W->work, back) == back) ? 1u : 0u; }   // the band's tickets are retired already (or none was filed): once more
      __syncthreads();
      par_c = par2;
    }
  }
  if (tid == 0 && (my_polls | my_dropped | my_switches)) { atomicAdd(&actl->polls, my_polls); atomicAdd(&actl->dropped, my_dropped); atomicAdd(&actl->switches, my_switches); }
}
