// mnav.hip -- MI355X (gfx950) wavefront planner: HIP kernels + the C ABI of include/mnav.h.
//
// Hot path replaced (reference file:line):
//   DijkstraMeshPlanner::dijkstra       dijkstra_mesh_planner/src/dijkstra_mesh_planner.cpp:217-398
//   DijkstraMeshPlanner::computeVectorMap                                            :189-209
//   CVPMeshPlanner::waveFrontPropagation cvp_mesh_planner/src/cvp_mesh_planner.cpp:651-918
//   CVPMeshPlanner::waveFrontUpdate                                                  :369-556
//   CVPMeshPlanner::computeVectorMap                                                 :204-239
//   MeshMap::computeEdgeWeights          mesh_map/src/mesh_map.cpp:517-561
//
// Design: the priority-queue loops become distance bands settled by a gather
// rule iterated to its fixed point (mnav_eval.h).  One step = one launch of k_step over the
// current work list of every plan in the batch; steps are enqueued back-to-back from a
// hipGraph with no host round trip, all loop control (band advance, goal_dist arming,
// termination) is recomputed by every workgroup from the previous step's counters.
// Memory-bound irregular gather work: no MFMA, coalescing via CSR rows + 24-byte corner
// records, wave-aggregated atomics for the work lists.
//
// This file never computes a plan on the CPU: every entry point fails when no GPU is usable.
#include <hip/hip_runtime.h>

#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>
#include <unordered_map>
#include <unistd.h>

#include "../../include/mnav.h"
#include "mnav_build.h"
#include "mnav_eval.h"

using namespace mnav;

namespace {

// Pointers that reach a kernel through a struct in memory are generic to the compiler: it emits
// flat_load + s_waitcnt vmcnt(0) lgkmcnt(0) around every LDS access.  Device-memory arrays are
// therefore re-typed as address_space(1) before use (global_load, waits only on vmcnt).
#define MNAV_GLOBAL __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ MNAV_GLOBAL T* as_global(T* p) { return (MNAV_GLOBAL T*)p; }
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));   // one Nbr {u, w bits}
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // 16-byte copy unit

constexpr int kBlock = 256;   // streaming kernels (init, vector map, input preparation)
constexpr int kChunk = 96;  // step launches per graph replay; multiple of 6 (slot parities)

// ---------------------------------------------------------------------------------------------
// Step kernel.  One wave (64 lanes) per workgroup, 8 lanes cooperate on one work-list entry:
// the lanes of a group fetch the CSR row / the corner records of the vertex in parallel, the
// gather rule of mnav_eval.h is then evaluated with in-group shuffles, and list pushes are
// aggregated per wave (one atomicAdd per wave and push round).  The serial rules in mnav_eval.h
// (eval_dijkstra / eval_cvp / process_entry) are the specification; this is the same arithmetic
// spread over lanes, and tests compare both against the oracle.
// ---------------------------------------------------------------------------------------------
constexpr int kWave = 64;
constexpr int kGroup = 8;                 // lanes per work-list entry
constexpr int kGroupsPerWave = kWave / kGroup;

struct StepCtx {
  const Plan* P;
  Cnt* cnt;
  uint32_t* next;
  uint32_t sv;          // dedup stamp of this step
  float lmin;
  uint32_t levals;
  bool lchanged;
  uint32_t* wcur;       // waiting list of the current epoch (Plan.wlist), entries before this step, epoch id
  uint32_t wbase, epoch;
  float lcut;           // min pop time over in-band vertices that moved (Cnt.minchg)
};

// dedup'd, wave-aggregated append of v to the next work list (all lanes of the wave that reach
// this point take part; `want` selects the lanes that actually push)
template <bool DIRTY>
__device__ __forceinline__ void push_agg(StepCtx& S, bool want, uint32_t v)
{
  bool ok = false;
  if (want) {
    if (DIRTY) S.P->dirty[v] = S.sv;                       // "a neighbour moved": re-evaluate next step
    if (S.P->stamp[v] != S.sv) ok = atomicExch(&S.P->stamp[v], S.sv) != S.sv;
  }
  const unsigned long long m = __ballot(ok);
  if (m == 0ull) return;
  const int leader = __ffsll((long long)m) - 1;
  const int lane = threadIdx.x & (kWave - 1);
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(&S.cnt->n_next, (uint32_t)__popcll(m));
  base = __shfl(base, leader);
  if (ok) {
    const uint32_t idx = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (idx < S.P->cap) S.next[idx] = v;
  }
}

// dedup'd (per epoch), wave-aggregated append of v to the waiting list (spec: Ops::park, mnav_eval.h)
__device__ __forceinline__ void park_agg(StepCtx& S, bool want, uint32_t v)
{
  bool ok = false;
  if (want && S.P->wstamp[v] != S.epoch) ok = atomicExch(&S.P->wstamp[v], S.epoch) != S.epoch;
  const unsigned long long m = __ballot(ok);
  if (m == 0ull) return;
  const int leader = __ffsll((long long)m) - 1;
  const int lane = threadIdx.x & (kWave - 1);
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(&S.cnt->n_wait, (uint32_t)__popcll(m));
  base = __shfl(base, leader);
  if (ok) {
    const uint32_t idx = S.wbase + base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (idx < S.P->cap) S.wcur[idx] = v;
  }
}

template <class T>
__device__ __forceinline__ T gshfl(T x, int src) { return __shfl(x, src, kGroup); }

// --- Dijkstra gather over 8 lanes (spec: mnav_eval.h::eval_dijkstra) ---------------------------
__device__ __forceinline__ Eval group_eval_dijkstra(const Plan& P, const Ctl& c, uint32_t v, int sub)
{
  float best_s = inf_f(), best_du = inf_f();
  uint32_t best_u = v;
  const uint32_t beg = P.row_ptr[v], end = P.row_ptr[v + 1];
  for (uint32_t i = beg + sub; i < end; i += kGroup) {
    const Nbr nb = P.nbr[i];
    const float du = P.dist[nb.u];
    if (!(du < c.thr) || du > c.goal_dist) continue;
    const float s = du + nb.w;                                    // dijkstra :331
    if (s < best_s || (s == best_s && s < inf_f() && (du < best_du || (du == best_du && nb.u < best_u)))) {
      best_s = s; best_du = du; best_u = nb.u;
    }
  }
#pragma unroll
  for (int o = 1; o < kGroup; o <<= 1) {
    const float os = __shfl_xor(best_s, o, kGroup), odu = __shfl_xor(best_du, o, kGroup);
    const uint32_t ou = __shfl_xor(best_u, o, kGroup);
    if (os < best_s || (os == best_s && os < inf_f() && (odu < best_du || (odu == best_du && ou < best_u)))) {
      best_s = os; best_du = odu; best_u = ou;
    }
  }
  Eval e; e.d = best_s; e.t = best_s; e.key = key_inf(); e.pred = (best_s < inf_f()) ? best_u : v; e.dir = 0.0f; e.cut = kNone;
  return e;
}

// --- CVP replay over 8 lanes (spec: mnav_eval.h::eval_cvp) ------------------------------------
// corners per lane in the 8-lane replay: 2 = vertices of up to 16 faces in parallel, the rest through the serial rule.
// 1 saves 18 VGPRs (152 instead of 170 unconstrained) but not enough for a fourth wave per SIMD without spilling, and
// measured the same (207 vs 203 plans/s in batches of 128)
#ifndef MNAV_CVP_ROUNDS
#define MNAV_CVP_ROUNDS 2
#endif
constexpr int kCvpRounds = MNAV_CVP_ROUNDS;
struct CornerItem { KeyRef fk; uint32_t trig; bool valid; bool first; CvpCand k; uint32_t v1, v2, face; };

__device__ __forceinline__ KeyRef gshfl_key(const KeyRef& r, int src)
{
  KeyRef o;
  o.k.hi = gshfl(r.k.hi, src); o.k.up = gshfl(r.k.up, src); o.k.lvl = gshfl(r.k.lvl, src); o.own = gshfl(r.own, src);
  return o;
}

// Lanes hold one corner each (fire event + float64 candidate, computed in parallel); the replay walks
// the triggers in pop order with in-group shuffles.  Pop keys of different main-front pops compare by
// the integer `hi` alone; only keys inside one cascade need key_less()'s walk over the cascade tree
// (PopKey, mnav_eval.h), which every lane of the group then performs on the same operands.
__device__ __forceinline__ Eval group_eval_cvp(const Plan& P, const Ctl& c, uint32_t v, int sub)
{
  const uint32_t beg = P.crn_ptr[v], end = P.crn_ptr[v + 1];
  if (end - beg > kCvpRounds * kGroup) return eval_cvp(P, c, v);   // rare high-valence vertex: serial rule
  const bool infl = P.seed_mask != nullptr;
  const bool mute = infl && P.seed_mask[v] == kInflMute;
  CornerItem it[kCvpRounds];
#pragma unroll
  for (int r = 0; r < kCvpRounds; ++r) {
    const uint32_t i = beg + sub + r * kGroup;
    it[r].fk = key_ref_of(key_inf(), inf_f(), 0); it[r].trig = kNone; it[r].valid = false;
    it[r].v1 = kNone; it[r].v2 = kNone; it[r].face = kNone; it[r].first = false;
    it[r].k.u3tmp = 0.0; it[r].k.cand = 0.0; it[r].k.dir = 0.0f; it[r].k.sel = 0; it[r].k.kind = 0;
    if (i < end) {
      const Corner k = P.crn[i];
      const Fire f = corner_fire(P, c, k);
      if (f.trig != kNone && !key_descends_from(P, f.trig, v)) {       // (spec: eval_cvp)
        it[r].valid = true; it[r].fk = f.key; it[r].trig = f.trig;
        if (infl) {                                                    // inflation wave: float32 rule (spec: eval_cvp)
          const InflCand u = infl_candidate(P.dist[k.v1], P.dist[k.v2], k.a, k.b, k.c, P.infl_max);
          it[r].k.u3tmp = (double)u.u3tmp; it[r].k.cand = 0.0; it[r].k.dir = 0.0f; it[r].k.sel = u.requeue ? 1 : 0; it[r].k.kind = u.ok ? 3 : 0;
        } else
        it[r].k = cvp_candidate(P.dist[k.v1], P.dist[k.v2], k.a, k.b, k.c);
        it[r].v1 = k.v1; it[r].v2 = k.v2; it[r].face = corner_face(k); it[r].first = corner_first_for(k, f.trig);
      }
    }
  }
  const int gbase = (threadIdx.x & (kWave - 1)) & ~(kGroup - 1);
  Eval e; e.d = inf_f(); e.t = inf_f(); e.key = key_inf(); e.pred = v; e.dir = 0.0f; e.cut = kNone; e.keyd = inf_f();
  constexpr unsigned long long kNoKey = ~0ull;
  KeyRef last = key_ref_of(key_inf(), inf_f(), 0);
  bool first = true, queued = false;
  const uint32_t max_pass = 2u * (end - beg) + 2u;                  // (spec: eval_cvp)
  for (uint32_t pass_no = 0;; ++pass_no) {
    if (pass_no == max_pass) { raise_flag(P, kFlagWalkLimit); break; }
    // next trigger pop strictly after the last one: smallest `hi` first, the tree decides among equals
    bool el[kCvpRounds];
    unsigned long long mh = kNoKey;
#pragma unroll
    for (int r = 0; r < kCvpRounds; ++r) {
      el[r] = it[r].valid && (first || key_less(P, last, it[r].fk));
      if (el[r] && it[r].fk.k.hi < mh) mh = it[r].fk.k.hi;
    }
#pragma unroll
    for (int o = 1; o < kGroup; o <<= 1) { const unsigned long long om = __shfl_xor(mh, o, kGroup); mh = om < mh ? om : mh; }
    if (mh == kNoKey) break;
    KeyRef m = last; uint32_t m_trig = kNone;
#pragma unroll
    for (int r = 0; r < kCvpRounds; ++r) {
      unsigned gm = (unsigned)((__ballot(el[r] && it[r].fk.k.hi == mh) >> gbase) & 0xFFull);
      while (gm) {
        const int src = __ffs((int)gm) - 1;
        gm &= gm - 1;
        const uint32_t ct = gshfl(it[r].trig, src);
        if (ct == m_trig) continue;
        const KeyRef cand = gshfl_key(it[r].fk, src);
        if (m_trig == kNone || key_less(P, cand, m)) { m = cand; m_trig = ct; }
      }
    }
    if (queued && !key_less(P, m, key_ref_of(e.key, e.keyd, v))) break;   // v pops before this trigger
    bool any = false;
    float ins_d = 0.0f;
#pragma unroll
    for (int pr = 0; pr < 2 * kCvpRounds; ++pr) {                  // trigger's circulator order: flagged face first
      const int r = pr % kCvpRounds;
      const bool want_first = pr < kCvpRounds;
      unsigned gm = (unsigned)((__ballot(it[r].valid && it[r].trig == m_trig && it[r].first == want_first) >> gbase) & 0xFFull);
      while (gm) {
        const int src = __ffs((int)gm) - 1;
        gm &= gm - 1;
        CvpCand k;
        k.u3tmp = gshfl(it[r].k.u3tmp, src); k.cand = gshfl(it[r].k.cand, src); k.dir = gshfl(it[r].k.dir, src);
        k.sel = gshfl(it[r].k.sel, src); k.kind = gshfl(it[r].k.kind, src);
        int sel = 0; float dir = 0.0f;
        if (k.kind == 3) {                                           // inflation :252,:298-311
          const float u3tmp = (float)k.u3tmp;
          if (e.d != 0.0f && u3tmp < e.d) {
            e.d = u3tmp; e.pred = gshfl(it[r].v1, src); e.cut = gshfl(it[r].v2, src);   // supports of the last lowering update (vector field)
            if (k.sel) { any = true; ins_d = e.d; }
          }
        } else if (k.kind != 0 && cvp_apply(k, e.d, sel, dir)) {
          const uint32_t v1 = gshfl(it[r].v1, src), v2 = gshfl(it[r].v2, src);
          e.pred = (sel == 1) ? v1 : v2; e.dir = dir; e.cut = gshfl(it[r].face, src);
          any = true; ins_d = e.d;
        }
      }
    }
    if (any && !mute) { e.key = key_for(P, ins_d, v, m); e.keyd = ins_d; queued = true; }   // ordinary pop, or a place inside this trigger's cascade
    last = m; first = false;
  }
  if (!queued) { if (!infl) e.pred = v; e.key = key_inf(); e.keyd = inf_f(); }
  e.t = key_time(e.key);
  return e;
}

template <uint32_t PLANNER>
__device__ __forceinline__ Eval group_eval(const Plan& P, const Ctl& c, uint32_t v, int sub)
{
  if constexpr (PLANNER == kPlannerCvp) return group_eval_cvp(P, c, v, sub);
  else return group_eval_dijkstra(P, c, v, sub);
}

// push the neighbourhood of v (spec: process_entry)
template <uint32_t PLANNER>
__device__ __forceinline__ void group_push_neighbours(StepCtx& S, const Plan& P, uint32_t v, int sub, bool want)
{
  if constexpr (PLANNER == kPlannerCvp) {
    const uint32_t beg = P.crn_ptr[v], end = P.crn_ptr[v + 1];
    const uint32_t rounds = want ? (end - beg + kGroup - 1) / kGroup : 0;
    // every lane of the wave must reach push_agg the same number of times -> wave-max of rounds
    uint32_t wr = rounds;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wr = max(wr, (uint32_t)__shfl_xor((int)wr, o));
    for (uint32_t r = 0; r < wr; ++r) {
      const uint32_t i = beg + sub + r * kGroup;
      uint32_t a = kNone, b = kNone;
      if (want && i < end) { const Corner k = P.crn[i]; if (k.v1 != kNone) { a = k.v1; b = k.v2; } }
      push_agg<true>(S, a != kNone, a);
      push_agg<true>(S, b != kNone, b);
    }
  } else {
    const uint32_t beg = P.row_ptr[v], end = P.row_ptr[v + 1];
    const uint32_t rounds = want ? (end - beg + kGroup - 1) / kGroup : 0;
    uint32_t wr = rounds;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wr = max(wr, (uint32_t)__shfl_xor((int)wr, o));
    for (uint32_t r = 0; r < wr; ++r) {
      const uint32_t i = beg + sub + r * kGroup;
      const bool w = want && i < end;
      const uint32_t u = w ? P.nbr[i].u : kNone;
      push_agg<true>(S, w, u);
    }
  }
}

// one work-list entry per 8-lane group; `active` = this group has an entry (inactive groups only
// take part in the wave-wide pushes).  Spec: mnav_eval.h::process_entry / process_repair.
template <uint32_t PLANNER, bool REPAIR>
__device__ __forceinline__ void group_process(StepCtx& S, const Plan& P, const Ctl& c, bool active, uint32_t v, int sub)
{
  constexpr bool cvp = (PLANNER == kPlannerCvp);
  bool push_nb = false, retain = false, self_again = false;
  float t_new = inf_f(), t_old_for_cut = inf_f();
  if (active && !is_seed(P, v)) {
    const float old_d = P.dist[v];
    PopKey old_key = key_inf();
    if constexpr (cvp) old_key = P.tkey[v];
    const float old_t = cvp ? key_time(old_key) : old_d;
    t_old_for_cut = old_t;
    bool go;
    if (REPAIR) go = (old_d < inf_f());
    else go = !(old_t < c.thr_fixed) && !(cvp && P.blocked[v]);
    // parked out of band and no neighbour moved since the last evaluation: keep waiting, as is
    const bool parked = !REPAIR && go && !c.band_new && !(old_t < c.thr) && old_t < inf_f() && P.dirty[v] != (uint32_t)c.it;
    if (parked) { retain = true; t_new = old_t; }
    else if (go) {
      if (!REPAIR || old_t > c.goal_dist) {                          // spec: process_repair (pop time, not value)
        if (sub == 0) ++S.levals;
        const Eval e = group_eval<PLANNER>(P, c, v, sub);
        bool changed = (f2u(e.d) != f2u(old_d)) || (f2u(e.t) != f2u(old_t)) || (e.pred != P.pred[v]);
        if (cvp) changed = changed || (e.key != old_key) || (e.cut != P.cutf[v]) || (f2u(e.dir) != f2u(P.dirn[v])) ||
                           (P.keyd && f2u(e.keyd) != f2u(P.keyd[v]));
#ifdef MNAV_DEBUG_FLIP                    // debugging aid: who keeps changing in a band that does not settle
        if (changed && sub == 0 && !REPAIR && c.band_steps >= 40 && c.band_steps < 44)
          printf("flip it %d v %u d %.9g->%.9g t %.9g->%.9g key hi %llx->%llx up %d->%d lvl %u->%u keyd %.9g->%.9g\n", c.it, v, old_d, e.d, old_t, e.t,
                 old_key.hi, e.key.hi, (int)old_key.up, (int)e.key.up, old_key.lvl, e.key.lvl, P.keyd ? P.keyd[v] : 0.f, e.keyd);
#endif
        if ((changed || REPAIR) && sub == 0) {
          P.dist[v] = e.d; P.pred[v] = e.pred;
          if constexpr (cvp) { P.tkey[v] = e.key; P.dirn[v] = e.dir; P.cutf[v] = e.cut; if (P.keyd) P.keyd[v] = e.keyd; }
        }
        t_new = e.t;
        if (REPAIR && cvp && sub == 0 && (f2u(e.d) != f2u(old_d) || e.key != old_key)) S.lchanged = true;   // sweep again
        if (!REPAIR) {
          const bool was_in = old_t < c.thr, now_in = e.t < c.thr;
          push_nb = (changed && (was_in || now_in)) || (now_in && c.band_new);
          retain = !now_in && e.t < inf_f();
          if constexpr (cvp) self_again = (e.key.lvl > 0u || old_key.lvl > 0u) && e.key != old_key;   // spec: process_entry
        }
      } else {
        t_new = old_t;
      }
      if (REPAIR) retain = (t_new >= c.thr) && (t_new < inf_f());
    }
  }
  if ((push_nb || self_again) && sub == 0) {
    S.lchanged = true;
    if (push_nb && ((t_old_for_cut < c.thr) != (t_new < c.thr))) S.lcut = fminf(S.lcut, fminf(t_old_for_cut, t_new));   // crossed the bound (spec: note_cut)
  }
  group_push_neighbours<PLANNER>(S, P, v, sub, push_nb);
  push_agg<true>(S, self_again && sub == 0, v);
  park_agg(S, retain && sub == 0, v);
  if (retain && sub == 0) S.lmin = fminf(S.lmin, t_new);
}

__device__ __forceinline__ float wave_min(float x)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x = fminf(x, __shfl_xor(x, o));
  return x;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t x)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
  return x;
}

// grid = (waves per plan, plans).  slot j (0..5) selects the ping-pong control block (j&1) and
// the counter block (j%3).
#ifndef MNAV_STEP_OCC                     // waves per SIMD the register allocator must reach: 3 (<= 168 VGPRs).  The CVP replay sits
#define MNAV_STEP_OCC 3                   // right at that edge (159-170 VGPRs); at 2 waves a batch is 20 % slower, forcing 4 or 5
#endif                                    // spills and is slower still (measured: 179 / 150 / 120 plans/s at 3 / 4 / 5)
#define MNAV_STEP_BOUNDS __launch_bounds__(kWave, MNAV_STEP_OCC)
// PRECTL: the step's control block was computed by k_cvp_ctl (batches on the wide kernel); this kernel then only serves the plans
// that are in a repair / rebuild / cut step, which sweep over all vertices with the 8-lane code below.
template <uint32_t PLANNER, bool PRECTL>
__device__ __forceinline__ void step_body(const Plan* __restrict__ plans, int j, uint32_t plan_index)
{
  const Plan& P = plans[plan_index];
  const int lane = threadIdx.x;
  __shared__ Ctl s_ctl;
  if (lane == 0) {
    if constexpr (PRECTL) s_ctl = P.ctl[j & 1];
    else {
      const Ctl prev = P.ctl[(j + 1) & 1];
      const Cnt cprev = P.cnt[(j + 2) % 3];
      const Ctl cur = controller(P, prev, cprev);
      s_ctl = cur;
      if (blockIdx.x == 0) {
        P.ctl[j & 1] = cur;
        Cnt z; z.n_next = 0; z.changed = 0; z.minkey = 0x7f800000u; z.evals = 0; z.n_wait = 0; z.minchg = 0x7f800000u; z.pad[0] = z.pad[1] = 0;
        P.cnt[(j + 1) % 3] = z;
      }
    }
  }
  __syncthreads();
  const Ctl cur = s_ctl;
  if (cur.done) return;
  if constexpr (PRECTL) { if (cur.repair <= 2 && P.seed_mask == nullptr) return; }   // k_step_wide's
  Cnt* cnt = &P.cnt[j % 3];
  StepCtx S{ &P, cnt, P.list[(cur.it + 1) & 1], (uint32_t)cur.it + 1u, inf_f(), 0u, false, P.wlist[cur.wsel & 1u], cur.wbase, cur.epoch, inf_f() };
  const int sub = lane & (kGroup - 1), grp = lane >> 3;
  const uint32_t ngroups = gridDim.x * kGroupsPerWave;
  const uint32_t g0 = blockIdx.x * kGroupsPerWave + grp;
  if (cur.repair == 3) {                                             // spec: process_cut -- no evaluation
    const uint32_t nthreads = gridDim.x * kWave, tid = blockIdx.x * kWave + lane;
    const uint32_t* list = P.list[cur.it & 1];
    for (uint32_t base = 0; base < cur.n; base += nthreads) {         // the work list is carried over
      const uint32_t i = base + tid;
      push_agg<true>(S, i < cur.n, i < cur.n ? list[i] : 0u);
    }
    for (uint32_t base = 0; base < P.V; base += nthreads) {           // keyed vertices at or above the cut wait for the restarted band
      const uint32_t v = base + tid;
      bool want = false; float t = inf_f();
      if (v < P.V && !is_seed(P, v)) {
        t = (PLANNER == kPlannerCvp) ? key_time(P.tkey[v]) : P.dist[v];
        want = t >= cur.thr && t < inf_f();
      }
      park_agg(S, want, v);
      if (want) S.lmin = fminf(S.lmin, t);
    }
  } else if (cur.repair == 1) {                                      // spec: process_repair
    const uint32_t rounds = (P.V + ngroups - 1) / ngroups;
    for (uint32_t r = 0; r < rounds; ++r) {
      const uint32_t v = g0 + r * ngroups;
      group_process<PLANNER, true>(S, P, cur, v < P.V, v < P.V ? v : 0u, sub);
    }
  } else if (cur.repair == 2) {                                      // spec: process_rebuild (band shrink, band_new == 1)
    const uint32_t rounds = (P.V + ngroups - 1) / ngroups;
    for (uint32_t r = 0; r < rounds; ++r) {
      const uint32_t v = g0 + r * ngroups;
      const bool active = v < P.V && P.dist[v < P.V ? v : 0u] < inf_f();
      group_process<PLANNER, false>(S, P, cur, active, active ? v : 0u, sub);
    }
  } else {
    // the work list; in the first step of a band also the waiting list the previous band left behind
    const uint32_t* list = P.list[cur.it & 1];
    const uint32_t* wprev = P.wlist[(cur.wsel ^ 1u) & 1u];
    const uint32_t ntot = cur.n + cur.wread;
    const uint32_t rounds = (ntot + ngroups - 1) / ngroups;
    for (uint32_t r = 0; r < rounds; ++r) {
      const uint32_t i = g0 + r * ngroups;
      const bool active = i < ntot;
      const uint32_t v = active ? (i < cur.n ? list[i] : wprev[i - cur.n]) : 0u;
      group_process<PLANNER, false>(S, P, cur, active, v, sub);
    }
  }
  const float wmin = wave_min(S.lmin);
  const float wcut = wave_min(S.lcut);
  const uint32_t wev = wave_sum(S.levals);
  const bool wch = __any(S.lchanged);
  if (lane == 0) {
    if (wmin < inf_f()) atomicMin(&cnt->minkey, f2u(wmin));
    if (wcut < inf_f()) atomicMin(&cnt->minchg, f2u(wcut));
    if (wev) atomicAdd(&cnt->evals, wev);
    if (wch) atomicOr(&cnt->changed, 1u);
  }
}

template <uint32_t PLANNER>
__global__ MNAV_STEP_BOUNDS void k_step(const Plan* __restrict__ plans, int j) { step_body<PLANNER, false>(plans, j, blockIdx.y); }

#include "mnav_cvp_wide.h"   // CVP batches: wide_round, k_cvp_ctl, k_step_wide, k_step_repair

// CVP verification sweep, run once after the last step (the CVP counterpart of k_dij_finalize's fixed-point
// check): every vertex is evaluated once more on the CONVERGED state.  (1) Its stored (potential, pop key,
// predecessor, direction, cutting face) must be reproduced exactly -- a vertex that was evaluated against a
// stale or torn key of a far cascade ancestor and never re-queued shows up here; (2) a walk over the cascade
// tree that hits its bound on the converged tree would silently change the pop order -- during the iteration
// such hits are transient and ignored (k_flags_reset clears them), here they count.  Either way the plan
// returns INTERNAL_ERROR instead of a potential that may not be the reference's.
__global__ void k_flags_reset(const Plan* __restrict__ plans)
{
  if (threadIdx.x == 0) { Cnt z; memset(&z, 0, sizeof(z)); z.minkey = 0x7f800000u; plans[blockIdx.x].cnt[3] = z; }
}

__global__ __launch_bounds__(kWave) void k_cvp_verify(const Plan* __restrict__ plans, int fix, uint32_t* __restrict__ any_bad)
{
  const Plan& P = plans[blockIdx.y];
  const int lane = threadIdx.x;
  const Ctl a = P.ctl[0], b = P.ctl[1];
  const Ctl cur = (a.it > b.it) ? a : b;
  if (!cur.done || cur.overflow) return;                               // reported as an error anyway
  const int sub = lane & (kGroup - 1), grp = lane >> 3;
  const uint32_t ngroups = gridDim.x * kGroupsPerWave;
  const uint32_t rounds = (P.V + ngroups - 1) / ngroups;
  uint32_t bad = 0;
  for (uint32_t r = 0; r < rounds; ++r) {
    const uint32_t v = blockIdx.x * kGroupsPerWave + grp + r * ngroups;
    const bool act = v < P.V && !is_seed(P, v < P.V ? v : 0u) && !P.blocked[v < P.V ? v : 0u];
    if (!act) continue;                                               // whole 8-lane groups skip together
    const Eval e = group_eval_cvp(P, cur, v, sub);
    if (sub == 0 && !verify_entry(P, cur, v, e, fix != 0)) ++bad;     // spec: mnav_eval.h (with fix: stores the re-evaluated state)
  }
  bad = wave_sum(bad);
  if (lane == 0 && bad) { atomicAdd(&P.cnt[3].changed, bad); atomicOr(any_bad, 1u); }
}

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
  const uint32_t *vptr, *verts, *hptr, *halo_verts, *halo_tile, *eptr, *rptr;
  const uint16_t* rowptr;
  const uint16_t* col;     // per local edge: local target            } split arrays: 6 B per edge in LDS;
  const float* tw;         // per local edge: push weight (+inf on padding) } tiles padded to 8 entries
  float* dist;
  uint32_t* pend[2];       // per-tile wake-up value (float bits), ping-pong by round parity
  float* tlast;            // per-tile threshold of its last solve
  TCtl* ctl;               // [2]
  TCnt* cnt;               // [3]
  uint32_t seed, target;
  double offset;
  float band;
  uint32_t max_rounds;
  uint32_t max_nv, max_nh, max_ne;
  const uint32_t* cancel;  // device word set by mnav_cancel (polled by the persistent kernels), may be null
  uint32_t t_lo, t_hi;     // tiles this process owns (sharded single plan, mnav_shard_*); t_hi == 0: all tiles
  const uint8_t* owned;    // partitioned mesh (mnav_shard_setup_partition): 1 = this process owns the vertex, 0 = halo copy whose
                           // neighbourhood is incomplete here (its value arrives through the exchange); null: every vertex is owned
};

constexpr int kTileBlock = 256;
#ifndef MNAV_PERSIST_WG_PER_CU
#define MNAV_PERSIST_WG_PER_CU 6        // register budget of k_plan_persistent: 6 workgroups (24 waves) per CU -> <= 80 VGPRs
#endif
// Tiles solved per best-first scan of k_plan_persistent (<= kTileBlock / 64) and how far behind the best one a further
// candidate may lie, in bands.  Measured on C2 (5120 plans, ms per launch): 1 -> 408.5; 2 within one band -> 398.4;
// 4 within one band -> 453.7 (the order matters more than the scans cost); 4 within 0.1 / 0.25 / 0.5 bands -> 405.6 /
// 408.5 / 413.0; 2 within 0.5 -> 402.6.
#ifndef MNAV_SCAN_SLACK
#define MNAV_SCAN_SLACK 1.0f
#endif
#ifndef MNAV_SCAN_CANDS
#define MNAV_SCAN_CANDS 2
#endif
constexpr int kTileVpt = 8;              // owned vertices per thread: tile_size <= 2048
constexpr int kTileTodo = 64;            // tiles one workgroup takes per round
constexpr uint32_t kInfBits = 0x7f800000u;

__host__ __device__ inline uint32_t pad_to(uint32_t x, uint32_t a) { return (x + a - 1) / a * a; }

// LDS image of one tile (dynamic shared memory), shared by k_tile_round and k_plan_persistent
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

// ---------------------------------------------------------------------------------------------
// Persistent per-plan variant (batches): ONE workgroup owns a plan from seed to convergence and
// walks its tiles best-first -- always the tile with the smallest wake-up value, with the band
// [m, m + band) -- without any launch or grid-wide round in between.  Independent plans never
// talk to each other, so there is no inter-workgroup protocol at all; hundreds of plans run
// concurrently (2 workgroups per CU).  Same tile solve as k_tile_round (LDS queue sweeps, ds_min
// on float bits); state that the workgroup re-reads after writing it (dist, wake-ups, tlast) is
// read with L1-bypassing (non-temporal) loads, i.e. served by the L2 and never by a stale L1 line.
// ---------------------------------------------------------------------------------------------
// Non-temporal loads bypass the per-CU L1 (served by the L2) like agent-scope atomic loads do, but
// unlike those they are ordinary loads: many stay in flight, one wait at the first use.  Stores are
// write-through to the L2 anyway; everything this workgroup re-reads is read through these.
__device__ __forceinline__ uint32_t ldg_u32(MNAV_GLOBAL const uint32_t* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ float ldg_f32(MNAV_GLOBAL const float* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void stg_u32(MNAV_GLOBAL uint32_t* p, uint32_t v) { *p = v; }
__device__ __forceinline__ void stg_f32(MNAV_GLOBAL float* p, float v) { *p = v; }

template <int VPT>   // owned vertices per thread: tile_size <= VPT * 256
__global__ __launch_bounds__(kTileBlock, MNAV_PERSIST_WG_PER_CU) void k_plan_persistent(const TilePlan* __restrict__ plans)
{
  const TilePlan& P = plans[blockIdx.x];
  const int tid = threadIdx.x;
  __shared__ unsigned long long s_best[kTileBlock / 64];
  __shared__ uint32_t s_hdr[8];
  __shared__ uint32_t s_nq[3];
  __shared__ float s_bound;
  __shared__ uint32_t s_stop;
  if (tid == 0) s_stop = 0u;
  MNAV_GLOBAL uint32_t* pend = as_global(P.pend[0]);
  MNAV_GLOBAL const uint32_t* g_vptr = as_global(P.vptr);
  MNAV_GLOBAL const uint32_t* g_hptr = as_global(P.hptr);
  MNAV_GLOBAL const uint32_t* g_eptr = as_global(P.eptr);
  MNAV_GLOBAL const uint32_t* g_rptr = as_global(P.rptr);
  MNAV_GLOBAL const uint32_t* g_verts = as_global(P.verts);
  MNAV_GLOBAL const uint32_t* g_halo_verts = as_global(P.halo_verts);
  MNAV_GLOBAL const uint32_t* g_halo_tile = as_global(P.halo_tile);
  MNAV_GLOBAL float* g_dist = as_global(P.dist);
  MNAV_GLOBAL float* g_tlast = as_global(P.tlast);

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const TileLds L = tile_lds_layout(smem, P.max_nv, P.max_nh, P.max_ne);
  uint32_t* const ldu = L.ldu; uint32_t* const lh0 = L.lh0; uint16_t* const q0 = L.q0;

  uint32_t acts = 0, sweeps_total = 0;
  uint32_t status = 0;   // 0 converged, 2 activation cap hit
#ifdef MNAV_TILE_TIMING
  unsigned long long tt[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#define PT_STAMP(k) do { if (tid == 0 && blockIdx.x == 0) tt[k] = clock64(); } while (0)
#else
#define PT_STAMP(k) do { } while (0)
#endif
  for (;;) {
    PT_STAMP(0);
    // best-first: the tile with the smallest wake-up value
    unsigned long long best = ~0ull;
    for (uint32_t t0 = tid; t0 < P.ntiles; t0 += 8 * kTileBlock) {        // 8 loads in flight per thread
      uint32_t pv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const uint32_t t = t0 + u * kTileBlock; pv[u] = (t < P.ntiles) ? ldg_u32(pend + t) : kInfBits; }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const unsigned long long k = ((unsigned long long)pv[u] << 32) | (t0 + u * kTileBlock);
        best = k < best ? k : best;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long ob = __shfl_xor(best, o); best = ob < best ? ob : best; }
    if ((tid & 63) == 0) s_best[tid >> 6] = best;
    if (tid == 0) {
      const float dt = ldg_f32(g_dist + P.target);
      s_bound = (float)((double)dt + fmax(P.offset, 0.0));         // >= the final goal_dist (dijkstra :296); negative offsets: goal_cut
      // mnav_cancel (dijkstra :287 `&& !cancel_planning_`): a word in device memory that mnav_cancel sets with a
      // 4-byte copy on its own stream; one agent-scope load every 16 tile activations (~0.3 ms)
      if ((acts & 15u) == 0u) s_stop = P.cancel ? __hip_atomic_load(P.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    }
    __syncthreads();
    if (s_stop) { status = 3; break; }
    // the four waves scanned disjoint quarters of the tiles: their four minima, in ascending order, are the candidates
    // of this scan.  Every candidate below the band threshold is solved without another scan (MNAV_SCAN_CANDS of them at
    // most; label-correcting: the order of the solves does not change the fixed point, only the work).
    unsigned long long cand[kTileBlock / 64];
#pragma unroll
    for (int w = 0; w < kTileBlock / 64; ++w) cand[w] = s_best[w];
#pragma unroll
    for (int a = 0; a < kTileBlock / 64; ++a)
#pragma unroll
      for (int b = a + 1; b < kTileBlock / 64; ++b)
        if (cand[b] < cand[a]) { const unsigned long long x = cand[a]; cand[a] = cand[b]; cand[b] = x; }
    best = cand[0];
    const float bound = s_bound;
    const float m = u2f((uint32_t)(best >> 32));
    if (!(m < inf_f()) || m > bound) break;                        // nothing left that may propagate
    if (acts >= P.max_rounds) { status = 2; break; }
    float thr = m + P.band;
    if (!(thr > m)) thr = next_up(m);
    PT_STAMP(1);
#pragma unroll 1
    for (int ci = 0; ci < MNAV_SCAN_CANDS; ++ci) {
    const float mc = u2f((uint32_t)(cand[ci] >> 32));
    if (ci > 0 && (!(mc < m + MNAV_SCAN_SLACK * P.band) || mc > bound)) break;   // only tiles about as urgent as the best one (uniform over the workgroup)
    const uint32_t t = (uint32_t)cand[ci];
    if (tid == 0) {
      stg_u32(pend + t, kInfBits);
      s_hdr[0] = g_vptr[t]; s_hdr[1] = g_vptr[t + 1]; s_hdr[2] = g_hptr[t]; s_hdr[3] = g_hptr[t + 1];
      s_hdr[4] = g_eptr[t]; s_hdr[5] = g_eptr[t + 1]; s_hdr[6] = g_rptr[t]; s_hdr[7] = f2u(ldg_f32(g_tlast + t));
      s_nq[0] = 0; s_nq[1] = 0; s_nq[2] = 0;
    }
    __syncthreads();
    const uint32_t v0 = s_hdr[0], nv = s_hdr[1] - v0;
    const uint32_t h0 = s_hdr[2], nh = s_hdr[3] - h0;
    const uint32_t e0 = s_hdr[4], ne = s_hdr[5] - e0;
    const uint32_t r0 = s_hdr[6];
    const uint32_t nl = nv + nh;
    const float tl = u2f(s_hdr[7]);
    PT_STAMP(2);
    // stage (see k_tile_round)
    uint32_t gi[VPT];
#pragma unroll
    for (int k = 0; k < VPT; ++k) { const uint32_t i = tid + k * kTileBlock; gi[k] = (i < nv) ? g_verts[v0 + i] : 0u; }
    uint32_t hi[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) { const uint32_t i = tid + k * kTileBlock; hi[k] = (i < nh) ? g_halo_verts[h0 + i] : 0u; }
    stage_tile_graph(P, L, e0, ne, r0, nl, tid);
    uint32_t orig[VPT];
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
      const uint32_t i = tid + k * kTileBlock;
      orig[k] = 0u;
      if (i < nv) {
        const float d = ldg_f32(g_dist + gi[k]);
        orig[k] = f2u(d); ldu[i] = orig[k];
        if (d < thr && d <= bound && !(d < tl)) q0[atomicAdd(&s_nq[0], 1u)] = (uint16_t)i;
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const uint32_t i = tid + k * kTileBlock;
      if (i < nh) { const float d = ldg_f32(g_dist + hi[k]); ldu[nv + i] = f2u(d); lh0[i] = f2u(d); if (d < thr && d <= bound) q0[atomicAdd(&s_nq[0], 1u)] = (uint16_t)(nv + i); }
    }
    for (uint32_t i = tid + 2 * kTileBlock; i < nh; i += kTileBlock) {
      const float d = ldg_f32(g_dist + g_halo_verts[h0 + i]);
      ldu[nv + i] = f2u(d); lh0[i] = f2u(d); if (d < thr && d <= bound) q0[atomicAdd(&s_nq[0], 1u)] = (uint16_t)(nv + i);
    }
    __syncthreads();
    PT_STAMP(3);
    const uint32_t sweep = tile_sweeps(L, nv, thr, bound, s_nq, tid);
    PT_STAMP(4);
    // wake-ups for the owners of undercut halo vertices, write-back, own left-over
    uint32_t own_left = kInfBits;
    for (uint32_t i = tid; i < nh; i += kTileBlock) {
      const uint32_t b = ldu[nv + i];
      if (b < lh0[i]) atomicMin((uint32_t*)&pend[g_halo_tile[h0 + i]], b);
    }
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
      const uint32_t i = tid + k * kTileBlock;
      if (i < nv) {
        const uint32_t db = ldu[i];
        if (db != orig[k]) stg_f32(g_dist + gi[k], u2f(db));
        const float d = u2f(db);
        if (!(d < thr) && d <= bound) own_left = min(own_left, db);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) own_left = min(own_left, (uint32_t)__shfl_xor((int)own_left, o));
    if ((tid & 63) == 0 && own_left != kInfBits) atomicMin((uint32_t*)&pend[t], own_left);
    if (tid == 0) stg_f32(g_tlast + t, thr);
    ++acts; sweeps_total += sweep;
    // every store / atomic of this activation must have reached the L2 before the next scan
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    }                                                              // candidates of this scan
#ifdef MNAV_TILE_TIMING
    if (tid == 0 && blockIdx.x == 0) {
      tt[5] = clock64(); tt[6] = sweep; tt[7] = nl;
      const unsigned int k = atomicAdd(&g_tile_timing_n, 1u);
      if (k < 4096) for (int q = 0; q < 8; ++q) g_tile_timing[k * 8 + q] = tt[q];
    }
#endif
  }
  if (tid == 0) {
    TCtl c; memset(&c, 0, sizeof(c));
    c.it = (int32_t)acts; c.done = 1; c.acts = acts; c.sweeps = sweeps_total; c.pad[0] = status;
    P.ctl[0] = c; P.ctl[1] = c;
  }
}

#include "mnav_async.h"   // k_plan_async: the tiles without rounds (engine 6, opt-in)

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

#include "mnav_shard.h"   // kernels of the sharded single plan (k_shard_*)

struct PlanResult {
  uint32_t code;
  uint32_t path_len;
  uint32_t steps, bands, armed, overflow;
  float goal_dist;
  uint32_t shrinks;
  unsigned long long settled;
  unsigned long long evals;
};

// Exact cut-off semantics + predecessors in one gather pass over all vertices (8 lanes per
// vertex).  After the tile rounds every vertex with dist <= goal_dist holds its final value.
// A vertex above goal_dist keeps, in the reference, the tentative value it got from expanded
// (dist <= goal_dist) neighbours only, or +inf -- exactly eval_dijkstra with thr = +inf.
// pred = first-popped neighbour attaining the minimum (DESIGN.md tie rule).
constexpr int kFinVpt = 3;               // local vertices (owned + halo) per thread held in registers
__host__ __device__ inline size_t finalize_lds_bytes(uint32_t max_nv, uint32_t max_nh, uint32_t max_ne)
{
  return tile_lds_bytes(max_nv, max_nh, max_ne) + 4 * (size_t)pad_to(max_nv, 4) + 4 * (size_t)pad_to(max_nv + max_nh, 4) + 8 * (size_t)max_nv;
}

__device__ __forceinline__ void store3(float* p, float x, float y, float z) { p[0] = x; p[1] = y; p[2] = z; }

// Source of the distances when the tile-batch engine ran (mnav_tb.h): its blocked per-(tile, plan) slices, addressed through
// vaddr[v] = {slice offset of v's tile, slice length << 8 | local index}; the engine's control words for the plan records.
struct FinBlocked { const float* D; const uint2* vaddr; uint32_t NP; const uint32_t* iters; const uint32_t* err; const uint32_t* n_cand;
                    const float* xyz; float* const* vecmaps; };   // vecmaps != null: computeVectorMap (dijkstra :189-209) in the same pass

// PG plans per workgroup share ONE staged tile graph (the staging -- 21 KB per tile out of L2 -- was most of this pass in
// large batches); BLOCKED: the distances are gathered from the tile-batch engine's slices, every value is written.
template <int PG, bool BLOCKED>
__global__ __launch_bounds__(kTileBlock) void k_dij_finalize(const Plan* __restrict__ plans, const TilePlan* __restrict__ tplans,
                                                             uint32_t* __restrict__ mismatch, PlanResult* __restrict__ res,
                                                             uint32_t tiles_per_block, uint32_t n_plans, FinBlocked B)
{
  // grid: x = group of PG plans, y = chunk of tiles.  Only tiles that were activated or woken are looked at: any
  // vertex that owes a value to an expanded source sits in such a tile (its source pushed to it
  // through a halo copy, which wakes the owner); everything else keeps dist = inf / pred = itself.
  // Per tile the push graph is staged in LDS like in the solve and read backwards: every edge
  // x -> y with an expanded source offers (d[x] + w, d[x], x) to its owned target y; pass 1 takes the
  // smallest sum (ds_min on the float bits), pass 2 the smallest (d[x], x) among the edges that attain
  // it (64-bit ds_min) -- the reference's predecessor under the (value, id) pop order.
  const uint32_t p0 = blockIdx.x * PG;
  const TilePlan& T = tplans[BLOCKED ? 0u : p0];                     // the mesh tables are the same in every record
  const int tid = threadIdx.x;
  MNAV_GLOBAL const uint32_t* g_vptr = as_global(T.vptr);
  MNAV_GLOBAL const uint32_t* g_hptr = as_global(T.hptr);
  MNAV_GLOBAL const uint32_t* g_eptr = as_global(T.eptr);
  MNAV_GLOBAL const uint32_t* g_rptr = as_global(T.rptr);
  MNAV_GLOBAL const uint32_t* g_verts = as_global(T.verts);
  MNAV_GLOBAL const uint32_t* g_halo_verts = as_global(T.halo_verts);
  MNAV_GLOBAL const u32x2* g_va = (MNAV_GLOBAL const u32x2*)as_global((const uint32_t*)B.vaddr);
  MNAV_GLOBAL const float* g_D = as_global(B.D);
  MNAV_GLOBAL const float* g_xyz = as_global(B.xyz);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const TileLds L = tile_lds_layout(smem, T.max_nv, T.max_nh, T.max_ne);
  uint32_t* const lsum = reinterpret_cast<uint32_t*>(smem + tile_lds_bytes(T.max_nv, T.max_nh, T.max_ne));
  uint32_t* const lgid = lsum + pad_to(T.max_nv, 4);
  unsigned long long* const lkey = reinterpret_cast<unsigned long long*>(lgid + pad_to(T.max_nv + T.max_nh, 4));
  const uint32_t np = (n_plans - p0 < (uint32_t)PG) ? n_plans - p0 : (uint32_t)PG;   // plans of this group
  // per plan of the group: seed, goal_dist, armed, settled count
  __shared__ uint32_t s_seed[PG], s_armed[PG], s_settled[PG], s_tie[PG];
  __shared__ float s_goal[PG], s_cut[PG];
  if (tid < PG) {
    const int q = tid;
    s_settled[q] = 0u; s_seed[q] = kNone; s_goal[q] = inf_f(); s_cut[q] = inf_f(); s_tie[q] = kNone; s_armed[q] = 0u;
    if ((uint32_t)q < np) {
      const Plan& P = plans[p0 + q];
      s_seed[q] = P.seed[0];
      const uint32_t tg = P.target[0];
      float dt;
      if (BLOCKED) { const uint2 a = B.vaddr[tg]; dt = B.D[(size_t)a.x * B.NP + (size_t)(p0 + q) * (a.y >> 8) + (a.y & 255u)]; }
      else dt = P.dist[tg];
      s_armed[q] = dt < inf_f() ? 1u : 0u;
      const GoalCut gc = goal_cut(dt, P.offset, tg);                 // dijkstra :296
      s_goal[q] = gc.goal; s_cut[q] = gc.cut; s_tie[q] = gc.tie;
    }
  }
  uint32_t bad = 0;
  __syncthreads();
  const uint32_t t_beg = T.t_lo + blockIdx.y * tiles_per_block;
  const uint32_t t_end = min(t_beg + tiles_per_block, T.t_hi ? T.t_hi : T.ntiles);
  for (uint32_t t = t_beg; t < t_end; ++t) {
    if (!BLOCKED) {                                                   // (PG == 1 there) uniform over the workgroup
      MNAV_GLOBAL const float* g_tlast = as_global((const float*)T.tlast);
      MNAV_GLOBAL const uint32_t* g_p0 = as_global((const uint32_t*)T.pend[0]);
      MNAV_GLOBAL const uint32_t* g_p1 = as_global((const uint32_t*)T.pend[1]);
      if (!(g_tlast[t] > -inf_f()) && g_p0[t] == kInfBits && g_p1[t] == kInfBits) continue;
    }
    const uint32_t v0 = g_vptr[t], nv = g_vptr[t + 1] - v0;
    const uint32_t h0 = g_hptr[t], nh = g_hptr[t + 1] - h0;
    const uint32_t e0 = g_eptr[t], ne = g_eptr[t + 1] - e0;
    const uint32_t r0 = g_rptr[t], nl = nv + nh;
    __syncthreads();                                               // the previous tile's LDS image is dead
    uint32_t g[kFinVpt];
    u32x2 va[kFinVpt];
#pragma unroll
    for (int k = 0; k < kFinVpt; ++k) {
      const uint32_t i = tid + k * kTileBlock;
      g[k] = (i < nv) ? g_verts[v0 + i] : (i < nl ? g_halo_verts[h0 + i - nv] : 0u);
      if (BLOCKED) va[k] = g_va[g[k]];
      if (i < nl) lgid[i] = g[k];
    }
    for (uint32_t i = tid + kFinVpt * kTileBlock; i < nl; i += kTileBlock) lgid[i] = (i < nv) ? g_verts[v0 + i] : g_halo_verts[h0 + i - nv];
    stage_tile_graph(T, L, e0, ne, r0, nl, tid);                   // (ends with a barrier: lgid is visible too)
    uint32_t dbn[kFinVpt];                                           // values of the first plan of the group
#pragma unroll
    for (int k = 0; k < kFinVpt; ++k) {
      if (BLOCKED) dbn[k] = f2u(g_D[(size_t)va[k].x * B.NP + (size_t)p0 * (va[k].y >> 8) + (va[k].y & 255u)]);
      else dbn[k] = f2u(as_global(plans[p0].dist)[g[k]]);
    }
#pragma unroll 1
    for (uint32_t q = 0; q < np; ++q) {
      const uint32_t p = p0 + q;
      const Plan& P = plans[p];
      MNAV_GLOBAL float* g_dist = as_global(P.dist);
      MNAV_GLOBAL uint32_t* g_pred = as_global(P.pred);
      GoalCut gcut; gcut.goal = s_goal[q]; gcut.cut = s_cut[q]; gcut.tie = s_tie[q];
      const float goal_dist = gcut.cut;                              // values above it are re-derived from the expanded sources
      const uint32_t seed_q = s_seed[q];
      MNAV_GLOBAL float* g_vm = (BLOCKED && B.vecmaps) ? as_global(B.vecmaps[p]) : nullptr;
      auto value_of = [&](uint32_t gid) -> uint32_t {
        if (BLOCKED) { const u32x2 a = g_va[gid]; return f2u(g_D[(size_t)a.x * B.NP + (size_t)p * (a.y >> 8) + (a.y & 255u)]); }
        return f2u(g_dist[gid]);
      };
      if (q) __syncthreads();                                        // the previous plan's ldu / lsum / lkey are dead
      uint32_t db[kFinVpt];
#pragma unroll
      for (int k = 0; k < kFinVpt; ++k) db[k] = dbn[k];
      if (BLOCKED && q + 1 < np) {                                   // the next plan's values are in flight during this plan's passes
#pragma unroll
        for (int k = 0; k < kFinVpt; ++k) dbn[k] = f2u(g_D[(size_t)va[k].x * B.NP + (size_t)(p + 1) * (va[k].y >> 8) + (va[k].y & 255u)]);
      }
      {
        // no reached vertex among the tile's own and halo vertices: nothing to derive here (dist = inf, pred = itself)
        int reached = 0;
#pragma unroll
        for (int k = 0; k < kFinVpt; ++k) reached |= ((uint32_t)(tid + k * kTileBlock) < nl && db[k] != kInfBits) ? 1 : 0;
        for (uint32_t i = tid + kFinVpt * kTileBlock; i < nl; i += kTileBlock) reached |= (value_of(lgid[i]) != kInfBits) ? 1 : 0;
        if (!__syncthreads_or(reached)) {                             // uniform over the workgroup
          if (BLOCKED) for (uint32_t i = tid; i < nv; i += kTileBlock) {
            const uint32_t gg = lgid[i]; g_dist[gg] = inf_f(); g_pred[gg] = gg;
            if (g_vm) store3((float*)g_vm + 3 * (size_t)gg, 0.f, 0.f, 0.f);
          }
          continue;
        }
      }
      int cut = 0;                                                   // owned vertices above goal_dist: their value is re-derived
#pragma unroll
      for (int k = 0; k < kFinVpt; ++k) {
        const uint32_t i = tid + k * kTileBlock;
        if (i < nl) {
          L.ldu[i] = db[k];
          if (i < nv) { const bool c = u2f(db[k]) > goal_dist; cut |= c; lsum[i] = c ? kInfBits : db[k]; lkey[i] = ~0ull; }
        }
      }
      for (uint32_t i = tid + kFinVpt * kTileBlock; i < nl; i += kTileBlock) {
        const uint32_t b = value_of(lgid[i]);
        L.ldu[i] = b;
        if (i < nv) { const bool c = u2f(b) > goal_dist; cut |= c; lsum[i] = c ? kInfBits : b; lkey[i] = ~0ull; }
      }
      cut = __syncthreads_or(cut);
      if (cut) {
        // pass 1 (tiles on the cut-off boundary only): smallest sum offered to the vertices above goal_dist
        for (uint32_t x = tid; x < nl; x += kTileBlock) {
          const float dx = u2f(L.ldu[x]);
          if (!expanded_source(gcut, dx, lgid[x])) continue;         // not expanded (dijkstra :293-300)
          for (uint32_t e = L.lrow[x], ee = L.lrow[x + 1]; e < ee; ++e) {
            const uint32_t y = L.lcol[e];
            if (y < nv && u2f(L.ldu[y]) > goal_dist) atomicMin(&lsum[y], f2u(dx + L.lw[e]));   // dijkstra :331
          }
        }
        __syncthreads();
      }
      // pass 2: every edge checks the fixed point (no expanded source may offer less than the target
      // holds) and the edges that attain the value compete with (d[x], x) for the predecessor
      for (uint32_t x = tid; x < nl; x += kTileBlock) {
        const uint32_t dxb = L.ldu[x];
        const uint32_t eb = L.lrow[x], ee = L.lrow[x + 1];
        const float dx = u2f(dxb);
        if (!expanded_source(gcut, dx, lgid[x])) continue;
        const unsigned long long key = ((unsigned long long)dxb << 32) | lgid[x];
        uint32_t y[8]; float w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { y[u] = L.lcol[eb + u]; w[u] = L.lw[eb + u]; }   // reads past the row stay inside the LDS image
        uint32_t sy[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const bool ok = eb + u < ee && y[u] < nv; y[u] = ok ? y[u] : 0xFFFFFFFFu; sy[u] = ok ? lsum[y[u]] : 0u; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (y[u] == 0xFFFFFFFFu) continue;
          const uint32_t sb = f2u(dx + w[u]);
          if (sb < sy[u]) ++bad;                                      // fixed point violated: internal error
          else if (sb == sy[u] && sb != kInfBits) atomicMin(&lkey[y[u]], key);
        }
        for (uint32_t e = eb + 8; e < ee; ++e) {                     // valence > 8: rare
          const uint32_t yy = L.lcol[e];
          if (yy >= nv) continue;
          const uint32_t sb = f2u(dx + L.lw[e]), syy = lsum[yy];
          if (sb < syy) ++bad;
          else if (sb == syy && sb != kInfBits) atomicMin(&lkey[yy], key);
        }
      }
      __syncthreads();
      uint32_t cnt = 0;
      for (uint32_t i = tid; i < nv; i += kTileBlock) {
        const uint32_t gg = lgid[i];
        if (gg == seed_q) {
          ++cnt;
          if (BLOCKED) { g_dist[gg] = u2f(L.ldu[i]); g_pred[gg] = gg; if (g_vm) store3((float*)g_vm + 3 * (size_t)gg, 0.f, 0.f, 0.f); }
          continue;
        }
        const uint32_t sb = lsum[i], ob = L.ldu[i];
        const unsigned long long key = lkey[i];
        if (sb != kInfBits && key == ~0ull && (!T.owned || T.owned[gg])) ++bad;   // a finite value no expanded neighbour supports (a halo copy's support may live on another process)
        const uint32_t pv = (sb != kInfBits) ? (uint32_t)key : gg;
        g_pred[gg] = pv;
        if (BLOCKED || sb != ob) g_dist[gg] = u2f(sb);                // (else only above goal_dist: tentative value, dijkstra :337-343)
        if (sb != kInfBits) ++cnt;
        if (BLOCKED && g_vm) {                                        // k_vecmap_dijkstra's arithmetic
          float x = 0.f, y = 0.f, z = 0.f;
          if (pv != gg) {                                             // :197
            x = g_xyz[3 * (size_t)pv] - g_xyz[3 * (size_t)gg];        // :204
            y = g_xyz[3 * (size_t)pv + 1] - g_xyz[3 * (size_t)gg + 1];
            z = g_xyz[3 * (size_t)pv + 2] - g_xyz[3 * (size_t)gg + 2];
            const float len = sqrtf(x * x + y * y + z * z);           // normalized(), :206
            x = x / len; y = y / len; z = z / len;
          }
          store3((float*)g_vm + 3 * (size_t)gg, x, y, z);
        }
      }
      cnt = wave_sum(cnt);
      if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&s_settled[q], cnt);
    }
  }
  bad = wave_sum(bad);
  if ((threadIdx.x & 63) == 0 && bad) atomicAdd(mismatch, bad);
  __syncthreads();
  if ((uint32_t)tid < np) {
    const int q = tid;
    if (s_settled[q]) atomicAdd(&res[p0 + q].settled, (unsigned long long)s_settled[q]);
    if (blockIdx.y == 0) {
      const Plan& P = plans[p0 + q];
      Ctl r; memset(&r, 0, sizeof(r));
      r.armed = s_armed[q]; r.goal_dist = s_goal[q]; r.thr = inf_f(); r.thr_fixed = inf_f();
      if (BLOCKED) { r.it = (int32_t)*B.iters; r.done = 1u; r.overflow = (*B.err || *B.n_cand) ? 1u : 0u; }
      else {
        const TilePlan& Tq = tplans[p0 + q];
        const TCtl a = Tq.ctl[0], b = Tq.ctl[1];
        const TCtl last = (a.it > b.it) ? a : b;
        r.it = last.it; r.done = last.done; r.bands = last.sweeps; r.evals = last.acts; r.overflow = last.pad[0];
      }
      P.ctl[0] = r; P.ctl[1] = r;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// plan state initialisation (dijkstra :266-270, cvp :710-714) and seeding (:272-277, :719-728)
// ---------------------------------------------------------------------------------------------
template <uint32_t PLANNER>
__global__ __launch_bounds__(kBlock) void k_init(const Plan* __restrict__ plans)
{
  const Plan& P = plans[blockIdx.y];
  const uint32_t stride = gridDim.x * kBlock;
  for (uint32_t v = blockIdx.x * kBlock + threadIdx.x; v < P.V; v += stride) {
    P.dist[v] = inf_f();
    P.pred[v] = v;
    if (P.stamp) { P.stamp[v] = 0u; P.dirty[v] = 0u; P.wstamp[v] = 0u; }   // work-list state of the band steps only
    if (PLANNER == kPlannerCvp) { P.tkey[v] = key_inf(); P.dirn[v] = 0.0f; P.cutf[v] = kNone; if (P.keyd) P.keyd[v] = inf_f(); }
  }
}

template <uint32_t PLANNER>
__global__ void k_seed(const Plan* __restrict__ plans)
{
  const Plan& P = plans[blockIdx.x];
  if (threadIdx.x != 0) return;
  constexpr int ns = (PLANNER == kPlannerCvp) ? 3 : 1;
  float m0 = inf_f();
  for (int k = 0; k < ns; ++k) {
    const uint32_t s = P.seed[k];
    P.dist[s] = P.seed_d[k];
    if (PLANNER == kPlannerCvp) { P.tkey[s] = make_key(P.seed_d[k], s); P.cutf[s] = P.seed_face; }
    m0 = fminf(m0, P.seed_d[k]);
  }
  uint32_t n = 0;
  uint32_t* l0 = P.list[0];
  for (int k = 0; k < ns; ++k) {
    const uint32_t s = P.seed[k];
    if (PLANNER == kPlannerCvp) {
      for (uint32_t i = P.crn_ptr[s]; i < P.crn_ptr[s + 1]; ++i) {
        const Corner c = P.crn[i];
        if (c.v1 == kNone) continue;
        if (P.stamp[c.v1] != 0xFFFFFFFFu) { P.stamp[c.v1] = 0xFFFFFFFFu; if (n < P.cap) l0[n] = c.v1; ++n; }
        if (P.stamp[c.v2] != 0xFFFFFFFFu) { P.stamp[c.v2] = 0xFFFFFFFFu; if (n < P.cap) l0[n] = c.v2; ++n; }
      }
    } else {
      for (uint32_t i = P.row_ptr[s]; i < P.row_ptr[s + 1]; ++i) {
        const uint32_t u = P.nbr[i].u;
        if (P.stamp[u] != 0xFFFFFFFFu) { P.stamp[u] = 0xFFFFFFFFu; if (n < P.cap) l0[n] = u; ++n; }
      }
    }
  }
  Ctl c0; memset(&c0, 0, sizeof(c0));
  c0.it = -1; c0.n = 0; c0.thr_fixed = -inf_f(); c0.goal_dist = inf_f();
  c0.thr = m0 + P.delta; if (!(c0.thr > m0)) c0.thr = next_up(m0);
  for (int k = 0; k < ns; ++k) if (!(P.seed_d[k] < c0.thr)) c0.thr = next_up(P.seed_d[k]);   // the first band holds every seed
  c0.band_new = 1; c0.width = P.delta; c0.wmin = inf_f(); c0.epoch = 1;
  P.ctl[1] = c0;
  P.ctl[0] = c0;
  Cnt ci; memset(&ci, 0, sizeof(ci)); ci.n_next = n; ci.changed = 1; ci.minkey = 0x7f800000u; ci.minchg = 0x7f800000u;
  P.cnt[2] = ci;                       // read by step 0 as "(0-1) mod 3"
  Cnt z; memset(&z, 0, sizeof(z)); z.minkey = 0x7f800000u; z.minchg = 0x7f800000u;
  P.cnt[0] = z; P.cnt[1] = z; P.cnt[3] = z;                         // cnt[3]: sticky flags (mnav_eval.h kFlag*)
}

// ---------------------------------------------------------------------------------------------
// result assembly: code + vertex path (dijkstra :358-373) / reachability (cvp :902-918), stats
// ---------------------------------------------------------------------------------------------

constexpr uint32_t kPathOverflow = 0xFFFFFFF0u;   // internal: the path row was too short; path_len then holds the FULL length and
                                                  // the host walks the overflowed plans again into exact-size rows
// Where the vertex path of plan k goes: rows of `stride` ids, or -- second pass, for the plans whose path did not fit --
// rows of exactly the needed size in a packed buffer (off / cap per plan, cap 0 = plan not part of this pass).
struct PathRows {
  uint32_t* base; uint32_t stride;
  const unsigned long long* off; const uint32_t* cap;
  __device__ __forceinline__ uint32_t* row(uint32_t k) const { return base + (off ? (size_t)off[k] : (size_t)k * stride); }
  __device__ __forceinline__ uint32_t capacity(uint32_t k) const { return cap ? cap[k] : stride; }
  __device__ __forceinline__ bool skip(uint32_t k) const { return cap && cap[k] == 0u; }
};

template <uint32_t PLANNER>
__global__ void k_finish(const Plan* __restrict__ plans, PlanResult* __restrict__ res, PathRows rows)
{
  const Plan& P = plans[blockIdx.x];
  if (threadIdx.x != 0 || rows.skip(blockIdx.x)) return;
  PlanResult& R = res[blockIdx.x];
  const Ctl a = P.ctl[0], b = P.ctl[1];
  const Ctl last = (a.it > b.it) ? a : b;
  R.steps = (uint32_t)(last.it < 0 ? 0 : last.it);
  R.bands = last.bands; R.armed = last.armed; R.overflow = last.overflow; R.goal_dist = last.goal_dist; R.shrinks = last.shrinks | (last.cuts << 16);
  R.evals = last.evals;
  R.path_len = 0;
  uint32_t code = kSuccess;
  if (PLANNER == kPlannerCvp) {                                     // k_cvp_verify
    if (P.cnt[3].n_next & kFlagWalkLimit) R.overflow |= 8u;         // cascade-tree walk bound hit on the converged tree
    if (P.cnt[3].changed) R.overflow |= 16u;                        // a vertex is not a fixed point of the gather rule
  }
  if (R.overflow || !last.done) code = kInternalError;
  else if (PLANNER == kPlannerDijkstra) {
    const uint32_t seed = P.seed[0], target = P.target[0];
    if (P.pred[target] == target) code = kNoPathFound;             // dijkstra :358
    else {
      uint32_t* path = rows.row(blockIdx.x);                       // written target-side first
      const uint32_t cap = rows.capacity(blockIdx.x);
      uint32_t n = 0, v = target;
      while (v != seed && n <= P.V) { v = P.pred[v]; if (n < cap) path[n] = v; ++n; }   // :369-373; the full length is counted
      if (v != seed) code = kInternalError;                         // a predecessor cycle
      else if (n > cap) code = kPathOverflow;                       // row too short: the host walks this plan again into an exact row
      R.path_len = n;
    }
  } else {
    bool any = false;
    for (int k = 0; k < 3; ++k) any = any || (P.pred[P.target[k]] != P.target[k]);   // cvp :904-911
    if (!any && !(P.target[0] == P.seed[0] && P.target[1] == P.seed[1] && P.target[2] == P.seed[2]))
      code = kNoPathFound;                                                           // :912-918
  }
  R.code = code;
}

// Paths without the finalize pass.  When a caller only wants the vertex path (no potential, predecessors or vector map:
// the batch bench, mbf_mesh_nav's getPath), k_dij_finalize -- which re-stages every touched tile to derive ALL
// predecessors and the tentative values beyond goal_dist -- is 10 % of a batch for nothing: after the tile rounds
// every vertex with dist <= goal_dist is final (its shortest paths only use such sources), the path only visits such
// vertices, and a path vertex's predecessor is the argmin (dist[u] + w, dist[u], u) over its neighbours of eval_dijkstra,
// computed here on the fly along the walk (one wave per plan, one neighbour per lane).  Every hop also checks that the
// minimum IS the vertex's distance (the fixed-point property k_dij_finalize verifies everywhere; here along the path).
__global__ __launch_bounds__(kWave) void k_path_lazy(const Plan* __restrict__ plans, const TilePlan* __restrict__ tplans, PlanResult* __restrict__ res,
                                                     PathRows rows, uint32_t* __restrict__ mismatch)
{
  if (rows.skip(blockIdx.x)) return;
  const Plan& P = plans[blockIdx.x];
  const TilePlan& T = tplans[blockIdx.x];
  const int lane = threadIdx.x;
  PlanResult& R = res[blockIdx.x];
  const TCtl a = T.ctl[0], b = T.ctl[1];
  const TCtl last = (a.it > b.it) ? a : b;
  const uint32_t seed = P.seed[0], target = P.target[0];
  const float dt = P.dist[target];
  const GoalCut gcut = goal_cut(dt, P.offset, target);
  const float goal_dist = gcut.goal;
  uint32_t code = kSuccess, n = 0, bad = 0;
  if (last.pad[0] || !last.done) code = kInternalError;               // activation cap hit / not finished
  else if (!(dt < inf_f())) code = kNoPathFound;                      // the target was never reached (dijkstra :358)
  else {
    uint32_t* path = rows.row(blockIdx.x);                            // written target-side first
    const uint32_t cap = rows.capacity(blockIdx.x);
    uint32_t v = target;
    while (v != seed && n <= P.V) {
      const float dv = P.dist[v];
      float best_s = inf_f(), best_du = inf_f();
      uint32_t best_u = v;
      const uint32_t beg = P.row_ptr[v], end = P.row_ptr[v + 1];
      for (uint32_t i = beg + lane; i < end; i += kWave) {
        const Nbr nb = P.nbr[i];
        const float du = P.dist[nb.u];
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
      v = best_u;
      if (lane == 0 && n < cap) path[n] = v;
      ++n;
    }
    if (bad || v != seed) code = kInternalError;
    else if (n > cap) code = kPathOverflow;
  }
  if (lane == 0) {
    R.code = code; R.path_len = (code == kSuccess || code == kPathOverflow) ? n : 0;
    R.steps = (uint32_t)(last.it < 0 ? 0 : last.it); R.bands = last.sweeps; R.armed = (dt < inf_f()) ? 1u : 0u; R.overflow = last.pad[0];
    R.goal_dist = goal_dist; R.evals = last.acts; R.shrinks = 0;
    if (bad) atomicAdd(mismatch, 1u);
  }
}

// vertex paths of a batch, packed back to back and turned into the reference's list order (seed ... pred[target]) on the
// device: ONE dense copy to a pinned buffer instead of a strided 2-D copy of n rows
__global__ __launch_bounds__(kBlock) void k_pack_paths(PathRows rows, PathRows over, const uint32_t* __restrict__ offs,
                                                       const uint32_t* __restrict__ lens, uint32_t* __restrict__ out)
{
  const uint32_t k = blockIdx.x, len = lens[k];
  const uint32_t* src = (over.cap && over.cap[k]) ? over.row(k) : rows.row(k);   // second-pass rows where the first ones were too short
  uint32_t* dst = out + offs[k];
  for (uint32_t q = threadIdx.x; q < len; q += kBlock) dst[q] = src[len - 1 - q];
}

// settled vertices of a lazily finished plan: the popped ones, dist <= goal_dist (conservative against k_dij_finalize's
// count, which includes the tentative ring beyond goal_dist)
__global__ __launch_bounds__(kBlock) void k_count_goal(const Plan* __restrict__ plans, PlanResult* __restrict__ res)
{
  const Plan& P = plans[blockIdx.y];
  const float dt = P.dist[P.target[0]];
  const float goal_dist = goal_cut(dt, P.offset, P.target[0]).cut;
  uint32_t c = 0;
  const uint32_t stride = gridDim.x * kBlock;
  for (uint32_t v = blockIdx.x * kBlock + threadIdx.x; v < P.V; v += stride) { const float d = P.dist[v]; c += (d < inf_f() && d <= goal_dist) ? 1u : 0u; }
  c = wave_sum(c);
  __shared__ uint32_t s_c[kBlock / 64];
  if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tot = 0;
    for (int k = 0; k < kBlock / 64; ++k) tot += s_c[k];
    if (tot) atomicAdd(&res[blockIdx.y].settled, (unsigned long long)tot);
  }
}

__global__ __launch_bounds__(kBlock) void k_count(const Plan* __restrict__ plans, PlanResult* __restrict__ res)
{
  const Plan& P = plans[blockIdx.y];
  uint32_t c = 0;
  const uint32_t stride = gridDim.x * kBlock;
  for (uint32_t v = blockIdx.x * kBlock + threadIdx.x; v < P.V; v += stride) c += (P.dist[v] < inf_f()) ? 1u : 0u;
  c = wave_sum(c);
  __shared__ uint32_t s_c[kBlock / 64];
  if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tot = 0;
    for (int k = 0; k < kBlock / 64; ++k) tot += s_c[k];
    if (tot) atomicAdd(&res[blockIdx.y].settled, (unsigned long long)tot);
  }
}

// ---------------------------------------------------------------------------------------------
// vector maps: dijkstra :189-209, cvp :204-239
// ---------------------------------------------------------------------------------------------

__global__ __launch_bounds__(kBlock) void k_vecmap_dijkstra(const Plan* __restrict__ plans, const float* __restrict__ xyz,
                                                            float* const* __restrict__ vecmaps)
{
  const Plan& P = plans[blockIdx.y];
  float* vm = vecmaps[blockIdx.y];
  const uint32_t stride = gridDim.x * kBlock;
  for (uint32_t v = blockIdx.x * kBlock + threadIdx.x; v < P.V; v += stride) {
    const uint32_t p = P.pred[v];
    float x = 0.f, y = 0.f, z = 0.f;
    if (p != v) {                                               // :197
      x = xyz[3 * (size_t)p] - xyz[3 * (size_t)v];              // :204
      y = xyz[3 * (size_t)p + 1] - xyz[3 * (size_t)v + 1];
      z = xyz[3 * (size_t)p + 2] - xyz[3 * (size_t)v + 2];
      const float len = sqrtf(x * x + y * y + z * z);           // normalized(), :206
      x = x / len; y = y / len; z = z / len;
    }
    store3(vm + 3 * (size_t)v, x, y, z);
  }
}

__global__ __launch_bounds__(kBlock) void k_vecmap_cvp(const Plan* __restrict__ plans, const float* __restrict__ xyz,
                                                       const float* __restrict__ nrm, float* const* __restrict__ vecmaps,
                                                       const float* __restrict__ seed_pos)
{
  const Plan& P = plans[blockIdx.y];
  float* vm = vecmaps[blockIdx.y];
  const uint32_t stride = gridDim.x * kBlock;
  for (uint32_t v = blockIdx.x * kBlock + threadIdx.x; v < P.V; v += stride) {
    const uint32_t p = P.pred[v];
    float x = 0.f, y = 0.f, z = 0.f;
    if (is_seed(P, v)) {                                        // cvp :722-724 (un-normalised diff)
      const float* sp = seed_pos + 3 * (size_t)blockIdx.y;
      x = sp[0] - xyz[3 * (size_t)v]; y = sp[1] - xyz[3 * (size_t)v + 1]; z = sp[2] - xyz[3 * (size_t)v + 2];
    } else if (p != v && P.cutf[v] != kNone) {                  // :218, :222-225
      const float dx = xyz[3 * (size_t)p] - xyz[3 * (size_t)v];
      const float dy = xyz[3 * (size_t)p + 1] - xyz[3 * (size_t)v + 1];
      const float dz = xyz[3 * (size_t)p + 2] - xyz[3 * (size_t)v + 2];
      const float nx = nrm[3 * (size_t)v], ny = nrm[3 * (size_t)v + 1], nz = nrm[3 * (size_t)v + 2];
      // rotated(normal, direction) :234 -- Rodrigues (CONVENTION, lvr2 un-vendored; see oracle)
      const float ang = P.dirn[v];
      const float c = cosf_ref(ang), s = sinf_ref(ang);   // the host libm's bits (mnav_eval.h): the field is the reference's bit for bit
      const float cx = ny * dz - nz * dy, cy = nz * dx - nx * dz, cz = nx * dy - ny * dx;
      const float ndv = nx * dx + ny * dy + nz * dz;
      const float k = ndv * (1.0f - c);
      x = dx * c + cx * s + nx * k; y = dy * c + cy * s + ny * k; z = dz * c + cz * s + nz * k;
      const float len = sqrtf(x * x + y * y + z * z);           // :236
      x = x / len; y = y / len; z = z / len;
    }
    store3(vm + 3 * (size_t)v, x, y, z);
  }
}

// ---------------------------------------------------------------------------------------------
// input preparation
// ---------------------------------------------------------------------------------------------
// MeshMap::computeEdgeWeights, mesh_map.cpp:517-561 (exact promotion order, no contraction)
// Combination layers on the device (mesh_layers/src/combination_layer.cpp:44-85 Max, :185-248 weighted
// sum): the inputs are dense V-sized layers (missing entries already replaced by the layer default,
// :62-65 / :201-205), combined in the order given, starting from defaultValue() = 0.
__global__ __launch_bounds__(kBlock) void k_combine(uint32_t V, int mode, uint32_t n_layers, const float* __restrict__ layers,
                                                    const float* __restrict__ weights, float* __restrict__ out)
{
  const uint32_t v = blockIdx.x * kBlock + threadIdx.x;
  if (v >= V) return;
  float cost = 0.0f;                                               // defaultValue(), combination_layer.h:52,94
  for (uint32_t l = 0; l < n_layers; ++l) {
    const float tmp = layers[(size_t)l * V + v];
    if (mode == 0) cost = (cost < tmp) ? tmp : cost;               // std::max(cost, tmp) :66
    else cost += weights[l] * tmp;                                 // :206 (float multiply, float add)
  }
  out[v] = cost;
}

__global__ __launch_bounds__(kBlock) void k_edge_weights(uint32_t E, const uint32_t* __restrict__ edge_vtx,
                                                         const float* __restrict__ edge_dist, const float* __restrict__ cost,
                                                         double factor, float* __restrict__ w)
{
  const uint32_t e = blockIdx.x * kBlock + threadIdx.x;
  if (e >= E) return;
  const float c1 = cost[edge_vtx[2 * (size_t)e]], c2 = cost[edge_vtx[2 * (size_t)e + 1]];   // :528-529
  if (isinf(c1) || isinf(c2)) { w[e] = inf_f(); return; }                                    // :538-542
  const float vd = edge_dist[e];                                                             // :548
  const float edge_cost = (float)((double)(vd * (c1 + c2)) / 2.0);                           // :550
  w[e] = (float)((double)vd + factor * (double)edge_cost);                                   // :552
}

// Incremental cost change (MeshMap::layerChanged mesh_map.cpp:454-493 + updateEdgeWeights :563-618): the changed
// vertices get their new cost, then only the edges around them are re-weighted -- same expressions as :550-552.
__global__ __launch_bounds__(kBlock) void k_scatter_costs(uint32_t n, const uint32_t* __restrict__ ids, const float* __restrict__ values,
                                                          float* __restrict__ cost)
{
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) cost[ids[i]] = values[i];
}
__global__ __launch_bounds__(kBlock) void k_update_edge_weights(uint32_t n, const uint32_t* __restrict__ ids, const uint32_t* __restrict__ row_ptr,
                                                                const uint32_t* __restrict__ nbr_u, const uint32_t* __restrict__ nbr_e,
                                                                const float* __restrict__ edge_dist, const float* __restrict__ cost,
                                                                double factor, float* __restrict__ w)
{
  // 8 lanes per changed vertex, one incident edge each (an edge between two changed vertices is written twice with
  // the same value)
  const uint32_t i = (blockIdx.x * kBlock + threadIdx.x) >> 3;
  const int sub = threadIdx.x & 7;
  if (i >= n) return;
  const uint32_t v = ids[i];
  const float c1 = cost[v];
  for (uint32_t k = row_ptr[v] + sub; k < row_ptr[v + 1]; k += 8) {
    const uint32_t e = nbr_e[k];
    const float c2 = cost[nbr_u[k]];
    if (isinf(c1) || isinf(c2)) { w[e] = inf_f(); continue; }       // :596-600
    const float vd = edge_dist[e];                                   // :606
    const float edge_cost = (float)((double)(vd * (c1 + c2)) / 2.0); // :608 (float sum: commutative, the endpoint order is free)
    w[e] = (float)((double)vd + factor * (double)edge_cost);        // :610
  }
}

// gather CSR for Dijkstra: {u, w(u,v)}; w=+inf when v is invalid (:328) or u is over the cost
// limit (:302, u would be popped but never expanded)
__global__ __launch_bounds__(kBlock) void k_build_nbr(uint32_t V, const uint32_t* __restrict__ row_ptr,
                                                      const uint32_t* __restrict__ nbr_u, const uint32_t* __restrict__ nbr_e,
                                                      const float* __restrict__ w, const float* __restrict__ cost,
                                                      const uint8_t* __restrict__ invalid, double cost_limit, Nbr* __restrict__ out)
{
  const uint32_t v = blockIdx.x * kBlock + threadIdx.x;
  if (v >= V) return;
  const bool vinv = invalid[v] != 0;
  for (uint32_t i = row_ptr[v]; i < row_ptr[v + 1]; ++i) {
    const uint32_t u = nbr_u[i];
    float ww = w[nbr_e[i]];
    if (vinv || (double)cost[u] > cost_limit) ww = inf_f();
    Nbr n; n.u = u; n.w = ww;
    out[i] = n;
  }
}

struct CornerIdx { uint32_t v1, v2, ea, eb, ec, face; };

__global__ __launch_bounds__(kBlock) void k_build_crn(uint32_t V, const uint32_t* __restrict__ crn_ptr,
                                                      const CornerIdx* __restrict__ idx, const float* __restrict__ w,
                                                      const float* __restrict__ cost, const uint8_t* __restrict__ invalid,
                                                      double cost_limit, Corner* __restrict__ out, uint8_t* __restrict__ blocked)
{
  const uint32_t v = blockIdx.x * kBlock + threadIdx.x;
  if (v >= V) return;
  const bool vinv = invalid[v] != 0;
  blocked[v] = ((double)cost[v] >= cost_limit || vinv) ? 1 : 0;      // cvp :802,825,848 / :785
  for (uint32_t i = crn_ptr[v]; i < crn_ptr[v + 1]; ++i) {
    const CornerIdx k = idx[i];
    Corner c;
    c.v1 = (vinv || invalid[k.v1] || invalid[k.v2]) ? kNone : k.v1;   // cvp :785
    c.v2 = k.v2; c.a = w[k.ea]; c.b = w[k.eb]; c.c = w[k.ec]; c.face = k.face;
    out[i] = c;
  }
}

// ---------------------------------------------------------------------------------------------
// Layers on the device (mesh_layers): Steepness (steepness_layer.cpp:157-166, :82-93), Inflation
// (inflation_layer.cpp:341-491 as a multi-source wave on the band engine; spec: mnav_eval.h eval_cvp / Plan.seed_mask)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_edge_dist(uint32_t E, const uint32_t* __restrict__ edge_vtx, const float* __restrict__ xyz,
                                                      float* __restrict__ out)
{
  const uint32_t e = blockIdx.x * kBlock + threadIdx.x;
  if (e >= E) return;
  const float* a = xyz + 3 * (size_t)edge_vtx[2 * (size_t)e];
  const float* b = xyz + 3 * (size_t)edge_vtx[2 * (size_t)e + 1];
  const float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  out[e] = sqrtf(dx * dx + dy * dy + dz * dz);                     // lvr2 BaseVector::distanceFrom in float (mesh_map.cpp:347)
}

__global__ __launch_bounds__(kBlock) void k_steepness(uint32_t V, const float* __restrict__ nrm, double threshold,
                                                      float* __restrict__ cost, uint8_t* __restrict__ lethal)
{
  const uint32_t v = blockIdx.x * kBlock + threadIdx.x;
  if (v >= V) return;
  const float st = acosf_ref(nrm[3 * (size_t)v + 2]);              // :165 (float overload of acos; the host libm's bits, mnav_eval.h)
  cost[v] = st;
  lethal[v] = ((double)st > threshold) ? 1 : 0;                    // :88
}

// corners with the edge DISTANCES as side lengths (waveCostInflation reads map->edgeDistances() :383); no face is
// skipped here: what may fire is decided by Plan.seed_mask
__global__ __launch_bounds__(kBlock) void k_build_crn_infl(uint32_t V, const uint32_t* __restrict__ crn_ptr,
                                                           const CornerIdx* __restrict__ idx, const float* __restrict__ w,
                                                           Corner* __restrict__ out)
{
  const uint32_t v = blockIdx.x * kBlock + threadIdx.x;
  if (v >= V) return;
  for (uint32_t i = crn_ptr[v]; i < crn_ptr[v + 1]; ++i) {
    const CornerIdx k = idx[i];
    Corner c;
    c.v1 = k.v1; c.v2 = k.v2; c.a = w[k.ea]; c.b = w[k.eb]; c.c = w[k.ec]; c.face = corner_face_for_inflation(k.face);
    out[i] = c;
  }
}

__global__ __launch_bounds__(kBlock) void k_infl_mask(uint32_t V, const uint8_t* __restrict__ lethal, const uint8_t* __restrict__ invalid,
                                                      uint8_t* __restrict__ mask)
{
  const uint32_t v = blockIdx.x * kBlock + threadIdx.x;
  if (v >= V) return;
  const bool l = lethal[v] != 0, inv = invalid && invalid[v] != 0;
  mask[v] = l ? (inv ? kInflSeedMute : kInflSeed) : (inv ? kInflMute : kInflFree);
}

// control blocks of the wave (the single-thread part of k_seed), then the seeds in parallel: every lethal vertex is
// fixed at distance 0 (:397-402) and the free vertices around it form the first work list
__global__ void k_infl_ctl(const Plan* __restrict__ plans)
{
  const Plan& P = plans[0];
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  Ctl c0; memset(&c0, 0, sizeof(c0));
  c0.it = -1; c0.n = 0; c0.thr_fixed = -inf_f(); c0.goal_dist = inf_f();
  c0.thr = P.delta; if (!(c0.thr > 0.0f)) c0.thr = next_up(0.0f);
  c0.band_new = 1; c0.width = P.delta; c0.wmin = inf_f(); c0.epoch = 1;
  P.ctl[1] = c0;
  P.ctl[0] = c0;
  Cnt ci; memset(&ci, 0, sizeof(ci)); ci.changed = 1; ci.minkey = 0x7f800000u; ci.minchg = 0x7f800000u;
  P.cnt[2] = ci;                       // read by step 0 as "(0-1) mod 3"; k_infl_seed counts the list into it
  Cnt z; memset(&z, 0, sizeof(z)); z.minkey = 0x7f800000u; z.minchg = 0x7f800000u;
  P.cnt[0] = z; P.cnt[1] = z; P.cnt[3] = z;
}

__global__ __launch_bounds__(kBlock) void k_infl_seed(const Plan* __restrict__ plans)
{
  const Plan& P = plans[0];
  const uint32_t stride = gridDim.x * kBlock;
  uint32_t* l0 = P.list[0];
  for (uint32_t v = blockIdx.x * kBlock + threadIdx.x; v < P.V; v += stride) {
    if (!is_seed(P, v)) continue;
    P.dist[v] = 0.0f; P.tkey[v] = make_key(0.0f, v); P.keyd[v] = 0.0f;
    for (uint32_t i = P.crn_ptr[v]; i < P.crn_ptr[v + 1]; ++i) {
      const Corner c = P.crn[i];
      const uint32_t nb[2] = { c.v1, c.v2 };
      for (int q = 0; q < 2; ++q) {
        const uint32_t u = nb[q];
        if (u == kNone || is_seed(P, u)) continue;
        if (P.stamp[u] != 0xFFFFFFFFu && atomicExch(&P.stamp[u], 0xFFFFFFFFu) != 0xFFFFFFFFu) {
          const uint32_t at = atomicAdd(&P.cnt[2].n_next, 1u);
          if (at < P.cap) l0[at] = u;
        }
      }
    }
  }
}

// The inflation layer's repulsive vector field from the converged wave (spec: mnav_eval.h infl_accumulate / infl_assign).
// state: 0 = open (a free vertex with a distance whose vector may still be assigned), 1 = final with a vector, 2 = final
// without one.  k_infl_assign is launched until nothing is open; it reads the states of the PREVIOUS launch and writes the
// next ones to a second array, so a vector is only ever read after the launch that wrote it has ended.
__global__ __launch_bounds__(kBlock) void k_infl_accum(const Plan* __restrict__ plans, const uint32_t* __restrict__ crn_walk,
                                                       const float* __restrict__ xyz, float* __restrict__ vec, uint8_t* __restrict__ state,
                                                       uint8_t* __restrict__ acc, uint32_t* __restrict__ ctl)
{
  const Plan& P = plans[0];
  const uint32_t v = blockIdx.x * kBlock + threadIdx.x;
  if (v >= P.V) return;
  float o[3] = { 0.f, 0.f, 0.f };
  const int r = infl_accumulate(P, crn_walk, xyz, v, o);
  if (r < 0) { atomicOr(&ctl[2], 1u); return; }
  vec[3 * (size_t)v] = o[0]; vec[3 * (size_t)v + 1] = o[1]; vec[3 * (size_t)v + 2] = o[2];
  acc[v] = r == 1 ? 1 : 0;
  const bool open = !is_seed(P, v) && P.dist[v] < inf_f();
  state[v] = open ? 0 : (r == 1 ? 1 : 2);
}

__global__ __launch_bounds__(kBlock) void k_infl_assign(const Plan* __restrict__ plans, float* __restrict__ vec, const uint8_t* __restrict__ state,
                                                        uint8_t* __restrict__ state_next, const uint8_t* __restrict__ acc, uint32_t* __restrict__ ctl)
{
  const Plan& P = plans[0];
  const uint32_t v = blockIdx.x * kBlock + threadIdx.x;
  if (v >= P.V) return;
  const uint8_t st = state[v];
  if (st != 0) { state_next[v] = st; return; }
  float o[3];
  const int r = infl_assign(P, vec, state, v, o);
  if (r == 2) { state_next[v] = 0; atomicAdd(&ctl[0], 1u); return; }   // a support is still open: next launch
  if (r == 1) { vec[3 * (size_t)v] = o[0]; vec[3 * (size_t)v + 1] = o[1]; vec[3 * (size_t)v + 2] = o[2]; }
  state_next[v] = (r == 1 || acc[v]) ? 1 : 2;
  atomicAdd(&ctl[1], 1u);
}

// riskiness from the distances: fading() :315-339; vertices the wave never reached keep the default 0
// (inflation_layer.h:74-77).  The exponential runs in float64 and is rounded to float32 (:326).
__global__ __launch_bounds__(kBlock) void k_infl_cost(uint32_t V, const float* __restrict__ dist, double inflation_radius,
                                                      double inscribed_radius, double inscribed_value, double lethal_value,
                                                      double cost_scaling_factor, float* __restrict__ cost)
{
  const uint32_t v = blockIdx.x * kBlock + threadIdx.x;
  if (v >= V) return;
  const float d = dist[v];
  float c;
  if (!(d < inf_f())) c = 0.0f;
  else if ((double)d > inflation_radius) c = 0.0f;                                        // :317-320
  else if ((double)d > inscribed_radius) {                                                // :323
    const float factor = (float)exp(-1.0 * cost_scaling_factor * ((double)d - inscribed_radius));   // :326
    c = (float)(inscribed_value * (double)factor);                                        // :327
  }
  else if (d > 0) c = (float)inscribed_value;                                             // :332-335
  else c = (float)lethal_value;                                                           // :338
  cost[v] = c;
}

__global__ __launch_bounds__(kBlock) void k_combine_resident(uint32_t V, int mode, uint32_t n_layers, const float* const* __restrict__ layers,
                                                             const float* __restrict__ weights, float* __restrict__ out)
{
  const uint32_t v = blockIdx.x * kBlock + threadIdx.x;
  if (v >= V) return;
  float cost = 0.0f;                                               // defaultValue(), combination_layer.h:52,94
  for (uint32_t l = 0; l < n_layers; ++l) {
    const float tmp = layers[l][v];
    if (mode == 0) cost = (cost < tmp) ? tmp : cost;               // std::max(cost, tmp) :66
    else cost += weights[l] * tmp;                                 // :206
  }
  out[v] = cost;
}

// CombinationLayer::onInputChanged (combination_layer.cpp:87-147 max, :250-302 weighted sum): only the changed vertices
__global__ __launch_bounds__(kBlock) void k_combine_resident_ids(uint32_t n, const uint32_t* __restrict__ ids, int mode, uint32_t n_layers,
                                                                 const float* const* __restrict__ layers, const float* __restrict__ weights,
                                                                 float* __restrict__ out, float* __restrict__ values)
{
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const uint32_t v = ids[i];
  float cost = 0.0f;
  for (uint32_t l = 0; l < n_layers; ++l) {
    const float tmp = layers[l][v];
    if (mode == 0) cost = (tmp < cost) ? cost : tmp;               // std::max(tmp, cost) :117
    else cost += weights[l] * tmp;                                 // :281
  }
  out[v] = cost;
  values[i] = cost;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
}  // namespace
#include "mnav_tb.h"
#include "mnav_walk.h"

// One back-tracking job: the plan's resident vector map and the two ends of the walk.
struct WalkJob { const float* vecmap; float seed[3]; uint32_t seed_face; float target[3]; uint32_t target_face; };

// CVPMeshPlanner's back-tracking (cvp_mesh_planner.cpp:920-951) on the resident vector map: one wave per plan.  The loop
// is a dependent chain (each step needs the position the previous one produced), so all 64 lanes walk in lockstep on the
// same data (uniform loads, uniform branches) and divide only the scan of searchNeighbourFaces' face list, which lives in
// LDS.  ctl[2*j] = walk status, ctl[2*j+1] = entries written (walk order, target first).
__global__ __launch_bounds__(64) void k_backtrack(WalkMesh M, WalkInflation L, const WalkJob* __restrict__ jobs, double step_width, uint32_t cap,
                                                  float* __restrict__ pos_out, uint32_t* __restrict__ face_out, int32_t* __restrict__ ctl)
{
  __shared__ uint32_t list[kWalkScratchWords];
  const uint32_t j = blockIdx.x;
  const WalkJob J = jobs[j];
  if (!J.vecmap) { if (threadIdx.x == 0) { ctl[2 * j] = 0; ctl[2 * j + 1] = 0; } return; }   // no plan behind this row
  WalkField Fd;
  Fd.vecmap = J.vecmap;
  for (int k = 0; k < 3; ++k) Fd.seed_vs[k] = M.faces[3 * (size_t)J.seed_face + k];
  uint32_t n = 0;
  const int st = walk_backtrack(M, Fd, L, w3(J.seed[0], J.seed[1], J.seed[2]), J.seed_face, w3(J.target[0], J.target[1], J.target[2]), J.target_face,
                                step_width, cap, pos_out + 3 * (size_t)cap * j, face_out + (size_t)cap * j, &n, list);
  ctl[2 * j] = st; ctl[2 * j + 1] = (int32_t)n;
}
namespace {

struct Slot {
  float *dist = nullptr, *dirn = nullptr, *vecmap = nullptr;
  PopKey* tkey = nullptr;
  uint32_t *pred = nullptr, *cutf = nullptr, *stamp = nullptr, *dirty = nullptr, *list0 = nullptr, *list1 = nullptr;
  uint32_t *wlist0 = nullptr, *wlist1 = nullptr, *wstamp = nullptr;
  Ctl* ctl = nullptr;
  Cnt* cnt = nullptr;
  bool cvp_ready = false, band_ready = false;
  // tiled engine
  uint32_t *tpend0 = nullptr, *tpend1 = nullptr;
  float* tlast = nullptr;
  TCtl* tctl = nullptr;
  TCnt* tcnt = nullptr;
  bool tile_ready = false;
};

}  // namespace

struct mnav_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  std::atomic<int> cancel{ 0 };
  uint32_t* d_cancel = nullptr;            // the same flag in device memory: long-running kernels poll it (agent-scope load)
  uint32_t* h_one = nullptr;               // pinned source word (1) for the copy mnav_cancel issues on its own stream
  hipStream_t cancel_stream = nullptr;
  // host copies needed for seeding
  uint32_t V = 0, F = 0, E = 0;
  std::vector<float> h_xyz, h_cost;
  std::vector<uint32_t> h_row_ptr, h_nbr_u;   // gather CSR (host copy): the tile-batch engine builds its streams from it on first use
  TbState tb; tb::Args tb_args{};             // tile-batch SSSP engine (mnav_tb.h); arguments of the last batch
  bool tb_args_valid = false;                 // ... which still describe live device memory (last tile-batch call succeeded, nothing freed since)
  bool want_vec = false;               // the running call asked for vector maps (lazy 12 B/vertex/plan)
  bool resident_vecmap = false;        // mnav_set_resident_outputs: always compute the vector map, leave it on the device
  std::vector<uint32_t> caller_slot;   // plan index of the caller's batch -> device slot of the last call (kNone: never ran)
  std::vector<uint32_t> h_faces;
  std::vector<uint32_t> h_vf_ptr, h_vf;    // getFacesOfVertex rows (host copy; uploaded on the first device back-tracking call)
  uint32_t *d_faces = nullptr, *d_vf_ptr = nullptr, *d_vf = nullptr; bool walk_mesh_valid = false;
  float* d_walk_pos = nullptr; uint32_t* d_walk_face = nullptr; size_t walk_cap = 0;   // k_backtrack outputs: rows of `cap` entries
  hipEvent_t ev_link[2]{};                 // stream links of the asynchronous shard calls (caller's stream <-> ours)
  struct WalkJob* d_walk_jobs = nullptr; int32_t* d_walk_ctl = nullptr; uint32_t walk_jobs_cap = 0;
  std::vector<uint8_t> h_invalid;
  bool have_mesh = false, have_costs = false, have_normals = false;
  // device mesh
  uint32_t *d_row_ptr = nullptr, *d_nbr_u = nullptr, *d_nbr_e = nullptr, *d_crn_ptr = nullptr, *d_edge_vtx = nullptr;
  CornerIdx* d_crn_idx = nullptr;
  uint32_t* d_crn_walk = nullptr;                                   // HostTopology::crn_walk (inflation vector field)
  float *d_xyz = nullptr, *d_nrm = nullptr, *d_cost = nullptr, *d_w = nullptr, *d_edge_dist = nullptr;
  uint8_t* d_invalid = nullptr;
  // materialised per cost_limit
  Nbr* d_nbr = nullptr; double nbr_limit = NAN; bool nbr_valid = false;
  Corner* d_crn = nullptr; uint8_t* d_blocked = nullptr; double crn_limit = NAN; bool crn_valid = false;
  FaceCirculation circ;                                            // caller-supplied getFacesOfVertex rows (optional)
  uint32_t* d_verify_any = nullptr; uint32_t verify_sweeps_used = 0;
  bool cvp_verify = true;                                          // k_cvp_verify after every CVP plan (MNAV_CVP_VERIFY=0 to skip)
  int walk_max = kKeyWalkMax, descend_max = kDescendWalkMax;       // cascade-tree walk bounds (MNAV_KEY_WALK_MAX / MNAV_DESCEND_WALK_MAX: tests)
  // plans
  std::vector<Slot> slots;
  Plan* d_plans = nullptr; uint32_t plans_cap = 0;
  PlanResult* d_res = nullptr; PlanResult* h_res = nullptr;
  float** d_vecptrs = nullptr;
  uint32_t* d_paths = nullptr; size_t paths_words = 0; uint32_t path_stride = 0;   // n plans x path_stride vertex ids
  uint32_t* d_over = nullptr; unsigned long long* d_over_off = nullptr; uint32_t* d_over_cap = nullptr;   // exact rows of the paths that did not fit
  uint32_t *d_pack = nullptr, *h_pack = nullptr, *d_pack_meta = nullptr; size_t pack_words = 0, pack_meta_n = 0;   // packed paths (device, pinned host), offsets + lengths
  std::unordered_map<void*, size_t> alloc_bytes;                   // sizes of the dev_upload buffers (re-used when unchanged)
  Ctl* h_ctl = nullptr;       // pinned, 2 per plan
  float* d_seed_pos = nullptr; uint32_t seed_pos_cap = 1;
  std::map<uint64_t, hipGraphExec_t> graphs;
  // tiled SSSP engine
  int dij_engine = 3;          // 0 tiled rounds, 1 band steps, 2 persistent per-plan, 3 auto, 5 tile-batch, 6 asynchronous tiles (opt-in)
  int last_engine = 0;
  uint32_t persistent_min_batch = 128;
  bool lazy_paths = false;      // this call only wants vertex paths: k_path_lazy instead of k_dij_finalize + k_finish
  bool allow_lazy_paths = true; // MNAV_LAZY_PATHS=0: always finalize (predecessors / tentative values for everybody)
  uint32_t max_steps = 1u << 20;   // per plan; set from the mesh size at upload (a wavefront needs O(diameter) steps)
  double max_wall_s = 120.0;   // host-side guard: a plan that takes longer is abandoned with an error
  uint32_t tile_size = 512;    // 4 workgroups of the tile kernels per CU (36 KB LDS each)
  float rounds_band_mult = 4.0f;   // the round engine (latency) prefers wider bands than the persistent one
  float tile_band_user = 0.f, tile_band_auto = 1.f;
  uint32_t* d_t_rptr = nullptr;
  HostTiles tiles_meta;        // only the small per-tile vectors are kept (vert_tile, sizes)
  uint32_t *d_t_vptr = nullptr, *d_t_verts = nullptr, *d_t_hptr = nullptr, *d_t_halo_verts = nullptr, *d_t_halo_tile = nullptr,
           *d_t_eptr = nullptr, *d_t_src = nullptr, *d_vert_tile = nullptr, *d_mismatch = nullptr;
  uint16_t *d_t_rowptr = nullptr, *d_t_col = nullptr;
  float* d_t_tw = nullptr; bool tw_valid = false; uint32_t t_nnz = 0;
  // sharded single plan (mnav_shard_*)
  struct Shard {
    bool ready = false, active = false;
    uint32_t rank = 0, world = 1, t_lo = 0, t_hi = 0, n_iface = 0, rounds_per_exchange = 8, j = 0;
    uint32_t seed = 0, target = 0; double offset = 0.3;
    uint32_t *d_iface_vert = nullptr, *d_wake_ptr = nullptr, *d_wake_tile = nullptr, *d_changed = nullptr, *d_minpend = nullptr;
    uint8_t* d_iface_owner = nullptr;
    std::vector<uint32_t> iface_vert;
    bool partition = false; uint8_t* d_owned = nullptr;              // mnav_shard_setup_partition
    uint32_t* d_walk = nullptr; uint32_t walk_cap = 0;               // mnav_shard_walk
    std::map<std::array<uint64_t, 3>, hipGraphExec_t> graphs;        // captured exchange sequences (shard_replay)
  } shard;
  double edge_cost_factor = 0.0;                                   // factor of the resident edge weights (mnav_update_costs)
  // layers computed / kept on the device (mnav_layer_*)
  struct Layer { float* cost = nullptr; uint8_t* lethal = nullptr; float* dist = nullptr; float* vec = nullptr; uint8_t* vstate = nullptr;   // vstate: 3 x V (two state arrays + the accumulate flags)
                 bool ready = false, have_vec = false;
                 double inflation_radius = 0, inscribed_radius = 0, inscribed_value = 0, lethal_value = 0; };   // InflationLayer config (vectorAt reads it)
  std::vector<Layer> layers;
  Corner* d_crn_infl = nullptr; bool crn_infl_valid = false;       // corners over the edge distances (inflation wave)
  uint8_t *d_infl_mask = nullptr, *d_zero_u8 = nullptr;
  float* d_infl_keyd = nullptr;
  uint32_t infl_steps = 0, infl_bands = 0; uint64_t infl_evals = 0; float infl_ms = 0.f, infl_ms_wave = 0.f;   // last inflation wave
  TilePlan* d_tplans = nullptr; uint32_t tplans_cap = 0;
  TCtl* h_tctl = nullptr;
  size_t tile_lds = 0, fin_lds = 0;
  bool use_graph = true;
  uint32_t* d_wide_prefix = nullptr; WideSched* d_wide_sched = nullptr; uint32_t wide_cap = 0;   // k_cvp_ctl -> k_step_wide
  float* d_vec3 = nullptr;                                           // mnav_vector_at after a paths-only batch
  uint32_t wide_groups = 1; hipStream_t stream_g[kWideGroupsMax] = {}; hipEvent_t ev_fork[kWideGroupsMax] = {};   // CVP batches in groups on their own streams ([0] unused / fork event)
  uint32_t cvp_wide_min_batch = 32;                                  // CVP batches of at least this many plans run k_step_wide
  float delta_user = 0.f, delta_auto = 0.f;
  uint32_t last_planner = 0, last_n = 0;
  std::vector<uint32_t> last_target; double last_offset = 0.0;   // Dijkstra: robot vertex per device slot, goal_dist_offset of the last call
  mnav_stats stats{};
  uint64_t algo_bytes = 0;
  hipEvent_t ev[8]{};
  hipEvent_t evc[2]{};         // bracket one graph replay (chunk of step/round launches)
  double ms_chunks = 0.0;      // sum of the bracketed chunk durations of the last call
  Ctl* d_ctl_pool = nullptr; uint32_t ctl_pool_cap = 0;      // contiguous control blocks: one D2H copy per chunk
  TCtl* d_tctl_pool = nullptr; uint32_t tctl_pool_cap = 0;
};

namespace {

#define MTRACE(msg) do { if (getenv("MNAV_TRACE")) { fprintf(stderr, "[mnav] %9.3f ms %s:%d %s\n", 1e-3 * (double)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(), __func__, __LINE__, msg); fflush(stderr); } } while (0)
#define HIPCHK(call)                                                                               \
  do {                                                                                             \
    hipError_t e_ = (call);                                                                        \
    if (e_ != hipSuccess) {                                                                        \
      ctx->err = std::string(#call) + ": " + hipGetErrorString(e_) + " (mnav.hip:" + std::to_string(__LINE__) + ")"; \
      return -1;                                                                                   \
    }                                                                                              \
  } while (0)

// temporary device buffer of one call: freed on every way out (the HIPCHK early returns included)
template <class T>
struct DevTmp {
  T* p = nullptr;
  DevTmp() = default;
  DevTmp(const DevTmp&) = delete;
  DevTmp& operator=(const DevTmp&) = delete;
  ~DevTmp() { if (p) (void)hipFree(p); }
  operator T*() const { return p; }
  void** out() { return (void**)&p; }
};

template <class T>
int dev_upload(mnav_ctx* ctx, T** dptr, const T* host, size_t n)
{
  const size_t bytes = sizeof(T) * (n ? n : 1) + 64;               // tail slack: clamped vector loads may touch element 0 of an empty tile
  auto it = *dptr ? ctx->alloc_bytes.find((void*)*dptr) : ctx->alloc_bytes.end();
  if (!*dptr || it == ctx->alloc_bytes.end() || it->second != bytes) {   // same size as last time (cost re-uploads): keep the buffer
    if (*dptr) { ctx->alloc_bytes.erase((void*)*dptr); (void)hipFree(*dptr); *dptr = nullptr; }
    HIPCHK(hipMalloc((void**)dptr, bytes));
    ctx->alloc_bytes[(void*)*dptr] = bytes;
  }
  if (n && host) HIPCHK(hipMemcpyAsync(*dptr, host, sizeof(T) * n, hipMemcpyHostToDevice, ctx->stream));
  return 0;
}

float ev_ms(hipEvent_t a, hipEvent_t b);

void drop_layers(mnav_ctx* ctx)
{
  for (auto& L : ctx->layers) { (void)hipFree(L.cost); (void)hipFree(L.lethal); (void)hipFree(L.dist); (void)hipFree(L.vec); (void)hipFree(L.vstate); }
  ctx->layers.clear();
  (void)hipFree(ctx->d_crn_infl); (void)hipFree(ctx->d_infl_mask); (void)hipFree(ctx->d_zero_u8); (void)hipFree(ctx->d_infl_keyd);
  ctx->d_crn_infl = nullptr; ctx->d_infl_mask = nullptr; ctx->d_zero_u8 = nullptr; ctx->d_infl_keyd = nullptr; ctx->crn_infl_valid = false;
}

void free_slot(Slot& s)
{
  (void)hipFree(s.dist); (void)hipFree(s.tkey); (void)hipFree(s.dirn); (void)hipFree(s.vecmap);
  (void)hipFree(s.pred); (void)hipFree(s.cutf); (void)hipFree(s.stamp); (void)hipFree(s.dirty); (void)hipFree(s.list0); (void)hipFree(s.list1);
  (void)hipFree(s.wlist0); (void)hipFree(s.wlist1); (void)hipFree(s.wstamp);
  (void)hipFree(s.cnt);
  (void)hipFree(s.tpend0); (void)hipFree(s.tpend1); (void)hipFree(s.tlast); (void)hipFree(s.tcnt);
  s = Slot{};
}

void drop_graphs(mnav_ctx* ctx)
{
  for (auto& kv : ctx->graphs) (void)hipGraphExecDestroy(kv.second);
  ctx->graphs.clear();
  for (auto& kv : ctx->shard.graphs) (void)hipGraphExecDestroy(kv.second);
  ctx->shard.graphs.clear();
}

int ensure_plan_tables(mnav_ctx* ctx, uint32_t n);

int ensure_slots(mnav_ctx* ctx, uint32_t n, bool cvp, bool band, bool vec)
{
  const size_t V = ctx->V ? ctx->V : 1;
  while (ctx->slots.size() < n) {
    Slot s;
    HIPCHK(hipMalloc((void**)&s.dist, 4 * V)); HIPCHK(hipMalloc((void**)&s.pred, 4 * V));   // 8 B per vertex and plan ...
    HIPCHK(hipMalloc((void**)&s.cnt, 4 * sizeof(Cnt)));                                       // 3 rotating + sticky flags
    ctx->slots.push_back(s);
  }
  for (uint32_t i = 0; i < n; ++i) {                                   // ... the rest only for the paths that use it
    Slot& s = ctx->slots[i];
    if (band && !s.band_ready) {                                       // work lists of the band/gather steps
      HIPCHK(hipMalloc((void**)&s.stamp, 4 * V)); HIPCHK(hipMalloc((void**)&s.dirty, 4 * V));
      HIPCHK(hipMalloc((void**)&s.list0, 4 * V)); HIPCHK(hipMalloc((void**)&s.list1, 4 * V));
      HIPCHK(hipMalloc((void**)&s.wlist0, 4 * V)); HIPCHK(hipMalloc((void**)&s.wlist1, 4 * V)); HIPCHK(hipMalloc((void**)&s.wstamp, 4 * V));
      s.band_ready = true;
    }
    if (vec && !s.vecmap) HIPCHK(hipMalloc((void**)&s.vecmap, 12 * V));
  }
  if (cvp)
    for (uint32_t i = 0; i < n; ++i) {
      Slot& s = ctx->slots[i];
      if (!s.cvp_ready) {
        HIPCHK(hipMalloc((void**)&s.tkey, sizeof(PopKey) * V)); HIPCHK(hipMalloc((void**)&s.dirn, 4 * V));
        HIPCHK(hipMalloc((void**)&s.cutf, 4 * V));
        s.cvp_ready = true;
      }
    }
  if (ctx->ctl_pool_cap < n) {
    if (ctx->d_ctl_pool) (void)hipFree(ctx->d_ctl_pool);
    ctx->d_ctl_pool = nullptr;
    HIPCHK(hipMalloc((void**)&ctx->d_ctl_pool, 2 * sizeof(Ctl) * n));
    ctx->ctl_pool_cap = n;
  }
  for (uint32_t i = 0; i < n; ++i) ctx->slots[i].ctl = ctx->d_ctl_pool + 2 * i;
  return ensure_plan_tables(ctx, n);
}

// per-plan tables shared by all engines (plan descriptors, results)
int ensure_plan_tables(mnav_ctx* ctx, uint32_t n)
{
  if (ctx->plans_cap < n) {
    if (ctx->d_plans) (void)hipFree(ctx->d_plans);
    if (ctx->d_res) (void)hipFree(ctx->d_res);
    if (ctx->h_res) (void)hipHostFree(ctx->h_res);
    if (ctx->h_ctl) (void)hipHostFree(ctx->h_ctl);
    if (ctx->d_vecptrs) (void)hipFree(ctx->d_vecptrs);
    ctx->d_plans = nullptr; ctx->d_res = nullptr; ctx->h_res = nullptr; ctx->h_ctl = nullptr; ctx->d_vecptrs = nullptr;
    drop_graphs(ctx);   // graphs captured the old d_plans pointer
    HIPCHK(hipMalloc((void**)&ctx->d_plans, sizeof(Plan) * n));
    HIPCHK(hipMalloc((void**)&ctx->d_res, sizeof(PlanResult) * n));
    HIPCHK(hipHostMalloc((void**)&ctx->h_res, sizeof(PlanResult) * n, hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void**)&ctx->h_ctl, sizeof(Ctl) * 2 * n, hipHostMallocDefault));
    HIPCHK(hipMalloc((void**)&ctx->d_vecptrs, sizeof(float*) * n));
    ctx->plans_cap = n;
  }
  return 0;
}

// Vertex paths of a batch: n rows of `stride` ids.  A path has ~1.4 sqrt(V) hops on a terrain, so rows of
// 16 sqrt(V) + 1024 ids hold it with a wide margin (84 MB instead of 20 GB for 5120 plans on the 1M mesh); a path
// that does not fit makes the path kernels report kPathOverflow with the full length, and only those plans are walked
// again into rows of exactly that size.
uint32_t default_path_stride(const mnav_ctx* ctx)
{
  const double s = 16.0 * std::sqrt((double)ctx->V) + 1024.0;
  return (uint32_t)std::min<double>(s, (double)(ctx->V ? ctx->V : 1));
}
int ensure_paths(mnav_ctx* ctx, uint32_t n, uint32_t stride)
{
  const size_t words = (size_t)n * (stride ? stride : 1);
  if (ctx->paths_words < words) {
    if (ctx->d_paths) (void)hipFree(ctx->d_paths);
    ctx->d_paths = nullptr; ctx->paths_words = 0;
    HIPCHK(hipMalloc((void**)&ctx->d_paths, sizeof(uint32_t) * words));
    ctx->paths_words = words;
  }
  ctx->path_stride = stride;
  return 0;
}
int ensure_paths(mnav_ctx* ctx, uint32_t n) { return ensure_paths(ctx, n, default_path_stride(ctx)); }

uint32_t blocks_per_plan(const mnav_ctx* ctx)
{
  // the work list of a planar mesh is O(sqrt(V)) long; 8 entries per wave and round.  Measured on the 1M mesh (CVP,
  // MI355X): 500 waves per plan are as fast as 1500 for a single plan (41.8 vs 41.4 ms) and 25 % faster in a batch of
  // 128 (225 vs 178 plans/s: fewer idle waves to dispatch per step); 256 waves cost 5 % / 10 %.
  const double want = 4.0 * std::sqrt((double)ctx->V) / kGroupsPerWave;
  uint32_t g = (uint32_t)std::ceil(want);
  if (const char* e = getenv("MNAV_BLOCKS_PER_PLAN")) g = (uint32_t)atoi(e);
  if (g < 4) g = 4;
  if (g > 4096) g = 4096;
  return g;
}

template <uint32_t PLANNER>
int launch_steps(mnav_ctx* ctx, uint32_t n, uint32_t G, int count, bool wide)
{
  const bool fork = wide && ctx->wide_groups > 1;
  if (fork) {                                                          // the other streams join (the capture of) the first
    HIPCHK(hipEventRecord(ctx->ev_fork[0], ctx->stream));
    for (uint32_t g = 1; g < ctx->wide_groups; ++g) HIPCHK(hipStreamWaitEvent(ctx->stream_g[g], ctx->ev_fork[0], 0));
  }
  for (int j = 0; j < count; ++j) {
    if (wide) {
      // per group of plans: controller + shares, the wide kernel with as many waves as stay resident, and the 8-lane kernel for the
      // plans in a band cut.  The groups run on two streams (two branches of the captured graph): a step of one group is ~0.6
      // rounds of the resident waves, so where one group's last round leaves the machine half empty the other group's step fills it
      uint32_t off = 0, poff = 0;
      for (uint32_t g = 0; g < ctx->wide_groups; ++g) {
        const uint32_t ng = (n * (g + 1u)) / ctx->wide_groups - off;
        hipStream_t st = g == 0 ? ctx->stream : ctx->stream_g[g];
        if (ng) {
          hipLaunchKernelGGL(k_cvp_ctl, dim3(1), dim3(256), 0, st, ctx->d_plans + off, ng, j % 6, ctx->d_wide_prefix + poff, ctx->d_wide_sched + g);
          hipLaunchKernelGGL(k_step_wide, dim3(G), dim3(kWave), 0, st, ctx->d_plans + off, ng, j % 6, ctx->d_wide_prefix + poff, ctx->d_wide_sched + g);
          hipLaunchKernelGGL(k_step_repair, dim3(blocks_per_plan(ctx), kRepairRows), dim3(kWave), 0, st, ctx->d_plans + off, j % 6,
                             ctx->d_wide_prefix + poff + ng + 1u, ctx->d_wide_sched + g);
        }
        off += ng; poff += 2u * ng + 2u;
      }
    }
    else hipLaunchKernelGGL(k_step<PLANNER>, dim3(G, n), dim3(kWave), 0, ctx->stream, ctx->d_plans, j % 6);
  }
  if (fork) {
    for (uint32_t g = 1; g < ctx->wide_groups; ++g) {
      HIPCHK(hipEventRecord(ctx->ev_fork[g], ctx->stream_g[g]));
      HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_fork[g], 0));
    }
  }
  HIPCHK(hipGetLastError());
  return 0;
}

template <uint32_t PLANNER>
int run_chunk(mnav_ctx* ctx, uint32_t n, uint32_t G, bool wide)
{
  if (!ctx->use_graph) return launch_steps<PLANNER>(ctx, n, G, getenv("MNAV_DEBUG_CHUNK") ? atoi(getenv("MNAV_DEBUG_CHUNK")) : kChunk, wide);   // debug: finer control-block trace
  const uint64_t key = ((uint64_t)PLANNER << 60) | ((uint64_t)(wide ? ctx->wide_groups : 0u) << 57) | ((uint64_t)n << 32) | G;
  auto it = ctx->graphs.find(key);
  if (it == ctx->graphs.end()) {
    hipGraph_t g = nullptr;
    HIPCHK(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    const int rc = launch_steps<PLANNER>(ctx, n, G, kChunk, wide);
    hipError_t e = hipStreamEndCapture(ctx->stream, &g);
    if (rc != 0 || e != hipSuccess) { ctx->err = "graph capture failed"; return -1; }
    hipGraphExec_t ge = nullptr;
    HIPCHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    (void)hipGraphDestroy(g);
    it = ctx->graphs.emplace(key, ge).first;
  }
  HIPCHK(hipGraphLaunch(it->second, ctx->stream));
  return 0;
}

int materialize(mnav_ctx* ctx, bool cvp, double cost_limit)
{
  const uint32_t V = ctx->V;
  const uint32_t gb = (V + kBlock - 1) / kBlock;
  if (!cvp) {
    if (ctx->nbr_valid && ctx->nbr_limit == cost_limit) return 0;
    if (!ctx->d_nbr) HIPCHK(hipMalloc((void**)&ctx->d_nbr, sizeof(Nbr) * (size_t)(ctx->E ? 2 * (size_t)ctx->E : 1)));
    hipLaunchKernelGGL(k_build_nbr, dim3(gb ? gb : 1), dim3(kBlock), 0, ctx->stream, V, ctx->d_row_ptr, ctx->d_nbr_u,
                       ctx->d_nbr_e, ctx->d_w, ctx->d_cost, ctx->d_invalid, cost_limit, ctx->d_nbr);
    HIPCHK(hipGetLastError());
    ctx->nbr_limit = cost_limit; ctx->nbr_valid = true; ctx->tw_valid = false; ctx->tb.w_valid = false;
  } else {
    if (ctx->crn_valid && ctx->crn_limit == cost_limit) return 0;
    if (!ctx->d_crn) HIPCHK(hipMalloc((void**)&ctx->d_crn, sizeof(Corner) * (size_t)(ctx->F ? 3 * (size_t)ctx->F : 1)));
    if (!ctx->d_blocked) HIPCHK(hipMalloc((void**)&ctx->d_blocked, V ? V : 1));
    hipLaunchKernelGGL(k_build_crn, dim3(gb ? gb : 1), dim3(kBlock), 0, ctx->stream, V, ctx->d_crn_ptr, ctx->d_crn_idx,
                       ctx->d_w, ctx->d_cost, ctx->d_invalid, cost_limit, ctx->d_crn, ctx->d_blocked);
    HIPCHK(hipGetLastError());
    ctx->crn_limit = cost_limit; ctx->crn_valid = true;
  }
  return 0;
}

// k_cvp_verify until a sweep finds every vertex at its fixed point: sweeps that find stale deep-cascade members store
// the re-evaluated state (verify_entry) and are followed by another one; the last allowed sweep only checks.
int verify_sweeps(mnav_ctx* ctx, uint32_t n)
{
  if (!ctx->d_verify_any) HIPCHK(hipMalloc((void**)&ctx->d_verify_any, 4));
  uint32_t gv = (ctx->V / kGroupsPerWave + 3) / 4;                  // ~4 vertices per 8-lane group
  if (gv < 1) gv = 1;
  if (gv > 8192) gv = 8192;
  ctx->verify_sweeps_used = 0;
  for (int sweep = 0; sweep <= kVerifySweeps; ++sweep) {
    uint32_t any = 0;
    HIPCHK(hipMemsetAsync(ctx->d_verify_any, 0, 4, ctx->stream));
    hipLaunchKernelGGL(k_flags_reset, dim3(n), dim3(64), 0, ctx->stream, ctx->d_plans);
    hipLaunchKernelGGL(k_cvp_verify, dim3(gv, n), dim3(kWave), 0, ctx->stream, ctx->d_plans, sweep < kVerifySweeps ? 1 : 0, ctx->d_verify_any);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(&any, ctx->d_verify_any, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (!any) break;
    ++ctx->verify_sweeps_used;
  }
  if (getenv("MNAV_TRACE")) fprintf(stderr, "[mnav] verification: %u fixing sweep(s)\n", ctx->verify_sweeps_used);
  return 0;
}

struct PlanIn {
  uint32_t seed[3], target[3];
  float seed_d[3];
  uint32_t seed_face;
  uint32_t seed_expands[3], target_expands[3];
};

// Runs n plans of one planner to completion on the device.  Returns 0, -1 (error) or 1 (cancelled).
template <uint32_t PLANNER>
int run_plans(mnav_ctx* ctx, uint32_t n, const std::vector<PlanIn>& in, double offset, bool want_path)
{
  constexpr bool cvp = PLANNER == kPlannerCvp;
  if (ensure_slots(ctx, n, cvp, true, cvp || ctx->want_vec)) return -1;
  if (want_path && ensure_paths(ctx, n)) return -1;
  // default band width: 3 mean edge weights for the Dijkstra gather steps, 12 for CVP (measured on C3:
  // fewer, fuller bands -- 20 % less time for one plan and for batches; results do not depend on it)
  const float delta = ctx->delta_user > 0.f ? ctx->delta_user : (cvp ? 4.0f * ctx->delta_auto : ctx->delta_auto);
  std::vector<Plan> hp(n);
  std::vector<float*> vecs(n);
  for (uint32_t i = 0; i < n; ++i) {
    Slot& s = ctx->slots[i];
    Plan& P = hp[i];
    memset(&P, 0, sizeof(P));
    P.planner = PLANNER; P.V = ctx->V;
    P.row_ptr = ctx->d_row_ptr; P.nbr = ctx->d_nbr; P.crn_ptr = ctx->d_crn_ptr; P.crn = ctx->d_crn; P.blocked = ctx->d_blocked;
    P.dist = s.dist; P.tkey = cvp ? s.tkey : nullptr; P.pred = s.pred; P.dirn = s.dirn; P.cutf = s.cutf; P.stamp = s.stamp; P.dirty = s.dirty;
    P.list[0] = s.list0; P.list[1] = s.list1; P.wlist[0] = s.wlist0; P.wlist[1] = s.wlist1; P.wstamp = s.wstamp; P.cap = ctx->V; P.ctl = s.ctl; P.cnt = s.cnt;
    P.delta = delta; P.offset = offset; P.max_steps = ctx->max_steps; P.walk_max = ctx->walk_max; P.descend_max = ctx->descend_max;
    for (int k = 0; k < 3; ++k) {
      P.seed[k] = in[i].seed[k]; P.target[k] = in[i].target[k]; P.seed_d[k] = in[i].seed_d[k];
      P.seed_expands[k] = in[i].seed_expands[k]; P.target_expands[k] = in[i].target_expands[k];
    }
    P.seed_face = in[i].seed_face;
    vecs[i] = s.vecmap;
  }
  HIPCHK(hipMemcpyAsync(ctx->d_plans, hp.data(), sizeof(Plan) * n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->d_vecptrs, vecs.data(), sizeof(float*) * n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->d_res, 0, sizeof(PlanResult) * n, ctx->stream));

  HIPCHK(hipEventRecord(ctx->ev[1], ctx->stream));
  uint32_t gi = (ctx->V + kBlock * 4 - 1) / (kBlock * 4);
  if (gi < 1) gi = 1;
  if (gi > 4096) gi = 4096;
  hipLaunchKernelGGL(k_init<PLANNER>, dim3(gi, n), dim3(kBlock), 0, ctx->stream, ctx->d_plans);
  hipLaunchKernelGGL(k_seed<PLANNER>, dim3(n), dim3(64), 0, ctx->stream, ctx->d_plans);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(ctx->ev[2], ctx->stream));

  // CVP batches: the wide step kernel (64 work-list entries per wave and round); single plans keep the 8-lane replay, whose
  // many small waves finish a short work list sooner
  bool wide = cvp && n >= ctx->cvp_wide_min_batch;
  if (const char* e = getenv("MNAV_CVP_WIDE")) wide = cvp && atoi(e) != 0;
  uint32_t G = blocks_per_plan(ctx);
  if (wide) {
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device);
    G = (kWideVerts == 64u ? 7u : 12u) * (uint32_t)ncu;               // waves of the whole batch, not per plan: what stays resident
    if (const char* e = getenv("MNAV_WIDE_WAVES")) G = (uint32_t)std::max(1, atoi(e));
    ctx->wide_groups = std::min(4u, std::max(1u, n / 40u));            // measured on the benched C3 configuration, plans/s with 1 / 2 / 3 / 4 / 8 groups:
                                                                      // 128 plans 277 / 315 / 320 / 320 / 221, 512 plans 330 / 425 / 463 / 468 / 422
    if (const char* e = getenv("MNAV_CVP_GROUPS")) ctx->wide_groups = (uint32_t)std::min(std::max(1, atoi(e)), (int)kWideGroupsMax);
    if (ctx->wide_groups > n) ctx->wide_groups = 1;
    if (ctx->wide_cap < n + 1u) {
      (void)hipFree(ctx->d_wide_prefix); ctx->d_wide_prefix = nullptr;
      HIPCHK(hipMalloc((void**)&ctx->d_wide_prefix, 4 * (size_t)(2u * n + 2u * kWideGroupsMax + 8u)));   // per group: prefix sums [ng + 1], then the list of plans in a band cut [ng]
      ctx->wide_cap = n + 1u;
      drop_graphs(ctx);                                               // (captured with the old pointer)
    }
    if (!ctx->d_wide_sched) HIPCHK(hipMalloc((void**)&ctx->d_wide_sched, kWideGroupsMax * sizeof(WideSched)));
    if (!ctx->stream_g[1]) {
      for (uint32_t g = 1; g < kWideGroupsMax; ++g) HIPCHK(hipStreamCreateWithFlags(&ctx->stream_g[g], hipStreamNonBlocking));
      for (auto& e : ctx->ev_fork) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
  }
  uint32_t launches = 0;
  int rc = 0;
  const auto t_start = std::chrono::steady_clock::now();
  ctx->ms_chunks = 0.0;
  for (;;) {
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() > ctx->max_wall_s) {
      ctx->err = "wavefront steps exceeded the wall-clock guard"; return -1;
    }
    HIPCHK(hipEventRecord(ctx->evc[0], ctx->stream));
    if (run_chunk<PLANNER>(ctx, n, G, wide)) return -1;
    HIPCHK(hipEventRecord(ctx->evc[1], ctx->stream));
    launches += kChunk;
    HIPCHK(hipMemcpyAsync(ctx->h_ctl, ctx->d_ctl_pool, 2 * sizeof(Ctl) * n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->ms_chunks += ev_ms(ctx->evc[0], ctx->evc[1]);
    bool all_done = true;
    for (uint32_t i = 0; i < n; ++i) {
      const Ctl& a = ctx->h_ctl[2 * i];
      const Ctl& b = ctx->h_ctl[2 * i + 1];
      const Ctl& last = a.it > b.it ? a : b;
      if (!last.done) all_done = false;
    }
    if (all_done) break;
    if (ctx->cancel.load(std::memory_order_relaxed)) { rc = 1; break; }
  }
  ctx->stats.launches = launches;
  if (cvp && rc == 0 && ctx->cvp_verify && verify_sweeps(ctx, n)) return -1;
  HIPCHK(hipEventRecord(ctx->ev[3], ctx->stream));
  return rc;
}

int ensure_tile_state(mnav_ctx* ctx, uint32_t n)
{
  const size_t nt = ctx->tiles_meta.ntiles ? ctx->tiles_meta.ntiles : 1;
  for (uint32_t i = 0; i < n; ++i) {
    Slot& s = ctx->slots[i];
    if (!s.tile_ready) {
      HIPCHK(hipMalloc((void**)&s.tpend0, 4 * nt)); HIPCHK(hipMalloc((void**)&s.tpend1, 4 * nt));
      HIPCHK(hipMalloc((void**)&s.tlast, 4 * nt));
      HIPCHK(hipMalloc((void**)&s.tcnt, 3 * sizeof(TCnt)));
      s.tile_ready = true;
    }
  }
  if (ctx->tctl_pool_cap < n) {
    if (ctx->d_tctl_pool) (void)hipFree(ctx->d_tctl_pool);
    ctx->d_tctl_pool = nullptr;
    HIPCHK(hipMalloc((void**)&ctx->d_tctl_pool, 2 * sizeof(TCtl) * n));
    ctx->tctl_pool_cap = n;
  }
  for (uint32_t i = 0; i < n; ++i) ctx->slots[i].tctl = ctx->d_tctl_pool + 2 * i;
  if (ctx->tplans_cap < n) {
    if (ctx->d_tplans) (void)hipFree(ctx->d_tplans);
    if (ctx->h_tctl) (void)hipHostFree(ctx->h_tctl);
    ctx->d_tplans = nullptr; ctx->h_tctl = nullptr;
    drop_graphs(ctx);
    HIPCHK(hipMalloc((void**)&ctx->d_tplans, sizeof(TilePlan) * n));
    HIPCHK(hipHostMalloc((void**)&ctx->h_tctl, sizeof(TCtl) * 2 * n, hipHostMallocDefault));
    ctx->tplans_cap = n;
  }
  if (!ctx->d_mismatch) HIPCHK(hipMalloc((void**)&ctx->d_mismatch, 4));
  return 0;
}

// one workgroup per (plan, chunk of tiles); small batches get more, smaller chunks to fill the chip
void launch_finalize(mnav_ctx* ctx, uint32_t n, uint32_t ntiles_in = 0, size_t fin_lds_in = 0)
{
  const uint32_t nt_ = ntiles_in ? ntiles_in : ctx->tiles_meta.ntiles;
  const size_t fin_lds = fin_lds_in ? fin_lds_in : ctx->fin_lds;
  const uint32_t ntiles = nt_ ? nt_ : 1u;
  uint32_t chunks = (4096u + n - 1) / n;                 // >= 4096 workgroups in flight
  if (chunks > ntiles) chunks = ntiles;
  if (chunks < 1) chunks = 1;
  const uint32_t per = (ntiles + chunks - 1) / chunks;
  chunks = (ntiles + per - 1) / per;
  hipLaunchKernelGGL((k_dij_finalize<1, false>), dim3(n, chunks), dim3(kTileBlock), fin_lds, ctx->stream, ctx->d_plans, ctx->d_tplans,
                     ctx->d_mismatch, ctx->d_res, per, n, FinBlocked{});
}

// the tile-batch engine's batches: groups of kFinGroup plans per staged tile, distances straight from the engine's slices
constexpr int kFinGroup = 8;
void launch_finalize_blocked(mnav_ctx* ctx, uint32_t n, const FinBlocked& B)
{
  const uint32_t ntiles = ctx->tiles_meta.ntiles ? ctx->tiles_meta.ntiles : 1u;
  const uint32_t groups = (n + kFinGroup - 1) / kFinGroup;
  uint32_t chunks = (8192u + groups - 1) / groups;
  if (chunks > ntiles) chunks = ntiles;
  if (chunks < 1) chunks = 1;
  const uint32_t per = (ntiles + chunks - 1) / chunks;
  chunks = (ntiles + per - 1) / per;
  hipLaunchKernelGGL((k_dij_finalize<kFinGroup, true>), dim3(groups, chunks), dim3(kTileBlock), ctx->fin_lds, ctx->stream, ctx->d_plans, ctx->d_tplans,
                     ctx->d_mismatch, ctx->d_res, per, n, B);
}

int tile_weights(mnav_ctx* ctx)
{
  if (ctx->tw_valid) return 0;
  const uint32_t n = ctx->t_nnz;
  const uint32_t gb = (n + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(k_tile_weights, dim3(gb ? gb : 1), dim3(kBlock), 0, ctx->stream, n, ctx->d_t_src, ctx->d_t_col, ctx->d_nbr, ctx->d_t_tw);
  HIPCHK(hipGetLastError());
  ctx->tw_valid = true;
  return 0;
}

constexpr int kTileChunk = 24;   // rounds per graph replay (multiple of 6)

int launch_tile_rounds(mnav_ctx* ctx, uint32_t n, uint32_t G, int count)
{
  for (int j = 0; j < count; ++j)
    hipLaunchKernelGGL(k_tile_round, dim3(G, n), dim3(kTileBlock), ctx->tile_lds, ctx->stream, ctx->d_tplans, j % 6);
  HIPCHK(hipGetLastError());
  return 0;
}

int run_tile_chunk(mnav_ctx* ctx, uint32_t n, uint32_t G)
{
  if (!ctx->use_graph) return launch_tile_rounds(ctx, n, G, kTileChunk);
  const uint64_t key = (7ull << 60) | ((uint64_t)n << 32) | G;
  auto it = ctx->graphs.find(key);
  if (it == ctx->graphs.end()) {
    hipGraph_t g = nullptr;
    HIPCHK(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    const int rc = launch_tile_rounds(ctx, n, G, kTileChunk);
    hipError_t e = hipStreamEndCapture(ctx->stream, &g);
    if (rc != 0 || e != hipSuccess) { ctx->err = "graph capture failed"; return -1; }
    hipGraphExec_t ge = nullptr;
    HIPCHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    (void)hipGraphDestroy(g);
    it = ctx->graphs.emplace(key, ge).first;
  }
  HIPCHK(hipGraphLaunch(it->second, ctx->stream));
  return 0;
}

// Dijkstra through the tiled engine.  Returns 0, -1 (error) or 1 (cancelled).
int run_dijkstra_tiled(mnav_ctx* ctx, uint32_t n, const std::vector<PlanIn>& in, double offset)
{
  if (ensure_slots(ctx, n, false, false, ctx->want_vec)) return -1;
  if (ensure_paths(ctx, n)) return -1;
  if (ensure_tile_state(ctx, n)) return -1;
  if (tile_weights(ctx)) return -1;
  const HostTiles& M = ctx->tiles_meta;
  std::vector<Plan> hp(n);
  std::vector<TilePlan> tp(n);
  std::vector<float*> vecs(n);
  for (uint32_t i = 0; i < n; ++i) {
    Slot& s = ctx->slots[i];
    Plan& P = hp[i];
    memset(&P, 0, sizeof(P));
    P.planner = kPlannerDijkstra; P.V = ctx->V;
    P.row_ptr = ctx->d_row_ptr; P.nbr = ctx->d_nbr; P.crn_ptr = ctx->d_crn_ptr; P.crn = ctx->d_crn; P.blocked = ctx->d_blocked;
    P.dist = s.dist; P.tkey = nullptr; P.pred = s.pred; P.dirn = s.dirn; P.cutf = s.cutf; P.stamp = s.stamp; P.dirty = s.dirty;
    P.list[0] = s.list0; P.list[1] = s.list1; P.wlist[0] = s.wlist0; P.wlist[1] = s.wlist1; P.wstamp = s.wstamp; P.cap = ctx->V; P.ctl = s.ctl; P.cnt = s.cnt;
    P.delta = 0.f; P.offset = offset; P.max_steps = 0x7FFFFFF0u; P.walk_max = kKeyWalkMax; P.descend_max = kDescendWalkMax;
    for (int k = 0; k < 3; ++k) {
      P.seed[k] = in[i].seed[k]; P.target[k] = in[i].target[k]; P.seed_d[k] = 0.f; P.seed_expands[k] = 1; P.target_expands[k] = 1;
    }
    P.seed_face = kNone;
    vecs[i] = s.vecmap;
    TilePlan& T = tp[i];
    memset(&T, 0, sizeof(T));
    T.V = ctx->V; T.ntiles = M.ntiles;
    T.vptr = ctx->d_t_vptr; T.verts = ctx->d_t_verts; T.hptr = ctx->d_t_hptr; T.halo_verts = ctx->d_t_halo_verts;
    T.halo_tile = ctx->d_t_halo_tile; T.eptr = ctx->d_t_eptr; T.rptr = ctx->d_t_rptr; T.rowptr = ctx->d_t_rowptr; T.col = ctx->d_t_col; T.tw = ctx->d_t_tw;
    T.dist = s.dist; T.pend[0] = s.tpend0; T.pend[1] = s.tpend1; T.tlast = s.tlast; T.ctl = s.tctl; T.cnt = s.tcnt;
    T.seed = in[i].seed[0]; T.target = in[i].target[0]; T.offset = offset; T.max_rounds = ctx->max_steps;
    T.band = ctx->tile_band_user > 0.f ? ctx->tile_band_user : ctx->tile_band_auto * ctx->rounds_band_mult;
    T.max_nv = M.max_nv; T.max_nh = M.max_nh; T.max_ne = M.max_ne;
  }
  HIPCHK(hipMemcpyAsync(ctx->d_plans, hp.data(), sizeof(Plan) * n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->d_tplans, tp.data(), sizeof(TilePlan) * n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->d_vecptrs, vecs.data(), sizeof(float*) * n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->d_res, 0, sizeof(PlanResult) * n, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->d_mismatch, 0, 4, ctx->stream));

  HIPCHK(hipEventRecord(ctx->ev[1], ctx->stream));
  uint32_t gi = (ctx->V + kBlock * 4 - 1) / (kBlock * 4);
  if (gi < 1) gi = 1;
  if (gi > 4096) gi = 4096;
  hipLaunchKernelGGL(k_init<kPlannerDijkstra>, dim3(gi, n), dim3(kBlock), 0, ctx->stream, ctx->d_plans);
  {
    uint32_t gt = (M.ntiles + kBlock - 1) / kBlock;
    if (gt < 1) gt = 1;
    hipLaunchKernelGGL(k_tile_init, dim3(gt, n), dim3(kBlock), 0, ctx->stream, ctx->d_tplans, ctx->d_vert_tile, -inf_f());
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(ctx->ev[2], ctx->stream));

  // active tiles form a ring along the wavefront: O(sqrt(ntiles)); every workgroup scans a
  // strided share of the tile table, so any grid size is correct
  uint32_t G = (uint32_t)std::ceil(8.0 * std::sqrt((double)M.ntiles)) + 8;
  if (const char* e = getenv("MNAV_TILE_BLOCKS")) G = (uint32_t)atoi(e);
  if (G > M.ntiles) G = M.ntiles;
  if (G < 1) G = 1;
  uint32_t launches = 0;
  int rc = 0;
  const auto t_start = std::chrono::steady_clock::now();
  ctx->ms_chunks = 0.0;
  for (;;) {
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() > ctx->max_wall_s) {
      ctx->err = "tile rounds exceeded the wall-clock guard"; return -1;
    }
    HIPCHK(hipEventRecord(ctx->evc[0], ctx->stream));
    if (run_tile_chunk(ctx, n, G)) return -1;
    HIPCHK(hipEventRecord(ctx->evc[1], ctx->stream));
    launches += kTileChunk;
    HIPCHK(hipMemcpyAsync(ctx->h_tctl, ctx->d_tctl_pool, 2 * sizeof(TCtl) * n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->ms_chunks += ev_ms(ctx->evc[0], ctx->evc[1]);
    bool all_done = true;
    for (uint32_t i = 0; i < n; ++i) {
      const TCtl& a = ctx->h_tctl[2 * i];
      const TCtl& b = ctx->h_tctl[2 * i + 1];
      const TCtl& last = a.it > b.it ? a : b;
      if (!last.done) all_done = false;
    }
    if (all_done) break;
    if (ctx->cancel.load(std::memory_order_relaxed)) { rc = 1; break; }
  }
  ctx->stats.launches = launches;
  if (rc == 0 && !ctx->lazy_paths) {
    launch_finalize(ctx, n);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipEventRecord(ctx->ev[3], ctx->stream));
  return rc;
}

// Dijkstra batches through the persistent per-plan kernel.  Returns 0, -1 (error) or 1 (cancelled).
int run_dijkstra_persistent(mnav_ctx* ctx, uint32_t n, const std::vector<PlanIn>& in, double offset)
{
  if (ensure_slots(ctx, n, false, false, ctx->want_vec)) return -1;
  if (ensure_paths(ctx, n)) return -1;
  if (ensure_tile_state(ctx, n)) return -1;
  if (tile_weights(ctx)) return -1;
  const HostTiles& M = ctx->tiles_meta;
  std::vector<Plan> hp(n);
  std::vector<TilePlan> tp(n);
  std::vector<float*> vecs(n);
  for (uint32_t i = 0; i < n; ++i) {
    Slot& s = ctx->slots[i];
    Plan& P = hp[i];
    memset(&P, 0, sizeof(P));
    P.planner = kPlannerDijkstra; P.V = ctx->V;
    P.row_ptr = ctx->d_row_ptr; P.nbr = ctx->d_nbr; P.crn_ptr = ctx->d_crn_ptr; P.crn = ctx->d_crn; P.blocked = ctx->d_blocked;
    P.dist = s.dist; P.tkey = nullptr; P.pred = s.pred; P.dirn = s.dirn; P.cutf = s.cutf; P.stamp = s.stamp; P.dirty = s.dirty;
    P.list[0] = s.list0; P.list[1] = s.list1; P.wlist[0] = s.wlist0; P.wlist[1] = s.wlist1; P.wstamp = s.wstamp; P.cap = ctx->V; P.ctl = s.ctl; P.cnt = s.cnt;
    P.delta = 0.f; P.offset = offset; P.max_steps = ctx->max_steps;
    for (int k = 0; k < 3; ++k) { P.seed[k] = in[i].seed[k]; P.target[k] = in[i].target[k]; P.seed_d[k] = 0.f; P.seed_expands[k] = 1; P.target_expands[k] = 1; }
    P.seed_face = kNone;
    vecs[i] = s.vecmap;
    TilePlan& T = tp[i];
    memset(&T, 0, sizeof(T));
    T.V = ctx->V; T.ntiles = M.ntiles;
    T.vptr = ctx->d_t_vptr; T.verts = ctx->d_t_verts; T.hptr = ctx->d_t_hptr; T.halo_verts = ctx->d_t_halo_verts;
    T.halo_tile = ctx->d_t_halo_tile; T.eptr = ctx->d_t_eptr; T.rptr = ctx->d_t_rptr; T.rowptr = ctx->d_t_rowptr; T.col = ctx->d_t_col; T.tw = ctx->d_t_tw;
    T.dist = s.dist; T.pend[0] = s.tpend0; T.pend[1] = s.tpend1; T.tlast = s.tlast; T.ctl = s.tctl; T.cnt = s.tcnt;
    T.seed = in[i].seed[0]; T.target = in[i].target[0]; T.offset = offset;
    T.max_rounds = 64u * (M.ntiles ? M.ntiles : 1u) + 1024u;          // activation cap per plan
    T.cancel = ctx->d_cancel;
    T.band = ctx->tile_band_user > 0.f ? ctx->tile_band_user : ctx->tile_band_auto;
    T.max_nv = M.max_nv; T.max_nh = M.max_nh; T.max_ne = M.max_ne;
  }
  HIPCHK(hipMemcpyAsync(ctx->d_plans, hp.data(), sizeof(Plan) * n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->d_tplans, tp.data(), sizeof(TilePlan) * n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->d_vecptrs, vecs.data(), sizeof(float*) * n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->d_res, 0, sizeof(PlanResult) * n, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->d_mismatch, 0, 4, ctx->stream));
  HIPCHK(hipEventRecord(ctx->ev[1], ctx->stream));
  uint32_t gi = (ctx->V + kBlock * 4 - 1) / (kBlock * 4);
  if (gi < 1) gi = 1;
  if (gi > 4096) gi = 4096;
  hipLaunchKernelGGL(k_init<kPlannerDijkstra>, dim3(gi, n), dim3(kBlock), 0, ctx->stream, ctx->d_plans);
  {
    uint32_t gt = (M.ntiles + kBlock - 1) / kBlock;
    if (gt < 1) gt = 1;
    hipLaunchKernelGGL(k_tile_init, dim3(gt, n), dim3(kBlock), 0, ctx->stream, ctx->d_tplans, ctx->d_vert_tile, -inf_f());
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(ctx->ev[2], ctx->stream));
  ctx->ms_chunks = 0.0;
  HIPCHK(hipEventRecord(ctx->evc[0], ctx->stream));
  if (ctx->tile_size <= 2 * kTileBlock) hipLaunchKernelGGL(k_plan_persistent<2>, dim3(n), dim3(kTileBlock), ctx->tile_lds, ctx->stream, ctx->d_tplans);
  else if (ctx->tile_size <= 4 * kTileBlock) hipLaunchKernelGGL(k_plan_persistent<4>, dim3(n), dim3(kTileBlock), ctx->tile_lds, ctx->stream, ctx->d_tplans);
  else hipLaunchKernelGGL(k_plan_persistent<8>, dim3(n), dim3(kTileBlock), ctx->tile_lds, ctx->stream, ctx->d_tplans);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(ctx->evc[1], ctx->stream));
  if (!ctx->lazy_paths) launch_finalize(ctx, n);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(ctx->ev[3], ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->ms_chunks = ev_ms(ctx->evc[0], ctx->evc[1]);
  ctx->stats.launches = 1;
  if (ctx->cancel.load(std::memory_order_relaxed)) return 1;        // the kernel left its loops early (status 3): :350-354
  return 0;
}

// Dijkstra through the asynchronous tile engine (mnav_async.h): ONE launch for the whole call.  Returns 0, -1 or 1 (cancelled).
int run_dijkstra_async(mnav_ctx* ctx, uint32_t n, const std::vector<PlanIn>& in, double offset)
{
  if (ensure_slots(ctx, n, false, false, ctx->want_vec)) return -1;
  if (ensure_paths(ctx, n)) return -1;
  if (ensure_tile_state(ctx, n)) return -1;
  if (tile_weights(ctx)) return -1;
  const HostTiles& M = ctx->tiles_meta;
  std::vector<Plan> hp(n);
  std::vector<TilePlan> tp(n);
  std::vector<float*> vecs(n);
  for (uint32_t i = 0; i < n; ++i) {
    Slot& s = ctx->slots[i];
    Plan& P = hp[i];
    memset(&P, 0, sizeof(P));
    P.planner = kPlannerDijkstra; P.V = ctx->V;
    P.row_ptr = ctx->d_row_ptr; P.nbr = ctx->d_nbr; P.crn_ptr = ctx->d_crn_ptr; P.crn = ctx->d_crn; P.blocked = ctx->d_blocked;
    P.dist = s.dist; P.tkey = nullptr; P.pred = s.pred; P.dirn = s.dirn; P.cutf = s.cutf; P.stamp = s.stamp; P.dirty = s.dirty;
    P.list[0] = s.list0; P.list[1] = s.list1; P.wlist[0] = s.wlist0; P.wlist[1] = s.wlist1; P.wstamp = s.wstamp; P.cap = ctx->V; P.ctl = s.ctl; P.cnt = s.cnt;
    P.delta = 0.f; P.offset = offset; P.max_steps = ctx->max_steps;
    for (int k = 0; k < 3; ++k) { P.seed[k] = in[i].seed[k]; P.target[k] = in[i].target[k]; P.seed_d[k] = 0.f; P.seed_expands[k] = 1; P.target_expands[k] = 1; }
    P.seed_face = kNone;
    vecs[i] = s.vecmap;
    TilePlan& T = tp[i];
    memset(&T, 0, sizeof(T));
    T.V = ctx->V; T.ntiles = M.ntiles;
    T.vptr = ctx->d_t_vptr; T.verts = ctx->d_t_verts; T.hptr = ctx->d_t_hptr; T.halo_verts = ctx->d_t_halo_verts;
    T.halo_tile = ctx->d_t_halo_tile; T.eptr = ctx->d_t_eptr; T.rptr = ctx->d_t_rptr; T.rowptr = ctx->d_t_rowptr; T.col = ctx->d_t_col; T.tw = ctx->d_t_tw;
    T.dist = s.dist; T.pend[0] = s.tpend0; T.pend[1] = s.tpend1; T.tlast = s.tlast; T.ctl = s.tctl; T.cnt = s.tcnt;
    T.seed = in[i].seed[0]; T.target = in[i].target[0]; T.offset = offset;
    T.max_rounds = 0x7FFFFFF0u;
    T.cancel = ctx->d_cancel;
    T.band = ctx->tile_band_user > 0.f ? ctx->tile_band_user : ctx->tile_band_auto;
    T.max_nv = M.max_nv; T.max_nh = M.max_nh; T.max_ne = M.max_ne;
  }
  AsyncCtl* const actl = reinterpret_cast<AsyncCtl*>(ctx->d_cancel + 4);   // words 4..7 of the 64-byte control line (word 0: mnav_cancel)
  HIPCHK(hipMemcpyAsync(ctx->d_plans, hp.data(), sizeof(Plan) * n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->d_tplans, tp.data(), sizeof(TilePlan) * n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->d_vecptrs, vecs.data(), sizeof(float*) * n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->d_res, 0, sizeof(PlanResult) * n, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->d_mismatch, 0, 4, ctx->stream));
  HIPCHK(hipMemsetAsync(actl, 0, sizeof(AsyncCtl), ctx->stream));     // every polled word is zeroed on the stream before every launch
  HIPCHK(hipEventRecord(ctx->ev[1], ctx->stream));
  uint32_t gi = (ctx->V + kBlock * 4 - 1) / (kBlock * 4);
  if (gi < 1) gi = 1;
  if (gi > 4096) gi = 4096;
  hipLaunchKernelGGL(k_init<kPlannerDijkstra>, dim3(gi, n), dim3(kBlock), 0, ctx->stream, ctx->d_plans);
  {
    uint32_t gt = (M.ntiles + kBlock - 1) / kBlock;
    if (gt < 1) gt = 1;
    hipLaunchKernelGGL(k_tile_init, dim3(gt, n), dim3(kBlock), 0, ctx->stream, ctx->d_tplans, ctx->d_vert_tile, -inf_f());
    hipLaunchKernelGGL(k_async_init, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, ctx->d_tplans, n);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(ctx->ev[2], ctx->stream));
  // Workgroups: what is resident at once, and no more per plan than its wavefront has tiles for (idle workgroups poll).
  int ncu = 256;
  (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device);
  uint32_t per_cu = 2, per_plan = 48;
  if (const char* e = getenv("MNAV_ASYNC_WG_PER_CU")) per_cu = (uint32_t)std::max(1, atoi(e));
  if (const char* e = getenv("MNAV_ASYNC_WG_PER_PLAN")) per_plan = (uint32_t)std::max(1, atoi(e));
  uint32_t G = std::min<uint64_t>((uint64_t)ncu * per_cu, (uint64_t)n * per_plan);
  if (G > M.ntiles * n) G = M.ntiles * n;
  if (G < 1) G = 1;
  double guard_s = std::min(ctx->max_wall_s, 10.0);                   // in-kernel give-up (100 MHz wall clock)
  if (const char* e = getenv("MNAV_ASYNC_MAX_S")) guard_s = atof(e);
  const unsigned long long limit_ticks = (unsigned long long)(guard_s * 1.0e8);
  ctx->ms_chunks = 0.0;
  HIPCHK(hipEventRecord(ctx->evc[0], ctx->stream));
  if (ctx->tile_size <= 2 * kTileBlock) hipLaunchKernelGGL(k_plan_async<2>, dim3(G), dim3(kTileBlock), ctx->tile_lds, ctx->stream, ctx->d_tplans, n, actl, limit_ticks);
  else if (ctx->tile_size <= 4 * kTileBlock) hipLaunchKernelGGL(k_plan_async<4>, dim3(G), dim3(kTileBlock), ctx->tile_lds, ctx->stream, ctx->d_tplans, n, actl, limit_ticks);
  else hipLaunchKernelGGL(k_plan_async<8>, dim3(G), dim3(kTileBlock), ctx->tile_lds, ctx->stream, ctx->d_tplans, n, actl, limit_ticks);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(ctx->evc[1], ctx->stream));
  AsyncCtl h{};
  HIPCHK(hipMemcpyAsync(&h, actl, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->ms_chunks = ev_ms(ctx->evc[0], ctx->evc[1]);
  ctx->stats.launches = 1;
  if (getenv("MNAV_VERBOSE"))
    fprintf(stderr, "[mnav] async: %u plans, %u workgroups, %.3f ms, abort %u, claim fails %u, idle passes %u\n", n, G, ctx->ms_chunks, h.abort, h.claim_fails, h.idle_passes);
  if (h.abort == 3u || ctx->cancel.load(std::memory_order_relaxed)) return 1;   // :350-354
  if (h.abort) { ctx->err = "asynchronous tile engine gave up (in-kernel wall-clock guard)"; return -1; }
  if (h.done_plans != n) { ctx->err = "asynchronous tile engine left plans unfinished"; return -1; }
  if (!ctx->lazy_paths) launch_finalize(ctx, n);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(ctx->ev[3], ctx->stream));
  return 0;
}

#include "mnav_tb_host.h"

float ev_ms(hipEvent_t a, hipEvent_t b)
{
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, a, b) != hipSuccess) return 0.f;
  return ms;
}

void finish_stats(mnav_ctx* ctx, uint32_t n, bool cvp)
{
  mnav_stats& st = ctx->stats;
  st.n_plans = n;
  st.steps = 0; st.bands = 0; st.armed = 0; st.goal_dist = INFINITY; st.settled = 0; st.evals = 0; st.band_shrinks = 0; st.band_cuts = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const PlanResult& r = ctx->h_res[i];
    if (r.steps > st.steps) st.steps = r.steps;
    if (r.bands > st.bands) st.bands = r.bands;
    st.band_shrinks += r.shrinks & 0xFFFFu; st.band_cuts += r.shrinks >> 16;
    st.armed += r.armed;
    if (i == 0) st.goal_dist = r.goal_dist;
    st.settled += r.settled;
    st.evals += r.evals;
  }
  st.ms_init = ev_ms(ctx->ev[1], ctx->ev[2]);
  st.ms_propagation = ev_ms(ctx->ev[2], ctx->ev[3]);
  st.ms_path = ev_ms(ctx->ev[3], ctx->ev[4]);
  st.ms_vector_map = ev_ms(ctx->ev[4], ctx->ev[5]);
  st.ms_download = ev_ms(ctx->ev[5], ctx->ev[6]);
  st.ms_total = ev_ms(ctx->ev[0], ctx->ev[6]);
  st.ms_step_kernels = (float)ctx->ms_chunks;
  // early-exit variant = settled vertices and their incident edges / faces
  const double V = ctx->V ? ctx->V : 1;
  const double frac = (double)st.settled / V;   // summed over plans
  if (!cvp) ctx->algo_bytes = (uint64_t)(24.0 * st.settled + 24.0 * frac * ctx->E);
  else ctx->algo_bytes = (uint64_t)(32.0 * st.settled + 68.0 * frac * ctx->F);
}

int check_ready(mnav_ctx* ctx)
{
  if (!ctx) return -1;
  if (!ctx->have_mesh) { ctx->err = "mnav_upload_mesh has not been called"; return -1; }
  if (!ctx->have_costs) { ctx->err = "mnav_upload_costs has not been called"; return -1; }
  return 0;
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

mnav_ctx* mnav_create(int device)
{
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return nullptr;
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  mnav_ctx* ctx = new mnav_ctx();
  ctx->device = device;
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return nullptr; }
  for (auto& e : ctx->ev)
    if (hipEventCreate(&e) != hipSuccess) { delete ctx; return nullptr; }
  for (auto& e : ctx->evc)
    if (hipEventCreate(&e) != hipSuccess) { delete ctx; return nullptr; }
  if (hipMalloc((void**)&ctx->d_seed_pos, 3 * sizeof(float)) != hipSuccess) { delete ctx; return nullptr; }
  if (hipMalloc((void**)&ctx->d_cancel, 64) != hipSuccess || hipMemset(ctx->d_cancel, 0, 64) != hipSuccess ||
      hipHostMalloc((void**)&ctx->h_one, 64, hipHostMallocDefault) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->cancel_stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return nullptr; }
  *ctx->h_one = 1u;
  if (const char* e = getenv("MNAV_NO_GRAPH")) ctx->use_graph = !(atoi(e) != 0);
  if (const char* e = getenv("MNAV_DIJKSTRA_ENGINE")) {
    if (!strcmp(e, "band") || !strcmp(e, "1")) ctx->dij_engine = 1;
    else if (!strcmp(e, "tiled") || !strcmp(e, "0")) ctx->dij_engine = 0;
    else if (!strcmp(e, "persistent") || !strcmp(e, "2")) ctx->dij_engine = 2;
    else if (!strcmp(e, "async") || !strcmp(e, "6")) ctx->dij_engine = 6;
    else ctx->dij_engine = 3;
  }
  if (const char* e = getenv("MNAV_PERSISTENT_MIN_BATCH")) ctx->persistent_min_batch = (uint32_t)atoi(e);
  return ctx;
}

void mnav_destroy(mnav_ctx* ctx)
{
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  drop_graphs(ctx);
  tb_free(ctx);
  for (auto& s : ctx->slots) free_slot(s);
  drop_layers(ctx);
  (void)hipFree(ctx->d_row_ptr); (void)hipFree(ctx->d_nbr_u); (void)hipFree(ctx->d_nbr_e); (void)hipFree(ctx->d_crn_ptr);
  (void)hipFree(ctx->d_edge_vtx); (void)hipFree(ctx->d_crn_idx); (void)hipFree(ctx->d_crn_walk); (void)hipFree(ctx->d_xyz); (void)hipFree(ctx->d_nrm);
  (void)hipFree(ctx->d_cost); (void)hipFree(ctx->d_w); (void)hipFree(ctx->d_edge_dist); (void)hipFree(ctx->d_invalid);
  (void)hipFree(ctx->d_nbr); (void)hipFree(ctx->d_crn); (void)hipFree(ctx->d_blocked); (void)hipFree(ctx->d_plans);
  (void)hipFree(ctx->d_res); (void)hipFree(ctx->d_vecptrs); (void)hipFree(ctx->d_paths); (void)hipFree(ctx->d_seed_pos);
  (void)hipFree(ctx->d_t_vptr); (void)hipFree(ctx->d_t_verts); (void)hipFree(ctx->d_t_hptr); (void)hipFree(ctx->d_t_halo_verts);
  (void)hipFree(ctx->d_t_halo_tile); (void)hipFree(ctx->d_t_eptr); (void)hipFree(ctx->d_t_src); (void)hipFree(ctx->d_vert_tile);
  (void)hipFree(ctx->d_t_rptr); (void)hipFree(ctx->d_mismatch); (void)hipFree(ctx->d_t_rowptr); (void)hipFree(ctx->d_t_col); (void)hipFree(ctx->d_t_tw);
  (void)hipFree(ctx->d_tplans);
  (void)hipFree(ctx->shard.d_iface_vert); (void)hipFree(ctx->shard.d_iface_owner); (void)hipFree(ctx->shard.d_wake_ptr); (void)hipFree(ctx->shard.d_wake_tile);
  (void)hipFree(ctx->d_wide_prefix); (void)hipFree(ctx->d_wide_sched); (void)hipFree(ctx->d_vec3);
  for (auto& st : ctx->stream_g) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
  for (auto& e : ctx->ev_fork) if (e) (void)hipEventDestroy(e);
  (void)hipFree(ctx->shard.d_owned); (void)hipFree(ctx->shard.d_changed); (void)hipFree(ctx->shard.d_minpend); (void)hipFree(ctx->shard.d_walk);
  if (ctx->cancel_stream) { (void)hipStreamSynchronize(ctx->cancel_stream); (void)hipStreamDestroy(ctx->cancel_stream); }
  if (ctx->h_one) (void)hipHostFree(ctx->h_one);
  (void)hipFree(ctx->d_cancel); (void)hipFree(ctx->d_verify_any);
  (void)hipFree(ctx->d_over); (void)hipFree(ctx->d_over_off); (void)hipFree(ctx->d_over_cap);
  (void)hipFree(ctx->d_pack); (void)hipFree(ctx->d_pack_meta); if (ctx->h_pack) (void)hipHostFree(ctx->h_pack);
  if (ctx->h_tctl) (void)hipHostFree(ctx->h_tctl);
  if (ctx->h_res) (void)hipHostFree(ctx->h_res);
  if (ctx->h_ctl) (void)hipHostFree(ctx->h_ctl);
  for (auto& e : ctx->ev) if (e) (void)hipEventDestroy(e);
  for (auto& e : ctx->evc) if (e) (void)hipEventDestroy(e);
  for (auto& e : ctx->ev_link) if (e) (void)hipEventDestroy(e);
  (void)hipFree(ctx->d_ctl_pool); (void)hipFree(ctx->d_tctl_pool);
  (void)hipFree(ctx->d_faces); (void)hipFree(ctx->d_vf_ptr); (void)hipFree(ctx->d_vf); (void)hipFree(ctx->d_walk_pos); (void)hipFree(ctx->d_walk_face);
  (void)hipFree(ctx->d_walk_jobs); (void)hipFree(ctx->d_walk_ctl);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char* mnav_last_error(const mnav_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int mnav_set_face_circulation(mnav_ctx* ctx, uint32_t V, uint32_t F, const uint32_t* vf_ptr, const uint32_t* vf)
{
  if (!ctx) return -1;
  ctx->err.clear();
  ctx->circ = FaceCirculation();
  if (!vf_ptr || !vf) return 0;                                  // back to the built-in half-edge replay
  ctx->circ.ptr.assign(vf_ptr, vf_ptr + (size_t)V + 1);
  ctx->circ.faces.assign(vf, vf + 3 * (size_t)F);
  ctx->circ.ok = true;
  return 0;
}

int mnav_upload_mesh(mnav_ctx* ctx, uint32_t V, uint32_t F, uint32_t E, const float* xyz, const uint32_t* face_vtx,
                     const uint32_t* edge_vtx, const float* vertex_normals)
{
  if (!ctx) return -1;
  ctx->err.clear();
  if ((V && !xyz) || (F && !face_vtx) || (E && !edge_vtx)) { ctx->err = "null mesh array"; return -1; }
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return -1; }
  HostTopology t;
  try {
    if (!ctx->circ.ptr.empty()) {
      if (ctx->circ.ptr.size() != (size_t)V + 1 || ctx->circ.faces.size() != 3 * (size_t)F || ctx->circ.ptr[V] != 3 * (size_t)F)
        throw std::invalid_argument("face circulation does not fit this mesh");
      t = build_topology(V, F, E, face_vtx, edge_vtx, &ctx->circ);
    } else {
      t = build_topology(V, F, E, face_vtx, edge_vtx);
    }
  }
  catch (const std::exception& ex) { ctx->err = ex.what(); return -2; }
  (void)hipStreamSynchronize(ctx->stream);
  for (auto& s : ctx->slots) free_slot(s);
  ctx->slots.clear();
  drop_layers(ctx);
  if (ctx->d_edge_dist) { ctx->alloc_bytes.erase((void*)ctx->d_edge_dist); (void)hipFree(ctx->d_edge_dist); ctx->d_edge_dist = nullptr; }   // belongs to the old mesh
  drop_graphs(ctx);
  (void)hipFree(ctx->d_paths); ctx->d_paths = nullptr; ctx->paths_words = 0;
  (void)hipFree(ctx->d_nbr); ctx->d_nbr = nullptr; (void)hipFree(ctx->d_crn); ctx->d_crn = nullptr;
  (void)hipFree(ctx->d_blocked); ctx->d_blocked = nullptr;
  (void)hipFree(ctx->d_cost); ctx->d_cost = nullptr; (void)hipFree(ctx->d_w); ctx->d_w = nullptr;
  (void)hipFree(ctx->d_invalid); ctx->d_invalid = nullptr; (void)hipFree(ctx->d_edge_dist); ctx->d_edge_dist = nullptr;
  ctx->nbr_valid = ctx->crn_valid = false; ctx->have_costs = false;
  ctx->V = V; ctx->F = F; ctx->E = E;
  {
    // a planar wavefront needs a few steps per hop of the mesh diameter ~ sqrt(V); generous cap
    const double cap = 400.0 * std::sqrt((double)V) + 20000.0;
    ctx->max_steps = cap > 2.0e9 ? 2000000000u : (uint32_t)cap;
    if (const char* e = getenv("MNAV_MAX_STEPS")) ctx->max_steps = (uint32_t)atoll(e);
    if (const char* e = getenv("MNAV_MAX_WALL_S")) ctx->max_wall_s = atof(e);
    if (const char* e = getenv("MNAV_CVP_VERIFY")) ctx->cvp_verify = atoi(e) != 0;
    if (const char* e = getenv("MNAV_KEY_WALK_MAX")) ctx->walk_max = atoi(e);
    if (const char* e = getenv("MNAV_DESCEND_WALK_MAX")) ctx->descend_max = atoi(e);
  }
  ctx->h_xyz.assign(xyz, xyz + 3 * (size_t)V);
  ctx->h_faces.assign(face_vtx, face_vtx + 3 * (size_t)F);
  ctx->h_row_ptr = t.row_ptr; ctx->h_nbr_u = t.nbr_u;
  ctx->h_vf_ptr = std::move(t.vf_ptr); ctx->h_vf = std::move(t.vf); ctx->walk_mesh_valid = false;
  tb_free(ctx);
  if (dev_upload(ctx, &ctx->d_row_ptr, t.row_ptr.data(), t.row_ptr.size())) return -1;
  if (dev_upload(ctx, &ctx->d_nbr_u, t.nbr_u.data(), t.nbr_u.size())) return -1;
  if (dev_upload(ctx, &ctx->d_nbr_e, t.nbr_e.data(), t.nbr_e.size())) return -1;
  if (dev_upload(ctx, &ctx->d_crn_ptr, t.crn_ptr.data(), t.crn_ptr.size())) return -1;
  if (dev_upload(ctx, &ctx->d_edge_vtx, edge_vtx, 2 * (size_t)E)) return -1;
  std::vector<CornerIdx> ci(t.crn_v1.size());
  for (size_t i = 0; i < ci.size(); ++i) ci[i] = CornerIdx{ t.crn_v1[i], t.crn_v2[i], t.crn_ea[i], t.crn_eb[i], t.crn_ec[i], t.crn_face[i] };
  if (dev_upload(ctx, &ctx->d_crn_walk, t.crn_walk.data(), t.crn_walk.size())) return -1;
  if (dev_upload(ctx, &ctx->d_crn_idx, ci.data(), ci.size())) return -1;
  if (dev_upload(ctx, &ctx->d_xyz, xyz, 3 * (size_t)V)) return -1;
  ctx->have_normals = vertex_normals != nullptr;
  if (dev_upload(ctx, &ctx->d_nrm, vertex_normals, vertex_normals ? 3 * (size_t)V : 0)) return -1;
  // LDS tiles of the SSSP engine
  {
    if (const char* e = getenv("MNAV_TILE_SIZE")) ctx->tile_size = (uint32_t)atoi(e);
    if (const char* e = getenv("MNAV_LAZY_PATHS")) ctx->allow_lazy_paths = atoi(e) != 0;
    if (ctx->tile_size < 64) ctx->tile_size = 64;
    if (ctx->tile_size > (uint32_t)(kTileBlock * kTileVpt)) ctx->tile_size = kTileBlock * kTileVpt;
    HostTiles T;
    for (;;) {
      try { T = build_tiles(t, xyz, ctx->tile_size); }
      catch (const std::exception& ex) { ctx->err = ex.what(); return -2; }
      ctx->tile_lds = tile_lds_bytes(T.max_nv, T.max_nh, T.max_ne);
      ctx->fin_lds = finalize_lds_bytes(T.max_nv, T.max_nh, T.max_ne);
      if (ctx->fin_lds <= 150 * 1024 || ctx->tile_size <= 64) break;
      ctx->tile_size /= 2;                       // irregular mesh: shrink until a tile fits the 160 KiB LDS
    }
    if (ctx->fin_lds > 160 * 1024) { ctx->err = "mesh valence too high for the LDS tile engine"; return -2; }
    if (ctx->fin_lds > 64 * 1024)
      HIPCHK(hipFuncSetAttribute((const void*)k_dij_finalize<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->fin_lds));
      HIPCHK(hipFuncSetAttribute((const void*)k_dij_finalize<kFinGroup, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->fin_lds));
    if (ctx->tile_lds > 64 * 1024)
      HIPCHK(hipFuncSetAttribute((const void*)k_tile_round, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->tile_lds));
    if (ctx->tile_lds > 64 * 1024) {
      HIPCHK(hipFuncSetAttribute((const void*)k_plan_persistent<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->tile_lds));
      HIPCHK(hipFuncSetAttribute((const void*)k_plan_persistent<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->tile_lds));
      HIPCHK(hipFuncSetAttribute((const void*)k_plan_persistent<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->tile_lds));
      HIPCHK(hipFuncSetAttribute((const void*)k_plan_async<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->tile_lds));
      HIPCHK(hipFuncSetAttribute((const void*)k_plan_async<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->tile_lds));
      HIPCHK(hipFuncSetAttribute((const void*)k_plan_async<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->tile_lds));
    }
    if (getenv("MNAV_VERBOSE"))
      fprintf(stderr, "[mnav] tiles: size %u, %u tiles, max owned %u, halo %u, edges %u -> LDS %zu B (solve), %zu B (finalize)\n",
              ctx->tile_size, T.ntiles, T.max_nv, T.max_nh, T.max_ne, ctx->tile_lds, ctx->fin_lds);
    (void)hipFree(ctx->d_t_tw); ctx->d_t_tw = nullptr; ctx->tw_valid = false;
    ctx->t_nnz = (uint32_t)T.col.size();
    if (dev_upload(ctx, &ctx->d_t_vptr, T.vptr.data(), T.vptr.size())) return -1;
    if (dev_upload(ctx, &ctx->d_t_verts, T.verts.data(), T.verts.size())) return -1;
    if (dev_upload(ctx, &ctx->d_t_hptr, T.hptr.data(), T.hptr.size())) return -1;
    if (dev_upload(ctx, &ctx->d_t_halo_verts, T.halo_verts.data(), T.halo_verts.size())) return -1;
    if (dev_upload(ctx, &ctx->d_t_halo_tile, T.halo_tile.data(), T.halo_tile.size())) return -1;
    if (dev_upload(ctx, &ctx->d_t_eptr, T.eptr.data(), T.eptr.size())) return -1;
    if (dev_upload(ctx, &ctx->d_t_rptr, T.rptr.data(), T.rptr.size())) return -1;
    if (dev_upload(ctx, &ctx->d_t_rowptr, T.rowptr.data(), T.rowptr.size())) return -1;
    if (dev_upload(ctx, &ctx->d_t_col, T.col.data(), T.col.size())) return -1;
    if (dev_upload(ctx, &ctx->d_t_src, T.src.data(), T.src.size())) return -1;
    if (dev_upload(ctx, &ctx->d_vert_tile, T.vert_tile.data(), T.vert_tile.size())) return -1;
    if (dev_upload(ctx, &ctx->d_t_tw, (const float*)nullptr, T.col.size())) return -1;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    // keep the sizes, and the small per-vertex maps the partitioner of mnav_shard_setup needs, on the host
    T.halo_tile.clear(); T.halo_tile.shrink_to_fit(); T.rowptr.clear(); T.rowptr.shrink_to_fit();
    T.col.clear(); T.col.shrink_to_fit(); T.src.clear(); T.src.shrink_to_fit();
    ctx->tiles_meta = std::move(T);
    ctx->shard.ready = false;
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));   // host vectors go out of scope
  ctx->have_mesh = true;
  return 0;
}

static void auto_delta(mnav_ctx* ctx, const float* w, uint32_t E)
{
  double s = 0; uint64_t c = 0;
  for (uint32_t e = 0; e < E; ++e) if (std::isfinite(w[e])) { s += w[e]; ++c; }
  ctx->delta_auto = c ? (float)(3.0 * s / (double)c) : 1.0f;
  if (!(ctx->delta_auto > 0.f)) ctx->delta_auto = 1.0f;
  // tile band ~ the potential difference across one tile (sqrt(tile_size) mean edges)
  ctx->tile_band_auto = (ctx->delta_auto / 3.0f) * std::sqrt((float)ctx->tile_size);
  if (const char* e = getenv("MNAV_TILE_BAND")) ctx->tile_band_user = (float)atof(e);
  if (const char* e = getenv("MNAV_ROUNDS_BAND_MULT")) ctx->rounds_band_mult = (float)atof(e);
}

int mnav_upload_costs(mnav_ctx* ctx, const float* vertex_costs, const float* edge_weights, const uint8_t* invalid)
{
  if (!ctx) return -1;
  ctx->err.clear();
  if (!ctx->have_mesh) { ctx->err = "mnav_upload_mesh has not been called"; return -1; }
  if ((ctx->V && !vertex_costs) || (ctx->E && !edge_weights)) { ctx->err = "null cost array"; return -1; }
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return -1; }
  if (dev_upload(ctx, &ctx->d_cost, vertex_costs, ctx->V)) return -1;
  if (dev_upload(ctx, &ctx->d_w, edge_weights, ctx->E)) return -1;
  std::vector<uint8_t> zero;
  if (!invalid) { zero.assign(ctx->V ? ctx->V : 1, 0); invalid = zero.data(); }
  if (dev_upload(ctx, &ctx->d_invalid, invalid, ctx->V)) return -1;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->h_cost.assign(vertex_costs, vertex_costs + ctx->V);
  ctx->h_invalid.assign(invalid, invalid + ctx->V);
  auto_delta(ctx, edge_weights, ctx->E);
  ctx->nbr_valid = ctx->crn_valid = false;
  ctx->have_costs = true;
  ctx->edge_cost_factor = 0.0;                                       // weights came from the caller: mnav_update_costs only touches vertex costs
  return 0;
}

// device pass shared by mnav_compute_edge_weights / mnav_combine_costs: d_cost and d_edge_dist are resident, d_w is
// (re)computed; host mirrors (cost for the seed cut-offs, mean weight for the band widths) are refreshed
static int edge_weight_pass(mnav_ctx* ctx, double edge_cost_factor, const uint8_t* invalid, float* vertex_costs_out, float* edge_weights_out)
{
  if (dev_upload(ctx, &ctx->d_w, (const float*)nullptr, ctx->E)) return -1;
  std::vector<uint8_t> zero;
  if (!invalid) { zero.assign(ctx->V ? ctx->V : 1, 0); invalid = zero.data(); }
  if (dev_upload(ctx, &ctx->d_invalid, invalid, ctx->V)) return -1;
  const uint32_t gb = (ctx->E + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(k_edge_weights, dim3(gb ? gb : 1), dim3(kBlock), 0, ctx->stream, ctx->E, ctx->d_edge_vtx, ctx->d_edge_dist,
                     ctx->d_cost, edge_cost_factor, ctx->d_w);
  HIPCHK(hipGetLastError());
  std::vector<float> w(ctx->E ? ctx->E : 1);
  ctx->h_cost.resize(ctx->V);
  HIPCHK(hipMemcpyAsync(w.data(), ctx->d_w, sizeof(float) * ctx->E, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->h_cost.data(), ctx->d_cost, sizeof(float) * ctx->V, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (edge_weights_out) memcpy(edge_weights_out, w.data(), sizeof(float) * ctx->E);
  if (vertex_costs_out) memcpy(vertex_costs_out, ctx->h_cost.data(), sizeof(float) * ctx->V);
  ctx->h_invalid.assign(invalid, invalid + ctx->V);
  auto_delta(ctx, w.data(), ctx->E);
  ctx->nbr_valid = ctx->crn_valid = false;
  ctx->edge_cost_factor = edge_cost_factor;
  ctx->have_costs = true;
  return 0;
}

int mnav_compute_edge_weights(mnav_ctx* ctx, const float* vertex_costs, const float* edge_distances, double edge_cost_factor,
                              const uint8_t* invalid, float* edge_weights_out)
{
  if (!ctx) return -1;
  ctx->err.clear();
  if (!ctx->have_mesh) { ctx->err = "mnav_upload_mesh has not been called"; return -1; }
  if ((ctx->V && !vertex_costs) || (ctx->E && !edge_distances)) { ctx->err = "null cost array"; return -1; }
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return -1; }
  if (dev_upload(ctx, &ctx->d_cost, vertex_costs, ctx->V)) return -1;
  if (dev_upload(ctx, &ctx->d_edge_dist, edge_distances, ctx->E)) return -1;
  ctx->crn_infl_valid = false;
  return edge_weight_pass(ctx, edge_cost_factor, invalid, nullptr, edge_weights_out);
}

int mnav_combine_costs(mnav_ctx* ctx, int mode, uint32_t n_layers, const float* const* layer_costs, const float* weights,
                       const float* edge_distances, double edge_cost_factor, const uint8_t* invalid, float* vertex_costs_out,
                       float* edge_weights_out)
{
  if (!ctx) return -1;
  ctx->err.clear();
  if (!ctx->have_mesh) { ctx->err = "mnav_upload_mesh has not been called"; return -1; }
  if (mode != 0 && mode != 1) { ctx->err = "combination mode must be 0 (max) or 1 (weighted sum)"; return -1; }
  if ((n_layers && !layer_costs) || (mode == 1 && n_layers && !weights) || (ctx->E && !edge_distances && !ctx->d_edge_dist)) { ctx->err = "null input array"; return -1; }
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return -1; }
  const uint32_t V = ctx->V;
  DevTmp<float> d_layers, d_wts;
  HIPCHK(hipMalloc(d_layers.out(), sizeof(float) * ((size_t)n_layers * V + 1)));
  HIPCHK(hipMalloc(d_wts.out(), sizeof(float) * (n_layers + 1)));
  int rc = 0;
  for (uint32_t l = 0; l < n_layers && rc == 0; ++l) {
    if (!layer_costs[l]) { ctx->err = "null layer"; rc = -1; break; }
    if (hipMemcpyAsync(d_layers + (size_t)l * V, layer_costs[l], sizeof(float) * V, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { ctx->err = "layer upload failed"; rc = -1; }
  }
  if (rc == 0 && mode == 1 && n_layers &&
      hipMemcpyAsync(d_wts, weights, sizeof(float) * n_layers, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { ctx->err = "weight upload failed"; rc = -1; }
  if (rc == 0 && dev_upload(ctx, &ctx->d_cost, (const float*)nullptr, V)) rc = -1;
  if (rc == 0 && edge_distances) { ctx->crn_infl_valid = false; if (dev_upload(ctx, &ctx->d_edge_dist, edge_distances, ctx->E)) rc = -1; }   // NULL: keep the resident ones
  if (rc == 0) {
    const uint32_t gb = (V + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(k_combine, dim3(gb ? gb : 1), dim3(kBlock), 0, ctx->stream, V, mode, n_layers, d_layers, d_wts, ctx->d_cost);
    if (hipGetLastError() != hipSuccess) { ctx->err = "cost combination failed"; rc = -1; }
  }
  // the combined costs never leave the device: the edge-weight pass reads them where they are
  if (rc == 0) rc = edge_weight_pass(ctx, edge_cost_factor, invalid, vertex_costs_out, edge_weights_out);
  (void)hipStreamSynchronize(ctx->stream);
  return rc;
}

int mnav_update_costs(mnav_ctx* ctx, uint32_t n, const uint32_t* vertex_ids, const float* values)
{
  if (!ctx) return -1;
  ctx->err.clear();
  if (!ctx->have_costs) { ctx->err = "no costs resident yet (mnav_upload_costs / mnav_compute_edge_weights / mnav_combine_costs first)"; return -1; }
  if (n == 0) return 0;
  if (!vertex_ids || !values) { ctx->err = "null input array"; return -1; }
  for (uint32_t i = 0; i < n; ++i) if (vertex_ids[i] >= ctx->V) { ctx->err = "vertex id out of range"; return -1; }
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return -1; }
  DevTmp<uint32_t> d_ids; DevTmp<float> d_vals;
  HIPCHK(hipMalloc(d_ids.out(), sizeof(uint32_t) * n));
  HIPCHK(hipMalloc(d_vals.out(), sizeof(float) * n));
  int rc = 0;
  if (hipMemcpyAsync(d_ids, vertex_ids, sizeof(uint32_t) * n, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
      hipMemcpyAsync(d_vals, values, sizeof(float) * n, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { ctx->err = "upload failed"; rc = -1; }
  if (rc == 0) {
    hipLaunchKernelGGL(k_scatter_costs, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, n, d_ids, d_vals, ctx->d_cost);
    // "Edge costs are only affected by vertex costs if layer_factor is not 0" (:568-572)
    if (ctx->edge_cost_factor != 0.0) {
      if (!ctx->d_edge_dist) { ctx->err = "edge distances are not resident (weights were uploaded, not computed here)"; rc = -1; }
      else hipLaunchKernelGGL(k_update_edge_weights, dim3((8 * n + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, n, d_ids, ctx->d_row_ptr,
                              ctx->d_nbr_u, ctx->d_nbr_e, ctx->d_edge_dist, ctx->d_cost, ctx->edge_cost_factor, ctx->d_w);
    }
    if (rc == 0 && hipGetLastError() != hipSuccess) { ctx->err = "cost update failed"; rc = -1; }
  }
  (void)hipStreamSynchronize(ctx->stream);
  if (rc) return rc;
  for (uint32_t i = 0; i < n; ++i) ctx->h_cost[vertex_ids[i]] = values[i];
  ctx->nbr_valid = ctx->crn_valid = false;                          // the cost-limit folded copies are rebuilt on the next plan
  return 0;
}

int mnav_update_edge_weights(mnav_ctx* ctx, uint32_t n, const uint32_t* edge_ids, const float* values)
{
  if (!ctx) return -1;
  ctx->err.clear();
  if (!ctx->have_costs) { ctx->err = "no costs resident yet (mnav_upload_costs / mnav_compute_edge_weights / mnav_combine_costs first)"; return -1; }
  if (n == 0) return 0;
  if (!edge_ids || !values) { ctx->err = "null input array"; return -1; }
  for (uint32_t i = 0; i < n; ++i) if (edge_ids[i] >= ctx->E) { ctx->err = "edge id out of range"; return -1; }
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return -1; }
  DevTmp<uint32_t> d_ids; DevTmp<float> d_vals;
  HIPCHK(hipMalloc(d_ids.out(), sizeof(uint32_t) * n));
  HIPCHK(hipMalloc(d_vals.out(), sizeof(float) * n));
  int rc = 0;
  if (hipMemcpyAsync(d_ids, edge_ids, sizeof(uint32_t) * n, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
      hipMemcpyAsync(d_vals, values, sizeof(float) * n, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { ctx->err = "upload failed"; rc = -1; }
  if (rc == 0) {
    hipLaunchKernelGGL(k_scatter_costs, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, n, d_ids, d_vals, ctx->d_w);   // (a scatter of floats by index)
    if (hipGetLastError() != hipSuccess) { ctx->err = "edge weight update failed"; rc = -1; }
  }
  (void)hipStreamSynchronize(ctx->stream);
  if (rc) return rc;
  ctx->nbr_valid = ctx->crn_valid = false;                          // the cost-limit folded copies (and the tiles' weights behind them) are rebuilt on the next plan
  return 0;
}

int mnav_download_costs(mnav_ctx* ctx, float* vertex_costs_out, float* edge_weights_out)
{
  if (!ctx || !ctx->have_costs) { if (ctx) ctx->err = "no costs resident"; return -1; }
  if (hipSetDevice(ctx->device) != hipSuccess) return -1;
  if (vertex_costs_out) HIPCHK(hipMemcpyAsync(vertex_costs_out, ctx->d_cost, sizeof(float) * ctx->V, hipMemcpyDeviceToHost, ctx->stream));
  if (edge_weights_out) HIPCHK(hipMemcpyAsync(edge_weights_out, ctx->d_w, sizeof(float) * ctx->E, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

// -- layers on the device ------------------------------------------------------------------------
static int layer_slot(mnav_ctx* ctx, uint32_t layer, bool want_dist)
{
  if (!ctx->have_mesh) { ctx->err = "mnav_upload_mesh has not been called"; return -1; }
  if (layer >= 64) { ctx->err = "layer index out of range (64 layers)"; return -1; }
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return -1; }
  if (ctx->layers.size() <= layer) ctx->layers.resize(layer + 1);
  mnav_ctx::Layer& L = ctx->layers[layer];
  const size_t V = ctx->V ? ctx->V : 1;
  if (!L.cost) HIPCHK(hipMalloc((void**)&L.cost, 4 * V));
  if (!L.lethal) HIPCHK(hipMalloc((void**)&L.lethal, V));
  if (want_dist && !L.dist) HIPCHK(hipMalloc((void**)&L.dist, 4 * V));
  return 0;
}

static int ensure_edge_distances(mnav_ctx* ctx)
{
  if (ctx->d_edge_dist) return 0;
  if (dev_upload(ctx, &ctx->d_edge_dist, (const float*)nullptr, ctx->E)) return -1;
  const uint32_t gb = (ctx->E + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(k_edge_dist, dim3(gb ? gb : 1), dim3(kBlock), 0, ctx->stream, ctx->E, ctx->d_edge_vtx, ctx->d_xyz, ctx->d_edge_dist);
  HIPCHK(hipGetLastError());
  ctx->crn_infl_valid = false;
  return 0;
}

int mnav_layer_upload(mnav_ctx* ctx, uint32_t layer, const float* costs, const uint8_t* lethal)
{
  if (!ctx) return -1;
  ctx->err.clear();
  if (layer_slot(ctx, layer, false)) return -1;
  mnav_ctx::Layer& L = ctx->layers[layer];
  if (!costs) { ctx->err = "null cost array"; return -1; }
  HIPCHK(hipMemcpyAsync(L.cost, costs, sizeof(float) * ctx->V, hipMemcpyHostToDevice, ctx->stream));
  if (lethal) HIPCHK(hipMemcpyAsync(L.lethal, lethal, ctx->V, hipMemcpyHostToDevice, ctx->stream));
  else HIPCHK(hipMemsetAsync(L.lethal, 0, ctx->V ? ctx->V : 1, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  L.ready = true;
  return 0;
}

int mnav_layer_steepness(mnav_ctx* ctx, uint32_t layer, double threshold)
{
  if (!ctx) return -1;
  ctx->err.clear();
  if (layer_slot(ctx, layer, false)) return -1;
  if (!ctx->have_normals) { ctx->err = "vertex normals are not resident (mnav_upload_mesh with vertex_normals)"; return -1; }
  mnav_ctx::Layer& L = ctx->layers[layer];
  const uint32_t gb = (ctx->V + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(k_steepness, dim3(gb ? gb : 1), dim3(kBlock), 0, ctx->stream, ctx->V, ctx->d_nrm, threshold, L.cost, L.lethal);
  HIPCHK(hipGetLastError());
  L.ready = true;
  return 0;
}

// InflationLayer::computeLayer (inflation_layer.cpp:96-178 / :577-600): lethals of the input layer -> distances_
// (multi-source wave) -> riskiness.  Runs the wave on the band engine with plan slot 0's work arrays.
int mnav_layer_inflation(mnav_ctx* ctx, uint32_t layer, uint32_t input_layer, double inflation_radius, double inscribed_radius,
                         double inscribed_value, double lethal_value, double cost_scaling_factor, const uint8_t* invalid)
{
  if (!ctx) return -1;
  ctx->err.clear();
  if (input_layer >= ctx->layers.size() || !ctx->layers[input_layer].ready) { ctx->err = "input layer is not resident"; return -1; }
  if (layer == input_layer) { ctx->err = "a layer cannot inflate itself"; return -1; }
  if (layer_slot(ctx, layer, true)) return -1;
  if (ctx->V > (1u << kKeyIdBits)) { ctx->err = "the ordered wave supports meshes of up to 2^26 vertices"; return -1; }
  if (ensure_edge_distances(ctx)) return -1;
  const uint32_t V = ctx->V;
  const uint32_t gb = (V + kBlock - 1) / kBlock ? (V + kBlock - 1) / kBlock : 1;
  if (!ctx->d_crn_infl) HIPCHK(hipMalloc((void**)&ctx->d_crn_infl, sizeof(Corner) * (size_t)(ctx->F ? 3 * (size_t)ctx->F : 1)));
  if (!ctx->crn_infl_valid) {
    hipLaunchKernelGGL(k_build_crn_infl, dim3(gb), dim3(kBlock), 0, ctx->stream, V, ctx->d_crn_ptr, ctx->d_crn_idx, ctx->d_edge_dist, ctx->d_crn_infl);
    HIPCHK(hipGetLastError());
    ctx->crn_infl_valid = true;
  }
  const size_t Vn = V ? V : 1;
  if (!ctx->d_infl_mask) HIPCHK(hipMalloc((void**)&ctx->d_infl_mask, Vn));
  if (!ctx->d_zero_u8) { HIPCHK(hipMalloc((void**)&ctx->d_zero_u8, Vn)); HIPCHK(hipMemsetAsync(ctx->d_zero_u8, 0, Vn, ctx->stream)); }
  if (!ctx->d_infl_keyd) HIPCHK(hipMalloc((void**)&ctx->d_infl_keyd, 4 * Vn));
  DevTmp<uint8_t> d_inv;
  if (invalid) { HIPCHK(hipMalloc(d_inv.out(), Vn)); HIPCHK(hipMemcpyAsync(d_inv, invalid, V, hipMemcpyHostToDevice, ctx->stream)); }
  mnav_ctx::Layer& L = ctx->layers[layer];
  mnav_ctx::Layer& In = ctx->layers[input_layer];
  L.inflation_radius = inflation_radius; L.inscribed_radius = inscribed_radius; L.inscribed_value = inscribed_value; L.lethal_value = lethal_value;
  hipLaunchKernelGGL(k_infl_mask, dim3(gb), dim3(kBlock), 0, ctx->stream, V, In.lethal, d_inv, ctx->d_infl_mask);
  HIPCHK(hipMemcpyAsync(L.lethal, In.lethal, V, hipMemcpyDeviceToDevice, ctx->stream));    // lethal_vertices_ = input->lethals() :170,:584
  if (ensure_slots(ctx, 1, true, true, false)) return -1;
  Slot& s = ctx->slots[0];
  ctx->caller_slot.assign(1, kNone);                                // the wave works in plan slot 0: the last plan's resident outputs are gone
  Plan P;
  memset(&P, 0, sizeof(P));
  P.planner = kPlannerCvp; P.V = V;
  P.row_ptr = ctx->d_row_ptr; P.nbr = nullptr; P.crn_ptr = ctx->d_crn_ptr; P.crn = ctx->d_crn_infl; P.blocked = ctx->d_zero_u8;
  P.dist = L.dist; P.tkey = s.tkey; P.pred = s.pred; P.dirn = s.dirn; P.cutf = s.cutf; P.stamp = s.stamp; P.dirty = s.dirty;
  P.list[0] = s.list0; P.list[1] = s.list1; P.wlist[0] = s.wlist0; P.wlist[1] = s.wlist1; P.wstamp = s.wstamp; P.cap = V; P.ctl = s.ctl; P.cnt = s.cnt;
  const float maxd = (float)inflation_radius;                                               // :438 (const float&)
  P.delta = maxd > 0.f ? maxd : 1.0f;                                                       // one band per radius: the wave dies out within ~2
  P.offset = 0.0; P.max_steps = ctx->max_steps; P.walk_max = ctx->walk_max; P.descend_max = ctx->descend_max;
  for (int k = 0; k < 3; ++k) { P.seed[k] = kNone; P.target[k] = kNone; P.seed_d[k] = 0.f; P.seed_expands[k] = 1; P.target_expands[k] = 0; }
  P.seed_face = kNone;
  P.seed_mask = ctx->d_infl_mask; P.keyd = ctx->d_infl_keyd; P.infl_max = maxd;
  HIPCHK(hipMemcpyAsync(ctx->d_plans, &P, sizeof(Plan), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipEventRecord(ctx->ev[1], ctx->stream));
  uint32_t gi = (V + kBlock * 4 - 1) / (kBlock * 4);
  if (gi < 1) gi = 1;
  if (gi > 4096) gi = 4096;
  hipLaunchKernelGGL(k_init<kPlannerCvp>, dim3(gi, 1), dim3(kBlock), 0, ctx->stream, ctx->d_plans);
  hipLaunchKernelGGL(k_infl_ctl, dim3(1), dim3(64), 0, ctx->stream, ctx->d_plans);
  hipLaunchKernelGGL(k_infl_seed, dim3(gi), dim3(kBlock), 0, ctx->stream, ctx->d_plans);
  HIPCHK(hipGetLastError());
  // the wave front of an inflation is as long as the lethal contours, not O(sqrt V): more waves than a plan gets
  uint32_t G = blocks_per_plan(ctx) * 12u;
  if (G > 8192u) G = 8192u;
  const auto t_start = std::chrono::steady_clock::now();
  Ctl last{};
  for (;;) {
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() > ctx->max_wall_s) {
      ctx->err = "inflation wave exceeded the wall-clock guard"; return -1;
    }
    if (run_chunk<kPlannerCvp>(ctx, 1, G, false)) return -1;
    HIPCHK(hipMemcpyAsync(ctx->h_ctl, ctx->d_ctl_pool, 2 * sizeof(Ctl), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    last = ctx->h_ctl[0].it > ctx->h_ctl[1].it ? ctx->h_ctl[0] : ctx->h_ctl[1];
    if (getenv("MNAV_TRACE"))
      fprintf(stderr, "[mnav] inflation it %d n %u thr %.6f fixed %.6f width %.4g bands %u band_steps %u shrinks %u cuts %u repair %u evals %u wread %u wbase %u done %u (verify sweeps of the previous wave %u)\n", last.it,
              last.n, last.thr, last.thr_fixed, last.width, last.bands, last.band_steps, last.shrinks, last.cuts, last.repair, last.evals, last.wread, last.wbase, last.done, ctx->verify_sweeps_used);
    if (last.done) break;
  }
  HIPCHK(hipEventRecord(ctx->ev[2], ctx->stream));
  // verification: every vertex must be a fixed point of the replay rule on the converged state (k_cvp_verify)
  if (verify_sweeps(ctx, 1)) return -1;
  hipLaunchKernelGGL(k_infl_cost, dim3(gb), dim3(kBlock), 0, ctx->stream, V, L.dist, inflation_radius, inscribed_radius, inscribed_value,
                     lethal_value, cost_scaling_factor, L.cost);
  HIPCHK(hipGetLastError());
  // vector_map_ (:277-309): accumulation over the lethal contours, then assignments in pop order (launches until settled)
  L.have_vec = false;
  if (!L.vec) HIPCHK(hipMalloc((void**)&L.vec, 12 * Vn));
  if (!L.vstate) HIPCHK(hipMalloc((void**)&L.vstate, 3 * Vn));
  if (!ctx->d_verify_any) HIPCHK(hipMalloc((void**)&ctx->d_verify_any, 4));
  uint32_t* d_vctl = nullptr;
  HIPCHK(hipMalloc((void**)&d_vctl, 16));
  HIPCHK(hipMemsetAsync(d_vctl, 0, 16, ctx->stream));
  uint8_t *st0 = L.vstate, *st1 = L.vstate + Vn, *acc = L.vstate + 2 * Vn;
  hipLaunchKernelGGL(k_infl_accum, dim3(gb), dim3(kBlock), 0, ctx->stream, ctx->d_plans, ctx->d_crn_walk, ctx->d_xyz, L.vec, st0, acc, d_vctl);
  uint32_t vctl[4] = { 0, 0, 0, 0 };
  bool vec_ok = true;
  for (int sweep = 0; sweep < 4096; ++sweep) {
    HIPCHK(hipMemsetAsync(d_vctl, 0, 8, ctx->stream));
    hipLaunchKernelGGL(k_infl_assign, dim3(gb), dim3(kBlock), 0, ctx->stream, ctx->d_plans, L.vec, st0, st1, acc, d_vctl);
    HIPCHK(hipMemcpyAsync(vctl, d_vctl, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    std::swap(st0, st1);
    if (vctl[2]) { vec_ok = false; break; }                          // a vertex with too many neighbours for the walk positions
    if (vctl[0] == 0) break;
    if (vctl[1] == 0) { vec_ok = false; break; }                     // nothing moved although something waits: not on a verified state
  }
  if (st0 != L.vstate) HIPCHK(hipMemcpyAsync(L.vstate, st0, Vn, hipMemcpyDeviceToDevice, ctx->stream));   // final states in the first array
  (void)hipFree(d_vctl);
  L.have_vec = vec_ok;
  HIPCHK(hipEventRecord(ctx->ev[3], ctx->stream));
  Cnt flags{};
  HIPCHK(hipMemcpyAsync(&flags, s.cnt + 3, sizeof(Cnt), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->infl_steps = (uint32_t)(last.it < 0 ? 0 : last.it); ctx->infl_bands = last.bands; ctx->infl_evals = last.evals;
  ctx->infl_ms = ev_ms(ctx->ev[1], ctx->ev[3]); ctx->infl_ms_wave = ev_ms(ctx->ev[1], ctx->ev[2]);
  if (last.overflow) { ctx->err = "inflation wave did not converge (work-list overflow or step limit)"; return -1; }
  if (flags.n_next & kFlagWalkLimit) { ctx->err = "inflation wave: cascade-tree walk bound hit; the result may not be the reference's"; return -1; }
  if (flags.changed) { ctx->err = "inflation wave: the converged state is not a fixed point of the replay rule"; return -1; }
  L.ready = true;
  return 0;
}

int mnav_layer_download(mnav_ctx* ctx, uint32_t layer, float* costs_out, uint8_t* lethal_out, float* distances_out)
{
  if (!ctx) return -1;
  ctx->err.clear();
  if (layer >= ctx->layers.size() || !ctx->layers[layer].ready) { ctx->err = "layer is not resident"; return -1; }
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return -1; }
  mnav_ctx::Layer& L = ctx->layers[layer];
  if (costs_out) HIPCHK(hipMemcpyAsync(costs_out, L.cost, sizeof(float) * ctx->V, hipMemcpyDeviceToHost, ctx->stream));
  if (lethal_out) HIPCHK(hipMemcpyAsync(lethal_out, L.lethal, ctx->V, hipMemcpyDeviceToHost, ctx->stream));
  if (distances_out) {
    if (!L.dist) { ctx->err = "this layer keeps no distances"; return -1; }
    HIPCHK(hipMemcpyAsync(distances_out, L.dist, sizeof(float) * ctx->V, hipMemcpyDeviceToHost, ctx->stream));
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

int mnav_layer_download_vectors(mnav_ctx* ctx, uint32_t layer, float* vectors_out, uint8_t* has_vector_out)
{
  if (!ctx) return -1;
  ctx->err.clear();
  if (layer >= ctx->layers.size() || !ctx->layers[layer].ready) { ctx->err = "layer is not resident"; return -1; }
  mnav_ctx::Layer& L = ctx->layers[layer];
  if (!L.have_vec) { ctx->err = "this layer has no vector field (not an inflation layer, or a vertex with more than 15 neighbours)"; return -1; }
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return -1; }
  const uint32_t V = ctx->V;
  if (vectors_out) HIPCHK(hipMemcpyAsync(vectors_out, L.vec, 12 * (size_t)V, hipMemcpyDeviceToHost, ctx->stream));
  std::vector<uint8_t> st(V ? V : 1);
  if (has_vector_out) HIPCHK(hipMemcpyAsync(st.data(), L.vstate, V, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (has_vector_out) for (uint32_t v = 0; v < V; ++v) has_vector_out[v] = st[v] == 1 ? 1 : 0;
  return 0;
}

int mnav_layer_stats(const mnav_ctx* ctx, uint32_t* steps, uint32_t* bands, uint64_t* evals, float* ms, uint32_t* verify_sweeps, float* ms_wave)
{
  if (!ctx) return -1;
  if (steps) *steps = ctx->infl_steps;
  if (bands) *bands = ctx->infl_bands;
  if (evals) *evals = ctx->infl_evals;
  if (ms) *ms = ctx->infl_ms;
  if (verify_sweeps) *verify_sweeps = ctx->verify_sweeps_used;
  if (ms_wave) *ms_wave = ctx->infl_ms_wave;
  return 0;
}

// CombinationLayer (max :44-85 / weighted sum :185-248) over resident layers, then MeshMap::computeEdgeWeights:
// the whole cost preparation of a map without a host copy of a single V-sized array.
int mnav_combine_layers(mnav_ctx* ctx, int mode, uint32_t n_layers, const uint32_t* layers, const float* weights, double edge_cost_factor,
                        const uint8_t* invalid)
{
  if (!ctx) return -1;
  ctx->err.clear();
  if (!ctx->have_mesh) { ctx->err = "mnav_upload_mesh has not been called"; return -1; }
  if (mode != 0 && mode != 1) { ctx->err = "combination mode must be 0 (max) or 1 (weighted sum)"; return -1; }
  if ((n_layers && !layers) || (mode == 1 && n_layers && !weights) || n_layers > 64) { ctx->err = "bad layer list"; return -1; }
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return -1; }
  std::vector<const float*> ptrs(n_layers ? n_layers : 1, nullptr);
  for (uint32_t l = 0; l < n_layers; ++l) {
    if (layers[l] >= ctx->layers.size() || !ctx->layers[layers[l]].ready) { ctx->err = "layer is not resident"; return -1; }
    ptrs[l] = ctx->layers[layers[l]].cost;
  }
  if (ensure_edge_distances(ctx)) return -1;
  DevTmp<const float*> d_ptrs; DevTmp<float> d_wts;
  HIPCHK(hipMalloc(d_ptrs.out(), sizeof(float*) * (n_layers + 1)));
  HIPCHK(hipMalloc(d_wts.out(), sizeof(float) * (n_layers + 1)));
  int rc = 0;
  if (n_layers && hipMemcpyAsync(d_ptrs, ptrs.data(), sizeof(float*) * n_layers, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = -1;
  if (rc == 0 && mode == 1 && n_layers && hipMemcpyAsync(d_wts, weights, sizeof(float) * n_layers, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = -1;
  if (rc == 0 && dev_upload(ctx, &ctx->d_cost, (const float*)nullptr, ctx->V)) rc = -1;
  if (rc == 0) {
    const uint32_t gb = (ctx->V + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(k_combine_resident, dim3(gb ? gb : 1), dim3(kBlock), 0, ctx->stream, ctx->V, mode, n_layers, d_ptrs, d_wts, ctx->d_cost);
    if (hipGetLastError() != hipSuccess) rc = -1;
  }
  if (rc != 0 && ctx->err.empty()) ctx->err = "layer combination failed";
  if (rc == 0) rc = edge_weight_pass(ctx, edge_cost_factor, invalid, nullptr, nullptr);
  (void)hipStreamSynchronize(ctx->stream);
  return rc;
}

// The incremental counterpart of mnav_combine_layers: CombinationLayer::onInputChanged + MeshMap::layerChanged +
// updateEdgeWeights(changed) for the n vertices a layer reported as changed (the layers themselves were updated on the
// device or re-uploaded before).  Same combination mode / layer list / weights as the full pass.
int mnav_combine_layers_update(mnav_ctx* ctx, int mode, uint32_t n_layers, const uint32_t* layers, const float* weights, uint32_t n,
                               const uint32_t* vertex_ids)
{
  if (!ctx) return -1;
  ctx->err.clear();
  if (!ctx->have_costs) { ctx->err = "no combined costs resident yet (mnav_combine_layers first)"; return -1; }
  if (mode != 0 && mode != 1) { ctx->err = "combination mode must be 0 (max) or 1 (weighted sum)"; return -1; }
  if ((n_layers && !layers) || (mode == 1 && n_layers && !weights) || n_layers > 64) { ctx->err = "bad layer list"; return -1; }
  if (n == 0) return 0;
  if (!vertex_ids) { ctx->err = "null id array"; return -1; }
  for (uint32_t i = 0; i < n; ++i) if (vertex_ids[i] >= ctx->V) { ctx->err = "vertex id out of range"; return -1; }
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return -1; }
  std::vector<const float*> ptrs(n_layers ? n_layers : 1, nullptr);
  for (uint32_t l = 0; l < n_layers; ++l) {
    if (layers[l] >= ctx->layers.size() || !ctx->layers[layers[l]].ready) { ctx->err = "layer is not resident"; return -1; }
    ptrs[l] = ctx->layers[layers[l]].cost;
  }
  DevTmp<const float*> d_ptrs; DevTmp<float> d_wts, d_vals; DevTmp<uint32_t> d_ids;
  HIPCHK(hipMalloc(d_ptrs.out(), sizeof(float*) * (n_layers + 1)));
  HIPCHK(hipMalloc(d_wts.out(), sizeof(float) * (n_layers + 1)));
  HIPCHK(hipMalloc(d_vals.out(), sizeof(float) * n));
  HIPCHK(hipMalloc(d_ids.out(), sizeof(uint32_t) * n));
  std::vector<float> vals(n);
  int rc = 0;
  if (n_layers && hipMemcpyAsync(d_ptrs, ptrs.data(), sizeof(float*) * n_layers, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = -1;
  if (rc == 0 && mode == 1 && n_layers && hipMemcpyAsync(d_wts, weights, sizeof(float) * n_layers, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = -1;
  if (rc == 0 && hipMemcpyAsync(d_ids, vertex_ids, sizeof(uint32_t) * n, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = -1;
  if (rc == 0) {
    hipLaunchKernelGGL(k_combine_resident_ids, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, n, d_ids, mode, n_layers, d_ptrs, d_wts,
                       ctx->d_cost, d_vals);
    if (ctx->edge_cost_factor != 0.0)                               // "Edge costs are only affected by vertex costs if layer_factor is not 0" (:568-572)
      hipLaunchKernelGGL(k_update_edge_weights, dim3((8 * n + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, n, d_ids, ctx->d_row_ptr,
                         ctx->d_nbr_u, ctx->d_nbr_e, ctx->d_edge_dist, ctx->d_cost, ctx->edge_cost_factor, ctx->d_w);
    if (hipGetLastError() != hipSuccess) rc = -1;
  }
  if (rc == 0 && hipMemcpyAsync(vals.data(), d_vals, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = -1;
  (void)hipStreamSynchronize(ctx->stream);
  if (rc != 0) { if (ctx->err.empty()) ctx->err = "incremental combination failed"; return rc; }
  for (uint32_t i = 0; i < n; ++i) ctx->h_cost[vertex_ids[i]] = vals[i];
  ctx->nbr_valid = ctx->crn_valid = false;                          // the cost-limit folded copies are rebuilt on the next plan
  return 0;
}

static uint32_t dijkstra_impl(mnav_ctx* ctx, uint32_t n, const uint32_t* seeds, const uint32_t* targets, double offset,
                              double cost_limit, uint32_t* codes_out, float* dist_out, uint32_t* pred_out, uint32_t* path_out,
                              uint32_t path_cap, uint32_t* path_len, float* vecmap_out, bool want_vecmap)
{
  if (check_ready(ctx)) return MNAV_INTERNAL_ERROR;
  ctx->err.clear();
  // goal_dist = dist[target] + offset cuts the wave off BEHIND the robot (dijkstra :296).  A negative offset (the reference
  // takes any double) stops the expansion AT the robot vertex: the engines run with the bound of offset 0 -- everything at or
  // below dist[target] is final then -- and the passes that derive tentative values, predecessors and paths apply the
  // reference's expanded set through goal_cut() (mnav_eval.h).
  if (offset != offset) { ctx->err = "goal_dist_offset is NaN"; return MNAV_INTERNAL_ERROR; }
  ctx->cancel.store(0);                                               // dijkstra :238
  if (ctx->d_cancel) { (void)hipStreamSynchronize(ctx->cancel_stream); (void)hipMemsetAsync(ctx->d_cancel, 0, 4, ctx->stream); }
  want_vecmap = want_vecmap || ctx->resident_vecmap;
  ctx->want_vec = want_vecmap;
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return MNAV_INTERNAL_ERROR; }
  MTRACE("start");
  const uint32_t V = ctx->V;
  uint32_t worst = MNAV_SUCCESS;
  // id checks stand in for the optional-handle tests of dijkstra :240-243
  std::vector<PlanIn> in; std::vector<uint32_t> map;   // map: device plan -> caller index
  std::vector<uint32_t> codes(n, MNAV_SUCCESS);
  std::vector<uint8_t> cleared(n, 0);
  for (uint32_t i = 0; i < n; ++i) {
    if (path_len) path_len[i] = 0;
    if (seeds[i] >= V) { codes[i] = MNAV_INVALID_START; continue; }
    if (targets[i] >= V) { codes[i] = MNAV_INVALID_GOAL; continue; }
    if (seeds[i] == targets[i]) { cleared[i] = 1; continue; }   // :252-255 SUCCESS right after clearing the maps
    PlanIn p{};
    for (int k = 0; k < 3; ++k) { p.seed[k] = kNone; p.target[k] = kNone; p.seed_d[k] = 0.f; p.seed_expands[k] = 1; p.target_expands[k] = 1; }
    p.seed[0] = seeds[i]; p.target[0] = targets[i]; p.seed_face = kNone;
    in.push_back(p); map.push_back(i);
  }
  // Longest plans first: a plan's work grows with the area its wave sweeps before it reaches the robot
  // vertex, i.e. with the squared seed-target distance.  Workgroups are dispatched in plan order, so when a
  // batch holds more plans than the device runs at once the short ones back-fill behind the long ones
  // instead of leaving a tail (results are mapped back through `map`).
  // engine: 0 = tiled rounds, 1 = band steps, 2 = persistent per-plan, 3 = auto, 5 = tile-batch (plan-vectorised, large batches)
  int engine = ctx->dij_engine;
  // paths only (nothing V-sized asked for, nothing kept resident): no finalize pass, predecessors along the path only
  ctx->lazy_paths = ctx->allow_lazy_paths && !dist_out && !pred_out && !want_vecmap && !ctx->resident_vecmap;
  {
    const uint32_t m0 = (uint32_t)in.size();
    // auto: one wave per (tile, 64 plans) for large batches, one workgroup per plan for medium ones, tile rounds otherwise
    if (engine == 3) {
      // The tile-batch engine advances all plans tile by tile, 16 plans per quarter of a wave (k_tb_solve_q).  Its floor is one
      // stream pass per iteration (~190 iterations x ~180 us on the 1M mesh, ~600 x 300 us at 10M), whatever the batch; above
      // that it beats the per-plan engines at every size measured (round 4, plans/s, tiled / persistent / tile-batch --
      // 1M: 64 plans 1084 / 496 / 1359, 256: 1879 / 1929 / 4392, 1024: 2223 / 7107 / 12871;
      // 10M: 64 plans 196 / 41 / 189, 256: 252 / 163 / 409, 1024: 254 / 594 / 924).
      const double tiles = std::max(1.0, (double)V / (0.9 * ctx->tb.T));
      const bool fills = m0 >= ctx->tb.min_batch && m0 <= 65535u && (double)m0 >= tiles / 1000.0;
      engine = fills ? 5 : (m0 >= ctx->persistent_min_batch) ? 2 : 0;
    }
    if (engine == 5 && m0 > 65535u) engine = 2;
    if (engine == 1 && offset < 0.0) engine = 0;                      // the band steps arm goal_dist inside the loop: tile rounds for a negative offset
  }
  if (engine == 5 && in.size() > 1) {
    // plans whose waves start close to each other are neighbours in the batch: their slices of a tile are adjacent in memory
    // and they tend to have work on the same tiles at the same time
    if (tb_build(ctx)) return MNAV_INTERNAL_ERROR;
    std::vector<uint32_t> ord(in.size());
    std::iota(ord.begin(), ord.end(), 0u);
    const std::vector<uint32_t>& vt = ctx->tb.vert_tile;
    std::stable_sort(ord.begin(), ord.end(), [&](uint32_t x, uint32_t y) { return vt[in[x].seed[0]] < vt[in[y].seed[0]]; });
    std::vector<PlanIn> in2(in.size()); std::vector<uint32_t> map2(in.size());
    for (size_t i = 0; i < in.size(); ++i) { in2[i] = in[ord[i]]; map2[i] = map[ord[i]]; }
    in.swap(in2); map.swap(map2);
  } else
  if (in.size() > 1 && ctx->h_xyz.size() == 3 * (size_t)V) {
    std::vector<uint32_t> ord(in.size());
    std::iota(ord.begin(), ord.end(), 0u);
    std::vector<float> est(in.size());
    for (size_t i = 0; i < in.size(); ++i) {
      const float* a = &ctx->h_xyz[3 * (size_t)in[i].seed[0]];
      const float* b = &ctx->h_xyz[3 * (size_t)in[i].target[0]];
      est[i] = (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]);
    }
    std::stable_sort(ord.begin(), ord.end(), [&](uint32_t x, uint32_t y) { return est[x] > est[y]; });
    std::vector<PlanIn> in2(in.size()); std::vector<uint32_t> map2(in.size());
    for (size_t i = 0; i < in.size(); ++i) { in2[i] = in[ord[i]]; map2[i] = map[ord[i]]; }
    in.swap(in2); map.swap(map2);
  }
  ctx->caller_slot.assign(n, kNone);
  for (size_t i = 0; i < map.size(); ++i) ctx->caller_slot[map[i]] = (uint32_t)i;
  (void)hipEventRecord(ctx->ev[0], ctx->stream);
  const uint32_t m = (uint32_t)in.size();
  ctx->tb.count_pending = false; ctx->tb_args_valid = false;
  ctx->last_planner = kPlannerDijkstra; ctx->last_n = 0;            // outputs of the previous call are gone; this call's count once its engine succeeded
  ctx->last_target.resize(m); for (uint32_t k = 0; k < m; ++k) ctx->last_target[k] = in[k].target[0];
  ctx->last_offset = offset;
  if (m) {
    if (materialize(ctx, false, cost_limit)) return MNAV_INTERNAL_ERROR;
    const bool want_path = true;
    const int rc = (engine == 0) ? run_dijkstra_tiled(ctx, m, in, offset)
                 : (engine == 2) ? run_dijkstra_persistent(ctx, m, in, offset)
                 : (engine == 6) ? run_dijkstra_async(ctx, m, in, offset)
                 : (engine == 5) ? run_dijkstra_tb(ctx, m, in, offset)
                                 : run_plans<kPlannerDijkstra>(ctx, m, in, offset, want_path);
    MTRACE("engine returned");
    if (rc != 0) (void)hipStreamSynchronize(ctx->stream);             // nothing of a failed / cancelled call stays in flight
    if (rc < 0) return MNAV_INTERNAL_ERROR;
    if (rc == 1) { for (uint32_t i = 0; i < n; ++i) if (codes_out) codes_out[i] = MNAV_CANCELED; return MNAV_CANCELED; }   // :350-354
    ctx->last_engine = engine; ctx->last_n = m;                       // only a call whose engine succeeded leaves outputs behind
    if (engine == 1) ctx->lazy_paths = false;                         // the band steps keep their predecessors as they go
    const PathRows rows1{ ctx->d_paths, ctx->path_stride, nullptr, nullptr };
    PathRows rows2{ nullptr, 0u, nullptr, nullptr };
    const uint32_t gc = (V + kBlock * 4 - 1) / (kBlock * 4);
    if (engine == 5 && ctx->lazy_paths) {
      hipLaunchKernelGGL(k_tb_path, dim3(m), dim3(kWave), 0, ctx->stream, ctx->tb_args, ctx->d_row_ptr, ctx->d_nbr, V, ctx->d_res, rows1, ctx->d_mismatch);
      ctx->tb.count_pending = true;                                  // settled vertices (a statistic): counted when somebody asks, mnav_get_stats
    } else if (ctx->lazy_paths) {
      hipLaunchKernelGGL(k_path_lazy, dim3(m), dim3(kWave), 0, ctx->stream, ctx->d_plans, ctx->d_tplans, ctx->d_res, rows1, ctx->d_mismatch);
      hipLaunchKernelGGL(k_count_goal, dim3(gc ? gc : 1, m), dim3(kBlock), 0, ctx->stream, ctx->d_plans, ctx->d_res);
    } else
    hipLaunchKernelGGL(k_finish<kPlannerDijkstra>, dim3(m), dim3(64), 0, ctx->stream, ctx->d_plans, ctx->d_res, rows1);
    if (engine == 1)   // the tile engines count the settled vertices in k_dij_finalize
      hipLaunchKernelGGL(k_count, dim3(gc ? gc : 1, m), dim3(kBlock), 0, ctx->stream, ctx->d_plans, ctx->d_res);
    (void)hipEventRecord(ctx->ev[4], ctx->stream);
    if (want_vecmap && !(engine == 5 && !ctx->lazy_paths))            // (the tile-batch engine's finalize pass writes the vector map itself)
      hipLaunchKernelGGL(k_vecmap_dijkstra, dim3(gc ? gc : 1, m), dim3(kBlock), 0, ctx->stream, ctx->d_plans, ctx->d_xyz, ctx->d_vecptrs);
    (void)hipEventRecord(ctx->ev[5], ctx->stream);
    if (hipMemcpyAsync(ctx->h_res, ctx->d_res, sizeof(PlanResult) * m, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) { ctx->err = "result download failed"; return MNAV_INTERNAL_ERROR; }
    {
      // paths longer than the default rows (corridors, mazes): ONLY those plans are walked again, into rows of exactly their
      // length (the first walk counted it) in one packed buffer
      size_t over_words = 0;
      std::vector<unsigned long long> ooff(m, 0ull); std::vector<uint32_t> ocap(m, 0u);
      for (uint32_t k = 0; k < m; ++k)
        if (ctx->h_res[k].code == kPathOverflow) { ooff[k] = over_words; ocap[k] = ctx->h_res[k].path_len; over_words += ctx->h_res[k].path_len; }
      if (over_words) {
        std::vector<PlanResult> keep(ctx->h_res, ctx->h_res + m);   // settled / evals were accumulated by other kernels
        (void)hipFree(ctx->d_over); (void)hipFree(ctx->d_over_off); (void)hipFree(ctx->d_over_cap);
        ctx->d_over = nullptr; ctx->d_over_off = nullptr; ctx->d_over_cap = nullptr;
        if (hipMalloc((void**)&ctx->d_over, 4 * over_words) != hipSuccess || hipMalloc((void**)&ctx->d_over_off, 8 * (size_t)m) != hipSuccess ||
            hipMalloc((void**)&ctx->d_over_cap, 4 * (size_t)m) != hipSuccess ||
            hipMemcpyAsync(ctx->d_over_off, ooff.data(), 8 * (size_t)m, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
            hipMemcpyAsync(ctx->d_over_cap, ocap.data(), 4 * (size_t)m, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
          { ctx->err = "path buffers: out of memory"; return MNAV_INTERNAL_ERROR; }
        rows2 = PathRows{ ctx->d_over, 0u, ctx->d_over_off, ctx->d_over_cap };
        if (engine == 5 && ctx->lazy_paths) hipLaunchKernelGGL(k_tb_path, dim3(m), dim3(kWave), 0, ctx->stream, ctx->tb_args, ctx->d_row_ptr, ctx->d_nbr, V, ctx->d_res, rows2, ctx->d_mismatch);
        else if (ctx->lazy_paths) hipLaunchKernelGGL(k_path_lazy, dim3(m), dim3(kWave), 0, ctx->stream, ctx->d_plans, ctx->d_tplans, ctx->d_res, rows2, ctx->d_mismatch);
        else hipLaunchKernelGGL(k_finish<kPlannerDijkstra>, dim3(m), dim3(64), 0, ctx->stream, ctx->d_plans, ctx->d_res, rows2);
        if (hipMemcpyAsync(ctx->h_res, ctx->d_res, sizeof(PlanResult) * m, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) { ctx->err = "result download failed"; return MNAV_INTERNAL_ERROR; }
        for (uint32_t k = 0; k < m; ++k) ctx->h_res[k].settled = keep[k].settled;
      }
    }
    if (ctx->last_engine != 1) {
      uint32_t mism = 0;
      if (hipMemcpy(&mism, ctx->d_mismatch, 4, hipMemcpyDeviceToHost) != hipSuccess || mism != 0) {
        ctx->err = "tiled SSSP did not reach its fixed point (" + std::to_string(mism) + " vertices)";
        return MNAV_INTERNAL_ERROR;
      }
    }
    MTRACE("results downloaded");
    // all vertex paths: packed and reversed on the device (k_pack_paths), one dense copy into a pinned buffer
    std::vector<uint32_t> offs(m + 1, 0), lens(m, 0);
    for (uint32_t k = 0; k < m; ++k) {
      lens[k] = (ctx->h_res[k].code == MNAV_SUCCESS) ? ctx->h_res[k].path_len : 0u;
      offs[k + 1] = offs[k] + lens[k];
    }
    const size_t total = offs[m];
    if (total && path_out && path_cap) {
      if (ctx->pack_words < total) {
        if (ctx->d_pack) (void)hipFree(ctx->d_pack);
        if (ctx->h_pack) (void)hipHostFree(ctx->h_pack);
        ctx->d_pack = nullptr; ctx->h_pack = nullptr; ctx->pack_words = 0;
        const size_t want = total + total / 4 + 1024;
        if (hipMalloc((void**)&ctx->d_pack, 4 * want) != hipSuccess || hipHostMalloc((void**)&ctx->h_pack, 4 * want, hipHostMallocDefault) != hipSuccess)
          { ctx->err = "path buffers: out of memory"; return MNAV_INTERNAL_ERROR; }
        ctx->pack_words = want;
      }
      if (ctx->pack_meta_n < 2 * (size_t)m) {
        if (ctx->d_pack_meta) (void)hipFree(ctx->d_pack_meta);
        ctx->d_pack_meta = nullptr; ctx->pack_meta_n = 0;
        if (hipMalloc((void**)&ctx->d_pack_meta, 4 * 2 * (size_t)m) != hipSuccess) { ctx->err = "path buffers: out of memory"; return MNAV_INTERNAL_ERROR; }
        ctx->pack_meta_n = 2 * (size_t)m;
      }
      if (hipMemcpyAsync(ctx->d_pack_meta, offs.data(), 4 * (size_t)m, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
          hipMemcpyAsync(ctx->d_pack_meta + m, lens.data(), 4 * (size_t)m, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        { ctx->err = "path download failed"; return MNAV_INTERNAL_ERROR; }
      hipLaunchKernelGGL(k_pack_paths, dim3(m), dim3(kBlock), 0, ctx->stream, rows1, rows2, ctx->d_pack_meta, ctx->d_pack_meta + m, ctx->d_pack);
      if (hipMemcpyAsync(ctx->h_pack, ctx->d_pack, 4 * total, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
          hipStreamSynchronize(ctx->stream) != hipSuccess) { ctx->err = "path download failed"; return MNAV_INTERNAL_ERROR; }
    }
    MTRACE("paths downloaded");
    for (uint32_t k = 0; k < m; ++k) {
      const uint32_t i = map[k];
      const PlanResult& r = ctx->h_res[k];
      codes[i] = r.code;
      if (r.code == MNAV_SUCCESS) {
        if (path_len) path_len[i] = r.path_len;
        if (path_out && path_cap && r.path_len)                         // reference list order: seed ... pred[target]
          memcpy(path_out + (size_t)i * path_cap, ctx->h_pack + offs[k], 4 * (size_t)std::min(r.path_len, path_cap));
      }
      if (dist_out && hipMemcpyAsync(dist_out + (size_t)i * V, ctx->slots[k].dist, 4 * (size_t)V, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
        { ctx->err = "dist download failed"; return MNAV_INTERNAL_ERROR; }
      if (pred_out && hipMemcpyAsync(pred_out + (size_t)i * V, ctx->slots[k].pred, 4 * (size_t)V, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
        { ctx->err = "pred download failed"; return MNAV_INTERNAL_ERROR; }
      if (vecmap_out && want_vecmap && hipMemcpyAsync(vecmap_out + (size_t)i * 3 * V, ctx->slots[k].vecmap, 12 * (size_t)V, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
        { ctx->err = "vecmap download failed"; return MNAV_INTERNAL_ERROR; }
    }
    MTRACE("paths copied out");
    (void)hipEventRecord(ctx->ev[6], ctx->stream);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) { ctx->err = "sync failed"; return MNAV_INTERNAL_ERROR; }
    finish_stats(ctx, m, false);
    MTRACE("stats done");
  }
  // plans rejected before reaching the device: the reference has cleared its maps by then
  for (uint32_t i = 0; i < n; ++i) {
    if (codes[i] == MNAV_INVALID_START || codes[i] == MNAV_INVALID_GOAL || cleared[i]) {
      if (dist_out) for (uint32_t v = 0; v < V; ++v) dist_out[(size_t)i * V + v] = INFINITY;
      if (pred_out) for (uint32_t v = 0; v < V; ++v) pred_out[(size_t)i * V + v] = v;
    }
    if (codes_out) codes_out[i] = codes[i];
    if (codes[i] != MNAV_SUCCESS && worst == MNAV_SUCCESS) worst = codes[i];
  }
  return worst;
}

uint32_t mnav_plan_dijkstra(mnav_ctx* ctx, uint32_t seed_vertex, uint32_t target_vertex, double goal_dist_offset, double cost_limit,
                            float* dist_out, uint32_t* pred_out, uint32_t* path_out, uint32_t path_cap, uint32_t* path_len,
                            float* vecmap_out)
{
  if (!ctx) return MNAV_INTERNAL_ERROR;
  uint32_t code = MNAV_INTERNAL_ERROR;
  uint32_t len = 0;
  const uint32_t rc = dijkstra_impl(ctx, 1, &seed_vertex, &target_vertex, goal_dist_offset, cost_limit, &code, dist_out, pred_out,
                                    path_out, path_cap, &len, vecmap_out, true);
  if (path_len) *path_len = len;
  return rc == MNAV_INTERNAL_ERROR || rc == MNAV_CANCELED ? rc : code;
}

uint32_t mnav_plan_dijkstra_batch(mnav_ctx* ctx, uint32_t n, const uint32_t* seeds, const uint32_t* targets, double goal_dist_offset,
                                  double cost_limit, uint32_t* codes_out, float* dist_out, uint32_t* pred_out, uint32_t* path_out,
                                  uint32_t path_cap, uint32_t* path_len)
{
  if (!ctx) return MNAV_INTERNAL_ERROR;
  if (n == 0) return MNAV_SUCCESS;
  if (!seeds || !targets) { ctx->err = "null seeds/targets"; return MNAV_INTERNAL_ERROR; }
  return dijkstra_impl(ctx, n, seeds, targets, goal_dist_offset, cost_limit, codes_out, dist_out, pred_out, path_out, path_cap,
                       path_len, nullptr, false);
}

static uint32_t cvp_impl(mnav_ctx* ctx, uint32_t n, const float* seed_pos, const uint32_t* seed_faces, const uint32_t* target_faces,
                         double goal_dist_offset, double cost_limit, uint32_t* codes_out, float* dist_out, uint32_t* pred_out,
                         float* direction_out, uint32_t* cutface_out, float* vecmap_out)
{
  if (check_ready(ctx)) return MNAV_INTERNAL_ERROR;
  ctx->err.clear();
  if (!(goal_dist_offset >= 0.0)) { ctx->err = "goal_dist_offset must be >= 0"; return MNAV_INTERNAL_ERROR; }   // see dijkstra_impl
  ctx->cancel.store(0);                                               // cvp :679
  if (ctx->d_cancel) { (void)hipStreamSynchronize(ctx->cancel_stream); (void)hipMemsetAsync(ctx->d_cancel, 0, 4, ctx->stream); }
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return MNAV_INTERNAL_ERROR; }
  const uint32_t V = ctx->V;
  if (!ctx->have_normals) { ctx->err = "vertex normals were not uploaded"; return MNAV_INTERNAL_ERROR; }
  if (V > (1u << kKeyIdBits)) { ctx->err = "CVP pop keys hold 26-bit vertex ids: mesh too large"; return MNAV_INTERNAL_ERROR; }
  std::vector<uint32_t> codes(n, MNAV_SUCCESS), map;
  std::vector<PlanIn> in;
  std::vector<float> sp;
  for (uint32_t i = 0; i < n; ++i) {
    if (seed_faces[i] >= ctx->F) { codes[i] = MNAV_INVALID_START; continue; }      // cvp :681-685
    if (target_faces[i] >= ctx->F) { codes[i] = MNAV_INVALID_GOAL; continue; }     // cvp :686-690
    PlanIn p{};
    for (int k = 0; k < 3; ++k) {
      p.seed[k] = ctx->h_faces[3 * (size_t)seed_faces[i] + k];
      p.target[k] = ctx->h_faces[3 * (size_t)target_faces[i] + k];
    }
    for (int k = 0; k < 3; ++k) {
      // cvp :721-723  diff = start - vertex; dist = diff.length()  (float arithmetic)
      const float dx = seed_pos[3 * (size_t)i] - ctx->h_xyz[3 * (size_t)p.seed[k]];
      const float dy = seed_pos[3 * (size_t)i + 1] - ctx->h_xyz[3 * (size_t)p.seed[k] + 1];
      const float dz = seed_pos[3 * (size_t)i + 2] - ctx->h_xyz[3 * (size_t)p.seed[k] + 2];
      const float l2 = dx * dx + dy * dy + dz * dz;
      p.seed_d[k] = sqrtf(l2);
      p.seed_expands[k] = (!((double)ctx->h_cost[p.seed[k]] >= cost_limit) && !ctx->h_invalid[p.seed[k]]) ? 1u : 0u;       // cvp :757,760
      p.target_expands[k] = (!((double)ctx->h_cost[p.target[k]] >= cost_limit) && !ctx->h_invalid[p.target[k]]) ? 1u : 0u;
    }
    p.seed_face = seed_faces[i];
    in.push_back(p); map.push_back(i);
    sp.insert(sp.end(), seed_pos + 3 * (size_t)i, seed_pos + 3 * (size_t)i + 3);
  }
  const uint32_t m = (uint32_t)in.size();
  ctx->caller_slot.assign(n, kNone);
  for (size_t i = 0; i < map.size(); ++i) ctx->caller_slot[map[i]] = (uint32_t)i;
  (void)hipEventRecord(ctx->ev[0], ctx->stream);
  ctx->tb.count_pending = false; ctx->tb_args_valid = false;         // (a lazy settled-vertex count of an earlier Dijkstra batch is void now)
  ctx->last_planner = kPlannerCvp; ctx->last_n = 0; ctx->last_engine = 1;
  uint32_t worst = MNAV_SUCCESS;
  if (m) {
    if (materialize(ctx, true, cost_limit)) return MNAV_INTERNAL_ERROR;
    if (ctx->seed_pos_cap < m) {
      (void)hipFree(ctx->d_seed_pos); ctx->d_seed_pos = nullptr;
      if (hipMalloc((void**)&ctx->d_seed_pos, 12 * (size_t)m) != hipSuccess) { ctx->err = "alloc failed"; return MNAV_INTERNAL_ERROR; }
      ctx->seed_pos_cap = m;
    }
    if (hipMemcpyAsync(ctx->d_seed_pos, sp.data(), 12 * (size_t)m, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
      { ctx->err = "seed upload failed"; return MNAV_INTERNAL_ERROR; }
    const int rc = run_plans<kPlannerCvp>(ctx, m, in, goal_dist_offset, false);
    if (rc != 0) (void)hipStreamSynchronize(ctx->stream);
    if (rc < 0) return MNAV_INTERNAL_ERROR;
    if (rc == 1) { if (codes_out) for (uint32_t i = 0; i < n; ++i) codes_out[i] = MNAV_CANCELED; return MNAV_CANCELED; }   // cvp :888-892
    ctx->last_n = m;
    hipLaunchKernelGGL(k_finish<kPlannerCvp>, dim3(m), dim3(64), 0, ctx->stream, ctx->d_plans, ctx->d_res, PathRows{ nullptr, 0u, nullptr, nullptr });
    const uint32_t gc = (V + kBlock * 4 - 1) / (kBlock * 4);
    hipLaunchKernelGGL(k_count, dim3(gc ? gc : 1, m), dim3(kBlock), 0, ctx->stream, ctx->d_plans, ctx->d_res);
    (void)hipEventRecord(ctx->ev[4], ctx->stream);
    hipLaunchKernelGGL(k_vecmap_cvp, dim3(gc ? gc : 1, m), dim3(kBlock), 0, ctx->stream, ctx->d_plans, ctx->d_xyz, ctx->d_nrm,
                       ctx->d_vecptrs, ctx->d_seed_pos);                // cvp :897
    (void)hipEventRecord(ctx->ev[5], ctx->stream);
    bool ok = hipMemcpyAsync(ctx->h_res, ctx->d_res, sizeof(PlanResult) * m, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
    for (uint32_t k = 0; k < m && ok; ++k) {
      const uint32_t i = map[k];
      Slot& s = ctx->slots[k];
      if (ok && dist_out) ok = hipMemcpyAsync(dist_out + (size_t)i * V, s.dist, 4 * (size_t)V, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
      if (ok && pred_out) ok = hipMemcpyAsync(pred_out + (size_t)i * V, s.pred, 4 * (size_t)V, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
      if (ok && direction_out) ok = hipMemcpyAsync(direction_out + (size_t)i * V, s.dirn, 4 * (size_t)V, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
      if (ok && cutface_out) ok = hipMemcpyAsync(cutface_out + (size_t)i * V, s.cutf, 4 * (size_t)V, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
      if (ok && vecmap_out) ok = hipMemcpyAsync(vecmap_out + (size_t)i * 3 * V, s.vecmap, 12 * (size_t)V, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
    }
    (void)hipEventRecord(ctx->ev[6], ctx->stream);
    if (!ok || hipStreamSynchronize(ctx->stream) != hipSuccess) { ctx->err = "output download failed"; return MNAV_INTERNAL_ERROR; }
    finish_stats(ctx, m, true);
    for (uint32_t k = 0; k < m; ++k) {
      codes[map[k]] = ctx->h_res[k].code;
      if (ctx->h_res[k].code == MNAV_INTERNAL_ERROR) {
        const uint32_t o = ctx->h_res[k].overflow;
        ctx->err = (o & 8u) ? "CVP: a walk over the cascade tree hit its bound on the converged state (pop order not guaranteed)"
                 : (o & 16u) ? "CVP: verification sweep found a vertex that is not a fixed point of the gather rule"
                             : "CVP wavefront did not converge (step cap)";
      }
    }
  }
  for (uint32_t i = 0; i < n; ++i) {
    if (codes_out) codes_out[i] = codes[i];
    if (codes[i] != MNAV_SUCCESS && worst == MNAV_SUCCESS) worst = codes[i];
  }
  return worst;
}

uint32_t mnav_plan_cvp(mnav_ctx* ctx, const float seed_pos[3], uint32_t seed_face, uint32_t target_face, double goal_dist_offset,
                       double cost_limit, float* dist_out, uint32_t* pred_out, float* direction_out, uint32_t* cutface_out,
                       float* vecmap_out)
{
  if (!ctx) return MNAV_INTERNAL_ERROR;
  if (!seed_pos) return MNAV_INVALID_START;
  uint32_t code = MNAV_INTERNAL_ERROR;
  const uint32_t rc = cvp_impl(ctx, 1, seed_pos, &seed_face, &target_face, goal_dist_offset, cost_limit, &code, dist_out, pred_out,
                               direction_out, cutface_out, vecmap_out);
  return (rc == MNAV_INTERNAL_ERROR || rc == MNAV_CANCELED) ? rc : code;
}

uint32_t mnav_plan_cvp_batch(mnav_ctx* ctx, uint32_t n, const float* seed_pos, const uint32_t* seed_faces, const uint32_t* target_faces,
                             double goal_dist_offset, double cost_limit, uint32_t* codes_out, float* dist_out, uint32_t* pred_out,
                             float* vecmap_out)
{
  if (!ctx) return MNAV_INTERNAL_ERROR;
  if (n == 0) return MNAV_SUCCESS;
  if (!seed_pos || !seed_faces || !target_faces) { ctx->err = "null seeds/targets"; return MNAV_INTERNAL_ERROR; }
  return cvp_impl(ctx, n, seed_pos, seed_faces, target_faces, goal_dist_offset, cost_limit, codes_out, dist_out, pred_out, nullptr, nullptr,
                  vecmap_out);
}

#include "mnav_shard_capi.h"   // mnav_shard_* (one plan over several GPUs)

void mnav_cancel(mnav_ctx* ctx)
{
  if (!ctx) return;
  ctx->cancel.store(1, std::memory_order_relaxed);
  // thread-safe (MBF calls cancel() from the action-server thread), fire and forget: SDMA copy beside the running kernel
  if (ctx->d_cancel && ctx->h_one) (void)hipMemcpyAsync(ctx->d_cancel, ctx->h_one, 4, hipMemcpyHostToDevice, ctx->cancel_stream);
}

// The settled-vertex count of a paths-only tile-batch call is instrumentation (it only feeds mnav_stats.settled and
// mnav_algorithmic_bytes): it is taken from the resident distances on the first query after the call, not inside it.
static void settle_stats(mnav_ctx* ctx)
{
  if (!ctx->tb.count_pending) return;
  ctx->tb.count_pending = false;
  // only the arguments of a tile-batch call that succeeded and whose buffers are still allocated may be dereferenced
  if (!ctx->tb_args_valid || ctx->last_planner != kPlannerDijkstra || ctx->last_engine != 5 || !ctx->tb.built) return;
  const uint32_t m = ctx->last_n;
  if (!m || hipSetDevice(ctx->device) != hipSuccess) return;
  hipLaunchKernelGGL(k_tb_count, dim3(ctx->tb.ntiles ? ctx->tb.ntiles : 1, 16), dim3(kBlock), 0, ctx->stream, ctx->tb_args, ctx->tb.T, ctx->d_res);
  if (hipMemcpyAsync(ctx->h_res, ctx->d_res, sizeof(PlanResult) * m, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess) return;
  finish_stats(ctx, m, false);
}

int mnav_get_stats(const mnav_ctx* ctx, mnav_stats* out)
{
  if (!ctx || !out) return -1;
  settle_stats(const_cast<mnav_ctx*>(ctx));
  *out = ctx->stats;
  return 0;
}

int mnav_get_timing(const mnav_ctx* ctx, mnav_stats* out)
{
  if (!ctx || !out) return -1;
  *out = ctx->stats;
  if (ctx->tb.count_pending) out->settled = 0;
  return 0;
}

int mnav_set_band_width(mnav_ctx* ctx, float delta)
{
  if (!ctx) return -1;
  ctx->delta_user = delta > 0.f ? delta : 0.f;
  return 0;
}

int mnav_set_dijkstra_engine(mnav_ctx* ctx, int engine)
{
  if (!ctx || engine < 0 || engine > 6 || engine == 4) return -1;   // 4 was the one-wave-per-plan experiment (removed)
  ctx->dij_engine = engine;
  return 0;
}

const void* mnav_device_output(const mnav_ctx* ctx, uint32_t slot, int what)
{
  if (!ctx) return nullptr;
  if (slot >= ctx->caller_slot.size()) return nullptr;                    // not a plan of the last call (a stale slot of an earlier, larger batch is never handed out)
  slot = ctx->caller_slot[slot];                                          // caller's plan index -> device slot (kNone: never reached the device)
  if (slot >= ctx->last_n || slot >= ctx->slots.size()) return nullptr;   // last_n == 0: the last call failed or was cancelled
  const Slot& s = ctx->slots[slot];
  // a paths-only Dijkstra call finalized nothing: values beyond goal_dist are engine-tentative, predecessors were derived
  // along the path only (mnav_download_output what = 5 returns the popped potential of such a call)
  const bool lazy = ctx->lazy_paths && ctx->last_planner == kPlannerDijkstra;
  switch (what) {
    case 0: return lazy ? nullptr : s.dist;
    case 1: return lazy ? nullptr : s.pred;
    case 2: return s.dirn;
    case 3: return s.cutf;
    case 4: return (ctx->last_planner == kPlannerDijkstra && !ctx->want_vec) ? nullptr : s.vecmap;   // (a Dijkstra call that was not asked for vector maps
                                                                                                       //  must not hand out an earlier call's)
    default: return nullptr;
  }
}

int mnav_set_resident_outputs(mnav_ctx* ctx, int on)
{
  if (!ctx) return -1;
  ctx->resident_vecmap = on != 0;
  return 0;
}

int mnav_download_output(mnav_ctx* ctx, uint32_t slot, int what, void* host_out)
{
  if (!ctx || !host_out) return -1;
  if (what == 5) {                                                   // popped potential (Dijkstra): exact wherever dist <= goal_dist, +inf elsewhere
    if (ctx->last_planner != kPlannerDijkstra) { ctx->err = "popped potential: the last call was not a Dijkstra plan"; return -1; }
    if (slot < ctx->caller_slot.size()) slot = ctx->caller_slot[slot];
    if (slot >= ctx->last_n) { ctx->err = "output not resident"; return -1; }
    if (hipSetDevice(ctx->device) != hipSuccess) return -1;
    float* tmp = nullptr;
    HIPCHK(hipMalloc((void**)&tmp, 4 * (size_t)(ctx->V ? ctx->V : 1)));
    const uint32_t g = std::min<uint32_t>((ctx->V + kBlock - 1) / kBlock + 1, 4096);
    if (ctx->last_engine == 5 && ctx->lazy_paths && !ctx->tb_args_valid) { (void)hipFree(tmp); ctx->err = "output not resident"; return -1; }
    if (ctx->last_engine == 5 && ctx->lazy_paths) hipLaunchKernelGGL(k_tb_popped, dim3(g), dim3(kBlock), 0, ctx->stream, ctx->tb_args, slot, ctx->V, tmp);
    else hipLaunchKernelGGL(k_popped, dim3(g), dim3(kBlock), 0, ctx->stream, ctx->slots[slot].dist, ctx->last_target[slot], ctx->last_offset, ctx->V, tmp);
    const hipError_t e1 = hipMemcpyAsync(host_out, tmp, 4 * (size_t)ctx->V, hipMemcpyDeviceToHost, ctx->stream);
    const hipError_t e2 = hipStreamSynchronize(ctx->stream);
    (void)hipFree(tmp);
    if (e1 != hipSuccess || e2 != hipSuccess) { ctx->err = "popped potential: copy failed"; return -1; }
    return 0;
  }
  const void* src = mnav_device_output(ctx, slot, what);
  if (!src) { ctx->err = "output not resident"; return -1; }
  if (hipSetDevice(ctx->device) != hipSuccess) return -1;
  const size_t bytes = (what == 4 ? 12 : 4) * (size_t)ctx->V;
  HIPCHK(hipMemcpyAsync(host_out, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

// MeshMap::directionAtPosition (mesh_map.cpp:625-650) on the resident vector map of plan `slot`: what the controller
// samples at the robot pose.  36 bytes cross PCIe instead of the 12 MB field.  A vertex counts as "has a vector" when
// its entry is not the all-zero vector the vector-map kernels write for vertices the wave did not set.
int mnav_vector_at(mnav_ctx* ctx, uint32_t slot, const uint32_t vs[3], const float bary[3], float out[3])
{
  if (!ctx || !vs || !bary || !out) return -1;
  for (int k = 0; k < 3; ++k) if (vs[k] >= ctx->V) { ctx->err = "vertex id out of range"; return -1; }
  float v[3][3];
  if (ctx->last_planner == kPlannerDijkstra && ctx->last_engine == 5 && ctx->lazy_paths && ctx->tb_args_valid) {
    // paths-only batch of the tile-batch engine: no vector map was written -- the three entries are derived from the blocked distances
    uint32_t p = slot;
    if (p < ctx->caller_slot.size()) p = ctx->caller_slot[p];
    if (p >= ctx->last_n) { ctx->err = "output not resident"; return -1; }
    if (hipSetDevice(ctx->device) != hipSuccess) return -1;
    if (!ctx->d_vec3) HIPCHK(hipMalloc((void**)&ctx->d_vec3, 64));
    hipLaunchKernelGGL(k_tb_vector3, dim3(3), dim3(kWave), 0, ctx->stream, ctx->tb_args, ctx->d_row_ptr, ctx->d_nbr, ctx->d_xyz, p,
                       make_uint3(vs[0], vs[1], vs[2]), ctx->d_vec3);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(&v[0][0], ctx->d_vec3, 36, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  } else {
    const float* vm = static_cast<const float*>(mnav_device_output(ctx, slot, 4));
    if (!vm) { ctx->err = "vector map not resident"; return -1; }
    if (hipSetDevice(ctx->device) != hipSuccess) return -1;
    for (int k = 0; k < 3; ++k) HIPCHK(hipMemcpyAsync(v[k], vm + 3 * (size_t)vs[k], 12, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  float acc[3] = { 0.f, 0.f, 0.f };
  bool any = false;
  for (int k = 0; k < 3; ++k) {
    const bool has = !(v[k][0] == 0.f && v[k][1] == 0.f && v[k][2] == 0.f);
    if (!has) continue;
    any = true;
    for (int c = 0; c < 3; ++c) acc[c] += v[k][c] * bary[k];       // :636-638
  }
  if (!any || !(std::isfinite(acc[0]) && std::isfinite(acc[1]) && std::isfinite(acc[2]))) return 0;   // :639-646
  out[0] = acc[0]; out[1] = acc[1]; out[2] = acc[2];
  return 1;
}

// f3 of SURVEY.md section 8: the consumer of the CVP vector field on the device.  Plan i of the last mnav_plan_cvp(_batch)
// call is walked from its target (the robot) back to its seed (the goal) over the vector map resident in HBM; what
// crosses PCIe is the path, not the 12 B/vertex field.
int mnav_backtrack_cvp_batch(mnav_ctx* ctx, uint32_t n, const float* seed_pos, const uint32_t* seed_faces, const float* target_pos,
                             const uint32_t* target_faces, double step_width, int32_t inflation_layer, uint32_t cap, float* positions_out,
                             uint32_t* faces_out, uint32_t* n_out, int32_t* status_out)
{
  if (!ctx) return -1;
  ctx->err.clear();
  if (!n) return 0;
  if (!seed_pos || !seed_faces || !target_pos || !target_faces || !positions_out || !faces_out || !n_out || !status_out) { ctx->err = "null argument"; return -1; }
  if (cap < 2 || (uint64_t)cap * n > (1ull << 28)) { ctx->err = "path capacity out of range"; return -1; }
  if (!(step_width > 0.0)) { ctx->err = "step_width must be positive"; return -1; }   // a zero step never leaves the start
  if (ctx->last_planner != kPlannerCvp) { ctx->err = "back-tracking: the last call was not a CVP plan"; return -1; }
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return -1; }
  if (n != ctx->caller_slot.size()) { ctx->err = "back-tracking: n differs from the last mnav_plan_cvp(_batch) call"; return -1; }
  std::vector<WalkJob> jobs(n);
  for (uint32_t i = 0; i < n; ++i) {
    const float* vm = static_cast<const float*>(mnav_device_output(ctx, i, 4));
    jobs[i].vecmap = nullptr;
    if (ctx->caller_slot[i] == kNone) continue;                       // a plan rejected before it reached the device (INVALID_START / _GOAL): status 0, no entries
    if (!vm) { ctx->err = "vector map not resident (mnav_set_resident_outputs, or pass vecmap_out to the plan call)"; return -1; }
    if (seed_faces[i] >= ctx->F || target_faces[i] >= ctx->F) { ctx->err = "face id out of range"; return -1; }
    jobs[i].vecmap = vm; jobs[i].seed_face = seed_faces[i]; jobs[i].target_face = target_faces[i];
    for (int k = 0; k < 3; ++k) { jobs[i].seed[k] = seed_pos[3 * (size_t)i + k]; jobs[i].target[k] = target_pos[3 * (size_t)i + k]; }
  }
  WalkInflation L{};
  if (inflation_layer >= 0) {
    if ((size_t)inflation_layer >= ctx->layers.size() || !ctx->layers[inflation_layer].ready || !ctx->layers[inflation_layer].dist || !ctx->layers[inflation_layer].have_vec) {
      ctx->err = "back-tracking: not a resident inflation layer with a vector field"; return -1;
    }
    const mnav_ctx::Layer& Ly = ctx->layers[inflation_layer];
    L.distances = Ly.dist; L.vectors = Ly.vec; L.has_vector = Ly.vstate;
    L.inflation_radius = Ly.inflation_radius; L.inscribed_radius = Ly.inscribed_radius; L.inscribed_value = Ly.inscribed_value; L.lethal_value = Ly.lethal_value;
    L.repulsive_field = 1;
  }
  if (!ctx->walk_mesh_valid) {
    if (dev_upload(ctx, &ctx->d_faces, ctx->h_faces.data(), ctx->h_faces.size())) return -1;
    if (dev_upload(ctx, &ctx->d_vf_ptr, ctx->h_vf_ptr.data(), ctx->h_vf_ptr.size())) return -1;
    if (dev_upload(ctx, &ctx->d_vf, ctx->h_vf.data(), ctx->h_vf.size())) return -1;
    ctx->walk_mesh_valid = true;
  }
  const size_t need = (size_t)cap * n;
  if (need > ctx->walk_cap) {
    (void)hipFree(ctx->d_walk_pos); (void)hipFree(ctx->d_walk_face); ctx->d_walk_pos = nullptr; ctx->d_walk_face = nullptr; ctx->walk_cap = 0;
    HIPCHK(hipMalloc((void**)&ctx->d_walk_pos, 12 * need)); HIPCHK(hipMalloc((void**)&ctx->d_walk_face, 4 * need));
    ctx->walk_cap = need;
  }
  if (n > ctx->walk_jobs_cap) {
    (void)hipFree(ctx->d_walk_jobs); (void)hipFree(ctx->d_walk_ctl); ctx->d_walk_jobs = nullptr; ctx->d_walk_ctl = nullptr; ctx->walk_jobs_cap = 0;
    HIPCHK(hipMalloc((void**)&ctx->d_walk_jobs, sizeof(WalkJob) * (size_t)n)); HIPCHK(hipMalloc((void**)&ctx->d_walk_ctl, 8 * (size_t)n));
    ctx->walk_jobs_cap = n;
  }
  HIPCHK(hipMemcpyAsync(ctx->d_walk_jobs, jobs.data(), sizeof(WalkJob) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  WalkMesh M{ ctx->d_xyz, ctx->d_faces, ctx->d_vf_ptr, ctx->d_vf, ctx->V, ctx->F };
  hipLaunchKernelGGL(k_backtrack, dim3(n), dim3(64), 0, ctx->stream, M, L, ctx->d_walk_jobs, step_width, cap, ctx->d_walk_pos, ctx->d_walk_face, ctx->d_walk_ctl);
  HIPCHK(hipGetLastError());
  std::vector<int32_t> ctl(2 * (size_t)n);
  HIPCHK(hipMemcpyAsync(ctl.data(), ctx->d_walk_ctl, 8 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  for (uint32_t i = 0; i < n; ++i) {                                  // the walked entries only: O(path) bytes per plan
    const uint32_t m = (uint32_t)ctl[2 * (size_t)i + 1];
    if (m) {
      HIPCHK(hipMemcpyAsync(positions_out + 3 * (size_t)cap * i, ctx->d_walk_pos + 3 * (size_t)cap * i, 12 * (size_t)m, hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(hipMemcpyAsync(faces_out + (size_t)cap * i, ctx->d_walk_face + (size_t)cap * i, 4 * (size_t)m, hipMemcpyDeviceToHost, ctx->stream));
    }
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t m = (uint32_t)ctl[2 * (size_t)i + 1];
    float* pp = positions_out + 3 * (size_t)cap * i; uint32_t* pf = faces_out + (size_t)cap * i;
    for (uint32_t a = 0, b = m ? m - 1 : 0; a < b; ++a, --b) {        // the reference push_front()s: list order is seed first
      for (int k = 0; k < 3; ++k) std::swap(pp[3 * (size_t)a + k], pp[3 * (size_t)b + k]);
      std::swap(pf[a], pf[b]);
    }
    n_out[i] = m; status_out[i] = ctl[2 * (size_t)i];
  }
  return 0;
}

int mnav_backtrack_cvp(mnav_ctx* ctx, const float seed_pos[3], uint32_t seed_face, const float target_pos[3], uint32_t target_face, double step_width,
                       int32_t inflation_layer, uint32_t cap, float* positions_out, uint32_t* faces_out, uint32_t* n_out)
{
  int32_t status = 0;
  const int rc = mnav_backtrack_cvp_batch(ctx, 1, seed_pos, &seed_face, target_pos, &target_face, step_width, inflation_layer, cap, positions_out, faces_out, n_out, &status);
  return rc < 0 ? rc : status;
}

uint64_t mnav_algorithmic_bytes(const mnav_ctx* ctx)
{
  if (!ctx) return 0;
  settle_stats(const_cast<mnav_ctx*>(ctx));
  return ctx->algo_bytes;
}

#ifdef MNAV_TILE_TIMING
int mnav_debug_tile_timing(unsigned long long* out, unsigned int cap)
{
  unsigned int n = 0;
  (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_tile_timing_n), sizeof(n));
  if (n > 4096) n = 4096;
  if (n > cap) n = cap;
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tile_timing), sizeof(unsigned long long) * 8 * n);
  unsigned int z = 0;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tile_timing_n), &z, sizeof(z));
  return (int)n;
}
#endif

}  // extern "C"

#ifdef MNAV_WIDE_TIMING
extern "C" int mnav_debug_wide_timing(unsigned long long* out)
{
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wide_timing), sizeof(unsigned long long) * 12) != hipSuccess) return -1;
  unsigned long long z[12] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
  return hipMemcpyToSymbol(HIP_SYMBOL(g_wide_timing), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
