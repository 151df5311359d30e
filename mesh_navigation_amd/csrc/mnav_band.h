// mnav_band.h -- the distance-band gather engine's step kernel (k_step: CVP, the inflation wave, Dijkstra on request), the wide CVP
// batch kernels (mnav_cvp_wide.h, included here) and the CVP verification sweep.  Included by mnav.hip inside its anonymous
// namespace; not a stand-alone header.
#pragma once

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
  if (cur.repair == 5) return;                                       // idle: the host runs the exact band routine next (controller_core)
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

// The exact band routine (mnav_eval.h: exact_*), run by the host between two chunks of steps when a plan's control block says
// exact_wanted: ONE wave.  Lanes work side by side over the reset list, over the candidate list (minimum by key) and over the
// corner neighbours of a popped vertex; an evaluation reads only final state and its own vertex, so the lanes of a round do not
// depend on each other -- all evaluate, barrier, all store.  Lists: the plan's two work-list buffers (idle while the steps idle).
__global__ __launch_bounds__(kWave) void k_exact_band(const Plan* __restrict__ plans, uint32_t* __restrict__ pops_out)
{
  const Plan& P = plans[blockIdx.x];
  const int lane = threadIdx.x;
  const Ctl a = P.ctl[0], b = P.ctl[1];
  const int slot = (a.it > b.it) ? 0 : 1;
  const Ctl cur = slot == 0 ? a : b;
  if (!cur.exact_wanted || cur.done) return;
  Ctl x = exact_ctl(cur);
  uint32_t* const A = P.list[0];
  uint32_t* const C = P.list[1];
  __shared__ uint32_t s_nA, s_nC;
  if (lane == 0) { s_nA = 0u; s_nC = 0u; }
  __syncthreads();
  const uint32_t tagA = kExactStamp - 2u, tagC = kExactStamp - 1u;
  auto add = [&](uint32_t* L, uint32_t* n, uint32_t tag, uint32_t v) {
    if (v == kNone || v >= P.V) return;
    if (atomicExch(&P.stamp[v], tag) == tag) return;                  // listed already
    const uint32_t at = atomicAdd(n, 1u);
    if (at < P.cap) L[at] = v;
  };
  // 1. the band's vertices and their corner neighbours ...
  for (uint32_t base = 0; base < P.V; base += kWave) {
    const uint32_t v = base + (uint32_t)lane;
    if (v >= P.V || is_seed(P, v)) continue;
    const float t = key_time(P.tkey[v]);
    if (!(t >= cur.thr_fixed && t < cur.thr)) continue;
    add(A, &s_nA, tagA, v);
    for (uint32_t i = P.crn_ptr[v]; i < P.crn_ptr[v + 1]; ++i) { const Corner k = P.crn[i]; if (k.v1 == kNone) continue; add(A, &s_nA, tagA, k.v1); add(A, &s_nA, tagA, k.v2); }
  }
  __syncthreads();
  const uint32_t nA = min(s_nA, P.cap);
  // ... forget what was derived from provisional supports ...
  for (uint32_t i = lane; i < nA; i += kWave) {
    const uint32_t v = A[i];
    if (is_seed(P, v) || P.blocked[v] || key_time(P.tkey[v]) < cur.thr_fixed) continue;
    P.dist[v] = inf_f(); P.pred[v] = v; P.tkey[v] = key_inf(); P.dirn[v] = 0.0f; P.cutf[v] = kNone; if (P.keyd) P.keyd[v] = inf_f();
  }
  __threadfence(); __syncthreads();
  // ... and evaluate them on settled supports only
  for (uint32_t base = 0; base < nA; base += kWave) {
    const uint32_t i = base + (uint32_t)lane;
    const uint32_t v = i < nA ? A[i] : kNone;
    const ExactEval r = exact_eval(P, x, v);
    __syncthreads();
    if (exact_store(P, x, v, r)) add(C, &s_nC, tagC, v);
    __threadfence(); __syncthreads();
  }
  // 2. one pop at a time
  uint32_t pops = 0;
  for (;;) {
    const uint32_t nC = min(s_nC, P.cap);
    uint32_t best = kNone;
    for (uint32_t i = lane; i < nC; i += kWave) { const uint32_t u = C[i]; if (exact_better(P, x, u, best)) best = u; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint32_t other = (uint32_t)__shfl_xor((int)best, o);
      if (other != kNone && (best == kNone || (other != best && key_less(P, key_ref(P, other), key_ref(P, best))))) best = other;
    }
    best = (uint32_t)__shfl((int)best, 0);
    if (best == kNone) break;
    x.bound_v = best; ++pops;
    const uint32_t beg = P.crn_ptr[best], n2 = 2u * (P.crn_ptr[best + 1] - beg);
    for (uint32_t base = 0; base < n2; base += kWave) {
      const uint32_t i = base + (uint32_t)lane;
      uint32_t w = kNone;
      if (i < n2) { const Corner k = P.crn[beg + (i >> 1)]; if (k.v1 != kNone) w = (i & 1u) ? k.v2 : k.v1; }
      const ExactEval r = exact_eval(P, x, w);
      __syncthreads();
      if (exact_store(P, x, w, r)) add(C, &s_nC, tagC, w);              // (a vertex met in two corners is stored twice with the same state)
      __threadfence(); __syncthreads();
    }
  }
  // 3. back to the band steps
  if (lane == 0) { P.ctl[slot] = exact_done_ctl(cur); if (pops_out) atomicAdd(pops_out, pops); }
}

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

// the vertices around each plan's seed face once more, with BOTH fire events of the faces that have a seed support
// (mnav_eval.h: corner_fire_second, seed_ring_fix): predecessor / direction / cutting face of the last successful application
__global__ __launch_bounds__(kWave) void k_cvp_seed_ring(const Plan* __restrict__ plans)
{
  const Plan& P = plans[blockIdx.x];
  const Ctl a = P.ctl[0], b = P.ctl[1];
  const Ctl cur = (a.it > b.it) ? a : b;
  if (!cur.done || cur.overflow) return;
  for (int q = 0; q < 3; ++q) {
    const uint32_t s = P.seed[q];
    if (s == kNone || s >= P.V) continue;
    const uint32_t beg = P.crn_ptr[s], end = P.crn_ptr[s + 1];
    for (uint32_t i = beg + (threadIdx.x >> 1); i < end; i += kWave / 2) {
      const Corner k = P.crn[i];
      seed_ring_fix(P, cur, (threadIdx.x & 1) ? k.v2 : k.v1);        // (a vertex met twice is written twice with the same values)
    }
  }
}

// Serial repair, the last resort of the verification (verify_sweeps): the vertices a sweep flags (fix == 2: listed in the plan's
// first work-list buffer, counted in cnt[3].n_wait) are evaluated again ONE AFTER THE OTHER by one 8-lane group; a vertex whose
// state moves hands its corner neighbours on.  The concurrent fixing sweeps store while their neighbours read: a cluster of tied
// pop times can keep flipping under them exactly like under the band steps (1 of 7 611 random sparse-lethal inflation maps,
// configuration 7515 of tools/gpu_infl_fuzz.py: 4, 4, 1, 1, 9, 3, 7, 5 vertices off after sweeps 1..8); one at a time, every
// evaluation sees its neighbours' stores (what the CPU model's verification does: it settles that map in two sweeps).
__global__ __launch_bounds__(kWave) void k_verify_serial(const Plan* __restrict__ plans, uint32_t max_evals)
{
  const Plan& P = plans[blockIdx.x];
  const int lane = threadIdx.x;
  const Ctl a = P.ctl[0], b = P.ctl[1];
  const Ctl cur = (a.it > b.it) ? a : b;
  if (!cur.done || cur.overflow) return;
  const int sub = lane & (kGroup - 1);
  uint32_t* const L = P.list[0];
  __shared__ uint32_t s_head, s_tail;
  if (lane == 0) { s_head = 0u; s_tail = min(P.cnt[3].n_wait, P.cap); }
  __syncthreads();
  const uint32_t tag = kExactStamp - 3u;
  if (lane == 0) for (uint32_t i = 0; i < s_tail; ++i) P.stamp[L[i]] = tag;
  __threadfence(); __syncthreads();
  for (uint32_t n = 0; n < max_evals; ++n) {
    if (s_head >= s_tail) break;
    const uint32_t v = L[s_head % P.cap];
    __syncthreads();
    bool moved = false;
    if (lane < kGroup) {                                               // (one group: the others idle)
      const Eval e = group_eval_cvp(P, cur, v, sub);
      if (sub == 0) moved = !verify_entry(P, cur, v, e, true);
    }
    moved = __shfl((int)moved, 0) != 0;
    if (lane == 0) {
      ++s_head;
      P.stamp[v] = 0u;                                                 // may be listed again
      if (moved)
        for (uint32_t i = P.crn_ptr[v]; i < P.crn_ptr[v + 1]; ++i) {
          const Corner k = P.crn[i];
          if (k.v1 == kNone) continue;
          for (int q = 0; q < 2; ++q) {
            const uint32_t w = q ? k.v2 : k.v1;
            if (is_seed(P, w) || P.blocked[w] || P.stamp[w] == tag || s_tail - s_head >= P.cap) continue;
            P.stamp[w] = tag; L[s_tail % P.cap] = w; ++s_tail;
          }
        }
    }
    __threadfence(); __syncthreads();
  }
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
    if (sub == 0 && !verify_entry(P, cur, v, e, fix == 1)) {          // spec: mnav_eval.h (fix 1: stores the re-evaluated state)
      ++bad;
      if (fix == 2) { const uint32_t at = atomicAdd(&P.cnt[3].n_wait, 1u); if (at < P.cap) P.list[0][at] = v; }   // listed for k_verify_serial
    }
  }
  bad = wave_sum(bad);
  if (lane == 0 && bad) { atomicAdd(&P.cnt[3].changed, bad); atomicOr(any_bad, 1u); }
}
