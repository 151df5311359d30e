// mnav_cvp_wide.h -- CVP wavefront (cvp_mesh_planner.cpp:651-918, loop :747-886) for BATCHES of plans: the wide step kernel.
// Included by mnav.hip inside its anonymous namespace, after StepCtx / push_agg / park_agg / step_body / k_step are defined.
// The per-vertex rules are mnav_eval.h's (make_cvp_item_pre, eval_cvp_items, eval_cvp): this file only spreads them over lanes.
#pragma once

// ---------------------------------------------------------------------------------------------
// Wide CVP step (batches): a wave takes 32 (first version: 64) work-list entries per round instead of 8.
// The 8-lane replay (mnav_band.h: group_eval_cvp) spends most of its instructions on in-group shuffles and serves 8 vertices per wave instruction.
// Here the evaluation is cut where its data dependence allows (mnav_eval.h: make_cvp_item / eval_cvp_items):
//   phase A, one lane per incident FACE of the round's vertices (~192 faces = 3-4 passes of 64 lanes): fire event and float64
//            candidate, neither depends on the vertex's own state -> a 48-byte item in LDS;
//   phase B, one lane per VERTEX: the replay over its own items, serially, without a single shuffle;
//   pushes,  one lane per face again (wave-aggregated list appends as before).
// Same functions, same decisions as eval_cvp (held against it in the CPU model on every evaluation); which vertices are
// evaluated concurrently differs, which the fixed-point iteration does not care about.
// ---------------------------------------------------------------------------------------------
#ifndef MNAV_WIDE_VERTS
#define MNAV_WIDE_VERTS 32
#endif
constexpr uint32_t kWideVerts = MNAV_WIDE_VERTS;   // work-list entries per wave and round (64, or 32: half the LDS image and shorter rounds for more resident waves)
constexpr uint32_t kWideSlots = (kWideVerts * 13u) / 2u;   // items per wave and round: 6.5 faces per vertex (32 vertices: 10 KB of LDS + the table below = 12 KB; 64 vertices: 22 KB)
constexpr int kWideOcc = kWideVerts == 64u ? 2 : 3;  // waves per SIMD the register allocator must reach
constexpr uint32_t kWideSeen = 512;           // direct-mapped table of vertices this wave has pushed in this launch (see push_many)
constexpr uint32_t kWideMaxFaces = 32;        // faces of one vertex that go through the items; beyond: the serial rule (eval_cvp)
// items field-major: phase A stores a field of 64 consecutive slots at a time, phase B lanes read only the fields they look at
// (an array of 48-byte structs costs an 8-way bank conflict per read there: the lanes' items lie 6 x 48 bytes apart)
struct WideLds {
  unsigned long long hi[kWideSlots], own[kWideSlots];
  double u3tmp[kWideSlots], cand[kWideSlots];
  uint32_t up[kWideSlots], lvl[kWideSlots], meta[kWideSlots];
  float dir[kWideSlots];
  uint8_t owner[kWideSlots];
  uint32_t seen[kWideSeen];
};
struct WideItems {
  const WideLds* L; uint32_t off;
  __device__ __forceinline__ unsigned long long hi(uint32_t k) const { return L->hi[off + k]; }
  __device__ __forceinline__ uint32_t up(uint32_t k) const { return L->up[off + k]; }
  __device__ __forceinline__ uint32_t lvl(uint32_t k) const { return L->lvl[off + k]; }
  __device__ __forceinline__ unsigned long long own(uint32_t k) const { return L->own[off + k]; }
  __device__ __forceinline__ double u3tmp(uint32_t k) const { return L->u3tmp[off + k]; }
  __device__ __forceinline__ double cand(uint32_t k) const { return L->cand[off + k]; }
  __device__ __forceinline__ float dir(uint32_t k) const { return L->dir[off + k]; }
  __device__ __forceinline__ uint32_t meta(uint32_t k) const { return L->meta[off + k]; }
};

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t x)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x = max(x, (uint32_t)__shfl_xor((int)x, o));
  return x;
}

// dedup'd, wave-aggregated append of up to N vertices per lane (kNone: none) to the next work list: the stamp looks, the
// exchanges and the ONE counter atomic of the whole batch are each in flight together (push_agg per vertex is a chain of three
// dependent round trips)
// Neighbouring vertices share most of their neighbours, and a wave's 64 work-list entries are neighbours: most candidates of a
// batch are duplicates of each other.  They are filtered in LDS first -- `seen` is a direct-mapped table of the vertices this
// wave has handed on during this launch (one step of one plan: the global stamp of such a vertex is set already, so dropping a
// repeat is exactly what the stamp would do; a slot taken over by another vertex only lets a repeat through) -- which leaves
// a third of the global look / exchange pairs.
template <int N>
__device__ __forceinline__ void push_many(StepCtx& S, uint32_t (&u)[N], uint32_t* seen, int lane)
{
  uint32_t st[N];
  bool ok[N];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    if (u[k] != kNone) { const uint32_t old = atomicExch(&seen[u[k] & (kWideSeen - 1u)], u[k]); if (old == u[k]) u[k] = kNone; }
  }
#pragma unroll
  for (int k = 0; k < N; ++k) { st[k] = S.sv; if (u[k] != kNone) { S.P->dirty[u[k]] = S.sv; st[k] = S.P->stamp[u[k]]; } }
#pragma unroll
  for (int k = 0; k < N; ++k) { uint32_t o = S.sv; if (st[k] != S.sv) o = atomicExch(&S.P->stamp[u[k]], S.sv); st[k] = o; }
  uint32_t mine = 0;
#pragma unroll
  for (int k = 0; k < N; ++k) { ok[k] = st[k] != S.sv; mine += ok[k] ? 1u : 0u; }
  uint32_t incl = mine;
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) { const uint32_t x = (uint32_t)__shfl_up((int)incl, o); if (lane >= o) incl += x; }
  const uint32_t total = (uint32_t)__shfl((int)incl, kWave - 1);
  if (total == 0u) return;
  uint32_t base = 0;
  if (lane == 0) base = atomicAdd(&S.cnt->n_next, total);
  uint32_t idx = (uint32_t)__shfl((int)base, 0) + incl - mine;
#pragma unroll
  for (int k = 0; k < N; ++k) if (ok[k]) { if (idx < S.P->cap) S.next[idx] = u[k]; ++idx; }
}

constexpr int kWidePassesPerBatch = kWideVerts == 64u ? 4 : 2;   // face passes whose loads are in flight together

#ifdef MNAV_WIDE_TIMING                   // debugging aid: cycles per phase of wide_round, summed over all waves
__device__ unsigned long long g_wide_timing[12];   // [0..6] cycles per phase, [8] rounds, [9] active entries, [10] evaluated, [11] serial-rule vertices
#define WD_STAMP(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); wt[k] += now_ - w_last; w_last = now_; } while (0)
#else
#define WD_STAMP(k) do { } while (0)
#endif

// mode 0: an ordinary step (spec: process_entry); 1: the repair sweep after goal_dist was armed (process_repair: every reached
// vertex is looked at, those whose pop time lies above goal_dist are evaluated again under the final cut-off and stored, nothing is
// pushed); 2: the rebuild after a band shrink (process_entry over every reached vertex, band_new == 1)
template <int MODE>
__device__ __forceinline__ void wide_round(StepCtx& S, const Plan& P, const Ctl& c, WideLds& L, bool active, uint32_t v, int lane)
{
#ifdef MNAV_WIDE_TIMING
  unsigned long long wt[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, w_last = __builtin_readcyclecounter();
#endif
  // ---- per vertex: does it have to be evaluated?  (spec: process_entry)
  bool evaluate = false, retain = false, push_nb = false, self_again = false;
  float old_d = inf_f(), old_t = inf_f(), t_new = inf_f(), old_dir = 0.0f;
  uint32_t old_pred = kNone, old_cut = kNone, beg = 0, nf = 0;
  PopKey old_key = key_inf();
  if (active && !is_seed(P, v)) {
    old_d = P.dist[v]; old_key = P.tkey[v];
    const uint8_t blk = P.blocked[v];
    const uint32_t dirty = P.dirty[v];
    const uint32_t b0 = P.crn_ptr[v], b1 = P.crn_ptr[v + 1];
    old_pred = P.pred[v]; old_cut = P.cutf[v]; old_dir = P.dirn[v];  // (all of the vertex's state in flight together)
    old_t = key_time(old_key);
    if constexpr (MODE == 1) {
      if (old_d < inf_f()) {
        if (old_t > c.goal_dist) { evaluate = true; beg = b0; nf = b1 - b0; }
        else { t_new = old_t; retain = (t_new >= c.thr) && (t_new < inf_f()); }
      }
    } else {
      const bool go = !(old_t < c.thr_fixed) && !blk && (MODE == 0 || old_d < inf_f());
      const bool parked = go && !c.band_new && !(old_t < c.thr) && old_t < inf_f() && dirty != (uint32_t)c.it;
      if (parked) { retain = true; t_new = old_t; }
      else if (go) { evaluate = true; beg = b0; nf = b1 - b0; }
    }
  }
  // ---- item slots: prefix sum of the face counts.  A vertex of very high valence, and whatever does not fit, takes the serial rule
  const uint32_t want = (nf <= kWideMaxFaces) ? nf : 0u;
  uint32_t incl = want;
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) { const uint32_t x = (uint32_t)__shfl_up((int)incl, o); if (lane >= o) incl += x; }
  const bool slots = want > 0u && incl <= kWideSlots;
  const uint32_t off = incl - want;
  const uint32_t T = wave_max_u32(slots ? incl : 0u);
  __syncthreads();                                                     // the previous round's readers of the LDS image are done
  if (slots) for (uint32_t k = 0; k < nf; ++k) L.owner[off + k] = (uint8_t)lane;
  __syncthreads();
  WD_STAMP(0);
  // ---- phase A: one lane per face; the records of a batch of passes are loaded before the first face is looked at
  for (uint32_t q00 = 0; q00 < T; q00 += kWidePassesPerBatch * kWave) {
    Corner ck[kWidePassesPerBatch];
    uint32_t vs[kWidePassesPerBatch];
#pragma unroll
    for (int p = 0; p < kWidePassesPerBatch; ++p) {
      const uint32_t q = q00 + p * kWave + (uint32_t)lane;
      ck[p].v1 = kNone; ck[p].v2 = kNone; ck[p].a = 0.f; ck[p].b = 0.f; ck[p].c = 0.f; ck[p].face = 0u; vs[p] = 0u;
      if (q00 + p * kWave < T) {                                       // (wave-uniform)
        const int s = L.owner[q < T ? q : T - 1u];
        vs[p] = (uint32_t)__shfl((int)v, s);
        const uint32_t bs = (uint32_t)__shfl((int)beg, s), os = (uint32_t)__shfl((int)off, s);
        if (q < T) ck[p] = P.crn[bs + (q - os)];
      }
    }
    PopKey t1[kWidePassesPerBatch], t2[kWidePassesPerBatch];
    float d1[kWidePassesPerBatch], d2[kWidePassesPerBatch];
#pragma unroll
    for (int p = 0; p < kWidePassesPerBatch; ++p) {
      t1[p] = key_inf(); t2[p] = key_inf(); d1[p] = inf_f(); d2[p] = inf_f();
      if (ck[p].v1 != kNone) { t1[p] = P.tkey[ck[p].v1]; t2[p] = P.tkey[ck[p].v2]; d1[p] = P.dist[ck[p].v1]; d2[p] = P.dist[ck[p].v2]; }
    }
#ifdef MNAV_WIDE_TIMING
    { float z = 0.f; for (int p = 0; p < kWidePassesPerBatch; ++p) z += d1[p] + d2[p]; asm volatile("" :: "v"(z)); }   // wait for the loads
#endif
    WD_STAMP(1);
#pragma unroll
    for (int p = 0; p < kWidePassesPerBatch; ++p) {
      const uint32_t q = q00 + p * kWave + (uint32_t)lane;
      if (q < T) {
        const CvpItem it = make_cvp_item_pre(P, c, vs[p], ck[p], t1[p], t2[p], d1[p], d2[p]);
        L.hi[q] = it.hi; L.own[q] = it.own; L.u3tmp[q] = it.u3tmp; L.cand[q] = it.cand;
        L.up[q] = it.up; L.lvl[q] = it.lvl; L.meta[q] = it.meta; L.dir[q] = it.dir;
      }
    }
    WD_STAMP(2);
  }
  __syncthreads();
  // ---- phase B: one lane per vertex
  Eval e; e.d = inf_f(); e.t = inf_f(); e.key = key_inf(); e.pred = v; e.dir = 0.0f; e.cut = kNone; e.keyd = inf_f();
  if (slots) {
    uint32_t win; int sel;
    WideItems mine; mine.L = &L; mine.off = off;
    e = eval_cvp_items_any(P, v, nf, mine, win, sel);
    if (win != kNone) { const Corner k = P.crn[beg + win]; e.pred = (sel == 1) ? k.v1 : k.v2; e.cut = corner_face(k); }
  } else if (evaluate) {
    e = eval_cvp(P, c, v);                                             // no faces / too many / no room left in this round
  }
  WD_STAMP(3);
  if (evaluate) {
    ++S.levals;
    const bool changed = (f2u(e.d) != f2u(old_d)) || (f2u(e.t) != f2u(old_t)) || (e.pred != old_pred) || (e.key != old_key) ||
                         (e.cut != old_cut) || (f2u(e.dir) != f2u(old_dir));
    if (changed) { P.dist[v] = e.d; P.pred[v] = e.pred; P.tkey[v] = e.key; P.dirn[v] = e.dir; P.cutf[v] = e.cut; }
    t_new = e.t;
    if constexpr (MODE == 1) {
      if (f2u(e.d) != f2u(old_d) || e.key != old_key) S.lchanged = true;       // sweep again (spec: process_repair)
      retain = (t_new >= c.thr) && (t_new < inf_f());
    } else {
      const bool was_in = old_t < c.thr, now_in = e.t < c.thr;
      push_nb = (changed && (was_in || now_in)) || (now_in && c.band_new);
      retain = !now_in && e.t < inf_f();
      self_again = (e.key.lvl > 0u || old_key.lvl > 0u) && e.key != old_key;   // spec: process_entry
    }
  }
  if (push_nb || self_again) {
    S.lchanged = true;
    if (push_nb && ((old_t < c.thr) != (t_new < c.thr))) S.lcut = fminf(S.lcut, fminf(old_t, t_new));   // crossed the bound (spec: note_cut)
  }
  // ---- pushes: one lane per face of the vertices that moved, the vertex itself when its cascade key moved
  {
    constexpr int kPasses = (int)((kWideSlots + kWave - 1) / kWave);
    uint32_t u[2 * kPasses + 1];
#pragma unroll
    for (int p = 0; p < kPasses; ++p) {
      u[2 * p] = kNone; u[2 * p + 1] = kNone;
      if ((uint32_t)(p * kWave) < T) {                                  // (wave-uniform)
        const uint32_t q = p * kWave + (uint32_t)lane;
        const int s = L.owner[q < T ? q : T - 1u];
        const bool w = __shfl((int)push_nb, s) != 0 && q < T;
        const uint32_t bs = (uint32_t)__shfl((int)beg, s), os = (uint32_t)__shfl((int)off, s);
        if (w) { const Corner k = P.crn[bs + (q - os)]; if (k.v1 != kNone) { u[2 * p] = k.v1; u[2 * p + 1] = k.v2; } }
      }
    }
    u[2 * kPasses] = self_again ? v : kNone;
    WD_STAMP(4);
    push_many(S, u, L.seen, lane);
  }
  WD_STAMP(5);
  unsigned long long sm = __ballot(push_nb && !slots);                 // vertices that took the serial rule: the wave walks their faces
  while (sm) {
    const int src = __ffsll((long long)sm) - 1;
    sm &= sm - 1ull;
    const uint32_t vb = (uint32_t)__shfl((int)beg, src), vn = (uint32_t)__shfl((int)nf, src);
    for (uint32_t i0 = 0; i0 < vn; i0 += kWave) {
      const uint32_t i = i0 + (uint32_t)lane;
      uint32_t a = kNone, b = kNone;
      if (i < vn) { const Corner k = P.crn[vb + i]; if (k.v1 != kNone) { a = k.v1; b = k.v2; } }
      push_agg<true>(S, a != kNone, a);
      push_agg<true>(S, b != kNone, b);
    }
  }
  park_agg(S, retain, v);
  if (retain) S.lmin = fminf(S.lmin, t_new);
  WD_STAMP(6);
#ifdef MNAV_WIDE_TIMING
  {
    const unsigned long long ne = __popcll(__ballot(evaluate)), na = __popcll(__ballot(active)), ns = __popcll(__ballot(!slots && evaluate));
    if (lane == 0) { atomicAdd(&g_wide_timing[8], 1ull); atomicAdd(&g_wide_timing[9], na); atomicAdd(&g_wide_timing[10], ne); atomicAdd(&g_wide_timing[11], ns); }
  }
  if (lane == 0) for (int k = 0; k < 8; ++k) if (wt[k]) atomicAdd(&g_wide_timing[k], wt[k]);
#endif
}

// ---- CVP batches: one launch of persistent waves per step, the work of ALL plans dealt out in chunks of kWideVerts entries ------------
// With a grid of (waves per plan, plans) most workgroups of a step find nothing to do -- a plan's work list is a few hundred to a
// few ten thousand entries, the grid must cover the largest -- and for a kernel with a 22 KB LDS image every one of them holds
// an LDS slot while it starts and exits: on the benched C3 configuration that kept the wide kernel at the 8-lane kernel's
// throughput.  k_cvp_ctl evaluates every plan's controller once (what each workgroup of k_step does for itself), writes the
// control blocks and the prefix sums of the plans' chunk counts; k_step_wide then runs exactly as many waves as stay resident,
// each taking an equal, contiguous share of the step's chunks, whatever plans they belong to.
struct WideSched { uint32_t total, n_repair, pad[2]; };
constexpr uint32_t kWideGroupsMax = 8;        // groups of plans a CVP batch is stepped in, each on its own stream (branch of the captured graph)
constexpr uint32_t kRepairRows = 16;          // grid rows of k_step_repair: plans in a repair step are rare, a row takes several if there are more

// the plans that k_cvp_ctl found in a repair / rebuild / cut step (rep_list = prefix + n + 1 ...): the 8-lane sweeps over all vertices
__global__ MNAV_STEP_BOUNDS void k_step_repair(const Plan* __restrict__ plans, int j, const uint32_t* __restrict__ rep_list, const WideSched* __restrict__ sched)
{
  const uint32_t nr = sched->n_repair;
  for (uint32_t r = blockIdx.y; r < nr; r += kRepairRows) {
    step_body<kPlannerCvp, true>(plans, j, rep_list[r]);
    __syncthreads();                                                   // (s_ctl of the next plan)
  }
}

__global__ __launch_bounds__(256) void k_cvp_ctl(const Plan* __restrict__ plans, uint32_t n, int j, uint32_t* __restrict__ prefix, WideSched* __restrict__ sched)
{
  __shared__ uint32_t s_base, s_rep;
  __shared__ uint32_t s_wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) { s_base = 0u; s_rep = 0u; }
  __syncthreads();
  for (uint32_t p0 = 0; p0 < n; p0 += 256) {
    const uint32_t p = p0 + tid;
    uint32_t chunks = 0;
    if (p < n) {
      const Plan& P = plans[p];
      const Ctl prev = P.ctl[(j + 1) & 1];
      const Cnt cprev = P.cnt[(j + 2) % 3];
      const Ctl cur = controller(P, prev, cprev);
      P.ctl[j & 1] = cur;
      Cnt z; z.n_next = 0; z.changed = 0; z.minkey = 0x7f800000u; z.evals = 0; z.n_wait = 0; z.minchg = 0x7f800000u; z.pad[0] = z.pad[1] = 0;
      P.cnt[(j + 1) % 3] = z;
      if (!cur.done) {
        if (P.seed_mask == nullptr && cur.repair == 0) chunks = (cur.n + cur.wread + kWideVerts - 1) / kWideVerts;
        else if (P.seed_mask == nullptr && cur.repair <= 2) chunks = (P.V + kWideVerts - 1) / kWideVerts;   // repair sweep / rebuild: over all vertices
        else prefix[n + 1u + atomicAdd(&s_rep, 1u)] = p;              // band cut (no evaluation): k_step_repair; its list follows the prefix sums
      }
    }
    uint32_t incl = chunks;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t x = (uint32_t)__shfl_up((int)incl, o); if (lane >= o) incl += x; }
    if (lane == 63) s_wsum[wid] = incl;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
    for (int w = 0; w < 4; ++w) { if (w < wid) woff += s_wsum[w]; tot += s_wsum[w]; }
    if (p < n) prefix[p] = s_base + woff + incl - chunks;
    __syncthreads();
    if (tid == 0) s_base += tot;
    __syncthreads();
  }
  if (tid == 0) { prefix[n] = s_base; sched->total = s_base; sched->n_repair = s_rep; }
}

// the plan that chunk c belongs to: the largest p with prefix[p] <= c (all lanes search together, 64 entries per look)
__device__ __forceinline__ uint32_t wide_find_plan(const uint32_t* __restrict__ prefix, uint32_t n, uint32_t c, int lane)
{
  uint32_t lo = 0, len = n + 1u;                                       // the answer lies in [lo, lo + len)
  while (len > 1u) {
    const uint32_t step = (len + kWave - 1u) / kWave;
    const uint32_t idx = lo + (uint32_t)lane * step;
    const bool le = idx < lo + len && prefix[idx] <= c;
    const uint32_t k = (uint32_t)__popcll(__ballot(le));                // samples are ascending: the first k of them are <= c (k >= 1)
    const uint32_t nlo = lo + (k - 1u) * step;
    len = min(step, lo + len - nlo);
    lo = nlo;
  }
  return lo;
}

// 32 entries per round: 12 KB of LDS and <= 168 VGPRs, 12 resident waves per CU (64 entries: 22 KB, 213 VGPRs, 7 waves)
__global__ __launch_bounds__(kWave, kWideOcc) void k_step_wide(const Plan* __restrict__ plans, uint32_t n, int j, const uint32_t* __restrict__ prefix,
                                                        const WideSched* __restrict__ sched)
{
  __shared__ WideLds s_wide;
  const int lane = threadIdx.x;
  const uint32_t total = sched->total;
  uint32_t c0 = (uint32_t)(((unsigned long long)total * blockIdx.x) / gridDim.x);
  const uint32_t c1 = (uint32_t)(((unsigned long long)total * (blockIdx.x + 1u)) / gridDim.x);
  if (c0 >= c1) return;
  uint32_t p = wide_find_plan(prefix, n, c0, lane);
  while (c0 < c1) {
    const uint32_t pb = prefix[p], pe = prefix[p + 1];
    if (pe <= c0) { ++p; continue; }                                   // (a plan without chunks in this step)
    const Plan& P = plans[p];
    const Ctl cur = P.ctl[j & 1];
    Cnt* cnt = &P.cnt[j % 3];
    StepCtx S{ &P, cnt, P.list[(cur.it + 1) & 1], (uint32_t)cur.it + 1u, inf_f(), 0u, false, P.wlist[cur.wsel & 1u], cur.wbase, cur.epoch, inf_f() };
    __syncthreads();
    for (uint32_t k = (uint32_t)lane; k < kWideSeen; k += kWave) s_wide.seen[k] = kNone;   // vertex ids of another plan (wide_round starts with a barrier)
    const uint32_t* list = P.list[cur.it & 1];
    const uint32_t* wprev = P.wlist[(cur.wsel ^ 1u) & 1u];
    const uint32_t ntot = cur.n + cur.wread;
    const uint32_t ce = min(c1, pe);
    if (cur.repair == 0) {
      for (uint32_t c = c0; c < ce; ++c) {
        const uint32_t i = (c - pb) * kWideVerts + (uint32_t)lane;
        const bool active = (uint32_t)lane < kWideVerts && i < ntot;
        const uint32_t v = active ? (i < cur.n ? list[i] : wprev[i - cur.n]) : 0u;
        wide_round<0>(S, P, cur, s_wide, active, v, lane);
      }
    } else {
      for (uint32_t c = c0; c < ce; ++c) {                             // a sweep over the vertices themselves
        const uint32_t v = (c - pb) * kWideVerts + (uint32_t)lane;
        const bool act = (uint32_t)lane < kWideVerts && v < P.V;
        if (cur.repair == 1) wide_round<1>(S, P, cur, s_wide, act, act ? v : 0u, lane);
        else wide_round<2>(S, P, cur, s_wide, act, act ? v : 0u, lane);
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
    c0 = ce; ++p;
  }
}

