// mnav_plan_kernels.h -- per-plan kernels shared by the engines: state initialisation and seeding, the path walks (k_finish,
// k_path_lazy), path packing, settled-vertex counts.  Included by mnav.hip inside its anonymous namespace; not a stand-alone header.
#pragma once

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
  c0.it = -1; c0.n = 0; c0.thr_fixed = -inf_f(); c0.goal_dist = inf_f(); c0.arm_vertex = kNone;
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
