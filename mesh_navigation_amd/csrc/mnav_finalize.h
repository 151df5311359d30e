// mnav_finalize.h -- PlanResult and k_dij_finalize: the reference's exact cut-off semantics (tentative values beyond goal_dist,
// negative offsets through goal_cut) and the predecessors of every vertex in one gather pass over the LDS tiles (the per-plan
// engines: tile rounds, persistent, asynchronous, sharded; the tile-batch engine has its own pass, mnav_tb_finalize.h).
// Included by mnav.hip inside its anonymous namespace; not a stand-alone header.
#pragma once

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

// One workgroup per (plan, chunk of tiles).
__global__ __launch_bounds__(kTileBlock) void k_dij_finalize(const Plan* __restrict__ plans, const TilePlan* __restrict__ tplans,
                                                             uint32_t* __restrict__ mismatch, PlanResult* __restrict__ res,
                                                             uint32_t tiles_per_block, uint32_t n_plans)
{
  // grid: x = plan, y = chunk of tiles.  Only tiles that were activated or woken are looked at: any
  // vertex that owes a value to an expanded source sits in such a tile (its source pushed to it
  // through a halo copy, which wakes the owner); everything else keeps dist = inf / pred = itself.
  // Per tile the push graph is staged in LDS like in the solve and read backwards: every edge
  // x -> y with an expanded source offers (d[x] + w, d[x], x) to its owned target y; pass 1 takes the
  // smallest sum (ds_min on the float bits), pass 2 the smallest (d[x], x) among the edges that attain
  // it (64-bit ds_min) -- the reference's predecessor under the (value, id) pop order.
  constexpr int PG = 1;                                              // (plans per workgroup; the per-plan arrays below are what is left of a grouped variant)
  const uint32_t p0 = blockIdx.x * PG;
  const TilePlan& T = tplans[p0];
  const int tid = threadIdx.x;
  MNAV_GLOBAL const uint32_t* g_vptr = as_global(T.vptr);
  MNAV_GLOBAL const uint32_t* g_hptr = as_global(T.hptr);
  MNAV_GLOBAL const uint32_t* g_eptr = as_global(T.eptr);
  MNAV_GLOBAL const uint32_t* g_rptr = as_global(T.rptr);
  MNAV_GLOBAL const uint32_t* g_verts = as_global(T.verts);
  MNAV_GLOBAL const uint32_t* g_halo_verts = as_global(T.halo_verts);
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
      dt = P.dist[tg];
      s_armed[q] = dt < inf_f() ? 1u : 0u;
      const GoalCut gc = goal_cut(dt, P.offset, P.goal_tie1 ? P.goal_tie1 - 1u : tg);   // dijkstra :296
      s_goal[q] = gc.goal; s_cut[q] = gc.cut; s_tie[q] = gc.tie;
    }
  }
  uint32_t bad = 0;
  __syncthreads();
  const uint32_t t_beg = T.t_lo + blockIdx.y * tiles_per_block;
  const uint32_t t_end = min(t_beg + tiles_per_block, T.t_hi ? T.t_hi : T.ntiles);
  for (uint32_t t = t_beg; t < t_end; ++t) {
    {                                                                 // uniform over the workgroup
      MNAV_GLOBAL const float* g_tlast = as_global((const float*)T.tlast);
      MNAV_GLOBAL const uint32_t* g_p0 = as_global((const uint32_t*)T.pend[0]);
      MNAV_GLOBAL const uint32_t* g_p1 = as_global((const uint32_t*)T.pend[1]);
      // (after the asynchronous engine pend[1] is the tiles' state word, 0 when the plan has finished: every tile that was ever woken
      //  was solved or dropped there, and both mark tlast)
      if (!(g_tlast[t] > -inf_f()) && g_p0[t] == kInfBits && (T.pend1_is_state || g_p1[t] == kInfBits)) continue;
    }
    const uint32_t v0 = g_vptr[t], nv = g_vptr[t + 1] - v0;
    const uint32_t h0 = g_hptr[t], nh = g_hptr[t + 1] - h0;
    const uint32_t e0 = g_eptr[t], ne = g_eptr[t + 1] - e0;
    const uint32_t r0 = g_rptr[t], nl = nv + nh;
    __syncthreads();                                               // the previous tile's LDS image is dead
    uint32_t g[kFinVpt];
#pragma unroll
    for (int k = 0; k < kFinVpt; ++k) {
      const uint32_t i = tid + k * kTileBlock;
      g[k] = (i < nv) ? g_verts[v0 + i] : (i < nl ? g_halo_verts[h0 + i - nv] : 0u);
      if (i < nl) lgid[i] = g[k];
    }
    for (uint32_t i = tid + kFinVpt * kTileBlock; i < nl; i += kTileBlock) lgid[i] = (i < nv) ? g_verts[v0 + i] : g_halo_verts[h0 + i - nv];
    stage_tile_graph(T, L, e0, ne, r0, nl, tid);                   // (ends with a barrier: lgid is visible too)
    uint32_t dbn[kFinVpt];                                           // values of the first plan of the group
#pragma unroll
    for (int k = 0; k < kFinVpt; ++k) {
      dbn[k] = f2u(as_global(plans[p0].dist)[g[k]]);
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
      auto value_of = [&](uint32_t gid) -> uint32_t {
        return f2u(g_dist[gid]);
      };
      if (q) __syncthreads();                                        // the previous plan's ldu / lsum / lkey are dead
      uint32_t db[kFinVpt];
#pragma unroll
      for (int k = 0; k < kFinVpt; ++k) db[k] = dbn[k];
      {
        // no reached vertex among the tile's own and halo vertices: nothing to derive here (dist = inf, pred = itself)
        int reached = 0;
#pragma unroll
        for (int k = 0; k < kFinVpt; ++k) reached |= ((uint32_t)(tid + k * kTileBlock) < nl && db[k] != kInfBits) ? 1 : 0;
        for (uint32_t i = tid + kFinVpt * kTileBlock; i < nl; i += kTileBlock) reached |= (value_of(lgid[i]) != kInfBits) ? 1 : 0;
        if (!__syncthreads_or(reached)) {                             // uniform over the workgroup
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
          continue;
        }
        const uint32_t sb = lsum[i], ob = L.ldu[i];
        const unsigned long long key = lkey[i];
        if (sb != kInfBits && key == ~0ull && (!T.owned || T.owned[gg])) ++bad;   // a finite value no expanded neighbour supports (a halo copy's support may live on another process)
        const uint32_t pv = (sb != kInfBits) ? (uint32_t)key : gg;
        g_pred[gg] = pv;
        if (sb != ob) g_dist[gg] = u2f(sb);                // (else only above goal_dist: tentative value, dijkstra :337-343)
        if (sb != kInfBits) ++cnt;
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
      {
        const TilePlan& Tq = tplans[p0 + q];
        const TCtl a = Tq.ctl[0], b = Tq.ctl[1];
        const TCtl last = (a.it > b.it) ? a : b;
        r.it = last.it; r.done = last.done; r.bands = last.sweeps; r.evals = last.acts; r.overflow = last.pad[0];
      }
      P.ctl[0] = r; P.ctl[1] = r;
    }
  }
}
