// mnav_tb_finalize.h -- the V-sized outputs of a tile-batch batch: potential with the reference's exact cut-off semantics
// (dijkstra :293-300: tentative values beyond goal_dist come from expanded sources only), predecessors (:331-343 under the
// (value, id) pop order) and the vector map (computeVectorMap :189-209), straight from the engine's blocked distances.
// Included by mnav.hip after mnav_tb.h; not a stand-alone header.
//
// Round 3/4 ran k_dij_finalize<8, true> for this: the 512-vertex LDS tiles of the per-plan engines, distances gathered through
// vaddr[], the tile's PUSH graph read backwards with ds_min / 64-bit ds_min and six workgroup barriers per (tile, plan) -- 91 ms
// of the 323 ms headline step, LDS 41 % busy with half of it bank conflicts (profiles/r05_c2_before_sq.md).  This pass works on
// the tile-batch engine's own tiles instead: one WAVE per (tile, range of plans); a (tile, plan) slice is one contiguous run of
// <= 200 floats that already holds the ghosts -- one coalesced load, no gather --; a lane owns two of the tile's vertices and
// keeps their <= 8 SOURCES (slice index, weight, vertex id: the pull form of the tile's graph, mnav_tb_build.h) in registers
// for all its plans, so a vertex is 8 LDS reads, 8 float adds and a (sum, value, id) argmin in registers: no atomics, no
// barrier (a wave is its own workgroup).  Adjacent tiles of the same plans are handled by neighbouring workgroups of the SAME
// XCD (blockIdx -> (xcd, tile, plan range)), so that the short runs of a tile's rows meet in one L2 before they are written.
#pragma once

namespace {

typedef float f32x3_u __attribute__((ext_vector_type(3), aligned(4)));   // one 12-byte store per vector-map entry (global_store_dwordx3)

// What k_tb_finalize needs of a plan besides its slices, one 48-byte record (k_tb_fin_plans writes it): a wave loads the record of
// its NEXT plan with ONE vector load -- lane l takes dword l -- together with that plan's slice, and reads the dwords out of the
// register with v_readlane when the plan's turn comes.
struct FinRec { float* dist; uint32_t* pred; float* vm; uint32_t seed; float cut; uint32_t tie; uint32_t pad_[3]; };
static_assert(sizeof(FinRec) == 48, "FinRec: 12 dwords");
constexpr uint32_t kFinRecWords = 12;

struct FinTb {
  const uint16_t* src; const float* w; const TbFinOvf* ovf; const float* ovf_w;   // finalize tables (mnav_tb_build.h), weights materialised per cost limit
  const uint32_t* verts; const uint32_t* ghost_gid; const uint32_t* order;   // order: the tiles by their smallest vertex id
  const float* xyz; float* const* vecmaps;        // vecmaps != null: the vector map in the same pass
  const Plan* plans; PlanResult* res; uint32_t* mismatch; const FinRec* recs;
  uint32_t plans_per_wave, tiles_per_xcd, max_sl;
};

// per plan: the plan record the path walk reads (k_finish) -- what k_dij_finalize's first tile chunk used to write
__global__ __launch_bounds__(kBlock) void k_tb_fin_plans(tb::Args A, const Plan* __restrict__ plans, float* const* __restrict__ vecmaps, FinRec* __restrict__ recs)
{
  const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= A.NP) return;
  const Plan& P = plans[p];
  const float dt = A.D[tb::slot_addr(A.vaddr[A.target[p]], A.NP, p)];
  const GoalCut gc = goal_cut(dt, A.offset, A.target[p]);              // dijkstra :296
  Ctl r; memset(&r, 0, sizeof(r));
  r.armed = dt < inf_f() ? 1u : 0u; r.goal_dist = gc.goal; r.thr = inf_f(); r.thr_fixed = inf_f();
  r.it = (int32_t)A.ctl->iters; r.done = 1u; r.overflow = (A.ctl->err || A.ctl->n_cand[0]) ? 1u : 0u;
  P.ctl[0] = r; P.ctl[1] = r;
  FinRec fr; memset(&fr, 0, sizeof(fr));
  fr.dist = P.dist; fr.pred = P.pred; fr.vm = vecmaps ? vecmaps[p] : nullptr; fr.seed = A.seed[p]; fr.cut = gc.cut; fr.tie = gc.tie;
  recs[p] = fr;
}

__global__ __launch_bounds__(kBlock) void k_tb_fin_weights(size_t n, const uint32_t* __restrict__ wsrc, const Nbr* __restrict__ nbr, float* __restrict__ w)
{
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) w[i] = (wsrc[i] == kNone) ? inf_f() : nbr[wsrc[i]].w;
}

// kFinWaves waves per workgroup: one tile each -- consecutive tiles of the bisection order, a compact patch -- walking the SAME
// plans in step (a barrier per plan).  A tile's rows are runs of ~11 vertices = a third of a 128-byte line of a vertex-order
// output array; written by one wave alone the lines leave the L2 partially filled long before the neighbouring tile gets to
// the same plan (first version of this kernel: no faster than the pass it replaced).  In step, the patch's lines are completed
// within a microsecond.  Measured (ms of the pass per 7168-plan batch on the 1M mesh, waves per workgroup / waves per SIMD the
// register budget allows): 1 / 3 -> 96, 4 / 3 -> 90, 8 / 3 -> 125, 8 / 4 (31 spilled dwords) -> 149, 16 / 4 -> 151: in step costs more
// than the merged lines save beyond four tiles -- the pass is bound by its instruction count, not by the writes.
#ifndef MNAV_FIN_WAVES
#define MNAV_FIN_WAVES 4
#endif
#ifndef MNAV_FIN_OCC
#define MNAV_FIN_OCC 3
#endif
constexpr int kFinWaves = MNAV_FIN_WAVES;
#ifndef MNAV_FIN_NOTAB
// The vector-map entry of a vertex is the unit vector towards its predecessor (computeVectorMap :204-206): one of its <= 8 sources,
// whichever plan asks.  A wave computes the unit vectors of its vertices' first kFinTabSlots sources ONCE (the same float32
// subtraction, square root and three divisions, so the same bits) into an LDS table [component][slot][row] and a plan reads its
// three floats by the slot that won the argmin: 6 instructions per (vertex, plan) instead of 75 (a fifth of the pass's
// instructions; the square root and the reciprocals issue at a quarter of the rate on top).  Six slots is every source of a
// regular terrain mesh; a predecessor in slot 6 or 7, from the overflow list or chosen by the exact branch is computed as before.
// (Round 5 kept this table in HBM -- 86 GB of gathers per batch, slower -- and found no room for it in registers; 9 KB of LDS per
// wave still leaves three workgroups per CU.)
constexpr uint32_t kFinTabSlots = 6;
#else
constexpr uint32_t kFinTabSlots = 0;
#endif
constexpr uint32_t kFinTabRows = 128, kFinTabWords = 3 * kFinTabSlots * kFinTabRows, kFinNoSlot = 255;
__host__ __device__ constexpr uint32_t fin_lds_words(uint32_t max_sl, bool vecmap) { return 5u * max_sl + (vecmap ? kFinTabWords : 0u); }
template <int T>
__global__ __launch_bounds__(64 * kFinWaves, MNAV_FIN_OCC) void k_tb_finalize(tb::Args A, FinTb F)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t fin_lds_all[];   // per wave: [max_sl] distances of the plan | [max_sl] vertex ids | [3 * max_sl] positions
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t* const fin_lds = fin_lds_all + (size_t)wave * fin_lds_words(F.max_sl, F.vecmaps != nullptr);
  float* const ltab = reinterpret_cast<float*>(fin_lds + 5u * F.max_sl);   // (vector maps only) unit vectors [3][kFinTabSlots][kFinTabRows]
  // blockIdx -> (xcd, tile group of that xcd, plan range): workgroups are dealt to the XCDs round-robin
  const uint32_t xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
  const uint32_t grp = xcd * F.tiles_per_xcd + j % F.tiles_per_xcd, pg = j / F.tiles_per_xcd;   // (tiles_per_xcd: GROUPS of kFinWaves tiles per xcd)
  const uint32_t t_raw = grp * kFinWaves + wave;
  const bool live = t_raw < A.ntiles;
#ifdef MNAV_FIN_BISECTION_ORDER
  const uint32_t t = live ? t_raw : A.ntiles - 1u;                    // (a wave without a tile shadows the last one and stores nothing)
#else
  const uint32_t t = F.order[live ? t_raw : A.ntiles - 1u];           // (a wave without a tile shadows the last one and stores nothing)
#endif
  const uint32_t p_beg = pg * F.plans_per_wave, p_end = min(p_beg + F.plans_per_wave, A.NP);
  if (grp * kFinWaves >= A.ntiles || p_beg >= p_end) return;          // (uniform over the workgroup)
  const TbTile W = A.tiles[t];
  uint32_t* const ld = fin_lds;
  uint32_t* const lgid = fin_lds + F.max_sl;
  float* const lxyz = reinterpret_cast<float*>(fin_lds + 2 * F.max_sl);
  const uint32_t NP = A.NP;
  // ---- once per wave: ids (and positions) of the slice's vertices, this lane's two vertices and their sources
  for (uint32_t i = lane; i < W.sl; i += 64) {
    uint32_t g = kNone;
    if (i < W.nv) g = F.verts[W.v0 + i];
    else if (i >= (uint32_t)T && i - T < W.nh) g = F.ghost_gid[W.goff + (i - T)];
    lgid[i] = g;
    if (F.vecmaps && g != kNone) { lxyz[3 * i] = F.xyz[3 * (size_t)g]; lxyz[3 * i + 1] = F.xyz[3 * (size_t)g + 1]; lxyz[3 * i + 2] = F.xyz[3 * (size_t)g + 2]; }
  }
  uint32_t ys[2] = { (uint32_t)lane, (uint32_t)lane + 64u };
  bool own[2]; uint32_t gid[2];
  uint32_t src[2][kTbFinSlots], srck[2][kTbFinSlots]; float wt[2][kTbFinSlots];   // srck: the source's slice index | its slot << 16 (what the argmin selects)
#pragma unroll
  for (int v = 0; v < 2; ++v) {
    own[v] = live && ys[v] < W.nv;
#pragma unroll
    for (int k = 0; k < (int)kTbFinSlots; ++k) {
      const size_t at = ((size_t)t * kTbFinSlots + k) * T + (own[v] ? ys[v] : 0u);
      src[v][k] = own[v] ? (uint32_t)F.src[at] : (uint32_t)kTbFinNone;
      wt[v][k] = own[v] ? F.w[at] : inf_f();
    }
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int v = 0; v < 2; ++v) {
    gid[v] = own[v] ? lgid[ys[v]] : kNone;
#ifdef MNAV_FIN_TILEMAJOR_EXPERIMENT                                  // (timing only, wrong results: what would tile-major output arrays cost?)
    if (own[v]) gid[v] = min(t * (uint32_t)T + ys[v], 999999u);
#endif
#pragma unroll
    for (int k = 0; k < (int)kTbFinSlots; ++k)
      if (src[v][k] == kTbFinNone) src[v][k] = ys[v] < (uint32_t)T ? ys[v] : 0u;   // (an unused slot reads the vertex's own row; its weight is +inf: the sum is +inf)
#pragma unroll
    for (int k = 0; k < (int)kTbFinSlots; ++k) srck[v][k] = src[v][k] | ((uint32_t)k << 16);
  }
  float px[2] = { 0.f, 0.f }, py[2] = { 0.f, 0.f }, pz[2] = { 0.f, 0.f };
  if (F.vecmaps) {
#pragma unroll
    for (int v = 0; v < 2; ++v) if (own[v]) { px[v] = lxyz[3 * ys[v]]; py[v] = lxyz[3 * ys[v] + 1]; pz[v] = lxyz[3 * ys[v] + 2]; }
#pragma unroll
    for (int v = 0; v < 2; ++v) if (own[v]) {
#pragma unroll
      for (int k = 0; k < (int)kFinTabSlots; ++k) {                     // (an unused slot: its own row, 0 / 0 -- never selected, its sum is +inf)
        const uint32_t sx = src[v][k];
        float x = lxyz[3 * sx] - px[v], y = lxyz[3 * sx + 1] - py[v], z = lxyz[3 * sx + 2] - pz[v];   // :204
        const float len = sqrtf(x * x + y * y + z * z);                // normalized(), :206
        x = x / len; y = y / len; z = z / len;
        float* const tp = ltab + (uint32_t)k * kFinTabRows + ys[v];
        tp[0] = x; tp[kFinTabSlots * kFinTabRows] = y; tp[2 * kFinTabSlots * kFinTabRows] = z;
      }
    }
  }
  // ---- the plans of this wave: the next plan's slice is in flight while a plan is worked on
  MNAV_GLOBAL const float* const gD = as_global(A.D) + (size_t)W.soff * NP;
  const uint32_t i0 = lane, i1 = lane + 64u, i2 = lane + 128u, i3 = lane + 192u;
  // Everything a plan needs is loaded one plan ahead, without a branch and record first.  (Until round 6 the record's fields were
  // separate loads issued AFTER the next slice's and needed at once, and the slice loads sat under `i < sl` branches, which leaves
  // the compiler without a count of the loads in flight: either way the wave waited for the slice it had just asked for -- the
  // prefetch overlapped nothing, SQ_WAIT_ANY 67 % of the wave time.)  Lanes past the slice's end repeat its last slot.
  const uint32_t slm = W.sl - 1u, j0 = min(i0, slm), j1 = min(i1, slm), j2 = min(i2, slm), j3 = min(i3, slm);
  MNAV_GLOBAL const uint32_t* const g_recs = (MNAV_GLOBAL const uint32_t*)as_global(F.recs) + min((uint32_t)lane, kFinRecWords - 1u);
  auto load_slice = [&](uint32_t p, uint32_t& rec, uint32_t (&r)[4]) {
    MNAV_GLOBAL const uint32_t* s = (MNAV_GLOBAL const uint32_t*)(gD + (size_t)p * W.sl);
    rec = g_recs[(size_t)p * kFinRecWords];
    r[0] = s[j0]; r[1] = s[j1]; r[2] = s[j2]; r[3] = s[j3];
  };
  uint32_t cur[4], nxt[4], rcur, rnxt;
  load_slice(p_beg, rcur, cur);
  rnxt = rcur;
  uint32_t bad = 0;
  for (uint32_t p = p_beg; p < p_end; ++p) {
#ifndef MNAV_FIN_NOSYNC
    __syncthreads();                                                  // the tiles of the patch write plan p together
#endif
    if (p + 1 < p_end) load_slice(p + 1, rnxt, nxt);
    auto rw = [&](int k) { return (uint32_t)__builtin_amdgcn_readlane((int)rcur, k); };
    MNAV_GLOBAL float* const g_dist = (MNAV_GLOBAL float*)(uintptr_t)(((unsigned long long)rw(1) << 32) | rw(0));
    MNAV_GLOBAL uint32_t* const g_pred = (MNAV_GLOBAL uint32_t*)(uintptr_t)(((unsigned long long)rw(3) << 32) | rw(2));
    MNAV_GLOBAL float* const g_vm = (MNAV_GLOBAL float*)(uintptr_t)(((unsigned long long)rw(5) << 32) | rw(4));
    const uint32_t seed = rw(6);
    const bool reached = __ballot(cur[0] != kTbInfBits || cur[1] != kTbInfBits || cur[2] != kTbInfBits || cur[3] != kTbInfBits) != 0ull;
    uint32_t cnt = 0;
    if (!reached) {                                                    // the plan's wave never came near this tile: dist = inf, pred = itself
#pragma unroll
      for (int v = 0; v < 2; ++v) if (own[v]) {
        g_dist[gid[v]] = inf_f(); g_pred[gid[v]] = gid[v];
        if (g_vm) { const f32x3_u z3 = { 0.f, 0.f, 0.f }; *(MNAV_GLOBAL f32x3_u*)(g_vm + 3 * (size_t)gid[v]) = z3; }
      }
    } else {
      GoalCut gcut; gcut.goal = 0.f; gcut.cut = u2f(rw(7)); gcut.tie = rw(8);   // dijkstra :296 (k_tb_fin_plans)
      __builtin_amdgcn_wave_barrier();                                 // (the previous plan's reads of ld[] are done: one wave, program order)
      if (i0 < W.sl) ld[i0] = cur[0];
      if (i1 < W.sl) ld[i1] = cur[1];
      if (i2 < W.sl) ld[i2] = cur[2];
      if (i3 < W.sl) ld[i3] = cur[3];
      for (uint32_t i = lane + 256u; i < W.sl; i += 64) ld[i] = ((MNAV_GLOBAL const uint32_t*)(gD + (size_t)p * W.sl))[i];   // (slices beyond 256 slots: rare meshes)
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        // ---- the common case without a branch: no source AT the cut value, one source attains the vertex's value
        const uint32_t dyb = ld[ys[v]];
        const float dy = u2f(dyb);
        uint32_t sum[kTbFinSlots], dsb[kTbFinSlots];
        bool at_cut = false;
#pragma unroll
        for (int k = 0; k < (int)kTbFinSlots; ++k) {
          dsb[k] = ld[src[v][k]];
          const float ds = u2f(dsb[k]);
          at_cut |= ds == gcut.cut;
          sum[k] = (ds < gcut.cut) ? f2u(ds + wt[v][k]) : kTbInfBits;   // expanded source (dijkstra :293-300), :331; an unused slot's weight is +inf
        }
        const uint32_t m = min(min(min(sum[0], sum[1]), min(sum[2], sum[3])), min(min(sum[4], sum[5]), min(sum[6], sum[7])));
        const bool is_seed = gid[v] == seed;
        const bool cut = dy > gcut.cut;                                // beyond goal_dist: the value is re-derived from the expanded sources
        uint32_t val = cut ? m : dyb;
        // the sources that attain the value compete with (d[x], x) for the predecessor: smallest d first
        uint32_t c[kTbFinSlots];
#pragma unroll
        for (int k = 0; k < (int)kTbFinSlots; ++k) c[k] = (sum[k] == val) ? dsb[k] : kTbInfBits;
        const uint32_t bd = min(min(min(c[0], c[1]), min(c[2], c[3])), min(min(c[4], c[5]), min(c[6], c[7])));
        uint32_t best = ys[v] | (kFinNoSlot << 16), nbest = 0;
#pragma unroll
        for (int k = (int)kTbFinSlots - 1; k >= 0; --k) { const bool hit = c[k] == bd; best = hit ? srck[v][k] : best; nbest += hit ? 1u : 0u; }
        uint32_t best_s = best & 0xFFFFu, bk = best >> 16;
        const bool finite = val != kTbInfBits;
        unsigned long long key = finite && bd != kTbInfBits ? (((unsigned long long)bd << 32) | lgid[best_s]) : ~0ull;
        uint32_t nbad = (m < val && !is_seed) ? 1u : 0u;
        // ---- the rest, exactly: a source AT the cut value (the vertex id decides whether it was expanded: negative offsets, ties with
        // goal_dist), several sources of the same potential attaining the value (the smaller id wins), valence above kTbFinSlots
        if (own[v] && (at_cut || (finite && nbest > 1u) || W.ovf_n)) {
          nbad = 0u;
#pragma unroll
          for (int k = 0; k < (int)kTbFinSlots; ++k) {
            const float ds = u2f(dsb[k]);
            bool ex = ds < inf_f() && ds < gcut.cut;
            if (ds == gcut.cut && ds < inf_f()) ex = lgid[src[v][k]] < gcut.tie;
            sum[k] = ex ? f2u(ds + wt[v][k]) : kTbInfBits;
          }
          val = cut ? kTbInfBits : dyb;
          if (cut) {
#pragma unroll
            for (int k = 0; k < (int)kTbFinSlots; ++k) val = min(val, sum[k]);
          }
          key = ~0ull; best_s = ys[v]; bk = kFinNoSlot;
#pragma unroll
          for (int k = 0; k < (int)kTbFinSlots; ++k) {
            if (sum[k] < val && !is_seed) ++nbad;
            if (sum[k] == val && val != kTbInfBits) {
              const unsigned long long kk = ((unsigned long long)dsb[k] << 32) | lgid[src[v][k]];
              if (kk < key) { key = kk; best_s = src[v][k]; }
            }
          }
          for (uint32_t e = 0; e < W.ovf_n; ++e) {
            const TbFinOvf o = F.ovf[W.ovf_off + e];
            if (o.y != ys[v]) continue;
            const uint32_t db = ld[o.src], g2 = lgid[o.src];
            if (!expanded_source(gcut, u2f(db), g2)) continue;
            const uint32_t sm = f2u(u2f(db) + F.ovf_w[W.ovf_off + e]);
            if (cut && sm < val) { val = sm; key = ~0ull; }            // (a smaller sum from the overflow list restarts the argmin)
            else if (sm < val && !is_seed) ++nbad;
            if (sm == val && val != kTbInfBits) {
              const unsigned long long kk = ((unsigned long long)db << 32) | g2;
              if (kk < key) { key = kk; best_s = o.src; }
            }
          }
        }
        if (!own[v]) continue;
        bad += nbad;
        uint32_t pv = gid[v];
        float outd = u2f(val);
        if (is_seed) { outd = dy; ++cnt; }
        else {
          if (val != kTbInfBits && key == ~0ull) ++bad;                // a finite value no expanded neighbour supports
          if (val != kTbInfBits) { pv = (uint32_t)key; ++cnt; }
        }
#ifdef MNAV_FIN_NOSTORE                                                // (timing experiments: 1 = no stores at all, 2 = no vector-map stores)
        if (MNAV_FIN_NOSTORE == 1) { bad += (f2u(outd) ^ pv) == 0x12345u ? 1u : 0u; }
        else { g_dist[gid[v]] = outd; g_pred[gid[v]] = pv; }
#else
        g_dist[gid[v]] = outd; g_pred[gid[v]] = pv;
#endif
        if (g_vm) {                                                    // k_vecmap_dijkstra's arithmetic
          float x = 0.f, y = 0.f, z = 0.f;
          if (pv != gid[v]) {                                          // :197
            if (bk < kFinTabSlots) {
              const float* const tp = ltab + bk * kFinTabRows + ys[v];
              x = tp[0]; y = tp[kFinTabSlots * kFinTabRows]; z = tp[2 * kFinTabSlots * kFinTabRows];
            } else {
              x = lxyz[3 * best_s] - px[v]; y = lxyz[3 * best_s + 1] - py[v]; z = lxyz[3 * best_s + 2] - pz[v];   // :204
              const float len = sqrtf(x * x + y * y + z * z);          // normalized(), :206
              x = x / len; y = y / len; z = z / len;
            }
          }
          const f32x3_u o3 = { x, y, z };
#ifdef MNAV_FIN_NOSTORE
          bad += (f2u(x) ^ f2u(y) ^ f2u(z)) == 0x12345u ? 1u : 0u;
#else
          *(MNAV_GLOBAL f32x3_u*)(g_vm + 3 * (size_t)gid[v]) = o3;
#endif
        }
      }
    }
    cnt = wave_sum(cnt);
    if (lane == 0 && cnt && live) atomicAdd(&F.res[p].settled, (unsigned long long)cnt);
#pragma unroll
    for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
    rcur = rnxt;
  }
  bad = wave_sum(bad);
  if (lane == 0 && bad) atomicAdd(F.mismatch, bad);
}

}  // namespace
