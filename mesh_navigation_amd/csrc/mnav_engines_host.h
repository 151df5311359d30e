// mnav_engines_host.h -- host drivers of the per-plan engines: the band steps (run_plans: CVP, inflation wave, Dijkstra on request),
// the tile rounds (run_dijkstra_tiled) and the asynchronous tiles
// (run_dijkstra_async), with the finalize launchers and the per-slot tile state.  Included by mnav.hip inside its anonymous
// namespace, after mnav_ctx and its helpers; not a stand-alone header.
#pragma once

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
  if (opt_set(ctx->opt.cvp_wide)) wide = cvp && ctx->opt.cvp_wide != 0.0;
  uint32_t G = blocks_per_plan(ctx);
  if (wide) {
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device);
    G = (kWideVerts == 64u ? 7u : 12u) * (uint32_t)ncu;               // waves of the whole batch, not per plan: what stays resident
    if (opt_set(ctx->opt.wide_waves)) G = std::max(1u, opt_u32(ctx->opt.wide_waves, G));
    ctx->wide_groups = std::min(4u, std::max(1u, n / 40u));            // measured on the benched C3 configuration, plans/s with 1 / 2 / 3 / 4 / 8 groups:
                                                                      // 128 plans 277 / 315 / 320 / 320 / 221, 512 plans 330 / 425 / 463 / 468 / 422
    if (opt_set(ctx->opt.cvp_groups)) ctx->wide_groups = std::min(std::max(1u, opt_u32(ctx->opt.cvp_groups, 1u)), (uint32_t)kWideGroupsMax);
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
  if (cvp && rc == 0) {                                               // second fire events around the seed faces (k_cvp_seed_ring)
    hipLaunchKernelGGL(k_cvp_seed_ring, dim3(n), dim3(kWave), 0, ctx->stream, ctx->d_plans);
    HIPCHK(hipGetLastError());
  }
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
  hipLaunchKernelGGL(k_dij_finalize, dim3(n, chunks), dim3(kTileBlock), fin_lds, ctx->stream, ctx->d_plans, ctx->d_tplans,
                     ctx->d_mismatch, ctx->d_res, per, n);
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
  if (opt_set(ctx->opt.tile_blocks)) G = opt_u32(ctx->opt.tile_blocks, G);
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

// Dijkstra through the asynchronous tile engine (mnav_async.h): ONE launch for the whole call.  Returns 0, -1, 1 (cancelled),
// 2 (the ticket ring ran out: nothing of the call is usable, run it on another engine) or 3 (the in-kernel watchdog gave up).
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
    T.pend1_is_state = 1u;
    // band of the plan in tile widths (<= 0: one band, every solve runs to its tile's fixed point); measured round 5, ms per call at
    // 1 / 8 / 47 / 64 plans on the 1M mesh: 2 widths 7.1 / 11.1 / 20.8 / 25.4, 3: 6.9 / 10.8 / 21.9 / 26.8, 4: 6.8 / 9.9 / 23.2 / 27.7, 6: 7.2 / 10.3 / 25.7 / 31.8, 8: 7.1 / 10.6 / 28.2 / 34.8
    T.band = (float)((opt_set(ctx->opt.async_band_mult) ? ctx->opt.async_band_mult : (n >= 32u ? 2.0 : 4.0)) * ctx->tile_band_auto);
    T.max_nv = M.max_nv; T.max_nh = M.max_nh; T.max_ne = M.max_ne;
  }
  AsyncCtl* const actl = reinterpret_cast<AsyncCtl*>(ctx->d_cancel + 4);   // words 4..11 of the 64-byte control line (word 0: mnav_cancel)
  // the ticket ring: every slot is written at most once per call, so it only has to hold what a call can file (a tile is
  // re-filed when a neighbour undercuts it after its solve: a handful of times) -- 16 per tile and plan; a call that runs out
  // gives up (abort 5) and is re-run on the tile rounds by the caller
  // the ticket rings, one per plan: every slot is written at most once per call, so a ring only has to hold what a plan can file
  // (a tile is re-filed when a neighbour undercuts it after its solve: a handful of times) -- 16 per tile; a call that runs out
  // gives up (abort 5) and is re-run on the tile rounds by the caller
  const bool ring_forced = opt_set(ctx->opt.async_ring_cap);
  const uint32_t cap1 = ring_forced ? std::max(2u, opt_u32(ctx->opt.async_ring_cap, 0u)) : std::max(1024u, std::min(M.ntiles * 16u, 1u << 24));
  if (ctx->ring_words < (size_t)cap1 * n) {
    (void)hipFree(ctx->d_ring); ctx->d_ring = nullptr; ctx->ring_words = 0;
    HIPCHK(hipMalloc((void**)&ctx->d_ring, 4 * (size_t)cap1 * n));
    ctx->ring_words = (size_t)cap1 * n;
  }
  ctx->ring_cap = cap1;
  {
    const size_t per_plan = 2u * (size_t)aq::kParkedLists * M.ntiles;   // the two parked lists of a plan
    if (ctx->parked_words < per_plan * n) {
      (void)hipFree(ctx->d_parked); ctx->d_parked = nullptr; ctx->parked_words = 0;
      HIPCHK(hipMalloc((void**)&ctx->d_parked, 4 * per_plan * n));
      ctx->parked_words = per_plan * n;
    }
    for (uint32_t i = 0; i < n; ++i) tp[i].parked = ctx->d_parked + per_plan * i;
  }
  HIPCHK(hipMemcpyAsync(ctx->d_plans, hp.data(), sizeof(Plan) * n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->d_tplans, tp.data(), sizeof(TilePlan) * n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->d_vecptrs, vecs.data(), sizeof(float*) * n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->d_res, 0, sizeof(PlanResult) * n, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->d_mismatch, 0, 4, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->d_ring, 0xFF, 4 * (size_t)ctx->ring_cap * n, ctx->stream));
  HIPCHK(hipEventRecord(ctx->ev[1], ctx->stream));
  uint32_t gi = (ctx->V + kBlock * 4 - 1) / (kBlock * 4);
  if (gi < 1) gi = 1;
  if (gi > 4096) gi = 4096;
  hipLaunchKernelGGL(k_init<kPlannerDijkstra>, dim3(gi, n), dim3(kBlock), 0, ctx->stream, ctx->d_plans);
  {
    uint32_t gt = (M.ntiles + kBlock - 1) / kBlock;
    if (gt < 1) gt = 1;
    hipLaunchKernelGGL(k_tile_init, dim3(gt, n), dim3(kBlock), 0, ctx->stream, ctx->d_tplans, ctx->d_vert_tile, -inf_f());
    hipLaunchKernelGGL(k_async_init, dim3(gt, n), dim3(kBlock), 0, ctx->stream, ctx->d_tplans, ctx->d_ring, ctx->d_vert_tile, actl, n, ctx->ring_cap);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(ctx->ev[2], ctx->stream));
  // Workgroups: what is resident at once (4 per CU: 36 KB of LDS each), and no more per plan than its wave front has tiles for
  // (a workgroup without a ticket polls one word)
  int ncu = 256;
  (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device);
  const uint32_t per_cu = std::max(1u, opt_u32(ctx->opt.async_wg_per_cu, 4u)), per_plan = std::max(1u, opt_u32(ctx->opt.async_wg_per_plan, 128u));
  uint32_t G = (uint32_t)std::min<uint64_t>((uint64_t)ncu * per_cu, (uint64_t)n * per_plan);
  if (G > M.ntiles * n) G = M.ntiles * n;
  if (G < 1) G = 1;
  // in-kernel give-up (100 MHz wall clock): async_max_s when set; a max_wall_s the caller set himself is honoured; 10 s otherwise
  const double guard_s = opt_set(ctx->opt.async_max_s) && ctx->opt.async_max_s > 0.0 ? ctx->opt.async_max_s
                       : opt_set(ctx->opt.max_wall_s) ? ctx->max_wall_s : std::min(ctx->max_wall_s, 10.0);
  const unsigned long long limit_ticks = (unsigned long long)(guard_s * 1.0e8);
  ctx->ms_chunks = 0.0;
  HIPCHK(hipEventRecord(ctx->evc[0], ctx->stream));
  if (ctx->tile_size <= 2 * kTileBlock) hipLaunchKernelGGL(k_plan_async<2>, dim3(G), dim3(kTileBlock), ctx->tile_lds, ctx->stream, ctx->d_tplans, n, actl, ctx->d_ring, limit_ticks);
  else if (ctx->tile_size <= 4 * kTileBlock) hipLaunchKernelGGL(k_plan_async<4>, dim3(G), dim3(kTileBlock), ctx->tile_lds, ctx->stream, ctx->d_tplans, n, actl, ctx->d_ring, limit_ticks);
  else hipLaunchKernelGGL(k_plan_async<8>, dim3(G), dim3(kTileBlock), ctx->tile_lds, ctx->stream, ctx->d_tplans, n, actl, ctx->d_ring, limit_ticks);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(ctx->evc[1], ctx->stream));
  AsyncCtl& h = *ctx->h_actl;                                         // (pinned)
  HIPCHK(hipMemcpyAsync(&h, actl, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->ms_chunks = ev_ms(ctx->evc[0], ctx->evc[1]);
  ctx->stats.launches = 1;
  if (opt_on(ctx->opt.verbose))
    fprintf(stderr, "[mnav] async: %u plans, %u workgroups, %.3f ms, abort %u, tickets %u, polls %u, retired beyond the bound %u, plan switches %u\n", n, G, ctx->ms_chunks, h.abort, h.tickets, h.polls, h.dropped, h.switches);
  if (h.abort == 3u || ctx->cancel.load(std::memory_order_relaxed)) return 1;   // :350-354
  if (h.abort == 5u) return 2;                                         // out of ticket slots: the caller re-runs the call on the tile rounds
  if (h.abort) { ctx->err = "asynchronous tile engine gave up (in-kernel wall-clock guard)"; return 3; }   // (the caller turns 3 into a failure unless `auto` picked the engine)
  if (h.done_plans != n) { ctx->err = "asynchronous tile engine left plans unfinished"; return -1; }
  if (!ctx->lazy_paths) launch_finalize(ctx, n);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(ctx->ev[3], ctx->stream));
  return 0;
}
