// mnav_tb_host.h -- host driver of the tile-batch SSSP engine (mnav_tb.h).  Included by mnav.hip inside its anonymous
// namespace, after mnav_ctx and the helpers (HIPCHK, dev_upload, ev_ms, PlanIn) are defined.
#pragma once

void tb_free_batch(mnav_ctx* ctx)
{
  TbState& S = ctx->tb;
  (void)hipFree(S.D); (void)hipFree(S.pend); (void)hipFree(S.pflag); (void)hipFree(S.pairs); S.pairs = nullptr; (void)hipFree(S.bucket); (void)hipFree(S.bcnt); (void)hipFree(S.items); (void)hipFree(S.ctl); (void)hipFree(S.wstat); S.wstat = nullptr;
  (void)hipFree(S.marr[0]); (void)hipFree(S.marr[1]);
  (void)hipFree(S.thr); (void)hipFree(S.bnd); (void)hipFree(S.seed); (void)hipFree(S.target); (void)hipFree(S.d_recs); S.d_recs = nullptr;
  if (S.h_ctl) (void)hipHostFree(S.h_ctl);
  for (int k = 0; k < 2; ++k) { if (S.graph[k]) (void)hipGraphExecDestroy(S.graph[k]); S.graph[k] = nullptr; }
  if (S.fill_stream) (void)hipStreamSynchronize(S.fill_stream);
  (void)hipFree(S.D2); S.D2 = nullptr; S.d2_clean = false; S.d2_wanted_np = 0u;
  S.D = nullptr; S.pend = nullptr; S.pflag = nullptr; S.bucket = nullptr; S.bcnt = nullptr; S.items = nullptr; S.ctl = nullptr; S.h_ctl = nullptr;
  S.marr[0] = S.marr[1] = nullptr; S.thr = S.bnd = nullptr; S.seed = S.target = nullptr;
  S.cap_np = 0;
  S.count_pending = false; ctx->tb_args_valid = false;               // the last batch's arguments point into freed memory now
}

void tb_free(mnav_ctx* ctx)
{
  TbState& S = ctx->tb;
  tb_free_batch(ctx);
  for (void* p : { (void*)S.d_tiles, (void*)S.d_stream, (void*)S.d_wsrc, (void*)S.d_exps, (void*)S.d_vaddr, (void*)S.d_vert_tile, (void*)S.d_verts,
                   (void*)S.d_vstream, (void*)S.d_vwsrc, (void*)S.d_vtile, (void*)S.d_vgroups, (void*)S.d_vexps,
                   (void*)S.d_fin_src, (void*)S.d_fin_wsrc, (void*)S.d_fin_ovf, (void*)S.d_fin_ovf_wsrc, (void*)S.d_ghost_gid, (void*)S.d_fin_order })
    if (p) { ctx->alloc_bytes.erase(p); (void)hipFree(p); }
  S.d_tiles = nullptr; S.d_stream = nullptr; S.d_wsrc = nullptr; S.d_exps = nullptr; S.d_vaddr = nullptr; S.d_vert_tile = nullptr; S.d_verts = nullptr;
  S.d_vstream = nullptr; S.d_vwsrc = nullptr; S.d_vtile = nullptr; S.d_vgroups = nullptr; S.d_vexps = nullptr;
  S.d_fin_src = nullptr; S.d_fin_wsrc = nullptr; S.d_fin_ovf = nullptr; S.d_fin_ovf_wsrc = nullptr; S.d_ghost_gid = nullptr; S.d_fin_order = nullptr;
  (void)hipFree(S.d_fin_w); S.d_fin_w = nullptr; (void)hipFree(S.d_fin_ovf_w); S.d_fin_ovf_w = nullptr; S.fin_w_valid = false;
  S.built = false; S.w_valid = false; S.vert_tile.clear();
  S.count_pending = false; ctx->tb_args_valid = false;
}

// mesh-dependent streams: built on the first batch that takes this engine (a few seconds of host work at 1M vertices)
int tb_build(mnav_ctx* ctx)
{
  TbState& S = ctx->tb;
  if (S.built) return 0;
  if (opt_set(ctx->opt.tb_tile)) S.T = opt_u32(ctx->opt.tb_tile, S.T);
  if (S.T != 64 && S.T != 96 && S.T != 120 && S.T != 128) S.T = 120;
  HostTopology t;
  t.V = ctx->V; t.E = ctx->E; t.F = ctx->F;
  t.row_ptr = ctx->h_row_ptr; t.nbr_u = ctx->h_nbr_u;
  HostTb H;
  try { H = build_tb(t, ctx->h_xyz.data(), S.T); }
  catch (const std::exception& ex) { ctx->err = ex.what(); return -1; }
  std::vector<uint2> vaddr(ctx->V);
  for (uint32_t v = 0; v < ctx->V; ++v) {
    const TbTile& W = H.tiles[H.vert_tile[v]];
    if (W.sl >= (1u << 24)) { ctx->err = "tile-batch engine: a tile has too many ghosts"; return -1; }
    vaddr[v] = make_uint2(W.soff, (W.sl << 8) | H.vert_local[v]);
  }
  if (dev_upload(ctx, &S.d_tiles, H.tiles.data(), H.tiles.size())) return -1;
  if (dev_upload(ctx, &S.d_stream, H.stream.data(), H.stream.size())) return -1;
  if (dev_upload(ctx, &S.d_wsrc, H.wsrc.data(), H.wsrc.size())) return -1;
  if (dev_upload(ctx, &S.d_exps, H.exps.data(), H.exps.size())) return -1;
  if (dev_upload(ctx, &S.d_vstream, H.vstream.data(), H.vstream.size())) return -1;
  if (dev_upload(ctx, &S.d_vwsrc, H.vwsrc.data(), H.vwsrc.size())) return -1;
  if (dev_upload(ctx, &S.d_vtile, H.vtile.data(), H.vtile.size())) return -1;
  if (H.vgroups.empty()) H.vgroups.push_back(0u);
  if (dev_upload(ctx, &S.d_vgroups, H.vgroups.data(), H.vgroups.size())) return -1;
  if (dev_upload(ctx, &S.d_vexps, H.vexps.data(), H.vexps.size())) return -1;
  if (dev_upload(ctx, &S.d_vaddr, vaddr.data(), vaddr.size())) return -1;
  if (dev_upload(ctx, &S.d_vert_tile, H.vert_tile.data(), H.vert_tile.size())) return -1;
  if (dev_upload(ctx, &S.d_verts, H.verts.data(), H.verts.size())) return -1;
  // finalize tables (mnav_tb_finalize.h)
  if (dev_upload(ctx, &S.d_fin_src, H.fin_src.data(), H.fin_src.size())) return -1;
  if (dev_upload(ctx, &S.d_fin_wsrc, H.fin_wsrc.data(), H.fin_wsrc.size())) return -1;
  if (dev_upload(ctx, &S.d_fin_ovf, H.fin_ovf.data(), H.fin_ovf.size())) return -1;
  {
    std::vector<uint32_t> ow(H.fin_ovf.size());
    for (size_t i = 0; i < ow.size(); ++i) ow[i] = H.fin_ovf[i].wsrc;
    if (dev_upload(ctx, &S.d_fin_ovf_wsrc, ow.data(), ow.size())) return -1;
  }
  if (dev_upload(ctx, &S.d_ghost_gid, H.ghost_gid.data(), H.ghost_gid.size())) return -1;
  {
    // k_tb_finalize's workgroups take the tiles in the order of their smallest vertex id: the tiles of a workgroup -- and of the
    // workgroups running next to it -- then write neighbouring pieces of the vertex-order output arrays (on a row-major grid: a
    // strip of tiles side by side, whose 11-vertex runs join to whole cache lines in the L2)
    std::vector<uint32_t> order(H.tiles.size()), key(H.tiles.size(), 0xFFFFFFFFu);
    for (size_t t = 0; t < H.tiles.size(); ++t) {
      order[t] = (uint32_t)t;
      for (uint32_t i = 0; i < H.tiles[t].nv; ++i) key[t] = std::min(key[t], H.verts[H.tiles[t].v0 + i]);
    }
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
    if (order.empty()) order.push_back(0u);
    if (dev_upload(ctx, &S.d_fin_order, order.data(), order.size())) return -1;
  }
  (void)hipFree(S.d_fin_w); S.d_fin_w = nullptr; (void)hipFree(S.d_fin_ovf_w); S.d_fin_ovf_w = nullptr;
  S.fin_n = H.fin_src.size(); S.fin_novf = H.fin_ovf.size(); S.fin_w_valid = false; S.max_sl = H.max_sl;
  HIPCHK(hipMalloc((void**)&S.d_fin_w, 4 * std::max<size_t>(S.fin_n, 1)));
  HIPCHK(hipMalloc((void**)&S.d_fin_ovf_w, 4 * std::max<size_t>(S.fin_novf, 1)));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  S.nvrec = H.vstream.size();
  S.ntiles = H.ntiles; S.S = H.S; S.nrec = H.stream.size(); S.nexp = H.exps.size(); S.max_nh = H.max_nh;
  S.vert_tile = std::move(H.vert_tile);
  S.built = true; S.w_valid = false;
  if (opt_on(ctx->opt.verbose))
    fprintf(stderr, "[mnav] tile-batch engine: T %u, %u tiles, %.2f slots per vertex, max ghosts %u, %.1f MB of streams + %.1f MB of sweep streams in the V layout\n", S.T, S.ntiles,
            ctx->V ? (double)S.S / ctx->V : 0.0, S.max_nh, (4.0 * S.nrec + 16.0 * S.nexp) / 1e6, 4.0 * S.nvrec / 1e6);
  return 0;
}

int tb_weights(mnav_ctx* ctx)
{
  TbState& S = ctx->tb;
  if (S.w_valid) return 0;
  hipLaunchKernelGGL(k_tb_weights, dim3(4096), dim3(kBlock), 0, ctx->stream, S.nrec, S.d_wsrc, ctx->d_nbr, S.d_stream);
  hipLaunchKernelGGL(k_tb_weights, dim3(4096), dim3(kBlock), 0, ctx->stream, S.nvrec, S.d_vwsrc, ctx->d_nbr, S.d_vstream);
  HIPCHK(hipGetLastError());
  S.w_valid = true; S.fin_w_valid = false;
  return 0;
}

// weights of the finalize tables: like the streams' they follow the cost-limit folded gather CSR (w_valid is reset with it)
int tb_fin_weights(mnav_ctx* ctx)
{
  TbState& S = ctx->tb;
  if (S.fin_w_valid) return 0;
  hipLaunchKernelGGL(k_tb_fin_weights, dim3(4096), dim3(kBlock), 0, ctx->stream, S.fin_n, S.d_fin_wsrc, ctx->d_nbr, S.d_fin_w);
  if (S.fin_novf) hipLaunchKernelGGL(k_tb_fin_weights, dim3(64), dim3(kBlock), 0, ctx->stream, S.fin_novf, S.d_fin_ovf_wsrc, ctx->d_nbr, S.d_fin_ovf_w);
  HIPCHK(hipGetLastError());
  S.fin_w_valid = true;
  return 0;
}

int tb_ensure_batch(mnav_ctx* ctx, uint32_t np)
{
  TbState& S = ctx->tb;
  if (np <= S.cap_np) return 0;
  tb_free_batch(ctx);
  const size_t nt = S.ntiles ? S.ntiles : 1, pairs = nt * (size_t)np;
  HIPCHK(hipMalloc((void**)&S.D, 4 * (size_t)S.S * np + 64));
  if (8 * (size_t)S.S * np <= ((size_t)96 << 30) && !opt_on(ctx->opt.tb_no_prefill)) {
    if (hipMalloc((void**)&S.D2, 4 * (size_t)S.S * np + 64) != hipSuccess) { S.D2 = nullptr; (void)hipGetLastError(); }
    if (S.D2 && !S.fill_stream) { HIPCHK(hipStreamCreateWithFlags(&S.fill_stream, hipStreamNonBlocking)); HIPCHK(hipEventCreateWithFlags(&S.fill_done, hipEventDisableTiming)); }
  }
  HIPCHK(hipMalloc((void**)&S.pend, 4 * pairs + 64));
  HIPCHK(hipMalloc((void**)&S.pflag, nt * (((size_t)np + 63) / 64) + 64));
  HIPCHK(hipMalloc((void**)&S.pairs, 4 * (nt * (((size_t)np + 63) / 64) + 64)));   // worst case: every (tile, block) flagged
  HIPCHK(hipMalloc((void**)&S.bucket, 2 * pairs + 64));
  HIPCHK(hipMalloc((void**)&S.bcnt, 4 * nt));
  HIPCHK(hipMalloc((void**)&S.items, 8 * (nt + pairs / kTbItemPlans + 64)));
  HIPCHK(hipMalloc((void**)&S.ctl, sizeof(tb::Ctl)));
  if (!S.wstat) HIPCHK(hipMalloc((void**)&S.wstat, 32 * (size_t)kTbStatSlots));
  HIPCHK(hipHostMalloc((void**)&S.h_ctl, sizeof(tb::Ctl), hipHostMallocDefault));
  for (int k = 0; k < 2; ++k) {
    HIPCHK(hipMalloc((void**)&S.marr[k], 4 * (size_t)np));
  }
  HIPCHK(hipMalloc((void**)&S.thr, 4 * (size_t)np)); HIPCHK(hipMalloc((void**)&S.bnd, 4 * (size_t)np));
  HIPCHK(hipMalloc((void**)&S.seed, 4 * (size_t)np)); HIPCHK(hipMalloc((void**)&S.target, 4 * (size_t)np));
  HIPCHK(hipMalloc((void**)&S.d_recs, sizeof(FinRec) * (size_t)np));
  S.cap_np = np;
  return 0;
}

int tb_fill(mnav_ctx* ctx, void* p, size_t bytes, uint32_t v)
{
  const size_t n16 = (bytes + 15) / 16;                               // the buffers carry 64 bytes of slack
  uint32_t g = (uint32_t)std::min<size_t>((n16 + kBlock - 1) / kBlock, 256 * 32);
  if (g < 1) g = 1;
  hipLaunchKernelGGL(k_tb_fill, dim3(g), dim3(kBlock), 0, ctx->stream, (u32x4*)p, n16, v);
  HIPCHK(hipGetLastError());
  return 0;
}

int tb_launch_iterations(mnav_ctx* ctx, const tb::Args& A, int count, uint32_t waves)
{
  const uint32_t gp = (A.NP + kBlock - 1) / kBlock;
  for (int j = 0; j < count; ++j) {
    const int par = j & 1;
    hipLaunchKernelGGL(k_tb_plan, dim3(gp), dim3(kBlock), 0, ctx->stream, A, par);
    hipLaunchKernelGGL(k_tb_pairs, dim3((A.n_flag16 + kBlock * kTbPairUnits - 1) / (kBlock * kTbPairUnits)), dim3(kBlock), 0, ctx->stream, A);
    hipLaunchKernelGGL(k_tb_scan, dim3(kTbScanWaves / (kBlock / 64)), dim3(kBlock), 0, ctx->stream, A, par);
    hipLaunchKernelGGL(k_tb_items, dim3((A.ntiles + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, A);
    if (ctx->tb.kernel == 1) hipLaunchKernelGGL((k_tbv_solve<120>), dim3(waves), dim3(64), 0, ctx->stream, A, ctx->tb.d_vtile, ctx->tb.d_vstream, ctx->tb.d_vgroups, ctx->tb.d_vexps, par);
    else if (ctx->tb.T == 64) hipLaunchKernelGGL((k_tb_solve_q<64>), dim3(waves), dim3(64), 0, ctx->stream, A, par);
    else if (ctx->tb.T == 96) hipLaunchKernelGGL((k_tb_solve_q<96>), dim3(waves), dim3(64), 0, ctx->stream, A, par);
    else if (ctx->tb.T == 120) hipLaunchKernelGGL((k_tb_solve_q<120>), dim3(waves), dim3(64), 0, ctx->stream, A, par);
    else hipLaunchKernelGGL((k_tb_solve_q<128>), dim3(waves), dim3(64), 0, ctx->stream, A, par);
  }
  hipLaunchKernelGGL(k_tb_stats, dim3(1), dim3(1024), 0, ctx->stream, A);
  HIPCHK(hipGetLastError());
  return 0;
}

// per-plan arrays in vertex order + the finalize pass, for calls that want V-sized outputs
int tb_fields(mnav_ctx* ctx, uint32_t n, const std::vector<PlanIn>& in, double offset, const tb::Args& A)
{
  TbState& S = ctx->tb;
  if (ensure_slots(ctx, n, false, false, ctx->want_vec)) return -1;
  if (tb_fin_weights(ctx)) return -1;
  std::vector<Plan> hp(n);
  std::vector<float*> vecs(n);
  for (uint32_t i = 0; i < n; ++i) {
    Slot& s = ctx->slots[i];
    Plan& P = hp[i];
    memset(&P, 0, sizeof(P));
    P.planner = kPlannerDijkstra; P.V = ctx->V;
    P.row_ptr = ctx->d_row_ptr; P.nbr = ctx->d_nbr; P.crn_ptr = ctx->d_crn_ptr; P.crn = ctx->d_crn; P.blocked = ctx->d_blocked;
    P.dist = s.dist; P.pred = s.pred; P.cap = ctx->V; P.ctl = s.ctl; P.cnt = s.cnt;
    P.offset = offset; P.max_steps = ctx->max_steps;
    for (int k = 0; k < 3; ++k) { P.seed[k] = in[i].seed[k]; P.target[k] = in[i].target[k]; P.seed_d[k] = 0.f; P.seed_expands[k] = 1; P.target_expands[k] = 1; }
    P.seed_face = kNone;
    vecs[i] = s.vecmap;
  }
  HIPCHK(hipMemcpyAsync(ctx->d_plans, hp.data(), sizeof(Plan) * n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->d_vecptrs, vecs.data(), sizeof(float*) * n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));                          // hp / vecs go out of scope
  // potential (with the reference's tentative values beyond goal_dist), predecessors and vector map of every plan, in vertex
  // order, straight from the blocked distances: one wave per (tile, 64 plans), eight tiles per workgroup, mnav_tb_finalize.h
  FinTb F{};
  F.src = S.d_fin_src; F.w = S.d_fin_w; F.ovf = S.d_fin_ovf; F.ovf_w = S.d_fin_ovf_w; F.verts = S.d_verts; F.ghost_gid = S.d_ghost_gid; F.order = S.d_fin_order;
  F.xyz = ctx->d_xyz; F.vecmaps = ctx->want_vec ? ctx->d_vecptrs : nullptr;
  F.plans = ctx->d_plans; F.res = ctx->d_res; F.mismatch = ctx->d_mismatch; F.recs = S.d_recs;
  const uint32_t groups = (S.ntiles + kFinWaves - 1u) / kFinWaves;    // a workgroup = kFinWaves consecutive tiles of the bisection order
#ifndef MNAV_FIN_PPW
#define MNAV_FIN_PPW 64u                  // plans a wave of k_tb_finalize walks; ms of the pass per 7168-plan batch at 32 / 64 / 128 / 256: 65.8 / 66.1 / 70.6 / 79.3
#endif
  F.plans_per_wave = MNAV_FIN_PPW; F.tiles_per_xcd = (groups + 7u) / 8u; F.max_sl = S.max_sl;
  hipLaunchKernelGGL(k_tb_fin_plans, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, A, ctx->d_plans, F.vecmaps, S.d_recs);
  const uint32_t npg = (n + F.plans_per_wave - 1u) / F.plans_per_wave;
  const size_t lds = 4 * (size_t)fin_lds_words(S.max_sl, F.vecmaps != nullptr) * kFinWaves;
  const dim3 grid(8u * F.tiles_per_xcd * npg);
  if (S.T == 64) hipLaunchKernelGGL((k_tb_finalize<64>), grid, dim3(64 * kFinWaves), lds, ctx->stream, A, F);
  else if (S.T == 96) hipLaunchKernelGGL((k_tb_finalize<96>), grid, dim3(64 * kFinWaves), lds, ctx->stream, A, F);
  else if (S.T == 120) hipLaunchKernelGGL((k_tb_finalize<120>), grid, dim3(64 * kFinWaves), lds, ctx->stream, A, F);
  else hipLaunchKernelGGL((k_tb_finalize<128>), grid, dim3(64 * kFinWaves), lds, ctx->stream, A, F);
  HIPCHK(hipGetLastError());
  return 0;
}

// Dijkstra batches through the tile-batch engine.  Returns 0, -1 (error) or 1 (cancelled).
// The clean-up of the OTHER distance buffer for the next call, launched by the entry point when this call's last result has been
// downloaded: on its own stream, with nothing of this call left on the device.  Round 6 measured where it must NOT run
// (profiles/r06_fill_modes.txt, r06_fill_trace.txt): next to the call's own kernels.  Whatever kernel of the call's stream is in
// flight when the fill starts does not finish before the fill does -- a 5 us memset took 6.9 ms next to a 7.5 ms fill -- and that
// is the fill's DURATION, not its rate: 48 / 24 / 12 / 6 workgroups stretched the batch by 0 / 6 / 18 / 40 ms.  So it runs at
// full width (44 GB in 7 ms at 7168 plans) in the gap the host leaves between two calls; a call that comes sooner waits for the
// event, which costs what cleaning at its start would.
int tb_clean_other(mnav_ctx* ctx)
{
  TbState& S = ctx->tb;
  if (!S.D2 || !S.fill_stream || S.d2_wanted_np == 0u || S.d2_clean) return 0;
  const uint32_t n = S.d2_wanted_np;
  S.d2_wanted_np = 0u;
  const size_t n16 = (4 * (size_t)S.S * n + 15) / 16;
  const uint32_t g = (uint32_t)std::max<size_t>(std::min<size_t>((n16 + kBlock - 1) / kBlock, 256 * 32), 1);
  hipLaunchKernelGGL(k_tb_fill, dim3(g), dim3(kBlock), 0, S.fill_stream, (u32x4*)S.D2, n16, kTbInfBits);
  HIPCHK(hipEventRecord(S.fill_done, S.fill_stream));
  S.d2_clean = true; S.d2_clean_np = n;
  return 0;
}

int run_dijkstra_tb(mnav_ctx* ctx, uint32_t n, const std::vector<PlanIn>& in, double offset)
{
  TbState& S = ctx->tb;
  if (tb_build(ctx)) return -1;
  if (tb_weights(ctx)) return -1;
  if (n > 65535u) { ctx->err = "tile-batch engine: more than 65535 plans in one batch"; return -1; }
  if (tb_ensure_batch(ctx, n)) return -1;
  if (ensure_plan_tables(ctx, n)) return -1;
  if (ensure_paths(ctx, n)) return -1;
  if (!ctx->d_mismatch) HIPCHK(hipMalloc((void**)&ctx->d_mismatch, 4));
  std::vector<uint32_t> seeds(n), targets(n);
  for (uint32_t i = 0; i < n; ++i) { seeds[i] = in[i].seed[0]; targets[i] = in[i].target[0]; }
  HIPCHK(hipMemcpyAsync(S.seed, seeds.data(), 4 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(S.target, targets.data(), 4 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->d_res, 0, sizeof(PlanResult) * n, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->d_mismatch, 0, 4, ctx->stream));
  tb::Args A{};
  A.tiles = S.d_tiles; A.stream = S.d_stream; A.exps = S.d_exps; A.D = S.D; A.pend = S.pend; A.pflag = S.pflag; A.nblk = (n + 63u) / 64u; A.NP = n; A.ntiles = S.ntiles;
  A.bucket = S.bucket; A.bcnt = S.bcnt; A.items = S.items; A.ctl = S.ctl; A.wstat = S.wstat; A.wstat_slots = kTbStatSlots;
  A.pairs = S.pairs; A.n_flag16 = (uint32_t)(((size_t)S.ntiles * A.nblk + 15u) / 16u);
  tb_div_magic(A.nblk, &A.nblk_magic, &A.nblk_shift);
  if ((size_t)S.ntiles * A.nblk >= (1ull << 31)) { ctx->err = "tile-batch engine: flag matrix beyond 2^31 entries"; return -1; }
  A.marr[0] = S.marr[0]; A.marr[1] = S.marr[1];
  A.thr = S.thr; A.bnd = S.bnd; A.seed = S.seed; A.target = S.target; A.vaddr = S.d_vaddr; A.vert_tile = S.d_vert_tile;
  A.offset = offset;
  // Solve kernel and band.  k_tbv_solve (distances in registers, mnav_tbv.h) takes whole waves per tile.  A plan's front crosses
  // ~sqrt(tiles) tiles per iteration, so a tile sees about 0.85 n / sqrt(tiles) plans per iteration and band of 2 tile widths
  // (measured: 63 at C2 = 7168 plans on 9 260 tiles, 11 at C4 = 4096 plans on 92 600 tiles); a wider band puts more plans on a
  // tile per iteration at the price of more activations.  Measured round 6 (engine ms per batch, k_tb_solve_q band 2 /
  // k_tbv_solve band 2 / band 4; density = n / sqrt(tiles)): 1M mesh 128 plans (density 1.3) 49 / 55 / 53, 512 (5.3) 64 / 72 / 67,
  // 2048 (21) 100 / 86 / 83, 7168 (74) 194 / 124 / 133; 10M mesh 4096 plans (13.5) 1524 / 1292 / 1180 (band 6: 1176).
  const bool v_fits = S.T == 120 && S.max_nh <= kTbvGhostRows;        // (its register window holds 120 owned rows and 64 ghosts)
  const double density = (double)n / std::sqrt((double)std::max(S.ntiles, 1u));
  S.kernel = (v_fits && density >= 8.0) ? 1 : 0;
  if (opt_set(ctx->opt.tb_kernel)) S.kernel = (opt_u32(ctx->opt.tb_kernel, 0u) == 1u && v_fits) ? 1 : 0;
  A.item_plans = S.kernel == 1 ? 64u : kTbItemPlans;
  {
    const float band_mult = (S.kernel == 1 && density < 40.0) ? 4.0f : S.band_mult;
    float band = (ctx->delta_auto / 3.0f) * std::sqrt((float)S.T) * band_mult;   // potential across one tile
    if (opt_set(ctx->opt.tb_band_mult)) band = (ctx->delta_auto / 3.0f) * std::sqrt((float)S.T) * (float)ctx->opt.tb_band_mult;
    if (ctx->tile_band_user > 0.f) band = ctx->tile_band_user;
    A.band = band;
  }
  HIPCHK(hipEventRecord(ctx->ev[1], ctx->stream));
  bool prefilled = false;
  if (S.D2 && S.d2_clean && S.d2_clean_np >= n) {                     // a clean buffer was prepared behind the previous call
    std::swap(S.D, S.D2); A.D = S.D;
    HIPCHK(hipStreamWaitEvent(ctx->stream, S.fill_done, 0));
    prefilled = true;
  }
  S.d2_clean = false;
  if (!prefilled && tb_fill(ctx, S.D, 4 * (size_t)S.S * n, kTbInfBits)) return -1;
  if (tb_fill(ctx, S.pend, 4 * (size_t)S.ntiles * n, kTbInfBits)) return -1;
  HIPCHK(hipMemsetAsync(S.pflag, 0, (size_t)(S.ntiles ? S.ntiles : 1) * ((n + 63u) / 64u) + 64u, ctx->stream));   // (+ the padding k_tb_pairs reads as 16-byte units)
  if (tb_fill(ctx, S.marr[0], 4 * (size_t)n, kTbInfBits)) return -1;
  if (tb_fill(ctx, S.marr[1], 4 * (size_t)n, kTbInfBits)) return -1;
  HIPCHK(hipMemsetAsync(S.ctl, 0, sizeof(tb::Ctl), ctx->stream));
  HIPCHK(hipMemsetAsync(S.wstat, 0, 32 * (size_t)kTbStatSlots, ctx->stream));
  HIPCHK(hipMemsetAsync(S.bcnt, 0, 4 * (size_t)(S.ntiles ? S.ntiles : 1), ctx->stream));
  hipLaunchKernelGGL(k_tb_seed, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, A);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(ctx->ev[2], ctx->stream));
  // (the other buffer -- the previous call's distances, superseded the moment this call began -- is cleaned for the NEXT call at the
  // end of this one: tb_clean_other)
  S.d2_wanted_np = S.D2 ? n : 0u;

  int ncu = 256;
  (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device);
  uint32_t per_cu = (uint32_t)((160u * 1024u) / (S.T * 256u + kTbQStride * 4u));   // LDS: T x 256 bytes + staging per wave, 160 KB per CU
  if (S.kernel == 1) per_cu = 8u;                                      // k_tbv_solve: 256 VGPRs per wave, two waves per SIMD, no LDS
  if (opt_set(ctx->opt.tb_waves_per_cu)) S.waves_per_cu = (int)ctx->opt.tb_waves_per_cu;
  if (S.waves_per_cu > 0) per_cu = (uint32_t)S.waves_per_cu;
  const uint32_t waves = per_cu * (uint32_t)ncu;
  const int chunk = S.iters_per_replay & ~1;
  // graph of `chunk` iterations, re-captured when the kernel arguments change
  int gi = 0;                                                         // graph slot: the one captured with these arguments, else the older one
  for (int k = 0; k < 2; ++k) if (S.graph[k] && memcmp(&S.graph_args[k], &A, sizeof(A)) == 0) gi = k + 2;
  if (ctx->use_graph && gi < 2) {
    gi = (S.graph[0] && S.graph_args[0].D != A.D) ? 1 : 0;
    if (S.graph[gi]) { (void)hipGraphExecDestroy(S.graph[gi]); S.graph[gi] = nullptr; }
    hipGraph_t g = nullptr;
    HIPCHK(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    const int rc = tb_launch_iterations(ctx, A, chunk, waves);
    const hipError_t e = hipStreamEndCapture(ctx->stream, &g);
    if (rc != 0 || e != hipSuccess) { ctx->err = "graph capture failed"; return -1; }
    HIPCHK(hipGraphInstantiate(&S.graph[gi], g, nullptr, nullptr, 0));
    (void)hipGraphDestroy(g);
    memcpy(&S.graph_args[gi], &A, sizeof(A));
  } else gi -= 2;
  int rc = 0;
  uint32_t iters = 0;
  const auto t_start = std::chrono::steady_clock::now();
  HIPCHK(hipEventRecord(ctx->evc[0], ctx->stream));
  for (;;) {
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() > ctx->max_wall_s) {
      ctx->err = "tile-batch iterations exceeded the wall-clock guard"; return -1;
    }
    if (ctx->use_graph) HIPCHK(hipGraphLaunch(S.graph[gi], ctx->stream));
    else if (tb_launch_iterations(ctx, A, chunk, waves)) return -1;
    iters += (uint32_t)chunk;
    HIPCHK(hipMemcpyAsync(S.h_ctl, S.ctl, sizeof(tb::Ctl), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (S.h_ctl->err) { ctx->err = "tile-batch engine: sweep cap hit"; return -1; }
    if (S.h_ctl->n_cand[0] == 0u) break;                             // the chunk ends on odd parity: its pending count is counter 0
    if (ctx->cancel.load(std::memory_order_relaxed)) { rc = 1; break; }
    if (iters > ctx->max_steps) { ctx->err = "tile-batch engine: iteration cap hit"; return -1; }
  }
  HIPCHK(hipEventRecord(ctx->evc[1], ctx->stream));
  if (rc == 0 && !ctx->lazy_paths) {
    // V-sized outputs wanted (potential, predecessors, vector map): the finalize pass of the tile engines derives the
    // reference's exact cut-off semantics and predecessors straight from the blocked distances (k_tb_finalize)
    if (tb_fields(ctx, n, in, offset, A)) return -1;
  }
  HIPCHK(hipEventRecord(ctx->ev[3], ctx->stream));
  HIPCHK(hipEventSynchronize(ctx->evc[1]));
  ctx->ms_chunks = ev_ms(ctx->evc[0], ctx->evc[1]);
  S.last = *S.h_ctl;
  ctx->stats.launches = 1;                                           // one engine run per batch (iterations: stats.steps)
  ctx->tb_args = A; ctx->tb_args_valid = (rc == 0);
  if (opt_on(ctx->opt.trace))
    fprintf(stderr, "[mnav] tile-batch: %u iterations, %llu activations (%.1f per item), %llu items, %.2f sweeps per item, %llu wakes\n", S.h_ctl->iters,
            S.h_ctl->acts, S.h_ctl->items ? (double)S.h_ctl->acts / S.h_ctl->items : 0.0, S.h_ctl->items,
            S.h_ctl->items ? (double)S.h_ctl->sweeps / S.h_ctl->items : 0.0, S.h_ctl->wakes);
  return rc;
}
