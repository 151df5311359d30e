// async_model.cpp -- CPU model of the PROTOCOL of the asynchronous tile engine (mesh_navigation_amd/csrc/mnav_async.h).
//
// TEST INFRASTRUCTURE ONLY (lives under oracle/): never linked into, loaded by, or reachable from the product library.
//
// The engine has no rounds and no scans: workgroups serve a TICKET QUEUE of woken tiles, solve them and wake their neighbours
// concurrently; the first waker of an idle tile files its ticket, and a plan is finished when its count of filed-and-not-retired
// tickets reaches zero.  What can go wrong there is the protocol -- a lost wake-up (a tile woken while it is in solve that nobody
// files again), a premature "finished", two solvers on one tile, a ticket filed twice -- not the tile solve (k_tile_round's, tested
// on the device).  This model restates k_plan_async operation by operation (same words, same order of the shared-memory
// operations, same decisions) on the tile tables the product builds (mnav_build.h::build_tiles) and runs W virtual workgroups as
// threads of which exactly ONE runs at a time: before every shared-memory operation a workgroup hands the baton to a
// pseudo-randomly chosen one (seeded: reproducible), so a test sweeps thousands of different interleavings at the granularity of
// single atomics.  Checked while it runs: at every plan_finish no tile of the plan is pending, queued or in solve, and it happens
// once per plan; never two solvers on a tile; never two live tickets of one tile.  Checked by the test: the distances against the
// sequential oracle.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <random>
#include <thread>
#include <vector>

#include "../mesh_navigation_amd/csrc/mnav_build.h"

using namespace mnav;

namespace {
constexpr uint32_t kInf = 0x7f800000u;
constexpr uint32_t kWakeSlots = 32;       // kAsyncWake

struct PlanState {
  std::vector<uint32_t> dist, pend, lock, tlast;   // float bits; lock = the state word (1: a ticket is filed or the tile is in solve)
  std::vector<uint8_t> ticketed;                   // model only: a ticket of the tile is in the ring and not yet retired
  uint32_t thr = 0, par = 0, nparked[2] = { 0, 0 }, epochs = 0;   // the plan's band: threshold (float bits), parity of the current parked list
  std::vector<uint32_t> ring; uint32_t head = 0, tail = 0, done_flag = 0;   // the plan's ticket ring (kNone = not filed yet)
  std::vector<uint32_t> parked[2];
  uint32_t work = 1, acts = 0, sweeps = 0, done = 0, finishes = 0;
  uint32_t seed = 0, target = 0;
  std::vector<uint8_t> in_solve;                   // model only: a workgroup is between claim and unlock
};

struct Model {
  HostTiles T;
  std::vector<float> tw;
  std::vector<PlanState> plans;
  double offset = 0.0; float band = 1.f;
  // scheduler: one runner at a time
  std::mutex m; std::condition_variable cv;
  int turn = 0; std::vector<uint8_t> alive; std::mt19937 rng;
  uint64_t yields = 0, claim_fails = 0, drops = 0, violations = 0, solves_now = 0, max_solves = 0, putbacks = 0, raised = 0;
  uint32_t done_plans = 0, abort = 0;
  uint64_t budget = 0;                             // yield budget: the model's wall-clock guard
  // deliberately broken variants, to show that the checks see protocol errors (tests/test_async_model.py): 2 = the solver does not look at the wake-up value again after clearing the state word (a wake-up that
  // arrived during the solve is lost), 3 = wakers file a ticket whatever the state word says (two solvers on a tile)
  uint32_t mutate = 0;

  // Uniformly random hand-offs almost never stall ONE workgroup for the length of another one's whole solve -- the windows
  // protocol errors hide in (a waker between its atomicMin and its count, a claimer between its scan and its claim).  So
  // every now and then the running workgroup is put to sleep for up to a few thousand scheduling points.
  std::vector<uint64_t> sleep_until;
  int next_runner()
  {
    std::vector<int> c;
    for (size_t i = 0; i < alive.size(); ++i) if (alive[i] && sleep_until[i] <= yields) c.push_back((int)i);
    if (c.empty()) for (size_t i = 0; i < alive.size(); ++i) if (alive[i]) c.push_back((int)i);
    return c.empty() ? -1 : c[rng() % c.size()];
  }
  void pass(int me)
  {
    std::unique_lock<std::mutex> lk(m);
    ++yields;
    if (yields > budget) abort = 2;
    if (rng() % 48u == 0u) sleep_until[me] = yields + 1u + rng() % 6000u;
    turn = next_runner();
    cv.notify_all();
    cv.wait(lk, [&] { return turn == me; });
  }
  void leave(int me)
  {
    std::unique_lock<std::mutex> lk(m);
    alive[me] = 0;
    const int nx = next_runner();
    if (nx >= 0) { turn = nx; cv.notify_all(); }
  }
  void enter(int me)
  {
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [&] { return turn == me; });
  }
};

struct Wg {
  Model& M; int me; std::mt19937 rng;
  Wg(Model& m_, int me_, uint32_t seed) : M(m_), me(me_), rng(seed) {}
  // every shared-memory operation is preceded by a scheduling point
  uint32_t ld(const uint32_t& w) { M.pass(me); return w; }
  void st(uint32_t& w, uint32_t v) { M.pass(me); w = v; }
  uint32_t add(uint32_t& w, uint32_t v) { M.pass(me); const uint32_t o = w; w = o + v; return o; }
  uint32_t sub(uint32_t& w, uint32_t v) { M.pass(me); const uint32_t o = w; w = o - v; return o; }
  uint32_t amin(uint32_t& w, uint32_t v) { M.pass(me); const uint32_t o = w; if (v < o) w = v; return o; }
  uint32_t xchg(uint32_t& w, uint32_t v) { M.pass(me); const uint32_t o = w; w = v; return o; }
  bool cas(uint32_t& w, uint32_t e, uint32_t v) { M.pass(me); if (w != e) return false; w = v; return true; }

  uint32_t amax(uint32_t& w, uint32_t v) { M.pass(me); const uint32_t o = w; if (v > o) w = v; return o; }

  void plan_finish(PlanState& P)
  {
    // model-only checks, on a consistent snapshot (nobody else runs): nothing of the plan is pending, parked, queued or in solve
    for (size_t t = 0; t < P.pend.size(); ++t)
      if (P.pend[t] != kInf || P.lock[t] != 0u || P.in_solve[t] || P.ticketed[t]) ++M.violations;
    if (P.work != 0) ++M.violations;
    ++P.finishes;
    st(P.done_flag, 1u);
    add(M.done_plans, 1u);
  }
  void push(PlanState& P, uint32_t p, uint32_t t, uint32_t* filed = nullptr)
  {
    if (P.ticketed[t]) ++M.violations;                               // a second live ticket of the tile
    P.ticketed[t] = 1;
    if (filed) ++*filed;                                             // covered by the caller's reservation
    else add(P.work, 1u);                                            // counted before it can be seen
    (void)p;
    const uint32_t i = add(P.tail, 1u);
    if (i < P.ring.size()) st(P.ring[i], t); else st(M.abort, 5u);
  }
  void park(PlanState& P, uint32_t t, uint32_t par)
  {
    const uint32_t i = add(P.nparked[par], 1u);
    if (i < P.parked[par].size()) st(P.parked[par][i], t); else st(M.abort, 5u);
  }
  void route(PlanState& P, uint32_t p, uint32_t t2, uint32_t v, float thr, uint32_t par, uint32_t* filed = nullptr)
  {
    if (u2f(v) < thr) {
      if (M.mutate == 3u) { amax(P.lock[t2], 3u); push(P, p, t2, filed); return; }
      if (amax(P.lock[t2], 3u) < 3u) push(P, p, t2, filed);
    } else { const uint32_t pk = 1u + par; if (amax(P.lock[t2], pk) < pk) park(P, t2, par); }
  }
  void wake(PlanState& P, uint32_t p, uint32_t t2, uint32_t v, float thr, uint32_t par, uint32_t* filed = nullptr)
  {
    amin(P.pend[t2], v);
    route(P, p, t2, v, thr, par, filed);
  }

  void run(uint32_t n, uint32_t home)
  {
    M.enter(me);
    const HostTiles& T = M.T;
    uint32_t p_cur = home % n;                                       // the plan this workgroup serves
    for (;;) {
      // ---- the next ticket of the plan this workgroup serves; when that plan is finished: the next unfinished one
      uint32_t e = kNone - 1u;                                       // kTicketExit
      for (bool again = true; again;) {
        again = false;
        PlanState& Pc = M.plans[p_cur];
        const uint32_t i = add(Pc.head, 1u);
        if (i < Pc.ring.size()) {
          for (;;) {
            e = ld(Pc.ring[i]);
            if (e != kNone) break;
            e = kNone - 1u;
            if (ld(M.abort) || ld(M.done_plans) >= n) break;
            if (ld(Pc.done_flag)) {
              uint32_t q = p_cur;
              for (uint32_t k = 1; k < n; ++k) { const uint32_t c2 = (p_cur + k) % n; if (!ld(M.plans[c2].done_flag)) { q = c2; break; } }
              if (q != p_cur) { p_cur = q; again = true; }
              break;
            }
            e = kNone;
          }
        } else st(M.abort, 5u);
      }
      if (e == kNone - 1u) break;
      const uint32_t p = p_cur, t = e;
      PlanState& P = M.plans[p];
      const uint32_t v = xchg(P.pend[t], kInf);
      const float dt = u2f(ld(P.dist[P.target]));
      const float bound = (float)((double)dt + std::max(M.offset, 0.0));
      const float thr = u2f(ld(P.thr));
      const uint32_t par = ld(P.par);
      bool solve = false;
      if (v != kInf) {
        if (u2f(v) > bound) {
          if (!(u2f(ld(P.tlast[t])) > -inf_f())) st(P.tlast[t], f2u(-3.0e38f));
          ++M.drops;
        } else solve = true;
      }
      uint32_t sweep = 0, filed = 0;
      if (solve) {
        add(P.work, kWakeSlots + 1u);                                // reservation for the tickets this solve may file
        if (P.in_solve[t]) ++M.violations;                           // two solvers on one tile
        P.in_solve[t] = 1;
        if (++M.solves_now > M.max_solves) M.max_solves = M.solves_now;
        // ---- solve (k_tile_round's: queue of sources below thr and the bound, min on the float bits)
        const uint32_t v0 = T.vptr[t], nv = T.vptr[t + 1] - v0;
        const uint32_t h0 = T.hptr[t], nh = T.hptr[t + 1] - h0;
        const uint32_t e0 = T.eptr[t], r0 = T.rptr[t];
        const float tl = u2f(ld(P.tlast[t]));
        std::vector<uint32_t> ldu(nv + nh), orig(nv), lh0(nh);
        std::vector<uint32_t> q, qn;
        std::vector<uint8_t> queued(nv, 0);
        for (uint32_t k = 0; k < nv; ++k) {
          orig[k] = ld(P.dist[T.verts[v0 + k]]); ldu[k] = orig[k];
          const float d = u2f(orig[k]);
          if (d < thr && d <= bound && !(d < tl)) q.push_back(k);
        }
        for (uint32_t k = 0; k < nh; ++k) {
          const uint32_t b = ld(P.dist[T.halo_verts[h0 + k]]); ldu[nv + k] = b; lh0[k] = b;
          const float d = u2f(b);
          if (d < thr && d <= bound) q.push_back(nv + k);
        }
        while (!q.empty()) {
          qn.clear(); std::fill(queued.begin(), queued.end(), 0);
          for (uint32_t x : q) {
            const float di = u2f(ldu[x]);
            if (!(di < thr) || !(di <= bound)) continue;
            for (uint32_t ed = T.rowptr[r0 + x]; ed < T.rowptr[r0 + x + 1]; ++ed) {
              const uint32_t c = T.col[e0 + ed];
              const uint32_t nd = f2u(di + M.tw[e0 + ed]);
              if (nd < ldu[c]) { ldu[c] = nd; if (c < nv && !queued[c]) { queued[c] = 1; qn.push_back(c); } }
            }
          }
          q.swap(qn); ++sweep;
        }
        // ---- publish: distances, then (after the drain) the wake-ups
        uint32_t own_left = kInf;
        for (uint32_t k = 0; k < nv; ++k) {
          const uint32_t db = ldu[k];
          if (db != orig[k]) st(P.dist[T.verts[v0 + k]], db);
          const float d = u2f(db);
          if (!(d < thr) && d <= bound) own_left = std::min(own_left, db);
        }
        uint32_t wt[kWakeSlots], wv[kWakeSlots]; bool over = false;
        for (uint32_t sl = 0; sl < kWakeSlots; ++sl) { wt[sl] = kNone; wv[sl] = kInf; }
        auto collect = [&](uint32_t t2, uint32_t val) {
          uint32_t slot = (t2 * 0x9E3779B1u) >> 27;
          for (uint32_t probe = 0; probe < kWakeSlots; ++probe, slot = (slot + 1u) & (kWakeSlots - 1u))
            if (wt[slot] == kNone || wt[slot] == t2) { wt[slot] = t2; wv[slot] = std::min(wv[slot], val); return; }
          over = true;
        };
        for (uint32_t k = 0; k < nh; ++k) if (ldu[nv + k] < lh0[k]) collect(T.halo_tile[h0 + k], ldu[nv + k]);
        if (own_left != kInf) collect(t, own_left);
        for (uint32_t sl = 0; sl < kWakeSlots; ++sl) if (wt[sl] != kNone) wake(P, p, wt[sl], wv[sl], thr, par, &filed);
        if (over) {
          for (uint32_t k = 0; k < nh; ++k) if (ldu[nv + k] < lh0[k]) wake(P, p, T.halo_tile[h0 + k], ldu[nv + k], thr, par);
          if (own_left != kInf) wake(P, p, t, own_left, thr, par);
        }
        st(P.tlast[t], f2u(thr));
        add(P.acts, 1u); add(P.sweeps, sweep);
        P.in_solve[t] = 0; --M.solves_now;
      }
      // ---- retire the ticket
      P.ticketed[t] = 0;
      xchg(P.lock[t], 0u);
      if (M.mutate != 2u) { const uint32_t v2 = ld(P.pend[t]); if (v2 != kInf) route(P, p, t, v2, thr, par, solve ? &filed : nullptr); }
      const uint32_t back = 1u + (solve ? kWakeSlots + 1u - filed : 0u);
      bool advance = sub(P.work, back) == back;
      // ---- advance the band (exclusive until its first ticket is filed)
      uint32_t par_c = par;
      while (advance) {
        if (M.solves_now && P.work == 0) { for (size_t k = 0; k < P.in_solve.size(); ++k) if (P.in_solve[k]) ++M.violations; }   // somebody still solves a tile of the plan
        const uint32_t pk_old = 1u + par_c, par2 = par_c ^ 1u, pk_new = 1u + par2;
        const uint32_t cnt = std::min<uint32_t>(ld(P.nparked[par_c]), (uint32_t)P.parked[par_c].size());
        const float bound2 = (float)((double)u2f(ld(P.dist[P.target])) + std::max(M.offset, 0.0));
        uint32_t mn = kInf; bool beyond = false;
        for (uint32_t k = 0; k < cnt; ++k) {
          const uint32_t t2 = ld(P.parked[par_c][k]);
          if (ld(P.lock[t2]) != pk_old) continue;
          const uint32_t pv = ld(P.pend[t2]);
          if (u2f(pv) > bound2) beyond = true; else mn = std::min(mn, pv);
        }
        if (mn == kInf && !beyond) { plan_finish(P); break; }
        float thr2 = inf_f();
        if (mn != kInf && M.band > 0.f && M.band < inf_f()) { const float m = u2f(mn); thr2 = m + M.band; if (!(thr2 > m)) thr2 = next_up(m); }
        uint32_t filed2 = 0;
        st(P.work, 1u + cnt);                                        // own hold + a reservation for every ticket the pass may file
        st(P.thr, f2u(thr2)); st(P.par, par2); st(P.nparked[par2], 0u);
        add(P.epochs, 1u);
        for (uint32_t k = 0; k < cnt; ++k) {
          const uint32_t t2 = ld(P.parked[par_c][k]);
          if (ld(P.lock[t2]) != pk_old) continue;
          const uint32_t pv = ld(P.pend[t2]);
          if (u2f(pv) < thr2 || u2f(pv) > bound2) { if (amax(P.lock[t2], 3u) < 3u) push(P, p, t2, &filed2); }
          else if (cas(P.lock[t2], pk_old, pk_new)) park(P, t2, par2);
        }
        { const uint32_t back2 = 1u + cnt - filed2; advance = sub(P.work, back2) == back2; }
        par_c = par2;
      }
    }
    M.leave(me);
  }
};
}  // namespace

extern "C" {
// stats_out: [0] activations, [1] sweeps, [2] band advances, [3] tickets retired beyond the bound, [4] plan finishes, [5] scheduling
// points, [6] most concurrent solves, [7] invariant violations, [8] abort code, [9] tickets filed, [10] -, [11] tiles
uint32_t asm_run(uint32_t V, uint32_t F, uint32_t E, const uint32_t* face_vtx, const uint32_t* edge_vtx, const float* edge_weights,
                 const float* vertex_costs, const uint8_t* invalid, const float* xyz, uint32_t tile_size, uint32_t n, const uint32_t* seeds,
                 const uint32_t* targets, double offset, double cost_limit, float band, uint32_t n_wg, uint32_t sched_seed, uint64_t budget,
                 uint32_t mutate, uint32_t ring_cap, float* dist_out, uint64_t* stats_out)
{
  Model M;
  HostTopology topo = build_topology(V, F, E, face_vtx, edge_vtx);
  std::vector<Nbr> nbr; std::vector<Corner> crn; std::vector<uint8_t> blocked;
  materialize_host(topo, edge_weights, vertex_costs, invalid, cost_limit, nbr, crn, blocked);
  M.T = build_tiles(topo, xyz, tile_size);
  M.tw.resize(M.T.col.size());
  for (size_t i = 0; i < M.tw.size(); ++i) M.tw[i] = (M.T.src[i] == kNone) ? inf_f() : nbr[M.T.src[i]].w;   // k_tile_weights
  M.offset = offset; M.band = band; M.rng.seed(sched_seed); M.budget = budget; M.mutate = mutate;
  M.plans.resize(n);
  for (uint32_t p = 0; p < n; ++p) {
    PlanState& P = M.plans[p];
    P.dist.assign(V, kInf); P.pend.assign(M.T.ntiles, kInf); P.lock.assign(M.T.ntiles, 0u); P.tlast.assign(M.T.ntiles, f2u(-inf_f()));
    P.in_solve.assign(M.T.ntiles, 0); P.ticketed.assign(M.T.ntiles, 0);
    P.seed = seeds[p]; P.target = targets[p];
    const uint32_t st = M.T.vert_tile[P.seed];
    P.dist[P.seed] = 0u; P.pend[st] = 0u; P.lock[st] = 3u; P.ticketed[st] = 1; P.work = 1u;   // k_init, k_tile_init, k_async_init
    P.thr = f2u((band > 0.f && band < inf_f()) ? band : inf_f()); P.par = 0;
    P.parked[0].assign(4 * (size_t)M.T.ntiles, 0u); P.parked[1].assign(4 * (size_t)M.T.ntiles, 0u);
  }
  for (uint32_t p = 0; p < n; ++p) {
    PlanState& P = M.plans[p];
    P.ring.assign((size_t)std::max<uint32_t>(ring_cap, 2u), kNone);
    P.ring[0] = M.T.vert_tile[P.seed]; P.head = 0; P.tail = 1;
  }
  M.alive.assign(n_wg, 1); M.sleep_until.assign(n_wg, 0); M.turn = 0;
  std::vector<std::thread> th;
  std::vector<Wg*> wgs;
  for (uint32_t w = 0; w < n_wg; ++w) wgs.push_back(new Wg(M, (int)w, sched_seed * 7919u + w));
  for (uint32_t w = 0; w < n_wg; ++w) th.emplace_back([&, w] { wgs[w]->run(n, w); });   // (workgroup w starts on plan w mod n)
  for (auto& t : th) t.join();
  for (auto* w : wgs) delete w;
  uint64_t acts = 0, sweeps = 0, fins = 0;
  for (uint32_t p = 0; p < n; ++p) {
    const PlanState& P = M.plans[p];
    acts += P.acts; sweeps += P.sweeps; fins += P.finishes;
    if (P.finishes != 1u && !M.abort) ++M.violations;
    for (uint32_t v = 0; v < V; ++v) dist_out[(size_t)p * V + v] = u2f(P.dist[v]);
  }
  uint64_t epochs = 0; for (uint32_t p = 0; p < n; ++p) epochs += M.plans[p].epochs;
  stats_out[0] = acts; stats_out[1] = sweeps; stats_out[2] = epochs; stats_out[3] = M.drops; stats_out[4] = fins; stats_out[5] = M.yields;
  stats_out[6] = M.max_solves; stats_out[7] = M.violations; uint64_t tickets = 0; for (uint32_t p = 0; p < n; ++p) tickets += M.plans[p].tail;
  stats_out[8] = M.abort; stats_out[9] = tickets; stats_out[10] = 0;
  stats_out[11] = M.T.ntiles;
  return M.abort ? 1u : 0u;
}
}
