// schedule_model.cpp -- CPU model of the device band/gather schedule.
//
// TEST INFRASTRUCTURE ONLY (lives under oracle/): runs the very same per-vertex rules and band
// controller the HIP kernels run (mesh_navigation_amd/csrc/mnav_eval.h), serially on the host, so
// that tests without a GPU can check the *schedule* against the sequential oracle
// (mnav_oracle.c).  It is never linked into, loaded by, or reachable from the product library.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../mesh_navigation_amd/csrc/mnav_build.h"
#include "../mesh_navigation_amd/csrc/mnav_eval.h"
#include "../mesh_navigation_amd/csrc/mnav_walk.h"

using namespace mnav;

namespace {
struct HostOps {
  Plan* P; Cnt* cnt; uint32_t* next; uint32_t stamp_val;
  void push(uint32_t v)
  {
    if (P->stamp[v] == stamp_val) return;
    P->stamp[v] = stamp_val;
    const uint32_t i = cnt->n_next++;
    if (i < P->cap) next[i] = v;
  }
  void push_dirty(uint32_t v) { P->dirty[v] = stamp_val; push(v); }
  void note_changed() { cnt->changed++; }
  void note_cut(float t) { const uint32_t b = f2u(t); if (b < cnt->minchg) cnt->minchg = b; }
  void note_min(float t) { const uint32_t b = f2u(t); if (b < cnt->minkey) cnt->minkey = b; }
  void note_eval() { cnt->evals++; }
  // waiting list of the current epoch (mnav_eval.h Plan.wlist): dedup by epoch id, position = entries before this step + this step's
  uint32_t* wcur; uint32_t wbase; uint32_t epoch;
  void park(uint32_t v, float t)
  {
    note_min(t);
    if (P->wstamp[v] == epoch) return;
    P->wstamp[v] = epoch;
    const uint32_t i = wbase + cnt->n_wait++;
    if (i < P->cap) wcur[i] = v;
  }
};
}  // namespace

// the step loop of the model (controller + one pass over the work list per step) and the model of k_cvp_verify
static uint32_t drive(Plan& P, uint32_t planner, int order, Ctl* ctl, Cnt* cnt, std::vector<PopKey>& tkey,
                      const std::vector<uint8_t>& blocked, uint64_t* stats_out, float* goal_dist_out)
{
  const uint32_t V = P.V;
  float* dist = P.dist; uint32_t* pred = P.pred; float* dirn = P.dirn; uint32_t* cutf = P.cutf;
  std::vector<uint32_t> dirty_dummy;
  uint32_t* dirty = P.dirty;
  uint64_t evals = 0, rng = 88172645463325252ull;
  const uint32_t trace_v = getenv("SM_TRACE_V") ? (uint32_t)atoi(getenv("SM_TRACE_V")) : kNone;   // debugging aid
  int j = 0;
  Ctl cur;
  std::vector<uint32_t> perm;
  const int sm_debug = getenv("SM_DEBUG") ? atoi(getenv("SM_DEBUG")) : -1;   // debugging aids, read once
  const int sm_cycle = getenv("SM_CYCLE") ? atoi(getenv("SM_CYCLE")) : -1;
  for (;; ++j) {
    const Ctl& prev = ctl[(j + 1) & 1];
    const Cnt& cprev = cnt[(j + 2) % 3];
    cur = controller(P, prev, cprev);
    ctl[j & 1] = cur;
    Cnt& cnext = cnt[(j + 1) % 3];
    cnext.n_next = 0; cnext.changed = 0; cnext.minkey = f2u(inf_f()); cnext.evals = 0; cnext.n_wait = 0; cnext.minchg = f2u(inf_f());
    if (sm_debug >= 0 && j >= sm_debug && j < sm_debug + 140)
      fprintf(stderr, "step %d n=%u thr=%.9g fixed=%.9g width=%.3g repair=%u band_new=%u band_steps=%u | prev changed=%u minkey=%.9g\n", j, cur.n, cur.thr, cur.thr_fixed, cur.width, cur.repair, cur.band_new, cur.band_steps, cprev.changed, u2f(cprev.minkey));
    if (cur.done) break;
    if (cur.exact_wanted) {
      // the exact band routine (mnav_eval.h), what the host runs between two chunks of steps on the device (k_exact_band)
      Ctl x = exact_ctl(cur);
      std::vector<uint32_t> A, C;
      auto add = [&](std::vector<uint32_t>& L, uint32_t v) { if (v != kNone && P.stamp[v] != kExactStamp - 1u - (&L == &C ? 0u : 1u)) { P.stamp[v] = kExactStamp - 1u - (&L == &C ? 0u : 1u); L.push_back(v); } };
      for (uint32_t v = 0; v < V; ++v) {                              // 1. the band's vertices and their corner neighbours, unless settled
        const float t = key_time(tkey[v]);
        if (is_seed(P, v) || !(t >= cur.thr_fixed && t < cur.thr)) continue;
        add(A, v);
        for (uint32_t i = P.crn_ptr[v]; i < P.crn_ptr[v + 1]; ++i) { const Corner k = P.crn[i]; if (k.v1 == kNone) continue; add(A, k.v1); add(A, k.v2); }
      }
      for (uint32_t v : A) {                                          //    forget what was derived from provisional supports
        if (is_seed(P, v) || blocked[v] || key_time(tkey[v]) < cur.thr_fixed) continue;
        dist[v] = inf_f(); pred[v] = v; tkey[v] = key_inf(); dirn[v] = 0.0f; cutf[v] = kNone; if (P.keyd) P.keyd[v] = inf_f();
      }
      for (uint32_t v : A) if (exact_entry(P, x, v)) add(C, v);
      uint64_t pops = 0;
      for (;;) {                                                      // 2. one pop at a time
        uint32_t best = kNone;
        for (uint32_t u : C) if (exact_better(P, x, u, best)) best = u;
        if (best == kNone) break;
        x.bound_v = best; ++pops;
        for (uint32_t i = P.crn_ptr[best]; i < P.crn_ptr[best + 1]; ++i) {
          const Corner k = P.crn[i];
          if (k.v1 == kNone) continue;
          if (exact_entry(P, x, k.v1)) add(C, k.v1);
          if (exact_entry(P, x, k.v2)) add(C, k.v2);
        }
      }
      if (getenv("SM_EXACT_DEBUG")) fprintf(stderr, "exact band at step %d: [%.9g, %.9g), %zu reset, %zu candidates, %llu pops\n", j, cur.thr_fixed, cur.thr, A.size(), C.size(), (unsigned long long)pops);
      ctl[j & 1] = exact_done_ctl(cur);                               // 3. back to the band steps
      continue;
    }
    Cnt& cc = cnt[j % 3];
    HostOps ops{ &P, &cc, P.list[(j + 1) & 1], (uint32_t)(j + 1), P.wlist[cur.wsel], cur.wbase, cur.epoch };
    const uint32_t* list = P.list[j & 1];
    std::vector<uint32_t> ent(list, list + cur.n);                   // work list, plus the previous band's waiting list in an epoch step
    if (cur.wread) ent.insert(ent.end(), P.wlist[cur.wsel ^ 1u].p, P.wlist[cur.wsel ^ 1u].p + cur.wread);
    const uint32_t nent = (uint32_t)ent.size();
    list = ent.data();
    if (cur.repair == 3) {                                           // band cut: park what lies at or above the cut, keep the work list
      for (uint32_t i = 0; i < cur.n; ++i) ops.push_dirty(P.list[j & 1][i]);
      for (uint32_t v = 0; v < V; ++v) {
        if (planner == kPlannerCvp) process_cut<kPlannerCvp>(P, cur, v, ops);
        else process_cut<kPlannerDijkstra>(P, cur, v, ops);
      }
    } else if (cur.repair == 2) {
      for (uint32_t v = 0; v < V; ++v) {
        if (planner == kPlannerCvp) process_rebuild<kPlannerCvp>(P, cur, v, ops);
        else process_rebuild<kPlannerDijkstra>(P, cur, v, ops);
      }
    } else if (cur.repair) {
      for (uint32_t v = 0; v < V; ++v) {
        if (planner == kPlannerCvp) process_repair<kPlannerCvp>(P, cur, v, ops);
        else process_repair<kPlannerDijkstra>(P, cur, v, ops);
      }
    } else {
      perm.resize(nent);
      for (uint32_t i = 0; i < nent; ++i) perm[i] = i;
      if (order == 1) std::reverse(perm.begin(), perm.end());
      if (order == 2)
        for (uint32_t i = nent; i > 1; --i) {
          rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
          std::swap(perm[i - 1], perm[rng % i]);
        }
      if (order == 3 || order >= 4) {
        // snapshot of everything the rules read
        std::vector<float> sd(dist, dist + V), sdir; std::vector<uint32_t> sp(pred, pred + V), scut;
        std::vector<PopKey> sk(tkey); std::vector<float> skd; if (P.keyd) skd.assign(P.keyd.p, P.keyd.p + V);
        if (planner == kPlannerCvp) { sdir.assign(dirn, dirn + V); scut.assign(cutf, cutf + V); }
        Plan R = P;
        R.dist = sd.data(); R.pred = sp.data(); R.tkey = sk.data(); if (P.keyd) R.keyd = skd.data();
        if (planner == kPlannerCvp) { R.dirn = sdir.data(); R.cutf = scut.data(); }
        // order >= 4: a seeded mixture -- every entry reads either the snapshot (a neighbour's evaluation has not landed yet) or
        // the state in place (it has), in shuffled order: closer to the device's racy evaluation than pure Jacobi
        if (order >= 4 && j == 0) rng ^= 0x9E3779B97F4A7C15ull * (uint64_t)order;
        if (order >= 4)
          for (uint32_t i = nent; i > 1; --i) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; std::swap(perm[i - 1], perm[rng % i]); }
        for (uint32_t i = 0; i < nent; ++i) {
          bool snap = true;
          if (order >= 4) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; snap = (rng >> 33) & 1u; }
          const uint32_t vv = list[order >= 4 ? perm[i] : i];
          if (planner == kPlannerCvp) process_entry_rw<kPlannerCvp>(snap ? R : P, P, cur, vv, ops);
          else process_entry_rw<kPlannerDijkstra>(snap ? R : P, P, cur, vv, ops);
        }
      } else
      for (uint32_t i = 0; i < nent; ++i) {
        const uint32_t vv = list[perm[i]];
        const bool tr = trace_v == vv;
        const float bd = dist[vv];
        if (planner == kPlannerCvp) process_entry<kPlannerCvp>(P, cur, vv, ops);
        else process_entry<kPlannerDijkstra>(P, cur, vv, ops);
        if (tr) {
          fprintf(stderr, "step %d thr [%.6f, %.6f) band_new %u v %u: d %.7f -> %.7f key t0 %.7f up %.0f lvl %u dirty %u\n", j, cur.thr_fixed, cur.thr, cur.band_new, vv, bd, dist[vv], key_time(tkey[vv]), (double)tkey[vv].up, (unsigned)tkey[vv].lvl, dirty[vv]);
          for (uint32_t ci = P.crn_ptr[vv]; ci < P.crn_ptr[vv + 1]; ++ci) {
            const Corner k = P.crn[ci];
            const Fire f = corner_fire(P, cur, k);
            fprintf(stderr, "    corner face %u v1 %u (d %.7f t0 %.7f up %.0f lvl %u) v2 %u (d %.7f t0 %.7f up %.0f lvl %u) trig %u\n", k.face, k.v1, dist[k.v1], key_time(tkey[k.v1]), (double)tkey[k.v1].up, (unsigned)tkey[k.v1].lvl, k.v2, dist[k.v2], key_time(tkey[k.v2]), (double)tkey[k.v2].up, (unsigned)tkey[k.v2].lvl, f.trig);
          }
        }
      }
    }
    evals += cc.evals;
    if (sm_cycle >= 0 && j >= sm_cycle && j < sm_cycle + 8) {   // debugging aid
      static std::vector<float> pd; static std::vector<PopKey> pk;
      if (pd.size() == V) {
        fprintf(stderr, "== step %d thr [%.7f, %.7f) n=%u changed=%u\n", j, cur.thr_fixed, cur.thr, cur.n, cc.changed);
        for (uint32_t v = 0; v < V; ++v)
          if (f2u(pd[v]) != f2u(dist[v]) || pk[v] != tkey[v])
            fprintf(stderr, "   v %u: d %.7f -> %.7f  key (t0 %.7f root %u up %d lvl %u) -> (t0 %.7f root %u up %d lvl %u) pred %u\n", v, pd[v], dist[v],
                    key_time(pk[v]), pair_id(pk[v].hi), (int)pk[v].up, pk[v].lvl, key_time(tkey[v]), pair_id(tkey[v].hi), (int)tkey[v].up, tkey[v].lvl, pred[v]);
      }
      pd.assign(dist, dist + V); pk = tkey;
    }
  }
  // model of k_cvp_verify: one more evaluation of every vertex on the converged state must reproduce it, and no
  // walk over the cascade tree may hit its bound there (flags raised during the iteration are transient)
  uint64_t verify_bad = 0, verify_flags = 0, verify_sweeps = 0;
  if (planner == kPlannerCvp && cur.done && !cur.overflow) {
    for (int sweep = 0; sweep <= kVerifySweeps; ++sweep) {          // like run_plans: fix, sweep again, until clean
      cnt[3].n_next = 0; cnt[3].changed = 0;
      verify_bad = 0;
      for (uint32_t v = 0; v < V; ++v) {
        if (is_seed(P, v) || blocked[v]) continue;
        const Eval e = eval_cvp(P, cur, v);
        if (getenv("SM_VERIFY_DEBUG") && !verify_entry(P, cur, v, e, false))
          fprintf(stderr, "verify sweep %d: v %u stored d %.9g key(t0 %.9g root %u up %d lvl %u) | eval d %.9g key(t0 %.9g root %u up %d lvl %u)\n", sweep, v, dist[v],
                  key_time(tkey[v]), pair_id(tkey[v].hi), (int)tkey[v].up, tkey[v].lvl, e.d, key_time(e.key), pair_id(e.key.hi), (int)e.key.up, e.key.lvl);
        if (!verify_entry(P, cur, v, e, sweep < kVerifySweeps)) ++verify_bad;
      }
      verify_flags = cnt[3].n_next;
      if (verify_bad == 0) break;
      ++verify_sweeps;
    }
  }
  // model of k_cvp_seed_ring: the second fire events of the faces around the seed face (mnav_eval.h, corner_fire_second)
  if (planner == kPlannerCvp && cur.done && !cur.overflow && verify_bad == 0)
    for (int q = 0; q < 3; ++q) {
      const uint32_t sq = P.seed[q];
      if (sq == kNone || sq >= V) continue;
      for (uint32_t i = P.crn_ptr[sq]; i < P.crn_ptr[sq + 1]; ++i) { seed_ring_fix(P, cur, P.crn[i].v1); seed_ring_fix(P, cur, P.crn[i].v2); }
    }
  if (stats_out) { stats_out[0] = (uint64_t)j; stats_out[1] = cur.bands; stats_out[2] = evals; stats_out[3] = cur.armed; stats_out[4] = cur.shrinks;
                   stats_out[5] = verify_bad; stats_out[6] = verify_flags; stats_out[7] = verify_sweeps | ((uint64_t)cur.cuts << 32); }
  if (goal_dist_out) *goal_dist_out = cur.goal_dist;
  return cur.overflow ? kInternalError : kSuccess;   // overflow == 2: step cap hit (no convergence)
}

extern "C" {

// order: 0 = list order, 1 = reversed list order, 2 = pseudo-random permutation per step,
// 3 = Jacobi: every entry of a step reads the state as it was at the start of the step.
// seeds: Dijkstra -> seed_v[0]; CVP -> 3 seed-face vertices with seed_d[] Euclidean distances.
// stats_out[0]=steps, [1]=bands, [2]=evals, [3]=armed, goal_dist_out.
uint32_t sm_run(uint32_t planner, uint32_t V, uint32_t F, uint32_t E, const uint32_t* face_vtx,
                const uint32_t* edge_vtx, const float* edge_weights, const float* vertex_costs,
                const uint8_t* invalid, const uint32_t seed_v[3], const float seed_d[3],
                uint32_t seed_face, const uint32_t target_v[3], double offset, double cost_limit,
                float delta, int order, uint32_t max_steps, float* dist, uint32_t* pred, float* dirn, uint32_t* cutf,
                uint64_t* stats_out, float* goal_dist_out)
{
  HostTopology topo = build_topology(V, F, E, face_vtx, edge_vtx);
  std::vector<Nbr> nbr; std::vector<Corner> crn; std::vector<uint8_t> blocked;
  materialize_host(topo, edge_weights, vertex_costs, invalid, cost_limit, nbr, crn, blocked);

  std::vector<PopKey> tkey(V, key_inf());
  std::vector<uint32_t> stamp(V, 0), dirty(V, 0), l0(V), l1(V), w0(V), w1(V), wstamp(V, 0);
  Ctl ctl[2]; Cnt cnt[4];   // cnt[3]: sticky flags (mnav_eval.h kFlag*)
  cnt[3].n_next = 0; cnt[3].changed = 0;
  std::memset(ctl, 0, sizeof(ctl)); std::memset(cnt, 0, sizeof(cnt));

  Plan P{};
  P.planner = planner; P.V = V;
  P.row_ptr = topo.row_ptr.data(); P.nbr = nbr.data();
  P.crn_ptr = topo.crn_ptr.data(); P.crn = crn.data(); P.blocked = blocked.data();
  P.dist = dist; P.tkey = tkey.data();
  P.pred = pred; P.dirn = dirn; P.cutf = cutf; P.stamp = stamp.data(); P.dirty = dirty.data();
  P.list[0] = l0.data(); P.list[1] = l1.data(); P.cap = V;
  P.wlist[0] = w0.data(); P.wlist[1] = w1.data(); P.wstamp = wstamp.data();
  P.ctl = ctl; P.cnt = cnt;
  P.delta = delta; P.offset = offset; P.max_steps = max_steps ? max_steps : 100000000u;
  P.walk_max = getenv("MNAV_KEY_WALK_MAX") ? atoi(getenv("MNAV_KEY_WALK_MAX")) : kKeyWalkMax;
  P.descend_max = getenv("MNAV_DESCEND_WALK_MAX") ? atoi(getenv("MNAV_DESCEND_WALK_MAX")) : kDescendWalkMax;
  for (int k = 0; k < 3; ++k) { P.seed[k] = kNone; P.seed_expands[k] = 0; P.target[k] = kNone; P.target_expands[k] = 0; }

  for (uint32_t v = 0; v < V; ++v) { dist[v] = inf_f(); pred[v] = v; }
  if (planner == kPlannerCvp) for (uint32_t v = 0; v < V; ++v) { dirn[v] = 0.0f; cutf[v] = kNone; }

  // initial state
  float m0 = inf_f();
  const int ns = (planner == kPlannerCvp) ? 3 : 1;
  for (int k = 0; k < ns; ++k) {
    const uint32_t s = seed_v[k];
    P.seed[k] = s;
    const float d = (planner == kPlannerCvp) ? seed_d[k] : 0.0f;
    dist[s] = d; P.tkey[s] = make_key(d, s);
    if (planner == kPlannerCvp) {
      cutf[s] = seed_face;
      P.seed_expands[k] = !((double)vertex_costs[s] >= cost_limit) && !(invalid && invalid[s]);  // cvp :757,760
    } else {
      P.seed_expands[k] = 1;  // cost cut-off folded into the gather weights
    }
    m0 = fminf(m0, d);
  }
  for (int k = 0; k < ns; ++k) {
    P.target[k] = target_v[k];
    if (planner == kPlannerCvp)
      P.target_expands[k] = !((double)vertex_costs[target_v[k]] >= cost_limit) && !(invalid && invalid[target_v[k]]);
    else P.target_expands[k] = 1;
  }
  // initial list: neighbours of the seeds
  Cnt& c_init = cnt[2];   // "(0-1) mod 3"
  cnt[0].minkey = cnt[1].minkey = f2u(inf_f());                  // like k_seed
  cnt[0].minchg = cnt[1].minchg = cnt[2].minchg = f2u(inf_f());
  c_init.minkey = f2u(inf_f());
  {
    HostOps ops{ &P, &c_init, P.list[0], 0xFFFFFFFFu, nullptr, 0u, 0u };
    for (int k = 0; k < ns; ++k) {
      const uint32_t s = seed_v[k];
      if (planner == kPlannerCvp) {
        for (uint32_t i = P.crn_ptr[s]; i < P.crn_ptr[s + 1]; ++i) {
          if (P.crn[i].v1 == kNone) continue;
          ops.push(P.crn[i].v1); ops.push(P.crn[i].v2);
        }
      } else {
        for (uint32_t i = P.row_ptr[s]; i < P.row_ptr[s + 1]; ++i) ops.push(P.nbr[i].u);
      }
    }
  }
  c_init.changed = 1;
  Ctl& c0 = ctl[1];
  c0.it = -1; c0.n = 0; c0.thr_fixed = -inf_f(); c0.goal_dist = inf_f(); c0.arm_vertex = kNone;
  c0.thr = m0 + delta; if (!(c0.thr > m0)) c0.thr = next_up(m0);
  for (int k = 0; k < ns; ++k) {                                   // the first band holds every seed (like k_seed)
    const float d = (planner == kPlannerCvp) ? seed_d[k] : 0.0f;
    if (!(d < c0.thr)) c0.thr = next_up(d);
  }
  c0.band_new = 1; c0.width = delta; c0.wmin = inf_f(); c0.epoch = 1;

  return drive(P, planner, order, ctl, cnt, tkey, blocked, stats_out, goal_dist_out);
}

// Inflation wave (InflationLayer::waveCostInflation) through the same rules: lethal = sources, edge_dist = side
// lengths, `invalid` optional.  dist/keyd: V floats out.  The model of mnav_layer_inflation (mnav.hip).
uint32_t sm_run_inflation(uint32_t V, uint32_t F, uint32_t E, const uint32_t* face_vtx, const uint32_t* edge_vtx,
                          const float* edge_dist, const uint8_t* lethal, const uint8_t* invalid, float max_distance,
                          float delta, int order, uint32_t max_steps, float* dist, float* keyd, uint64_t* stats_out,
                          const float* xyz, float* vec_out, uint8_t* has_out)
{
  HostTopology topo = build_topology(V, F, E, face_vtx, edge_vtx);
  std::vector<Nbr> nbr; std::vector<Corner> crn; std::vector<uint8_t> blocked;
  std::vector<float> zero_cost(V, 0.0f);
  materialize_host(topo, edge_dist, zero_cost.data(), nullptr, INFINITY, nbr, crn, blocked);   // no face skipped, nothing blocked
  for (Corner& k : crn) k.face = corner_face_for_inflation(k.face);                           // like k_build_crn_infl
  std::vector<uint8_t> mask(V);
  for (uint32_t v = 0; v < V; ++v) {
    const bool l = lethal[v] != 0, inv = invalid && invalid[v];
    mask[v] = l ? (inv ? kInflSeedMute : kInflSeed) : (inv ? kInflMute : kInflFree);
  }
  std::vector<PopKey> tkey(V, key_inf());
  std::vector<uint32_t> stamp(V, 0), dirty(V, 0), l0(V), l1(V), w0(V), w1(V), wstamp(V, 0), pred(V), cutf(V, kNone);
  std::vector<float> dirn(V, 0.0f);
  Ctl ctl[2]; Cnt cnt[4];
  std::memset(ctl, 0, sizeof(ctl)); std::memset(cnt, 0, sizeof(cnt));
  Plan P{};
  P.planner = kPlannerCvp; P.V = V;
  P.row_ptr = topo.row_ptr.data(); P.nbr = nbr.data();
  P.crn_ptr = topo.crn_ptr.data(); P.crn = crn.data(); P.blocked = blocked.data();
  P.dist = dist; P.tkey = tkey.data(); P.keyd = keyd;
  P.pred = pred.data(); P.dirn = dirn.data(); P.cutf = cutf.data(); P.stamp = stamp.data(); P.dirty = dirty.data();
  P.list[0] = l0.data(); P.list[1] = l1.data(); P.cap = V;
  P.wlist[0] = w0.data(); P.wlist[1] = w1.data(); P.wstamp = wstamp.data();
  P.ctl = ctl; P.cnt = cnt;
  P.delta = delta; P.offset = 0.0; P.max_steps = max_steps ? max_steps : 100000000u;
  P.walk_max = getenv("MNAV_KEY_WALK_MAX") ? atoi(getenv("MNAV_KEY_WALK_MAX")) : kKeyWalkMax;
  P.descend_max = getenv("MNAV_DESCEND_WALK_MAX") ? atoi(getenv("MNAV_DESCEND_WALK_MAX")) : kDescendWalkMax;
  for (int k = 0; k < 3; ++k) { P.seed[k] = kNone; P.seed_expands[k] = 1; P.target[k] = kNone; P.target_expands[k] = 0; }
  P.seed_mask = mask.data(); P.infl_max = max_distance;
  for (uint32_t v = 0; v < V; ++v) { dist[v] = inf_f(); keyd[v] = inf_f(); pred[v] = v; }
  Cnt& c_init = cnt[2];
  cnt[0].minkey = cnt[1].minkey = f2u(inf_f());
  cnt[0].minchg = cnt[1].minchg = cnt[2].minchg = f2u(inf_f());
  c_init.minkey = f2u(inf_f());
  {
    HostOps ops{ &P, &c_init, P.list[0], 0xFFFFFFFFu, nullptr, 0u, 0u };
    for (uint32_t s = 0; s < V; ++s) {                              // like k_infl_seed
      if (!is_seed(P, s)) continue;
      dist[s] = 0.0f; tkey[s] = make_key(0.0f, s); keyd[s] = 0.0f;
      for (uint32_t i = P.crn_ptr[s]; i < P.crn_ptr[s + 1]; ++i) {
        if (!is_seed(P, P.crn[i].v1)) ops.push(P.crn[i].v1);
        if (!is_seed(P, P.crn[i].v2)) ops.push(P.crn[i].v2);
      }
    }
  }
  c_init.changed = 1;
  Ctl& c0 = ctl[1];                                                 // like k_infl_ctl
  c0.it = -1; c0.n = 0; c0.thr_fixed = -inf_f(); c0.goal_dist = inf_f(); c0.arm_vertex = kNone;
  c0.thr = delta; if (!(c0.thr > 0.0f)) c0.thr = next_up(0.0f);
  c0.band_new = 1; c0.width = delta; c0.wmin = inf_f(); c0.epoch = 1;
  const uint32_t code = drive(P, kPlannerCvp, order, ctl, cnt, tkey, blocked, stats_out, nullptr);
  // the layer's vector field from the converged wave (mnav_eval.h infl_accumulate / infl_assign), like k_infl_accum /
  // k_infl_assign do on the device
  if (code == kSuccess && xyz && vec_out && has_out) {
    std::vector<uint8_t> state(V, 0);
    for (uint32_t v = 0; v < V; ++v) {
      float o[3] = { 0.f, 0.f, 0.f };
      const int r = infl_accumulate(P, topo.crn_walk.data(), xyz, v, o);
      if (r < 0) return kInternalError;
      vec_out[3 * (size_t)v] = o[0]; vec_out[3 * (size_t)v + 1] = o[1]; vec_out[3 * (size_t)v + 2] = o[2];
      has_out[v] = r == 1;
      if (is_seed(P, v) || !(dist[v] < inf_f())) state[v] = r == 1 ? 1 : 2;
    }
    for (int sweep = 0; sweep < 100000; ++sweep) {
      uint32_t waiting = 0, moved = 0;
      for (uint32_t v = 0; v < V; ++v) {
        if (state[v]) continue;
        float o[3];
        const int r = infl_assign(P, vec_out, state.data(), v, o);
        if (r == 2) { ++waiting; continue; }
        if (r == 1) { vec_out[3 * (size_t)v] = o[0]; vec_out[3 * (size_t)v + 1] = o[1]; vec_out[3 * (size_t)v + 2] = o[2]; has_out[v] = 1; }
        state[v] = has_out[v] ? 1 : 2;
        ++moved;
      }
      if (!waiting) break;
      if (!moved) return kInternalError;                           // a dependency cycle: cannot happen on a verified state
    }
  }
  return code;
}

// the product's waveFrontUpdate arithmetic on plain numbers (mnav_eval.h infl_candidate): value offered to the free
// vertex, or NaN when it is not finite; *requeue = the :311 condition
// mnav_eval.h acosf_ref (the device's SteepnessLayer arithmetic) for the host-side check against libm
float sm_acosf_ref(float x) { return acosf_ref(x); }
// mnav_eval.h goal_cut / expanded_source: which sources the Dijkstra wave expands, as the finalize pass, the lazy path walks and
// the lazy vector entries of the device decide it (incl. negative goal_dist_offset); out[i] = 1 when vertex ids[i] with FINAL
// value d[i] is an expanded source of a plan whose robot vertex `target` has the final value dt
void sm_expanded_sources(float dt, double offset, uint32_t target, uint32_t n, const float* d, const uint32_t* ids, uint8_t* out, float* goal_cut_tie)
{
  const GoalCut g = goal_cut(dt, offset, target);
  for (uint32_t i = 0; i < n; ++i) out[i] = expanded_source(g, d[i], ids[i]) ? 1 : 0;
  if (goal_cut_tie) { goal_cut_tie[0] = g.goal; goal_cut_tie[1] = g.cut; goal_cut_tie[2] = (g.tie == kNone) ? -1.0f : (float)g.tie; }
}
float sm_cosf_ref(float x) { return cosf_ref(x); }
float sm_sinf_ref(float x) { return sinf_ref(x); }

// mnav_walk.h (what k_backtrack runs, one thread) on host arrays: the product's back-tracking arithmetic for the CPU
// parity test against mo_cvp_backtrack.  Returns the walk status; positions / faces in walk order (target first).
int sm_backtrack(uint32_t V, uint32_t F, const float* xyz, const uint32_t* faces, const uint32_t* vf_ptr, const uint32_t* vf, const float* vecmap,
                 const float* seed_pos, uint32_t seed_face, const float* target_pos, uint32_t target_face, double step_width, uint32_t cap,
                 const float* infl_dist, const float* infl_vec, const double* infl_cfg, int repulsive, float* pos_out, uint32_t* face_out, uint32_t* n_out)
{
  WalkMesh M{ xyz, faces, vf_ptr, vf, V, F };
  WalkField Fd{ vecmap, { faces[3 * (size_t)seed_face], faces[3 * (size_t)seed_face + 1], faces[3 * (size_t)seed_face + 2] } };
  WalkInflation L{ infl_dist, infl_vec, nullptr, infl_cfg ? infl_cfg[0] : 0.0, infl_cfg ? infl_cfg[1] : 0.0, infl_cfg ? infl_cfg[2] : 0.0, infl_cfg ? infl_cfg[3] : 0.0, repulsive };
  std::vector<uint32_t> list(kWalkListCap);
  return walk_backtrack(M, Fd, L, w3_load(seed_pos), seed_face, w3_load(target_pos), target_face, step_width, cap, pos_out, face_out, n_out, list.data());
}

float sm_infl_candidate(float u1, float u2, float a, float b, float c, float max_distance, int* requeue)
{
  const InflCand k = infl_candidate(u1, u2, a, b, c, max_distance);
  if (requeue) *requeue = k.requeue ? 1 : 0;
  return k.ok ? k.u3tmp : NAN;
}

}  // extern "C"
