// tbatch_model.cpp -- CPU work model of the tile-major, plan-vectorised SSSP engine (design study, not product code).
//
// Counts what the device engine would do on an N x N terrain for P plans (random goals, common robot vertex):
// iterations of the level-synchronous schedule, (tile, plan) activations, Gauss-Seidel sweeps per activation,
// ghost stores and wake-ups -- and checks the result against a heap Dijkstra with the same float32 adds.
//
//   g++ -O2 -std=c++17 -o /tmp/tbatch_model tools/tbatch_model.cpp && /tmp/tbatch_model N P T delta_mult order
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <queue>
#include <vector>

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static const float INF = INFINITY;

struct Mesh {
  uint32_t N, V;
  std::vector<float> xyz;
  std::vector<uint32_t> rp, nb; std::vector<float> w;   // CSR, symmetric
};

static uint64_t rng_state = 88172645463325252ull;
static double urand() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (rng_state >> 11) * (1.0 / 9007199254740992.0); }

static Mesh make_terrain(uint32_t N, float h)
{
  Mesh M; M.N = N; M.V = N * N; M.xyz.resize(3 * (size_t)M.V);
  double ph[5], ps[5];
  for (int k = 0; k < 5; ++k) { ph[k] = urand() * 6.283; ps[k] = urand() * 6.283; }
  for (uint32_t j = 0; j < N; ++j) for (uint32_t i = 0; i < N; ++i) {
    const double x = i * h + (urand() * 0.4 - 0.2) * h, y = j * h + (urand() * 0.4 - 0.2) * h;
    double z = 0;
    for (int k = 1; k <= 5; ++k) z += 2.0 / (1 << k) * sin(6.283185 * (1.0 / 50) * (1 << k) * x + ph[k - 1]) * cos(6.283185 * (1.0 / 50) * (1 << k) * y + ps[k - 1]);
    float* p = &M.xyz[3 * (size_t)(j * N + i)]; p[0] = (float)x; p[1] = (float)y; p[2] = (float)z;
  }
  std::vector<std::vector<uint32_t>> adj(M.V);
  auto add = [&](uint32_t a, uint32_t b) { adj[a].push_back(b); adj[b].push_back(a); };
  for (uint32_t j = 0; j < N; ++j) for (uint32_t i = 0; i < N; ++i) {
    const uint32_t v = j * N + i;
    if (i + 1 < N) add(v, v + 1);
    if (j + 1 < N) add(v, v + N);
    if (i + 1 < N && j + 1 < N) add(v, v + N + 1);
  }
  M.rp.assign(M.V + 1, 0);
  for (uint32_t v = 0; v < M.V; ++v) M.rp[v + 1] = M.rp[v] + (uint32_t)adj[v].size();
  M.nb.resize(M.rp[M.V]); M.w.resize(M.rp[M.V]);
  for (uint32_t v = 0; v < M.V; ++v) {
    std::sort(adj[v].begin(), adj[v].end());
    for (size_t k = 0; k < adj[v].size(); ++k) {
      const uint32_t u = adj[v][k];
      const float* a = &M.xyz[3 * (size_t)v]; const float* b = &M.xyz[3 * (size_t)u];
      const float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
      M.nb[M.rp[v] + k] = u; M.w[M.rp[v] + k] = sqrtf(dx * dx + dy * dy + dz * dz);
    }
  }
  return M;
}

static uint64_t spread2(uint64_t x)
{
  x &= 0xFFFFFFFFull; x = (x | (x << 16)) & 0x0000FFFF0000FFFFull; x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
  x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full; x = (x | (x << 2)) & 0x3333333333333333ull; x = (x | (x << 1)) & 0x5555555555555555ull;
  return x;
}

struct Tile {
  uint32_t v0, nv;                         // owned: positions [v0, v0+nv) of the tile order
  std::vector<uint32_t> ghost;             // global vertex ids, sorted by tile-order position
  // pull lists: for local target y (owned first, then ghosts): sources (local id, weight); ghost targets only list owned sources
  std::vector<uint32_t> ip; std::vector<uint16_t> isrc; std::vector<float> iw;
  std::vector<std::vector<uint16_t>> orders;   // sweep orders over the owned vertices
  // exports: for owned local u: (tile, ghost slot) pairs
  std::vector<uint32_t> xp; std::vector<uint32_t> xt, xs;
};

int main(int argc, char** argv)
{
  const uint32_t N = argc > 1 ? atoi(argv[1]) : 300;
  const uint32_t P = argc > 2 ? atoi(argv[2]) : 8;
  const uint32_t T = argc > 3 ? atoi(argv[3]) : 128;
  const float dmult = argc > 4 ? atof(argv[4]) : 1.0f;
  const int order_mode = argc > 5 ? atoi(argv[5]) : 0;     // 0: morton fwd/bwd, 1: 4 diagonal directions, 2: x/y 4 directions, 3: 8 directions
  const int jacobi_ghosts = argc > 6 ? atoi(argv[6]) : 1;  // 1: ghost stores become visible at the end of an iteration
  const double offset = 0.3;
  Mesh M = make_terrain(N, 0.1f);
  const uint32_t V = M.V;
  // tile order
  std::vector<uint32_t> ord(V), pos(V);
  {
    std::vector<uint64_t> key(V);
    for (uint32_t v = 0; v < V; ++v) key[v] = spread2(v % N) | (spread2(v / N) << 1);
    for (uint32_t v = 0; v < V; ++v) ord[v] = v;
    std::sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
    for (uint32_t i = 0; i < V; ++i) pos[ord[i]] = i;
  }
  const uint32_t nt = (V + T - 1) / T;
  std::vector<uint32_t> vt(V);
  for (uint32_t i = 0; i < V; ++i) vt[ord[i]] = i / T;
  std::vector<Tile> tiles(nt);
  size_t tot_ghost = 0, tot_in = 0;
  for (uint32_t t = 0; t < nt; ++t) {
    Tile& L = tiles[t];
    L.v0 = t * T; L.nv = std::min(T, V - L.v0);
    std::vector<uint32_t> gh;
    for (uint32_t i = 0; i < L.nv; ++i) {
      const uint32_t v = ord[L.v0 + i];
      for (uint32_t k = M.rp[v]; k < M.rp[v + 1]; ++k) if (vt[M.nb[k]] != t) gh.push_back(M.nb[k]);
    }
    std::sort(gh.begin(), gh.end(), [&](uint32_t a, uint32_t b) { return pos[a] < pos[b]; });
    gh.erase(std::unique(gh.begin(), gh.end()), gh.end());
    L.ghost = gh;
    tot_ghost += gh.size();
  }
  auto local_of = [&](const Tile& L, uint32_t t, uint32_t u) -> uint32_t {
    if (vt[u] == t) return pos[u] - L.v0;
    const auto it = std::lower_bound(L.ghost.begin(), L.ghost.end(), u, [&](uint32_t a, uint32_t b) { return pos[a] < pos[b]; });
    return L.nv + (uint32_t)(it - L.ghost.begin());
  };
  for (uint32_t t = 0; t < nt; ++t) {
    Tile& L = tiles[t];
    L.ip.push_back(0);
    for (uint32_t y = 0; y < L.nv + L.ghost.size(); ++y) {
      const uint32_t gy = y < L.nv ? ord[L.v0 + y] : L.ghost[y - L.nv];
      for (uint32_t k = M.rp[gy]; k < M.rp[gy + 1]; ++k) {
        const uint32_t u = M.nb[k];
        if (y >= L.nv && vt[u] != t) continue;          // ghost targets: owned sources only
        if (y < L.nv || vt[u] == t) { L.isrc.push_back((uint16_t)local_of(L, t, u)); L.iw.push_back(M.w[k]); }
      }
      L.ip.push_back((uint32_t)L.isrc.size());
    }
    tot_in += L.isrc.size();
    // exports
    L.xp.push_back(0);
    for (uint32_t i = 0; i < L.nv; ++i) {
      const uint32_t v = ord[L.v0 + i];
      std::vector<uint32_t> ts;
      for (uint32_t k = M.rp[v]; k < M.rp[v + 1]; ++k) if (vt[M.nb[k]] != t) ts.push_back(vt[M.nb[k]]);
      std::sort(ts.begin(), ts.end()); ts.erase(std::unique(ts.begin(), ts.end()), ts.end());
      for (uint32_t t2 : ts) { L.xt.push_back(t2); L.xs.push_back(local_of(tiles[t2], t2, v) - tiles[t2].nv); }
      L.xp.push_back((uint32_t)L.xt.size());
    }
    // sweep orders
    auto by = [&](double ax, double ay) {
      std::vector<uint16_t> o(L.nv);
      for (uint32_t i = 0; i < L.nv; ++i) o[i] = (uint16_t)i;
      std::stable_sort(o.begin(), o.end(), [&](uint16_t a, uint16_t b) {
        const float* pa = &M.xyz[3 * (size_t)ord[L.v0 + a]]; const float* pb = &M.xyz[3 * (size_t)ord[L.v0 + b]];
        return ax * pa[0] + ay * pa[1] < ax * pb[0] + ay * pb[1];
      });
      return o;
    };
    if (order_mode == 0) {
      std::vector<uint16_t> o(L.nv); for (uint32_t i = 0; i < L.nv; ++i) o[i] = (uint16_t)i;
      L.orders.push_back(o); std::reverse(o.begin(), o.end()); L.orders.push_back(o);
    } else if (order_mode == 1) {
      L.orders.push_back(by(1, 1)); L.orders.push_back(by(-1, 1)); L.orders.push_back(by(-1, -1)); L.orders.push_back(by(1, -1));
    } else if (order_mode == 2) {
      L.orders.push_back(by(1, 0.01)); L.orders.push_back(by(0.01, 1)); L.orders.push_back(by(-1, -0.01)); L.orders.push_back(by(-0.01, -1));
    } else {
      L.orders.push_back(by(1, 1)); L.orders.push_back(by(-1, -1)); L.orders.push_back(by(-1, 1)); L.orders.push_back(by(1, -1));
      L.orders.push_back(by(1, 0.01)); L.orders.push_back(by(-1, -0.01)); L.orders.push_back(by(0.01, 1)); L.orders.push_back(by(-0.01, -1));
    }
  }
  double mean_w = 0; for (float x : M.w) mean_w += x; mean_w /= M.w.size();
  const float delta = dmult * (float)(mean_w * sqrt((double)T));
  printf("V=%u tiles=%u T=%u ghosts/tile=%.1f in-edges/tile=%.1f slice words/plan=%zu (%.2fx V) delta=%.3f\n", V, nt, T, (double)tot_ghost / nt,
         (double)tot_in / nt, (size_t)V + tot_ghost, (double)(V + tot_ghost) / V, delta);
  std::vector<size_t> soff(nt + 1, 0);
  for (uint32_t t = 0; t < nt; ++t) soff[t + 1] = soff[t] + tiles[t].nv + tiles[t].ghost.size();
  const size_t S = soff[nt];

  const uint32_t robot = (uint32_t)(0.9 * (N - 1)) * N + (uint32_t)(0.9 * (N - 1));
  uint64_t tot_act = 0, tot_sweeps = 0, tot_gstores = 0, tot_wakes = 0, tot_settled = 0, tot_edge_ops = 0, tot_tiles_touched = 0, tot_changed_own = 0;
  std::vector<uint32_t> sweep_hist(64, 0);
  uint32_t max_iters = 0;
  std::vector<uint64_t> pairs_per_iter;
  for (uint32_t p = 0; p < P; ++p) {
    const uint32_t goal = (uint32_t)(urand() * V) % V;
    // reference: heap Dijkstra, float adds, early exit semantics irrelevant here (full field)
    std::vector<float> ref(V, INF);
    {
      typedef std::pair<float, uint32_t> QE;
      std::priority_queue<QE, std::vector<QE>, std::greater<QE>> pq;
      ref[goal] = 0; pq.push({ 0.f, goal });
      while (!pq.empty()) {
        auto [d, v] = pq.top(); pq.pop();
        if (d > ref[v]) continue;
        for (uint32_t k = M.rp[v]; k < M.rp[v + 1]; ++k) { const float nd = d + M.w[k]; if (nd < ref[M.nb[k]]) { ref[M.nb[k]] = nd; pq.push({ nd, M.nb[k] }); } }
      }
    }
    std::vector<float> D(S, INF);
    std::vector<uint32_t> pend(nt, f2u(INF)), acts(nt, 0);
    D[soff[vt[goal]] + (pos[goal] - tiles[vt[goal]].v0)] = 0.f;
    pend[vt[goal]] = f2u(0.f);
    { const Tile& L = tiles[vt[goal]]; const uint32_t i = pos[goal] - L.v0;   // the seed value is exported like any changed boundary value
      for (uint32_t k = L.xp[i]; k < L.xp[i + 1]; ++k) D[soff[L.xt[k]] + tiles[L.xt[k]].nv + L.xs[k]] = 0.f; }
    const size_t robot_slot = soff[vt[robot]] + (pos[robot] - tiles[vt[robot]].v0);
    uint32_t iters = 0;
    std::vector<std::pair<size_t, float>> deferred;
    std::vector<float> d;
    for (;; ++iters) {
      float m = INF;
      for (uint32_t t = 0; t < nt; ++t) m = std::min(m, u2f(pend[t]));
      const float bound = (float)((double)D[robot_slot] + offset);
      if (!(m < INF) || m > bound) break;
      float thr = m + delta; if (!(thr > m)) thr = INF;
      std::vector<uint32_t> ready;
      for (uint32_t t = 0; t < nt; ++t) { const float pv = u2f(pend[t]); if (pv < thr && pv <= bound) { ready.push_back(t); pend[t] = f2u(INF); } }
      if (pairs_per_iter.size() <= iters) pairs_per_iter.resize(iters + 1, 0);
      pairs_per_iter[iters] += ready.size();
      deferred.clear();
      if (getenv("TB_DEBUG") && iters > 200 && iters < 210) { printf("it %u m=%g bound=%g thr=%g ready=%zu:", iters, m, bound, thr, ready.size()); for (uint32_t t : ready) printf(" %u", t); printf("\n"); }
      for (uint32_t t : ready) {
        Tile& L = tiles[t];
        const uint32_t nv = L.nv, nh = (uint32_t)L.ghost.size();
        d.assign(D.begin() + soff[t], D.begin() + soff[t] + nv + nh);
        std::vector<float> orig(d.begin(), d.begin() + nv);
        if (!acts[t]) ++tot_tiles_touched;
        ++acts[t]; ++tot_act;
        uint32_t sweeps = 0;
        for (;; ) {
          bool changed = false;
          const std::vector<uint16_t>& o = L.orders[sweeps % L.orders.size()];
          for (uint32_t i = 0; i < nv; ++i) {
            const uint32_t y = o[i];
            float best = d[y];
            for (uint32_t k = L.ip[y]; k < L.ip[y + 1]; ++k) { const float c = d[L.isrc[k]] + L.iw[k]; if (c < best) best = c; }
            tot_edge_ops += L.ip[y + 1] - L.ip[y];
            if (best < d[y]) { d[y] = best; changed = true; }
          }
          ++sweeps;
          if (!changed) break;
        }
        tot_sweeps += sweeps; sweep_hist[std::min<uint32_t>(sweeps, 63)]++;
        // wake-ups: ghost candidates
        for (uint32_t hgi = 0; hgi < nh; ++hgi) {
          const uint32_t y = nv + hgi;
          float cand = INF;
          for (uint32_t k = L.ip[y]; k < L.ip[y + 1]; ++k) { const float c = d[L.isrc[k]] + L.iw[k]; if (c < cand) cand = c; }
          if (cand < d[y]) { const uint32_t t2 = vt[L.ghost[hgi]]; if (f2u(cand) < pend[t2]) pend[t2] = f2u(cand); ++tot_wakes; }
        }
        // write back + ghost stores
        for (uint32_t i = 0; i < nv; ++i) if (d[i] != orig[i]) {
          D[soff[t] + i] = d[i]; ++tot_changed_own;
          for (uint32_t k = L.xp[i]; k < L.xp[i + 1]; ++k) {
            const size_t slot = soff[L.xt[k]] + tiles[L.xt[k]].nv + L.xs[k];
            if (jacobi_ghosts) deferred.push_back({ slot, d[i] }); else D[slot] = d[i];
            ++tot_gstores;
          }
        }
      }
      for (auto& e : deferred) D[e.first] = e.second;
    }
    max_iters = std::max(max_iters, iters);
    // check: every vertex with ref <= goal_dist must match
    const float gd = (float)((double)ref[robot] + offset);
    uint32_t bad = 0, settled = 0;
    for (uint32_t v = 0; v < V; ++v) if (ref[v] <= gd) { ++settled; if (D[soff[vt[v]] + (pos[v] - tiles[vt[v]].v0)] != ref[v]) ++bad; }
    tot_settled += settled;
    if (bad || D[robot_slot] != ref[robot]) printf("plan %u: MISMATCH %u of %u (robot %g vs %g)\n", p, bad, settled, D[robot_slot], ref[robot]);
    if (p == 0) printf("plan 0: iters=%u settled=%u\n", iters, settled);
  }
  printf("plans=%u  iterations(max)=%u  settled/plan=%.0f\n", P, max_iters, (double)tot_settled / P);
  printf("activations/plan=%.0f  per touched tile=%.2f  touched tiles/plan=%.0f (settled/T=%.0f)\n", (double)tot_act / P, (double)tot_act / tot_tiles_touched,
         (double)tot_tiles_touched / P, (double)tot_settled / P / T);
  printf("sweeps/activation=%.2f  edge-ops/plan=%.3g (%.1f per settled vertex)\n", (double)tot_sweeps / tot_act, (double)tot_edge_ops / P, (double)tot_edge_ops / tot_settled);
  printf("ghost stores/activation=%.1f  changed owned/activation=%.1f  wakes/activation=%.1f\n", (double)tot_gstores / tot_act, (double)tot_changed_own / tot_act, (double)tot_wakes / tot_act);
  printf("sweep histogram:"); for (int i = 1; i < 40; ++i) if (sweep_hist[i]) printf(" %d:%.1f%%", i, 100.0 * sweep_hist[i] / tot_act); printf("\n");
  { double cdf = 0, emax = 0, prev = 0; for (int i = 1; i < 64; ++i) { cdf += (double)sweep_hist[i] / tot_act; const double c64 = pow(cdf, 64.0); emax += i * (c64 - prev); prev = c64; }
    printf("expected max sweeps over 64 lanes (iid): %.1f\n", emax); }
  uint64_t mx = 0, sum = 0; for (uint64_t x : pairs_per_iter) { mx = std::max(mx, x); sum += x; }
  printf("pairs/iteration/plan: mean %.1f max %.1f\n", (double)sum / pairs_per_iter.size() / P, (double)mx / P);
  return 0;
}
