// tb_model.cpp -- CPU model of the tile-batch SSSP engine (mesh_navigation_amd/csrc/mnav_tb.h).
//
// TEST INFRASTRUCTURE ONLY (lives under oracle/): it interprets the very same record streams the HIP kernel reads
// (mnav_tb_build.h), lane by lane, and runs the same level-synchronous schedule (k_tb_plan / k_tb_scan / k_tb_items /
// k_tb_solve_q) serially on the host, so that tests without a GPU can check the stream builder and the schedule against
// the sequential oracle (mnav_oracle.c).  Never linked into, loaded by, or reachable from the product library.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../mesh_navigation_amd/csrc/mnav_tb_build.h"

using namespace mnav;

namespace {
inline uint32_t fabs_bits_add(uint32_t v, uint32_t w) { return f2u(std::fabs(u2f(v)) + u2f(w)); }
}

extern "C" {

// jacobi: bit 0: 1 = every activation of an iteration sees the slices as they were when the iteration started (what concurrent
// waves may see at worst), 0 = in place, in item order.  bits 4-7: every sweep re-runs its
// last chunk that many times (what a quarter whose stream is shorter than its wave's does).
// stats_out: [0] iterations, [1] (tile, plan) activations, [2] sweeps summed over activations, [3] wake-ups, [4] max sweeps of
// one item, [5] tiles, [6] slots per plan, [7] items (waves of <= 64 plans), [8] sweep blocks visited, [9] sweep blocks evaluated;
// [2] counts sweeps per ITEM (the lanes of an item sweep in lockstep until no lane changes)
uint32_t tbm_run(uint32_t V, uint32_t F, uint32_t E, const uint32_t* face_vtx, const uint32_t* edge_vtx, const float* edge_weights,
                 const float* vertex_costs, const uint8_t* invalid, const float* xyz, uint32_t T, uint32_t n, const uint32_t* seeds,
                 const uint32_t* targets, double offset, double cost_limit, float band, int jacobi, float* dist_out, uint64_t* stats_out)
{
  HostTopology topo = build_topology(V, F, E, face_vtx, edge_vtx);
  std::vector<Nbr> nbr; std::vector<Corner> crn; std::vector<uint8_t> blocked;
  materialize_host(topo, edge_weights, vertex_costs, invalid, cost_limit, nbr, crn, blocked);
  const uint32_t rerun = (uint32_t)((jacobi >> 4) & 15);              // bits 4-7: every sweep re-runs its last chunk that many times (a quarter whose stream is shorter than its wave's)
  const bool vlayout = (jacobi & 2) != 0;                            // bit 1: the sweeps read the V layout of the streams (k_tbv_solve, mnav_tbv.h)
  jacobi &= 1;
  HostTb H;
  try { H = build_tb(topo, xyz, T); }
  catch (const std::exception& ex) { fprintf(stderr, "tbm_run: %s\n", ex.what()); return 64; }
  for (size_t i = 0; i < H.stream.size(); ++i) if (H.wsrc[i] != kNone) H.stream[i] = f2u(nbr[H.wsrc[i]].w);   // k_tb_weights
  for (size_t i = 0; i < H.vstream.size(); ++i) if (H.vwsrc[i] != kNone) H.vstream[i] = f2u(nbr[H.vwsrc[i]].w);
  const uint32_t NP = n, nt = H.ntiles;
  auto slot = [&](uint32_t t, uint32_t p, uint32_t i) { return (size_t)H.tiles[t].soff * NP + (size_t)p * H.tiles[t].sl + i; };
  std::vector<uint32_t> D((size_t)H.S * NP, kTbInfBits), Dsnap;
  std::vector<uint32_t> pend((size_t)nt * NP, kTbInfBits);
  std::vector<uint32_t> marr[2] = { std::vector<uint32_t>(NP, kTbInfBits), std::vector<uint32_t>(NP, kTbInfBits) };
  std::vector<std::pair<uint32_t, uint32_t>> cand[2];
  std::vector<float> thr(NP), bnd(NP);
  // k_tb_seed
  for (uint32_t p = 0; p < NP; ++p) {
    const uint32_t s = seeds[p], t = H.vert_tile[s], loc = H.vert_local[s];
    D[slot(t, p, loc)] = 0u;
    const TbTile& W = H.tiles[t];
    for (uint32_t k = 0; k < W.exp_n; ++k) { const TbExp& e = H.exps[W.exp_off + k]; if (e.u == loc * 256u) D[(size_t)e.soff * NP + (size_t)p * e.sl + e.off] = 0u; }
    pend[(size_t)t * NP + p] = 0u; marr[0][p] = 0u; cand[0].push_back({ t, p });
  }
  uint64_t iters = 0, acts = 0, sweeps_tot = 0, wakes = 0, max_sweeps = 0, items = 0, blocks_total = 0, blocks_eval = 0, stale_reads = 0;
  std::vector<std::vector<uint16_t>> bucket(nt);
  std::vector<std::vector<uint32_t>> ldsv(64, std::vector<uint32_t>(256));   // per lane: row -> value
  for (int par = 0;; par ^= 1) {
    if (cand[par].empty()) break;
    ++iters;
    // k_tb_plan
    for (uint32_t p = 0; p < NP; ++p) {
      const float m = u2f(marr[par][p]);
      marr[par ^ 1][p] = kTbInfBits;
      const uint32_t tg = targets[p];
      const float dt = u2f(D[slot(H.vert_tile[tg], p, H.vert_local[tg])]);
      const float bound = (float)((double)dt + offset);
      const bool done = !(m < INFINITY) || m > bound;
      float th = m + band; if (!(th > m)) th = next_up(m);
      thr[p] = done ? 0.0f : th; bnd[p] = bound;
    }
    // k_tb_scan
    cand[par ^ 1].clear();
    for (auto& e : cand[par]) {
      const size_t pi = (size_t)e.first * NP + e.second;
      const uint32_t pb = pend[pi]; const float pv = u2f(pb);
      if (pv > bnd[e.second]) pend[pi] = kTbInfBits;
      else if (pv < thr[e.second]) { pend[pi] = kTbInfBits; bucket[e.first].push_back((uint16_t)e.second); }
      else { cand[par ^ 1].push_back(e); marr[par ^ 1][e.second] = std::min(marr[par ^ 1][e.second], pb); }
    }
    if (jacobi) Dsnap = D;
    const std::vector<uint32_t>& Din = jacobi ? Dsnap : D;
    // k_tb_items + k_tb_solve_q: items (here of <= 64 plans; the kernel cuts them into quarters of 16, same values) of one tile, one plan per lane, the lanes in lockstep
    for (uint32_t t = 0; t < nt; ++t) {
      const TbTile& W = H.tiles[t];
      for (size_t start = 0; start < bucket[t].size(); start += 64) {
        const uint32_t cnt_l = (uint32_t)std::min<size_t>(64, bucket[t].size() - start);
        ++items; acts += cnt_l;
        for (uint32_t l = 0; l < cnt_l; ++l) {
          const uint32_t p = bucket[t][start + l];
          uint32_t* lds = ldsv[l].data();
          const size_t sl = slot(t, p, 0), gs = sl + T;
          for (uint32_t r = 0; r < T; ++r) lds[r] = Din[sl + r];
          // pre
          if (vlayout) {
            const uint32_t* M = &H.vtile[(size_t)kTbvTileWords * t];
            for (uint32_t c = 0; c < M[1]; ++c) {
              const uint32_t* C = &H.vstream[((size_t)M[0] + c) * kTbChunk];
              for (uint32_t j = 0; j < kTbBlocksPerChunk; ++j) {
                uint32_t blk[kTbBlock];
                for (uint32_t q = 0; q < kTbBlock; ++q) blk[q] = C[tb_sweep_index(j, q)];
                const uint32_t y = tbv_target(blk);
                if (y < kTbvGhostRows || y - kTbvGhostRows >= T || (blk[0] & 0xF000u) != kTbvDst) return 63;
                uint32_t m = kTbInfBits;
                for (uint32_t k = 0; k < kTbvSources; ++k) {
                  const uint32_t r = tbv_source(blk, k);
                  const uint32_t val = r < kTbvGhostRows ? Din[gs + r] : (lds[r - kTbvGhostRows] & 0x7fffffffu);   // (an unused slot names the target itself, weight +inf)
                  if (r < kTbvGhostRows && r >= W.nh) return 63;
                  m = std::min(m, f2u(u2f(val) + u2f(blk[8 + k])));
                }
                uint32_t& tv = lds[y - kTbvGhostRows];
                if (m < (tv & 0x7fffffffu)) tv = m | kTbDirty;
              }
            }
          } else
          for (uint32_t c = 0; c < W.pre_chunks; ++c) {
            const uint32_t* cur = &H.stream[((size_t)W.pre_off + c) * kTbChunk];
            const uint32_t* G = &Din[gs + 4 * (size_t)cur[12]];
            for (uint32_t j = 0; j < kTbBlocksPerChunk; ++j) {
              const uint32_t* K = cur + kTbBlock * j;
              const uint32_t hd = K[0], cnt = (hd >> 8) & 7u;
              if (!cnt) continue;
              const float g = u2f(G[hd & 3u]);
              for (uint32_t k = 0; k < kTbGhostEdges; ++k) if (k < cnt) {
                const uint32_t row = K[1 + k] / 256u;
                const uint32_t nd = f2u(g + u2f(K[6 + k]));
                if (nd < (lds[row] & 0x7fffffffu)) lds[row] = nd | kTbDirty;
              }
            }
          }
        }
        // sweeps, all lanes in lockstep until a sweep changes nothing in any of them
        uint32_t sweep = 0;
        if (vlayout) {
          // the register-resident kernel: one wave per item, blocks of up to six sources, rows by index; a tile none of whose
          // vertices has a source inside it has no blocks (one "sweep" that changes nothing)
          const uint32_t voff = H.vtile[(size_t)kTbvTileWords * t + 2], vch = H.vtile[(size_t)kTbvTileWords * t + 3];
          for (;;) {
            uint32_t chg = 0;
            for (uint32_t c = 0; c < vch; ++c) {
              const uint32_t* C = &H.vstream[((size_t)voff + (size_t)(sweep & 3u) * vch + c) * kTbChunk];
              blocks_total += kTbBlocksPerChunk;
              for (uint32_t j = 0; j < kTbBlocksPerChunk; ++j) {
                uint32_t blk[kTbBlock];
                for (uint32_t q = 0; q < kTbBlock; ++q) blk[q] = C[tb_sweep_index(j, q)];
                ++blocks_eval;
                const uint32_t y = tbv_target(blk) - kTbvGhostRows;
                if (y >= T) return 62;
                for (uint32_t l = 0; l < cnt_l; ++l) {
                  uint32_t* lds = ldsv[l].data();
                  const uint32_t acc0 = lds[y] & 0x7fffffffu;
                  uint32_t acc = acc0;
                  for (uint32_t k = 0; k < kTbvSources; ++k) { const uint32_t r = tbv_source(blk, k) - kTbvGhostRows; if (r >= T) return 62; acc = std::min(acc, fabs_bits_add(lds[r], blk[8 + k])); }
                  if (acc < acc0) { lds[y] = acc | kTbDirty; chg = 1; }
                }
              }
            }
            ++sweep;
            if (!chg) break;
            if (sweep >= 16u * T) return 60;
          }
        } else
        for (;;) {
          const uint32_t* B = &H.stream[((size_t)W.sweep_off + (size_t)(sweep & 3u) * W.sweep_chunks) * kTbChunk];
          uint32_t chg_cur = 0;
          for (uint32_t cc = 0; cc < W.sweep_chunks + (W.sweep_chunks ? rerun : 0u); ++cc) {
            const uint32_t c = std::min(cc, W.sweep_chunks - 1u);
            const uint32_t* C = B + (size_t)c * kTbChunk;
            blocks_total += kTbBlocksPerChunk;
            for (uint32_t j = 0; j < kTbBlocksPerChunk; ++j) {
              auto K = [&](uint32_t q) { return C[tb_sweep_index(j, q)]; };   // sweep chunks are stored transposed (mnav_tb_build.h)
              ++blocks_eval;
              const uint32_t y = K(0) / 256u;
              for (uint32_t l = 0; l < cnt_l; ++l) {
                uint32_t* lds = ldsv[l].data();
                const uint32_t acc0 = lds[y] & 0x7fffffffu;
                uint32_t acc = acc0;
                for (uint32_t k = 0; k < 7; ++k) acc = std::min(acc, fabs_bits_add(lds[K(1 + k) / 256u], K(8 + k)));
                if (acc < acc0) { lds[y] = acc | kTbDirty; chg_cur |= 1u << (y * 32u / T); }
              }
            }
          }
          ++sweep;
          if (!chg_cur) break;
          if (sweep >= 16u * T) return 60;
        }
        sweeps_tot += sweep; max_sweeps = std::max<uint64_t>(max_sweeps, sweep);
        for (uint32_t l = 0; l < cnt_l; ++l) {
          const uint32_t p = bucket[t][start + l];
          uint32_t* lds = ldsv[l].data();
          const size_t sl = slot(t, p, 0), gs = sl + T;
          // write back
          for (uint32_t r = 0; r < T; ++r) if (lds[r] & kTbDirty) D[sl + r] = lds[r] & 0x7fffffffu;
          // post
          uint32_t cnd = kTbInfBits, best = kTbInfBits;
          if (vlayout) {
            const uint32_t* M = &H.vtile[(size_t)kTbvTileWords * t];
            uint32_t gi = 0;
            for (uint32_t c = 0; c < M[5]; ++c) {
              const uint32_t* C = &H.vstream[((size_t)M[4] + c) * kTbChunk];
              for (uint32_t j = 0; j < kTbBlocksPerChunk; ++j) {
                uint32_t blk[kTbBlock];
                for (uint32_t q = 0; q < kTbBlock; ++q) blk[q] = C[tb_sweep_index(j, q)];
                const uint32_t h = tbv_target(blk), fl = tbv_flags(blk);
                if (h >= kTbvGhostRows || (blk[0] & 0xF000u) != kTbvSrc) return 63;
                for (uint32_t k = 0; k < kTbvSources; ++k) {
                  const uint32_t r = tbv_source(blk, k);
                  if (r < kTbvGhostRows) { if (r != h || blk[8 + k] != kTbInfBits) return 63; continue; }   // unused slot
                  cnd = std::min(cnd, fabs_bits_add(lds[r - kTbvGhostRows], blk[8 + k]));
                }
                if (fl & kTbvGhostEnd) { if (cnd < Din[gs + h]) best = std::min(best, cnd); cnd = kTbInfBits; }
                if (fl & kTbvTileEnd) {
                  if (gi >= M[7]) return 63;
                  const uint32_t owner = H.vgroups[(size_t)M[6] + gi++];
                  if (best != kTbInfBits) {
                    const size_t pi = (size_t)owner * NP + p;
                    const uint32_t old = pend[pi];
                    if (best < old) { pend[pi] = best; marr[par ^ 1][p] = std::min(marr[par ^ 1][p], best); }
                    if (old == kTbInfBits) cand[par ^ 1].push_back({ owner, p });
                    ++wakes;
                  }
                  best = kTbInfBits;
                }
              }
            }
            if (gi != M[7]) return 63;
          } else
          for (uint32_t c = 0; c < W.post_chunks; ++c) {
            const uint32_t* cur = &H.stream[((size_t)W.post_off + c) * kTbChunk];
            const uint32_t* G = &Din[gs + 4 * (size_t)cur[12]];
            for (uint32_t j = 0; j < kTbBlocksPerChunk; ++j) {
              const uint32_t* K = cur + kTbBlock * j;
              const uint32_t hd = K[0], cnt = (hd >> 8) & 7u;
              if (!cnt) continue;
              for (uint32_t k = 0; k < kTbGhostEdges; ++k) if (k < cnt) cnd = std::min(cnd, fabs_bits_add(lds[K[1 + k] / 256u], K[6 + k]));
              if (hd & kTbGhostEnd) { if (cnd < G[hd & 3u]) best = std::min(best, cnd); cnd = kTbInfBits; }
              if (hd & kTbTileEnd) {
                if (best != kTbInfBits) {
                  const size_t pi = (size_t)K[11] * NP + p;
                  const uint32_t old = pend[pi];
                  if (best < old) { pend[pi] = best; marr[par ^ 1][p] = std::min(marr[par ^ 1][p], best); }
                  if (old == kTbInfBits) cand[par ^ 1].push_back({ K[11], p });
                  ++wakes;
                }
                best = kTbInfBits;
              }
            }
          }
          // export
          if (vlayout) {
            const uint32_t* M = &H.vtile[(size_t)kTbvTileWords * t];
            uint32_t covered = 0;
            for (uint32_t k = 0; k < M[9]; ++k) {
              const TbvExp& e = H.vexps[(size_t)M[8] + k];
              if (e.n < 1 || e.n > 4) return 63;
              for (uint32_t q = 0; q < e.n; ++q) {
                const uint32_t row = (e.rows >> (8 * q)) & 0xFFu;
                if (row < kTbvGhostRows || row - kTbvGhostRows >= T) return 63;
                const uint32_t v = lds[row - kTbvGhostRows];
                if (v & kTbDirty) D[(size_t)e.soff * NP + (size_t)p * e.sl + e.off + q] = v & 0x7fffffffu;
                ++covered;
              }
            }
            if (covered != W.exp_n) return 63;
          } else
          for (uint32_t k = 0; k < W.exp_n; ++k) {
            const TbExp& e = H.exps[W.exp_off + k];
            const uint32_t v = lds[e.u / 256u];
            if (v & kTbDirty) D[(size_t)e.soff * NP + (size_t)p * e.sl + e.off] = v & 0x7fffffffu;
          }
        }
      }
      bucket[t].clear();
    }
    if (iters > 1000000) return 61;
  }
  for (uint32_t p = 0; p < NP; ++p)
    for (uint32_t v = 0; v < V; ++v) dist_out[(size_t)p * V + v] = u2f(D[slot(H.vert_tile[v], p, H.vert_local[v])]);
  if (stats_out) { stats_out[0] = iters; stats_out[1] = acts; stats_out[2] = sweeps_tot; stats_out[3] = wakes; stats_out[4] = max_sweeps; stats_out[5] = nt; stats_out[6] = H.S; stats_out[7] = items; stats_out[8] = blocks_total; stats_out[9] = blocks_eval; stats_out[10] = stale_reads; stats_out[11] = H.max_nh; }
  return 0;
}

// the pair of mnav::tb_div_magic against the division it replaces: every n of [n_lo, n_hi) in steps of `stride`, and the values
// around every multiple of d in that range when `edges` is set; returns the number of differences
unsigned long long tbm_div_magic_check(uint32_t d, uint32_t n_lo, uint32_t n_hi, uint32_t stride, int edges)
{
  uint32_t M = 0, s = 0;
  mnav::tb_div_magic(d, &M, &s);
  unsigned long long bad = 0;
  for (uint64_t n = n_lo; n < n_hi; n += stride) bad += mnav::tb_div_by_magic((uint32_t)n, d, M, s) != (uint32_t)n / d;
  if (edges && d)
    for (uint64_t m = (uint64_t)(n_lo / d) * d; m < n_hi; m += d)
      for (int k = -1; k <= 1; ++k) {
        const uint64_t n = m + (uint64_t)(int64_t)k;
        if (n >= n_lo && n < n_hi) bad += mnav::tb_div_by_magic((uint32_t)n, d, M, s) != (uint32_t)n / d;
      }
  return bad;
}

}  // extern "C"
