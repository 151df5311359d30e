// mnav_map_kernels.h -- the streaming kernels around the planners: vector maps (dijkstra :189-209, cvp :204-239), cost
// combination, edge weights (full and incremental), the materialised gather CSR / corner tables, Steepness, the Inflation
// layer's wave pieces.  Included by mnav.hip inside its anonymous namespace; not a stand-alone header.
#pragma once

// ---------------------------------------------------------------------------------------------
// vector maps: dijkstra :189-209, cvp :204-239
// ---------------------------------------------------------------------------------------------

__global__ __launch_bounds__(kBlock) void k_vecmap_dijkstra(const Plan* __restrict__ plans, const float* __restrict__ xyz,
                                                            float* const* __restrict__ vecmaps)
{
  const Plan& P = plans[blockIdx.y];
  float* vm = vecmaps[blockIdx.y];
  const uint32_t stride = gridDim.x * kBlock;
  for (uint32_t v = blockIdx.x * kBlock + threadIdx.x; v < P.V; v += stride) {
    const uint32_t p = P.pred[v];
    float x = 0.f, y = 0.f, z = 0.f;
    if (p != v) {                                               // :197
      x = xyz[3 * (size_t)p] - xyz[3 * (size_t)v];              // :204
      y = xyz[3 * (size_t)p + 1] - xyz[3 * (size_t)v + 1];
      z = xyz[3 * (size_t)p + 2] - xyz[3 * (size_t)v + 2];
      const float len = sqrtf(x * x + y * y + z * z);           // normalized(), :206
      x = x / len; y = y / len; z = z / len;
    }
    store3(vm + 3 * (size_t)v, x, y, z);
  }
}

__global__ __launch_bounds__(kBlock) void k_vecmap_cvp(const Plan* __restrict__ plans, const float* __restrict__ xyz,
                                                       const float* __restrict__ nrm, float* const* __restrict__ vecmaps,
                                                       const float* __restrict__ seed_pos)
{
  const Plan& P = plans[blockIdx.y];
  float* vm = vecmaps[blockIdx.y];
  const uint32_t stride = gridDim.x * kBlock;
  for (uint32_t v = blockIdx.x * kBlock + threadIdx.x; v < P.V; v += stride) {
    const uint32_t p = P.pred[v];
    float x = 0.f, y = 0.f, z = 0.f;
    if (is_seed(P, v)) {                                        // cvp :722-724 (un-normalised diff)
      const float* sp = seed_pos + 3 * (size_t)blockIdx.y;
      x = sp[0] - xyz[3 * (size_t)v]; y = sp[1] - xyz[3 * (size_t)v + 1]; z = sp[2] - xyz[3 * (size_t)v + 2];
    } else if (p != v && P.cutf[v] != kNone) {                  // :218, :222-225
      const float dx = xyz[3 * (size_t)p] - xyz[3 * (size_t)v];
      const float dy = xyz[3 * (size_t)p + 1] - xyz[3 * (size_t)v + 1];
      const float dz = xyz[3 * (size_t)p + 2] - xyz[3 * (size_t)v + 2];
      const float nx = nrm[3 * (size_t)v], ny = nrm[3 * (size_t)v + 1], nz = nrm[3 * (size_t)v + 2];
      // rotated(normal, direction) :234 -- Rodrigues (CONVENTION, lvr2 un-vendored; see oracle)
      const float ang = P.dirn[v];
      const float c = cosf_ref(ang), s = sinf_ref(ang);   // the host libm's bits (mnav_eval.h): the field is the reference's bit for bit
      const float cx = ny * dz - nz * dy, cy = nz * dx - nx * dz, cz = nx * dy - ny * dx;
      const float ndv = nx * dx + ny * dy + nz * dz;
      const float k = ndv * (1.0f - c);
      x = dx * c + cx * s + nx * k; y = dy * c + cy * s + ny * k; z = dz * c + cz * s + nz * k;
      const float len = sqrtf(x * x + y * y + z * z);           // :236
      x = x / len; y = y / len; z = z / len;
    }
    store3(vm + 3 * (size_t)v, x, y, z);
  }
}

// ---------------------------------------------------------------------------------------------
// input preparation
// ---------------------------------------------------------------------------------------------
// MeshMap::computeEdgeWeights, mesh_map.cpp:517-561 (exact promotion order, no contraction)
// Combination layers on the device (mesh_layers/src/combination_layer.cpp:44-85 Max, :185-248 weighted
// sum): the inputs are dense V-sized layers (missing entries already replaced by the layer default,
// :62-65 / :201-205), combined in the order given, starting from defaultValue() = 0.
__global__ __launch_bounds__(kBlock) void k_combine(uint32_t V, int mode, uint32_t n_layers, const float* __restrict__ layers,
                                                    const float* __restrict__ weights, float* __restrict__ out)
{
  const uint32_t v = blockIdx.x * kBlock + threadIdx.x;
  if (v >= V) return;
  float cost = 0.0f;                                               // defaultValue(), combination_layer.h:52,94
  for (uint32_t l = 0; l < n_layers; ++l) {
    const float tmp = layers[(size_t)l * V + v];
    if (mode == 0) cost = (cost < tmp) ? tmp : cost;               // std::max(cost, tmp) :66
    else cost += weights[l] * tmp;                                 // :206 (float multiply, float add)
  }
  out[v] = cost;
}

__global__ __launch_bounds__(kBlock) void k_edge_weights(uint32_t E, const uint32_t* __restrict__ edge_vtx,
                                                         const float* __restrict__ edge_dist, const float* __restrict__ cost,
                                                         double factor, float* __restrict__ w)
{
  const uint32_t e = blockIdx.x * kBlock + threadIdx.x;
  if (e >= E) return;
  const float c1 = cost[edge_vtx[2 * (size_t)e]], c2 = cost[edge_vtx[2 * (size_t)e + 1]];   // :528-529
  if (isinf(c1) || isinf(c2)) { w[e] = inf_f(); return; }                                    // :538-542
  const float vd = edge_dist[e];                                                             // :548
  const float edge_cost = (float)((double)(vd * (c1 + c2)) / 2.0);                           // :550
  w[e] = (float)((double)vd + factor * (double)edge_cost);                                   // :552
}

// Incremental cost change (MeshMap::layerChanged mesh_map.cpp:454-493 + updateEdgeWeights :563-618): the changed
// vertices get their new cost, then only the edges around them are re-weighted -- same expressions as :550-552.
__global__ __launch_bounds__(kBlock) void k_scatter_costs(uint32_t n, const uint32_t* __restrict__ ids, const float* __restrict__ values,
                                                          float* __restrict__ cost)
{
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) cost[ids[i]] = values[i];
}
__global__ __launch_bounds__(kBlock) void k_update_edge_weights(uint32_t n, const uint32_t* __restrict__ ids, const uint32_t* __restrict__ row_ptr,
                                                                const uint32_t* __restrict__ nbr_u, const uint32_t* __restrict__ nbr_e,
                                                                const float* __restrict__ edge_dist, const float* __restrict__ cost,
                                                                double factor, float* __restrict__ w)
{
  // 8 lanes per changed vertex, one incident edge each (an edge between two changed vertices is written twice with
  // the same value)
  const uint32_t i = (blockIdx.x * kBlock + threadIdx.x) >> 3;
  const int sub = threadIdx.x & 7;
  if (i >= n) return;
  const uint32_t v = ids[i];
  const float c1 = cost[v];
  for (uint32_t k = row_ptr[v] + sub; k < row_ptr[v + 1]; k += 8) {
    const uint32_t e = nbr_e[k];
    const float c2 = cost[nbr_u[k]];
    if (isinf(c1) || isinf(c2)) { w[e] = inf_f(); continue; }       // :596-600
    const float vd = edge_dist[e];                                   // :606
    const float edge_cost = (float)((double)(vd * (c1 + c2)) / 2.0); // :608 (float sum: commutative, the endpoint order is free)
    w[e] = (float)((double)vd + factor * (double)edge_cost);        // :610
  }
}

// gather CSR for Dijkstra: {u, w(u,v)}; w=+inf when v is invalid (:328) or u is over the cost
// limit (:302, u would be popped but never expanded)
__global__ __launch_bounds__(kBlock) void k_build_nbr(uint32_t V, const uint32_t* __restrict__ row_ptr,
                                                      const uint32_t* __restrict__ nbr_u, const uint32_t* __restrict__ nbr_e,
                                                      const float* __restrict__ w, const float* __restrict__ cost,
                                                      const uint8_t* __restrict__ invalid, double cost_limit, Nbr* __restrict__ out)
{
  const uint32_t v = blockIdx.x * kBlock + threadIdx.x;
  if (v >= V) return;
  const bool vinv = invalid[v] != 0;
  for (uint32_t i = row_ptr[v]; i < row_ptr[v + 1]; ++i) {
    const uint32_t u = nbr_u[i];
    float ww = w[nbr_e[i]];
    if (vinv || (double)cost[u] > cost_limit) ww = inf_f();
    Nbr n; n.u = u; n.w = ww;
    out[i] = n;
  }
}

struct CornerIdx { uint32_t v1, v2, ea, eb, ec, face; };

__global__ __launch_bounds__(kBlock) void k_build_crn(uint32_t V, const uint32_t* __restrict__ crn_ptr,
                                                      const CornerIdx* __restrict__ idx, const float* __restrict__ w,
                                                      const float* __restrict__ cost, const uint8_t* __restrict__ invalid,
                                                      double cost_limit, Corner* __restrict__ out, uint8_t* __restrict__ blocked)
{
  const uint32_t v = blockIdx.x * kBlock + threadIdx.x;
  if (v >= V) return;
  const bool vinv = invalid[v] != 0;
  blocked[v] = ((double)cost[v] >= cost_limit || vinv) ? 1 : 0;      // cvp :802,825,848 / :785
  for (uint32_t i = crn_ptr[v]; i < crn_ptr[v + 1]; ++i) {
    const CornerIdx k = idx[i];
    Corner c;
    c.v1 = (vinv || invalid[k.v1] || invalid[k.v2]) ? kNone : k.v1;   // cvp :785
    c.v2 = k.v2; c.a = w[k.ea]; c.b = w[k.eb]; c.c = w[k.ec]; c.face = k.face;
    out[i] = c;
  }
}

// ---------------------------------------------------------------------------------------------
// Layers on the device (mesh_layers): Steepness (steepness_layer.cpp:157-166, :82-93), Inflation
// (inflation_layer.cpp:341-491 as a multi-source wave on the band engine; spec: mnav_eval.h eval_cvp / Plan.seed_mask)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_edge_dist(uint32_t E, const uint32_t* __restrict__ edge_vtx, const float* __restrict__ xyz,
                                                      float* __restrict__ out)
{
  const uint32_t e = blockIdx.x * kBlock + threadIdx.x;
  if (e >= E) return;
  const float* a = xyz + 3 * (size_t)edge_vtx[2 * (size_t)e];
  const float* b = xyz + 3 * (size_t)edge_vtx[2 * (size_t)e + 1];
  const float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  out[e] = sqrtf(dx * dx + dy * dy + dz * dz);                     // lvr2 BaseVector::distanceFrom in float (mesh_map.cpp:347)
}

__global__ __launch_bounds__(kBlock) void k_steepness(uint32_t V, const float* __restrict__ nrm, double threshold,
                                                      float* __restrict__ cost, uint8_t* __restrict__ lethal)
{
  const uint32_t v = blockIdx.x * kBlock + threadIdx.x;
  if (v >= V) return;
  const float st = acosf_ref(nrm[3 * (size_t)v + 2]);              // :165 (float overload of acos; the host libm's bits, mnav_eval.h)
  cost[v] = st;
  lethal[v] = ((double)st > threshold) ? 1 : 0;                    // :88
}

// corners with the edge DISTANCES as side lengths (waveCostInflation reads map->edgeDistances() :383); no face is
// skipped here: what may fire is decided by Plan.seed_mask
__global__ __launch_bounds__(kBlock) void k_build_crn_infl(uint32_t V, const uint32_t* __restrict__ crn_ptr,
                                                           const CornerIdx* __restrict__ idx, const float* __restrict__ w,
                                                           Corner* __restrict__ out)
{
  const uint32_t v = blockIdx.x * kBlock + threadIdx.x;
  if (v >= V) return;
  for (uint32_t i = crn_ptr[v]; i < crn_ptr[v + 1]; ++i) {
    const CornerIdx k = idx[i];
    Corner c;
    c.v1 = k.v1; c.v2 = k.v2; c.a = w[k.ea]; c.b = w[k.eb]; c.c = w[k.ec]; c.face = corner_face_for_inflation(k.face);
    out[i] = c;
  }
}

__global__ __launch_bounds__(kBlock) void k_infl_mask(uint32_t V, const uint8_t* __restrict__ lethal, const uint8_t* __restrict__ invalid,
                                                      uint8_t* __restrict__ mask)
{
  const uint32_t v = blockIdx.x * kBlock + threadIdx.x;
  if (v >= V) return;
  const bool l = lethal[v] != 0, inv = invalid && invalid[v] != 0;
  mask[v] = l ? (inv ? kInflSeedMute : kInflSeed) : (inv ? kInflMute : kInflFree);
}

// control blocks of the wave (the single-thread part of k_seed), then the seeds in parallel: every lethal vertex is
// fixed at distance 0 (:397-402) and the free vertices around it form the first work list
__global__ void k_infl_ctl(const Plan* __restrict__ plans)
{
  const Plan& P = plans[0];
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  Ctl c0; memset(&c0, 0, sizeof(c0));
  c0.it = -1; c0.n = 0; c0.thr_fixed = -inf_f(); c0.goal_dist = inf_f(); c0.arm_vertex = kNone;
  c0.thr = P.delta; if (!(c0.thr > 0.0f)) c0.thr = next_up(0.0f);
  c0.band_new = 1; c0.width = P.delta; c0.wmin = inf_f(); c0.epoch = 1;
  P.ctl[1] = c0;
  P.ctl[0] = c0;
  Cnt ci; memset(&ci, 0, sizeof(ci)); ci.changed = 1; ci.minkey = 0x7f800000u; ci.minchg = 0x7f800000u;
  P.cnt[2] = ci;                       // read by step 0 as "(0-1) mod 3"; k_infl_seed counts the list into it
  Cnt z; memset(&z, 0, sizeof(z)); z.minkey = 0x7f800000u; z.minchg = 0x7f800000u;
  P.cnt[0] = z; P.cnt[1] = z; P.cnt[3] = z;
}

__global__ __launch_bounds__(kBlock) void k_infl_seed(const Plan* __restrict__ plans)
{
  const Plan& P = plans[0];
  const uint32_t stride = gridDim.x * kBlock;
  uint32_t* l0 = P.list[0];
  for (uint32_t v = blockIdx.x * kBlock + threadIdx.x; v < P.V; v += stride) {
    if (!is_seed(P, v)) continue;
    P.dist[v] = 0.0f; P.tkey[v] = make_key(0.0f, v); P.keyd[v] = 0.0f;
    for (uint32_t i = P.crn_ptr[v]; i < P.crn_ptr[v + 1]; ++i) {
      const Corner c = P.crn[i];
      const uint32_t nb[2] = { c.v1, c.v2 };
      for (int q = 0; q < 2; ++q) {
        const uint32_t u = nb[q];
        if (u == kNone || is_seed(P, u)) continue;
        if (P.stamp[u] != 0xFFFFFFFFu && atomicExch(&P.stamp[u], 0xFFFFFFFFu) != 0xFFFFFFFFu) {
          const uint32_t at = atomicAdd(&P.cnt[2].n_next, 1u);
          if (at < P.cap) l0[at] = u;
        }
      }
    }
  }
}

// The inflation layer's repulsive vector field from the converged wave (spec: mnav_eval.h infl_accumulate / infl_assign).
// state: 0 = open (a free vertex with a distance whose vector may still be assigned), 1 = final with a vector, 2 = final
// without one.  k_infl_assign is launched until nothing is open; it reads the states of the PREVIOUS launch and writes the
// next ones to a second array, so a vector is only ever read after the launch that wrote it has ended.
__global__ __launch_bounds__(kBlock) void k_infl_accum(const Plan* __restrict__ plans, const uint32_t* __restrict__ crn_walk,
                                                       const float* __restrict__ xyz, float* __restrict__ vec, uint8_t* __restrict__ state,
                                                       uint8_t* __restrict__ acc, uint32_t* __restrict__ ctl)
{
  const Plan& P = plans[0];
  const uint32_t v = blockIdx.x * kBlock + threadIdx.x;
  if (v >= P.V) return;
  float o[3] = { 0.f, 0.f, 0.f };
  const int r = infl_accumulate(P, crn_walk, xyz, v, o);
  if (r < 0) { atomicOr(&ctl[2], 1u); return; }
  vec[3 * (size_t)v] = o[0]; vec[3 * (size_t)v + 1] = o[1]; vec[3 * (size_t)v + 2] = o[2];
  acc[v] = r == 1 ? 1 : 0;
  const bool open = !is_seed(P, v) && P.dist[v] < inf_f();
  state[v] = open ? 0 : (r == 1 ? 1 : 2);
}

__global__ __launch_bounds__(kBlock) void k_infl_assign(const Plan* __restrict__ plans, float* __restrict__ vec, const uint8_t* __restrict__ state,
                                                        uint8_t* __restrict__ state_next, const uint8_t* __restrict__ acc, uint32_t* __restrict__ ctl)
{
  const Plan& P = plans[0];
  const uint32_t v = blockIdx.x * kBlock + threadIdx.x;
  if (v >= P.V) return;
  const uint8_t st = state[v];
  if (st != 0) { state_next[v] = st; return; }
  float o[3];
  const int r = infl_assign(P, vec, state, v, o);
  if (r == 2) { state_next[v] = 0; atomicAdd(&ctl[0], 1u); return; }   // a support is still open: next launch
  if (r == 1) { vec[3 * (size_t)v] = o[0]; vec[3 * (size_t)v + 1] = o[1]; vec[3 * (size_t)v + 2] = o[2]; }
  state_next[v] = (r == 1 || acc[v]) ? 1 : 2;
  atomicAdd(&ctl[1], 1u);
}

// riskiness from the distances: fading() :315-339; vertices the wave never reached keep the default 0
// (inflation_layer.h:74-77).  The exponential runs in float64 and is rounded to float32 (:326).
__global__ __launch_bounds__(kBlock) void k_infl_cost(uint32_t V, const float* __restrict__ dist, double inflation_radius,
                                                      double inscribed_radius, double inscribed_value, double lethal_value,
                                                      double cost_scaling_factor, float* __restrict__ cost)
{
  const uint32_t v = blockIdx.x * kBlock + threadIdx.x;
  if (v >= V) return;
  const float d = dist[v];
  float c;
  if (!(d < inf_f())) c = 0.0f;
  else if ((double)d > inflation_radius) c = 0.0f;                                        // :317-320
  else if ((double)d > inscribed_radius) {                                                // :323
    const float factor = (float)exp(-1.0 * cost_scaling_factor * ((double)d - inscribed_radius));   // :326
    c = (float)(inscribed_value * (double)factor);                                        // :327
  }
  else if (d > 0) c = (float)inscribed_value;                                             // :332-335
  else c = (float)lethal_value;                                                           // :338
  cost[v] = c;
}

__global__ __launch_bounds__(kBlock) void k_combine_resident(uint32_t V, int mode, uint32_t n_layers, const float* const* __restrict__ layers,
                                                             const float* __restrict__ weights, float* __restrict__ out)
{
  const uint32_t v = blockIdx.x * kBlock + threadIdx.x;
  if (v >= V) return;
  float cost = 0.0f;                                               // defaultValue(), combination_layer.h:52,94
  for (uint32_t l = 0; l < n_layers; ++l) {
    const float tmp = layers[l][v];
    if (mode == 0) cost = (cost < tmp) ? tmp : cost;               // std::max(cost, tmp) :66
    else cost += weights[l] * tmp;                                 // :206
  }
  out[v] = cost;
}

// CombinationLayer::onInputChanged (combination_layer.cpp:87-147 max, :250-302 weighted sum): only the changed vertices
__global__ __launch_bounds__(kBlock) void k_combine_resident_ids(uint32_t n, const uint32_t* __restrict__ ids, int mode, uint32_t n_layers,
                                                                 const float* const* __restrict__ layers, const float* __restrict__ weights,
                                                                 float* __restrict__ out, float* __restrict__ values)
{
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const uint32_t v = ids[i];
  float cost = 0.0f;
  for (uint32_t l = 0; l < n_layers; ++l) {
    const float tmp = layers[l][v];
    if (mode == 0) cost = (tmp < cost) ? cost : tmp;               // std::max(tmp, cost) :117
    else cost += weights[l] * tmp;                                 // :281
  }
  out[v] = cost;
  values[i] = cost;
}
