// mnav_shard.h -- device kernels of ONE Dijkstra plan (dijkstra_mesh_planner.cpp:287-348) on a mesh partitioned over several
// GPUs: pack / apply of the interface buffer, termination words, path segments.  Included by mnav.hip inside its
// anonymous namespace, after TilePlan / TCtl / TCnt are defined; the C ABI on top of them is mnav_shard_capi.h.
#pragma once

// ---------------------------------------------------------------------------------------------
// ONE plan on a mesh that is range-partitioned over several processes / GPUs (BASELINE config 4,
// SURVEY.md 8e).  The tiles are in Morton order; process r owns a contiguous range of them and, with
// them, their vertices.  Every process runs the ordinary tile rounds (k_tile_round) on its own tiles
// only; between blocks of rounds the distances of the INTERFACE vertices (vertices with a neighbour
// owned by somebody else, plus the robot vertex) are exchanged with one min-allreduce over a dense
// buffer (RCCL over xGMI; torch.distributed in the Python driver), and a vertex whose value dropped
// wakes the local tiles that have it in their halo.  Label-correcting: the fixed point, and with it
// every bit of the potential, is the one of the unpartitioned run.
// ---------------------------------------------------------------------------------------------
struct ShardDev {
  uint32_t n_iface, rank, target;
  uint32_t partition;            // 1: partitioned mesh -- iface_vert holds LOCAL ids (kNone: vertex not held here), every held copy is
                                 // packed (a valid upper bound) and every held copy takes a smaller reduced value
  const uint32_t* iface_vert;    // n_iface vertex ids, the same list on every process
  const uint8_t* iface_owner;    // n_iface owning process
  const uint32_t* wake_ptr;      // n_iface+1 -> wake_tile: local tiles that hold the vertex in their halo
  const uint32_t* wake_tile;
};

// pack: own interface values, +inf for the others (the min-allreduce then delivers every owner's value)
__global__ __launch_bounds__(kBlock) void k_shard_pack(ShardDev S, const TilePlan* __restrict__ plans, float* __restrict__ buf,
                                                       uint32_t* __restrict__ changed, uint32_t* __restrict__ minpend)
{
  if (blockIdx.x == 0 && threadIdx.x == 0) { *changed = 0u; *minpend = kInfBits; }   // the words the apply step accumulates into ("nothing pending")
  const float* dist = plans[0].dist;                                  // (the robot vertex comes from the plan record too: the captured
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;               //  exchange graphs hold nothing that changes from plan to plan)
  if (i < S.n_iface) {
    const uint32_t v = S.iface_vert[i];
    buf[i] = (v != kNone && (S.partition || S.iface_owner[i] == S.rank)) ? dist[v] : inf_f();
  }
  if (i == S.n_iface) buf[i] = dist[plans[0].target];   // last slot: the robot vertex (bound / goal_dist need it everywhere); stale copies are larger
}

// apply: ghost values that dropped are stored and wake the local tiles around them for the next round;
// the round controller is re-armed (its `done` is sticky) and told about the new smallest wake-up value
__global__ __launch_bounds__(kBlock) void k_shard_apply(ShardDev S, const TilePlan* __restrict__ plans, const float* __restrict__ buf,
                                                        uint32_t* __restrict__ changed)
{
  const TilePlan& P = plans[0];
  const TCtl a = P.ctl[0], b = P.ctl[1];
  const int32_t j = (a.it > b.it) ? a.it : b.it;                     // last round executed (-1: none yet)
  uint32_t* pn = P.pend[(j + 1) & 1];                                // the buffer round j+1 reads
  TCnt* cnt = &P.cnt[((j % 3) + 3) % 3];                             // ... and the counters it reads as "previous"
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i == 0) { P.ctl[0].done = 0; P.ctl[1].done = 0; }
  if (i == S.n_iface) { if (buf[i] < P.dist[P.target]) P.dist[P.target] = buf[i]; return; }
  if (i >= S.n_iface) return;
  const uint32_t v = S.iface_vert[i];
  if (v == kNone || (!S.partition && S.iface_owner[i] == S.rank)) return;
  const float nv = buf[i];
  if (!(nv < P.dist[v])) return;
  P.dist[v] = nv;
  const uint32_t bits = f2u(nv);
  // Partitioned mesh: v sits INSIDE a local tile (first entry of its wake list).  A tile only re-queues its own vertices
  // from the threshold of its last solve upwards (k_tile_round: sources in [tlast, thr)), so a value that arrives from
  // outside below that threshold pulls it down.  Signed min on the float bits: the negative marks (-inf: never solved) stay.
  if (S.partition) atomicMin((int*)&P.tlast[S.wake_tile[S.wake_ptr[i]]], (int)bits);
  for (uint32_t k = S.wake_ptr[i]; k < S.wake_ptr[i + 1]; ++k) atomicMin(&pn[S.wake_tile[k]], bits);
  if (S.wake_ptr[i + 1] > S.wake_ptr[i]) { atomicMin(&cnt->minpend, bits); atomicOr(changed, 1u); }
}

// smallest wake-up value among the owned tiles (what this process still has to do), as float bits
__global__ __launch_bounds__(kBlock) void k_shard_minpend(const TilePlan* __restrict__ plans, uint32_t* __restrict__ out)
{
  const TilePlan& P = plans[0];
  const TCtl a = P.ctl[0], b = P.ctl[1];
  const int32_t j = (a.it > b.it) ? a.it : b.it;
  const uint32_t* pn = P.pend[(j + 1) & 1];
  const uint32_t hi = P.t_hi ? P.t_hi : P.ntiles;
  uint32_t m = kInfBits;
  for (uint32_t t = P.t_lo + blockIdx.x * kBlock + threadIdx.x; t < hi; t += gridDim.x * kBlock) m = min(m, pn[t]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0 && m != kInfBits) atomicMin(out, m);
}

// the termination words of one exchange, written on the device: {smallest pending wake-up, dist[target], -status}
__global__ void k_shard_ctl(const uint32_t* __restrict__ minpend, const TilePlan* __restrict__ plans, const uint32_t* __restrict__ cancel,
                            float* __restrict__ ctl)
{
  if (threadIdx.x || blockIdx.x) return;
  const float* dist = plans[0].dist; const uint32_t target = plans[0].target;
  const uint32_t mp = *minpend;
  ctl[0] = (mp >= kInfBits) ? inf_f() : u2f(mp);
  ctl[1] = dist[target];
  ctl[2] = (cancel && __atomic_load_n(cancel, __ATOMIC_RELAXED)) ? -1.0f : 0.0f;
}

// One segment of the vertex path (dijkstra :358-373) inside this process's part: predecessors are followed from `start` while
// the vertex is owned here; out = {count, vertex the walk stopped at, status (1: a vertex without predecessor), ids...}
__global__ void k_shard_walk(const uint32_t* __restrict__ pred, const uint8_t* __restrict__ owned, uint32_t start, uint32_t seed, uint32_t cap,
                             uint32_t* __restrict__ out)
{
  if (threadIdx.x || blockIdx.x) return;
  uint32_t v = start, n = 0, status = 0;
  while (v != seed && (!owned || owned[v]) && n < cap) {
    const uint32_t p = pred[v];
    if (p == v) { status = 1; break; }
    out[3 + n++] = p;
    v = p;
  }
  out[0] = n; out[1] = v; out[2] = status;
}

// final gather buffers: owned entries, neutral elements elsewhere (min-allreduce over dist, pred)
__global__ __launch_bounds__(kBlock) void k_shard_owned(uint32_t V, const uint32_t* __restrict__ vert_tile, uint32_t t_lo, uint32_t t_hi,
                                                        const float* __restrict__ dist, const uint32_t* __restrict__ pred,
                                                        float* __restrict__ dist_out, uint32_t* __restrict__ pred_out)
{
  const uint32_t v = blockIdx.x * kBlock + threadIdx.x;
  if (v >= V) return;
  const uint32_t t = vert_tile[v];
  const bool mine = t >= t_lo && t < t_hi;
  dist_out[v] = mine ? dist[v] : inf_f();
  pred_out[v] = mine ? pred[v] : 0xFFFFFFFFu;
}

