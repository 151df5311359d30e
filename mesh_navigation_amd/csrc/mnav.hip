// mnav.hip -- MI355X (gfx950) wavefront planner: HIP kernels + the C ABI of include/mnav.h.
//
// Hot path replaced (reference file:line):
//   DijkstraMeshPlanner::dijkstra       dijkstra_mesh_planner/src/dijkstra_mesh_planner.cpp:217-398
//   DijkstraMeshPlanner::computeVectorMap                                            :189-209
//   CVPMeshPlanner::waveFrontPropagation cvp_mesh_planner/src/cvp_mesh_planner.cpp:651-918
//   CVPMeshPlanner::waveFrontUpdate                                                  :369-556
//   CVPMeshPlanner::computeVectorMap                                                 :204-239
//   MeshMap::computeEdgeWeights          mesh_map/src/mesh_map.cpp:517-561
//
// Design: the priority-queue loops become distance bands settled by a gather
// rule iterated to its fixed point (mnav_eval.h).  One step = one launch of k_step over the
// current work list of every plan in the batch; steps are enqueued back-to-back from a
// hipGraph with no host round trip, all loop control (band advance, goal_dist arming,
// termination) is recomputed by every workgroup from the previous step's counters.
// Memory-bound irregular gather work: no MFMA, coalescing via CSR rows + 24-byte corner
// records, wave-aggregated atomics for the work lists.
//
// This file never computes a plan on the CPU: every entry point fails when no GPU is usable.
#include <hip/hip_runtime.h>

#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>
#include <unordered_map>
#include <unistd.h>

#include "../../include/mnav.h"
#include "mnav_build.h"
#include "mnav_eval.h"
#include "mnav_options.h"

using namespace mnav;

namespace {

// Pointers that reach a kernel through a struct in memory are generic to the compiler: it emits
// flat_load + s_waitcnt vmcnt(0) lgkmcnt(0) around every LDS access.  Device-memory arrays are
// therefore re-typed as address_space(1) before use (global_load, waits only on vmcnt).
#define MNAV_GLOBAL __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ MNAV_GLOBAL T* as_global(T* p) { return (MNAV_GLOBAL T*)p; }
template <class T>
__device__ __forceinline__ MNAV_GLOBAL T* as_global(const GPtr<T>& p) { return (MNAV_GLOBAL T*)p.p; }
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));   // one Nbr {u, w bits}
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // 16-byte copy unit

constexpr int kBlock = 256;   // streaming kernels (init, vector map, input preparation)
constexpr int kChunk = 96;  // step launches per graph replay; multiple of 6 (slot parities)

#include "mnav_band.h"

#include "mnav_tiles.h"

#include "mnav_shard.h"   // kernels of the sharded single plan (k_shard_*)

#include "mnav_finalize.h"

#include "mnav_plan_kernels.h"

#include "mnav_map_kernels.h"

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
}  // namespace
#include "mnav_tb.h"
#include "mnav_tb_finalize.h"
#include "mnav_tbv.h"
#include "mnav_walk.h"

// One back-tracking job: the plan's resident vector map and the two ends of the walk.
struct WalkJob { const float* vecmap; float seed[3]; uint32_t seed_face; float target[3]; uint32_t target_face; };

// CVPMeshPlanner's back-tracking (cvp_mesh_planner.cpp:920-951) on the resident vector map: one wave per plan.  The loop
// is a dependent chain (each step needs the position the previous one produced), so all 64 lanes walk in lockstep on the
// same data (uniform loads, uniform branches) and divide only the scan of searchNeighbourFaces' face list, which lives in
// LDS.  ctl[2*j] = walk status, ctl[2*j+1] = entries written (walk order, target first).
__global__ __launch_bounds__(64) void k_backtrack(WalkMesh M, WalkInflation L, const WalkJob* __restrict__ jobs, double step_width, uint32_t cap,
                                                  float* __restrict__ pos_out, uint32_t* __restrict__ face_out, int32_t* __restrict__ ctl)
{
  __shared__ uint32_t list[kWalkScratchWords];
  const uint32_t j = blockIdx.x;
  const WalkJob J = jobs[j];
  if (!J.vecmap) { if (threadIdx.x == 0) { ctl[2 * j] = 0; ctl[2 * j + 1] = 0; } return; }   // no plan behind this row
  WalkField Fd;
  Fd.vecmap = J.vecmap;
  for (int k = 0; k < 3; ++k) Fd.seed_vs[k] = M.faces[3 * (size_t)J.seed_face + k];
  uint32_t n = 0;
  const int st = walk_backtrack(M, Fd, L, w3(J.seed[0], J.seed[1], J.seed[2]), J.seed_face, w3(J.target[0], J.target[1], J.target[2]), J.target_face,
                                step_width, cap, pos_out + 3 * (size_t)cap * j, face_out + (size_t)cap * j, &n, list);
  ctl[2 * j] = st; ctl[2 * j + 1] = (int32_t)n;
}
namespace {

struct Slot {
  float *dist = nullptr, *dirn = nullptr, *vecmap = nullptr;
  PopKey* tkey = nullptr;
  uint32_t *pred = nullptr, *cutf = nullptr, *stamp = nullptr, *dirty = nullptr, *list0 = nullptr, *list1 = nullptr;
  uint32_t *wlist0 = nullptr, *wlist1 = nullptr, *wstamp = nullptr;
  Ctl* ctl = nullptr;
  Cnt* cnt = nullptr;
  bool cvp_ready = false, band_ready = false;
  // tiled engine
  uint32_t *tpend0 = nullptr, *tpend1 = nullptr;
  float* tlast = nullptr;
  TCtl* tctl = nullptr;
  TCnt* tcnt = nullptr;
  bool tile_ready = false;
};

}  // namespace

struct mnav_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  std::atomic<int> cancel{ 0 };
  uint32_t* d_cancel = nullptr;            // the same flag in device memory: long-running kernels poll it (agent-scope load)
  uint32_t* h_one = nullptr;               // pinned source word (1) for the copy mnav_cancel issues on its own stream
  hipStream_t cancel_stream = nullptr;
  // host copies needed for seeding
  uint32_t V = 0, F = 0, E = 0;
  std::vector<float> h_xyz, h_cost;
  std::vector<uint32_t> h_row_ptr, h_nbr_u;   // gather CSR (host copy): the tile-batch engine builds its streams from it on first use
  TbState tb; tb::Args tb_args{};             // tile-batch SSSP engine (mnav_tb.h); arguments of the last batch
  bool tb_args_valid = false;                 // ... which still describe live device memory (last tile-batch call succeeded, nothing freed since)
  bool want_vec = false;               // the running call asked for vector maps (lazy 12 B/vertex/plan)
  bool resident_vecmap = false;        // mnav_set_resident_outputs: always compute the vector map, leave it on the device
  std::vector<uint32_t> caller_slot;   // plan index of the caller's batch -> device slot of the last call (kNone: never ran)
  std::vector<uint32_t> h_faces;
  std::vector<uint32_t> h_vf_ptr, h_vf;    // getFacesOfVertex rows (host copy; uploaded on the first device back-tracking call)
  uint32_t *d_faces = nullptr, *d_vf_ptr = nullptr, *d_vf = nullptr; bool walk_mesh_valid = false;
  float* d_walk_pos = nullptr; uint32_t* d_walk_face = nullptr; size_t walk_cap = 0;   // k_backtrack outputs: rows of `cap` entries
  hipEvent_t ev_link[2]{};                 // stream links of the asynchronous shard calls (caller's stream <-> ours)
  struct WalkJob* d_walk_jobs = nullptr; int32_t* d_walk_ctl = nullptr; uint32_t walk_jobs_cap = 0;
  std::vector<uint8_t> h_invalid;
  bool have_mesh = false, have_costs = false, have_normals = false;
  // device mesh
  uint32_t *d_row_ptr = nullptr, *d_nbr_u = nullptr, *d_nbr_e = nullptr, *d_crn_ptr = nullptr, *d_edge_vtx = nullptr;
  CornerIdx* d_crn_idx = nullptr;
  uint32_t* d_crn_walk = nullptr;                                   // HostTopology::crn_walk (inflation vector field)
  float *d_xyz = nullptr, *d_nrm = nullptr, *d_cost = nullptr, *d_w = nullptr, *d_edge_dist = nullptr;
  uint8_t* d_invalid = nullptr;
  // materialised per cost_limit
  Nbr* d_nbr = nullptr; double nbr_limit = NAN; bool nbr_valid = false;
  Corner* d_crn = nullptr; uint8_t* d_blocked = nullptr; double crn_limit = NAN; bool crn_valid = false;
  FaceCirculation circ;                                            // caller-supplied getFacesOfVertex rows (optional)
  uint32_t* d_verify_any = nullptr; uint32_t verify_sweeps_used = 0;
  bool cvp_verify = true;                                          // k_cvp_verify after every CVP plan (MNAV_CVP_VERIFY=0 to skip)
  int walk_max = kKeyWalkMax, descend_max = kDescendWalkMax;       // cascade-tree walk bounds (MNAV_KEY_WALK_MAX / MNAV_DESCEND_WALK_MAX: tests)
  // plans
  std::vector<Slot> slots;
  Plan* d_plans = nullptr; uint32_t plans_cap = 0;
  PlanResult* d_res = nullptr; PlanResult* h_res = nullptr;
  float** d_vecptrs = nullptr;
  uint32_t* d_paths = nullptr; size_t paths_words = 0; uint32_t path_stride = 0;   // n plans x path_stride vertex ids
  uint32_t* d_over = nullptr; unsigned long long* d_over_off = nullptr; uint32_t* d_over_cap = nullptr;   // exact rows of the paths that did not fit
  uint32_t *d_pack = nullptr, *h_pack = nullptr, *d_pack_meta = nullptr; size_t pack_words = 0, pack_meta_n = 0;   // packed paths (device, pinned host), offsets + lengths
  std::unordered_map<void*, size_t> alloc_bytes;                   // sizes of the dev_upload buffers (re-used when unchanged)
  Ctl* h_ctl = nullptr;       // pinned, 2 per plan
  uint32_t infl_exact_bands = 0;   // bands of the last inflation wave that went through the exact band routine (k_exact_band)
  float* d_seed_pos = nullptr; uint32_t seed_pos_cap = 1;
  std::map<uint64_t, hipGraphExec_t> graphs;
  // tiled SSSP engine
  int dij_engine = 3;          // 0 tiled rounds, 1 band steps, 3 auto, 5 tile-batch, 6 asynchronous tiles (2: the per-plan persistent kernel, retired round 5)
  int last_engine = 0;
  bool lazy_paths = false;      // this call only wants vertex paths: k_path_lazy instead of k_dij_finalize + k_finish
  bool allow_lazy_paths = true; // MNAV_LAZY_PATHS=0: always finalize (predecessors / tentative values for everybody)
  uint32_t max_steps = 1u << 20;   // per plan; set from the mesh size at upload (a wavefront needs O(diameter) steps)
  double max_wall_s = 120.0;   // host-side guard: a plan that takes longer is abandoned with an error
  uint32_t tile_size = 512;    // 4 workgroups of the tile kernels per CU (36 KB LDS each)
  float rounds_band_mult = 4.0f;   // the round engine (latency) prefers wide bands
  float tile_band_user = 0.f, tile_band_auto = 1.f;
  uint32_t* d_t_rptr = nullptr;
  HostTiles tiles_meta;        // only the small per-tile vectors are kept (vert_tile, sizes)
  uint32_t *d_t_vptr = nullptr, *d_t_verts = nullptr, *d_t_hptr = nullptr, *d_t_halo_verts = nullptr, *d_t_halo_tile = nullptr,
           *d_t_eptr = nullptr, *d_t_src = nullptr, *d_vert_tile = nullptr, *d_mismatch = nullptr;
  uint16_t *d_t_rowptr = nullptr, *d_t_col = nullptr;
  float* d_t_tw = nullptr; bool tw_valid = false; uint32_t t_nnz = 0;
  // sharded single plan (mnav_shard_*)
  struct Shard {
    bool ready = false, active = false, finalized = false;   // finalized: slot 0 holds the predecessors of the last sharded plan (mnav_shard_walk)
    uint32_t rank = 0, world = 1, t_lo = 0, t_hi = 0, n_iface = 0, rounds_per_exchange = 8, j = 0;
    uint32_t seed = 0, target = 0; double offset = 0.3; uint32_t goal_tie1 = 0;
    uint32_t *d_iface_vert = nullptr, *d_wake_ptr = nullptr, *d_wake_tile = nullptr, *d_changed = nullptr, *d_minpend = nullptr;
    uint8_t* d_iface_owner = nullptr;
    std::vector<uint32_t> iface_vert;
    bool partition = false; uint8_t* d_owned = nullptr;              // mnav_shard_setup_partition
    uint32_t* d_walk = nullptr; uint32_t walk_cap = 0;               // mnav_shard_walk
    std::map<std::array<uint64_t, 3>, hipGraphExec_t> graphs;        // captured exchange sequences (shard_replay)
  } shard;
  double edge_cost_factor = 0.0;                                   // factor of the resident edge weights (mnav_update_costs)
  // layers computed / kept on the device (mnav_layer_*)
  struct Layer { float* cost = nullptr; uint8_t* lethal = nullptr; float* dist = nullptr; float* vec = nullptr; uint8_t* vstate = nullptr;   // vstate: 3 x V (two state arrays + the accumulate flags)
                 bool ready = false, have_vec = false;
                 double inflation_radius = 0, inscribed_radius = 0, inscribed_value = 0, lethal_value = 0; };   // InflationLayer config (vectorAt reads it)
  std::vector<Layer> layers;
  Corner* d_crn_infl = nullptr; bool crn_infl_valid = false;       // corners over the edge distances (inflation wave)
  uint8_t *d_infl_mask = nullptr, *d_zero_u8 = nullptr;
  float* d_infl_keyd = nullptr;
  uint32_t infl_steps = 0, infl_bands = 0; uint64_t infl_evals = 0; float infl_ms = 0.f, infl_ms_wave = 0.f;   // last inflation wave
  TilePlan* d_tplans = nullptr; uint32_t tplans_cap = 0;
  TCtl* h_tctl = nullptr;
  size_t tile_lds = 0, fin_lds = 0;
  bool use_graph = true;
  uint32_t* d_wide_prefix = nullptr; WideSched* d_wide_sched = nullptr; uint32_t wide_cap = 0;   // k_cvp_ctl -> k_step_wide
  float* d_vec3 = nullptr;                                           // mnav_vector_at after a paths-only batch
  uint32_t wide_groups = 1; hipStream_t stream_g[kWideGroupsMax] = {}; hipEvent_t ev_fork[kWideGroupsMax] = {};   // CVP batches in groups on their own streams ([0] unused / fork event)
  uint32_t cvp_wide_min_batch = 32;                                  // CVP batches of at least this many plans run k_step_wide
  float delta_user = 0.f, delta_auto = 0.f;
  Options opt;                                                       // mnav_options.h: read from the environment once, by mnav_create
  uint32_t max_steps_auto = 1u << 20;
  uint32_t* d_ring = nullptr; uint32_t ring_cap = 0; size_t ring_words = 0; struct AsyncCtl* h_actl = nullptr; uint32_t* d_parked = nullptr; size_t parked_words = 0;   // asynchronous tile engine: ticket ring, pinned copy of its control words
  uint32_t last_planner = 0, last_n = 0;
  std::vector<uint32_t> last_target; double last_offset = 0.0;   // Dijkstra: robot vertex per device slot, goal_dist_offset of the last call
  mnav_stats stats{};
  uint64_t algo_bytes = 0;
  hipEvent_t ev[8]{};
  hipEvent_t evc[2]{};         // bracket one graph replay (chunk of step/round launches)
  double ms_chunks = 0.0;      // sum of the bracketed chunk durations of the last call
  Ctl* d_ctl_pool = nullptr; uint32_t ctl_pool_cap = 0;      // contiguous control blocks: one D2H copy per chunk
  TCtl* d_tctl_pool = nullptr; uint32_t tctl_pool_cap = 0;
};

namespace {

#define MTRACE(msg) do { if (opt_on(ctx->opt.trace)) { fprintf(stderr, "[mnav] %9.3f ms %s:%d %s\n", 1e-3 * (double)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(), __func__, __LINE__, msg); fflush(stderr); } } while (0)
#define HIPCHK(call)                                                                               \
  do {                                                                                             \
    hipError_t e_ = (call);                                                                        \
    if (e_ != hipSuccess) {                                                                        \
      ctx->err = std::string(#call) + ": " + hipGetErrorString(e_) + " (" __FILE_NAME__ ":" + std::to_string(__LINE__) + ")"; \
      return -1;                                                                                   \
    }                                                                                              \
  } while (0)

// temporary device buffer of one call: freed on every way out (the HIPCHK early returns included)
template <class T>
struct DevTmp {
  T* p = nullptr;
  DevTmp() = default;
  DevTmp(const DevTmp&) = delete;
  DevTmp& operator=(const DevTmp&) = delete;
  ~DevTmp() { if (p) (void)hipFree(p); }
  operator T*() const { return p; }
  void** out() { return (void**)&p; }
};

template <class T>
int dev_upload(mnav_ctx* ctx, T** dptr, const T* host, size_t n)
{
  const size_t bytes = sizeof(T) * (n ? n : 1) + 64;               // tail slack: clamped vector loads may touch element 0 of an empty tile
  auto it = *dptr ? ctx->alloc_bytes.find((void*)*dptr) : ctx->alloc_bytes.end();
  if (!*dptr || it == ctx->alloc_bytes.end() || it->second != bytes) {   // same size as last time (cost re-uploads): keep the buffer
    if (*dptr) { ctx->alloc_bytes.erase((void*)*dptr); (void)hipFree(*dptr); *dptr = nullptr; }
    HIPCHK(hipMalloc((void**)dptr, bytes));
    ctx->alloc_bytes[(void*)*dptr] = bytes;
  }
  if (n && host) HIPCHK(hipMemcpyAsync(*dptr, host, sizeof(T) * n, hipMemcpyHostToDevice, ctx->stream));
  return 0;
}

float ev_ms(hipEvent_t a, hipEvent_t b);

void drop_layers(mnav_ctx* ctx)
{
  for (auto& L : ctx->layers) { (void)hipFree(L.cost); (void)hipFree(L.lethal); (void)hipFree(L.dist); (void)hipFree(L.vec); (void)hipFree(L.vstate); }
  ctx->layers.clear();
  (void)hipFree(ctx->d_crn_infl); (void)hipFree(ctx->d_infl_mask); (void)hipFree(ctx->d_zero_u8); (void)hipFree(ctx->d_infl_keyd);
  ctx->d_crn_infl = nullptr; ctx->d_infl_mask = nullptr; ctx->d_zero_u8 = nullptr; ctx->d_infl_keyd = nullptr; ctx->crn_infl_valid = false;
}

void free_slot(Slot& s)
{
  (void)hipFree(s.dist); (void)hipFree(s.tkey); (void)hipFree(s.dirn); (void)hipFree(s.vecmap);
  (void)hipFree(s.pred); (void)hipFree(s.cutf); (void)hipFree(s.stamp); (void)hipFree(s.dirty); (void)hipFree(s.list0); (void)hipFree(s.list1);
  (void)hipFree(s.wlist0); (void)hipFree(s.wlist1); (void)hipFree(s.wstamp);
  (void)hipFree(s.cnt);
  (void)hipFree(s.tpend0); (void)hipFree(s.tpend1); (void)hipFree(s.tlast); (void)hipFree(s.tcnt);
  s = Slot{};
}

void drop_graphs(mnav_ctx* ctx)
{
  for (auto& kv : ctx->graphs) (void)hipGraphExecDestroy(kv.second);
  ctx->graphs.clear();
  for (auto& kv : ctx->shard.graphs) (void)hipGraphExecDestroy(kv.second);
  ctx->shard.graphs.clear();
}

// the options that are mirrored in context fields (everything else is read from ctx->opt where it is used)
void apply_options(mnav_ctx* ctx)
{
  const Options& o = ctx->opt;
  ctx->use_graph = !opt_on(o.no_graph);
  if (opt_set(o.dijkstra_engine)) { const int e = (int)o.dijkstra_engine; ctx->dij_engine = (e == 0 || e == 1 || e == 5 || e == 6) ? e : 3; }
  else ctx->dij_engine = 3;                                           // unset (NaN): the built-in default, auto
  ctx->max_steps = opt_u32(o.max_steps, ctx->max_steps_auto);
  ctx->max_wall_s = opt_set(o.max_wall_s) ? o.max_wall_s : 120.0;
  ctx->cvp_verify = !opt_set(o.cvp_verify) || o.cvp_verify != 0.0;
  ctx->walk_max = opt_set(o.key_walk_max) ? (int)o.key_walk_max : kKeyWalkMax;
  ctx->descend_max = opt_set(o.descend_walk_max) ? (int)o.descend_walk_max : kDescendWalkMax;
  ctx->allow_lazy_paths = !opt_set(o.lazy_paths) || o.lazy_paths != 0.0;
  ctx->tile_band_user = opt_set(o.tile_band) ? (float)o.tile_band : 0.f;
  ctx->rounds_band_mult = opt_set(o.rounds_band_mult) ? (float)o.rounds_band_mult : 4.0f;
}

int ensure_plan_tables(mnav_ctx* ctx, uint32_t n);

int ensure_slots(mnav_ctx* ctx, uint32_t n, bool cvp, bool band, bool vec)
{
  const size_t V = ctx->V ? ctx->V : 1;
  while (ctx->slots.size() < n) {
    Slot s;
    HIPCHK(hipMalloc((void**)&s.dist, 4 * V)); HIPCHK(hipMalloc((void**)&s.pred, 4 * V));   // 8 B per vertex and plan ...
    HIPCHK(hipMalloc((void**)&s.cnt, 4 * sizeof(Cnt)));                                       // 3 rotating + sticky flags
    ctx->slots.push_back(s);
  }
  for (uint32_t i = 0; i < n; ++i) {                                   // ... the rest only for the paths that use it
    Slot& s = ctx->slots[i];
    if (band && !s.band_ready) {                                       // work lists of the band/gather steps
      HIPCHK(hipMalloc((void**)&s.stamp, 4 * V)); HIPCHK(hipMalloc((void**)&s.dirty, 4 * V));
      HIPCHK(hipMalloc((void**)&s.list0, 4 * V)); HIPCHK(hipMalloc((void**)&s.list1, 4 * V));
      HIPCHK(hipMalloc((void**)&s.wlist0, 4 * V)); HIPCHK(hipMalloc((void**)&s.wlist1, 4 * V)); HIPCHK(hipMalloc((void**)&s.wstamp, 4 * V));
      s.band_ready = true;
    }
    if (vec && !s.vecmap) HIPCHK(hipMalloc((void**)&s.vecmap, 12 * V));
  }
  if (cvp)
    for (uint32_t i = 0; i < n; ++i) {
      Slot& s = ctx->slots[i];
      if (!s.cvp_ready) {
        HIPCHK(hipMalloc((void**)&s.tkey, sizeof(PopKey) * V)); HIPCHK(hipMalloc((void**)&s.dirn, 4 * V));
        HIPCHK(hipMalloc((void**)&s.cutf, 4 * V));
        s.cvp_ready = true;
      }
    }
  if (ctx->ctl_pool_cap < n) {
    if (ctx->d_ctl_pool) (void)hipFree(ctx->d_ctl_pool);
    ctx->d_ctl_pool = nullptr;
    HIPCHK(hipMalloc((void**)&ctx->d_ctl_pool, 2 * sizeof(Ctl) * n));
    ctx->ctl_pool_cap = n;
  }
  for (uint32_t i = 0; i < n; ++i) ctx->slots[i].ctl = ctx->d_ctl_pool + 2 * i;
  return ensure_plan_tables(ctx, n);
}

// per-plan tables shared by all engines (plan descriptors, results)
int ensure_plan_tables(mnav_ctx* ctx, uint32_t n)
{
  if (ctx->plans_cap < n) {
    if (ctx->d_plans) (void)hipFree(ctx->d_plans);
    if (ctx->d_res) (void)hipFree(ctx->d_res);
    if (ctx->h_res) (void)hipHostFree(ctx->h_res);
    if (ctx->h_ctl) (void)hipHostFree(ctx->h_ctl);
    if (ctx->d_vecptrs) (void)hipFree(ctx->d_vecptrs);
    ctx->d_plans = nullptr; ctx->d_res = nullptr; ctx->h_res = nullptr; ctx->h_ctl = nullptr; ctx->d_vecptrs = nullptr;
    drop_graphs(ctx);   // graphs captured the old d_plans pointer
    HIPCHK(hipMalloc((void**)&ctx->d_plans, sizeof(Plan) * n));
    HIPCHK(hipMalloc((void**)&ctx->d_res, sizeof(PlanResult) * n));
    HIPCHK(hipHostMalloc((void**)&ctx->h_res, sizeof(PlanResult) * n, hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void**)&ctx->h_ctl, sizeof(Ctl) * 2 * n, hipHostMallocDefault));
    HIPCHK(hipMalloc((void**)&ctx->d_vecptrs, sizeof(float*) * n));
    ctx->plans_cap = n;
  }
  return 0;
}

// Vertex paths of a batch: n rows of `stride` ids.  A path has ~1.4 sqrt(V) hops on a terrain, so rows of
// 16 sqrt(V) + 1024 ids hold it with a wide margin (84 MB instead of 20 GB for 5120 plans on the 1M mesh); a path
// that does not fit makes the path kernels report kPathOverflow with the full length, and only those plans are walked
// again into rows of exactly that size.
uint32_t default_path_stride(const mnav_ctx* ctx)
{
  const double s = 16.0 * std::sqrt((double)ctx->V) + 1024.0;
  return (uint32_t)std::min<double>(s, (double)(ctx->V ? ctx->V : 1));
}
int ensure_paths(mnav_ctx* ctx, uint32_t n, uint32_t stride)
{
  const size_t words = (size_t)n * (stride ? stride : 1);
  if (ctx->paths_words < words) {
    if (ctx->d_paths) (void)hipFree(ctx->d_paths);
    ctx->d_paths = nullptr; ctx->paths_words = 0;
    HIPCHK(hipMalloc((void**)&ctx->d_paths, sizeof(uint32_t) * words));
    ctx->paths_words = words;
  }
  ctx->path_stride = stride;
  return 0;
}
int ensure_paths(mnav_ctx* ctx, uint32_t n) { return ensure_paths(ctx, n, default_path_stride(ctx)); }

uint32_t blocks_per_plan(const mnav_ctx* ctx)
{
  // the work list of a planar mesh is O(sqrt(V)) long; 8 entries per wave and round.  Measured on the 1M mesh (CVP,
  // MI355X): 500 waves per plan are as fast as 1500 for a single plan (41.8 vs 41.4 ms) and 25 % faster in a batch of
  // 128 (225 vs 178 plans/s: fewer idle waves to dispatch per step); 256 waves cost 5 % / 10 %.
  const double want = 4.0 * std::sqrt((double)ctx->V) / kGroupsPerWave;
  uint32_t g = (uint32_t)std::ceil(want);
  if (opt_set(ctx->opt.blocks_per_plan)) g = opt_u32(ctx->opt.blocks_per_plan, g);
  if (g < 4) g = 4;
  if (g > 4096) g = 4096;
  return g;
}

template <uint32_t PLANNER>
int launch_steps(mnav_ctx* ctx, uint32_t n, uint32_t G, int count, bool wide)
{
  const bool fork = wide && ctx->wide_groups > 1;
  if (fork) {                                                          // the other streams join (the capture of) the first
    HIPCHK(hipEventRecord(ctx->ev_fork[0], ctx->stream));
    for (uint32_t g = 1; g < ctx->wide_groups; ++g) HIPCHK(hipStreamWaitEvent(ctx->stream_g[g], ctx->ev_fork[0], 0));
  }
  for (int j = 0; j < count; ++j) {
    if (wide) {
      // per group of plans: controller + shares, the wide kernel with as many waves as stay resident, and the 8-lane kernel for the
      // plans in a band cut.  The groups run on two streams (two branches of the captured graph): a step of one group is ~0.6
      // rounds of the resident waves, so where one group's last round leaves the machine half empty the other group's step fills it
      uint32_t off = 0, poff = 0;
      for (uint32_t g = 0; g < ctx->wide_groups; ++g) {
        const uint32_t ng = (n * (g + 1u)) / ctx->wide_groups - off;
        hipStream_t st = g == 0 ? ctx->stream : ctx->stream_g[g];
        if (ng) {
          hipLaunchKernelGGL(k_cvp_ctl, dim3(1), dim3(256), 0, st, ctx->d_plans + off, ng, j % 6, ctx->d_wide_prefix + poff, ctx->d_wide_sched + g);
          hipLaunchKernelGGL(k_step_wide, dim3(G), dim3(kWave), 0, st, ctx->d_plans + off, ng, j % 6, ctx->d_wide_prefix + poff, ctx->d_wide_sched + g);
          hipLaunchKernelGGL(k_step_repair, dim3(blocks_per_plan(ctx), kRepairRows), dim3(kWave), 0, st, ctx->d_plans + off, j % 6,
                             ctx->d_wide_prefix + poff + ng + 1u, ctx->d_wide_sched + g);
        }
        off += ng; poff += 2u * ng + 2u;
      }
    }
    else hipLaunchKernelGGL(k_step<PLANNER>, dim3(G, n), dim3(kWave), 0, ctx->stream, ctx->d_plans, j % 6);
  }
  if (fork) {
    for (uint32_t g = 1; g < ctx->wide_groups; ++g) {
      HIPCHK(hipEventRecord(ctx->ev_fork[g], ctx->stream_g[g]));
      HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_fork[g], 0));
    }
  }
  HIPCHK(hipGetLastError());
  return 0;
}

__global__ void k_stamp_to_host(unsigned long long* __restrict__ dst_host) { *dst_host = wall_clock64(); }   // (trace: the device's 100 MHz clock)

template <uint32_t PLANNER>
int run_chunk(mnav_ctx* ctx, uint32_t n, uint32_t G, bool wide)
{
  if (!ctx->use_graph) return launch_steps<PLANNER>(ctx, n, G, opt_set(ctx->opt.debug_chunk) ? (int)ctx->opt.debug_chunk : kChunk, wide);   // debug: finer control-block trace
  const uint64_t key = ((uint64_t)PLANNER << 60) | ((uint64_t)(wide ? ctx->wide_groups : 0u) << 57) | ((uint64_t)n << 32) | G;
  auto it = ctx->graphs.find(key);
  if (it == ctx->graphs.end()) {
    hipGraph_t g = nullptr;
    HIPCHK(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    const int rc = launch_steps<PLANNER>(ctx, n, G, kChunk, wide);
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

int materialize(mnav_ctx* ctx, bool cvp, double cost_limit)
{
  const uint32_t V = ctx->V;
  const uint32_t gb = (V + kBlock - 1) / kBlock;
  if (!cvp) {
    if (ctx->nbr_valid && ctx->nbr_limit == cost_limit) return 0;
    if (!ctx->d_nbr) HIPCHK(hipMalloc((void**)&ctx->d_nbr, sizeof(Nbr) * (size_t)(ctx->E ? 2 * (size_t)ctx->E : 1)));
    hipLaunchKernelGGL(k_build_nbr, dim3(gb ? gb : 1), dim3(kBlock), 0, ctx->stream, V, ctx->d_row_ptr, ctx->d_nbr_u,
                       ctx->d_nbr_e, ctx->d_w, ctx->d_cost, ctx->d_invalid, cost_limit, ctx->d_nbr);
    HIPCHK(hipGetLastError());
    ctx->nbr_limit = cost_limit; ctx->nbr_valid = true; ctx->tw_valid = false; ctx->tb.w_valid = false;
  } else {
    if (ctx->crn_valid && ctx->crn_limit == cost_limit) return 0;
    if (!ctx->d_crn) HIPCHK(hipMalloc((void**)&ctx->d_crn, sizeof(Corner) * (size_t)(ctx->F ? 3 * (size_t)ctx->F : 1)));
    if (!ctx->d_blocked) HIPCHK(hipMalloc((void**)&ctx->d_blocked, V ? V : 1));
    hipLaunchKernelGGL(k_build_crn, dim3(gb ? gb : 1), dim3(kBlock), 0, ctx->stream, V, ctx->d_crn_ptr, ctx->d_crn_idx,
                       ctx->d_w, ctx->d_cost, ctx->d_invalid, cost_limit, ctx->d_crn, ctx->d_blocked);
    HIPCHK(hipGetLastError());
    ctx->crn_limit = cost_limit; ctx->crn_valid = true;
  }
  return 0;
}

// k_cvp_verify until a sweep finds every vertex at its fixed point: sweeps that find stale deep-cascade members store
// the re-evaluated state (verify_entry) and are followed by another one; the last allowed sweep only checks.
int verify_sweeps(mnav_ctx* ctx, uint32_t n)
{
  if (!ctx->d_verify_any) HIPCHK(hipMalloc((void**)&ctx->d_verify_any, 4));
  uint32_t gv = (ctx->V / kGroupsPerWave + 3) / 4;                  // ~4 vertices per 8-lane group
  if (gv < 1) gv = 1;
  if (gv > 8192) gv = 8192;
  ctx->verify_sweeps_used = 0;
  for (int sweep = 0; sweep <= kVerifySweeps; ++sweep) {
    uint32_t any = 0;
    HIPCHK(hipMemsetAsync(ctx->d_verify_any, 0, 4, ctx->stream));
    hipLaunchKernelGGL(k_flags_reset, dim3(n), dim3(64), 0, ctx->stream, ctx->d_plans);
    hipLaunchKernelGGL(k_cvp_verify, dim3(gv, n), dim3(kWave), 0, ctx->stream, ctx->d_plans, sweep < kVerifySweeps ? 1 : 0, ctx->d_verify_any);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(&any, ctx->d_verify_any, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (any && n == 1 && opt_on(ctx->opt.trace)) {
      Cnt f{};
      (void)hipMemcpy(&f, ctx->slots[0].cnt + 3, sizeof(Cnt), hipMemcpyDeviceToHost);
      fprintf(stderr, "[mnav] verification sweep %d: %u vertices not at their fixed point\n", sweep, f.changed);
    }
    if (!any) break;
    ++ctx->verify_sweeps_used;
  }
  if (ctx->verify_sweeps_used > (uint32_t)kVerifySweeps && n == 1) {
    // the concurrent sweeps keep a tie cluster flipping: list what is off (fix == 2), repair it one vertex at a time, check again
    uint32_t any = 0;
    HIPCHK(hipMemsetAsync(ctx->d_verify_any, 0, 4, ctx->stream));
    hipLaunchKernelGGL(k_flags_reset, dim3(n), dim3(64), 0, ctx->stream, ctx->d_plans);
    hipLaunchKernelGGL(k_cvp_verify, dim3(gv, n), dim3(kWave), 0, ctx->stream, ctx->d_plans, 2, ctx->d_verify_any);
    hipLaunchKernelGGL(k_verify_serial, dim3(n), dim3(kWave), 0, ctx->stream, ctx->d_plans, 200000u);
    HIPCHK(hipMemsetAsync(ctx->d_verify_any, 0, 4, ctx->stream));
    hipLaunchKernelGGL(k_flags_reset, dim3(n), dim3(64), 0, ctx->stream, ctx->d_plans);
    hipLaunchKernelGGL(k_cvp_verify, dim3(gv, n), dim3(kWave), 0, ctx->stream, ctx->d_plans, 0, ctx->d_verify_any);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(&any, ctx->d_verify_any, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (opt_on(ctx->opt.trace)) fprintf(stderr, "[mnav] verification: serial repair, %s\n", any ? "still off" : "clean");
  }
  if (opt_on(ctx->opt.trace)) fprintf(stderr, "[mnav] verification: %u fixing sweep(s)\n", ctx->verify_sweeps_used);
  return 0;
}

#include "mnav_engines_host.h"

#include "mnav_tb_host.h"

float ev_ms(hipEvent_t a, hipEvent_t b)
{
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, a, b) != hipSuccess) return 0.f;
  return ms;
}

void finish_stats(mnav_ctx* ctx, uint32_t n, bool cvp)
{
  mnav_stats& st = ctx->stats;
  st.n_plans = n;
  st.steps = 0; st.bands = 0; st.armed = 0; st.goal_dist = INFINITY; st.settled = 0; st.evals = 0; st.band_shrinks = 0; st.band_cuts = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const PlanResult& r = ctx->h_res[i];
    if (r.steps > st.steps) st.steps = r.steps;
    if (r.bands > st.bands) st.bands = r.bands;
    st.band_shrinks += r.shrinks & 0xFFFFu; st.band_cuts += r.shrinks >> 16;
    st.armed += r.armed;
    if (i == 0) st.goal_dist = r.goal_dist;
    st.settled += r.settled;
    st.evals += r.evals;
  }
  st.ms_init = ev_ms(ctx->ev[1], ctx->ev[2]);
  st.ms_propagation = ev_ms(ctx->ev[2], ctx->ev[3]);
  st.ms_path = ev_ms(ctx->ev[3], ctx->ev[4]);
  st.ms_vector_map = ev_ms(ctx->ev[4], ctx->ev[5]);
  st.ms_download = ev_ms(ctx->ev[5], ctx->ev[6]);
  st.ms_total = ev_ms(ctx->ev[0], ctx->ev[6]);
  st.ms_step_kernels = (float)ctx->ms_chunks;
  // early-exit variant = settled vertices and their incident edges / faces
  const double V = ctx->V ? ctx->V : 1;
  const double frac = (double)st.settled / V;   // summed over plans
  if (!cvp) ctx->algo_bytes = (uint64_t)(24.0 * st.settled + 24.0 * frac * ctx->E);
  else ctx->algo_bytes = (uint64_t)(32.0 * st.settled + 68.0 * frac * ctx->F);
}

int check_ready(mnav_ctx* ctx)
{
  if (!ctx) return -1;
  if (!ctx->have_mesh) { ctx->err = "mnav_upload_mesh has not been called"; return -1; }
  if (!ctx->have_costs) { ctx->err = "mnav_upload_costs has not been called"; return -1; }
  return 0;
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

mnav_ctx* mnav_create(int device)
{
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return nullptr;
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  mnav_ctx* ctx = new mnav_ctx();
  ctx->device = device;
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return nullptr; }
  for (auto& e : ctx->ev)
    if (hipEventCreate(&e) != hipSuccess) { delete ctx; return nullptr; }
  for (auto& e : ctx->evc)
    if (hipEventCreate(&e) != hipSuccess) { delete ctx; return nullptr; }
  if (hipMalloc((void**)&ctx->d_seed_pos, 3 * sizeof(float)) != hipSuccess) { delete ctx; return nullptr; }
  if (hipMalloc((void**)&ctx->d_cancel, 64) != hipSuccess || hipMemset(ctx->d_cancel, 0, 64) != hipSuccess ||
      hipHostMalloc((void**)&ctx->h_one, 64, hipHostMallocDefault) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->cancel_stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return nullptr; }
  *ctx->h_one = 1u;
  if (hipHostMalloc((void**)&ctx->h_actl, sizeof(AsyncCtl), hipHostMallocDefault) != hipSuccess) { delete ctx; return nullptr; }
  ctx->opt.from_environment();                                        // the ONLY look at the environment (mnav_options.h)
  apply_options(ctx);
  return ctx;
}

void mnav_destroy(mnav_ctx* ctx)
{
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  drop_graphs(ctx);
  tb_free(ctx);
  for (auto& s : ctx->slots) free_slot(s);
  drop_layers(ctx);
  (void)hipFree(ctx->d_row_ptr); (void)hipFree(ctx->d_nbr_u); (void)hipFree(ctx->d_nbr_e); (void)hipFree(ctx->d_crn_ptr);
  (void)hipFree(ctx->d_edge_vtx); (void)hipFree(ctx->d_crn_idx); (void)hipFree(ctx->d_crn_walk); (void)hipFree(ctx->d_xyz); (void)hipFree(ctx->d_nrm);
  (void)hipFree(ctx->d_cost); (void)hipFree(ctx->d_w); (void)hipFree(ctx->d_edge_dist); (void)hipFree(ctx->d_invalid);
  (void)hipFree(ctx->d_nbr); (void)hipFree(ctx->d_crn); (void)hipFree(ctx->d_blocked); (void)hipFree(ctx->d_plans);
  (void)hipFree(ctx->d_res); (void)hipFree(ctx->d_vecptrs); (void)hipFree(ctx->d_paths); (void)hipFree(ctx->d_seed_pos);
  (void)hipFree(ctx->d_t_vptr); (void)hipFree(ctx->d_t_verts); (void)hipFree(ctx->d_t_hptr); (void)hipFree(ctx->d_t_halo_verts);
  (void)hipFree(ctx->d_t_halo_tile); (void)hipFree(ctx->d_t_eptr); (void)hipFree(ctx->d_t_src); (void)hipFree(ctx->d_vert_tile);
  (void)hipFree(ctx->d_t_rptr); (void)hipFree(ctx->d_mismatch); (void)hipFree(ctx->d_t_rowptr); (void)hipFree(ctx->d_t_col); (void)hipFree(ctx->d_t_tw);
  (void)hipFree(ctx->d_tplans);
  (void)hipFree(ctx->shard.d_iface_vert); (void)hipFree(ctx->shard.d_iface_owner); (void)hipFree(ctx->shard.d_wake_ptr); (void)hipFree(ctx->shard.d_wake_tile);
  (void)hipFree(ctx->d_wide_prefix); (void)hipFree(ctx->d_wide_sched); (void)hipFree(ctx->d_vec3);
  for (auto& st : ctx->stream_g) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
  for (auto& e : ctx->ev_fork) if (e) (void)hipEventDestroy(e);
  (void)hipFree(ctx->shard.d_owned); (void)hipFree(ctx->shard.d_changed); (void)hipFree(ctx->shard.d_minpend); (void)hipFree(ctx->shard.d_walk);
  if (ctx->cancel_stream) { (void)hipStreamSynchronize(ctx->cancel_stream); (void)hipStreamDestroy(ctx->cancel_stream); }
  if (ctx->h_one) (void)hipHostFree(ctx->h_one);
  (void)hipFree(ctx->d_cancel); (void)hipFree(ctx->d_verify_any); (void)hipFree(ctx->d_ring); (void)hipFree(ctx->d_parked); if (ctx->h_actl) (void)hipHostFree(ctx->h_actl);
  (void)hipFree(ctx->d_over); (void)hipFree(ctx->d_over_off); (void)hipFree(ctx->d_over_cap);
  (void)hipFree(ctx->d_pack); (void)hipFree(ctx->d_pack_meta); if (ctx->h_pack) (void)hipHostFree(ctx->h_pack);
  if (ctx->h_tctl) (void)hipHostFree(ctx->h_tctl);
  if (ctx->h_res) (void)hipHostFree(ctx->h_res);
  if (ctx->h_ctl) (void)hipHostFree(ctx->h_ctl);
  for (auto& e : ctx->ev) if (e) (void)hipEventDestroy(e);
  for (auto& e : ctx->evc) if (e) (void)hipEventDestroy(e);
  for (auto& e : ctx->ev_link) if (e) (void)hipEventDestroy(e);
  (void)hipFree(ctx->d_ctl_pool); (void)hipFree(ctx->d_tctl_pool);
  (void)hipFree(ctx->d_faces); (void)hipFree(ctx->d_vf_ptr); (void)hipFree(ctx->d_vf); (void)hipFree(ctx->d_walk_pos); (void)hipFree(ctx->d_walk_face);
  (void)hipFree(ctx->d_walk_jobs); (void)hipFree(ctx->d_walk_ctl);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char* mnav_last_error(const mnav_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int mnav_set_face_circulation(mnav_ctx* ctx, uint32_t V, uint32_t F, const uint32_t* vf_ptr, const uint32_t* vf)
{
  if (!ctx) return -1;
  ctx->err.clear();
  ctx->circ = FaceCirculation();
  if (!vf_ptr || !vf) return 0;                                  // back to the built-in half-edge replay
  ctx->circ.ptr.assign(vf_ptr, vf_ptr + (size_t)V + 1);
  ctx->circ.faces.assign(vf, vf + 3 * (size_t)F);
  ctx->circ.ok = true;
  return 0;
}

int mnav_upload_mesh(mnav_ctx* ctx, uint32_t V, uint32_t F, uint32_t E, const float* xyz, const uint32_t* face_vtx,
                     const uint32_t* edge_vtx, const float* vertex_normals)
{
  if (!ctx) return -1;
  ctx->err.clear();
  if ((V && !xyz) || (F && !face_vtx) || (E && !edge_vtx)) { ctx->err = "null mesh array"; return -1; }
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return -1; }
  HostTopology t;
  try {
    if (!ctx->circ.ptr.empty()) {
      if (ctx->circ.ptr.size() != (size_t)V + 1 || ctx->circ.faces.size() != 3 * (size_t)F || ctx->circ.ptr[V] != 3 * (size_t)F)
        throw std::invalid_argument("face circulation does not fit this mesh");
      t = build_topology(V, F, E, face_vtx, edge_vtx, &ctx->circ);
    } else {
      t = build_topology(V, F, E, face_vtx, edge_vtx);
    }
  }
  catch (const std::exception& ex) { ctx->err = ex.what(); return -2; }
  (void)hipStreamSynchronize(ctx->stream);
  for (auto& s : ctx->slots) free_slot(s);
  ctx->slots.clear();
  drop_layers(ctx);
  if (ctx->d_edge_dist) { ctx->alloc_bytes.erase((void*)ctx->d_edge_dist); (void)hipFree(ctx->d_edge_dist); ctx->d_edge_dist = nullptr; }   // belongs to the old mesh
  drop_graphs(ctx);
  (void)hipFree(ctx->d_paths); ctx->d_paths = nullptr; ctx->paths_words = 0;
  (void)hipFree(ctx->d_nbr); ctx->d_nbr = nullptr; (void)hipFree(ctx->d_crn); ctx->d_crn = nullptr;
  (void)hipFree(ctx->d_blocked); ctx->d_blocked = nullptr;
  (void)hipFree(ctx->d_cost); ctx->d_cost = nullptr; (void)hipFree(ctx->d_w); ctx->d_w = nullptr;
  (void)hipFree(ctx->d_invalid); ctx->d_invalid = nullptr; (void)hipFree(ctx->d_edge_dist); ctx->d_edge_dist = nullptr;
  ctx->nbr_valid = ctx->crn_valid = false; ctx->have_costs = false;
  ctx->V = V; ctx->F = F; ctx->E = E;
  {
    // a planar wavefront needs a few steps per hop of the mesh diameter ~ sqrt(V); generous cap
    const double cap = 400.0 * std::sqrt((double)V) + 20000.0;
    ctx->max_steps = cap > 2.0e9 ? 2000000000u : (uint32_t)cap;
    ctx->max_steps_auto = ctx->max_steps;
    apply_options(ctx);
  }
  ctx->h_xyz.assign(xyz, xyz + 3 * (size_t)V);
  ctx->h_faces.assign(face_vtx, face_vtx + 3 * (size_t)F);
  ctx->h_row_ptr = t.row_ptr; ctx->h_nbr_u = t.nbr_u;
  ctx->h_vf_ptr = std::move(t.vf_ptr); ctx->h_vf = std::move(t.vf); ctx->walk_mesh_valid = false;
  tb_free(ctx);
  if (dev_upload(ctx, &ctx->d_row_ptr, t.row_ptr.data(), t.row_ptr.size())) return -1;
  if (dev_upload(ctx, &ctx->d_nbr_u, t.nbr_u.data(), t.nbr_u.size())) return -1;
  if (dev_upload(ctx, &ctx->d_nbr_e, t.nbr_e.data(), t.nbr_e.size())) return -1;
  if (dev_upload(ctx, &ctx->d_crn_ptr, t.crn_ptr.data(), t.crn_ptr.size())) return -1;
  if (dev_upload(ctx, &ctx->d_edge_vtx, edge_vtx, 2 * (size_t)E)) return -1;
  std::vector<CornerIdx> ci(t.crn_v1.size());
  for (size_t i = 0; i < ci.size(); ++i) ci[i] = CornerIdx{ t.crn_v1[i], t.crn_v2[i], t.crn_ea[i], t.crn_eb[i], t.crn_ec[i], t.crn_face[i] };
  if (dev_upload(ctx, &ctx->d_crn_walk, t.crn_walk.data(), t.crn_walk.size())) return -1;
  if (dev_upload(ctx, &ctx->d_crn_idx, ci.data(), ci.size())) return -1;
  if (dev_upload(ctx, &ctx->d_xyz, xyz, 3 * (size_t)V)) return -1;
  ctx->have_normals = vertex_normals != nullptr;
  if (dev_upload(ctx, &ctx->d_nrm, vertex_normals, vertex_normals ? 3 * (size_t)V : 0)) return -1;
  // LDS tiles of the SSSP engine
  {
    ctx->tile_size = opt_u32(ctx->opt.tile_size, 512u);
    if (ctx->tile_size < 64) ctx->tile_size = 64;
    if (ctx->tile_size > (uint32_t)(kTileBlock * kTileVpt)) ctx->tile_size = kTileBlock * kTileVpt;
    HostTiles T;
    for (;;) {
      try { T = build_tiles(t, xyz, ctx->tile_size); }
      catch (const std::exception& ex) { ctx->err = ex.what(); return -2; }
      ctx->tile_lds = tile_lds_bytes(T.max_nv, T.max_nh, T.max_ne);
      ctx->fin_lds = finalize_lds_bytes(T.max_nv, T.max_nh, T.max_ne);
      if (ctx->fin_lds <= 150 * 1024 || ctx->tile_size <= 64) break;
      ctx->tile_size /= 2;                       // irregular mesh: shrink until a tile fits the 160 KiB LDS
    }
    if (ctx->fin_lds > 160 * 1024) { ctx->err = "mesh valence too high for the LDS tile engine"; return -2; }
    if (ctx->fin_lds > 64 * 1024)
      HIPCHK(hipFuncSetAttribute((const void*)k_dij_finalize, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->fin_lds));
    if (ctx->tile_lds > 64 * 1024)
      HIPCHK(hipFuncSetAttribute((const void*)k_tile_round, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->tile_lds));
    if (ctx->tile_lds > 64 * 1024) {
      HIPCHK(hipFuncSetAttribute((const void*)k_plan_async<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->tile_lds));
      HIPCHK(hipFuncSetAttribute((const void*)k_plan_async<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->tile_lds));
      HIPCHK(hipFuncSetAttribute((const void*)k_plan_async<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->tile_lds));
    }
    if (opt_on(ctx->opt.verbose))
      fprintf(stderr, "[mnav] tiles: size %u, %u tiles, max owned %u, halo %u, edges %u -> LDS %zu B (solve), %zu B (finalize)\n",
              ctx->tile_size, T.ntiles, T.max_nv, T.max_nh, T.max_ne, ctx->tile_lds, ctx->fin_lds);
    (void)hipFree(ctx->d_t_tw); ctx->d_t_tw = nullptr; ctx->tw_valid = false;
    ctx->t_nnz = (uint32_t)T.col.size();
    if (dev_upload(ctx, &ctx->d_t_vptr, T.vptr.data(), T.vptr.size())) return -1;
    if (dev_upload(ctx, &ctx->d_t_verts, T.verts.data(), T.verts.size())) return -1;
    if (dev_upload(ctx, &ctx->d_t_hptr, T.hptr.data(), T.hptr.size())) return -1;
    if (dev_upload(ctx, &ctx->d_t_halo_verts, T.halo_verts.data(), T.halo_verts.size())) return -1;
    if (dev_upload(ctx, &ctx->d_t_halo_tile, T.halo_tile.data(), T.halo_tile.size())) return -1;
    if (dev_upload(ctx, &ctx->d_t_eptr, T.eptr.data(), T.eptr.size())) return -1;
    if (dev_upload(ctx, &ctx->d_t_rptr, T.rptr.data(), T.rptr.size())) return -1;
    if (dev_upload(ctx, &ctx->d_t_rowptr, T.rowptr.data(), T.rowptr.size())) return -1;
    if (dev_upload(ctx, &ctx->d_t_col, T.col.data(), T.col.size())) return -1;
    if (dev_upload(ctx, &ctx->d_t_src, T.src.data(), T.src.size())) return -1;
    if (dev_upload(ctx, &ctx->d_vert_tile, T.vert_tile.data(), T.vert_tile.size())) return -1;
    if (dev_upload(ctx, &ctx->d_t_tw, (const float*)nullptr, T.col.size())) return -1;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    // keep the sizes, and the small per-vertex maps the partitioner of mnav_shard_setup needs, on the host
    T.halo_tile.clear(); T.halo_tile.shrink_to_fit(); T.rowptr.clear(); T.rowptr.shrink_to_fit();
    T.col.clear(); T.col.shrink_to_fit(); T.src.clear(); T.src.shrink_to_fit();
    ctx->tiles_meta = std::move(T);
    ctx->shard.ready = false;
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));   // host vectors go out of scope
  ctx->have_mesh = true;
  return 0;
}

static void auto_delta(mnav_ctx* ctx, const float* w, uint32_t E)
{
  double s = 0; uint64_t c = 0;
  for (uint32_t e = 0; e < E; ++e) if (std::isfinite(w[e])) { s += w[e]; ++c; }
  ctx->delta_auto = c ? (float)(3.0 * s / (double)c) : 1.0f;
  if (!(ctx->delta_auto > 0.f)) ctx->delta_auto = 1.0f;
  // tile band ~ the potential difference across one tile (sqrt(tile_size) mean edges)
  ctx->tile_band_auto = (ctx->delta_auto / 3.0f) * std::sqrt((float)ctx->tile_size);
}

int mnav_upload_costs(mnav_ctx* ctx, const float* vertex_costs, const float* edge_weights, const uint8_t* invalid)
{
  if (!ctx) return -1;
  ctx->err.clear();
  if (!ctx->have_mesh) { ctx->err = "mnav_upload_mesh has not been called"; return -1; }
  if ((ctx->V && !vertex_costs) || (ctx->E && !edge_weights)) { ctx->err = "null cost array"; return -1; }
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return -1; }
  if (dev_upload(ctx, &ctx->d_cost, vertex_costs, ctx->V)) return -1;
  if (dev_upload(ctx, &ctx->d_w, edge_weights, ctx->E)) return -1;
  std::vector<uint8_t> zero;
  if (!invalid) { zero.assign(ctx->V ? ctx->V : 1, 0); invalid = zero.data(); }
  if (dev_upload(ctx, &ctx->d_invalid, invalid, ctx->V)) return -1;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->h_cost.assign(vertex_costs, vertex_costs + ctx->V);
  ctx->h_invalid.assign(invalid, invalid + ctx->V);
  auto_delta(ctx, edge_weights, ctx->E);
  ctx->nbr_valid = ctx->crn_valid = false;
  ctx->have_costs = true;
  ctx->edge_cost_factor = 0.0;                                       // weights came from the caller: mnav_update_costs only touches vertex costs
  return 0;
}

// device pass shared by mnav_compute_edge_weights / mnav_combine_costs: d_cost and d_edge_dist are resident, d_w is
// (re)computed; host mirrors (cost for the seed cut-offs, mean weight for the band widths) are refreshed
static int edge_weight_pass(mnav_ctx* ctx, double edge_cost_factor, const uint8_t* invalid, float* vertex_costs_out, float* edge_weights_out)
{
  if (dev_upload(ctx, &ctx->d_w, (const float*)nullptr, ctx->E)) return -1;
  std::vector<uint8_t> zero;
  if (!invalid) { zero.assign(ctx->V ? ctx->V : 1, 0); invalid = zero.data(); }
  if (dev_upload(ctx, &ctx->d_invalid, invalid, ctx->V)) return -1;
  const uint32_t gb = (ctx->E + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(k_edge_weights, dim3(gb ? gb : 1), dim3(kBlock), 0, ctx->stream, ctx->E, ctx->d_edge_vtx, ctx->d_edge_dist,
                     ctx->d_cost, edge_cost_factor, ctx->d_w);
  HIPCHK(hipGetLastError());
  std::vector<float> w(ctx->E ? ctx->E : 1);
  ctx->h_cost.resize(ctx->V);
  HIPCHK(hipMemcpyAsync(w.data(), ctx->d_w, sizeof(float) * ctx->E, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->h_cost.data(), ctx->d_cost, sizeof(float) * ctx->V, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (edge_weights_out) memcpy(edge_weights_out, w.data(), sizeof(float) * ctx->E);
  if (vertex_costs_out) memcpy(vertex_costs_out, ctx->h_cost.data(), sizeof(float) * ctx->V);
  ctx->h_invalid.assign(invalid, invalid + ctx->V);
  auto_delta(ctx, w.data(), ctx->E);
  ctx->nbr_valid = ctx->crn_valid = false;
  ctx->edge_cost_factor = edge_cost_factor;
  ctx->have_costs = true;
  return 0;
}

int mnav_compute_edge_weights(mnav_ctx* ctx, const float* vertex_costs, const float* edge_distances, double edge_cost_factor,
                              const uint8_t* invalid, float* edge_weights_out)
{
  if (!ctx) return -1;
  ctx->err.clear();
  if (!ctx->have_mesh) { ctx->err = "mnav_upload_mesh has not been called"; return -1; }
  if ((ctx->V && !vertex_costs) || (ctx->E && !edge_distances)) { ctx->err = "null cost array"; return -1; }
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return -1; }
  if (dev_upload(ctx, &ctx->d_cost, vertex_costs, ctx->V)) return -1;
  if (dev_upload(ctx, &ctx->d_edge_dist, edge_distances, ctx->E)) return -1;
  ctx->crn_infl_valid = false;
  return edge_weight_pass(ctx, edge_cost_factor, invalid, nullptr, edge_weights_out);
}

int mnav_combine_costs(mnav_ctx* ctx, int mode, uint32_t n_layers, const float* const* layer_costs, const float* weights,
                       const float* edge_distances, double edge_cost_factor, const uint8_t* invalid, float* vertex_costs_out,
                       float* edge_weights_out)
{
  if (!ctx) return -1;
  ctx->err.clear();
  if (!ctx->have_mesh) { ctx->err = "mnav_upload_mesh has not been called"; return -1; }
  if (mode != 0 && mode != 1) { ctx->err = "combination mode must be 0 (max) or 1 (weighted sum)"; return -1; }
  if ((n_layers && !layer_costs) || (mode == 1 && n_layers && !weights) || (ctx->E && !edge_distances && !ctx->d_edge_dist)) { ctx->err = "null input array"; return -1; }
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return -1; }
  const uint32_t V = ctx->V;
  DevTmp<float> d_layers, d_wts;
  HIPCHK(hipMalloc(d_layers.out(), sizeof(float) * ((size_t)n_layers * V + 1)));
  HIPCHK(hipMalloc(d_wts.out(), sizeof(float) * (n_layers + 1)));
  int rc = 0;
  for (uint32_t l = 0; l < n_layers && rc == 0; ++l) {
    if (!layer_costs[l]) { ctx->err = "null layer"; rc = -1; break; }
    if (hipMemcpyAsync(d_layers + (size_t)l * V, layer_costs[l], sizeof(float) * V, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { ctx->err = "layer upload failed"; rc = -1; }
  }
  if (rc == 0 && mode == 1 && n_layers &&
      hipMemcpyAsync(d_wts, weights, sizeof(float) * n_layers, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { ctx->err = "weight upload failed"; rc = -1; }
  if (rc == 0 && dev_upload(ctx, &ctx->d_cost, (const float*)nullptr, V)) rc = -1;
  if (rc == 0 && edge_distances) { ctx->crn_infl_valid = false; if (dev_upload(ctx, &ctx->d_edge_dist, edge_distances, ctx->E)) rc = -1; }   // NULL: keep the resident ones
  if (rc == 0) {
    const uint32_t gb = (V + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(k_combine, dim3(gb ? gb : 1), dim3(kBlock), 0, ctx->stream, V, mode, n_layers, d_layers, d_wts, ctx->d_cost);
    if (hipGetLastError() != hipSuccess) { ctx->err = "cost combination failed"; rc = -1; }
  }
  // the combined costs never leave the device: the edge-weight pass reads them where they are
  if (rc == 0) rc = edge_weight_pass(ctx, edge_cost_factor, invalid, vertex_costs_out, edge_weights_out);
  (void)hipStreamSynchronize(ctx->stream);
  return rc;
}

int mnav_update_costs(mnav_ctx* ctx, uint32_t n, const uint32_t* vertex_ids, const float* values)
{
  if (!ctx) return -1;
  ctx->err.clear();
  if (!ctx->have_costs) { ctx->err = "no costs resident yet (mnav_upload_costs / mnav_compute_edge_weights / mnav_combine_costs first)"; return -1; }
  if (n == 0) return 0;
  if (!vertex_ids || !values) { ctx->err = "null input array"; return -1; }
  for (uint32_t i = 0; i < n; ++i) if (vertex_ids[i] >= ctx->V) { ctx->err = "vertex id out of range"; return -1; }
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return -1; }
  DevTmp<uint32_t> d_ids; DevTmp<float> d_vals;
  HIPCHK(hipMalloc(d_ids.out(), sizeof(uint32_t) * n));
  HIPCHK(hipMalloc(d_vals.out(), sizeof(float) * n));
  int rc = 0;
  if (hipMemcpyAsync(d_ids, vertex_ids, sizeof(uint32_t) * n, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
      hipMemcpyAsync(d_vals, values, sizeof(float) * n, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { ctx->err = "upload failed"; rc = -1; }
  if (rc == 0) {
    hipLaunchKernelGGL(k_scatter_costs, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, n, d_ids, d_vals, ctx->d_cost);
    // "Edge costs are only affected by vertex costs if layer_factor is not 0" (:568-572)
    if (ctx->edge_cost_factor != 0.0) {
      if (!ctx->d_edge_dist) { ctx->err = "edge distances are not resident (weights were uploaded, not computed here)"; rc = -1; }
      else hipLaunchKernelGGL(k_update_edge_weights, dim3((8 * n + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, n, d_ids, ctx->d_row_ptr,
                              ctx->d_nbr_u, ctx->d_nbr_e, ctx->d_edge_dist, ctx->d_cost, ctx->edge_cost_factor, ctx->d_w);
    }
    if (rc == 0 && hipGetLastError() != hipSuccess) { ctx->err = "cost update failed"; rc = -1; }
  }
  (void)hipStreamSynchronize(ctx->stream);
  if (rc) return rc;
  for (uint32_t i = 0; i < n; ++i) ctx->h_cost[vertex_ids[i]] = values[i];
  ctx->nbr_valid = ctx->crn_valid = false;                          // the cost-limit folded copies are rebuilt on the next plan
  return 0;
}

int mnav_update_edge_weights(mnav_ctx* ctx, uint32_t n, const uint32_t* edge_ids, const float* values)
{
  if (!ctx) return -1;
  ctx->err.clear();
  if (!ctx->have_costs) { ctx->err = "no costs resident yet (mnav_upload_costs / mnav_compute_edge_weights / mnav_combine_costs first)"; return -1; }
  if (n == 0) return 0;
  if (!edge_ids || !values) { ctx->err = "null input array"; return -1; }
  for (uint32_t i = 0; i < n; ++i) if (edge_ids[i] >= ctx->E) { ctx->err = "edge id out of range"; return -1; }
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return -1; }
  DevTmp<uint32_t> d_ids; DevTmp<float> d_vals;
  HIPCHK(hipMalloc(d_ids.out(), sizeof(uint32_t) * n));
  HIPCHK(hipMalloc(d_vals.out(), sizeof(float) * n));
  int rc = 0;
  if (hipMemcpyAsync(d_ids, edge_ids, sizeof(uint32_t) * n, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
      hipMemcpyAsync(d_vals, values, sizeof(float) * n, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { ctx->err = "upload failed"; rc = -1; }
  if (rc == 0) {
    hipLaunchKernelGGL(k_scatter_costs, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, n, d_ids, d_vals, ctx->d_w);   // (a scatter of floats by index)
    if (hipGetLastError() != hipSuccess) { ctx->err = "edge weight update failed"; rc = -1; }
  }
  (void)hipStreamSynchronize(ctx->stream);
  if (rc) return rc;
  ctx->nbr_valid = ctx->crn_valid = false;                          // the cost-limit folded copies (and the tiles' weights behind them) are rebuilt on the next plan
  return 0;
}

int mnav_download_costs(mnav_ctx* ctx, float* vertex_costs_out, float* edge_weights_out)
{
  if (!ctx || !ctx->have_costs) { if (ctx) ctx->err = "no costs resident"; return -1; }
  if (hipSetDevice(ctx->device) != hipSuccess) return -1;
  if (vertex_costs_out) HIPCHK(hipMemcpyAsync(vertex_costs_out, ctx->d_cost, sizeof(float) * ctx->V, hipMemcpyDeviceToHost, ctx->stream));
  if (edge_weights_out) HIPCHK(hipMemcpyAsync(edge_weights_out, ctx->d_w, sizeof(float) * ctx->E, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

// -- layers on the device ------------------------------------------------------------------------
static int layer_slot(mnav_ctx* ctx, uint32_t layer, bool want_dist)
{
  if (!ctx->have_mesh) { ctx->err = "mnav_upload_mesh has not been called"; return -1; }
  if (layer >= 64) { ctx->err = "layer index out of range (64 layers)"; return -1; }
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return -1; }
  if (ctx->layers.size() <= layer) ctx->layers.resize(layer + 1);
  mnav_ctx::Layer& L = ctx->layers[layer];
  const size_t V = ctx->V ? ctx->V : 1;
  if (!L.cost) HIPCHK(hipMalloc((void**)&L.cost, 4 * V));
  if (!L.lethal) HIPCHK(hipMalloc((void**)&L.lethal, V));
  if (want_dist && !L.dist) HIPCHK(hipMalloc((void**)&L.dist, 4 * V));
  return 0;
}

static int ensure_edge_distances(mnav_ctx* ctx)
{
  if (ctx->d_edge_dist) return 0;
  if (dev_upload(ctx, &ctx->d_edge_dist, (const float*)nullptr, ctx->E)) return -1;
  const uint32_t gb = (ctx->E + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(k_edge_dist, dim3(gb ? gb : 1), dim3(kBlock), 0, ctx->stream, ctx->E, ctx->d_edge_vtx, ctx->d_xyz, ctx->d_edge_dist);
  HIPCHK(hipGetLastError());
  ctx->crn_infl_valid = false;
  return 0;
}

int mnav_layer_upload(mnav_ctx* ctx, uint32_t layer, const float* costs, const uint8_t* lethal)
{
  if (!ctx) return -1;
  ctx->err.clear();
  if (layer_slot(ctx, layer, false)) return -1;
  mnav_ctx::Layer& L = ctx->layers[layer];
  if (!costs) { ctx->err = "null cost array"; return -1; }
  HIPCHK(hipMemcpyAsync(L.cost, costs, sizeof(float) * ctx->V, hipMemcpyHostToDevice, ctx->stream));
  if (lethal) HIPCHK(hipMemcpyAsync(L.lethal, lethal, ctx->V, hipMemcpyHostToDevice, ctx->stream));
  else HIPCHK(hipMemsetAsync(L.lethal, 0, ctx->V ? ctx->V : 1, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  L.ready = true;
  return 0;
}

int mnav_layer_steepness(mnav_ctx* ctx, uint32_t layer, double threshold)
{
  if (!ctx) return -1;
  ctx->err.clear();
  if (layer_slot(ctx, layer, false)) return -1;
  if (!ctx->have_normals) { ctx->err = "vertex normals are not resident (mnav_upload_mesh with vertex_normals)"; return -1; }
  mnav_ctx::Layer& L = ctx->layers[layer];
  const uint32_t gb = (ctx->V + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(k_steepness, dim3(gb ? gb : 1), dim3(kBlock), 0, ctx->stream, ctx->V, ctx->d_nrm, threshold, L.cost, L.lethal);
  HIPCHK(hipGetLastError());
  L.ready = true;
  return 0;
}

// InflationLayer::computeLayer (inflation_layer.cpp:96-178 / :577-600): lethals of the input layer -> distances_
// (multi-source wave) -> riskiness.  Runs the wave on the band engine with plan slot 0's work arrays.
int mnav_layer_inflation(mnav_ctx* ctx, uint32_t layer, uint32_t input_layer, double inflation_radius, double inscribed_radius,
                         double inscribed_value, double lethal_value, double cost_scaling_factor, const uint8_t* invalid)
{
  if (!ctx) return -1;
  ctx->err.clear();
  if (input_layer >= ctx->layers.size() || !ctx->layers[input_layer].ready) { ctx->err = "input layer is not resident"; return -1; }
  if (layer == input_layer) { ctx->err = "a layer cannot inflate itself"; return -1; }
  if (layer_slot(ctx, layer, true)) return -1;
  if (ctx->V > (1u << kKeyIdBits)) { ctx->err = "the ordered wave supports meshes of up to 2^26 vertices"; return -1; }
  if (ensure_edge_distances(ctx)) return -1;
  const uint32_t V = ctx->V;
  const uint32_t gb = (V + kBlock - 1) / kBlock ? (V + kBlock - 1) / kBlock : 1;
  if (!ctx->d_crn_infl) HIPCHK(hipMalloc((void**)&ctx->d_crn_infl, sizeof(Corner) * (size_t)(ctx->F ? 3 * (size_t)ctx->F : 1)));
  if (!ctx->crn_infl_valid) {
    hipLaunchKernelGGL(k_build_crn_infl, dim3(gb), dim3(kBlock), 0, ctx->stream, V, ctx->d_crn_ptr, ctx->d_crn_idx, ctx->d_edge_dist, ctx->d_crn_infl);
    HIPCHK(hipGetLastError());
    ctx->crn_infl_valid = true;
  }
  const size_t Vn = V ? V : 1;
  if (!ctx->d_infl_mask) HIPCHK(hipMalloc((void**)&ctx->d_infl_mask, Vn));
  if (!ctx->d_zero_u8) { HIPCHK(hipMalloc((void**)&ctx->d_zero_u8, Vn)); HIPCHK(hipMemsetAsync(ctx->d_zero_u8, 0, Vn, ctx->stream)); }
  if (!ctx->d_infl_keyd) HIPCHK(hipMalloc((void**)&ctx->d_infl_keyd, 4 * Vn));
  DevTmp<uint8_t> d_inv;
  if (invalid) { HIPCHK(hipMalloc(d_inv.out(), Vn)); HIPCHK(hipMemcpyAsync(d_inv, invalid, V, hipMemcpyHostToDevice, ctx->stream)); }
  mnav_ctx::Layer& L = ctx->layers[layer];
  mnav_ctx::Layer& In = ctx->layers[input_layer];
  L.inflation_radius = inflation_radius; L.inscribed_radius = inscribed_radius; L.inscribed_value = inscribed_value; L.lethal_value = lethal_value;
  hipLaunchKernelGGL(k_infl_mask, dim3(gb), dim3(kBlock), 0, ctx->stream, V, In.lethal, d_inv, ctx->d_infl_mask);
  HIPCHK(hipMemcpyAsync(L.lethal, In.lethal, V, hipMemcpyDeviceToDevice, ctx->stream));    // lethal_vertices_ = input->lethals() :170,:584
  if (ensure_slots(ctx, 1, true, true, false)) return -1;
  Slot& s = ctx->slots[0];
  ctx->caller_slot.assign(1, kNone);                                // the wave works in plan slot 0: the last plan's resident outputs are gone
  ctx->shard.finalized = false;
  Plan P;
  memset(&P, 0, sizeof(P));
  P.planner = kPlannerCvp; P.V = V;
  P.row_ptr = ctx->d_row_ptr; P.nbr = nullptr; P.crn_ptr = ctx->d_crn_ptr; P.crn = ctx->d_crn_infl; P.blocked = ctx->d_zero_u8;
  P.dist = L.dist; P.tkey = s.tkey; P.pred = s.pred; P.dirn = s.dirn; P.cutf = s.cutf; P.stamp = s.stamp; P.dirty = s.dirty;
  P.list[0] = s.list0; P.list[1] = s.list1; P.wlist[0] = s.wlist0; P.wlist[1] = s.wlist1; P.wstamp = s.wstamp; P.cap = V; P.ctl = s.ctl; P.cnt = s.cnt;
  const float maxd = (float)inflation_radius;                                               // :438 (const float&)
  P.delta = maxd > 0.f ? maxd : 1.0f;                                                       // one band per radius: the wave dies out within ~2
  P.offset = 0.0; P.max_steps = ctx->max_steps; P.walk_max = ctx->walk_max; P.descend_max = ctx->descend_max;
  for (int k = 0; k < 3; ++k) { P.seed[k] = kNone; P.target[k] = kNone; P.seed_d[k] = 0.f; P.seed_expands[k] = 1; P.target_expands[k] = 0; }
  P.seed_face = kNone;
  P.seed_mask = ctx->d_infl_mask; P.keyd = ctx->d_infl_keyd; P.infl_max = maxd;
  HIPCHK(hipMemcpyAsync(ctx->d_plans, &P, sizeof(Plan), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipEventRecord(ctx->ev[1], ctx->stream));
  uint32_t gi = (V + kBlock * 4 - 1) / (kBlock * 4);
  if (gi < 1) gi = 1;
  if (gi > 4096) gi = 4096;
  hipLaunchKernelGGL(k_init<kPlannerCvp>, dim3(gi, 1), dim3(kBlock), 0, ctx->stream, ctx->d_plans);
  hipLaunchKernelGGL(k_infl_ctl, dim3(1), dim3(64), 0, ctx->stream, ctx->d_plans);
  hipLaunchKernelGGL(k_infl_seed, dim3(gi), dim3(kBlock), 0, ctx->stream, ctx->d_plans);
  HIPCHK(hipGetLastError());
  // the wave front of an inflation is as long as the lethal contours, not O(sqrt V): more waves than a plan gets
  uint32_t G = blocks_per_plan(ctx) * 12u;
  if (G > 8192u) G = 8192u;
  const auto t_start = std::chrono::steady_clock::now();
  Ctl last{};
  ctx->infl_exact_bands = 0;
  for (;;) {
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() > ctx->max_wall_s) {
      ctx->err = "inflation wave exceeded the wall-clock guard"; return -1;
    }
    const auto t_c0 = std::chrono::steady_clock::now();
    const bool tr = opt_on(ctx->opt.trace);
    unsigned long long* const stamps = reinterpret_cast<unsigned long long*>(ctx->h_res);   // (trace: two device clock stamps in the pinned result record, unused during a wave)
    if (tr) { HIPCHK(hipEventRecord(ctx->ev[4], ctx->stream)); hipLaunchKernelGGL(k_stamp_to_host, dim3(1), dim3(1), 0, ctx->stream, stamps); }   // (trace: the chunk's time on the DEVICE, apart from the host's wait for it)
    if (run_chunk<kPlannerCvp>(ctx, 1, G, false)) return -1;
    if (tr) HIPCHK(hipEventRecord(ctx->ev[5], ctx->stream));
    const auto t_c1 = std::chrono::steady_clock::now();
    if (tr) hipLaunchKernelGGL(k_stamp_to_host, dim3(1), dim3(1), 0, ctx->stream, stamps + 1);
    HIPCHK(hipMemcpyAsync(ctx->h_ctl, ctx->d_ctl_pool, 2 * sizeof(Ctl), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const auto t_c2 = std::chrono::steady_clock::now();
    float dev_ms = 0.f;
    if (tr) (void)hipEventElapsedTime(&dev_ms, ctx->ev[4], ctx->ev[5]);
    last = ctx->h_ctl[0].it > ctx->h_ctl[1].it ? ctx->h_ctl[0] : ctx->h_ctl[1];
    if (opt_on(ctx->opt.trace))
      fprintf(stderr, "[mnav] inflation it %d n %u thr %.6f fixed %.6f width %.4g bands %u band_steps %u shrinks %u cuts %u repair %u evals %u wread %u wbase %u done %u (verify sweeps of the previous wave %u) host ms: since start %.3f, this chunk's launch %.3f, its wait %.3f, the chunk on the device %.3f, by the device's own clock %.3f\n", last.it,
              last.n, last.thr, last.thr_fixed, last.width, last.bands, last.band_steps, last.shrinks, last.cuts, last.repair, last.evals, last.wread, last.wbase, last.done, ctx->verify_sweeps_used,
              1e3 * std::chrono::duration<double>(t_c0 - t_start).count(), 1e3 * std::chrono::duration<double>(t_c1 - t_c0).count(), 1e3 * std::chrono::duration<double>(t_c2 - t_c1).count(), dev_ms, tr ? (double)(stamps[1] - stamps[0]) * 1e-5 : 0.0);
    if (last.done) break;
    if (last.exact_wanted) {
      // a band that neither the concurrent steps nor a serial band from a clean state settle (tied pop times around isolated
      // lethal vertices: 0.5 % of random such maps): its vertices are popped one at a time, the reference's own procedure
      // (mnav_eval.h exact_*; the steps idle meanwhile and resume with a cut step)
      hipLaunchKernelGGL(k_exact_band, dim3(1), dim3(kWave), 0, ctx->stream, ctx->d_plans, (uint32_t*)nullptr);
      HIPCHK(hipGetLastError());
      ++ctx->infl_exact_bands;
    }
  }
  HIPCHK(hipEventRecord(ctx->ev[2], ctx->stream));
  // verification: every vertex must be a fixed point of the replay rule on the converged state (k_cvp_verify)
  if (verify_sweeps(ctx, 1)) return -1;
  hipLaunchKernelGGL(k_infl_cost, dim3(gb), dim3(kBlock), 0, ctx->stream, V, L.dist, inflation_radius, inscribed_radius, inscribed_value,
                     lethal_value, cost_scaling_factor, L.cost);
  HIPCHK(hipGetLastError());
  // vector_map_ (:277-309): accumulation over the lethal contours, then assignments in pop order (launches until settled)
  L.have_vec = false;
  if (!L.vec) HIPCHK(hipMalloc((void**)&L.vec, 12 * Vn));
  if (!L.vstate) HIPCHK(hipMalloc((void**)&L.vstate, 3 * Vn));
  if (!ctx->d_verify_any) HIPCHK(hipMalloc((void**)&ctx->d_verify_any, 4));
  uint32_t* d_vctl = nullptr;
  HIPCHK(hipMalloc((void**)&d_vctl, 16));
  HIPCHK(hipMemsetAsync(d_vctl, 0, 16, ctx->stream));
  uint8_t *st0 = L.vstate, *st1 = L.vstate + Vn, *acc = L.vstate + 2 * Vn;
  hipLaunchKernelGGL(k_infl_accum, dim3(gb), dim3(kBlock), 0, ctx->stream, ctx->d_plans, ctx->d_crn_walk, ctx->d_xyz, L.vec, st0, acc, d_vctl);
  uint32_t vctl[4] = { 0, 0, 0, 0 };
  bool vec_ok = true;
  for (int sweep = 0; sweep < 4096; ++sweep) {
    HIPCHK(hipMemsetAsync(d_vctl, 0, 8, ctx->stream));
    hipLaunchKernelGGL(k_infl_assign, dim3(gb), dim3(kBlock), 0, ctx->stream, ctx->d_plans, L.vec, st0, st1, acc, d_vctl);
    HIPCHK(hipMemcpyAsync(vctl, d_vctl, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    std::swap(st0, st1);
    if (vctl[2]) { vec_ok = false; break; }                          // a vertex with too many neighbours for the walk positions
    if (vctl[0] == 0) break;
    if (vctl[1] == 0) { vec_ok = false; break; }                     // nothing moved although something waits: not on a verified state
  }
  if (st0 != L.vstate) HIPCHK(hipMemcpyAsync(L.vstate, st0, Vn, hipMemcpyDeviceToDevice, ctx->stream));   // final states in the first array
  (void)hipFree(d_vctl);
  L.have_vec = vec_ok;
  HIPCHK(hipEventRecord(ctx->ev[3], ctx->stream));
  Cnt flags{};
  HIPCHK(hipMemcpyAsync(&flags, s.cnt + 3, sizeof(Cnt), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->infl_steps = (uint32_t)(last.it < 0 ? 0 : last.it); ctx->infl_bands = last.bands; ctx->infl_evals = last.evals;
  ctx->infl_ms = ev_ms(ctx->ev[1], ctx->ev[3]); ctx->infl_ms_wave = ev_ms(ctx->ev[1], ctx->ev[2]);
  if (last.overflow) { ctx->err = "inflation wave did not converge (work-list overflow or step limit)"; return -1; }
  if (flags.n_next & kFlagWalkLimit) { ctx->err = "inflation wave: cascade-tree walk bound hit; the result may not be the reference's"; return -1; }
  if (flags.changed) { ctx->err = "inflation wave: the converged state is not a fixed point of the replay rule"; return -1; }
  L.ready = true;
  return 0;
}

int mnav_layer_download(mnav_ctx* ctx, uint32_t layer, float* costs_out, uint8_t* lethal_out, float* distances_out)
{
  if (!ctx) return -1;
  ctx->err.clear();
  if (layer >= ctx->layers.size() || !ctx->layers[layer].ready) { ctx->err = "layer is not resident"; return -1; }
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return -1; }
  mnav_ctx::Layer& L = ctx->layers[layer];
  if (costs_out) HIPCHK(hipMemcpyAsync(costs_out, L.cost, sizeof(float) * ctx->V, hipMemcpyDeviceToHost, ctx->stream));
  if (lethal_out) HIPCHK(hipMemcpyAsync(lethal_out, L.lethal, ctx->V, hipMemcpyDeviceToHost, ctx->stream));
  if (distances_out) {
    if (!L.dist) { ctx->err = "this layer keeps no distances"; return -1; }
    HIPCHK(hipMemcpyAsync(distances_out, L.dist, sizeof(float) * ctx->V, hipMemcpyDeviceToHost, ctx->stream));
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

int mnav_layer_download_vectors(mnav_ctx* ctx, uint32_t layer, float* vectors_out, uint8_t* has_vector_out)
{
  if (!ctx) return -1;
  ctx->err.clear();
  if (layer >= ctx->layers.size() || !ctx->layers[layer].ready) { ctx->err = "layer is not resident"; return -1; }
  mnav_ctx::Layer& L = ctx->layers[layer];
  if (!L.have_vec) { ctx->err = "this layer has no vector field (not an inflation layer, or a vertex with more than 15 neighbours)"; return -1; }
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return -1; }
  const uint32_t V = ctx->V;
  if (vectors_out) HIPCHK(hipMemcpyAsync(vectors_out, L.vec, 12 * (size_t)V, hipMemcpyDeviceToHost, ctx->stream));
  std::vector<uint8_t> st(V ? V : 1);
  if (has_vector_out) HIPCHK(hipMemcpyAsync(st.data(), L.vstate, V, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (has_vector_out) for (uint32_t v = 0; v < V; ++v) has_vector_out[v] = st[v] == 1 ? 1 : 0;
  return 0;
}

int mnav_layer_stats(const mnav_ctx* ctx, uint32_t* steps, uint32_t* bands, uint64_t* evals, float* ms, uint32_t* verify_sweeps, float* ms_wave)
{
  if (!ctx) return -1;
  if (steps) *steps = ctx->infl_steps;
  if (bands) *bands = ctx->infl_bands;
  if (evals) *evals = ctx->infl_evals;
  if (ms) *ms = ctx->infl_ms;
  if (verify_sweeps) *verify_sweeps = ctx->verify_sweeps_used;
  if (ms_wave) *ms_wave = ctx->infl_ms_wave;
  return 0;
}

// CombinationLayer (max :44-85 / weighted sum :185-248) over resident layers, then MeshMap::computeEdgeWeights:
// the whole cost preparation of a map without a host copy of a single V-sized array.
int mnav_combine_layers(mnav_ctx* ctx, int mode, uint32_t n_layers, const uint32_t* layers, const float* weights, double edge_cost_factor,
                        const uint8_t* invalid)
{
  if (!ctx) return -1;
  ctx->err.clear();
  if (!ctx->have_mesh) { ctx->err = "mnav_upload_mesh has not been called"; return -1; }
  if (mode != 0 && mode != 1) { ctx->err = "combination mode must be 0 (max) or 1 (weighted sum)"; return -1; }
  if ((n_layers && !layers) || (mode == 1 && n_layers && !weights) || n_layers > 64) { ctx->err = "bad layer list"; return -1; }
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return -1; }
  std::vector<const float*> ptrs(n_layers ? n_layers : 1, nullptr);
  for (uint32_t l = 0; l < n_layers; ++l) {
    if (layers[l] >= ctx->layers.size() || !ctx->layers[layers[l]].ready) { ctx->err = "layer is not resident"; return -1; }
    ptrs[l] = ctx->layers[layers[l]].cost;
  }
  if (ensure_edge_distances(ctx)) return -1;
  DevTmp<const float*> d_ptrs; DevTmp<float> d_wts;
  HIPCHK(hipMalloc(d_ptrs.out(), sizeof(float*) * (n_layers + 1)));
  HIPCHK(hipMalloc(d_wts.out(), sizeof(float) * (n_layers + 1)));
  int rc = 0;
  if (n_layers && hipMemcpyAsync(d_ptrs, ptrs.data(), sizeof(float*) * n_layers, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = -1;
  if (rc == 0 && mode == 1 && n_layers && hipMemcpyAsync(d_wts, weights, sizeof(float) * n_layers, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = -1;
  if (rc == 0 && dev_upload(ctx, &ctx->d_cost, (const float*)nullptr, ctx->V)) rc = -1;
  if (rc == 0) {
    const uint32_t gb = (ctx->V + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(k_combine_resident, dim3(gb ? gb : 1), dim3(kBlock), 0, ctx->stream, ctx->V, mode, n_layers, d_ptrs, d_wts, ctx->d_cost);
    if (hipGetLastError() != hipSuccess) rc = -1;
  }
  if (rc != 0 && ctx->err.empty()) ctx->err = "layer combination failed";
  if (rc == 0) rc = edge_weight_pass(ctx, edge_cost_factor, invalid, nullptr, nullptr);
  (void)hipStreamSynchronize(ctx->stream);
  return rc;
}

// The incremental counterpart of mnav_combine_layers: CombinationLayer::onInputChanged + MeshMap::layerChanged +
// updateEdgeWeights(changed) for the n vertices a layer reported as changed (the layers themselves were updated on the
// device or re-uploaded before).  Same combination mode / layer list / weights as the full pass.
int mnav_combine_layers_update(mnav_ctx* ctx, int mode, uint32_t n_layers, const uint32_t* layers, const float* weights, uint32_t n,
                               const uint32_t* vertex_ids)
{
  if (!ctx) return -1;
  ctx->err.clear();
  if (!ctx->have_costs) { ctx->err = "no combined costs resident yet (mnav_combine_layers first)"; return -1; }
  if (mode != 0 && mode != 1) { ctx->err = "combination mode must be 0 (max) or 1 (weighted sum)"; return -1; }
  if ((n_layers && !layers) || (mode == 1 && n_layers && !weights) || n_layers > 64) { ctx->err = "bad layer list"; return -1; }
  if (n == 0) return 0;
  if (!vertex_ids) { ctx->err = "null id array"; return -1; }
  for (uint32_t i = 0; i < n; ++i) if (vertex_ids[i] >= ctx->V) { ctx->err = "vertex id out of range"; return -1; }
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return -1; }
  std::vector<const float*> ptrs(n_layers ? n_layers : 1, nullptr);
  for (uint32_t l = 0; l < n_layers; ++l) {
    if (layers[l] >= ctx->layers.size() || !ctx->layers[layers[l]].ready) { ctx->err = "layer is not resident"; return -1; }
    ptrs[l] = ctx->layers[layers[l]].cost;
  }
  DevTmp<const float*> d_ptrs; DevTmp<float> d_wts, d_vals; DevTmp<uint32_t> d_ids;
  HIPCHK(hipMalloc(d_ptrs.out(), sizeof(float*) * (n_layers + 1)));
  HIPCHK(hipMalloc(d_wts.out(), sizeof(float) * (n_layers + 1)));
  HIPCHK(hipMalloc(d_vals.out(), sizeof(float) * n));
  HIPCHK(hipMalloc(d_ids.out(), sizeof(uint32_t) * n));
  std::vector<float> vals(n);
  int rc = 0;
  if (n_layers && hipMemcpyAsync(d_ptrs, ptrs.data(), sizeof(float*) * n_layers, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = -1;
  if (rc == 0 && mode == 1 && n_layers && hipMemcpyAsync(d_wts, weights, sizeof(float) * n_layers, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = -1;
  if (rc == 0 && hipMemcpyAsync(d_ids, vertex_ids, sizeof(uint32_t) * n, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = -1;
  if (rc == 0) {
    hipLaunchKernelGGL(k_combine_resident_ids, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, n, d_ids, mode, n_layers, d_ptrs, d_wts,
                       ctx->d_cost, d_vals);
    if (ctx->edge_cost_factor != 0.0)                               // "Edge costs are only affected by vertex costs if layer_factor is not 0" (:568-572)
      hipLaunchKernelGGL(k_update_edge_weights, dim3((8 * n + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, n, d_ids, ctx->d_row_ptr,
                         ctx->d_nbr_u, ctx->d_nbr_e, ctx->d_edge_dist, ctx->d_cost, ctx->edge_cost_factor, ctx->d_w);
    if (hipGetLastError() != hipSuccess) rc = -1;
  }
  if (rc == 0 && hipMemcpyAsync(vals.data(), d_vals, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = -1;
  (void)hipStreamSynchronize(ctx->stream);
  if (rc != 0) { if (ctx->err.empty()) ctx->err = "incremental combination failed"; return rc; }
  for (uint32_t i = 0; i < n; ++i) ctx->h_cost[vertex_ids[i]] = vals[i];
  ctx->nbr_valid = ctx->crn_valid = false;                          // the cost-limit folded copies are rebuilt on the next plan
  return 0;
}

static uint32_t dijkstra_impl(mnav_ctx* ctx, uint32_t n, const uint32_t* seeds, const uint32_t* targets, double offset,
                              double cost_limit, uint32_t* codes_out, float* dist_out, uint32_t* pred_out, uint32_t* path_out,
                              uint32_t path_cap, uint32_t* path_len, float* vecmap_out, bool want_vecmap)
{
  if (check_ready(ctx)) return MNAV_INTERNAL_ERROR;
  ctx->err.clear();
  // goal_dist = dist[target] + offset cuts the wave off BEHIND the robot (dijkstra :296).  A negative offset (the reference
  // takes any double) stops the expansion AT the robot vertex: the engines run with the bound of offset 0 -- everything at or
  // below dist[target] is final then -- and the passes that derive tentative values, predecessors and paths apply the
  // reference's expanded set through goal_cut() (mnav_eval.h).
  if (offset != offset) { ctx->err = "goal_dist_offset is NaN"; return MNAV_INTERNAL_ERROR; }
  ctx->cancel.store(0);                                               // dijkstra :238
  if (ctx->d_cancel) { (void)hipStreamSynchronize(ctx->cancel_stream); (void)hipMemsetAsync(ctx->d_cancel, 0, 4, ctx->stream); }
  want_vecmap = want_vecmap || ctx->resident_vecmap;
  ctx->want_vec = want_vecmap;
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return MNAV_INTERNAL_ERROR; }
  MTRACE("start");
  const uint32_t V = ctx->V;
  uint32_t worst = MNAV_SUCCESS;
  // id checks stand in for the optional-handle tests of dijkstra :240-243
  std::vector<PlanIn> in; std::vector<uint32_t> map;   // map: device plan -> caller index
  std::vector<uint32_t> codes(n, MNAV_SUCCESS);
  std::vector<uint8_t> cleared(n, 0);
  for (uint32_t i = 0; i < n; ++i) {
    if (path_len) path_len[i] = 0;
    if (seeds[i] >= V) { codes[i] = MNAV_INVALID_START; continue; }
    if (targets[i] >= V) { codes[i] = MNAV_INVALID_GOAL; continue; }
    if (seeds[i] == targets[i]) { cleared[i] = 1; continue; }   // :252-255 SUCCESS right after clearing the maps
    PlanIn p{};
    for (int k = 0; k < 3; ++k) { p.seed[k] = kNone; p.target[k] = kNone; p.seed_d[k] = 0.f; p.seed_expands[k] = 1; p.target_expands[k] = 1; }
    p.seed[0] = seeds[i]; p.target[0] = targets[i]; p.seed_face = kNone;
    in.push_back(p); map.push_back(i);
  }
  // Longest plans first: a plan's work grows with the area its wave sweeps before it reaches the robot
  // vertex, i.e. with the squared seed-target distance.  Workgroups are dispatched in plan order, so when a
  // batch holds more plans than the device runs at once the short ones back-fill behind the long ones
  // instead of leaving a tail (results are mapped back through `map`).
  // engine: 0 = tiled rounds, 1 = band steps, 3 = auto, 5 = tile-batch (plan-vectorised, large batches), 6 = asynchronous tiles
  int engine = ctx->dij_engine;
  const bool engine_auto = engine == 3;
  // paths only (nothing V-sized asked for, nothing kept resident): no finalize pass, predecessors along the path only
  ctx->lazy_paths = ctx->allow_lazy_paths && !dist_out && !pred_out && !want_vecmap && !ctx->resident_vecmap;
  {
    const uint32_t m0 = (uint32_t)in.size();
    // auto: one wave per (tile, 64 plans) for large batches, one workgroup per plan for medium ones, tile rounds otherwise
    if (engine == 3) {
      // The tile-batch engine advances all plans tile by tile, 16 plans per quarter of a wave (k_tb_solve_q).  Its floor is one
      // stream pass per iteration (~190 iterations x ~180 us on the 1M mesh, ~600 x 300 us at 10M), whatever the batch; above
      // that it beats the per-plan engines at every size measured (round 4, plans/s, tiled / persistent / tile-batch --
      // 1M: 64 plans 1084 / 496 / 1359, 256: 1879 / 1929 / 4392, 1024: 2223 / 7107 / 12871;
      // 10M: 64 plans 196 / 41 / 189, 256: 252 / 163 / 409, 1024: 254 / 594 / 924).
      const double tiles = std::max(1.0, (double)V / (0.9 * ctx->tb.T));
      const bool fills = m0 >= ctx->tb.min_batch && m0 <= 65535u && (double)m0 >= tiles / 1000.0;
      // Below that: the asynchronous tile engine (mnav_async.h: one launch, per-plan ticket queues of woken tiles; what a real
      // makePlan call -- ONE plan -- runs on), up to async_max_batch plans.  Measured round 5, ms per call on the 1M mesh, rounds /
      // tile-batch / asynchronous: 1 plan 10.1 / 20 / 6.8, 8 plans 28 / 29 / 9.7, 47 plans 55 / 38 / 22.9, 64 plans 65 / 46 / 27.5
      // (10M mesh: 8 plans 83 / - / 45, 47 plans 243 / - / 182, 64 plans 330 / 340 / 224); it grows linearly with the batch
      // where the tile-batch engine's iterations are shared by all plans: the engines cross at about a hundred plans.
      // Round 6 (a ticket's round trips cut, mnav_async.h): ms per call, asynchronous / tile-batch, 1M mesh 96 plans 29 / 51,
      // 128: 36 / 53, 192: 52 / 58, 256: 69 / 60; 10M mesh 96: 261 / 343, 128: 325 / 388, 192: 540 / 470
      // (profiles/r06_async_crossover.json): the engines now cross at about 160 plans on both meshes.
      engine = (m0 <= opt_u32(ctx->opt.async_max_batch, 160u)) ? 6 : fills ? 5 : 0;
    }
    if (engine == 5 && m0 > 65535u) engine = 0;                       // (plan ids are 16 bits in the tile-batch buckets)
    if (engine == 1 && offset < 0.0) engine = 0;                      // the band steps arm goal_dist inside the loop: tile rounds for a negative offset
  }
  if (engine == 5 && in.size() > 1) {
    // plans whose waves start close to each other are neighbours in the batch: their slices of a tile are adjacent in memory
    // and they tend to have work on the same tiles at the same time
    if (tb_build(ctx)) return MNAV_INTERNAL_ERROR;
    std::vector<uint32_t> ord(in.size());
    std::iota(ord.begin(), ord.end(), 0u);
    const std::vector<uint32_t>& vt = ctx->tb.vert_tile;
    std::stable_sort(ord.begin(), ord.end(), [&](uint32_t x, uint32_t y) { return vt[in[x].seed[0]] < vt[in[y].seed[0]]; });
    std::vector<PlanIn> in2(in.size()); std::vector<uint32_t> map2(in.size());
    for (size_t i = 0; i < in.size(); ++i) { in2[i] = in[ord[i]]; map2[i] = map[ord[i]]; }
    in.swap(in2); map.swap(map2);
  } else
  if (in.size() > 1 && ctx->h_xyz.size() == 3 * (size_t)V) {
    std::vector<uint32_t> ord(in.size());
    std::iota(ord.begin(), ord.end(), 0u);
    std::vector<float> est(in.size());
    for (size_t i = 0; i < in.size(); ++i) {
      const float* a = &ctx->h_xyz[3 * (size_t)in[i].seed[0]];
      const float* b = &ctx->h_xyz[3 * (size_t)in[i].target[0]];
      est[i] = (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]);
    }
    std::stable_sort(ord.begin(), ord.end(), [&](uint32_t x, uint32_t y) { return est[x] > est[y]; });
    std::vector<PlanIn> in2(in.size()); std::vector<uint32_t> map2(in.size());
    for (size_t i = 0; i < in.size(); ++i) { in2[i] = in[ord[i]]; map2[i] = map[ord[i]]; }
    in.swap(in2); map.swap(map2);
  }
  ctx->caller_slot.assign(n, kNone);
  ctx->shard.finalized = false;
  for (size_t i = 0; i < map.size(); ++i) ctx->caller_slot[map[i]] = (uint32_t)i;
  (void)hipEventRecord(ctx->ev[0], ctx->stream);
  const uint32_t m = (uint32_t)in.size();
  ctx->tb.count_pending = false; ctx->tb_args_valid = false;
  ctx->last_planner = kPlannerDijkstra; ctx->last_n = 0;            // outputs of the previous call are gone; this call's count once its engine succeeded
  ctx->last_target.resize(m); for (uint32_t k = 0; k < m; ++k) ctx->last_target[k] = in[k].target[0];
  ctx->last_offset = offset;
  if (m) {
    if (materialize(ctx, false, cost_limit)) return MNAV_INTERNAL_ERROR;
    const bool want_path = true;
    int rc = (engine == 0) ? run_dijkstra_tiled(ctx, m, in, offset)
           : (engine == 6) ? run_dijkstra_async(ctx, m, in, offset)
           : (engine == 5) ? run_dijkstra_tb(ctx, m, in, offset)
                           : run_plans<kPlannerDijkstra>(ctx, m, in, offset, want_path);
    // ticket ring exhausted: the rounds start over.  So they do when the in-kernel watchdog gave up (rc 3) on an engine that `auto`
    // picked: the rounds have the whole max_wall_s (a serialising profiler, a mesh far beyond the tuned sizes)
    if (engine == 6 && (rc == 2 || (rc == 3 && engine_auto))) { ctx->err.clear(); engine = 0; rc = run_dijkstra_tiled(ctx, m, in, offset); }
    if (rc == 3) rc = -1;
    MTRACE("engine returned");
    if (rc != 0) (void)hipStreamSynchronize(ctx->stream);             // nothing of a failed / cancelled call stays in flight
    if (rc < 0) return MNAV_INTERNAL_ERROR;
    if (rc == 1) { for (uint32_t i = 0; i < n; ++i) if (codes_out) codes_out[i] = MNAV_CANCELED; return MNAV_CANCELED; }   // :350-354
    ctx->last_engine = engine; ctx->last_n = m;                       // only a call whose engine succeeded leaves outputs behind
    if (engine == 1) ctx->lazy_paths = false;                         // the band steps keep their predecessors as they go
    const PathRows rows1{ ctx->d_paths, ctx->path_stride, nullptr, nullptr };
    PathRows rows2{ nullptr, 0u, nullptr, nullptr };
    const uint32_t gc = (V + kBlock * 4 - 1) / (kBlock * 4);
    if (engine == 5 && ctx->lazy_paths) {
      hipLaunchKernelGGL(k_tb_path, dim3(m), dim3(kWave), 0, ctx->stream, ctx->tb_args, ctx->d_row_ptr, ctx->d_nbr, V, ctx->d_res, rows1, ctx->d_mismatch);
      ctx->tb.count_pending = true;                                  // settled vertices (a statistic): counted when somebody asks, mnav_get_stats
    } else if (ctx->lazy_paths) {
      hipLaunchKernelGGL(k_path_lazy, dim3(m), dim3(kWave), 0, ctx->stream, ctx->d_plans, ctx->d_tplans, ctx->d_res, rows1, ctx->d_mismatch);
      hipLaunchKernelGGL(k_count_goal, dim3(gc ? gc : 1, m), dim3(kBlock), 0, ctx->stream, ctx->d_plans, ctx->d_res);
    } else
    hipLaunchKernelGGL(k_finish<kPlannerDijkstra>, dim3(m), dim3(64), 0, ctx->stream, ctx->d_plans, ctx->d_res, rows1);
    if (engine == 1)   // the tile engines count the settled vertices in k_dij_finalize
      hipLaunchKernelGGL(k_count, dim3(gc ? gc : 1, m), dim3(kBlock), 0, ctx->stream, ctx->d_plans, ctx->d_res);
    (void)hipEventRecord(ctx->ev[4], ctx->stream);
    if (want_vecmap && !(engine == 5 && !ctx->lazy_paths))            // (the tile-batch engine's finalize pass writes the vector map itself)
      hipLaunchKernelGGL(k_vecmap_dijkstra, dim3(gc ? gc : 1, m), dim3(kBlock), 0, ctx->stream, ctx->d_plans, ctx->d_xyz, ctx->d_vecptrs);
    (void)hipEventRecord(ctx->ev[5], ctx->stream);
    if (hipMemcpyAsync(ctx->h_res, ctx->d_res, sizeof(PlanResult) * m, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) { ctx->err = "result download failed"; return MNAV_INTERNAL_ERROR; }
    {
      // paths longer than the default rows (corridors, mazes): ONLY those plans are walked again, into rows of exactly their
      // length (the first walk counted it) in one packed buffer
      size_t over_words = 0;
      std::vector<unsigned long long> ooff(m, 0ull); std::vector<uint32_t> ocap(m, 0u);
      for (uint32_t k = 0; k < m; ++k)
        if (ctx->h_res[k].code == kPathOverflow) { ooff[k] = over_words; ocap[k] = ctx->h_res[k].path_len; over_words += ctx->h_res[k].path_len; }
      if (over_words) {
        std::vector<PlanResult> keep(ctx->h_res, ctx->h_res + m);   // settled / evals were accumulated by other kernels
        (void)hipFree(ctx->d_over); (void)hipFree(ctx->d_over_off); (void)hipFree(ctx->d_over_cap);
        ctx->d_over = nullptr; ctx->d_over_off = nullptr; ctx->d_over_cap = nullptr;
        if (hipMalloc((void**)&ctx->d_over, 4 * over_words) != hipSuccess || hipMalloc((void**)&ctx->d_over_off, 8 * (size_t)m) != hipSuccess ||
            hipMalloc((void**)&ctx->d_over_cap, 4 * (size_t)m) != hipSuccess ||
            hipMemcpyAsync(ctx->d_over_off, ooff.data(), 8 * (size_t)m, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
            hipMemcpyAsync(ctx->d_over_cap, ocap.data(), 4 * (size_t)m, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
          { ctx->err = "path buffers: out of memory"; return MNAV_INTERNAL_ERROR; }
        rows2 = PathRows{ ctx->d_over, 0u, ctx->d_over_off, ctx->d_over_cap };
        if (engine == 5 && ctx->lazy_paths) hipLaunchKernelGGL(k_tb_path, dim3(m), dim3(kWave), 0, ctx->stream, ctx->tb_args, ctx->d_row_ptr, ctx->d_nbr, V, ctx->d_res, rows2, ctx->d_mismatch);
        else if (ctx->lazy_paths) hipLaunchKernelGGL(k_path_lazy, dim3(m), dim3(kWave), 0, ctx->stream, ctx->d_plans, ctx->d_tplans, ctx->d_res, rows2, ctx->d_mismatch);
        else hipLaunchKernelGGL(k_finish<kPlannerDijkstra>, dim3(m), dim3(64), 0, ctx->stream, ctx->d_plans, ctx->d_res, rows2);
        if (hipMemcpyAsync(ctx->h_res, ctx->d_res, sizeof(PlanResult) * m, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) { ctx->err = "result download failed"; return MNAV_INTERNAL_ERROR; }
        for (uint32_t k = 0; k < m; ++k) ctx->h_res[k].settled = keep[k].settled;
      }
    }
    if (ctx->last_engine != 1) {
      uint32_t mism = 0;
      if (hipMemcpy(&mism, ctx->d_mismatch, 4, hipMemcpyDeviceToHost) != hipSuccess || mism != 0) {
        ctx->err = "tiled SSSP did not reach its fixed point (" + std::to_string(mism) + " vertices)";
        return MNAV_INTERNAL_ERROR;
      }
    }
    MTRACE("results downloaded");
    // all vertex paths: packed and reversed on the device (k_pack_paths), one dense copy into a pinned buffer
    std::vector<uint32_t> offs(m + 1, 0), lens(m, 0);
    for (uint32_t k = 0; k < m; ++k) {
      lens[k] = (ctx->h_res[k].code == MNAV_SUCCESS) ? ctx->h_res[k].path_len : 0u;
      offs[k + 1] = offs[k] + lens[k];
    }
    const size_t total = offs[m];
    if (total && path_out && path_cap) {
      if (ctx->pack_words < total) {
        if (ctx->d_pack) (void)hipFree(ctx->d_pack);
        if (ctx->h_pack) (void)hipHostFree(ctx->h_pack);
        ctx->d_pack = nullptr; ctx->h_pack = nullptr; ctx->pack_words = 0;
        const size_t want = total + total / 4 + 1024;
        if (hipMalloc((void**)&ctx->d_pack, 4 * want) != hipSuccess || hipHostMalloc((void**)&ctx->h_pack, 4 * want, hipHostMallocDefault) != hipSuccess)
          { ctx->err = "path buffers: out of memory"; return MNAV_INTERNAL_ERROR; }
        ctx->pack_words = want;
      }
      if (ctx->pack_meta_n < 2 * (size_t)m) {
        if (ctx->d_pack_meta) (void)hipFree(ctx->d_pack_meta);
        ctx->d_pack_meta = nullptr; ctx->pack_meta_n = 0;
        if (hipMalloc((void**)&ctx->d_pack_meta, 4 * 2 * (size_t)m) != hipSuccess) { ctx->err = "path buffers: out of memory"; return MNAV_INTERNAL_ERROR; }
        ctx->pack_meta_n = 2 * (size_t)m;
      }
      if (hipMemcpyAsync(ctx->d_pack_meta, offs.data(), 4 * (size_t)m, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
          hipMemcpyAsync(ctx->d_pack_meta + m, lens.data(), 4 * (size_t)m, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        { ctx->err = "path download failed"; return MNAV_INTERNAL_ERROR; }
      hipLaunchKernelGGL(k_pack_paths, dim3(m), dim3(kBlock), 0, ctx->stream, rows1, rows2, ctx->d_pack_meta, ctx->d_pack_meta + m, ctx->d_pack);
      if (hipMemcpyAsync(ctx->h_pack, ctx->d_pack, 4 * total, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
          hipStreamSynchronize(ctx->stream) != hipSuccess) { ctx->err = "path download failed"; return MNAV_INTERNAL_ERROR; }
    }
    MTRACE("paths downloaded");
    for (uint32_t k = 0; k < m; ++k) {
      const uint32_t i = map[k];
      const PlanResult& r = ctx->h_res[k];
      codes[i] = r.code;
      if (r.code == MNAV_SUCCESS) {
        if (path_len) path_len[i] = r.path_len;
        if (path_out && path_cap && r.path_len)                         // reference list order: seed ... pred[target]
          memcpy(path_out + (size_t)i * path_cap, ctx->h_pack + offs[k], 4 * (size_t)std::min(r.path_len, path_cap));
      }
      if (dist_out && hipMemcpyAsync(dist_out + (size_t)i * V, ctx->slots[k].dist, 4 * (size_t)V, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
        { ctx->err = "dist download failed"; return MNAV_INTERNAL_ERROR; }
      if (pred_out && hipMemcpyAsync(pred_out + (size_t)i * V, ctx->slots[k].pred, 4 * (size_t)V, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
        { ctx->err = "pred download failed"; return MNAV_INTERNAL_ERROR; }
      if (vecmap_out && want_vecmap && hipMemcpyAsync(vecmap_out + (size_t)i * 3 * V, ctx->slots[k].vecmap, 12 * (size_t)V, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
        { ctx->err = "vecmap download failed"; return MNAV_INTERNAL_ERROR; }
    }
    MTRACE("paths copied out");
    (void)hipEventRecord(ctx->ev[6], ctx->stream);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) { ctx->err = "sync failed"; return MNAV_INTERNAL_ERROR; }
    if (engine == 5 && tb_clean_other(ctx)) return MNAV_INTERNAL_ERROR;   // (the next call's distance buffer, in the gap between the calls)
    finish_stats(ctx, m, false);
    MTRACE("stats done");
  }
  // plans rejected before reaching the device: the reference has cleared its maps by then
  for (uint32_t i = 0; i < n; ++i) {
    if (codes[i] == MNAV_INVALID_START || codes[i] == MNAV_INVALID_GOAL || cleared[i]) {
      if (dist_out) for (uint32_t v = 0; v < V; ++v) dist_out[(size_t)i * V + v] = INFINITY;
      if (pred_out) for (uint32_t v = 0; v < V; ++v) pred_out[(size_t)i * V + v] = v;
    }
    if (codes_out) codes_out[i] = codes[i];
    if (codes[i] != MNAV_SUCCESS && worst == MNAV_SUCCESS) worst = codes[i];
  }
  return worst;
}

uint32_t mnav_plan_dijkstra(mnav_ctx* ctx, uint32_t seed_vertex, uint32_t target_vertex, double goal_dist_offset, double cost_limit,
                            float* dist_out, uint32_t* pred_out, uint32_t* path_out, uint32_t path_cap, uint32_t* path_len,
                            float* vecmap_out)
{
  if (!ctx) return MNAV_INTERNAL_ERROR;
  uint32_t code = MNAV_INTERNAL_ERROR;
  uint32_t len = 0;
  const uint32_t rc = dijkstra_impl(ctx, 1, &seed_vertex, &target_vertex, goal_dist_offset, cost_limit, &code, dist_out, pred_out,
                                    path_out, path_cap, &len, vecmap_out, true);
  if (path_len) *path_len = len;
  return rc == MNAV_INTERNAL_ERROR || rc == MNAV_CANCELED ? rc : code;
}

uint32_t mnav_plan_dijkstra_batch(mnav_ctx* ctx, uint32_t n, const uint32_t* seeds, const uint32_t* targets, double goal_dist_offset,
                                  double cost_limit, uint32_t* codes_out, float* dist_out, uint32_t* pred_out, uint32_t* path_out,
                                  uint32_t path_cap, uint32_t* path_len)
{
  if (!ctx) return MNAV_INTERNAL_ERROR;
  if (n == 0) return MNAV_SUCCESS;
  if (!seeds || !targets) { ctx->err = "null seeds/targets"; return MNAV_INTERNAL_ERROR; }
  return dijkstra_impl(ctx, n, seeds, targets, goal_dist_offset, cost_limit, codes_out, dist_out, pred_out, path_out, path_cap,
                       path_len, nullptr, false);
}

static uint32_t cvp_impl(mnav_ctx* ctx, uint32_t n, const float* seed_pos, const uint32_t* seed_faces, const uint32_t* target_faces,
                         double goal_dist_offset, double cost_limit, uint32_t* codes_out, float* dist_out, uint32_t* pred_out,
                         float* direction_out, uint32_t* cutface_out, float* vecmap_out)
{
  if (check_ready(ctx)) return MNAV_INTERNAL_ERROR;
  ctx->err.clear();
  if (goal_dist_offset != goal_dist_offset) { ctx->err = "goal_dist_offset is NaN"; return MNAV_INTERNAL_ERROR; }   // any double otherwise: a negative one
                                                                                                                      // stops the wave at the arming pop (passes_goal_cut)
  ctx->cancel.store(0);                                               // cvp :679
  if (ctx->d_cancel) { (void)hipStreamSynchronize(ctx->cancel_stream); (void)hipMemsetAsync(ctx->d_cancel, 0, 4, ctx->stream); }
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return MNAV_INTERNAL_ERROR; }
  const uint32_t V = ctx->V;
  if (!ctx->have_normals) { ctx->err = "vertex normals were not uploaded"; return MNAV_INTERNAL_ERROR; }
  if (V > (1u << kKeyIdBits)) { ctx->err = "CVP pop keys hold 26-bit vertex ids: mesh too large"; return MNAV_INTERNAL_ERROR; }
  std::vector<uint32_t> codes(n, MNAV_SUCCESS), map;
  std::vector<PlanIn> in;
  std::vector<float> sp;
  for (uint32_t i = 0; i < n; ++i) {
    if (seed_faces[i] >= ctx->F) { codes[i] = MNAV_INVALID_START; continue; }      // cvp :681-685
    if (target_faces[i] >= ctx->F) { codes[i] = MNAV_INVALID_GOAL; continue; }     // cvp :686-690
    PlanIn p{};
    for (int k = 0; k < 3; ++k) {
      p.seed[k] = ctx->h_faces[3 * (size_t)seed_faces[i] + k];
      p.target[k] = ctx->h_faces[3 * (size_t)target_faces[i] + k];
    }
    for (int k = 0; k < 3; ++k) {
      // cvp :721-723  diff = start - vertex; dist = diff.length()  (float arithmetic)
      const float dx = seed_pos[3 * (size_t)i] - ctx->h_xyz[3 * (size_t)p.seed[k]];
      const float dy = seed_pos[3 * (size_t)i + 1] - ctx->h_xyz[3 * (size_t)p.seed[k] + 1];
      const float dz = seed_pos[3 * (size_t)i + 2] - ctx->h_xyz[3 * (size_t)p.seed[k] + 2];
      const float l2 = dx * dx + dy * dy + dz * dz;
      p.seed_d[k] = sqrtf(l2);
      p.seed_expands[k] = (!((double)ctx->h_cost[p.seed[k]] >= cost_limit) && !ctx->h_invalid[p.seed[k]]) ? 1u : 0u;       // cvp :757,760
      p.target_expands[k] = (!((double)ctx->h_cost[p.target[k]] >= cost_limit) && !ctx->h_invalid[p.target[k]]) ? 1u : 0u;
    }
    p.seed_face = seed_faces[i];
    in.push_back(p); map.push_back(i);
    sp.insert(sp.end(), seed_pos + 3 * (size_t)i, seed_pos + 3 * (size_t)i + 3);
  }
  const uint32_t m = (uint32_t)in.size();
  ctx->caller_slot.assign(n, kNone);
  ctx->shard.finalized = false;
  for (size_t i = 0; i < map.size(); ++i) ctx->caller_slot[map[i]] = (uint32_t)i;
  (void)hipEventRecord(ctx->ev[0], ctx->stream);
  ctx->tb.count_pending = false; ctx->tb_args_valid = false;         // (a lazy settled-vertex count of an earlier Dijkstra batch is void now)
  ctx->last_planner = kPlannerCvp; ctx->last_n = 0; ctx->last_engine = 1;
  uint32_t worst = MNAV_SUCCESS;
  if (m) {
    if (materialize(ctx, true, cost_limit)) return MNAV_INTERNAL_ERROR;
    if (ctx->seed_pos_cap < m) {
      (void)hipFree(ctx->d_seed_pos); ctx->d_seed_pos = nullptr;
      if (hipMalloc((void**)&ctx->d_seed_pos, 12 * (size_t)m) != hipSuccess) { ctx->err = "alloc failed"; return MNAV_INTERNAL_ERROR; }
      ctx->seed_pos_cap = m;
    }
    if (hipMemcpyAsync(ctx->d_seed_pos, sp.data(), 12 * (size_t)m, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
      { ctx->err = "seed upload failed"; return MNAV_INTERNAL_ERROR; }
    const int rc = run_plans<kPlannerCvp>(ctx, m, in, goal_dist_offset, false);
    if (rc != 0) (void)hipStreamSynchronize(ctx->stream);
    if (rc < 0) return MNAV_INTERNAL_ERROR;
    if (rc == 1) { if (codes_out) for (uint32_t i = 0; i < n; ++i) codes_out[i] = MNAV_CANCELED; return MNAV_CANCELED; }   // cvp :888-892
    ctx->last_n = m;
    hipLaunchKernelGGL(k_finish<kPlannerCvp>, dim3(m), dim3(64), 0, ctx->stream, ctx->d_plans, ctx->d_res, PathRows{ nullptr, 0u, nullptr, nullptr });
    const uint32_t gc = (V + kBlock * 4 - 1) / (kBlock * 4);
    hipLaunchKernelGGL(k_count, dim3(gc ? gc : 1, m), dim3(kBlock), 0, ctx->stream, ctx->d_plans, ctx->d_res);
    (void)hipEventRecord(ctx->ev[4], ctx->stream);
    hipLaunchKernelGGL(k_vecmap_cvp, dim3(gc ? gc : 1, m), dim3(kBlock), 0, ctx->stream, ctx->d_plans, ctx->d_xyz, ctx->d_nrm,
                       ctx->d_vecptrs, ctx->d_seed_pos);                // cvp :897
    (void)hipEventRecord(ctx->ev[5], ctx->stream);
    bool ok = hipMemcpyAsync(ctx->h_res, ctx->d_res, sizeof(PlanResult) * m, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
    for (uint32_t k = 0; k < m && ok; ++k) {
      const uint32_t i = map[k];
      Slot& s = ctx->slots[k];
      if (ok && dist_out) ok = hipMemcpyAsync(dist_out + (size_t)i * V, s.dist, 4 * (size_t)V, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
      if (ok && pred_out) ok = hipMemcpyAsync(pred_out + (size_t)i * V, s.pred, 4 * (size_t)V, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
      if (ok && direction_out) ok = hipMemcpyAsync(direction_out + (size_t)i * V, s.dirn, 4 * (size_t)V, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
      if (ok && cutface_out) ok = hipMemcpyAsync(cutface_out + (size_t)i * V, s.cutf, 4 * (size_t)V, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
      if (ok && vecmap_out) ok = hipMemcpyAsync(vecmap_out + (size_t)i * 3 * V, s.vecmap, 12 * (size_t)V, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
    }
    (void)hipEventRecord(ctx->ev[6], ctx->stream);
    if (!ok || hipStreamSynchronize(ctx->stream) != hipSuccess) { ctx->err = "output download failed"; return MNAV_INTERNAL_ERROR; }
    finish_stats(ctx, m, true);
    for (uint32_t k = 0; k < m; ++k) {
      codes[map[k]] = ctx->h_res[k].code;
      if (ctx->h_res[k].code == MNAV_INTERNAL_ERROR) {
        const uint32_t o = ctx->h_res[k].overflow;
        ctx->err = (o & 8u) ? "CVP: a walk over the cascade tree hit its bound on the converged state (pop order not guaranteed)"
                 : (o & 16u) ? "CVP: verification sweep found a vertex that is not a fixed point of the gather rule"
                             : "CVP wavefront did not converge (step cap)";
      }
    }
  }
  for (uint32_t i = 0; i < n; ++i) {
    if (codes_out) codes_out[i] = codes[i];
    if (codes[i] != MNAV_SUCCESS && worst == MNAV_SUCCESS) worst = codes[i];
  }
  return worst;
}

uint32_t mnav_plan_cvp(mnav_ctx* ctx, const float seed_pos[3], uint32_t seed_face, uint32_t target_face, double goal_dist_offset,
                       double cost_limit, float* dist_out, uint32_t* pred_out, float* direction_out, uint32_t* cutface_out,
                       float* vecmap_out)
{
  if (!ctx) return MNAV_INTERNAL_ERROR;
  if (!seed_pos) return MNAV_INVALID_START;
  uint32_t code = MNAV_INTERNAL_ERROR;
  const uint32_t rc = cvp_impl(ctx, 1, seed_pos, &seed_face, &target_face, goal_dist_offset, cost_limit, &code, dist_out, pred_out,
                               direction_out, cutface_out, vecmap_out);
  return (rc == MNAV_INTERNAL_ERROR || rc == MNAV_CANCELED) ? rc : code;
}

uint32_t mnav_plan_cvp_batch(mnav_ctx* ctx, uint32_t n, const float* seed_pos, const uint32_t* seed_faces, const uint32_t* target_faces,
                             double goal_dist_offset, double cost_limit, uint32_t* codes_out, float* dist_out, uint32_t* pred_out,
                             float* vecmap_out)
{
  if (!ctx) return MNAV_INTERNAL_ERROR;
  if (n == 0) return MNAV_SUCCESS;
  if (!seed_pos || !seed_faces || !target_faces) { ctx->err = "null seeds/targets"; return MNAV_INTERNAL_ERROR; }
  return cvp_impl(ctx, n, seed_pos, seed_faces, target_faces, goal_dist_offset, cost_limit, codes_out, dist_out, pred_out, nullptr, nullptr,
                  vecmap_out);
}

#include "mnav_shard_capi.h"   // mnav_shard_* (one plan over several GPUs)

void mnav_cancel(mnav_ctx* ctx)
{
  if (!ctx) return;
  ctx->cancel.store(1, std::memory_order_relaxed);
  // thread-safe (MBF calls cancel() from the action-server thread), fire and forget: SDMA copy beside the running kernel
  if (ctx->d_cancel && ctx->h_one) (void)hipMemcpyAsync(ctx->d_cancel, ctx->h_one, 4, hipMemcpyHostToDevice, ctx->cancel_stream);
}

// The settled-vertex count of a paths-only tile-batch call is instrumentation (it only feeds mnav_stats.settled and
// mnav_algorithmic_bytes): it is taken from the resident distances on the first query after the call, not inside it.
static void settle_stats(mnav_ctx* ctx)
{
  if (!ctx->tb.count_pending) return;
  ctx->tb.count_pending = false;
  // only the arguments of a tile-batch call that succeeded and whose buffers are still allocated may be dereferenced
  if (!ctx->tb_args_valid || ctx->last_planner != kPlannerDijkstra || ctx->last_engine != 5 || !ctx->tb.built) return;
  const uint32_t m = ctx->last_n;
  if (!m || hipSetDevice(ctx->device) != hipSuccess) return;
  hipLaunchKernelGGL(k_tb_count, dim3(ctx->tb.ntiles ? ctx->tb.ntiles : 1, 16), dim3(kBlock), 0, ctx->stream, ctx->tb_args, ctx->tb.T, ctx->d_res);
  if (hipMemcpyAsync(ctx->h_res, ctx->d_res, sizeof(PlanResult) * m, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess) return;
  finish_stats(ctx, m, false);
}

int mnav_get_stats(const mnav_ctx* ctx, mnav_stats* out)
{
  if (!ctx || !out) return -1;
  settle_stats(const_cast<mnav_ctx*>(ctx));
  *out = ctx->stats;
  return 0;
}

int mnav_get_timing(const mnav_ctx* ctx, mnav_stats* out)
{
  if (!ctx || !out) return -1;
  *out = ctx->stats;
  if (ctx->tb.count_pending) out->settled = 0;
  return 0;
}

int mnav_set_band_width(mnav_ctx* ctx, float delta)
{
  if (!ctx) return -1;
  ctx->delta_user = delta > 0.f ? delta : 0.f;
  return 0;
}

int mnav_set_dijkstra_engine(mnav_ctx* ctx, int engine)
{
  if (!ctx || engine < 0 || engine > 6 || engine == 2 || engine == 4) return -1;   // 2: one workgroup per plan, 4: one wave per plan (both retired)
  ctx->dij_engine = engine;
  ctx->opt.dijkstra_engine = (double)engine;
  return 0;
}

int mnav_set_option(mnav_ctx* ctx, const char* name, double value)
{
  if (!ctx || !name) return -1;
  double* f = ctx->opt.find(name);
  if (!f) { ctx->err = std::string("unknown option: ") + name; return -1; }
  if (!strcmp(name, "dijkstra_engine") && value == value) {           // the retired engines are refused like by mnav_set_dijkstra_engine
    const int e = (int)value;
    if (value != (double)e || e < 0 || e > 6 || e == 2 || e == 4) { ctx->err = "dijkstra_engine: 0, 1, 3 (auto), 5 or 6"; return -1; }
  }
  *f = value;                                                         // NaN: back to the built-in default
  apply_options(ctx);
  drop_graphs(ctx);                                                   // (captured launch sequences may hold what the option decides)
  return 0;
}

double mnav_get_option(const mnav_ctx* ctx, const char* name)
{
  if (!ctx || !name) return NAN;
  const double* f = const_cast<mnav_ctx*>(ctx)->opt.find(name);
  return f ? *f : NAN;
}

const void* mnav_device_output(const mnav_ctx* ctx, uint32_t slot, int what)
{
  if (!ctx) return nullptr;
  if (slot >= ctx->caller_slot.size()) return nullptr;                    // not a plan of the last call (a stale slot of an earlier, larger batch is never handed out)
  slot = ctx->caller_slot[slot];                                          // caller's plan index -> device slot (kNone: never reached the device)
  if (slot >= ctx->last_n || slot >= ctx->slots.size()) return nullptr;   // last_n == 0: the last call failed or was cancelled
  const Slot& s = ctx->slots[slot];
  // a paths-only Dijkstra call finalized nothing: values beyond goal_dist are engine-tentative, predecessors were derived
  // along the path only (mnav_download_output what = 5 returns the popped potential of such a call)
  const bool lazy = ctx->lazy_paths && ctx->last_planner == kPlannerDijkstra;
  switch (what) {
    case 0: return lazy ? nullptr : s.dist;
    case 1: return lazy ? nullptr : s.pred;
    case 2: return s.dirn;
    case 3: return s.cutf;
    case 4: return (ctx->last_planner == kPlannerDijkstra && !ctx->want_vec) ? nullptr : s.vecmap;   // (a Dijkstra call that was not asked for vector maps
                                                                                                       //  must not hand out an earlier call's)
    default: return nullptr;
  }
}

int mnav_set_resident_outputs(mnav_ctx* ctx, int on)
{
  if (!ctx) return -1;
  ctx->resident_vecmap = on != 0;
  return 0;
}

int mnav_download_output(mnav_ctx* ctx, uint32_t slot, int what, void* host_out)
{
  if (!ctx || !host_out) return -1;
  if (what == 5) {                                                   // popped potential (Dijkstra): exact wherever dist <= goal_dist, +inf elsewhere
    if (ctx->last_planner != kPlannerDijkstra) { ctx->err = "popped potential: the last call was not a Dijkstra plan"; return -1; }
    if (slot < ctx->caller_slot.size()) slot = ctx->caller_slot[slot];
    if (slot >= ctx->last_n) { ctx->err = "output not resident"; return -1; }
    if (hipSetDevice(ctx->device) != hipSuccess) return -1;
    float* tmp = nullptr;
    HIPCHK(hipMalloc((void**)&tmp, 4 * (size_t)(ctx->V ? ctx->V : 1)));
    const uint32_t g = std::min<uint32_t>((ctx->V + kBlock - 1) / kBlock + 1, 4096);
    if (ctx->last_engine == 5 && ctx->lazy_paths && !ctx->tb_args_valid) { (void)hipFree(tmp); ctx->err = "output not resident"; return -1; }
    if (ctx->last_engine == 5 && ctx->lazy_paths) hipLaunchKernelGGL(k_tb_popped, dim3(g), dim3(kBlock), 0, ctx->stream, ctx->tb_args, slot, ctx->V, tmp);
    else hipLaunchKernelGGL(k_popped, dim3(g), dim3(kBlock), 0, ctx->stream, ctx->slots[slot].dist, ctx->last_target[slot], ctx->last_offset, ctx->V, tmp);
    const hipError_t e1 = hipMemcpyAsync(host_out, tmp, 4 * (size_t)ctx->V, hipMemcpyDeviceToHost, ctx->stream);
    const hipError_t e2 = hipStreamSynchronize(ctx->stream);
    (void)hipFree(tmp);
    if (e1 != hipSuccess || e2 != hipSuccess) { ctx->err = "popped potential: copy failed"; return -1; }
    return 0;
  }
  const void* src = mnav_device_output(ctx, slot, what);
  if (!src) { ctx->err = "output not resident"; return -1; }
  if (hipSetDevice(ctx->device) != hipSuccess) return -1;
  const size_t bytes = (what == 4 ? 12 : 4) * (size_t)ctx->V;
  HIPCHK(hipMemcpyAsync(host_out, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

// MeshMap::directionAtPosition (mesh_map.cpp:625-650) on the resident vector map of plan `slot`: what the controller
// samples at the robot pose.  36 bytes cross PCIe instead of the 12 MB field.  A vertex counts as "has a vector" when
// its entry is not the all-zero vector the vector-map kernels write for vertices the wave did not set.
int mnav_vector_at(mnav_ctx* ctx, uint32_t slot, const uint32_t vs[3], const float bary[3], float out[3])
{
  if (!ctx || !vs || !bary || !out) return -1;
  for (int k = 0; k < 3; ++k) if (vs[k] >= ctx->V) { ctx->err = "vertex id out of range"; return -1; }
  float v[3][3];
  if (ctx->last_planner == kPlannerDijkstra && ctx->last_engine == 5 && ctx->lazy_paths && ctx->tb_args_valid) {
    // paths-only batch of the tile-batch engine: no vector map was written -- the three entries are derived from the blocked distances
    uint32_t p = slot;
    if (p < ctx->caller_slot.size()) p = ctx->caller_slot[p];
    if (p >= ctx->last_n) { ctx->err = "output not resident"; return -1; }
    if (hipSetDevice(ctx->device) != hipSuccess) return -1;
    if (!ctx->d_vec3) HIPCHK(hipMalloc((void**)&ctx->d_vec3, 64));
    hipLaunchKernelGGL(k_tb_vector3, dim3(3), dim3(kWave), 0, ctx->stream, ctx->tb_args, ctx->d_row_ptr, ctx->d_nbr, ctx->d_xyz, p,
                       make_uint3(vs[0], vs[1], vs[2]), ctx->d_vec3);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(&v[0][0], ctx->d_vec3, 36, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  } else {
    const float* vm = static_cast<const float*>(mnav_device_output(ctx, slot, 4));
    if (!vm) { ctx->err = "vector map not resident"; return -1; }
    if (hipSetDevice(ctx->device) != hipSuccess) return -1;
    for (int k = 0; k < 3; ++k) HIPCHK(hipMemcpyAsync(v[k], vm + 3 * (size_t)vs[k], 12, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  float acc[3] = { 0.f, 0.f, 0.f };
  bool any = false;
  for (int k = 0; k < 3; ++k) {
    const bool has = !(v[k][0] == 0.f && v[k][1] == 0.f && v[k][2] == 0.f);
    if (!has) continue;
    any = true;
    for (int c = 0; c < 3; ++c) acc[c] += v[k][c] * bary[k];       // :636-638
  }
  if (!any || !(std::isfinite(acc[0]) && std::isfinite(acc[1]) && std::isfinite(acc[2]))) return 0;   // :639-646
  out[0] = acc[0]; out[1] = acc[1]; out[2] = acc[2];
  return 1;
}

// f3 of SURVEY.md section 8: the consumer of the CVP vector field on the device.  Plan i of the last mnav_plan_cvp(_batch)
// call is walked from its target (the robot) back to its seed (the goal) over the vector map resident in HBM; what
// crosses PCIe is the path, not the 12 B/vertex field.
int mnav_backtrack_cvp_batch(mnav_ctx* ctx, uint32_t n, const float* seed_pos, const uint32_t* seed_faces, const float* target_pos,
                             const uint32_t* target_faces, double step_width, int32_t inflation_layer, uint32_t cap, float* positions_out,
                             uint32_t* faces_out, uint32_t* n_out, int32_t* status_out)
{
  if (!ctx) return -1;
  ctx->err.clear();
  if (!n) return 0;
  if (!seed_pos || !seed_faces || !target_pos || !target_faces || !positions_out || !faces_out || !n_out || !status_out) { ctx->err = "null argument"; return -1; }
  if (cap < 2 || (uint64_t)cap * n > (1ull << 28)) { ctx->err = "path capacity out of range"; return -1; }
  if (!(step_width > 0.0)) { ctx->err = "step_width must be positive"; return -1; }   // a zero step never leaves the start
  if (ctx->last_planner != kPlannerCvp) { ctx->err = "back-tracking: the last call was not a CVP plan"; return -1; }
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return -1; }
  if (n != ctx->caller_slot.size()) { ctx->err = "back-tracking: n differs from the last mnav_plan_cvp(_batch) call"; return -1; }
  std::vector<WalkJob> jobs(n);
  for (uint32_t i = 0; i < n; ++i) {
    const float* vm = static_cast<const float*>(mnav_device_output(ctx, i, 4));
    jobs[i].vecmap = nullptr;
    if (ctx->caller_slot[i] == kNone) continue;                       // a plan rejected before it reached the device (INVALID_START / _GOAL): status 0, no entries
    if (!vm) { ctx->err = "vector map not resident (mnav_set_resident_outputs, or pass vecmap_out to the plan call)"; return -1; }
    if (seed_faces[i] >= ctx->F || target_faces[i] >= ctx->F) { ctx->err = "face id out of range"; return -1; }
    jobs[i].vecmap = vm; jobs[i].seed_face = seed_faces[i]; jobs[i].target_face = target_faces[i];
    for (int k = 0; k < 3; ++k) { jobs[i].seed[k] = seed_pos[3 * (size_t)i + k]; jobs[i].target[k] = target_pos[3 * (size_t)i + k]; }
  }
  WalkInflation L{};
  if (inflation_layer >= 0) {
    if ((size_t)inflation_layer >= ctx->layers.size() || !ctx->layers[inflation_layer].ready || !ctx->layers[inflation_layer].dist || !ctx->layers[inflation_layer].have_vec) {
      ctx->err = "back-tracking: not a resident inflation layer with a vector field"; return -1;
    }
    const mnav_ctx::Layer& Ly = ctx->layers[inflation_layer];
    L.distances = Ly.dist; L.vectors = Ly.vec; L.has_vector = Ly.vstate;
    L.inflation_radius = Ly.inflation_radius; L.inscribed_radius = Ly.inscribed_radius; L.inscribed_value = Ly.inscribed_value; L.lethal_value = Ly.lethal_value;
    L.repulsive_field = 1;
  }
  if (!ctx->walk_mesh_valid) {
    if (dev_upload(ctx, &ctx->d_faces, ctx->h_faces.data(), ctx->h_faces.size())) return -1;
    if (dev_upload(ctx, &ctx->d_vf_ptr, ctx->h_vf_ptr.data(), ctx->h_vf_ptr.size())) return -1;
    if (dev_upload(ctx, &ctx->d_vf, ctx->h_vf.data(), ctx->h_vf.size())) return -1;
    ctx->walk_mesh_valid = true;
  }
  const size_t need = (size_t)cap * n;
  if (need > ctx->walk_cap) {
    (void)hipFree(ctx->d_walk_pos); (void)hipFree(ctx->d_walk_face); ctx->d_walk_pos = nullptr; ctx->d_walk_face = nullptr; ctx->walk_cap = 0;
    HIPCHK(hipMalloc((void**)&ctx->d_walk_pos, 12 * need)); HIPCHK(hipMalloc((void**)&ctx->d_walk_face, 4 * need));
    ctx->walk_cap = need;
  }
  if (n > ctx->walk_jobs_cap) {
    (void)hipFree(ctx->d_walk_jobs); (void)hipFree(ctx->d_walk_ctl); ctx->d_walk_jobs = nullptr; ctx->d_walk_ctl = nullptr; ctx->walk_jobs_cap = 0;
    HIPCHK(hipMalloc((void**)&ctx->d_walk_jobs, sizeof(WalkJob) * (size_t)n)); HIPCHK(hipMalloc((void**)&ctx->d_walk_ctl, 8 * (size_t)n));
    ctx->walk_jobs_cap = n;
  }
  HIPCHK(hipMemcpyAsync(ctx->d_walk_jobs, jobs.data(), sizeof(WalkJob) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  WalkMesh M{ ctx->d_xyz, ctx->d_faces, ctx->d_vf_ptr, ctx->d_vf, ctx->V, ctx->F };
  hipLaunchKernelGGL(k_backtrack, dim3(n), dim3(64), 0, ctx->stream, M, L, ctx->d_walk_jobs, step_width, cap, ctx->d_walk_pos, ctx->d_walk_face, ctx->d_walk_ctl);
  HIPCHK(hipGetLastError());
  std::vector<int32_t> ctl(2 * (size_t)n);
  HIPCHK(hipMemcpyAsync(ctl.data(), ctx->d_walk_ctl, 8 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  for (uint32_t i = 0; i < n; ++i) {                                  // the walked entries only: O(path) bytes per plan
    const uint32_t m = (uint32_t)ctl[2 * (size_t)i + 1];
    if (m) {
      HIPCHK(hipMemcpyAsync(positions_out + 3 * (size_t)cap * i, ctx->d_walk_pos + 3 * (size_t)cap * i, 12 * (size_t)m, hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(hipMemcpyAsync(faces_out + (size_t)cap * i, ctx->d_walk_face + (size_t)cap * i, 4 * (size_t)m, hipMemcpyDeviceToHost, ctx->stream));
    }
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t m = (uint32_t)ctl[2 * (size_t)i + 1];
    float* pp = positions_out + 3 * (size_t)cap * i; uint32_t* pf = faces_out + (size_t)cap * i;
    for (uint32_t a = 0, b = m ? m - 1 : 0; a < b; ++a, --b) {        // the reference push_front()s: list order is seed first
      for (int k = 0; k < 3; ++k) std::swap(pp[3 * (size_t)a + k], pp[3 * (size_t)b + k]);
      std::swap(pf[a], pf[b]);
    }
    n_out[i] = m; status_out[i] = ctl[2 * (size_t)i];
  }
  return 0;
}

int mnav_backtrack_cvp(mnav_ctx* ctx, const float seed_pos[3], uint32_t seed_face, const float target_pos[3], uint32_t target_face, double step_width,
                       int32_t inflation_layer, uint32_t cap, float* positions_out, uint32_t* faces_out, uint32_t* n_out)
{
  int32_t status = 0;
  const int rc = mnav_backtrack_cvp_batch(ctx, 1, seed_pos, &seed_face, target_pos, &target_face, step_width, inflation_layer, cap, positions_out, faces_out, n_out, &status);
  return rc < 0 ? rc : status;
}

uint64_t mnav_algorithmic_bytes(const mnav_ctx* ctx)
{
  if (!ctx) return 0;
  settle_stats(const_cast<mnav_ctx*>(ctx));
  return ctx->algo_bytes;
}

int mnav_last_engine(const mnav_ctx* ctx)
{
  if (!ctx || ctx->last_planner != kPlannerDijkstra || ctx->last_n == 0) return -1;
  return ctx->last_engine + ((ctx->last_engine == 5 && ctx->tb.kernel == 1) ? 16 : 0);
}

#ifdef MNAV_TILE_TIMING
int mnav_debug_tile_timing(unsigned long long* out, unsigned int cap)
{
  unsigned int n = 0;
  (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_tile_timing_n), sizeof(n));
  if (n > 4096) n = 4096;
  if (n > cap) n = cap;
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tile_timing), sizeof(unsigned long long) * 8 * n);
  unsigned int z = 0;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tile_timing_n), &z, sizeof(z));
  return (int)n;
}
#endif

}  // extern "C"

#ifdef MNAV_WIDE_TIMING
extern "C" int mnav_debug_wide_timing(unsigned long long* out)
{
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wide_timing), sizeof(unsigned long long) * 12) != hipSuccess) return -1;
  unsigned long long z[12] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
  return hipMemcpyToSymbol(HIP_SYMBOL(g_wide_timing), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
