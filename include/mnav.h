/*
 * mnav.h -- C ABI of the MI355X-native wavefront planner (libmnav.so).
 *
 * The reference has no C ABI: its boundary is the C++ pure-virtual plugin class
 * mbf_mesh_core::MeshPlanner (mbf_mesh_core/include/mbf_mesh_core/mesh_planner.h:50-92)
 * loaded through pluginlib.  This header is what a MeshPlanner implementation binds
 * instead of running the priority-queue loops itself; INTEGRATION.md shows the
 * adapter (mesh_navigation_amd/csrc/adapter/) that keeps makePlan / cancel /
 * initialize unchanged on top of it.
 *
 * Conventions: plain pointers and sizes only; the caller owns every buffer it
 * passes; the context owns all device memory; one plan call in flight per
 * context (the reference reuses potential_/predecessors_ members the same way,
 * dijkstra_mesh_planner.h:189-197).  Vertex / face / edge ids are the
 * reference's handles' idx() values.  All plan functions return the MBF GetPath
 * result codes of dijkstra_mesh_planner.h:72-85.  The library never computes on
 * the CPU: without a usable GPU mnav_create() fails (returns NULL).
 */
#ifndef MNAV_H
#define MNAV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MNAV_SUCCESS 0u         /* mbf_msgs GetPath::Result::SUCCESS        */
#define MNAV_CANCELED 51u       /* ...::CANCELED                            */
#define MNAV_INVALID_START 52u  /* ...::INVALID_START                       */
#define MNAV_INVALID_GOAL 53u   /* ...::INVALID_GOAL                        */
#define MNAV_NO_PATH_FOUND 54u  /* ...::NO_PATH_FOUND                       */
#define MNAV_INTERNAL_ERROR 60u /* ...::INTERNAL_ERROR (device/runtime failure) */
#define MNAV_NONE 0xFFFFFFFFu

typedef struct mnav_ctx mnav_ctx;

/* Per-call statistics (the phases the reference logs, dijkstra_mesh_planner.cpp:391-394,
 * cvp_mesh_planner.cpp:956-960).  Times are HIP-event milliseconds on the context stream. */
typedef struct mnav_stats {
  uint32_t steps;          /* wavefront step-kernel launches that did work       */
  uint32_t launches;       /* step launches enqueued (incl. overshoot no-ops)    */
  uint32_t bands;          /* distance bands completed                           */
  uint32_t armed;          /* goal_dist was armed                                */
  float goal_dist;         /* armed value or +inf                                */
  uint32_t n_plans;        /* plans in the call (batch size)                     */
  uint64_t evals;          /* vertex evaluations (all plans)                     */
  uint64_t settled;        /* vertices with a finite potential (all plans)       */
  float ms_init;           /* state initialisation                               */
  float ms_propagation;    /* wavefront propagation (all step launches)          */
  float ms_vector_map;     /* computeVectorMap                                   */
  float ms_path;           /* predecessor walk                                   */
  float ms_download;       /* device -> host copies of requested outputs         */
  float ms_total;          /* whole call                                         */
  float ms_step_kernels;   /* sum over graph replays of the event-bracketed step/round launches
                              (excludes the host polls between replays)          */
  uint32_t band_shrinks;   /* bands cut down because they did not converge (all plans)   */
  uint32_t band_cuts;      /* bands restarted under a lower bound after kCutAfter steps     */
} mnav_stats;

/* -- life cycle -------------------------------------------------------------------------- */
/* Replaces: plugin construction + MeshPlanner::initialize(name, mesh_map, node),
 * mesh_planner.h:88 (dijkstra_mesh_planner.cpp:142-169, cvp_mesh_planner.cpp:148-186). */
mnav_ctx* mnav_create(int device);
void mnav_destroy(mnav_ctx* ctx);
/* Text of the last failure on this context ("" if none); valid until the next call. */
const char* mnav_last_error(const mnav_ctx* ctx);

/* Upload the half-edge mesh once (MeshMap::mesh(), mesh_map.h:276-279) as flat arrays:
 * xyz V*3, face_vtx F*3 (reference face order), edge_vtx E*2 (reference edge ids),
 * vertex_normals V*3 (MeshMap::vertexNormals(), mesh_map.h:326-337; may be NULL if only the
 * Dijkstra planner is used: mnav_plan_cvp needs them for its vector map).  Returns 0 on success,
 * <0 on error. */
int mnav_upload_mesh(mnav_ctx* ctx, uint32_t V, uint32_t F, uint32_t E, const float* xyz,
                     const uint32_t* face_vtx, const uint32_t* edge_vtx,
                     const float* vertex_normals);

/* Optional, BEFORE mnav_upload_mesh: the rows of lvr2::PMPMesh::getFacesOfVertex (half-edge circulator
 * order) as CSR, vf_ptr V+1, vf 3F.  CVPMeshPlanner applies the faces of a popped vertex in that order
 * (cvp_mesh_planner.cpp:775-778) and its update is not a pure minimum on cost-inflated triangles, so the
 * order is part of the result.  Without this call (or with NULL rows) the library derives the order itself
 * by replaying pmp::SurfaceMesh::add_face over face_vtx in index order, which is how the reference builds
 * its mesh from the map file (mesh_map.cpp:273).  Returns 0. */
int mnav_set_face_circulation(mnav_ctx* ctx, uint32_t V, uint32_t F, const uint32_t* vf_ptr,
                              const uint32_t* vf);

/* Upload the inputs the planners re-read on every plan (dijkstra_mesh_planner.cpp:214,
 * cvp_mesh_planner.cpp:245): vertex_costs V (MeshMap::vertexCosts(), mesh_map.h:292-295),
 * edge_weights E (MeshMap::edgeWeights(), mesh_map.h:342-345), invalid V bytes
 * (MeshMap::invalid, mesh_map.h:447; NULL = none). */
int mnav_upload_costs(mnav_ctx* ctx, const float* vertex_costs, const float* edge_weights,
                      const uint8_t* invalid);

/* Device version of MeshMap::computeEdgeWeights (mesh_map/src/mesh_map.cpp:517-561): uploads
 * vertex_costs V and edge_distances E and derives the edge weights on the GPU with the
 * reference's mixed float/double arithmetic.  edge_weights_out (E, may be NULL) receives them. */
int mnav_compute_edge_weights(mnav_ctx* ctx, const float* vertex_costs, const float* edge_distances,
                              double edge_cost_factor, const uint8_t* invalid,
                              float* edge_weights_out);

/* Device version of the combination layers that produce the planners' vertex costs
 * (mesh_layers/src/combination_layer.cpp:44-85 MaxCombinationLayer, :185-248 AvgCombinationLayer; the
 * default layer copied by MeshMap::copyVertexCostsFromDefaultLayer, mesh_map.cpp:495-515), followed by
 * the edge weights of mnav_compute_edge_weights: mode 0 = max, 1 = weighted sum (weights n_layers,
 * AbstractLayer::combinationWeight()); layer_costs = n_layers dense V-sized arrays (missing entries
 * already replaced by the layer's default value), combined in the given order starting from 0.
 * The result becomes the context's vertex costs / edge weights; vertex_costs_out V and
 * edge_weights_out E (either may be NULL) receive copies. */
int mnav_combine_costs(mnav_ctx* ctx, int mode, uint32_t n_layers, const float* const* layer_costs,
                       const float* weights, const float* edge_distances, double edge_cost_factor,
                       const uint8_t* invalid, float* vertex_costs_out, float* edge_weights_out);

/* -- planning ---------------------------------------------------------------------------- */
/* Replaces DijkstraMeshPlanner::dijkstra (7-arg) + computeVectorMap,
 * dijkstra_mesh_planner.cpp:217-398, :189-209.  seed_vertex = wave seed (navigation goal),
 * target_vertex = robot vertex (both already resolved by MeshMap::getNearestVertexHandle,
 * :235-236).  Outputs (any may be NULL): dist_out V (potential_), pred_out V
 * (predecessors_), vecmap_out V*3 (vector_map_, zero rows where the reference has no entry),
 * path_out/path_len: the vertex path in dijkstra()'s list order (seed first ... pred[target]),
 * at most path_cap entries are written, *path_len is the full length.
 * goal_dist_offset: any double like the reference's parameter (default 0.3).  A negative value stops the expansion at the
 * robot vertex (only vertices popped before it are sources, :293-300) and is reproduced exactly; NaN is refused.  The CVP
 * and sharded entry points below take negative offsets too (CVP: every pop up to and including the arming one expands,
 * cvp_mesh_planner.cpp:754 before :765-769). */
uint32_t mnav_plan_dijkstra(mnav_ctx* ctx, uint32_t seed_vertex, uint32_t target_vertex,
                            double goal_dist_offset, double cost_limit, float* dist_out,
                            uint32_t* pred_out, uint32_t* path_out, uint32_t path_cap,
                            uint32_t* path_len, float* vecmap_out);

/* Replaces CVPMeshPlanner::waveFrontPropagation up to and including computeVectorMap,
 * cvp_mesh_planner.cpp:651-918, :204-239 (the vector-field back-tracking :920-951 stays on
 * the host, see the adapter).  seed_pos = exact wave seed position (navigation goal),
 * seed_face / target_face = containing faces (MeshMap::getContainingFace, :673-674).
 * Outputs (any may be NULL): dist_out V, pred_out V, direction_out V (direction_),
 * cutface_out V (cutting_faces_, MNAV_NONE = no entry), vecmap_out V*3.  Entries of
 * direction/cutface/vecmap for vertices the wave did not update are 0 / MNAV_NONE / 0
 * (the reference leaves stale values of earlier plans there, cvp_mesh_planner.cpp:179). */
uint32_t mnav_plan_cvp(mnav_ctx* ctx, const float seed_pos[3], uint32_t seed_face,
                       uint32_t target_face, double goal_dist_offset, double cost_limit,
                       float* dist_out, uint32_t* pred_out, float* direction_out,
                       uint32_t* cutface_out, float* vecmap_out);

/* n independent Dijkstra plans on the same mesh in one sweep (BASELINE config 5: concurrent
 * goals).  seeds/targets: n each.  codes_out n.  dist_out/pred_out: n*V or NULL.
 * path_out: n*path_cap or NULL, path_len n.  * A call that asks for nothing V-sized (dist_out, pred_out, vecmap_out all NULL, resident outputs off) only derives the
 * predecessors along the returned path (no finalize pass over the touched tiles): mnav_device_output(slot, pred) is
 * NULL afterwards and the resident potential is final up to goal_dist only.  MNAV_LAZY_PATHS=0 restores the full pass.
 */
uint32_t mnav_plan_dijkstra_batch(mnav_ctx* ctx, uint32_t n, const uint32_t* seeds,
                                  const uint32_t* targets, double goal_dist_offset,
                                  double cost_limit, uint32_t* codes_out, float* dist_out,
                                  uint32_t* pred_out, uint32_t* path_out, uint32_t path_cap,
                                  uint32_t* path_len);

/* n independent CVP plans on the same mesh in one sweep.  seed_pos n*3, seed_faces / target_faces n,
 * codes_out n; dist_out / pred_out n*V, vecmap_out n*V*3 or NULL. */
uint32_t mnav_plan_cvp_batch(mnav_ctx* ctx, uint32_t n, const float* seed_pos, const uint32_t* seed_faces,
                             const uint32_t* target_faces, double goal_dist_offset, double cost_limit,
                             uint32_t* codes_out, float* dist_out, uint32_t* pred_out, float* vecmap_out);

/* Incremental cost change: MeshMap::layerChanged (mesh_map.cpp:454-493) + updateEdgeWeights(changed) (:563-618) on
 * the device.  The n vertices get their new costs; if the resident weights were computed here with a non-zero
 * edge_cost_factor (mnav_compute_edge_weights / mnav_combine_costs) the edges around them are re-weighted with the
 * same mixed float/double expression (:606-610), otherwise only the vertex costs change ("edge_cost_factor is 0,
 * skipping edge cost update", :568-572).  Nothing but the n ids and values crosses PCIe.  Returns 0 / <0. */
int mnav_update_costs(mnav_ctx* ctx, uint32_t n, const uint32_t* vertex_ids, const float* values);
/* The caller's own incremental edge weights (MeshMap::updateEdgeWeights, mesh_map.cpp:563-618, already run on the host by
 * MeshMap::layerChanged :454-493): `values[i]` replaces the weight of edge `edge_ids[i]`.  Together with mnav_update_costs on
 * weights that were uploaded (mnav_upload_costs: then mnav_update_costs only touches the vertex costs) a cost change of the
 * map costs O(changed) instead of a pass over all vertices and edges. */
int mnav_update_edge_weights(mnav_ctx* ctx, uint32_t n, const uint32_t* edge_ids, const float* values);
/* Copies of the resident vertex costs (V) and edge weights (E); either pointer may be NULL. */
int mnav_download_costs(mnav_ctx* ctx, float* vertex_costs_out, float* edge_weights_out);

/* -- cost layers on the device (mesh_layers) ------------------------------------------------------
 * The reference's layer plugins compute per-vertex costs on the CPU (AbstractLayer::computeLayer) and MeshMap
 * combines them.  These entry points keep that stack in HBM: layer `k` (0..63) is a resident cost array + lethal set.
 *   mnav_layer_upload      a layer computed elsewhere (ObstacleLayer, a costs file ...): costs V floats, lethal V bytes
 *                          or NULL
 *   mnav_layer_steepness   SteepnessLayer::computeLayer (steepness_layer.cpp:157-166) from the resident vertex normals:
 *                          cost = acos(n.z) in float, lethal when > threshold (:82-93)
 *   mnav_layer_inflation   InflationLayer::computeLayer (inflation_layer.cpp:96-178): the lethal set of `input_layer`
 *                          is the source of waveCostInflation (:341-491), a multi-source ordered fast-marching wave over
 *                          the edge distances, run here on the same ordered-wave engine as the CVP planner (replay of the
 *                          face updates in the reference's pop order, float32 Sethian update :181-234, re-queue rule
 *                          :311, invalid vertices never fixed :417); then riskiness = fading(distance) (:315-339).
 *                          `invalid` = the map's non-manifold flags (V bytes) or NULL.  Distances are bit-identical to
 *                          the reference's; a converged state that fails the verification sweep returns <0 instead of a
 *                          result; it works in the first plan slot, so the resident outputs of the last plan are gone
 *                          afterwards (mnav_download_output fails until the next plan).  The layer's repulsive vector field (vector_map_, :277-309: an order-dependent
 *                          accumulation over the lethal contours, then assignments in pop order) is computed as well
 *                          (mnav_layer_download_vectors), under the (value, id) heap-tie convention of the library.
 *   mnav_layer_download    copies of a layer's costs / lethal flags / wave distances (NULL to skip; distances only for an
 *                          inflation layer, +inf where the wave never arrived)
 *   mnav_combine_layers    CombinationLayer (mode 0 = max :44-85, 1 = weighted sum :185-248) over resident layers, then
 *                          MeshMap::computeEdgeWeights (mesh_map.cpp:495-561): the resident vertex costs and edge weights
 *                          the planners read are replaced; edge distances are computed on the device if none are resident
 * All return 0 / <0 (mnav_last_error). */
int mnav_layer_upload(mnav_ctx* ctx, uint32_t layer, const float* costs, const uint8_t* lethal);
int mnav_layer_steepness(mnav_ctx* ctx, uint32_t layer, double threshold);
int mnav_layer_inflation(mnav_ctx* ctx, uint32_t layer, uint32_t input_layer, double inflation_radius, double inscribed_radius,
                         double inscribed_value, double lethal_value, double cost_scaling_factor, const uint8_t* invalid);
int mnav_layer_download(mnav_ctx* ctx, uint32_t layer, float* costs_out, uint8_t* lethal_out, float* distances_out);
/* vector_map_ of an inflation layer: V*3 floats and V flags (1 = the reference's map holds an entry); NULL to skip. */
int mnav_layer_download_vectors(mnav_ctx* ctx, uint32_t layer, float* vectors_out, uint8_t* has_vector_out);
int mnav_combine_layers(mnav_ctx* ctx, int mode, uint32_t n_layers, const uint32_t* layers, const float* weights,
                        double edge_cost_factor, const uint8_t* invalid);
/* CombinationLayer::onInputChanged (combination_layer.cpp:87-147, :250-302) + MeshMap::layerChanged (mesh_map.cpp:454-493)
 * + updateEdgeWeights(changed) (:563-618): after a layer changed on n vertices (re-uploaded or recomputed on the device)
 * only those vertices are recombined and only the edges around them re-weighted.  Same mode / layers / weights as the full
 * mnav_combine_layers that came before. */
int mnav_combine_layers_update(mnav_ctx* ctx, int mode, uint32_t n_layers, const uint32_t* layers, const float* weights, uint32_t n,
                               const uint32_t* vertex_ids);
/* Counters of the last inflation wave: band steps, bands, vertex evaluations, device milliseconds (whole call), fixing
 * verification sweeps that were needed, device milliseconds of the wave alone.  Any pointer may be NULL. */
int mnav_layer_stats(const mnav_ctx* ctx, uint32_t* steps, uint32_t* bands, uint64_t* evals, float* ms, uint32_t* verify_sweeps,
                     float* ms_wave);

/* -- one plan over several GPUs (BASELINE config 4) ---------------------------------------------
 * The reference's loop (dijkstra_mesh_planner.cpp:287-348) on a mesh that is range-partitioned over `world`
 * processes, one per GPU: the LDS tiles are in Morton order and process `rank` owns a contiguous range of them.
 * Every process uploads the same mesh and costs, then
 *   n = mnav_shard_setup(ctx, rank, world)            floats in the exchange buffer (interface vertices + robot vertex)
 *   mnav_shard_begin(ctx, seed, target, offset, limit)
 *   repeat { mnav_shard_rounds(ctx, R, buf);           R local tile rounds, then buf[i] = own interface values / +inf
 *            min-allreduce(buf) over the processes      (RCCL over xGMI; ncclAllReduce(ncclMin) / torch.distributed)
 *            mnav_shard_apply(ctx, buf, &local_min, &target_dist);
 *          } until min-allreduce(local_min) is +inf or > target_dist + offset
 *   mnav_shard_finalize(ctx, dist_buf, pred_buf);       owned entries, +inf / 0xFFFFFFFF elsewhere: min-allreduce both
 * `buf`, `dist_buf` (V floats) and `pred_buf` (V uint32) are DEVICE pointers owned by the caller (the collective runs on
 * them in place).  The result is bit-identical to the single-GPU plan: the schedule is label-correcting, only the
 * fixed point matters.  Returns 0, <0 on error, 1 if cancelled. */
int mnav_shard_setup(mnav_ctx* ctx, uint32_t rank, uint32_t world);
int mnav_shard_info(const mnav_ctx* ctx, uint32_t* t_lo, uint32_t* t_hi, uint32_t* ntiles, uint32_t* n_exchange);
/* The same loop on a mesh whose DATA is partitioned (north_star: "the mesh is range-partitioned across the 8 GPUs ...
 * allreduce of halo-vertex distances only"): the mesh uploaded to this context is ONE PART -- the vertices this process
 * owns plus their 1-ring halo (the neighbours owned elsewhere), renumbered in ascending global id, with every edge that has
 * an owned endpoint.  `exchange_vertex[i]` (i < n_exchange, the same global list of interface vertices on every process:
 * the vertices that have a neighbour owned by another process) is the LOCAL id of interface vertex i, or 0xFFFFFFFF when
 * this process does not hold it; `owned[v]` (one byte per local vertex) is 1 for owned vertices, 0 for halo copies.
 * The exchange buffer has n_exchange + 1 floats (last: the robot vertex, passed to mnav_shard_begin as a local id).  All
 * local tiles run; every held copy of an interface vertex is packed (a value reached along real edges is an upper bound of
 * the true distance) and takes the reduced minimum; mnav_shard_finalize returns dist / pred of the LOCAL vertices (the
 * owned ones are final, predecessors are local ids), sized by the part, not by the mesh.  Returns n_exchange + 1. */
int mnav_shard_setup_partition(mnav_ctx* ctx, uint32_t n_exchange, const uint32_t* exchange_vertex, const uint8_t* owned);
/* After mnav_shard_finalize: one segment of the vertex path (dijkstra_mesh_planner.cpp:358-373) inside this process's part.
 * Predecessors are followed from `start_vertex` (local id) while the vertex is owned here, at most `cap` hops:
 * out_host[0] = hops, out_host[1] = the vertex the walk stopped at (the seed, or a halo copy: its owner continues),
 * out_host[2] = 1 if a vertex without predecessor was met (the wave never reached it), out_host[3..] = the predecessors
 * visited (local ids).  `out_host` holds cap + 3 words.  The potential / predecessor arrays never leave the device. */
int mnav_shard_walk(mnav_ctx* ctx, uint32_t start_vertex, uint32_t seed_vertex, uint32_t cap, uint32_t* out_host);
/* Device memory this context holds for mesh tables and per-plan state, in bytes (the partitioned plan's footprint test). */
uint64_t mnav_device_bytes(const mnav_ctx* ctx);

int mnav_shard_begin(mnav_ctx* ctx, uint32_t seed_vertex, uint32_t target_vertex, double goal_dist_offset, double cost_limit);
/* Partitioned data, negative goal_dist_offset only: with such an offset the reference expands exactly the vertices popped BEFORE
 * the robot vertex (dijkstra_mesh_planner.cpp:293-300), and among vertices of the robot vertex's potential the vertex id decides.
 * A part that does not hold the robot vertex passes the robot's RANK among its own (ascending) ids -- the number of local vertices
 * with a smaller global id -- before mnav_shard_begin; parts that hold it need not call this. */
int mnav_shard_set_goal_tie(mnav_ctx* ctx, uint32_t tie_id);
int mnav_shard_rounds(mnav_ctx* ctx, uint32_t rounds, float* iface_buf_dev);
int mnav_shard_apply(mnav_ctx* ctx, const float* iface_buf_dev, float* local_min_out, float* target_dist_out);
int mnav_shard_finalize(mnav_ctx* ctx, float* dist_buf_dev, uint32_t* pred_buf_dev);
/* The two steps of an exchange without a host round trip.  `caller_stream` (a hipStream_t) is the stream the caller's
 * collectives are ordered on; the library links its own stream to it with events in both directions, nothing waits on the
 * host.  mnav_shard_apply_async writes {smallest pending wake-up, dist[target], -1 if mnav_cancel arrived else 0} to the
 * DEVICE words ctl_dev[0..2]; the caller min-allreduces them and reads them back once every few exchanges -- an exchange
 * after convergence changes nothing, so checking late is safe. */
int mnav_shard_rounds_async(mnav_ctx* ctx, uint32_t rounds, float* iface_buf_dev, void* caller_stream);
int mnav_shard_apply_async(mnav_ctx* ctx, const float* iface_buf_dev, float* ctl_dev, void* caller_stream);

/* Replaces MeshPlanner::cancel(), mesh_planner.h:80 (dijkstra_mesh_planner.cpp:136-140):
 * async-signal/thread safe, only sets a flag that the running plan polls between step
 * batches; the plan then returns MNAV_CANCELED.  The flag is cleared when a plan starts
 * (dijkstra_mesh_planner.cpp:238). */
void mnav_cancel(mnav_ctx* ctx);

/* -- introspection / tuning -------------------------------------------------------------- */
int mnav_get_stats(const mnav_ctx* ctx, mnav_stats* out);
/* The same without the settled-vertex count when that has not been taken yet: after a paths-only batch of the tile-batch
 * engine the count (instrumentation for mnav_algorithmic_bytes) is made by the first mnav_get_stats / mnav_algorithmic_bytes
 * call, from the resident distances -- mnav_get_timing never triggers it and reports settled = 0 until then. */
int mnav_get_timing(const mnav_ctx* ctx, mnav_stats* out);
/* Band width of the wavefront engine in potential units; <= 0 selects the default
 * (3 x mean finite edge weight for the Dijkstra band steps, 12 x for CVP, recomputed on every cost
 * upload).  Results do not depend on it. */
int mnav_set_band_width(mnav_ctx* ctx, float delta);
/* Schedule of the Dijkstra planner: 0 = LDS-tiled label-correcting rounds (one launch per round),
 * 1 = the distance-band gather steps that the CVP planner uses,
 * (2, one persistent workgroup per plan, was retired in round 5: -1),
 * 3 = automatic (default): 6 for calls of up to `async_max_batch` = 160 plans (a call whose ticket ring overflows is re-run on 0),
 *     5 beyond that (batches that also hold >= tiles/1000 plans), else 0,
 * 5 = tile-batch: one plan per lane, the tile's graph as record streams (highest throughput for large batches); its tiles are
 *     solved by k_tbv_solve (one wave per tile, <= 64 plans, the distances in VGPRs) when a tile sees enough plans per iteration
 *     (plans / sqrt(tiles) >= 8, option "tb_kernel" overrides), else by k_tb_solve_q (16 plans per quarter of a wave, LDS),
 * 6 = the LDS tiles without rounds: resident workgroups serve a ticket queue of woken tiles, solve and wake tiles
 *     asynchronously, one launch per call (single plans -- what MeshPlanner::makePlan runs -- and batches up to ~100 plans).
 * All give identical results (the label-correcting fixed point does not depend on the schedule). */
int mnav_set_dijkstra_engine(mnav_ctx* ctx, int engine);
/* Tuning / debug options by name (the list with one line each: mesh_navigation_amd/csrc/mnav_options.h; e.g. "cvp_wide",
 * "cvp_groups", "no_graph", "lazy_paths", "max_wall_s", "async_wg_per_plan").  mnav_create reads the process environment ONCE
 * (MNAV_<NAME>); afterwards this call is the only way to change an option -- no plan path ever looks at the environment.
 * value = NaN restores the built-in default.  Options read at upload time ("tile_size", "tb_tile") act on the next
 * mnav_upload_mesh.  Returns 0, or -1 for an unknown name; mnav_get_option returns NaN for unset / unknown. */
int mnav_set_option(mnav_ctx* ctx, const char* name, double value);
double mnav_get_option(const mnav_ctx* ctx, const char* name);
/* Outputs that stay on the device.  mnav_set_resident_outputs(ctx, 1): every plan also computes its vector map
 * (computeVectorMap, dijkstra :189-209 / cvp :204-239) and leaves it in HBM even when no host buffer is passed.
 * mnav_download_output copies one V-sized output of plan `slot` to the host on demand (what as below; 12 B/vertex for
 * the vector map, 4 B otherwise).  mnav_vector_at is MeshMap::directionAtPosition (mesh_map.cpp:625-650) on the
 * resident vector map: the controller's sample at the robot pose (returns 1, 0 = no vector there, <0 error). */
int mnav_set_resident_outputs(mnav_ctx* ctx, int on);
int mnav_download_output(mnav_ctx* ctx, uint32_t slot, int what, void* host_out);
int mnav_vector_at(mnav_ctx* ctx, uint32_t slot, const uint32_t vs[3], const float bary[3], float out[3]);
/* CVPMeshPlanner's back-tracking over the vector field (cvp_mesh_planner.cpp:920-951: MeshMap::meshAhead mesh_map.cpp
 * :1070-1108 with projectedBarycentricCoords util.cpp:320-347, searchNeighbourFaces mesh_map.cpp:999-1068,
 * directionAtPosition :625-650 and InflationLayer::vectorAt inflation_layer.cpp:493-521) on the vector maps the last
 * mnav_plan_cvp(_batch) call left resident (mnav_set_resident_outputs, or a vecmap_out buffer): plan i walks from
 * target_pos/target_face (the robot) until it is within step_width of seed_pos/seed_face (the goal).  inflation_layer: a
 * layer computed by mnav_layer_inflation whose repulsive field is added to every step, or -1.  Row i of positions_out
 * (cap*3 floats) / faces_out (cap) receives n_out[i] entries in the reference's list order (seed first); status_out[i] =
 * 1 reached the seed, 0 no path (the walk left the field or the mesh, or ran into `cap`), -1 a face of the walk has a
 * vertex without an inflation entry (lvr2 panics there), -2 internal list overflow.  The same float32 operations in the
 * same order as the host: positions are bit-identical.  Returns 0, or -1 with mnav_last_error set (then nothing was
 * walked).  The single-plan form returns the status; -1 with a non-empty mnav_last_error is an argument/device error. */
int mnav_backtrack_cvp_batch(mnav_ctx* ctx, uint32_t n, const float* seed_pos, const uint32_t* seed_faces, const float* target_pos,
                             const uint32_t* target_faces, double step_width, int32_t inflation_layer, uint32_t cap,
                             float* positions_out, uint32_t* faces_out, uint32_t* n_out, int32_t* status_out);
int mnav_backtrack_cvp(mnav_ctx* ctx, const float seed_pos[3], uint32_t seed_face, const float target_pos[3], uint32_t target_face,
                       double step_width, int32_t inflation_layer, uint32_t cap, float* positions_out, uint32_t* faces_out,
                       uint32_t* n_out);
/* Device pointers of the last plan's resident outputs (slot = plan index in a batch):
 * what = 0 dist, 1 pred, 2 direction, 3 cutface, 4 vecmap.  NULL if not available -- in particular dist / pred after a
 * paths-only Dijkstra call (no dist_out / pred_out / vector map asked for): such a call runs no finalize pass, values
 * beyond goal_dist would be engine-tentative.  mnav_download_output additionally takes what = 5, the POPPED potential of
 * a Dijkstra plan: the reference's value wherever it popped the vertex (dist <= goal_dist), +inf elsewhere. */
const void* mnav_device_output(const mnav_ctx* ctx, uint32_t slot, int what);
/* Algorithmic bytes of the last call per SURVEY.md §8(d): SSSP 24*V' + 24*E', CVP 32*V' + 68*F'
 * with V' = settled vertices and E'/F' their incident edges/faces scaled from the full mesh. */
uint64_t mnav_algorithmic_bytes(const mnav_ctx* ctx);
/* Which engine / kernel ran the last Dijkstra call (for reports and profiles): the engine of mnav_set_dijkstra_engine that `auto`
 * resolved to (0, 1, 5, 6), + 16 when the tile-batch engine (5) solved its tiles with the register-resident kernel k_tbv_solve
 * (64 plans per wave, distances in VGPRs) instead of k_tb_solve_q (16 plans per quarter of a wave, distances in LDS).  -1: none. */
int mnav_last_engine(const mnav_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* MNAV_H */
