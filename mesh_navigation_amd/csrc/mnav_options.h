// mnav_options.h -- the tuning / debug options of a context.  The process environment is read ONCE, by mnav_create
// (MNAV_<NAME IN CAPITALS>); afterwards only mnav_set_option changes an option: no getenv on any plan path, a variable that
// appears in the environment of a running mbf_mesh_nav node changes nothing.  Unset (NaN) = the built-in default of the place
// that reads it.  Included by mnav.hip; not a stand-alone header.
#pragma once

#define MNAV_OPTION_LIST(X)                                                                                                              \
  X(verbose)             /* 1: one line per engine build / asynchronous call on stderr */                                                 \
  X(trace)               /* 1: host-side trace points on stderr */                                                                        \
  X(no_graph)            /* 1: launch the step / round kernels one by one instead of replaying hipGraphs (profilers that cannot see into graphs) */ \
  X(debug_chunk)         /* steps per host look when no_graph is set */                                                                   \
  X(dijkstra_engine)     /* 0 tile rounds, 1 band steps, 3 auto, 5 tile-batch, 6 asynchronous tiles */                            \
  X(blocks_per_plan)     /* band steps: workgroups per plan */                                                                            \
  X(tile_blocks)         /* tile rounds: workgroups per plan */                                                                           \
  X(cvp_wide)            /* 0 / 1: never / always the wide CVP step kernel (default: batches from 32 plans) */                            \
  X(wide_waves)          /* wide CVP kernel: resident waves */                                                                            \
  X(cvp_groups)          /* wide CVP kernel: groups of plans stepped on their own streams (default by batch size, <= 4) */                \
  X(max_steps) X(max_wall_s)                                                                                                             \
  X(cvp_verify)          /* 0: skip the verification sweeps after a CVP plan */                                                           \
  X(key_walk_max) X(descend_walk_max)   /* cascade-tree walk bounds (tests) */                                                            \
  X(tile_size)           /* LDS tiles, read at mnav_upload_mesh */                                                                        \
  X(lazy_paths)          /* 0: always run the finalize pass */                                                                            \
  X(tile_band) X(rounds_band_mult)                                                                                                       \
  X(tb_tile)             /* tile-batch engine: rows per tile, read when its streams are built */                                          \
  X(tb_no_prefill) X(tb_band_mult) X(tb_waves_per_cu)                                                                                    \
  X(tb_kernel)           /* tile-batch solve: 0 = k_tb_solve_q (16 plans per quarter of a wave, distances in LDS), 1 = k_tbv_solve (64 plans per wave, distances in registers); default: by the expected plans per tile */ \
  X(async_wg_per_cu) X(async_wg_per_plan) X(async_max_s)                                                                                 \
  X(async_band_mult)     /* band of the asynchronous engine in tile widths of potential (default 4; <= 0: no bands) */                         \
  X(async_max_batch)     /* auto: batches of up to this many plans take the asynchronous engine */                                      \
  X(async_ring_cap)      /* ticket slots of the asynchronous engine (default 16 per tile and plan; tests force the overflow path) */

struct Options {
#define X(name) double name = NAN;
  MNAV_OPTION_LIST(X)
#undef X
  struct Desc { const char* name; double Options::*field; };
  static const Desc* table(size_t* n)
  {
    static const Desc t[] = {
#define X(name) { #name, &Options::name },
      MNAV_OPTION_LIST(X)
#undef X
    };
    *n = sizeof(t) / sizeof(t[0]);
    return t;
  }
  double* find(const char* name)
  {
    size_t n; const Desc* t = table(&n);
    for (size_t i = 0; i < n; ++i) if (!strcmp(t[i].name, name)) return &(this->*t[i].field);
    return nullptr;
  }
  // MNAV_<NAME> for every option; dijkstra_engine also takes the engines' names
  void from_environment()
  {
    size_t n; const Desc* t = table(&n);
    for (size_t i = 0; i < n; ++i) {
      std::string var = "MNAV_";
      for (const char* c = t[i].name; *c; ++c) var += (char)toupper((unsigned char)*c);
      const char* e = getenv(var.c_str());
      if (!e || !*e) continue;
      double v = atof(e);
      if (!strcmp(t[i].name, "dijkstra_engine")) {
        v = !strcmp(e, "tiled") ? 0 : !strcmp(e, "band") ? 1 : !strcmp(e, "tile_batch") ? 5 : !strcmp(e, "async") ? 6
          : (e[0] >= '0' && e[0] <= '9') ? atof(e) : 3;
      }
      this->*t[i].field = v;
    }
  }
};
static inline bool opt_set(double v) { return v == v; }
static inline uint32_t opt_u32(double v, uint32_t dflt) { return v == v ? (uint32_t)(v < 0.0 ? 0.0 : v) : dflt; }
static inline bool opt_on(double v) { return v == v && v != 0.0; }
