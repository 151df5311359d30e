// mnav_tbv.h -- the REGISTER-RESIDENT solve of the tile-batch SSSP engine (k_tbv_solve): one wave per (tile, <= 64 plans), the
// plans' distances of the tile's 120 vertices in VGPRs, no LDS anywhere.  Included by mnav.hip after mnav_tb.h.
//
// Why (profiles/r05_c2_sq.md): k_tb_solve_q keeps the distances in LDS as [vertex][lane], 30.7 KB per wave = 1.25
// waves per SIMD, and a lone wave is bound by its own instruction issue: 22 VALU + 9 LDS + 7 SALU issue slots per block, one
// after the other (293 cycles per block and wave measured, 88 of them VALU).  The register file of a CU is 512 KB against 160 KB
// of LDS: with the distances in registers a wave needs no LDS, two to three waves share a SIMD and the scalar half of one wave's
// block (unpacking row indices, moving the register index) issues next to the vector half of another's.
//
// How: the lanes of the wave are plans of ONE tile, so a row index is wave-uniform -- the hardware's VGPR index mode
// (s_set_gpr_idx_*: M0[7:0] is added to the register number of the operands selected by M0[15:12]) addresses row r of the image
// as v[kImg + r].  The image lives in a FIXED window of physical registers, v[136:255]; the compiler is held below it
// (amdgpu_num_vgpr) and never sees it; everything that touches the window is inline assembly:
//   v[0:95]     compiler
//   v[96:119]   sweep: ring of six stream chunks (4 registers each)          } only inside tbv_sweeps
//   v[120:127]  sweep: six candidates + their minima                          }
//   v128        sweep: this lane's byte offset into a chunk
//   v[136:255]  the image: row r of the tile, this lane's plan
// The relaxation itself is dijkstra_mesh_planner.cpp:331 (one float32 add per edge, minimum over the sources): same arithmetic,
// same fixed point as k_tb_solve_q and every other engine, bit for bit.
#pragma once

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"     // (the clobber list names v255 on purpose: the kernel's register count must cover the image window)
namespace {
namespace tbv {

// amdgpu_num_vgpr(n) holds the compiler to 2 n registers on gfx950 (unified VGPR / AGPR file; measured: n = 32, 48, 64, 96 ->
// NumVgprs 64, 96, 128, 192): v[0:95] here
#define TBV_COMPILER_VGPRS __attribute__((amdgpu_num_vgpr(48)))
constexpr int kImg = 136;                 // first image register
constexpr int kRows = 120;                // rows of the image = TbState::T of this kernel

#define TBV_S2(x) #x
#define TBV_S(x) TBV_S2(x)

// ---- rows by a compile-time index
template <int R> __device__ __forceinline__ void img_set(uint32_t x) { asm volatile("v_mov_b32 v[136+%c1], %0" : : "v"(x), "n"(R)); }
template <int R> __device__ __forceinline__ uint32_t img_get() { uint32_t x; asm volatile("v_mov_b32 %0, v[136+%c1]" : "=v"(x) : "n"(R)); return x; }
// row R <- *p, rows 4C..4C+3 <- the 16 bytes at p + 16 C: loads straight into the window, all in flight together; img_loads_wait() before the rows are used
template <int R> __device__ __forceinline__ void img_load(const uint32_t* p) { asm volatile("global_load_dword v[136+%c1], %0, off" : : "v"(p), "n"(R) : "memory"); }
template <int C> __device__ __forceinline__ void img_load_quad(const void* p) { asm volatile("global_load_dwordx4 v[136+%c1:139+%c1], %0, off offset:%c2" : : "v"(p), "n"(4 * C), "n"(16 * C) : "memory"); }
__device__ __forceinline__ void img_loads_wait() { asm volatile("s_waitcnt vmcnt(0)" : : : "memory"); }
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f)
{
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
// ---- rows by a wave-uniform index (an SGPR)
__device__ __forceinline__ uint32_t img_read(uint32_t row)
{
  uint32_t x;
  asm volatile("s_set_gpr_idx_on %1, gpr_idx(SRC0)\n\tv_mov_b32 %0, v136\n\ts_set_gpr_idx_off" : "=v"(x) : "s"(row));
  return x;
}
__device__ __forceinline__ void img_write(uint32_t row, uint32_t x)
{
  asm volatile("s_set_gpr_idx_on %1, gpr_idx(DST)\n\tv_mov_b32 v136, %0\n\ts_set_gpr_idx_off" : : "v"(x), "s"(row));
}
// img[row] = min(img[row], x); returns the lanes that were lowered
__device__ __forceinline__ unsigned long long img_min(uint32_t row, uint32_t x)
{
  unsigned long long m;
  asm volatile("s_set_gpr_idx_on %2, gpr_idx(SRC1,DST)\n\tv_cmp_lt_u32 vcc, %1, v136\n\tv_min_u32 v136, %1, v136\n\ts_set_gpr_idx_off\n\ts_mov_b64 %0, vcc"
               : "=s"(m) : "v"(x), "s"(row) : "vcc");
  return m;
}
// |img[row]| + w  (the sign bit of an image value marks "lowered in this activation" once the sweeps are over)
__device__ __forceinline__ float img_abs_plus(uint32_t row, float w)
{
  float t;
  asm volatile("s_set_gpr_idx_on %1, gpr_idx(SRC0)\n\tv_add_f32_e64 %0, |v136|, %2\n\ts_set_gpr_idx_off" : "=v"(t) : "s"(row), "s"(w));
  return t;
}

// ---- the Gauss-Seidel sweeps of one activation.
// Stream: the tile's sweep chunks in the V layout (mnav_tb_build.h, tb_vsweep): a chunk = 4 blocks of 16 dwords, stored transposed
// like the Q layout (dword q of block j at chunk dword 4 q + j).  A block relaxes its target row from up to SIX source rows
// (dijkstra :331); its row indices come as ready-made values of M0 (index in bits 7:0, the operands the index applies to in bits
// 15:12 -- 0x2000: source 1, 0xA000: source 1 and destination):
//     d0 = (0xA000 | target) | (0x2000 | source0) << 16      d1 = (0x2000 | source1) | (0x2000 | source2) << 16
//     d2 = (0x2000 | source3) | (0x2000 | source4) << 16     d3 =  0x2000 | source5
//     d8 .. d13 = the six weights (an unused slot: source = target, weight +inf)
// The index dwords of a chunk are its first 64 bytes: ONE s_load_dwordx16 per chunk brings them into SGPRs (the chunk after the
// current one is in flight); the weights arrive through the vector path -- lane l of every 16-lane row loads the 16 bytes l & 15
// of the chunk and holds dword l & 15 of block j in register j -- and are consumed as DPP operands (row_newbcast: lane 8 + k of
// the row).  Per block 11 vector and 9 scalar instructions:
//     6 x (M0 <- index; v_add_f32_dpp t_k, w_k, img[source_k]), M0 <- 0, 2 v_min3_u32 + v_min_u32, M0 <- target,
//     v_cmp_lt_u32 (did the target improve: OR-ed into the sweep's flag), v_min_u32 img[target]
// (the first version of the routine -- indices through v_readlane, s_set_gpr_idx_idx per access, mode switches around the write:
//  16 + 16 instructions -- measured 129 cycles per block and SIMD with two waves on the SIMD, 205 with one: the SIMD issues one
//  instruction per four cycles whatever its kind, so the count is what matters.)
// Reads and writes of the image are register accesses: block j + 1 sees what block j wrote -- the plain Gauss-Seidel sweep with
// no forwarding rule.  Six chunks are in registers or in flight (the load cursor runs on into the next sweep's order).
// Returns the number of sweeps (the last one changed nothing in any lane); `overrun` when `cap` sweeps did not suffice.
#define TBV_ADD(t, D, k) "v_add_f32_dpp v" #t ", v[" #D "], v136 row_newbcast:" #k " row_mask:0xf bank_mask:0xf\n\t"
#define TBV_BLOCK(D, i0, i1, i2, i3)                                                                                    \
  "s_lshr_b32 m0, s[" #i0 "], 16\n\t"      TBV_ADD(120, D, 8)                                                          \
  "s_and_b32 m0, s[" #i1 "], 0xffff\n\t"   TBV_ADD(121, D, 9)                                                          \
  "s_lshr_b32 m0, s[" #i1 "], 16\n\t"      TBV_ADD(122, D, 10)                                                         \
  "s_and_b32 m0, s[" #i2 "], 0xffff\n\t"   TBV_ADD(123, D, 11)                                                         \
  "s_lshr_b32 m0, s[" #i2 "], 16\n\t"      TBV_ADD(124, D, 12)                                                         \
  "s_mov_b32 m0, s[" #i3 "]\n\t"           TBV_ADD(125, D, 13)                                                         \
  "s_mov_b32 m0, 0\n\t"                                                                                                \
  "v_min3_u32 v126, v120, v121, v122\n\t"                                                                              \
  "v_min3_u32 v127, v123, v124, v125\n\t"                                                                              \
  "v_min_u32 v127, v126, v127\n\t"                                                                                     \
  "s_and_b32 m0, s[" #i0 "], 0xffff\n\t"                                                                               \
  "v_cmp_lt_u32 vcc, v127, v136\n\t"                                                                                   \
  "v_min_u32 v136, v127, v136\n\t"                                                                                     \
  "s_or_b64 s[70:71], s[70:71], vcc\n\t"
// the chunk under the vector load cursor (byte offset s68 from the stream's first sweep chunk) into ring slot `b`; cursor + 1 chunk, cyclic
#define TBV_LOAD(b)                                                                                                    \
  "s_add_u32 s86, s72, s68\n\t"  "s_addc_u32 s87, s73, 0\n\t"                                                          \
  "global_load_dwordx4 v[" #b ":" #b "+3], v128, s[86:87]\n\t"                                                         \
  "s_add_u32 s68, s68, 0x100\n\t"  "s_cmp_eq_u32 s68, s75\n\t"  "s_cselect_b32 s68, 0, s68\n\t"
// the index dwords of the chunk under the scalar cursor (s78) into the SGPR set that starts at `S`; cursor + 1 chunk, cyclic
#define TBV_SLOAD(S)                                                                                                   \
  "s_load_dwordx16 s[" #S ":" #S "+15], s[72:73], s78\n\t"                                                             \
  "s_add_u32 s78, s78, 0x100\n\t"  "s_cmp_eq_u32 s78, s75\n\t"  "s_cselect_b32 s78, 0, s78\n\t"
// one chunk: its weights in ring slot b, its indices in set SA; the next chunk's indices go to set SB
#define TBV_CHUNK(b, SA, SB, tag)                                                                                      \
  "s_waitcnt vmcnt(5) lgkmcnt(0)\n\t"                                                                                  \
  TBV_SLOAD(SB)                                                                                                        \
  "s_set_gpr_idx_on s74, gpr_idx(SRC1)\n\t"                                                                            \
  TBV_BLOCK(b, SA, SA + 4, SA + 8, SA + 12)  TBV_BLOCK(b + 1, SA + 1, SA + 5, SA + 9, SA + 13)                         \
  TBV_BLOCK(b + 2, SA + 2, SA + 6, SA + 10, SA + 14)  TBV_BLOCK(b + 3, SA + 3, SA + 7, SA + 11, SA + 15)               \
  "s_set_gpr_idx_off\n\t"                                                                                              \
  TBV_LOAD(b)                                                                                                          \
  "s_sub_u32 s80, s80, 1\n\t"                                                                                          \
  "s_cmp_lg_u32 s80, 0\n\t"                                                                                            \
  "s_cbranch_scc1 Ltbv_nx" #tag "_%=\n\t"                                                                              \
  "s_mov_b32 s80, s74\n\t"                                                                                             \
  "s_add_u32 s84, s84, 1\n\t"                                                                                          \
  "s_or_b64 s[70:71], s[70:71], s[82:83]\n\t"                                                                          \
  "s_cmp_eq_u64 s[70:71], 0\n\t"                                                                                       \
  "s_cbranch_scc1 Ltbv_done_%=\n\t"                                                                                    \
  "s_cmp_ge_u32 s84, s81\n\t"                                                                                          \
  "s_cbranch_scc1 Ltbv_over_%=\n\t"                                                                                    \
  "s_mov_b64 s[70:71], 0\n\t"                                                                                          \
  "Ltbv_nx" #tag "_%=:\n\t"

// `force` != 0: every sweep counts as "changed" (timing runs: exactly `cap` sweeps).  nch >= 1.
__device__ __forceinline__ uint32_t tbv_sweeps(const uint32_t* sweep0, uint32_t nch, uint32_t first_order, uint32_t cap, uint32_t lane_off,
                                               uint32_t force, bool& overrun)
{
  uint32_t sweeps, ovr;
  // (every scalar operand is made uniform explicitly: a 64-bit "s" operand the compiler thinks divergent is handed over in VGPRs)
  const unsigned long long st = (unsigned long long)(uintptr_t)sweep0;
  const uint32_t st_lo = tb::rfl((uint32_t)st), st_hi = tb::rfl((uint32_t)(st >> 32));
  nch = tb::rfl(nch); first_order = tb::rfl(first_order); cap = tb::rfl(cap); force = tb::rfl(force);
  asm volatile(
      "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
      "v_mov_b32 v128, %[off]\n\t"
      "s_mov_b32 s72, %[stlo]\n\t"
      "s_mov_b32 s73, %[sthi]\n\t"
      "s_mov_b32 s74, %[nch]\n\t"
      "s_lshl_b32 s75, s74, 10\n\t"                                   // bytes of the four orders
      "s_mov_b32 s81, %[cap]\n\t"
      "s_mov_b32 s82, %[force]\n\t"
      "s_mov_b32 s83, %[force]\n\t"
      "s_lshl_b32 s68, s74, 8\n\t"
      "s_mul_i32 s68, s68, %[ord]\n\t"                               // both cursors start at the first order's chunk 0
      "s_mov_b32 s78, s68\n\t"
      "s_mov_b32 s80, s74\n\t"
      "s_mov_b32 s84, 0\n\t"
      "s_mov_b64 s[70:71], 0\n\t"
      TBV_LOAD(96) TBV_LOAD(100) TBV_LOAD(104) TBV_LOAD(108) TBV_LOAD(112) TBV_LOAD(116)
      TBV_SLOAD(36)
      "Ltbv_top_%=:\n\t"
      TBV_CHUNK(96, 36, 52, c0) TBV_CHUNK(100, 52, 36, c1) TBV_CHUNK(104, 36, 52, c2) TBV_CHUNK(108, 52, 36, c3) TBV_CHUNK(112, 36, 52, c4) TBV_CHUNK(116, 52, 36, c5)
      "s_branch Ltbv_top_%=\n\t"
      "Ltbv_over_%=:\n\t"
      "s_mov_b32 %[ovr], 1\n\t"
      "s_branch Ltbv_end_%=\n\t"
      "Ltbv_done_%=:\n\t"
      "s_mov_b32 %[ovr], 0\n\t"
      "Ltbv_end_%=:\n\t"
      "s_mov_b32 %[sw], s84\n\t"
      "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"                              // the speculative loads of the next sweep land in the ring / the index set: nothing may follow them
      : [sw] "=s"(sweeps), [ovr] "=s"(ovr)
      : [stlo] "s"(st_lo), [sthi] "s"(st_hi), [nch] "s"(nch), [ord] "s"(first_order), [cap] "s"(cap), [off] "v"(lane_off), [force] "s"(force)
      : "memory", "vcc", "scc", "m0", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53",
        "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s70", "s71", "s72", "s73", "s74", "s75", "s78",
        "s80", "s81", "s82", "s83", "s84", "s86", "s87", "v96", "v127", "v128", "v255");
  overrun = ovr != 0u;
  return sweeps;
}

}  // namespace tbv

// ---------------------------------------------------------------------------------------------
// The register-resident solve: one wave per work item = (tile, <= 64 plans of its bucket), one plan per lane.
// ---------------------------------------------------------------------------------------------
// Same item protocol, same slices, same wake-ups and exports as k_tb_solve_q (mnav_tb.h) -- the two kernels are interchangeable
// per item --, but everything that was uniform per QUARTER there is uniform per WAVE here: the tile header, the ghost streams and
// the export records are read through the scalar cache (constant address space, s_load), the sweeps through tbv_sweeps.  During
// the ghost phase and the sweeps the image holds plain values; the "lowered in this activation" marks (sign bits) the write-back,
// the wake-ups and the exports go by are set afterwards, by comparing the image with the slice as it was loaded.
namespace tbv {
typedef const MNAV_CONST uint32_t* cwords_t;
template <class P> __device__ __forceinline__ cwords_t cwords(P* p) { return (cwords_t)(uintptr_t)p; }
// rows 4C..4C+3 against the values they were loaded with: sign bit where lowered; returns the marked rows
template <int R> __device__ __forceinline__ uint32_t mark_row(uint32_t orig, uint32_t sign)
{
  uint32_t x;
  asm volatile("v_sub_u32 %0, v[136+%c3], %1\n\tv_and_or_b32 %0, %0, %2, v[136+%c3]\n\tv_mov_b32 v[136+%c3], %0" : "=&v"(x) : "v"(orig), "s"(sign), "n"(R));
  return x;
}
// img[row] = min(img[row], x); 1 in the lanes that were lowered
__device__ __forceinline__ uint32_t img_min_flag(uint32_t row, uint32_t x)
{
  uint32_t f;
  asm volatile("s_set_gpr_idx_on %2, gpr_idx(SRC1,DST)\n\tv_cmp_lt_u32 vcc, %1, v136\n\tv_min_u32 v136, %1, v136\n\ts_set_gpr_idx_off\n\tv_cndmask_b32 %0, 0, 1, vcc"
               : "=v"(f) : "v"(x), "s"(row) : "vcc");
  return f;
}
}  // namespace tbv

template <int T>
__global__ __launch_bounds__(64) TBV_COMPILER_VGPRS
void k_tbv_solve(tb::Args A, const uint32_t* __restrict__ vtile, const uint32_t* __restrict__ vstream, int par)
{
  static_assert(T == tbv::kRows && T % 20 == 0, "the image window holds 120 rows");
  const int lane = threadIdx.x;
  const uint32_t NP = A.NP;
  const uint32_t n_items = tb::rfl(A.ctl->n_items);
  const tbv::cwords_t cstream = tbv::cwords(A.stream), ctiles = tbv::cwords(A.tiles), cexps = tbv::cwords(A.exps), cvtile = tbv::cwords(vtile);
  uint32_t my_items = 0, my_acts = 0, my_sweeps = 0, my_wakes = 0;
#ifdef MNAV_TB_TIMING
  unsigned long long tt[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, t_last = __builtin_readcyclecounter();
#endif
  for (;;) {
    uint32_t it = 0;
    if (lane == 0) it = atomicAdd(&A.ctl->next_item, 1u);
    it = tb::rfl(it);
    if (it >= n_items) break;
    const u32x2 item = ((MNAV_GLOBAL const u32x2*)as_global(A.items))[it];
    const uint32_t t = tb::rfl(item.x), start = tb::rfl(item.y & 0xFFFFu), count = tb::rfl(item.y >> 16);
    const tbv::cwords_t hw = ctiles + (size_t)t * 16u;
    const uint32_t soff = hw[tb::kTwSoff], slen = hw[tb::kTwSl];
    const uint32_t pre_off = hw[tb::kTwPreOff], pre_chunks = hw[tb::kTwPreChunks], post_off = hw[tb::kTwPostOff], post_chunks = hw[tb::kTwPostChunks];
    const uint32_t exp_off = hw[tb::kTwExpOff], exp_n = hw[tb::kTwExpN];
    const uint32_t voff = cvtile[2u * t], vch = cvtile[2u * t + 1u];
    ++my_items; my_acts += (lane == 0) ? count : 0u;
    // lanes beyond `count` shadow the last plan of the item and store nothing
    const bool active = (uint32_t)lane < count;
    const uint32_t p = A.bucket[(size_t)t * NP + start + min((uint32_t)lane, max(count, 1u) - 1u)];
    MNAV_GLOBAL float* const sl = as_global(A.D) + ((size_t)soff * NP + (size_t)p * slen);
    TB_STAMP(0);
    // ---- the owned slots: straight into the image window, all loads in flight together
    tbv::static_for<0, T / 4>([&](auto c) { tbv::img_load_quad<decltype(c)::value>((const void*)sl); });
    tbv::img_loads_wait();
    MNAV_GLOBAL const u32x4* const g4p = (MNAV_GLOBAL const u32x4*)(sl + T);
    uint32_t first_order = 0;
    TB_STAMP(1);
    // ---- ghosts -> owned (the ghosts are constant during the activation)
    if (pre_chunks) {
      tbv::cwords_t S = cstream + (size_t)pre_off * kTbChunk;
      float gmin = inf_f();                                           // smallest ghost value that lowered one of this lane's vertices ...
      uint32_t gord = 0;                                              // ... and the sweep order that runs with a wave entering there
      u32x4 G = g4p[S[12]];
      for (uint32_t c = 0; c < pre_chunks; ++c, S += kTbChunk) {
        const u32x4 Gn = g4p[(c + 1u < pre_chunks) ? S[13] : 0u];     // the next chunk's ghost values
#pragma unroll
        for (int j = 0; j < (int)kTbBlocksPerChunk; ++j) {
          const uint32_t hd = S[16 * j], n = (hd >> 8) & 7u;
          if (n) {
            const uint32_t jj = hd & 3u;
            const float g = u2f(jj == 0 ? G.x : jj == 1 ? G.y : jj == 2 ? G.z : G.w);
            uint32_t lowered = 0;
#pragma unroll
            for (int k = 0; k < (int)kTbGhostEdges; ++k)
              if ((uint32_t)k < n) lowered |= tbv::img_min_flag(S[16 * j + 1 + k] >> 8, f2u(g + u2f(S[16 * j + 6 + k])));
            if (lowered && g < gmin) { gmin = g; gord = (hd >> kTbOrderShift) & 3u; }
          }
        }
        G = Gn;
      }
      const bool votes = active && gmin < inf_f();
      uint32_t bestc = 0;
#pragma unroll
      for (uint32_t o = 0; o < 4; ++o) {
        const uint32_t cn = (uint32_t)__popcll(__ballot(votes && gord == o));
        if (cn > bestc) { bestc = cn; first_order = o; }
      }
    }
    TB_STAMP(2);
    // ---- Gauss-Seidel sweeps to the tile-local fixed point of every lane
    uint32_t sweep = 1;
    if (vch) {
      bool overrun;
      sweep = tbv::tbv_sweeps(vstream + (size_t)voff * kTbChunk, vch, first_order, 16u * T, 16u * ((uint32_t)lane & 15u), 0u, overrun);
      if (overrun && lane == 0) A.ctl->err = 1u;
    }
    my_sweeps += sweep;
    TB_STAMP(3);
    // ---- mark what was lowered (image against the slice as loaded), write back the 16-byte chunks that hold a lowered value
    {
      MNAV_GLOBAL u32x4* const s4 = (MNAV_GLOBAL u32x4*)sl;
      const uint32_t sign = kTbDirty;
      tbv::static_for<0, T / 20>([&](auto b) {
        constexpr int B = decltype(b)::value;
        u32x4 o[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) o[q] = s4[5 * B + q];
        tbv::static_for<0, 5>([&](auto qq) {
          constexpr int Q = decltype(qq)::value, C = 5 * B + Q;
          u32x4 x;
          x.x = tbv::mark_row<4 * C + 0>(o[Q].x, sign); x.y = tbv::mark_row<4 * C + 1>(o[Q].y, sign);
          x.z = tbv::mark_row<4 * C + 2>(o[Q].z, sign); x.w = tbv::mark_row<4 * C + 3>(o[Q].w, sign);
          if (active && ((x.x | x.y | x.z | x.w) & kTbDirty)) {
            x.x &= 0x7fffffffu; x.y &= 0x7fffffffu; x.z &= 0x7fffffffu; x.w &= 0x7fffffffu;
            s4[C] = x;
          }
        });
      });
    }
    TB_STAMP(4);
    // ---- owned -> ghosts: a neighbour tile is woken when a candidate undercuts what we know of its vertex (the three-stage
    // pipeline of k_tb_solve_q: look at the pending value, atomicMin it, learn from the old value whether this is the pair's first wake-up)
    if (post_chunks) {
      tbv::cwords_t S = cstream + (size_t)post_off * kTbChunk;
      u32x4 G = g4p[S[12]];
      uint32_t cand = kTbInfBits, best = kTbInfBits;
      MNAV_GLOBAL uint32_t* const pend_p = as_global(A.pend) + p;
      MNAV_GLOBAL uint8_t* const pflag_p = as_global(A.pflag) + (p >> 6);
      MNAV_GLOBAL uint32_t* const pm = as_global(A.marr[par ^ 1]) + p;
      uint32_t t2_1 = 0, best_1 = kTbInfBits, cur_1 = 0, best_2 = kTbInfBits, old_2 = 0;
      bool want_1 = false, did_2 = false;
      uint32_t n_first = 0;
      auto advance = [&](uint32_t t2_new, uint32_t best_new, bool want_new) {
        bool first = false;
        if (did_2) {
          first = old_2 == kTbInfBits;
          if (best_2 < old_2) atomicMin((uint32_t*)pm, best_2);
          ++my_wakes;
        }
        n_first += first ? 1u : 0u;
        did_2 = want_1 && best_1 < cur_1;
        best_2 = best_1;
        if (did_2) { old_2 = atomicMin((uint32_t*)(pend_p + (size_t)t2_1 * NP), best_1); pflag_p[(size_t)t2_1 * A.nblk] = 1; }
        want_1 = want_new; t2_1 = t2_new; best_1 = best_new;
        if (want_new) cur_1 = pend_p[(size_t)t2_new * NP];
      };
      for (uint32_t c = 0; c < post_chunks; ++c, S += kTbChunk) {
        const u32x4 Gn = g4p[(c + 1u < post_chunks) ? S[13] : 0u];
#pragma unroll
        for (int j = 0; j < (int)kTbBlocksPerChunk; ++j) {
          const uint32_t hd = S[16 * j], n = (hd >> 8) & 7u;
          if (n) {
#pragma unroll
            for (int k = 0; k < (int)kTbGhostEdges; ++k)
              if ((uint32_t)k < n) cand = min(cand, f2u(tbv::img_abs_plus(S[16 * j + 1 + k] >> 8, u2f(S[16 * j + 6 + k]))));
            if (hd & kTbGhostEnd) {
              const uint32_t jj = hd & 3u;
              const uint32_t g = jj == 0 ? G.x : jj == 1 ? G.y : jj == 2 ? G.z : G.w;
              if (cand < g) best = min(best, cand);
              cand = kTbInfBits;
            }
            if (hd & kTbTileEnd) {
              advance(S[16 * j + 11], best, active && best != kTbInfBits);   // d11: owner tile of the ghosts just closed
              best = kTbInfBits;
            }
          }
        }
        G = Gn;
      }
      advance(0u, kTbInfBits, false);                                  // drain the two stages in flight
      advance(0u, kTbInfBits, false);
      n_first = wave_sum(n_first);
      if (lane == 0 && n_first) atomicAdd(&A.ctl->n_cand[par ^ 1], n_first);
    }
    TB_STAMP(5);
    // ---- export the lowered boundary values to the ghost slots that mirror them
    for (uint32_t k = 0; k < exp_n; ++k) {
      const tbv::cwords_t X = cexps + 4u * ((size_t)exp_off + k);
      const uint32_t v = tbv::img_read(X[0] >> 8);
      if (active && (v & kTbDirty)) as_global(A.D)[(size_t)X[1] * NP + ((size_t)p * X[2] + X[3])] = u2f(v & 0x7fffffffu);
    }
    TB_STAMP(6);
  }
#ifdef MNAV_TB_TIMING
  if (lane == 0) for (int k = 0; k < 8; ++k) if (tt[k]) atomicAdd(&g_tb_timing[k], tt[k]);
#endif
  my_wakes = wave_sum(my_wakes); my_acts = wave_sum(my_acts);
  if (lane == 0 && my_items) {
    atomicAdd(&A.ctl->items, (unsigned long long)my_items); atomicAdd(&A.ctl->acts, (unsigned long long)my_acts);
    atomicAdd(&A.ctl->sweeps, (unsigned long long)my_sweeps); atomicAdd(&A.ctl->wakes, (unsigned long long)my_wakes);
  }
}

// Test / timing entry of the sweep routine on its own (mnav_debug_tbv_sweeps): every wave loads ITS image [row][lane] from `img`,
// runs the sweeps over the one stream and stores the image back.
__global__ __launch_bounds__(64) TBV_COMPILER_VGPRS
void k_tbv_micro(const uint32_t* __restrict__ stream, uint32_t nch, uint32_t first_order, uint32_t cap, uint32_t force, uint32_t reps,
                 uint32_t* __restrict__ img, uint32_t* __restrict__ out)
{
  const int lane = threadIdx.x;
  uint32_t* const my = img + (size_t)blockIdx.x * tbv::kRows * 64 + lane;
  tbv::static_for<0, tbv::kRows>([&](auto r) { constexpr int R = decltype(r)::value; tbv::img_load<R>(my + R * 64); });
  tbv::img_loads_wait();
  uint32_t total = 0, over = 0;
  for (uint32_t k = 0; k < reps; ++k) {
    bool ovr;
    total += tbv::tbv_sweeps(stream, nch, first_order, cap, 16u * ((uint32_t)lane & 15u), force ? ~0u : 0u, ovr);
    over |= ovr ? 1u : 0u;
  }
  tbv::static_for<0, tbv::kRows>([&](auto r) { constexpr int R = decltype(r)::value; my[R * 64] = tbv::img_get<R>(); });
  if (lane == 0) { out[2 * blockIdx.x] = total; out[2 * blockIdx.x + 1] = over; }
}

}  // namespace

extern "C" int mnav_debug_tbv_sweeps(const uint32_t* stream_host, uint32_t nch, uint32_t first_order, uint32_t cap, uint32_t force, uint32_t reps,
                                     uint32_t waves, uint32_t* img_host, uint32_t* out_host, float* ms_out)
{
  uint32_t *d_s = nullptr, *d_i = nullptr, *d_o = nullptr;
  const size_t ns = (size_t)4 * nch * 64 + 8 * 64, ni = (size_t)waves * tbv::kRows * 64;
  if (hipMalloc((void**)&d_s, 4 * ns) != hipSuccess || hipMalloc((void**)&d_i, 4 * ni) != hipSuccess || hipMalloc((void**)&d_o, 8 * (size_t)waves) != hipSuccess) return -1;
  (void)hipMemset(d_s, 0, 4 * ns);
  (void)hipMemcpy(d_s, stream_host, (size_t)4 * 4 * nch * 64, hipMemcpyHostToDevice);
  (void)hipMemcpy(d_i, img_host, 4 * ni, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k_tbv_micro, dim3(waves), dim3(64), 0, 0, d_s, nch, first_order, cap, force, reps, d_i, d_o);
  (void)hipEventRecord(e1, 0);
  const hipError_t e = hipDeviceSynchronize();
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  if (ms_out) *ms_out = ms;
  (void)hipMemcpy(img_host, d_i, 4 * ni, hipMemcpyDeviceToHost);
  (void)hipMemcpy(out_host, d_o, 8 * (size_t)waves, hipMemcpyDeviceToHost);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(d_s); (void)hipFree(d_i); (void)hipFree(d_o);
  return e == hipSuccess ? 0 : -2;
}
#pragma clang diagnostic pop
