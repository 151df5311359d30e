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
//   v[120:127]  sweep: seven candidates + their minimum                       }
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
// Stream: the tile's sweep chunks in the V layout (mnav_tb_build.h, tb_vsweep): a chunk = 4 blocks, stored transposed like the
// Q layout (dword q of block j at chunk dword 4 q + j), so that lane l of every 16-lane row loads the 16 bytes l & 15 and holds
// dword l & 15 of block j in register j of the load.  A block: d0 = target | source0 << 16, d1 = source1 | source2 << 16,
// d2 = source3 | source4 << 16, d3 = source5 | source6 << 16 (row indices), d8..d14 = the seven weights (unused slot: source =
// target, weight +inf).  The four index dwords go to SGPRs (v_readlane), the weights are consumed as DPP operands
// (row_newbcast: lane 8 + k of the row).  Per block 16 vector and 16 scalar instructions:
//     4 v_readlane, 7 x (s_set_gpr_idx_idx [+ s_lshr]; v_add_f32_dpp t_k, w_k, img[source_k]), 3 v_min3_u32,
//     v_cmp_lt_u32 (did the target improve: OR-ed into the sweep's flag), v_min_u32 img[target]
// Reads and writes of the image are register accesses: block j + 1 sees what block j wrote -- the plain Gauss-Seidel sweep with
// no forwarding rule.  Six chunks are in registers or in flight (the load cursor runs on into the next sweep's order).
// Returns the number of sweeps (the last one changed nothing in any lane); `overrun` when `cap` sweeps did not suffice.
#define TBV_ADD(t, D, k) "v_add_f32_dpp v" #t ", v[" #D "], v136 row_newbcast:" #k " row_mask:0xf bank_mask:0xf\n\t"
#define TBV_BLOCK(D)                                                                                                   \
  "v_readlane_b32 s76, v[" #D "], 0\n\t"                                                                                 \
  "v_readlane_b32 s77, v[" #D "], 1\n\t"                                                                                 \
  "v_readlane_b32 s78, v[" #D "], 2\n\t"                                                                                 \
  "v_readlane_b32 s79, v[" #D "], 3\n\t"                                                                                 \
  "s_lshr_b32 s80, s76, 16\n\t"  "s_set_gpr_idx_idx s80\n\t" TBV_ADD(120, D, 8)                                        \
  "s_set_gpr_idx_idx s77\n\t" TBV_ADD(121, D, 9)                                                                       \
  "s_lshr_b32 s80, s77, 16\n\t"  "s_set_gpr_idx_idx s80\n\t" TBV_ADD(122, D, 10)                                       \
  "s_set_gpr_idx_idx s78\n\t" TBV_ADD(123, D, 11)                                                                      \
  "s_lshr_b32 s80, s78, 16\n\t"  "s_set_gpr_idx_idx s80\n\t" TBV_ADD(124, D, 12)                                       \
  "s_set_gpr_idx_idx s79\n\t" TBV_ADD(125, D, 13)                                                                      \
  "s_lshr_b32 s80, s79, 16\n\t"  "s_set_gpr_idx_idx s80\n\t" TBV_ADD(126, D, 14)                                       \
  "s_set_gpr_idx_idx 0\n\t"                                                                                            \
  "v_min3_u32 v127, v120, v121, v122\n\t"                                                                              \
  "v_min3_u32 v127, v127, v123, v124\n\t"                                                                              \
  "v_min3_u32 v127, v127, v125, v126\n\t"                                                                              \
  "s_set_gpr_idx_idx s76\n\t"                                                                                          \
  "v_cmp_lt_u32 vcc, v127, v136\n\t"                                                                                   \
  "s_or_b64 s[70:71], s[70:71], vcc\n\t"                                                                               \
  "s_set_gpr_idx_mode gpr_idx(SRC1,DST)\n\t"                                                                           \
  "v_min_u32 v136, v127, v136\n\t"                                                                                     \
  "s_set_gpr_idx_mode gpr_idx(SRC1)\n\t"
// load the chunk under the load cursor into ring slot `b` (first register) and advance the cursor (chunk, order)
#define TBV_LOAD(b, tag)                                                                                               \
  "global_load_dwordx4 v[" #b ":" #b "+3], v128, s[64:65]\n\t"                                                         \
  "s_add_u32 s64, s64, 0x100\n\t"  "s_addc_u32 s65, s65, 0\n\t"                                                        \
  "s_add_u32 s66, s66, 1\n\t"                                                                                          \
  "s_cmp_lt_u32 s66, s74\n\t"                                                                                          \
  "s_cbranch_scc1 Ltbv_ld" #tag "_%=\n\t"                                                                              \
  "s_mov_b32 s66, 0\n\t"                                                                                               \
  "s_add_u32 s67, s67, 1\n\t"  "s_and_b32 s67, s67, 3\n\t"                                                             \
  "s_mul_i32 s80, s67, s75\n\t"                                                                                        \
  "s_add_u32 s64, s72, s80\n\t"  "s_addc_u32 s65, s73, 0\n\t"                                                          \
  "Ltbv_ld" #tag "_%=:\n\t"
#define TBV_CHUNK(b, tag)                                                                                              \
  "s_waitcnt vmcnt(5)\n\t"                                                                                             \
  "s_set_gpr_idx_on s76, gpr_idx(SRC1)\n\t"                                                                            \
  TBV_BLOCK(b) TBV_BLOCK(b + 1) TBV_BLOCK(b + 2) TBV_BLOCK(b + 3)                                                      \
  "s_set_gpr_idx_off\n\t"                                                                                              \
  TBV_LOAD(b, tag)                                                                                                     \
  "s_add_u32 s68, s68, 1\n\t"                                                                                          \
  "s_cmp_lt_u32 s68, s74\n\t"                                                                                          \
  "s_cbranch_scc1 Ltbv_nx" #tag "_%=\n\t"                                                                              \
  "s_mov_b32 s68, 0\n\t"                                                                                               \
  "s_add_u32 s69, s69, 1\n\t"                                                                                          \
  "s_or_b64 s[70:71], s[70:71], s[82:83]\n\t"                                                                          \
  "s_cmp_eq_u64 s[70:71], 0\n\t"                                                                                       \
  "s_cbranch_scc1 Ltbv_done_%=\n\t"                                                                                    \
  "s_cmp_ge_u32 s69, s81\n\t"                                                                                          \
  "s_cbranch_scc1 Ltbv_over_%=\n\t"                                                                                    \
  "s_mov_b64 s[70:71], 0\n\t"                                                                                          \
  "Ltbv_nx" #tag "_%=:\n\t"

// `force` != 0: every sweep counts as "changed" (timing runs: exactly `cap` sweeps)
__device__ __forceinline__ uint32_t tbv_sweeps(const uint32_t* sweep0, uint32_t nch, uint32_t first_order, uint32_t cap, uint32_t lane_off,
                                               uint32_t force, bool& overrun)
{
  uint32_t sweeps, ovr;
  // (every scalar operand is made uniform explicitly: a 64-bit "s" operand the compiler thinks divergent is handed over in VGPRs)
  const unsigned long long st = (unsigned long long)(uintptr_t)sweep0;
  const uint32_t st_lo = tb::rfl((uint32_t)st), st_hi = tb::rfl((uint32_t)(st >> 32));
  nch = tb::rfl(nch); first_order = tb::rfl(first_order); cap = tb::rfl(cap); force = tb::rfl(force);
  asm volatile(
      "s_waitcnt vmcnt(0)\n\t"
      "v_mov_b32 v128, %[off]\n\t"
      "s_mov_b32 s72, %[stlo]\n\t"
      "s_mov_b32 s73, %[sthi]\n\t"
      "s_mov_b32 s74, %[nch]\n\t"
      "s_lshl_b32 s75, s74, 8\n\t"
      "s_mov_b32 s81, %[cap]\n\t"
      "s_mov_b32 s82, %[force]\n\t"
      "s_mov_b32 s83, %[force]\n\t"
      "s_mov_b32 s67, %[ord]\n\t"
      "s_mul_i32 s80, s67, s75\n\t"
      "s_add_u32 s64, s72, s80\n\t"
      "s_addc_u32 s65, s73, 0\n\t"
      "s_mov_b32 s66, 0\n\t"
      "s_mov_b32 s68, 0\n\t"
      "s_mov_b32 s69, 0\n\t"
      "s_mov_b64 s[70:71], 0\n\t"
      TBV_LOAD(96, p0) TBV_LOAD(100, p1) TBV_LOAD(104, p2) TBV_LOAD(108, p3) TBV_LOAD(112, p4) TBV_LOAD(116, p5)
      "Ltbv_top_%=:\n\t"
      TBV_CHUNK(96, c0) TBV_CHUNK(100, c1) TBV_CHUNK(104, c2) TBV_CHUNK(108, c3) TBV_CHUNK(112, c4) TBV_CHUNK(116, c5)
      "s_branch Ltbv_top_%=\n\t"
      "Ltbv_over_%=:\n\t"
      "s_mov_b32 %[ovr], 1\n\t"
      "s_branch Ltbv_end_%=\n\t"
      "Ltbv_done_%=:\n\t"
      "s_mov_b32 %[ovr], 0\n\t"
      "Ltbv_end_%=:\n\t"
      "s_mov_b32 %[sw], s69\n\t"
      "s_waitcnt vmcnt(0)\n\t"                                         // the speculative loads of the next sweep land in the ring: nothing may follow them
      : [sw] "=s"(sweeps), [ovr] "=s"(ovr)
      : [stlo] "s"(st_lo), [sthi] "s"(st_hi), [nch] "s"(nch), [ord] "s"(first_order), [cap] "s"(cap), [off] "v"(lane_off), [force] "s"(force)
      : "memory", "vcc", "scc", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80",
        "s81", "s82", "s83", "v96", "v127", "v128", "v255");
  overrun = ovr != 0u;
  return sweeps;
}

}  // namespace tbv

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
