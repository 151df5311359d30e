// mnav_tbv.h -- the REGISTER-RESIDENT solve of the tile-batch SSSP engine (k_tbv_solve): one wave per (tile, <= 64 plans), the
// plans' distances of the tile's 120 vertices AND of its ghosts in VGPRs, no LDS anywhere.  Included by mnav.hip after mnav_tb.h.
//
// Why (profiles/r05_c2_sq.md): k_tb_solve_q keeps the distances in LDS as [vertex][lane], 30.7 KB per wave = 1.25
// waves per SIMD, and a lone wave is bound by its own instruction issue: 22 VALU + 9 LDS + 7 SALU issue slots per block, one
// after the other (293 cycles per block and wave measured, 88 of them VALU).  The register file of a CU is 512 KB against 160 KB
// of LDS: with the distances in registers a wave needs no LDS, two waves share a SIMD, and a block is 11 vector + 9 scalar
// instructions (88 cycles per block and SIMD measured, tools/gpu_tbv_micro.py: the SIMD issues one instruction per four cycles).
//
// How: the lanes of the wave are plans of ONE tile, so a row index is wave-uniform -- the hardware's VGPR index mode
// (MODE.gpr_idx_en; M0[7:0] is added to the register number of the operands selected by M0[15:12]) addresses row r of the
// window as v[72 + r].  The window lives in FIXED physical registers; the compiler is held below it (amdgpu_num_vgpr) and never
// sees it; everything that touches it is inline assembly:
//   v[0:39]     compiler
//   v[40:55]    ring of four stream chunks (4 registers each)                 }
//   v[56:63]    six candidates + two minima                                    } only inside the three stream passes
//   v[64:68]    lane's byte offset into a chunk; best / candidate / +inf / tmp }
//   v[72:135]   window rows 0..63: the ghost slots of the slice (copies of the neighbour tiles' boundary vertices)
//   v[136:255]  window rows 64..183: owned vertex r of the tile at row 64 + r
// All three phases that touch the tile's graph -- ghosts -> owned ("pre"), the Gauss-Seidel sweeps, owned -> ghosts ("post") -- are
// streams of the SAME 16-dword block (mnav_tb_build.h, V layout): a target row, six source rows, six weights.  The relaxation
// itself is dijkstra_mesh_planner.cpp:331 (one float32 add per edge, minimum over the sources): same arithmetic, same fixed point
// as k_tb_solve_q and every other engine, bit for bit.
#pragma once

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"     // (the clobber list names v255 on purpose: the kernel's register count must cover the window)
namespace {
namespace tbv {

// amdgpu_num_vgpr(n) holds the compiler to 2 n registers on gfx950 (unified VGPR / AGPR file; measured: n = 32, 48, 64, 96 ->
// NumVgprs 64, 96, 128, 192): v[0:39] here
#define TBV_COMPILER_VGPRS __attribute__((amdgpu_num_vgpr(20)))
constexpr int kRows = 120;                // owned rows = TbState::T of this kernel
constexpr int kGhostRows = (int)kTbvGhostRows;
static_assert(kGhostRows == 64, "the window starts at v72: ghosts v[72:135], owned rows v[136:255]");

// ---- owned rows by a compile-time index
template <int R> __device__ __forceinline__ void img_set(uint32_t x) { asm volatile("v_mov_b32 v[136+%c1], %0" : : "v"(x), "n"(R)); }
template <int R> __device__ __forceinline__ uint32_t img_get() { uint32_t x; asm volatile("v_mov_b32 %0, v[136+%c1]" : "=v"(x) : "n"(R)); return x; }
// loads straight into the window, all in flight together; img_loads_wait() before the rows are used
template <int R> __device__ __forceinline__ void img_load(const uint32_t* p) { asm volatile("global_load_dword v[136+%c1], %0, off" : : "v"(p), "n"(R) : "memory"); }
template <int C> __device__ __forceinline__ void img_load_quad(const void* p) { asm volatile("global_load_dwordx4 v[136+%c1:139+%c1], %0, off offset:%c2" : : "v"(p), "n"(4 * C), "n"(16 * C) : "memory"); }
template <int Q> __device__ __forceinline__ void ghost_load_quad(const void* p) { asm volatile("global_load_dwordx4 v[72+%c1:75+%c1], %0, off offset:%c2" : : "v"(p), "n"(4 * Q), "n"(16 * (kRows / 4 + Q)) : "memory"); }
template <int C> __device__ __forceinline__ void img_store_quad(void* p) { asm volatile("global_store_dwordx4 %0, v[136+%c1:139+%c1], off offset:%c2" : : "v"(p), "n"(4 * C), "n"(16 * C) : "memory"); }
__device__ __forceinline__ void img_loads_wait() { asm volatile("s_waitcnt vmcnt(0)" : : : "memory"); }
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f)
{
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
// ---- window rows by a wave-uniform index (an SGPR)
__device__ __forceinline__ uint32_t win_read(uint32_t row)
{
  uint32_t x;
  asm volatile("s_set_gpr_idx_on %1, gpr_idx(SRC0)\n\tv_mov_b32 %0, v72\n\ts_set_gpr_idx_off" : "=v"(x) : "s"(row));
  return x;
}
// ---- the stream passes.
// A chunk = 4 blocks of 16 dwords, stored transposed (dword q of block j at chunk dword 4 q + j).  The row indices of a block come
// as ready-made values of M0 (index in bits 7:0, the operands it applies to in bits 15:12 -- 0x2000: source 1, 0xA000: source 1 and
// destination):  d0 = (mode | target) | (0x2000 | source0) << 16,  d1 = (0x2000 | source1) | (0x2000 | source2) << 16,
// d2 = (0x2000 | source3) | (0x2000 | source4) << 16,  d3 = (0x2000 | source5) | flags << 16,  d8 .. d13 = the six weights.
// The index dwords of a chunk are its first 64 bytes: ONE s_load_dwordx16 per chunk brings them into SGPRs (the chunk after the
// current one is in flight, two SGPR sets in turn); the weights arrive through the vector path -- lane l of every 16-lane row
// loads the 16 bytes l & 15 of the chunk and holds dword l & 15 of block j in register j -- and are consumed as DPP operands
// (row_newbcast: lane 8 + k of the row).  Common part of a block, 8 scalar + 9 vector instructions:
//     6 x (M0 <- index; v_add_f32_dpp t_k, w_k, row[source_k]), M0 <- 0, 2 v_min3_u32 + v_min_u32, [post: fold the carried candidate], M0 <- target
// then the phase's tail (below).  Reads and writes of the window are register accesses: block j + 1 sees what block j wrote -- the
// plain Gauss-Seidel sweep, no forwarding rule.  Four chunks are in registers or in flight.
// (The first version of the sweeps -- indices through v_readlane, s_set_gpr_idx_idx per access, mode switches around the write:
//  16 + 16 instructions -- measured 129 cycles per block and SIMD with two waves on the SIMD, 205 with one; this one 88 / 141.)
#define TBV_ADD(t, D, k) "v_add_f32_dpp v" #t ", v[" #D "], v72 row_newbcast:" #k " row_mask:0xf bank_mask:0xf\n\t"
#define TBV_HEAD(D, i0, i1, i2, i3, FOLD)                                                                               \
  "s_lshr_b32 m0, s[" #i0 "], 16\n\t"      TBV_ADD(56, D, 8)                                                           \
  "s_and_b32 m0, s[" #i1 "], 0xffff\n\t"   TBV_ADD(57, D, 9)                                                           \
  "s_lshr_b32 m0, s[" #i1 "], 16\n\t"      TBV_ADD(58, D, 10)                                                          \
  "s_and_b32 m0, s[" #i2 "], 0xffff\n\t"   TBV_ADD(59, D, 11)                                                          \
  "s_lshr_b32 m0, s[" #i2 "], 16\n\t"      TBV_ADD(60, D, 12)                                                          \
  "s_and_b32 m0, s[" #i3 "], 0xffff\n\t"   TBV_ADD(61, D, 13)                                                          \
  "s_mov_b32 m0, 0\n\t"                                                                                                \
  "v_min3_u32 v63, v56, v57, v58\n\t"                                                                                  \
  "v_min3_u32 v62, v59, v60, v61\n\t"                                                                                  \
  "v_min_u32 v62, v63, v62\n\t"                                                                                        \
  FOLD                                                                                                                 \
  "s_and_b32 m0, s[" #i0 "], 0xffff\n\t"
// sweeps: did the target improve (OR-ed into the sweep's flag), target <- min
#define TBV_BLOCK_SWEEP(D, i0, i1, i2, i3, tag)                                                                         \
  TBV_HEAD(D, i0, i1, i2, i3, "")                                                                                       \
  "v_cmp_lt_u32 vcc, v62, v72\n\t"                                                                                     \
  "v_min_u32 v72, v62, v72\n\t"                                                                                        \
  "s_or_b64 s[70:71], s[70:71], vcc\n\t"
// pre: the same, and per lane the smallest value a ghost lowered a row to (v65) with the sweep order of that row (v66, flags bits 0-1)
#define TBV_BLOCK_PRE(D, i0, i1, i2, i3, tag)                                                                           \
  TBV_HEAD(D, i0, i1, i2, i3, "")                                                                                       \
  "v_cmp_lt_u32 vcc, v62, v72\n\t"                                                                                     \
  "v_min_u32 v72, v62, v72\n\t"                                                                                        \
  "s_mov_b32 m0, 0\n\t"                                                                                                \
  "v_cndmask_b32 v68, v67, v62, vcc\n\t"                                                                               \
  "v_cmp_lt_u32 vcc, v68, v65\n\t"                                                                                     \
  "s_lshr_b32 s85, s[" #i3 "], 16\n\t"                                                                                 \
  "v_cndmask_b32 v65, v65, v68, vcc\n\t"                                                                               \
  "v_mov_b32 v68, s85\n\t"                                                                                             \
  "v_cndmask_b32 v66, v66, v68, vcc\n\t"
// post: the block's target is a ghost, read only: the candidate (carried over continuation blocks in v66) against the ghost's
// value; at the ghost's last block the undercutting candidates go into the owner tile's best (v65); at the owner's last ghost the
// best is parked in window row `group` (the ghosts of that row and of every row before it have been looked at) and the next
// owner's begins
#define TBV_BLOCK_POST(D, i0, i1, i2, i3, tag)                                                                          \
  TBV_HEAD(D, i0, i1, i2, i3, "v_min_u32 v62, v62, v66\n\t")                                                            \
  "v_cmp_lt_u32 vcc, v62, v72\n\t"                                                                                     \
  "s_mov_b32 m0, 0\n\t"                                                                                                \
  "s_bitcmp1_b32 s[" #i3 "], 16\n\t"                                                                                   \
  "s_cbranch_scc0 Ltbv_pc" #tag "_%=\n\t"                                                                              \
  "v_cndmask_b32 v68, v67, v62, vcc\n\t"                                                                               \
  "v_min_u32 v65, v65, v68\n\t"                                                                                        \
  "v_mov_b32 v66, v67\n\t"                                                                                             \
  "s_bitcmp1_b32 s[" #i3 "], 17\n\t"                                                                                   \
  "s_cbranch_scc0 Ltbv_pn" #tag "_%=\n\t"                                                                              \
  "s_or_b32 m0, s85, 0x8000\n\t"                                                                                       \
  "v_mov_b32 v72, v65\n\t"                                                                                             \
  "s_mov_b32 m0, 0\n\t"                                                                                                \
  "v_mov_b32 v65, v67\n\t"                                                                                             \
  "s_add_u32 s85, s85, 1\n\t"                                                                                          \
  "s_branch Ltbv_pn" #tag "_%=\n\t"                                                                                    \
  "Ltbv_pc" #tag "_%=:\n\t"                                                                                            \
  "v_mov_b32 v66, v62\n\t"                                                                                             \
  "Ltbv_pn" #tag "_%=:\n\t"
// the chunk under the vector load cursor (byte offset s68 from the pass's first chunk) into ring slot `b`; cursor + 1 chunk, cyclic
#define TBV_LOAD(b)                                                                                                    \
  "s_add_u32 s86, s72, s68\n\t"  "s_addc_u32 s87, s73, 0\n\t"                                                          \
  "global_load_dwordx4 v[" #b ":" #b "+3], v64, s[86:87]\n\t"                                                          \
  "s_add_u32 s68, s68, 0x100\n\t"  "s_cmp_eq_u32 s68, s75\n\t"  "s_cselect_b32 s68, 0, s68\n\t"
// the index dwords of the chunk under the scalar cursor (s78) into the SGPR set that starts at `S`; cursor + 1 chunk, cyclic
// (tried instead: the index dwords out of the ring registers by v_readlane, no scalar loads -- four more vector instructions per
//  block: 105 instead of 88 cycles per block and SIMD on its own, 128 instead of 123.5 ms per 7168-plan engine run)
#define TBV_SLOAD(S)                                                                                                   \
  "s_load_dwordx16 s[" #S ":" #S "+15], s[72:73], s78\n\t"                                                             \
  "s_add_u32 s78, s78, 0x100\n\t"  "s_cmp_eq_u32 s78, s75\n\t"  "s_cselect_b32 s78, 0, s78\n\t"
// one chunk: its weights in ring slot b, its indices in set SA; the next chunk's indices go to set SB.  END: what follows the chunk
#define TBV_CHUNK(BLOCK, b, SA, SB, tag, END)                                                                          \
  "s_waitcnt vmcnt(3) lgkmcnt(0)\n\t"                                                                                  \
  TBV_SLOAD(SB)                                                                                                        \
  "s_set_gpr_idx_on s74, gpr_idx(SRC1)\n\t"                                                                            \
  BLOCK(b, SA, SA + 4, SA + 8, SA + 12, tag##0)  BLOCK(b + 1, SA + 1, SA + 5, SA + 9, SA + 13, tag##1)                 \
  BLOCK(b + 2, SA + 2, SA + 6, SA + 10, SA + 14, tag##2)  BLOCK(b + 3, SA + 3, SA + 7, SA + 11, SA + 15, tag##3)       \
  "s_set_gpr_idx_off\n\t"                                                                                              \
  TBV_LOAD(b)                                                                                                          \
  "s_sub_u32 s80, s80, 1\n\t"                                                                                          \
  "s_cmp_lg_u32 s80, 0\n\t"                                                                                            \
  "s_cbranch_scc1 Ltbv_nx" #tag "_%=\n\t"                                                                              \
  END                                                                                                                  \
  "Ltbv_nx" #tag "_%=:\n\t"
// end of a sweep: count it; nothing changed in any lane -> done; the cap -> overrun
#define TBV_END_SWEEP                                                                                                  \
  "s_mov_b32 s80, s74\n\t"                                                                                             \
  "s_add_u32 s84, s84, 1\n\t"                                                                                          \
  "s_or_b64 s[88:89], s[88:89], s[70:71]\n\t"                                                                          \
  "s_or_b64 s[70:71], s[70:71], s[82:83]\n\t"                                                                          \
  "s_cmp_eq_u64 s[70:71], 0\n\t"                                                                                       \
  "s_cbranch_scc1 Ltbv_done_%=\n\t"                                                                                    \
  "s_cmp_ge_u32 s84, s81\n\t"                                                                                          \
  "s_cbranch_scc1 Ltbv_over_%=\n\t"                                                                                    \
  "s_mov_b64 s[70:71], 0\n\t"
#define TBV_END_PASS "s_branch Ltbv_done_%=\n\t"
#define TBV_LOOP(BLOCK, END)                                                                                           \
  TBV_LOAD(40) TBV_LOAD(44) TBV_LOAD(48) TBV_LOAD(52)                                                                  \
  TBV_SLOAD(36)                                                                                                        \
  "Ltbv_top_%=:\n\t"                                                                                                   \
  TBV_CHUNK(BLOCK, 40, 36, 52, c0, END) TBV_CHUNK(BLOCK, 44, 52, 36, c1, END) TBV_CHUNK(BLOCK, 48, 36, 52, c2, END) TBV_CHUNK(BLOCK, 52, 52, 36, c3, END) \
  "s_branch Ltbv_top_%=\n\t"
#define TBV_CLOBBERS "memory", "vcc", "scc", "m0", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", \
  "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s70", "s71", "s72", "s73", "s74", "s75", \
  "s78", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "v40", "v68", "v255"

struct Uni { uint32_t lo, hi, nch; };   // a stream's first chunk (address) and its chunk count, made uniform for the "s" operands
__device__ __forceinline__ Uni uni(const uint32_t* first, uint32_t nch)
{
  // (every scalar operand is made uniform explicitly: a 64-bit "s" operand the compiler thinks divergent is handed over in VGPRs)
  const unsigned long long a = (unsigned long long)(uintptr_t)first;
  return Uni{ tb::rfl((uint32_t)a), tb::rfl((uint32_t)(a >> 32)), tb::rfl(nch) };
}

// The Gauss-Seidel sweeps of one activation over the four orders' chunks (order k at sweep0 + k * nch chunks), starting with
// `first_order`, until a sweep changes nothing in any lane.  Returns the number of sweeps; `overrun` when `cap` did not suffice.
// `force` != 0: every sweep counts as "changed" (timing runs: exactly `cap` sweeps).  nch >= 1.  `lowered`: the lanes in which
// some sweep lowered some row.
__device__ __forceinline__ uint32_t tbv_sweeps(const uint32_t* sweep0, uint32_t nch, uint32_t first_order, uint32_t cap, uint32_t lane_off,
                                               uint32_t force, bool& overrun, unsigned long long& lowered)
{
  uint32_t sweeps, ovr, low_lo, low_hi;
  const Uni u = uni(sweep0, nch);
  first_order = tb::rfl(first_order); cap = tb::rfl(cap); force = tb::rfl(force);
  asm volatile(
      "v_mov_b32 v64, %[off]\n\t"
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
      "s_mov_b64 s[88:89], 0\n\t"
      TBV_LOOP(TBV_BLOCK_SWEEP, TBV_END_SWEEP)
      "Ltbv_over_%=:\n\t"
      "s_mov_b32 %[ovr], 1\n\t"
      "s_branch Ltbv_end_%=\n\t"
      "Ltbv_done_%=:\n\t"
      "s_mov_b32 %[ovr], 0\n\t"
      "Ltbv_end_%=:\n\t"
      "s_mov_b32 %[sw], s84\n\t"
      "s_mov_b32 %[llo], s88\n\t"
      "s_mov_b32 %[lhi], s89\n\t"
      "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"                              // the speculative loads of the next sweep land in the ring / the index set: nothing may follow them
      : [sw] "=s"(sweeps), [ovr] "=s"(ovr), [llo] "=s"(low_lo), [lhi] "=s"(low_hi)
      : [stlo] "s"(u.lo), [sthi] "s"(u.hi), [nch] "s"(u.nch), [ord] "s"(first_order), [cap] "s"(cap), [off] "v"(lane_off), [force] "s"(force)
      : TBV_CLOBBERS);
  overrun = ovr != 0u;
  lowered = ((unsigned long long)low_hi << 32) | low_lo;
  return sweeps;
}

// ghosts -> owned: one pass over the tile's pre chunks.  Per lane: the smallest value a ghost lowered one of its rows to (bits;
// +inf: nothing was lowered) and the sweep order that runs with a wave entering at that row.
__device__ __forceinline__ void tbv_pre(const uint32_t* pre0, uint32_t nch, uint32_t lane_off, uint32_t& min_new, uint32_t& order)
{
  const Uni u = uni(pre0, nch);
  asm volatile(
      "v_mov_b32 v64, %[off]\n\t"
      "v_mov_b32 v67, 0x7f800000\n\t"
      "v_mov_b32 v65, v67\n\t"
      "v_mov_b32 v66, 0\n\t"
      "s_mov_b32 s72, %[stlo]\n\t"
      "s_mov_b32 s73, %[sthi]\n\t"
      "s_mov_b32 s74, %[nch]\n\t"
      "s_lshl_b32 s75, s74, 8\n\t"
      "s_mov_b32 s68, 0\n\t"
      "s_mov_b32 s78, 0\n\t"
      "s_mov_b32 s80, s74\n\t"
      TBV_LOOP(TBV_BLOCK_PRE, TBV_END_PASS)
      "Ltbv_done_%=:\n\t"
      "v_mov_b32 %[mn], v65\n\t"
      "v_mov_b32 %[od], v66\n\t"
      "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
      : [mn] "=v"(min_new), [od] "=v"(order)
      : [stlo] "s"(u.lo), [sthi] "s"(u.hi), [nch] "s"(u.nch), [off] "v"(lane_off)
      : TBV_CLOBBERS);
}

// owned -> ghosts: one pass over the tile's post chunks.  Afterwards window row g (g < the tile's groups) holds, per lane, the
// smallest candidate that undercuts a ghost owned by the tile of group g (+inf: none) -- what that tile is woken with.
__device__ __forceinline__ void tbv_post(const uint32_t* post0, uint32_t nch, uint32_t lane_off)
{
  const Uni u = uni(post0, nch);
  asm volatile(
      "v_mov_b32 v64, %[off]\n\t"
      "v_mov_b32 v67, 0x7f800000\n\t"
      "v_mov_b32 v65, v67\n\t"
      "v_mov_b32 v66, v67\n\t"
      "s_mov_b32 s85, 0\n\t"
      "s_mov_b32 s72, %[stlo]\n\t"
      "s_mov_b32 s73, %[sthi]\n\t"
      "s_mov_b32 s74, %[nch]\n\t"
      "s_lshl_b32 s75, s74, 8\n\t"
      "s_mov_b32 s68, 0\n\t"
      "s_mov_b32 s78, 0\n\t"
      "s_mov_b32 s80, s74\n\t"
      TBV_LOOP(TBV_BLOCK_POST, TBV_END_PASS)
      "Ltbv_done_%=:\n\t"
      "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
      :
      : [stlo] "s"(u.lo), [sthi] "s"(u.hi), [nch] "s"(u.nch), [off] "v"(lane_off)
      : TBV_CLOBBERS);
}

typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 16 bytes at any dword address (ghost slots of a run)
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
typedef const MNAV_CONST uint32_t* cwords_t;
template <class P> __device__ __forceinline__ cwords_t cwords(P* p) { return (cwords_t)(uintptr_t)p; }

}  // namespace tbv

// ---------------------------------------------------------------------------------------------
// The register-resident solve: one wave per work item = (tile, <= 64 plans of its bucket), one plan per lane.
// ---------------------------------------------------------------------------------------------
// Same item protocol, same slices, same wake-ups and exports as k_tb_solve_q (mnav_tb.h) -- the two kernels are interchangeable
// per item --, but everything that was uniform per QUARTER there is uniform per WAVE here: the tile's words and the export records
// are read through the scalar cache (constant address space, s_load), the three graph phases are stream passes over the window.
// The window holds plain values throughout (no "lowered in this activation" marks): a lane in which nothing was lowered stores
// nothing, the others store their slice and boundary values whole.
template <int T>
__global__ __launch_bounds__(64) TBV_COMPILER_VGPRS
void k_tbv_solve(tb::Args A, const uint32_t* __restrict__ vtile, const uint32_t* __restrict__ vstream, const uint32_t* __restrict__ vgroups,
                 const TbvExp* __restrict__ vexps, int par)
{
  static_assert(T == tbv::kRows && T % 20 == 0, "the window holds 120 owned rows");
  const int lane = threadIdx.x;
  const uint32_t NP = A.NP;
  const uint32_t n_items = tb::rfl(A.ctl->n_items);
  const uint32_t lane_off = 16u * ((uint32_t)lane & 15u);
  const tbv::cwords_t ctiles = tbv::cwords(A.tiles), cvtile = tbv::cwords(vtile), cgroups = tbv::cwords(vgroups);
  uint32_t my_items = 0, my_acts = 0, my_sweeps = 0, my_wakes = 0, my_first = 0;
#ifdef MNAV_TB_TIMING
  unsigned long long tt[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, t_last = __builtin_readcyclecounter();
#endif
  // The ticket of the NEXT item is taken while this item's slice is in flight, its item words while the passes run: two of the four
  // dependent round trips between two items (ticket -> item words -> tile words / bucket entry -> slice) are off the chain.
  // (the first ticket of a wave is its own index, the counter hands out the tickets behind the grid's: see k_tb_solve_q)
  uint32_t it = blockIdx.x;
  u32x2 item = { 0u, 0u };
  if (it < n_items) item = ((MNAV_GLOBAL const u32x2*)as_global(A.items))[it];
  while (it < n_items) {
    const uint32_t t = tb::rfl(item.x), start = tb::rfl(item.y & 0xFFFFu), count = tb::rfl(item.y >> 16);
    const tbv::cwords_t hw = ctiles + (size_t)t * 16u, vw = cvtile + (size_t)t * kTbvTileWords;
    const uint32_t soff = hw[tb::kTwSoff], slen = hw[tb::kTwSl], nh = hw[tb::kTwNh];
    const uint32_t pre_off = vw[0], pre_chunks = vw[1], sweep_off = vw[2], sweep_chunks = vw[3], post_off = vw[4], post_chunks = vw[5], grp_off = vw[6], grp_n = vw[7];
    const uint32_t vexp_off = vw[8], vexp_n = vw[9];
    ++my_items; my_acts += (lane == 0) ? count : 0u;
    // lanes beyond `count` shadow the last plan of the item and store nothing
    const bool active = (uint32_t)lane < count;
    const uint32_t p = A.bucket[(size_t)t * NP + start + min((uint32_t)lane, max(count, 1u) - 1u)];
    MNAV_GLOBAL float* const sl = as_global(A.D) + ((size_t)soff * NP + (size_t)p * slen);
    TB_STAMP(0);
    // ---- the slice: owned slots and ghosts straight into the window, all loads in flight together
    tbv::static_for<0, T / 4>([&](auto c) { tbv::img_load_quad<decltype(c)::value>((const void*)sl); });
    {
      const uint32_t nq = (nh + 3u) >> 2;
      tbv::static_for<0, tbv::kGhostRows / 4>([&](auto q) { if ((uint32_t)decltype(q)::value < nq) tbv::ghost_load_quad<decltype(q)::value>((const void*)sl); });
    }
    uint32_t it_n = 0;
    if (lane == 0) it_n = atomicAdd(&A.ctl->next_item, 1u);           // (returns with the slice)
    tbv::img_loads_wait();
    it_n = gridDim.x + tb::rfl(it_n);
    u32x2 item_n = { 0u, 0u };
    if (it_n < n_items) item_n = ((MNAV_GLOBAL const u32x2*)as_global(A.items))[it_n];   // in flight during the passes
    TB_STAMP(1);
    // ---- ghosts -> owned (the ghosts are constant during the activation); the order most lanes ask for starts the sweeps
    uint32_t first_order = 0;
    bool changed = false;                                              // this lane: a ghost or a sweep lowered one of its rows
    if (pre_chunks) {
      uint32_t mn, gord;
      tbv::tbv_pre(vstream + (size_t)pre_off * kTbChunk, pre_chunks, lane_off, mn, gord);
      changed = mn != kTbInfBits;
      const bool votes = active && changed;
      uint32_t bestc = 0;
#pragma unroll
      for (uint32_t o = 0; o < 4; ++o) {
        const uint32_t cn = (uint32_t)__popcll(__ballot(votes && (gord & 3u) == o));
        if (cn > bestc) { bestc = cn; first_order = o; }
      }
    }
    TB_STAMP(2);
    // ---- Gauss-Seidel sweeps to the tile-local fixed point of every lane
    uint32_t sweep = 1;
    if (sweep_chunks) {
      bool overrun;
      unsigned long long lowered;
      sweep = tbv::tbv_sweeps(vstream + (size_t)sweep_off * kTbChunk, sweep_chunks, first_order, 16u * T, lane_off, 0u, overrun, lowered);
      if (overrun && lane == 0) A.ctl->err = 1u;
      changed = changed || ((lowered >> lane) & 1ull);                 // (the wave source's tile: lowered by no ghost, yet its rows move)
    }
    my_sweeps += sweep;
    TB_STAMP(3);
    // ---- owned -> ghosts: a neighbour tile is woken when a candidate undercuts what we know of its vertex.  The pass leaves
    // each neighbour's best candidate in a window row (the first rows: ghosts that have been looked at by then).  The wake-ups of
    // four neighbours are in flight together: atomicMin on the pair's wake-up value (its old value says whether this is the
    // pair's first wake-up: those are counted once per item), the flag of its block of 64 plans, and ONE atomicMin per lane on
    // the plan's smallest pending value at the end.  (k_tb_solve_q looks at the pending value with a plain load first and
    // pipelines look / atomic / count over the neighbours: three dependent round trips per neighbour that one wave per SIMD
    // could overlap with nothing else; here the chain is one round trip per four neighbours.)
    if (post_chunks) {
      tbv::tbv_post(vstream + (size_t)post_off * kTbChunk, post_chunks, lane_off);
      TB_STAMP(7);
      MNAV_GLOBAL uint32_t* const pend_p = as_global(A.pend) + p;
      MNAV_GLOBAL uint8_t* const pflag_p = as_global(A.pflag) + (p >> 6);
      uint32_t n_first = 0, mbest = kTbInfBits;
      for (uint32_t g0 = 0; g0 < grp_n; g0 += 4u) {
        uint32_t best[4], old[4];
        bool want[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool live = g0 + q < grp_n;                            // (uniform)
          best[q] = live ? tbv::win_read(g0 + q) : kTbInfBits;
          want[q] = live && active && best[q] != kTbInfBits;
          old[q] = 0u;
          if (want[q]) {
            const size_t at = (size_t)cgroups[grp_off + g0 + q] * NP;
            old[q] = atomicMin((uint32_t*)(pend_p + at), best[q]);
            pflag_p[(size_t)cgroups[grp_off + g0 + q] * A.nblk] = 1;
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (want[q]) {
            n_first += (old[q] == kTbInfBits) ? 1u : 0u;
            if (best[q] < old[q]) { mbest = min(mbest, best[q]); ++my_wakes; }
          }
      }
      if (mbest != kTbInfBits) atomicMin((uint32_t*)(as_global(A.marr[par ^ 1]) + p), mbest);
      my_first |= n_first;                                             // (per lane; ONE store per wave at the end: a store per item on
                                                                       // the one word every wave writes cost 57 ms of 100 -- the next pass waits for it)
    }
    TB_STAMP(5);
    // ---- write back and export.  A lane in which neither a ghost nor a sweep lowered a row has changed nothing: it stores
    // nothing.  The others store their whole slice and every boundary value --
    // unconditionally: telling the lowered rows from the rest takes the slice as it was loaded (a second read of it: a round trip
    // per five quads, 16 % of the kernel when it was done), the stores cost no round trip at all.
    const bool wrote = active && changed;
    if (wrote) tbv::static_for<0, T / 4>([&](auto c) { tbv::img_store_quad<decltype(c)::value>((void*)sl); });
    TB_STAMP(4);
    // export runs (mnav_tb_build.h, TbvExp: up to four boundary rows whose ghost copies in one neighbour's slice are adjacent): lane l
    // fetches run k0 + l, the wave walks them by v_readlane; one store of 4 / 8 / 12 / 16 bytes per run
    for (uint32_t k0 = 0; k0 < vexp_n; k0 += 64u) {
      MNAV_GLOBAL const u32x4* const rp = (MNAV_GLOBAL const u32x4*)as_global(vexps) + 2u * ((size_t)vexp_off + k0 + (uint32_t)lane);   // (64 runs of slack behind the last tile's)
      const u32x4 ra = rp[0];
      const uint32_t rn = rp[1].x;
      const uint32_t n = min(64u, vexp_n - k0);
      for (uint32_t j = 0; j < n; ++j) {
        const uint32_t rows = (uint32_t)__builtin_amdgcn_readlane((int)ra.x, (int)j), so = (uint32_t)__builtin_amdgcn_readlane((int)ra.y, (int)j);
        const uint32_t sl2 = (uint32_t)__builtin_amdgcn_readlane((int)ra.z, (int)j), off = (uint32_t)__builtin_amdgcn_readlane((int)ra.w, (int)j);
        const uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)rn, (int)j);
        tbv::f32x4u v;
        v.x = u2f(tbv::win_read(rows & 255u)); v.y = u2f(tbv::win_read((rows >> 8) & 255u));
        v.z = u2f(tbv::win_read((rows >> 16) & 255u)); v.w = u2f(tbv::win_read(rows >> 24));
        if (wrote) {
          MNAV_GLOBAL float* const dst = as_global(A.D) + ((size_t)so * NP + ((size_t)p * sl2 + off));
          if (cnt == 4u) *(MNAV_GLOBAL tbv::f32x4u*)dst = v;
          else if (cnt == 3u) { *(MNAV_GLOBAL tbv::f32x2u*)dst = tbv::f32x2u{ v.x, v.y }; dst[2] = v.z; }
          else if (cnt == 2u) *(MNAV_GLOBAL tbv::f32x2u*)dst = tbv::f32x2u{ v.x, v.y };
          else dst[0] = v.x;
        }
      }
    }
    TB_STAMP(6);
    it = it_n; item = item_n;
  }
#ifdef MNAV_TB_TIMING
  if (lane == 0) for (int k = 0; k < 8; ++k) if (tt[k]) atomicAdd(&g_tb_timing[k], tt[k]);
#endif
  my_wakes = wave_sum(my_wakes); my_acts = wave_sum(my_acts);
  if (__any(my_first != 0u) && lane == 0) A.ctl->n_cand[par ^ 1] = 1u;   // "pairs are pending" (k_tb_scan)
  if (lane == 0 && my_items && blockIdx.x < A.wstat_slots) {         // (its own slot, a plain read-modify-write: see k_tb_stats)
    unsigned long long* const ws = A.wstat + 4u * blockIdx.x;
    ws[0] += (unsigned long long)my_items; ws[1] += (unsigned long long)my_acts; ws[2] += (unsigned long long)my_sweeps; ws[3] += (unsigned long long)my_wakes;
  }
}

// Test / timing entry of the sweep routine on its own (mnav_debug_tbv_sweeps): every wave loads ITS owned rows [row][lane] from
// `img`, runs the sweeps over the one stream and stores the rows back.
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
    unsigned long long low;
    total += tbv::tbv_sweeps(stream, nch, first_order, cap, 16u * ((uint32_t)lane & 15u), force ? ~0u : 0u, ovr, low);
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
