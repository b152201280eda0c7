// valu_issue.hip — development micro-benchmark (not part of the product): how many cycles a SIMD of gfx950 needs per wave64
// instruction of the kinds the ANIm extension engines are made of (integer VOP2 / VOP3, selects, compares, DPP moves, scalar ALU),
// with v_fma_f32 as the reference the guide quotes (2 cycles).  The issue roofline of bench.py / DESIGN.md §9 is priced with what
// this prints (profiles/archive/r04_valu_issue_ubench.txt).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_issue.hip -o tools/ubench/valu_issue && tools/ubench/valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ __launch_bounds__(64) void issue_kernel(unsigned* out, int iters) {
  unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = blockIdx.x | 1u;
  float f0 = a0, f1 = a1, f2 = a2, f3 = a3, f4 = a4, f5 = a5, f6 = a6, f7 = a7, g = 1.0001f;
  unsigned s0 = blockIdx.x, s1 = 3;
  asm volatile("s_mov_b32 vcc_lo, 0x55555555\n s_mov_b32 vcc_hi, 0x33333333\n s_mov_b32 s20, 0x0f0f0f0f\n s_mov_b32 s21, 0x00ff00ff" ::: "vcc", "s20", "s21");
  for (int it = 0; it < iters; ++it) {
    // 8 independent chains x 8 = 64 instructions per REP64 block, 4 blocks per iteration
#define BODY(INS) REP8(asm volatile(INS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");)
    if (KIND == 0) { REP8(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(g));) }
    if (KIND == 1) { BODY("v_max_u32 %0, %0, %8\n v_max_u32 %1, %1, %8\n v_max_u32 %2, %2, %8\n v_max_u32 %3, %3, %8\n v_max_u32 %4, %4, %8\n v_max_u32 %5, %5, %8\n v_max_u32 %6, %6, %8\n v_max_u32 %7, %7, %8") }
    if (KIND == 2) { BODY("v_sub_u32_e64 %0, %0, %8 clamp\n v_sub_u32_e64 %1, %1, %8 clamp\n v_sub_u32_e64 %2, %2, %8 clamp\n v_sub_u32_e64 %3, %3, %8 clamp\n v_sub_u32_e64 %4, %4, %8 clamp\n v_sub_u32_e64 %5, %5, %8 clamp\n v_sub_u32_e64 %6, %6, %8 clamp\n v_sub_u32_e64 %7, %7, %8 clamp") }
    if (KIND == 3) { BODY("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc") }
    if (KIND == 4) { BODY("v_and_or_b32 %0, %0, %8, %1\n v_and_or_b32 %1, %1, %8, %2\n v_and_or_b32 %2, %2, %8, %3\n v_and_or_b32 %3, %3, %8, %4\n v_and_or_b32 %4, %4, %8, %5\n v_and_or_b32 %5, %5, %8, %6\n v_and_or_b32 %6, %6, %8, %7\n v_and_or_b32 %7, %7, %8, %0") }
    if (KIND == 5) { BODY("v_max3_u32 %0, %0, %8, %1\n v_max3_u32 %1, %1, %8, %2\n v_max3_u32 %2, %2, %8, %3\n v_max3_u32 %3, %3, %8, %4\n v_max3_u32 %4, %4, %8, %5\n v_max3_u32 %5, %5, %8, %6\n v_max3_u32 %6, %6, %8, %7\n v_max3_u32 %7, %7, %8, %0") }
    if (KIND == 6) { BODY("v_cmp_le_u32 vcc, %8, %0\n v_cmp_le_u32 vcc, %8, %1\n v_cmp_le_u32 vcc, %8, %2\n v_cmp_le_u32 vcc, %8, %3\n v_cmp_le_u32 vcc, %8, %4\n v_cmp_le_u32 vcc, %8, %5\n v_cmp_le_u32 vcc, %8, %6\n v_cmp_le_u32 vcc, %8, %7") }
    if (KIND == 7) { BODY("v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %2, %3 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %4 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %4, %5 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %5, %6 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %6, %7 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %7, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1") }
    if (KIND == 8) { REP8(asm volatile("s_add_i32 %0, %0, %1\n s_max_i32 %0, %0, %1\n s_add_i32 %0, %0, %1\n s_max_i32 %0, %0, %1\n s_add_i32 %0, %0, %1\n s_max_i32 %0, %0, %1\n s_add_i32 %0, %0, %1\n s_max_i32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");) }
    if (KIND == 10) { REP8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s20", "s21");) }
    if (KIND == 11) { REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cndmask_b32_e64 %7, %7, %8, s[20:21]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s20", "s21");) }
    if (KIND == 12) { REP8(asm volatile("v_cmp_le_u32 vcc, %8, %0\n v_cndmask_b32 %0, %0, %8, vcc\n v_cmp_le_u32 vcc, %8, %1\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_le_u32 vcc, %8, %2\n v_cndmask_b32 %2, %2, %8, vcc\n v_cmp_le_u32 vcc, %8, %3\n v_cndmask_b32 %3, %3, %8, vcc\n v_cmp_le_u32 vcc, %8, %4\n v_cndmask_b32 %4, %4, %8, vcc\n v_cmp_le_u32 vcc, %8, %5\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_le_u32 vcc, %8, %6\n v_cndmask_b32 %6, %6, %8, vcc\n v_cmp_le_u32 vcc, %8, %7\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s20", "s21");) }
    if (KIND == 13) { REP8(asm volatile("v_max_u32 %0, %0, %8\n v_cndmask_b32 %4, %4, %8, vcc\n v_max_u32 %1, %1, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_max_u32 %2, %2, %8\n v_cndmask_b32 %6, %6, %8, vcc\n v_max_u32 %3, %3, %8\n v_cndmask_b32 %7, %7, %8, vcc\n v_max_u32 %4, %4, %8\n v_cndmask_b32 %0, %0, %8, vcc\n v_max_u32 %5, %5, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_max_u32 %6, %6, %8\n v_cndmask_b32 %2, %2, %8, vcc\n v_max_u32 %7, %7, %8\n v_cndmask_b32 %3, %3, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s20", "s21");) }
    if (KIND == 14) { REP8(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s20", "s21");) }
    if (KIND == 15) { REP8(asm volatile("v_lshlrev_b32 %0, 2, %0\n v_lshlrev_b32 %1, 2, %1\n v_lshlrev_b32 %2, 2, %2\n v_lshlrev_b32 %3, 2, %3\n v_lshlrev_b32 %4, 2, %4\n v_lshlrev_b32 %5, 2, %5\n v_lshlrev_b32 %6, 2, %6\n v_lshlrev_b32 %7, 2, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s20", "s21");) }
    if (KIND == 16) { REP8(asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s20", "s21");) }
    if (KIND == 17) { REP8(asm volatile("v_bfi_b32 %0, %8, %0, %1\n v_bfi_b32 %1, %8, %1, %2\n v_bfi_b32 %2, %8, %2, %3\n v_bfi_b32 %3, %8, %3, %4\n v_bfi_b32 %4, %8, %4, %5\n v_bfi_b32 %5, %8, %5, %6\n v_bfi_b32 %6, %8, %6, %7\n v_bfi_b32 %7, %8, %7, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s20", "s21");) }
    if (KIND == 18) { REP8(asm volatile("v_sub_u32 %0, %0, %8\n v_sub_u32 %1, %1, %8\n v_sub_u32 %2, %2, %8\n v_sub_u32 %3, %3, %8\n v_sub_u32 %4, %4, %8\n v_sub_u32 %5, %5, %8\n v_sub_u32 %6, %6, %8\n v_sub_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s20", "s21");) }
    if (KIND == 19) { REP8(asm volatile("v_min_u32 %0, %0, %8\n v_min_u32 %1, %1, %8\n v_min_u32 %2, %2, %8\n v_min_u32 %3, %3, %8\n v_min_u32 %4, %4, %8\n v_min_u32 %5, %5, %8\n v_min_u32 %6, %6, %8\n v_min_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s20", "s21");) }
    if (KIND == 20) { REP8(asm volatile("v_cmp_le_u32_e64 s[20:21], %8, %0\n v_cmp_le_u32_e64 s[20:21], %8, %1\n v_cmp_le_u32_e64 s[20:21], %8, %2\n v_cmp_le_u32_e64 s[20:21], %8, %3\n v_cmp_le_u32_e64 s[20:21], %8, %4\n v_cmp_le_u32_e64 s[20:21], %8, %5\n v_cmp_le_u32_e64 s[20:21], %8, %6\n v_cmp_le_u32_e64 s[20:21], %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s20", "s21");) }
    if (KIND == 21) { REP8(asm volatile("v_readlane_b32 s20, %0, 63\n v_readlane_b32 s20, %1, 63\n v_readlane_b32 s20, %2, 63\n v_readlane_b32 s20, %3, 63\n v_readlane_b32 s20, %4, 63\n v_readlane_b32 s20, %5, 63\n v_readlane_b32 s20, %6, 63\n v_readlane_b32 s20, %7, 63" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s20", "s21");) }
    if (KIND == 22) { REP8(asm volatile("v_ashrrev_i32 %0, 31, %0\n v_ashrrev_i32 %1, 31, %1\n v_ashrrev_i32 %2, 31, %2\n v_ashrrev_i32 %3, 31, %3\n v_ashrrev_i32 %4, 31, %4\n v_ashrrev_i32 %5, 31, %5\n v_ashrrev_i32 %6, 31, %6\n v_ashrrev_i32 %7, 31, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s20", "s21");) }
    if (KIND == 23) { REP8(asm volatile("v_mad_u32_u24 %0, %0, %8, %1\n v_mad_u32_u24 %1, %1, %8, %2\n v_mad_u32_u24 %2, %2, %8, %3\n v_mad_u32_u24 %3, %3, %8, %4\n v_mad_u32_u24 %4, %4, %8, %5\n v_mad_u32_u24 %5, %5, %8, %6\n v_mad_u32_u24 %6, %6, %8, %7\n v_mad_u32_u24 %7, %7, %8, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s20", "s21");) }
    if (KIND == 24) { REP8(asm volatile("v_or_b32 %0, %0, %8\n v_or_b32 %1, %1, %8\n v_or_b32 %2, %2, %8\n v_or_b32 %3, %3, %8\n v_or_b32 %4, %4, %8\n v_or_b32 %5, %5, %8\n v_or_b32 %6, %6, %8\n v_or_b32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s20", "s21");) }
    if (KIND == 25) { REP8(asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s20", "s21");) }
    if (KIND == 26) { REP8(asm volatile("v_lshrrev_b32 %0, 1, %0\n v_lshrrev_b32 %1, 1, %1\n v_lshrrev_b32 %2, 1, %2\n v_lshrrev_b32 %3, 1, %3\n v_lshrrev_b32 %4, 1, %4\n v_lshrrev_b32 %5, 1, %5\n v_lshrrev_b32 %6, 1, %6\n v_lshrrev_b32 %7, 1, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s20", "s21");) }
    if (KIND == 27) { REP8(asm volatile("v_bfe_i32 %0, %0, 3, 1\n v_bfe_i32 %1, %1, 3, 1\n v_bfe_i32 %2, %2, 3, 1\n v_bfe_i32 %3, %3, 3, 1\n v_bfe_i32 %4, %4, 3, 1\n v_bfe_i32 %5, %5, 3, 1\n v_bfe_i32 %6, %6, 3, 1\n v_bfe_i32 %7, %7, 3, 1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s20", "s21");) }
    if (KIND == 28) { REP8(asm volatile("v_and_b32_dpp %0, %1, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_and_b32_dpp %1, %2, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_and_b32_dpp %2, %3, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_and_b32_dpp %3, %4, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_and_b32_dpp %4, %5, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_and_b32_dpp %5, %6, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_and_b32_dpp %6, %7, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_and_b32_dpp %7, %0, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s20", "s21");) }
    if (KIND == 29) { REP8(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s20", "s21");) }
    if (KIND == 30) { REP8(asm volatile("v_subrev_u32 %0, %8, %0\n v_subrev_u32 %1, %8, %1\n v_subrev_u32 %2, %8, %2\n v_subrev_u32 %3, %8, %3\n v_subrev_u32 %4, %8, %4\n v_subrev_u32 %5, %8, %5\n v_subrev_u32 %6, %8, %6\n v_subrev_u32 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s20", "s21");) }
    if (KIND == 31) { REP8(asm volatile("v_add3_u32 %0, %0, %8, %1\n v_add3_u32 %1, %1, %8, %2\n v_add3_u32 %2, %2, %8, %3\n v_add3_u32 %3, %3, %8, %4\n v_add3_u32 %4, %4, %8, %5\n v_add3_u32 %5, %5, %8, %6\n v_add3_u32 %6, %6, %8, %7\n v_add3_u32 %7, %7, %8, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s20", "s21");) }
    if (KIND == 32) { REP8(asm volatile("v_lshl_add_u32 %0, %0, 1, %8\n v_lshl_add_u32 %1, %1, 1, %8\n v_lshl_add_u32 %2, %2, 1, %8\n v_lshl_add_u32 %3, %3, 1, %8\n v_lshl_add_u32 %4, %4, 1, %8\n v_lshl_add_u32 %5, %5, 1, %8\n v_lshl_add_u32 %6, %6, 1, %8\n v_lshl_add_u32 %7, %7, 1, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s20", "s21");) }
    if (KIND == 33) { REP8(asm volatile("v_max_i32 %0, %0, %8\n v_max_i32 %1, %1, %8\n v_max_i32 %2, %2, %8\n v_max_i32 %3, %3, %8\n v_max_i32 %4, %4, %8\n v_max_i32 %5, %5, %8\n v_max_i32 %6, %6, %8\n v_max_i32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s20", "s21");) }
    if (KIND == 34) { REP8(asm volatile("v_pk_max_u16 %0, %0, %8\n v_pk_max_u16 %1, %1, %8\n v_pk_max_u16 %2, %2, %8\n v_pk_max_u16 %3, %3, %8\n v_pk_max_u16 %4, %4, %8\n v_pk_max_u16 %5, %5, %8\n v_pk_max_u16 %6, %6, %8\n v_pk_max_u16 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s20", "s21");) }
    if (KIND == 35) { REP8(asm volatile("v_pk_sub_u16 %0, %0, %8 clamp\n v_pk_sub_u16 %1, %1, %8 clamp\n v_pk_sub_u16 %2, %2, %8 clamp\n v_pk_sub_u16 %3, %3, %8 clamp\n v_pk_sub_u16 %4, %4, %8 clamp\n v_pk_sub_u16 %5, %5, %8 clamp\n v_pk_sub_u16 %6, %6, %8 clamp\n v_pk_sub_u16 %7, %7, %8 clamp" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s20", "s21");) }
    if (KIND == 9) {      // the engines' mix: a vector instruction and a scalar one alternating (do the two pipes overlap across waves?)
      REP8(asm volatile("v_max_u32 %0, %0, %10\n s_add_i32 %8, %8, %9\n v_max_u32 %1, %1, %10\n s_max_i32 %8, %8, %9\n v_max_u32 %2, %2, %10\n s_add_i32 %8, %8, %9\n v_max_u32 %3, %3, %10\n s_max_i32 %8, %8, %9\n"
                        "v_max_u32 %4, %4, %10\n s_add_i32 %8, %8, %9\n v_max_u32 %5, %5, %10\n s_max_i32 %8, %8, %9\n v_max_u32 %6, %6, %10\n s_add_i32 %8, %8, %9\n v_max_u32 %7, %7, %10\n s_max_i32 %8, %8, %9"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0) : "s"(s1), "v"(b) : "scc");)
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (unsigned)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7) + s0;
}

template <int KIND>
static void run(const char* name, int num_cu, int waves_per_simd, int per_iter, unsigned* out) {
  const int iters = 20000, blocks = num_cu * 4 * waves_per_simd;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(issue_kernel<KIND>, dim3(blocks), dim3(64), 0, 0, out, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL(issue_kernel<KIND>, dim3(blocks), dim3(64), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_simd = (double)waves_per_simd * iters * per_iter;
  printf("%-44s %d waves/SIMD: %8.3f ms  %6.2f ns per instruction and SIMD = %5.2f cycles at 2.4 GHz\n", name, waves_per_simd, ms,
         ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cu = p.multiProcessorCount;
  printf("%s: %d CUs, clock %d kHz\n", p.name, cu, p.clockRate);
  unsigned* out;
  hipMalloc(&out, (size_t)cu * 4 * 8 * 64 * 4);
  for (int w : {1, 8}) {
    run<0>("v_fma_f32", cu, w, 64, out);
    run<1>("v_max_u32 (VOP2)", cu, w, 64, out);
    run<2>("v_sub_u32 clamp (VOP3)", cu, w, 64, out);
    run<3>("v_cndmask_b32 (vcc)", cu, w, 64, out);
    run<4>("v_and_or_b32 (VOP3, 3 operands)", cu, w, 64, out);
    run<5>("v_max3_u32", cu, w, 64, out);
    run<6>("v_cmp_le_u32 -> vcc", cu, w, 64, out);
    run<7>("v_mov_b32 dpp wave_shl:1", cu, w, 64, out);
    run<8>("s_add_i32 / s_max_i32 (per SIMD; one unit per CU)", cu, w, 64, out);
    run<9>("v_max_u32 + s_add/s_max alternating (pairs)", cu, w, 64, out);
    run<10>("v_cndmask_b32 (vcc set before the loop)", cu, w, 64, out);
    run<11>("v_cndmask_b32_e64 (mask in an SGPR pair)", cu, w, 64, out);
    run<12>("v_cmp_le_u32 + v_cndmask_b32 pairs (pairs)", cu, w, 64, out);
    run<13>("v_max_u32 + v_cndmask_b32 alternating (pairs)", cu, w, 64, out);
    run<14>("v_add_u32", cu, w, 64, out);
    run<15>("v_lshlrev_b32", cu, w, 64, out);
    run<16>("v_and_b32", cu, w, 64, out);
    run<17>("v_bfi_b32", cu, w, 64, out);
    run<18>("v_sub_u32 (no clamp)", cu, w, 64, out);
    run<19>("v_min_u32", cu, w, 64, out);
    run<20>("v_cmp_le_u32 -> SGPR pair (e64)", cu, w, 64, out);
    run<21>("v_readlane_b32 (-> SGPR)", cu, w, 64, out);
    run<22>("v_ashrrev_i32", cu, w, 64, out);
    run<23>("v_mad_u32_u24", cu, w, 64, out);
    run<24>("v_or_b32", cu, w, 64, out);
    run<25>("v_xor_b32", cu, w, 64, out);
    run<26>("v_lshrrev_b32", cu, w, 64, out);
    run<27>("v_bfe_i32", cu, w, 64, out);
    run<28>("v_and_b32 dpp wave_shr:1", cu, w, 64, out);
    run<29>("v_mov_b32", cu, w, 64, out);
    run<30>("v_subrev_u32", cu, w, 64, out);
    run<31>("v_add3_u32", cu, w, 64, out);
    run<32>("v_lshl_add_u32", cu, w, 64, out);
    run<33>("v_max_i32", cu, w, 64, out);
    run<34>("v_pk_max_u16", cu, w, 64, out);
    run<35>("v_pk_sub_u16 clamp", cu, w, 64, out);
  }
  return 0;
}
