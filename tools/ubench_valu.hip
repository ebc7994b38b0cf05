// ubench_valu.hip -- issue cadence of the integer-VALU instructions the Viterbi kernel (k_viterbi3.hpp) is made of, on gfx950.
// Question: does a wave64 VALU instruction occupy its SIMD for 4 cycles or for 2?  The kernel's
// "VALU issue slots used" figure depends on it, and so does the lever (instruction count vs occupancy/latency).
// Method: one workgroup per CU (grid = #CUs), W waves per SIMD (blockDim = 256 W), every wave runs `iters` trips of a
// block of 64 instructions, either ONE dependent chain or EIGHT independent chains; s_memtime (shader clock) around the
// loop.  Reported per configuration: the mean and the slowest wave's cycles per instruction.  The waves of a SIMD do NOT share it
// evenly (the arbiter favours the older wave, which runs at nearly its solo speed while the younger fills gaps), so the SIMD's
// throughput is instructions x waves / (cycles until the LAST wave ends): simd_cycles_per_inst = slowest_wave / waves_per_simd.
// Its floor over W is the issue cadence of the instruction class.
// Build: hipcc --offload-arch=gfx950 -O2 -o ubench_valu tools/ubench_valu.hip ; run: ./ubench_valu > profiles/rNN_ubench_valu.json
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
#include <algorithm>

#define REAL_STEPS \
          "v_perm_b32 %2, 0, %5, %6\n v_pk_add_u16 %3, %0, %2\n v_pk_sub_i16 %0, %0, %2\n v_pk_sub_i16 %4, %1, %2\n v_pk_add_u16 %1, %1, %2\n" \
          "v_mov_b32_dpp %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_pk_max_i16 %0, %3, %0\n v_mov_b32_dpp %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
          "v_and_or_b32 %0, %0, %8, %7\n v_pk_max_i16 %1, %4, %1\n v_perm_b32 %2, 0, %5, %6\n v_and_or_b32 %1, %1, %8, %7\n" \
          "v_pk_add_u16 %3, %0, %2\n v_pk_sub_i16 %0, %0, %2\n v_pk_sub_i16 %4, %1, %2\n v_pk_add_u16 %1, %1, %2\n" \
          "v_mov_b32_dpp %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_pk_max_i16 %0, %3, %0\n v_mov_b32_dpp %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
          "v_and_or_b32 %0, %0, %8, %7\n v_pk_max_i16 %1, %4, %1\n v_perm_b32 %2, 0, %5, %6\n v_and_or_b32 %1, %1, %8, %7\n" \
          "v_pk_add_u16 %3, %0, %2\n v_pk_sub_i16 %0, %0, %2\n v_pk_add_u16 %4, %1, %2\n v_pk_sub_i16 %1, %1, %2\n" \
          "v_mov_b32_dpp %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_pk_max_i16 %0, %3, %0\n v_mov_b32_dpp %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
          "v_and_or_b32 %0, %0, %8, %7\n v_pk_max_i16 %1, %4, %1\n v_perm_b32 %2, 0, %5, %6\n v_and_or_b32 %1, %1, %8, %7\n" \
          "v_pk_add_u16 %3, %0, %2\n v_pk_sub_i16 %0, %0, %2\n v_perm_b32 %2, 0, %5, %7\n v_pk_add_u16 %4, %1, %2\n v_pk_sub_i16 %1, %1, %2\n" \
          "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_pk_max_i16 %0, %3, %0\n v_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
          "v_pk_max_i16 %1, %4, %1\n v_and_b32 %0, 0xfe3ffe3f, %0\n v_and_b32 %1, 0xfe3ffe3f, %1\n v_or_b32 %0, 0x400000, %0\n v_or_b32 %1, 0x1c00180, %1\n" \

#define REGS_KERNEL \
          "v_perm_b32 v23, 0, v72, v34\n" \
          "v_pk_add_u16 v68, v20, v23\n" \
          "v_pk_sub_i16 v20, v20, v23\n" \
          "v_pk_sub_i16 v69, v22, v23\n" \
          "v_pk_add_u16 v22, v22, v23\n" \
          "v_mov_b32_dpp v20, v20 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
          "v_pk_max_i16 v20, v68, v20\n" \
          "v_mov_b32_dpp v22, v22 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
          "v_and_or_b32 v20, v20, s58, v28\n" \
          "v_pk_max_i16 v22, v69, v22\n" \
          "v_and_or_b32 v22, v22, s58, v28\n" \
          "v_perm_b32 v23, 0, v72, v34\n" \
          "v_pk_add_u16 v68, v20, v23\n" \
          "v_pk_sub_i16 v20, v20, v23\n" \
          "v_pk_sub_i16 v69, v22, v23\n" \
          "v_pk_add_u16 v22, v22, v23\n" \
          "v_mov_b32_dpp v20, v20 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
          "v_pk_max_i16 v20, v68, v20\n" \
          "v_mov_b32_dpp v22, v22 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
          "v_and_or_b32 v20, v20, s58, v28\n" \
          "v_pk_max_i16 v22, v69, v22\n" \
          "v_and_or_b32 v22, v22, s58, v28\n" \
          "v_perm_b32 v23, 0, v72, v34\n" \
          "v_pk_add_u16 v68, v20, v23\n" \
          "v_pk_sub_i16 v20, v20, v23\n" \
          "v_pk_sub_i16 v69, v22, v23\n" \
          "v_pk_add_u16 v22, v22, v23\n" \
          "v_mov_b32_dpp v20, v20 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
          "v_pk_max_i16 v20, v68, v20\n" \
          "v_mov_b32_dpp v22, v22 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
          "v_and_or_b32 v20, v20, s58, v28\n" \
          "v_pk_max_i16 v22, v69, v22\n" \
          "v_and_or_b32 v22, v22, s58, v28\n" \
          "v_perm_b32 v23, 0, v72, v34\n" \
          "v_pk_add_u16 v68, v20, v23\n" \
          "v_pk_sub_i16 v20, v20, v23\n" \
          "v_pk_sub_i16 v69, v22, v23\n" \
          "v_pk_add_u16 v22, v22, v23\n" \
          "v_mov_b32_dpp v20, v20 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
          "v_pk_max_i16 v20, v68, v20\n" \
          "v_mov_b32_dpp v22, v22 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
          "v_and_or_b32 v20, v20, s58, v28\n" \
          "v_pk_max_i16 v22, v69, v22\n" \
          "v_and_or_b32 v22, v22, s58, v28\n" \

#define REGS_SPREAD \
          "v_perm_b32 v22, 0, v25, v29\n" \
          "v_pk_add_u16 v23, v20, v22\n" \
          "v_pk_sub_i16 v20, v20, v22\n" \
          "v_pk_sub_i16 v27, v21, v22\n" \
          "v_pk_add_u16 v21, v21, v22\n" \
          "v_mov_b32_dpp v20, v20 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
          "v_pk_max_i16 v20, v23, v20\n" \
          "v_mov_b32_dpp v21, v21 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
          "v_and_or_b32 v20, v20, s58, v26\n" \
          "v_pk_max_i16 v21, v27, v21\n" \
          "v_and_or_b32 v21, v21, s58, v26\n" \
          "v_perm_b32 v22, 0, v25, v29\n" \
          "v_pk_add_u16 v23, v20, v22\n" \
          "v_pk_sub_i16 v20, v20, v22\n" \
          "v_pk_sub_i16 v27, v21, v22\n" \
          "v_pk_add_u16 v21, v21, v22\n" \
          "v_mov_b32_dpp v20, v20 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
          "v_pk_max_i16 v20, v23, v20\n" \
          "v_mov_b32_dpp v21, v21 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
          "v_and_or_b32 v20, v20, s58, v26\n" \
          "v_pk_max_i16 v21, v27, v21\n" \
          "v_and_or_b32 v21, v21, s58, v26\n" \
          "v_perm_b32 v22, 0, v25, v29\n" \
          "v_pk_add_u16 v23, v20, v22\n" \
          "v_pk_sub_i16 v20, v20, v22\n" \
          "v_pk_sub_i16 v27, v21, v22\n" \
          "v_pk_add_u16 v21, v21, v22\n" \
          "v_mov_b32_dpp v20, v20 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
          "v_pk_max_i16 v20, v23, v20\n" \
          "v_mov_b32_dpp v21, v21 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
          "v_and_or_b32 v20, v20, s58, v26\n" \
          "v_pk_max_i16 v21, v27, v21\n" \
          "v_and_or_b32 v21, v21, s58, v26\n" \
          "v_perm_b32 v22, 0, v25, v29\n" \
          "v_pk_add_u16 v23, v20, v22\n" \
          "v_pk_sub_i16 v20, v20, v22\n" \
          "v_pk_sub_i16 v27, v21, v22\n" \
          "v_pk_add_u16 v21, v21, v22\n" \
          "v_mov_b32_dpp v20, v20 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
          "v_pk_max_i16 v20, v23, v20\n" \
          "v_mov_b32_dpp v21, v21 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
          "v_and_or_b32 v20, v20, s58, v26\n" \
          "v_pk_max_i16 v21, v27, v21\n" \
          "v_and_or_b32 v21, v21, s58, v26\n" \

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

enum { K_PK_ADD, K_PK_MAX, K_PERM, K_AND_OR, K_DPP_ROR8, K_DPP_QUAD, K_ALIGNBIT, K_MAX_DPP, K_ADD_U32, K_ACS_MIX, K_ADD_E64, K_AND_E32, K_MAX_I16_E32, K_ADD_LIT, K_HALF_MIX, K_REAL, K_REAL_3K, K_REAL_24K, K_REGS_KERNEL, K_REGS_SPREAD, K_COUNT };
static const char *kNames[] = {"v_pk_add_i16", "v_pk_max_i16", "v_perm_b32", "v_and_or_b32", "v_mov_b32_dpp row_ror:8", "v_mov_b32_dpp quad_perm",
                               "v_alignbit_b32", "v_max_i32_dpp row_mirror", "v_add_u32", "acs mix (perm,pk_add,pk_sub,dpp,pk_max,and_or)",
                               "v_add_u32_e64 (VOP3 encoding, 8 bytes)", "v_and_b32 (VOP2, 4 bytes)", "v_max_i16 (VOP2, 4 bytes)", "v_add_u32 + 32-bit literal (8 bytes)", "alternating v_add_u32 (4 B) / v_pk_add_i16 (8 B)",
                               "4 trellis steps as compiled in viterbi3_kernel (SGPR mask, 4 DPP controls)",
                               "the same, loop body of 3 KB (8 copies)", "the same, straight-line loop body of 24 KB (64 copies)",
                               "4 steps with the VGPR numbers of the compiled kernel (sources share register banks)", "4 steps with VGPR numbers spread over the 4 banks"};

template <int KIND, bool DEP> __global__ __launch_bounds__(1024) void k(int iters, long long *cycles, int *sink, unsigned *hwid)
{
  int a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 ^ 0x55, a5 = a0 + 77, a6 = a0 * 11, a7 = a0 - 5;
  int b = 0x00010001 + (threadIdx.x & 3), c = 0x03020100;
  extern __shared__ int dyn_lds[];
  if (iters < 0) { iters = -iters; dyn_lds[threadIdx.x] = a0; for (int i = 0, n = (blockIdx.x * 37) & 255; i < n; i++) asm volatile("s_sleep 1"); }   // one-wave workgroups: start skew
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
    if (KIND == K_PK_ADD) {
      if (DEP) asm volatile(REP64("v_pk_add_i16 %0, %0, %1\n") : "+v"(a0) : "v"(b));
      else asm volatile(REP8("v_pk_add_i16 %0, %0, %8\n v_pk_add_i16 %1, %1, %8\n v_pk_add_i16 %2, %2, %8\n v_pk_add_i16 %3, %3, %8\n"
                             "v_pk_add_i16 %4, %4, %8\n v_pk_add_i16 %5, %5, %8\n v_pk_add_i16 %6, %6, %8\n v_pk_add_i16 %7, %7, %8\n")
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    } else if (KIND == K_PK_MAX) {
      if (DEP) asm volatile(REP64("v_pk_max_i16 %0, %0, %1\n") : "+v"(a0) : "v"(b));
      else asm volatile(REP8("v_pk_max_i16 %0, %0, %8\n v_pk_max_i16 %1, %1, %8\n v_pk_max_i16 %2, %2, %8\n v_pk_max_i16 %3, %3, %8\n"
                             "v_pk_max_i16 %4, %4, %8\n v_pk_max_i16 %5, %5, %8\n v_pk_max_i16 %6, %6, %8\n v_pk_max_i16 %7, %7, %8\n")
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    } else if (KIND == K_PERM) {
      if (DEP) asm volatile(REP64("v_perm_b32 %0, %0, %1, %2\n") : "+v"(a0) : "v"(b), "v"(c));
      else asm volatile(REP8("v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n"
                             "v_perm_b32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %8, %9\n v_perm_b32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %8, %9\n")
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    } else if (KIND == K_AND_OR) {
      if (DEP) asm volatile(REP64("v_and_or_b32 %0, %0, %1, %2\n") : "+v"(a0) : "v"(c), "v"(b));
      else asm volatile(REP8("v_and_or_b32 %0, %0, %9, %8\n v_and_or_b32 %1, %1, %9, %8\n v_and_or_b32 %2, %2, %9, %8\n v_and_or_b32 %3, %3, %9, %8\n"
                             "v_and_or_b32 %4, %4, %9, %8\n v_and_or_b32 %5, %5, %9, %8\n v_and_or_b32 %6, %6, %9, %8\n v_and_or_b32 %7, %7, %9, %8\n")
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    } else if (KIND == K_DPP_ROR8) {
      if (DEP) asm volatile(REP64("v_mov_b32_dpp %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n") : "+v"(a0));
      else asm volatile(REP8("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %2, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %4, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %6, %7 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n")
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if (KIND == K_DPP_QUAD) {
      if (DEP) asm volatile(REP64("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n") : "+v"(a0));
      else asm volatile(REP8("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %6, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if (KIND == K_ALIGNBIT) {
      if (DEP) asm volatile(REP64("v_alignbit_b32 %0, %0, %0, 16\n") : "+v"(a0));
      else asm volatile(REP8("v_alignbit_b32 %0, %0, %0, 16\n v_alignbit_b32 %1, %1, %1, 16\n v_alignbit_b32 %2, %2, %2, 16\n v_alignbit_b32 %3, %3, %3, 16\n"
                             "v_alignbit_b32 %4, %4, %4, 16\n v_alignbit_b32 %5, %5, %5, 16\n v_alignbit_b32 %6, %6, %6, 16\n v_alignbit_b32 %7, %7, %7, 16\n")
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if (KIND == K_MAX_DPP) {
      if (DEP) asm volatile(REP64("v_max_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n") : "+v"(a0));
      else asm volatile(REP8("v_max_i32_dpp %0, %1, %0 row_mirror row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %1, %2, %1 row_mirror row_mask:0xf bank_mask:0xf\n"
                             "v_max_i32_dpp %2, %3, %2 row_mirror row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %3, %4, %3 row_mirror row_mask:0xf bank_mask:0xf\n"
                             "v_max_i32_dpp %4, %5, %4 row_mirror row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %5, %6, %5 row_mirror row_mask:0xf bank_mask:0xf\n"
                             "v_max_i32_dpp %6, %7, %6 row_mirror row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %7, %0, %7 row_mirror row_mask:0xf bank_mask:0xf\n")
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if (KIND == K_ADD_U32) {
      if (DEP) asm volatile(REP64("v_add_u32 %0, %0, %1\n") : "+v"(a0) : "v"(b));
      else asm volatile(REP8("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                             "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n")
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    } else if (KIND == K_ADD_E64) {
      asm volatile(REP8("v_add_u32_e64 %0, %0, %8\n v_add_u32_e64 %1, %1, %8\n v_add_u32_e64 %2, %2, %8\n v_add_u32_e64 %3, %3, %8\n"
                        "v_add_u32_e64 %4, %4, %8\n v_add_u32_e64 %5, %5, %8\n v_add_u32_e64 %6, %6, %8\n v_add_u32_e64 %7, %7, %8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    } else if (KIND == K_AND_E32) {
      asm volatile(REP8("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n"
                        "v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    } else if (KIND == K_MAX_I16_E32) {
      asm volatile(REP8("v_max_i16 %0, %0, %8\n v_max_i16 %1, %1, %8\n v_max_i16 %2, %2, %8\n v_max_i16 %3, %3, %8\n"
                        "v_max_i16 %4, %4, %8\n v_max_i16 %5, %5, %8\n v_max_i16 %6, %6, %8\n v_max_i16 %7, %7, %8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    } else if (KIND == K_ADD_LIT) {
      asm volatile(REP8("v_add_u32 %0, 0x12345678, %0\n v_add_u32 %1, 0x12345678, %1\n v_add_u32 %2, 0x12345678, %2\n v_add_u32 %3, 0x12345678, %3\n"
                        "v_add_u32 %4, 0x12345678, %4\n v_add_u32 %5, 0x12345678, %5\n v_add_u32 %6, 0x12345678, %6\n v_add_u32 %7, 0x12345678, %7\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if (KIND == K_HALF_MIX) {
      asm volatile(REP8("v_add_u32 %0, %0, %8\n v_pk_add_i16 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_pk_add_i16 %3, %3, %8\n"
                        "v_add_u32 %4, %4, %8\n v_pk_add_i16 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_pk_add_i16 %7, %7, %8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    } else if (KIND == K_REAL) {
      // four steps (phases 2..5) copied from the compiled kernel: 44 VALU instructions; %0,%1 state, %2..%4 temporaries, %5 step word, %6 selector, %7 bias, %8 SGPR mask
      asm volatile(REAL_STEPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4) : "v"(b), "v"(c), "v"(a5), "s"(0xfefffeff));
    } else if (KIND == K_REGS_KERNEL) {
      asm volatile("s_mov_b32 s58, 0xfefffeff\n" REP8(REGS_KERNEL) ::: "v20", "v22", "v23", "v68", "v69", "v72", "v34", "v28", "s58");
    } else if (KIND == K_REGS_SPREAD) {
      asm volatile("s_mov_b32 s58, 0xfefffeff\n" REP8(REGS_SPREAD) ::: "v20", "v21", "v22", "v23", "v27", "v25", "v29", "v26", "s58");
    } else if (KIND == K_REAL_3K) {
      asm volatile(REP8(REAL_STEPS) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4) : "v"(b), "v"(c), "v"(a5), "s"(0xfefffeff));
    } else if (KIND == K_REAL_24K) {
      asm volatile(REP64(REAL_STEPS) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4) : "v"(b), "v"(c), "v"(a5), "s"(0xfefffeff));
    } else {
      // one trellis step of k_viterbi3.hpp on its two VGPRs (a0, a1): the dependency structure of the real loop.
      // DEP = as in the kernel (one decoder state per wave); !DEP = two independent decoders interleaved (a0,a1 | a4,a5)
      if (DEP) asm volatile(REP8(
          "v_perm_b32 %2, 0, %6, %7\n v_pk_add_i16 %3, %0, %2\n v_pk_sub_i16 %4, %0, %2\n v_pk_add_i16 %5, %1, %2\n v_pk_sub_i16 %2, %1, %2\n"
          "v_mov_b32_dpp %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n"
          "v_pk_max_i16 %3, %3, %4\n v_pk_max_i16 %5, %5, %2\n v_and_or_b32 %0, %3, %7, %6\n v_and_or_b32 %1, %5, %7, %6\n")
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(b), "v"(c));
      else asm volatile(REP8(
          "v_perm_b32 %2, 0, %8, %9\n v_perm_b32 %6, 0, %8, %9\n v_pk_add_i16 %3, %0, %2\n v_pk_add_i16 %7, %4, %6\n v_pk_sub_i16 %2, %0, %2\n v_pk_sub_i16 %6, %4, %6\n"
          "v_mov_b32_dpp %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n"
          "v_pk_max_i16 %3, %3, %2\n v_pk_max_i16 %7, %7, %6\n v_and_or_b32 %0, %3, %9, %8\n v_and_or_b32 %4, %7, %9, %8\n")
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    }
  }
  const long long t1 = clock64();
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    cycles[w] = t1 - t0;
    unsigned id, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    hwid[w] = (id & 0xffffu) | ((xcc & 0xfu) << 16);       // wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13] | xcc
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x7fffffff) *sink = a0;
}

// W waves per SIMD = WG workgroups per CU x WPW waves per workgroup and SIMD
template <int KIND, bool DEP> static void run(int WG, int WPW, int ncu, long long *d_cyc, int *d_sink, unsigned *d_hw, bool last)
{
  const int iters = KIND == K_REAL_24K ? 40 : KIND == K_REAL_3K ? 300 : 2000, W = WG * WPW, nwaves = ncu * 4 * W;
  const int per_trip = KIND == K_ACS_MIX ? (DEP ? 88 : 96) : KIND == K_REAL ? 47 : KIND == K_REAL_3K ? 376 : (KIND == K_REGS_KERNEL || KIND == K_REGS_SPREAD) ? 352 : KIND == K_REAL_24K ? 3008 : 64;
  std::vector<long long> h(nwaves); std::vector<unsigned> hw(nwaves);
  double best = 1e30, bestmax = 1e30; int occ_min = 0, occ_max = 0;
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL((k<KIND, DEP>), dim3(ncu * WG), dim3(256 * WPW), 0, 0, iters, d_cyc, d_sink, d_hw);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), d_cyc, sizeof(long long) * nwaves, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hw.data(), d_hw, sizeof(unsigned) * nwaves, hipMemcpyDeviceToHost);
    std::vector<int> cnt(1 << 18, 0);                      // waves per (xcc, se, sh, cu, simd)
    for (unsigned v : hw) cnt[((v >> 16) << 14) | (((v >> 8) & 0xff) << 2) | ((v >> 4) & 3)]++;
    occ_min = 1 << 30; occ_max = 0;
    for (int c : cnt) if (c) { occ_min = std::min(occ_min, c); occ_max = std::max(occ_max, c); }
    std::sort(h.begin(), h.end());
    if ((double)h[nwaves / 2] < best) { best = (double)h[nwaves / 2]; bestmax = (double)h[nwaves - 1]; }
  }
  const double per_inst_wave = best / ((double)iters * per_trip);
  printf("  {\"inst\": \"%s\", \"chain\": \"%s\", \"waves_per_simd\": %d, \"waves_per_simd_observed\": [%d, %d], \"cycles_per_inst_one_wave\": %.3f, \"slowest_wave\": %.3f, \"simd_cycles_per_inst\": %.3f}%s\n",
         kNames[KIND], DEP ? "dependent" : (KIND == K_ACS_MIX ? "2 decoders interleaved" : "8 independent"), W, occ_min, occ_max, per_inst_wave,
         bestmax / ((double)iters * per_trip), bestmax / ((double)iters * per_trip) / W, last ? "" : ",");
}

// the same with ONE-WAVE workgroups (as viterbi3_kernel runs): grid = cus x 4 W, residency capped at 4 W per CU by dynamic LDS
template <int KIND> static void solo(int W, int ncu, long long *d_cyc, int *d_sink, unsigned *d_hw, bool last)
{
  const int iters = KIND == K_REAL_24K ? 40 : (KIND == K_REAL_3K || KIND == K_REGS_KERNEL || KIND == K_REGS_SPREAD) ? 300 : 2000, nwaves = ncu * 4 * W, per_trip = KIND == K_REAL ? 47 : KIND == K_REAL_3K ? 376 : (KIND == K_REGS_KERNEL || KIND == K_REGS_SPREAD) ? 352 : KIND == K_REAL_24K ? 3008 : 64;
  const size_t lds = (size_t)(160 * 1024 / (4 * W) - 512);
  (void)hipFuncSetAttribute((const void *)k<KIND, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  std::vector<long long> h(nwaves); std::vector<unsigned> hw(nwaves);
  double best = 1e30, bestmax = 0; int occ_min = 0, occ_max = 0; double occ_avg = 0;
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL((k<KIND, false>), dim3(nwaves), dim3(64), lds, 0, -iters, d_cyc, d_sink, d_hw);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), d_cyc, sizeof(long long) * nwaves, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hw.data(), d_hw, sizeof(unsigned) * nwaves, hipMemcpyDeviceToHost);
    std::vector<int> cnt(1 << 18, 0);
    for (unsigned v : hw) cnt[((v >> 16) << 14) | (((v >> 8) & 0xff) << 2) | ((v >> 4) & 3)]++;
    occ_min = 1 << 30; occ_max = 0; int nsimd = 0;
    for (int c : cnt) if (c) { occ_min = std::min(occ_min, c); occ_max = std::max(occ_max, c); nsimd++; }
    occ_avg = (double)nwaves / nsimd;
    double sum = 0; for (long long v : h) sum += (double)v;
    std::sort(h.begin(), h.end());
    if (sum / nwaves < best) { best = sum / nwaves; bestmax = (double)h[nwaves - 1]; }
  }
  const double per_inst_wave = best / ((double)iters * per_trip);
  printf("  {\"inst\": \"%s\", \"chain\": \"one-wave workgroups, skewed starts\", \"waves_per_simd\": %d, \"waves_per_simd_observed\": [%d, %d], \"simds_used_avg_waves\": %.2f, \"cycles_per_inst_one_wave\": %.3f, \"slowest_wave\": %.3f, \"simd_cycles_per_inst\": %.3f}%s\n",
         kNames[KIND], W, occ_min, occ_max, occ_avg, per_inst_wave, bestmax / ((double)iters * per_trip), bestmax / ((double)iters * per_trip) / occ_max, last ? "" : ",");
}

template <int KIND> static void both(int ncu, long long *d_cyc, int *d_sink, unsigned *d_hw, bool last, bool dep_too = true)
{
  const int cfg[][2] = {{1, 1}, {1, 2}, {1, 3}, {1, 4}, {2, 3}, {2, 4}};
  if (dep_too) for (auto &c : cfg) run<KIND, true>(c[0], c[1], ncu, d_cyc, d_sink, d_hw, false);
  for (int i = 0; i < 6; i++) run<KIND, false>(cfg[i][0], cfg[i][1], ncu, d_cyc, d_sink, d_hw, last && i == 5);
}

int main()
{
  int ncu = 256; (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  int clk = 0; (void)hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  long long *d_cyc; int *d_sink; unsigned *d_hw;
  (void)hipMalloc((void **)&d_cyc, sizeof(long long) * ncu * 32); (void)hipMalloc((void **)&d_sink, 4); (void)hipMalloc((void **)&d_hw, sizeof(unsigned) * ncu * 32);
  hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
  printf("{\"device\": \"%s\", \"gcn_arch\": \"%s\", \"cus\": %d, \"clock_khz\": %d, \"timer\": \"s_memtime via clock64()\",\n \"note\": \"simd_cycles_per_inst = slowest wave's cycles / (instructions x waves on its SIMD) = SIMD issue time per wave64 instruction (waves do not share a SIMD evenly: use the slowest, not the mean). waves_per_simd_observed = [min, max] waves per SIMD from HW_ID\",\n \"results\": [\n",
         pr.name, pr.gcnArchName, ncu, clk);
  both<K_ADD_U32>(ncu, d_cyc, d_sink, d_hw, false); both<K_ADD_E64>(ncu, d_cyc, d_sink, d_hw, false, false); both<K_ADD_LIT>(ncu, d_cyc, d_sink, d_hw, false, false);
  both<K_AND_E32>(ncu, d_cyc, d_sink, d_hw, false, false); both<K_MAX_I16_E32>(ncu, d_cyc, d_sink, d_hw, false, false); both<K_HALF_MIX>(ncu, d_cyc, d_sink, d_hw, false, false);
  both<K_PK_ADD>(ncu, d_cyc, d_sink, d_hw, false); both<K_PK_MAX>(ncu, d_cyc, d_sink, d_hw, false, false); both<K_PERM>(ncu, d_cyc, d_sink, d_hw, false);
  both<K_AND_OR>(ncu, d_cyc, d_sink, d_hw, false, false); both<K_ALIGNBIT>(ncu, d_cyc, d_sink, d_hw, false, false); both<K_DPP_ROR8>(ncu, d_cyc, d_sink, d_hw, false); both<K_DPP_QUAD>(ncu, d_cyc, d_sink, d_hw, false, false);
  both<K_MAX_DPP>(ncu, d_cyc, d_sink, d_hw, false, false); both<K_ACS_MIX>(ncu, d_cyc, d_sink, d_hw, false); both<K_REAL>(ncu, d_cyc, d_sink, d_hw, false, false);
  for (int W : {1, 2, 3, 4, 6, 8}) solo<K_REAL>(W, ncu, d_cyc, d_sink, d_hw, false);
  for (int W : {1, 2, 4, 8}) solo<K_REAL_3K>(W, ncu, d_cyc, d_sink, d_hw, false);
  for (int W : {1, 2, 4, 8}) solo<K_REAL_24K>(W, ncu, d_cyc, d_sink, d_hw, false);
  for (int W : {1, 2, 4, 8}) solo<K_REGS_KERNEL>(W, ncu, d_cyc, d_sink, d_hw, false);
  for (int W : {1, 2, 4, 8}) solo<K_REGS_SPREAD>(W, ncu, d_cyc, d_sink, d_hw, false);
  for (int W : {1, 2, 3, 4, 6, 8}) solo<K_PK_ADD>(W, ncu, d_cyc, d_sink, d_hw, false);
  for (int W : {1, 2, 3, 4, 6, 8}) solo<K_ADD_U32>(W, ncu, d_cyc, d_sink, d_hw, W == 8);
  printf("]}\n");
  return 0;
}
