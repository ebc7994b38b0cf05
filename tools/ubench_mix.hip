// ubench_mix.hip -- does a full-rate VOP2 instruction (v_add_u32, v_and_b32: 2.2 cycles per wave64 instruction with >= 2 waves per
// SIMD, profiles/r02_ubench_valu.json) keep its rate when it sits between half-rate ones (v_pk_*, v_perm, DPP: 4 cycles)?  The strict
// alternation measured in ubench_valu.hip costs 4 cycles per instruction for BOTH.  This asks the same for runs of 2 and 4 and for the
// instruction sequence a "light-op" add-compare-select step would use (k_viterbi3.hpp).
// Method as ubench_valu.hip: one workgroup per CU, W waves per SIMD, s_memtime around the loop, slowest wave / W.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/ubench_mix tools/ubench_mix.hip ; run: ./tools/ubench_mix > profiles/rNN_ubench_mix.json
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP4(x) x x x x
#define REP8(x) x x x x x x x x

#define A(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define P(n) "v_pk_add_i16 %" #n ", %" #n ", %8\n"
#define D(n) "v_add_u32_dpp %" #n ", %" #n ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define M(n) "v_pk_max_u16 %" #n ", %" #n ", %8\n"
#define N(n) "v_and_b32 %" #n ", %" #n ", %8\n"
#define R(n) "v_perm_b32 %" #n ", %" #n ", %8, %9\n"
#define O(n) "v_and_or_b32 %" #n ", %" #n ", %9, %8\n"
#define V(n) "v_mov_b32_dpp %" #n ", %" #n " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"

enum { K_ALT, K_RUN2, K_RUN4, K_ADD_DPP, K_LIGHT, K_NOW, K_LIGHT_SCHED, K_COUNT };
static const char *kNames[] = {
  "alternating v_add_u32 / v_pk_add_i16", "runs of 2: add add pk pk", "runs of 4: 4 x add, 4 x pk", "v_add_u32_dpp quad_perm",
  "light step: perm sub add add add_dpp add_dpp pk_max pk_max and and (10)", "current step: perm pk_add pk_sub pk_add pk_sub dpp dpp pk_max pk_max and_or and_or (11)",
  "light step, full-rate instructions adjacent: perm add_dpp add_dpp pk_max pk_max | sub add add and and (10)"};
static const int kPer[] = {64, 64, 64, 64, 80, 88, 80};

template <int KIND> __global__ __launch_bounds__(1024) void k(int iters, long long *cycles, int *sink)
{
  int a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 ^ 0x55, a5 = a0 + 77, a6 = a0 * 11, a7 = a0 - 5;
  int b = 0x00010001 + (threadIdx.x & 3), c = 0x03020100;
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
    if (KIND == K_ALT) asm volatile(REP8(A(0) P(1) A(2) P(3) A(4) P(5) A(6) P(7)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    else if (KIND == K_RUN2) asm volatile(REP8(A(0) A(1) P(2) P(3) A(4) A(5) P(6) P(7)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    else if (KIND == K_RUN4) asm volatile(REP8(A(0) A(1) A(2) A(3) P(4) P(5) P(6) P(7)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    else if (KIND == K_ADD_DPP) asm volatile(REP8(D(0) D(1) D(2) D(3) D(4) D(5) D(6) D(7)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    // the step's data flow on two state registers (a0, a1) and temporaries: E = perm(W); F = K - E; X = v + E; Y = dpp(v) + F; v = max(X, Y) & mask
    else if (KIND == K_LIGHT) asm volatile(REP8(
        "v_perm_b32 %2, %8, %8, %9\n v_sub_u32 %3, %9, %2\n v_add_u32 %4, %0, %2\n v_add_u32 %5, %1, %2\n"
        "v_add_u32_dpp %0, %0, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32_dpp %1, %1, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_pk_max_u16 %0, %4, %0\n v_pk_max_u16 %1, %5, %1\n v_and_b32 %0, %0, %9\n v_and_b32 %1, %1, %9\n")
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    else if (KIND == K_NOW) asm volatile(REP8(
        "v_perm_b32 %2, %8, %8, %9\n v_pk_add_i16 %4, %0, %2\n v_pk_sub_i16 %0, %0, %2\n v_pk_add_i16 %5, %1, %2\n v_pk_sub_i16 %1, %1, %2\n"
        "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_pk_max_i16 %0, %4, %0\n v_pk_max_i16 %1, %5, %1\n v_and_or_b32 %0, %0, %9, %8\n v_and_or_b32 %1, %1, %9, %8\n")
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    // two decoders' worth (a0,a1 | a6,a7) is NOT what the kernel has; this keeps one decoder and only reorders: the and of step n next to sub/add of step n+1
    else asm volatile(REP8(
        "v_perm_b32 %2, %8, %8, %9\n"
        "v_add_u32_dpp %6, %0, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32_dpp %7, %1, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_pk_max_u16 %0, %4, %6\n v_pk_max_u16 %1, %5, %7\n"
        "v_sub_u32 %3, %9, %2\n v_and_b32 %0, %0, %9\n v_and_b32 %1, %1, %9\n v_add_u32 %4, %0, %2\n v_add_u32 %5, %1, %2\n")
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
  }
  const long long t1 = clock64();
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x7fffffff) *sink = a0;
}

template <int KIND> static void run(int ncu, long long *d_cyc, int *d_sink, bool last)
{
  const int cfg[][2] = {{1, 1}, {1, 2}, {1, 3}, {1, 4}, {2, 4}};
  for (int ci = 0; ci < 5; ci++) {
    const int WG = cfg[ci][0], WPW = cfg[ci][1], W = WG * WPW, nwaves = ncu * 4 * W, iters = 2000;
    std::vector<long long> h(nwaves);
    double best = 1e30, bestmax = 1e30;
    for (int rep = 0; rep < 3; rep++) {
      hipLaunchKernelGGL((k<KIND>), dim3(ncu * WG), dim3(256 * WPW), 0, 0, iters, d_cyc, d_sink);
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(h.data(), d_cyc, sizeof(long long) * nwaves, hipMemcpyDeviceToHost);
      std::sort(h.begin(), h.end());
      if ((double)h[nwaves - 1] < bestmax) { best = (double)h[nwaves / 2]; bestmax = (double)h[nwaves - 1]; }
    }
    const double n = (double)iters * kPer[KIND];
    printf("  {\"seq\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_inst_median_wave\": %.3f, \"slowest_wave\": %.3f, \"simd_cycles_per_inst\": %.3f, \"simd_cycles_per_group\": %.2f}%s\n",
           kNames[KIND], W, best / n, bestmax / n, bestmax / n / W, bestmax / n / W * (kPer[KIND] / 8), (last && ci == 4) ? "" : ",");
  }
}

int main()
{
  int ncu = 256; (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  long long *d_cyc; int *d_sink;
  (void)hipMalloc((void **)&d_cyc, sizeof(long long) * ncu * 32); (void)hipMalloc((void **)&d_sink, 4);
  hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
  printf("{\"gcn_arch\": \"%s\", \"cus\": %d, \"note\": \"simd_cycles_per_group = SIMD issue cycles for one copy of the sequence (8 instructions for the first four rows, one trellis step on two registers for the rest)\",\n \"results\": [\n", pr.gcnArchName, ncu);
  run<K_ALT>(ncu, d_cyc, d_sink, false); run<K_RUN2>(ncu, d_cyc, d_sink, false); run<K_RUN4>(ncu, d_cyc, d_sink, false); run<K_ADD_DPP>(ncu, d_cyc, d_sink, false);
  run<K_LIGHT>(ncu, d_cyc, d_sink, false); run<K_NOW>(ncu, d_cyc, d_sink, false); run<K_LIGHT_SCHED>(ncu, d_cyc, d_sink, true);
  printf("]}\n");
  return 0;
}
