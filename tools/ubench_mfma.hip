// ubench_mfma.hip -- can the matrix pipe take work off the VALU port for free?  (the three radix-16 passes of the 8k symbol kernel as
// 16x16 complex DFTs on v_mfma_f32_16x16x4_f32.)  With steps in flight the whole step is bound by VALU issue (95 % busy), the matrix pipe idle.  Questions:
//   (1) what does a v_mfma_f32_16x16x4_f32 cost on its own (cycles per instruction and SIMD; exact f32, 1024 MACs);
//   (2) a wavefront that runs the Viterbi kernel's VALU mix (v_pk_add / v_pk_sub / v_pk_max / v_perm / DPP) while ANOTHER wavefront of the same SIMD issues
//       MFMAs back to back: does the VALU wavefront keep its 4.0-4.5 cycles per instruction?
//   (3) the same inside ONE wavefront: an MFMA every n VALU instructions (independent registers).
// Method as ubench_valu.hip: one workgroup per CU with W wavefronts per SIMD, s_memtime around the loop, the slowest wavefront of each kind counts.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/ubench_mfma tools/ubench_mfma.hip ; run: ./tools/ubench_mfma > profiles/rNN_ubench_mfma.json
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float f4 __attribute__((ext_vector_type(4)));

#define VALU8 \
  "v_perm_b32 %2, %8, %8, %9\n v_pk_add_i16 %4, %0, %2\n v_pk_sub_i16 %0, %0, %2\n v_pk_add_i16 %5, %1, %2\n v_pk_sub_i16 %1, %1, %2\n" \
  "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_pk_max_i16 %0, %4, %0\n v_pk_max_i16 %1, %5, %1\n"

// role 0: VALU mix only; role 1: MFMA only (4 independent accumulators); role 2: per 8 VALU instructions `mfma_per` MFMAs in the same wavefront
__global__ __launch_bounds__(1024) void k(int iters, int valu_waves_per_simd, int mfma_waves_per_simd, int mixed, long long *cycles, float *sink)
{
  const int wave = threadIdx.x >> 6;                         // waves are dealt round-robin to the 4 SIMDs: wave >> 2 = slot on its SIMD
  const int slot = wave >> 2;
  const bool is_mfma = !mixed && slot >= valu_waves_per_simd;
  int a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0, b = 0x00010001 + (threadIdx.x & 3), c = 0x03020100;
  f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  const float x = 1.0f + (threadIdx.x & 7) * 0.125f, y = 0.5f;
  __syncthreads();
  const long long t0 = clock64();
  if (mixed) {
    for (int it = 0; it < iters; it++) {
      asm volatile(VALU8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c0, 0, 0, 0);
      asm volatile(VALU8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
      if (mixed >= 2) c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c1, 0, 0, 0);
      asm volatile(VALU8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
      if (mixed >= 4) c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c2, 0, 0, 0);
      asm volatile(VALU8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
      if (mixed >= 4) c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c3, 0, 0, 0);
    }
  } else if (is_mfma) {
    for (int it = 0; it < iters; it++) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c3, 0, 0, 0);
    }
  } else {
    for (int it = 0; it < iters; it++) {
      asm volatile(VALU8 VALU8 VALU8 VALU8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    }
  }
  const long long t1 = clock64();
  if ((threadIdx.x & 63) == 0) cycles[(size_t)blockIdx.x * 16 + wave] = t1 - t0;
  sink[blockIdx.x * 1024 + threadIdx.x] = (float)(a0 + a1 + a4 + a5) + c0.x + c1.y + c2.z + c3.w;
}

int main()
{
  int ncu = 256; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  long long *d_cyc; float *d_sink; hipMalloc(&d_cyc, sizeof(long long) * ncu * 16); hipMalloc(&d_sink, sizeof(float) * ncu * 1024);
  const int iters = 20000;
  // clock64() counts at 100 MHz on gfx9 (s_memtime); convert with the shader clock measured by a pure VALU loop of known cost?  Report raw ratios instead:
  // everything below is relative to the VALU-only run of the same occupancy
  struct Case { const char *name; int vw, mw, mixed; };
  const Case cases[] = {
    {"VALU mix alone, 1 wave per SIMD", 1, 0, 0}, {"VALU mix alone, 2 waves per SIMD", 2, 0, 0}, {"MFMA alone, 1 wave per SIMD", 0, 1, 0}, {"MFMA alone, 2 waves per SIMD", 0, 2, 0},
    {"1 VALU wave + 1 MFMA wave per SIMD", 1, 1, 0}, {"2 VALU waves + 1 MFMA wave per SIMD", 2, 1, 0}, {"2 VALU waves + 2 MFMA waves per SIMD", 2, 2, 0},
    {"one wave: 1 MFMA per 32 VALU (x2 waves per SIMD)", 2, 0, 1}, {"one wave: 2 MFMA per 32 VALU (x2 waves per SIMD)", 2, 0, 2}, {"one wave: 4 MFMA per 32 VALU (x2 waves per SIMD)", 2, 0, 4}};
  printf("{\"note\": \"ticks = s_memtime (100 MHz) per loop iteration of the slowest wavefront of the kind; a VALU iteration = 32 instructions of the Viterbi mix, an MFMA iteration = 4 v_mfma_f32_16x16x4_f32, a mixed iteration = 32 VALU + n MFMA\", \"iters\": %d, \"cases\": [\n", iters);
  for (size_t ci = 0; ci < sizeof cases / sizeof cases[0]; ci++) {
    const Case &c = cases[ci];
    const int waves = 4 * (c.vw + c.mw);
    hipMemset(d_cyc, 0, sizeof(long long) * ncu * 16);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k, dim3(ncu), dim3(64 * waves), 0, 0, iters, c.vw, c.mw, c.mixed, d_cyc, d_sink);
    hipDeviceSynchronize();
    std::vector<long long> h(ncu * 16); hipMemcpy(h.data(), d_cyc, sizeof(long long) * ncu * 16, hipMemcpyDeviceToHost);
    long long mv = 0, mm = 0;
    for (int b = 0; b < ncu; b++) for (int w = 0; w < waves; w++) { const bool mf = !c.mixed && (w >> 2) >= c.vw; long long v = h[(size_t)b * 16 + w]; if (mf) mm = std::max(mm, v); else mv = std::max(mv, v); }
    printf("  {\"case\": \"%s\", \"valu_ticks_per_iter\": %.4f, \"mfma_ticks_per_iter\": %.4f}%s\n", c.name, c.vw || c.mixed ? (double)mv / iters : 0.0, c.mw ? (double)mm / iters : 0.0,
           ci + 1 < sizeof cases / sizeof cases[0] ? "," : "");
  }
  printf("]}\n");
  return 0;
}
