// ubench_transpose.hip -- VERDICT r04 item 5, second proposal: "do the first two radix-16 passes of symbol8k_kernel wave-locally (DPP) so that only the last pass
// needs a workgroup barrier".  With the index split n = tid + 512 i (pass 1 on the registers), pass 2 over the LOW four bits of tid and pass 3 over the other five,
// the hand-over between pass 1 and pass 2 is a 16 x 16 transposition of (lane of a DPP row) x (register): lane l holds Y[k1 = 0..15][l] and must hold Y[l][0..15].
// That replaces ONE of the kernel's LDS round trips (16 ds_write_b64 + barrier + 16 ds_read_b64 + the barrier that protects the image from the next writers) by
// register traffic.  What does each cost IN SITU -- between two radix-16 passes (the kernel's own s8_dft16 + s8_twiddle16), 512-thread workgroups at 128 VGPRs, two
// per CU, so that the LDS latency and the barrier can hide behind the other workgroup's arithmetic exactly as they do in the kernel?
//   mode 0: pass arithmetic only                      mode 1: + the LDS exchange with the kernel's swizzled addresses and its two barriers
//   mode 2: + the transposition by DPP: four exchange stages (lane ^ 8, ^ 4, ^ 2, ^ 1); a stage sends one register of each of 8 pairs to the partner lane:
//           per 32-bit word one select (what to send), the DPP move(s), two selects (where it lands)
// Output: microseconds per 1000 iterations of the whole grid, cycles per iteration and workgroup pair (one CU), and the static VALU / LDS instruction counts of the
// loops (llvm-objdump, see tools/README.md).  The transposition is checked on the device first (mode 3).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I gr_dvbt_amd/csrc -I include -o tools/ubench_transpose tools/ubench_transpose.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "k_symbol8k.hpp"

using namespace dvbt;

template <int CTRL> __device__ __forceinline__ int t_dpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
// the word of lane (l ^ B) of the same DPP row
template <int B> __device__ __forceinline__ int t_xor(int v)
{
  if (B == 1) return t_dpp<0xB1>(v);                 // quad_perm [1,0,3,2]
  if (B == 2) return t_dpp<0x4E>(v);                 // quad_perm [2,3,0,1]
  if (B == 8) return t_dpp<0x128>(v);                // row_ror:8
  return t_dpp<0x1B>(t_dpp<0x141>(v));               // l ^ 4 = (l ^ 7) ^ 3: row_half_mirror, then quad_perm [3,2,1,0] (no single DPP control exchanges across quads)
}
template <int B> __device__ __forceinline__ void t_stage(v2f (&a)[16], int lane)
{
  // (opaque to the optimiser: left alone it composes the four stages' selects into one select tree per element over the lane number -- ~900 compares)
  int bit = lane & B;
  asm volatile("" : "+v"(bit));
#pragma unroll
  for (int r = 0; r < 16; r++) asm volatile("" : "+v"(a[r]));
  const bool hi = bit != 0;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    if (r & B) continue;
    const v2f send = hi ? a[r] : a[r | B];
    const unsigned long long u = __builtin_bit_cast(unsigned long long, send);           // both words travel (written on the vector's elements the compiler moved only .x)
    const unsigned lo = (unsigned)t_xor<B>((int)(unsigned)u), up = (unsigned)t_xor<B>((int)(unsigned)(u >> 32));
    const v2f recv = __builtin_bit_cast(v2f, (unsigned long long)lo | ((unsigned long long)up << 32));
    if (hi) a[r] = recv; else a[r | B] = recv;
  }
}
// a[r] of lane l  <->  a[l] of lane r, inside every DPP row of 16 lanes
__device__ __forceinline__ void t_transpose(v2f (&a)[16], int lane)
{
  t_stage<8>(a, lane); t_stage<4>(a, lane); t_stage<2>(a, lane); t_stage<1>(a, lane);
}

template <int MODE> __global__ __launch_bounds__(512, 4) void k(int iters, const float2 *__restrict__ tw, float *__restrict__ sink, int *__restrict__ bad)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  v2f *x = reinterpret_cast<v2f *>(smem_raw);
  const int tid0 = threadIdx.x;
  v2f a[16];
#pragma unroll
  for (int i = 0; i < 16; i++) a[i] = (v2f){(float)((tid0 & 15) * 16 + i), (float)(tid0 >> 4)};
  if (MODE == 3) {                                     // self-check: lane l, register r must end up with lane r's register l
    t_transpose(a, tid0 & 15);
    int wrong = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) wrong += !(a[i].x == (float)(i * 16 + (tid0 & 15)) && a[i].y == (float)(tid0 >> 4));
    if (wrong) atomicAdd(bad, wrong);
    return;
  }
  const v2f w1A = s8_v(tw[tid0]);
  for (int it = 0; it < iters; it++) {
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const S8Roots R = s8_roots();
    s8_dft16(a, R);
    { v2f wA = w1A; asm volatile("" : "+v"(wA), "+v"(a[15])); s8_twiddle16(a, wA); }
    if (MODE == 1) {
      __syncthreads();                                 // the previous readers of the image are done
      const int b0 = tid ^ ((tid >> 5) & 15);
#pragma unroll
      for (int k1 = 0; k1 < 16; k1++) x[k1 * 512 + (b0 ^ ((k1 & 1) << 4))] = a[k1];
      __syncthreads();
      const int k1 = tid >> 5, mm = (tid & 31) ^ ((k1 & 1) << 4), rb = k1 * 512;
#pragma unroll
      for (int m1 = 0; m1 < 16; m1++) a[m1] = x[rb + 32 * m1 + (mm ^ m1)];
    } else if (MODE == 2) {
      t_transpose(a, tid & 15);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; i++) s += a[i].x + a[i].y;
  sink[(size_t)blockIdx.x * 512 + tid0] = s;
}

template <int MODE> static double run(int grid, int iters, const float2 *d_tw, float *d_sink, int *d_bad)
{
  (void)hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 8192);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  double best = 1e30;
  for (int rep = 0; rep < 4; rep++) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(512), 65536 + 8192, 0, iters, d_tw, d_sink, d_bad);   // the kernel's LDS footprint in every mode: two workgroups per CU
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  return best;
}

int main()
{
  int ncu = 256; (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  int khz = 0; (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
  const int grid = 2 * ncu, iters = 4000;
  std::vector<float2> tw(512);
  for (int i = 0; i < 512; i++) tw[i] = make_float2(cosf(-6.283185307f * i / 8192.f), sinf(-6.283185307f * i / 8192.f));
  float2 *d_tw; float *d_sink; int *d_bad;
  (void)hipMalloc((void **)&d_tw, sizeof(float2) * 512); (void)hipMalloc((void **)&d_sink, sizeof(float) * grid * 512); (void)hipMalloc((void **)&d_bad, 4);
  (void)hipMemcpy(d_tw, tw.data(), sizeof(float2) * 512, hipMemcpyHostToDevice); (void)hipMemset(d_bad, 0, 4);
  hipLaunchKernelGGL((k<3>), dim3(4), dim3(512), 65536 + 8192, 0, 1, d_tw, d_sink, d_bad);
  int bad = -1; (void)hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost);
  const double t0 = run<0>(grid, iters, d_tw, d_sink, d_bad), t1 = run<1>(grid, iters, d_tw, d_sink, d_bad), t2 = run<2>(grid, iters, d_tw, d_sink, d_bad);
  const double mhz = khz > 0 ? khz / 1e3 : 2400.0;
  auto cyc = [&](double ms) { return ms * 1e-3 / iters * mhz * 1e6; };                                 // cycles per iteration of a CU's two workgroups
  printf("{\"what\": \"one radix-16 pass of symbol8k_kernel (s8_dft16 + s8_twiddle16) followed by the hand-over to the next pass, 512-thread workgroups, 128 VGPRs, two per CU (%d workgroups), %d iterations\",\n"
         " \"transpose_self_check_wrong_words\": %d, \"clock_mhz\": %.0f,\n"
         " \"ms\": {\"pass_only\": %.3f, \"pass_plus_lds_exchange_two_barriers\": %.3f, \"pass_plus_dpp_transpose\": %.3f},\n"
         " \"cycles_per_iteration_and_cu\": {\"pass_only\": %.0f, \"pass_plus_lds_exchange_two_barriers\": %.0f, \"pass_plus_dpp_transpose\": %.0f},\n"
         " \"hand_over_cycles_per_cu\": {\"lds_exchange\": %.0f, \"dpp_transpose\": %.0f}}\n",
         grid, iters, bad, mhz, t0, t1, t2, cyc(t0), cyc(t1), cyc(t2), cyc(t1 - t0), cyc(t2 - t0));
  return bad != 0;
}
