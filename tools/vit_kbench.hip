// vit_kbench.hip -- viterbi3_kernel alone on synthetic input: launch time vs chunk size / occupancy build, without the rest of
// the chain.  Experiment tool (not part of the product library).  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DV3_EXP=bits]
// -I gr_dvbt_amd/csrc -o tools/vit_kbench tools/vit_kbench.hip ; run: tools/vit_kbench [symbols=17475] [chunk_bytes...]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "k_viterbi3.hpp"
using namespace dvbt;

// one wavefront samples the shader clock (s_memtime) against the constant 100 MHz counter (s_memrealtime) while the decoder runs
__global__ void clock_probe(unsigned long long *o, long long spin)
{
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  while ((long long)(wall_clock64() - w0) < spin) __builtin_amdgcn_s_sleep(8);
  o[0] = clock64() - c0; o[1] = wall_clock64() - w0;
}

int main(int argc, char **argv)
{
  const int nsym = argc > 1 ? atoi(argv[1]) : 17475;
  Dims d = make_dims(2, 0, 4, 0, 1);                            // 8k QAM64 7/8
  const long long nin = (long long)nsym * d.payload / 1024 * 1024, steps = nin / 1024 * 5376, nout = steps / 8 - d.ntb;
  std::vector<uint8_t> h(nin);
  unsigned x = 12345; for (auto &b : h) { x = x * 1664525u + 1013904223u; b = (x >> 24) & 63; }
  uint8_t *din, *dout; (void)hipMalloc((void **)&din, nin + 64); (void)hipMalloc((void **)&dout, nout + 64);
  (void)hipMemcpy(din, h.data(), nin, hipMemcpyHostToDevice);
  int ncu = 256; (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  std::vector<int> chunks;
  for (int i = 2; i < argc; i++) chunks.push_back(atoi(argv[i]));
  if (chunks.empty()) {
    for (int wpc : {8, 12, 16}) for (int rounds : {1, 2, 3}) chunks.push_back((int)((nout + (long long)ncu * wpc * 4 * rounds - 1) / ((long long)ncu * wpc * 4 * rounds)));
  }
#if V3_EXP & 16
  unsigned long long *dbg = nullptr;
  (void)hipMalloc((void **)&dbg, 8 * 3 * 65536); (void)hipMemcpyToSymbol(HIP_SYMBOL(v3_dbg), &dbg, sizeof dbg);
#endif
  unsigned long long sum0 = 0;
  for (int cb : chunks) {
    VitParams vp = make_vit_params(d, 768, cb);
    const long long nchunks = (nout + cb - 1) / cb; const dim3 grid((unsigned)((nchunks + 4 * V3_WGW - 1) / (4 * V3_WGW)));
    float best = 1e30f; double mhz = 0;
    {
      static hipStream_t s2 = nullptr; static unsigned long long *pr = nullptr;
      if (!s2) { (void)hipStreamCreate(&s2); (void)hipMalloc((void **)&pr, 16); }
      hipLaunchKernelGGL(viterbi3_kernel<24>, grid, dim3(64 * V3_WGW), 0, 0, (const uint8_t *)din, dout, (const RxState *)nullptr, steps, vp, V3Aux{}, 0ll, 0ll);
      hipLaunchKernelGGL(clock_probe, dim3(1), dim3(64), 0, s2, pr, 200000ll);       // 2 ms at 100 MHz, inside the decoder's launch
      (void)hipDeviceSynchronize();
      unsigned long long h2[2]; (void)hipMemcpy(h2, pr, 16, hipMemcpyDeviceToHost);
      mhz = (double)h2[0] / (double)h2[1] * 100.0;
    }
    for (int rep = 0; rep < 4; rep++) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(viterbi3_kernel<24>, grid, dim3(64 * V3_WGW), 0, 0, (const uint8_t *)din, dout, (const RxState *)nullptr, steps, vp, V3Aux{}, 0ll, 0ll);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
#if V3_EXP & 16
    {   // per-wavefront lifetimes (100 MHz ticks) and residency per SIMD
      const size_t nw = (size_t)grid.x * V3_WGW;
      (void)hipMemset(dbg, 0, 8 * 3 * nw);
      hipLaunchKernelGGL(viterbi3_kernel<24>, grid, dim3(64 * V3_WGW), 0, 0, (const uint8_t *)din, dout, (const RxState *)nullptr, steps, vp, V3Aux{}, 0ll, 0ll);
      (void)hipDeviceSynchronize();
      std::vector<unsigned long long> hd(3 * nw); (void)hipMemcpy(hd.data(), dbg, 8 * 3 * nw, hipMemcpyDeviceToHost);
      unsigned long long tmin = ~0ull, tmax = 0; std::vector<double> durs;
      for (size_t i = 0; i < nw; i++) if (hd[3 * i + 2]) { tmin = std::min(tmin, hd[3 * i + 1]); tmax = std::max(tmax, hd[3 * i + 2]); durs.push_back((double)(hd[3 * i + 2] - hd[3 * i + 1])); }
      std::sort(durs.begin(), durs.end());
      printf("{\"dbg\": \"waves %zu, launch span %.1f us, wave lifetime us min %.1f p10 %.1f median %.1f p90 %.1f max %.1f; start spread: last start %.1f us after the first;", durs.size(), (tmax - tmin) / 100.0,
             durs[0] / 100, durs[durs.size() / 10] / 100, durs[durs.size() / 2] / 100, durs[durs.size() * 9 / 10] / 100, durs.back() / 100, 0.0);
      for (int frac = 1; frac <= 3; frac++) {
        int hist[16] = {0};
        const unsigned long long tq = tmin + (tmax - tmin) * frac / 4;
        std::vector<int> cnt(1 << 18, 0);
        for (size_t i = 0; i < nw; i++) if (hd[3 * i + 2] && hd[3 * i + 1] <= tq && tq < hd[3 * i + 2]) { unsigned v = (unsigned)hd[3 * i]; cnt[((v >> 16) << 14) | (((v >> 8) & 0xff) << 2) | ((v >> 4) & 3)]++; }
        int used = 0; for (int c : cnt) if (c) { hist[c < 15 ? c : 15]++; used++; }
        printf(" at %d/4: SIMDs busy %d:", frac, used);
        for (int c = 1; c < 16; c++) if (hist[c]) printf(" %dw x%d", c, hist[c]);
        printf(";");
      }
      printf("\"}\n");
    }
#endif
    std::vector<uint8_t> o(nout); (void)hipMemcpy(o.data(), dout, nout, hipMemcpyDeviceToHost);
    unsigned long long sum = 0; for (long long i = 0; i < nout; i++) sum = sum * 1099511628211ull + o[i];
    if (!sum0) sum0 = sum;
    printf("{\"exp_bits\": %d, \"shader_mhz_under_load\": %.0f, \"chunk_bytes\": %d, \"wavefronts\": %u, \"waves_per_cu\": %.2f, \"ms\": %.4f, \"decoded_gbit_s\": %.1f, \"same_output\": %s}\n", V3_EXP, mhz, cb, grid.x * V3_WGW,
           (double)grid.x * V3_WGW / ncu, best, nout * 8 / (best * 1e-3) / 1e9, sum == sum0 ? "true" : "false");
  }
  return 0;
}
