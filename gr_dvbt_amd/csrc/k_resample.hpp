// k_resample.hpp -- front of the RX flowgraph: rational_resampler_ccc(64, 70) + multiply_const as one
// kernel.  Stock GNU Radio blocks in apps/dvbt_rx_demo*.grc (rational_resampler_xxx_0, blocks_multiply_const_vxx_0); the
// algorithm restated is gr-filter's rational_resampler_base general_work + fir_filter (see dvbt_tables.hpp).
//   output M (counted from the stream start):  n = floor(M rd / ri),  branch = (M rd) mod ri,
//   out[M] = scale * sum_k branch[k] * x[n - k]      (k = nt-1 .. 0, i.e. ascending input index; x[<0] = 0)
// One thread per output; a workgroup of 256 outputs stages the ~ 256 rd/ri + nt input samples it needs and the
// branch table in LDS.  HBM-bound: 8 B in x rd/ri + 8 B out per output sample.
#pragma once
#include <hip/hip_runtime.h>
#include "k_frontend.hpp"

namespace dvbt {

constexpr int RS_MAX_BRANCH_FLOATS = 64 * 48;      // ri * nt of the largest supported design
constexpr int RS_TILE_IN = 768;                    // input samples staged per 256 outputs (256 * rd/ri + nt must fit)

// x[i] holds the input sample with stream index xbase + i (xbase <= 0 means the buffer starts with history that lies
// before the stream start and is never read), xlen samples; outputs M0 .. M0+count-1
__global__ __launch_bounds__(256) void resample_scale_kernel(const float2 *__restrict__ x, long long xbase, long long xlen, long long M0, long long count,
                                                            int ri, int rd, int nt, const float *__restrict__ br, float scale, float2 *__restrict__ out)
{
  __shared__ float s_br[RS_MAX_BRANCH_FLOATS];
  __shared__ float2 s_x[RS_TILE_IN];
  const int tid = threadIdx.x;
  const long long m_blk = (long long)blockIdx.x * 256;
  if (m_blk >= count) return;
  for (int i = tid; i < ri * nt; i += 256) s_br[i] = br[i];
  const long long Mf = M0 + m_blk, Ml = (M0 + (m_blk + 255 < count - 1 ? m_blk + 255 : count - 1));
  const long long n_lo = (Mf * rd) / ri - (nt - 1), n_hi = (Ml * rd) / ri;       // stream indices needed by the block
  const int span = (int)(n_hi - n_lo + 1);
  for (int i = tid; i < span; i += 256) {
    const long long n = n_lo + i, q = n - xbase;
    s_x[i] = (n >= 0 && q >= 0 && q < xlen) ? x[q] : make_float2(0.f, 0.f);
  }
  __syncthreads();
  const long long m = m_blk + tid;
  if (m >= count) return;
  const long long M = M0 + m, pr = M * rd;
  const long long n = pr / ri; const int ctr = (int)(pr - n * ri);
  const float *t = s_br + ctr * nt;
  const int base = (int)(n - n_lo);                                               // s_x index of x[n]
  float ar = 0.f, ai = 0.f;
  for (int k = nt - 1; k >= 0; k--) { const float2 v = s_x[base - k]; ar += v.x * t[k]; ai += v.y * t[k]; }
  out[m] = make_float2(ar * scale, ai * scale);
}

}  // namespace dvbt
