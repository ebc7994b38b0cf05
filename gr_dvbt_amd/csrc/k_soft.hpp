// k_soft.hpp -- soft-decision demapping and decoding (second half; gr-dvbt's TODO.txt:25-26 "soft decision demapper / Viterbi", the disabled
// soft-metric table lib/d_metrics.c:34-133).  An OPT-IN mode of the segment API (dvbt_rx_params.soft_decision = 1): the reference decodes hard decisions only,
// so there is no oracle for it and no parity claim -- it is validated against the hard path (identical TS on a clean loopback; lower packet error rate under
// noise, tests/test_gpu_soft.py).  What changes between the symbol kernel and the byte de-interleaver:
//   hard:  label byte per carrier -> inner_kernel (A5 + A6 on labels) -> viterbi3_kernel (packed 16-bit cells, hard branch metrics)
//   soft:  equalised carrier + channel state per carrier (EQ tap, CSI tap of the symbol kernel) -> soft_demap_kernel: one 8-bit log-likelihood ratio per coded
//          bit, max-log over the two axes' levels, weighted with the carrier's channel power, written straight to its place after A5 + A6 (a scatter inside
//          the workgroup's LDS image of the symbol) -> viterbi_soft4_kernel (k_soft4.hpp): depuncturing while staging, four chunks per wavefront on the hard
//          kernel's cell layout, 16-bit path metrics, correlation branch metrics, erasures = 0, decisions parked in HBM, traceback by lane.
// Same chunking idea as the hard kernel (every chunk decoded by an independent decoder with a warm-up in front and a look-ahead behind) and the same output
// stream: byte j = information bits 8 j .. 8 j + 7, total_steps / 8 - ntraceback bytes, so that the byte de-interleaver, RS decoder and descrambler run unchanged.
#pragma once
#include "k_viterbi3.hpp"

namespace dvbt {

constexpr int SOFT_CLAMP = 31;           // soft values live in [-31, 31]; > 0: the coded bit is more likely 0
constexpr float SOFT_UNIT = 8.0f;        // a carrier on a constellation point, one level from the decision boundary, full channel power: +-8

// one workgroup per delivered symbol: mean channel power of the symbol, then per carrier and bit the max-log LLR.  The constellation is a product of two
// axes (dvbt_tables.hpp constellation_points: even label bits belong to I, odd bits to Q), so min over the points with bit j = b of |e - point|^2 splits into
// a min over that axis' levels plus a term of the other axis that cancels in d1 - d0: 2 x 2^(m/2) distances per carrier instead of 2^m.  The soft values
// go to their places after both inner de-interleavers in an LDS image of the symbol (byte scatter), which then leaves as 16-byte rows.
template <int PA>   // bits per axis: m = 2 PA
__device__ __forceinline__ void soft_demap_symbol(const float2 *__restrict__ eq, const float *__restrict__ csi, int P, float inv_mean, float inv_step2,
                                                  const float (*lev)[8], const uint16_t *__restrict__ dst, int8_t *s_out, int t)
{
  constexpr int M = 2 * PA, NL = 1 << PA;
  for (int i2 = t; i2 < P / 2; i2 += 256) {
#pragma unroll
    for (int c = 0; c < 2; c++) {
      const int i = 2 * i2 + c;
      const float2 e = eq[i];
      float w = csi[i] * inv_mean; w = fminf(w, 4.0f);               // a carrier in a fade counts less, one on a peak at most 4x
      const float scale = inv_step2 * w * SOFT_UNIT;
#pragma unroll
      for (int a = 0; a < 2; a++) {
        const float z = a ? e.y : e.x;
        float d0[PA], d1[PA];
#pragma unroll
        for (int jj = 0; jj < PA; jj++) { d0[jj] = 3.0e38f; d1[jj] = 3.0e38f; }
#pragma unroll
        for (int u = 0; u < NL; u++) {
          const float dz = z - lev[a][u], d = dz * dz;
#pragma unroll
          for (int jj = 0; jj < PA; jj++) { if ((u >> (PA - 1 - jj)) & 1) d1[jj] = fminf(d1[jj], d); else d0[jj] = fminf(d0[jj], d); }
        }
#pragma unroll
        for (int jj = 0; jj < PA; jj++) {
          float v = (d1[jj] - d0[jj]) * scale;
          v = fminf(fmaxf(rintf(v), (float)-SOFT_CLAMP), (float)SOFT_CLAMP);
          s_out[dst[i * M + 2 * jj + a]] = (int8_t)(int)v;           // label bit j = 2 jj + a, MSB first, to its place after A5 + A6
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void soft_demap_kernel(const float2 *__restrict__ eq, const float *__restrict__ csi, const RxState *st, InnerParams ip,
                                                        const float2 *__restrict__ points, float inv_step2, const int *__restrict__ sym_index,
                                                        const uint16_t *__restrict__ dst_tab, int8_t *__restrict__ out)
{
  extern __shared__ __attribute__((aligned(16))) int8_t s_out[];    // the symbol's P m soft values in the order after both inner de-interleavers
  __shared__ float s_lev[2][8];
  __shared__ float s_red[256];
  const int os = blockIdx.x, t = threadIdx.x, P = ip.payload, m = ip.m, pa = m / 2, nl = 1 << pa;
  if (st->first_out < 0 || os >= st->n_out_symbols) return;
  const size_t s = (size_t)st->first_out + os;
  if (t < 2 * nl) {                                                  // level u of axis a: the point whose bits of that axis are u and whose other bits are 0
    const int a = t / nl, u = t - a * nl;
    int label = 0;
    for (int jj = 0; jj < pa; jj++) label |= ((u >> (pa - 1 - jj)) & 1) << (m - 1 - (2 * jj + a));
    s_lev[a][u] = a ? points[label].y : points[label].x;
  }
  float acc = 0.f;
  for (int i = t; i < P; i += 256) acc += csi[s * P + i];
  s_red[t] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (t < o) s_red[t] += s_red[t + o]; __syncthreads(); }
  const float inv_mean = (float)P / fmaxf(s_red[0], 1e-30f);
  const float2 *e = eq + s * P; const float *c = csi + s * P;
  const uint16_t *dst = dst_tab + (size_t)(sym_index[s] & 1) * P * m;
  if (pa == 1) soft_demap_symbol<1>(e, c, P, inv_mean, inv_step2, s_lev, dst, s_out, t);
  else if (pa == 2) soft_demap_symbol<2>(e, c, P, inv_mean, inv_step2, s_lev, dst, s_out, t);
  else soft_demap_symbol<3>(e, c, P, inv_mean, inv_step2, s_lev, dst, s_out, t);
  __syncthreads();
  const uint4 *src = reinterpret_cast<const uint4 *>(s_out);         // P m is a multiple of 16 (P = 1512 or 6048)
  uint4 *o = reinterpret_cast<uint4 *>(out + (size_t)os * P * m);
  for (int i = t; i < P * m / 16; i += 256) o[i] = src[i];
}

// scatter table of A5 + A6 on soft values (symbol_inner_interleaver_impl.cc:197-209, bit_inner_deinterleaver_impl.cc:120-184): soft value x = v q + k of a
// symbol AFTER both de-interleavers (bit k, MSB first, of word q = 126 b + i) is bit e = perm(v, v i + k) of the word w = (i - off[e]) mod 126 of block b after
// the symbol de-interleaver, which is carrier H(126 b + w) (even symbols) / H^-1 (odd symbols): dst[odd][carrier v + e] = x
__global__ __launch_bounds__(256) void soft_tab_kernel(InnerParams ip, const uint16_t *__restrict__ H, const uint16_t *__restrict__ Hinv, uint16_t *__restrict__ dst)
{
  const int P = ip.payload, v = ip.m;
  const int off[6] = {0, 63, 105, 42, 21, 84};
  for (int x = blockIdx.x * 256 + threadIdx.x; x < 2 * P * v; x += gridDim.x * 256) {
    const int odd = x >= P * v, xx = odd ? x - P * v : x;
    const int q = xx / v, k = xx - q * v, b = q / 126, i = q - b * 126;
    const int e = ((v * i + k) % v) / (v / 2) + 2 * ((v * i + k) % (v / 2));
    int w = i - off[e]; w += w < 0 ? 126 : 0; w += w < 0 ? 126 : 0;
    const int qq = b * 126 + w;
    dst[(size_t)odd * P * v + (odd ? Hinv[qq] : H[qq]) * v + e] = (uint16_t)xx;
  }
}

}  // namespace dvbt
