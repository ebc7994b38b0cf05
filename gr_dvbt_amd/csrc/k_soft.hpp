// k_soft.hpp -- soft-decision demapping and decoding (second half; gr-dvbt's TODO.txt:25-26 "soft decision demapper / Viterbi", the disabled
// soft-metric table lib/d_metrics.c:34-133).  An OPT-IN mode of the segment API (dvbt_rx_params.soft_decision = 1): the reference decodes hard decisions only,
// so there is no oracle for it and no parity claim -- it is validated against the hard path (identical TS on a clean loopback; lower packet error rate under
// noise, tests/test_gpu_soft.py).  What changes between the symbol kernel and the byte de-interleaver:
//   hard:  label byte per carrier -> inner_kernel (A5 + A6 on labels) -> viterbi3_kernel (packed 16-bit cells, hard branch metrics)
//   soft:  equalised carrier + channel state per carrier (EQ tap, CSI tap of the symbol kernel) -> soft_demap_kernel: one 8-bit log-likelihood ratio per coded
//          bit, max-log over the constellation table, weighted with the carrier's channel power -> soft_inner_kernel: A5 + A6 as one gather on the soft values
//          -> viterbi_soft_kernel: one wavefront per chunk, a lane per state, 32-bit path metrics, correlation branch metrics, erasures = 0, decisions by ballot
//          into LDS, traceback by one lane.
// Same chunking idea as the hard kernel (every chunk decoded by an independent decoder with a warm-up in front and a look-ahead behind) and the same output
// stream: byte j = information bits 8 j .. 8 j + 7, total_steps / 8 - ntraceback bytes, so that the byte de-interleaver, RS decoder and descrambler run unchanged.
// Not tuned: ~10x the hard kernel's time (an ACS step costs two ds_bpermute and a dozen VALU instructions for ONE decoder).
#pragma once
#include "k_viterbi3.hpp"

namespace dvbt {

constexpr int SOFT_CLAMP = 31;           // soft values live in [-31, 31]; > 0: the coded bit is more likely 0
constexpr float SOFT_UNIT = 8.0f;        // a carrier on a constellation point, one level from the decision boundary, full channel power: +-8

// one workgroup per delivered symbol: mean channel power of the symbol, then per carrier and bit the max-log LLR
__global__ __launch_bounds__(256) void soft_demap_kernel(const float2 *__restrict__ eq, const float *__restrict__ csi, const RxState *st, InnerParams ip,
                                                        const float2 *__restrict__ points, float inv_step2, int8_t *__restrict__ out)
{
  __shared__ float s_pts[128];
  __shared__ float s_red[256];
  const int os = blockIdx.x, t = threadIdx.x, P = ip.payload, m = ip.m;
  if (st->first_out < 0 || os >= st->n_out_symbols) return;
  const size_t s = (size_t)st->first_out + os;
  if (t < ip.csize) { s_pts[2 * t] = points[t].x; s_pts[2 * t + 1] = points[t].y; }
  float acc = 0.f;
  for (int i = t; i < P; i += 256) acc += csi[s * P + i];
  s_red[t] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (t < o) s_red[t] += s_red[t + o]; __syncthreads(); }
  const float inv_mean = (float)P / fmaxf(s_red[0], 1e-30f);
  for (int i = t; i < P; i += 256) {
    const float2 e = eq[s * P + i];
    float w = csi[s * P + i] * inv_mean; w = fminf(w, 4.0f);          // a carrier in a fade counts less, one on a peak at most 4x
    float d0[6], d1[6];
#pragma unroll
    for (int j = 0; j < 6; j++) { d0[j] = 3.0e38f; d1[j] = 3.0e38f; }
    for (int c = 0; c < ip.csize; c++) {
      const float dx = e.x - s_pts[2 * c], dy = e.y - s_pts[2 * c + 1], d = dx * dx + dy * dy;
#pragma unroll
      for (int j = 0; j < 6; j++) if (j < m) { if ((c >> (m - 1 - j)) & 1) d1[j] = fminf(d1[j], d); else d0[j] = fminf(d0[j], d); }
    }
#pragma unroll
    for (int j = 0; j < 6; j++) if (j < m) {
      float v = (d1[j] - d0[j]) * inv_step2 * w * SOFT_UNIT;
      v = fminf(fmaxf(rintf(v), (float)-SOFT_CLAMP), (float)SOFT_CLAMP);
      out[((size_t)os * P + i) * m + j] = (int8_t)(int)v;
    }
  }
}

// A5 + A6 on soft values (symbol_inner_interleaver_impl.cc:197-209, bit_inner_deinterleaver_impl.cc:120-184): the bit k (MSB first) of word i of 126-word
// block b after both de-interleavers is bit e = perm(v, v i + k) of the word w = (i - off[e]) mod 126 of the same block after the symbol de-interleaver,
// which is carrier H(q) (even symbols) / H^-1(q) (odd symbols) of the demapper's output
__global__ __launch_bounds__(256) void soft_inner_kernel(const int8_t *__restrict__ in, const RxState *st, InnerParams ip, const int *__restrict__ sym_index,
                                                        const uint16_t *__restrict__ H, const uint16_t *__restrict__ Hinv, int8_t *__restrict__ out)
{
  const int os = blockIdx.x, P = ip.payload, v = ip.m;
  if (st->first_out < 0 || os >= st->n_out_symbols) return;
  const int odd = sym_index[st->first_out + os] & 1;
  const int8_t *src = in + (size_t)os * P * v;
  int8_t *dst = out + (size_t)os * P * v;
  const int off[6] = {0, 63, 105, 42, 21, 84};
  for (int x = threadIdx.x; x < P * v; x += 256) {
    const int q = x / v, k = x - q * v, b = q / 126, i = q - b * 126;
    const int e = ((v * i + k) % v) / (v / 2) + 2 * ((v * i + k) % (v / 2));
    int w = i - off[e]; w += w < 0 ? 126 : 0; w += w < 0 ? 126 : 0;   // (w + off[e]) % 126 == i
    const int qq = b * 126 + w;
    const int carrier = odd ? Hinv[qq] : H[qq];
    dst[x] = src[(size_t)carrier * v + e];
  }
}

// ---- the decoder.  Encoder (inner_coder_impl.cc:33-48): r = state | b << 6, X = parity(r & 0x79), Y = parity(r & 0x5b), next state = r >> 1.
constexpr int SV_B = 256;                  // decoded bytes per chunk (one wavefront)
constexpr int SV_WARM = 256;               // warm-up steps in front of a chunk
constexpr int SV_WAVES = 2;                // wavefronts (chunks) per workgroup (50 KB of LDS: three workgroups per CU)
constexpr int SV_MAXSTEPS = SV_WARM + 8 * SV_B + 8 * 24;

__global__ __launch_bounds__(64 * SV_WAVES) void viterbi_soft_kernel(const int8_t *__restrict__ in, uint8_t *__restrict__ out, const RxState *st, VitParams vp)
{
  __shared__ unsigned long long s_dec[SV_WAVES][SV_MAXSTEPS];     // decisions: bit s of word t = the survivor of state s at step t came from predecessor 2 (s & 31) + 1
  __shared__ short s_sym[SV_WAVES][SV_MAXSTEPS];                  // depunctured soft pair of a step: low byte X, high byte Y
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned long long *dec = s_dec[wv]; short *sym = s_sym[wv];
  const long long total_steps = st->n_vit_steps, total_out = total_steps / 8 - vp.ntb;
  const long long chunk = (long long)blockIdx.x * SV_WAVES + wv;
  const long long b0 = chunk * SV_B;
  if (b0 >= total_out) return;
  const long long b1 = b0 + SV_B < total_out ? b0 + SV_B : total_out;
  long long t0 = 8 * b0 - SV_WARM; if (t0 < 0) t0 = 0;            // first step of this decoder
  long long t1 = 8 * b1 + 8 * vp.ntb; if (t1 > total_steps) t1 = total_steps;
  const int T = (int)(t1 - t0);
  // ---- staging: the soft values of the steps, depunctured (viterbi_decoder_impl.cc:241-256 with 0 for the erasures)
  const long long n_soft = st->n_vit_in * vp.m;                   // soft values that exist
  for (int i = lane; i < T; i += 64) {
    const long long t = t0 + i, mb = 2 * t;
    const long long q = mb / vp.plen; const int ph = (int)(mb - q * vp.plen);
    int sx = 0, sy = 0;
    if (vp.punct[ph]) { const long long r = q * vp.n + vp.prefix[ph]; if (r < n_soft) sx = in[r]; }
    if (vp.punct[ph + 1]) { const long long r = q * vp.n + vp.prefix[ph + 1]; if (r < n_soft) sy = in[r]; }
    sym[i] = (short)((sx & 0xff) | (sy << 8));
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  // ---- add-compare-select: lane = new state ns; predecessors p0 = 2 (ns & 31), p1 = p0 + 1; input bit = ns >> 5
  const int p0 = (lane & 31) << 1, p1 = p0 | 1, bin = lane >> 5;
  auto par = [](int x) { return __popc((unsigned)x) & 1; };
  const int r0 = p0 | (bin << 6), r1 = p1 | (bin << 6);
  const int sx0 = 1 - 2 * par(r0 & 0x79), sy0 = 1 - 2 * par(r0 & 0x5b), sx1 = 1 - 2 * par(r1 & 0x79), sy1 = 1 - 2 * par(r1 & 0x5b);
  int M = 0;                                                      // all states equal at the start (a segment begins at a superframe start of a running stream:
                                                                  // the encoder's state there is not known, as for the hard decoder)
  for (int i = 0; i < T; i++) {
    const int pr = sym[i];
    const int sx = (int)(signed char)(pr & 0xff), sy = pr >> 8;
    const int m0 = __shfl(M, p0), m1 = __shfl(M, p1);
    const int c0 = m0 + sx * sx0 + sy * sy0, c1 = m1 + sx * sx1 + sy * sy1;
    const bool d = c1 > c0;
    M = d ? c1 : c0;
    const unsigned long long bal = __ballot(d);
    if (lane == 0) dec[i] = bal;
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  // ---- best end state, traceback by one lane (the input bit of step t is bit 5 of the state after it), bytes MSB first
  int best = M, bl = lane;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const int om = __shfl_xor(best, o), ol = __shfl_xor(bl, o); if (om > best || (om == best && ol < bl)) { best = om; bl = ol; } }
  if (lane == 0) {
    int s = bl;
    unsigned byte = 0;
    for (int i = T - 1; i >= 0; i--) {
      const long long t = t0 + i;
      const int bit = s >> 5;
      if (t < 8 * b1 && t >= 8 * b0) {
        byte |= (unsigned)bit << (7 - (int)(t & 7));
        if ((t & 7) == 0) { out[t >> 3] = (uint8_t)byte; byte = 0; }
      }
      s = ((s & 31) << 1) | (int)((dec[i] >> s) & 1ull);
      if (t < 8 * b0) break;
    }
  }
}

}  // namespace dvbt
