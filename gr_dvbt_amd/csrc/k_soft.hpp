// k_soft.hpp -- soft-decision demapping and decoding (second half; gr-dvbt's TODO.txt:25-26 "soft decision demapper / Viterbi", the disabled
// soft-metric table lib/d_metrics.c:34-133).  An OPT-IN mode of the segment API (dvbt_rx_params.soft_decision = 1): the reference decodes hard decisions only,
// so there is no oracle for it and no parity claim -- it is validated against the hard path (identical TS on a clean loopback; lower packet error rate under
// noise, tests/test_gpu_soft.py).  What changes between the symbol kernel and the byte de-interleaver:
//   hard:  label byte per carrier -> inner_kernel (A5 + A6 on labels) -> viterbi3_kernel (packed 16-bit cells, hard branch metrics)
//   soft:  equalised carrier + channel state per carrier (EQ tap, CSI tap of the symbol kernel) -> soft_demap_kernel: one 8-bit log-likelihood ratio per coded
//          bit, max-log over the constellation table, weighted with the carrier's channel power -> viterbi_soft_kernel: A5 + A6 as a table gather while
//          fetching, one wavefront per chunk, a lane per state, 32-bit path metrics, correlation branch metrics, erasures = 0, decisions = the compare's
//          SGPR pair, parked per 64 steps in a scratch slot in HBM, scalar traceback.
// Same chunking idea as the hard kernel (every chunk decoded by an independent decoder with a warm-up in front and a look-ahead behind) and the same output
// stream: byte j = information bits 8 j .. 8 j + 7, total_steps / 8 - ntraceback bytes, so that the byte de-interleaver, RS decoder and descrambler run unchanged.
#pragma once
#include "k_viterbi3.hpp"

namespace dvbt {

constexpr int SOFT_CLAMP = 31;           // soft values live in [-31, 31]; > 0: the coded bit is more likely 0
constexpr float SOFT_UNIT = 8.0f;        // a carrier on a constellation point, one level from the decision boundary, full channel power: +-8

// one workgroup per delivered symbol: mean channel power of the symbol, then per carrier and bit the max-log LLR.  The constellation is a product of two
// axes (dvbt_tables.hpp constellation_points: even label bits belong to I, odd bits to Q), so min over the points with bit j = b of |e - point|^2 splits into
// a min over that axis' levels plus a term of the other axis that cancels in d1 - d0: 2 x 2^(m/2) distances per carrier instead of 2^m.  A thread handles
// two neighbouring carriers and stores their 2 m soft values as m / 2 aligned dwords.
template <int PA>   // bits per axis: m = 2 PA
__device__ __forceinline__ void soft_demap_symbol(const float2 *__restrict__ eq, const float *__restrict__ csi, int P, float inv_mean, float inv_step2,
                                                  const float (*lev)[8], int8_t *__restrict__ out, int t)
{
  constexpr int M = 2 * PA, NL = 1 << PA;
  uint32_t *out32 = reinterpret_cast<uint32_t *>(out);
  for (int i2 = t; i2 < P / 2; i2 += 256) {
    uint32_t word[PA] = {};                                          // 2 carriers x M bytes = PA dwords
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
          const int byte = c * M + 2 * jj + a;                       // label bit j = 2 jj + a, MSB first
          word[byte >> 2] |= (uint32_t)((int)v & 0xff) << (8 * (byte & 3));
        }
      }
    }
#pragma unroll
    for (int k = 0; k < PA; k++) out32[i2 * PA + k] = word[k];
  }
}

__global__ __launch_bounds__(256) void soft_demap_kernel(const float2 *__restrict__ eq, const float *__restrict__ csi, const RxState *st, InnerParams ip,
                                                        const float2 *__restrict__ points, float inv_step2, int8_t *__restrict__ out)
{
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
  const float2 *e = eq + s * P; const float *c = csi + s * P; int8_t *o = out + (size_t)os * P * m;
  if (pa == 1) soft_demap_symbol<1>(e, c, P, inv_mean, inv_step2, s_lev, o, t);
  else if (pa == 2) soft_demap_symbol<2>(e, c, P, inv_mean, inv_step2, s_lev, o, t);
  else soft_demap_symbol<3>(e, c, P, inv_mean, inv_step2, s_lev, o, t);
}

// ---- the decoder.  Encoder (inner_coder_impl.cc:33-48): r = state | b << 6, X = parity(r & 0x79), Y = parity(r & 0x5b), next state = r >> 1.
// One wavefront per chunk, a lane per state, and NO LDS allocation: what bounded the first version (27.9 ms on 17 superframes of 8k QAM64 7/8, 25x the hard
// kernel) was residency -- 25 KB of decisions + staged symbols per decoder = 6 wavefronts per CU, each one a dependent chain of two ds_bpermute per step.  Now
//   * the soft pairs of 64 steps are fetched by the 64 lanes (one step each, straight from the demapper's output THROUGH the A5 + A6 gather table, so the soft
//     de-interleaver launch and its buffer are gone), one block ahead of the block being decoded, and handed to the step loop by v_readlane;
//   * every lane shifts its state's decision bit into a register (32 steps per register) and every 64 steps the wavefront writes 2 x 256 coalesced bytes to
//     its slot of a scratch buffer in HBM (8 bytes per step: written once, read once);
//   * the traceback reads the blocks back the same way and walks them with v_readlane (lane = the wave-uniform state) + scalar instructions, 8 bytes out per block.
// Residency is then bounded by the wavefront slots (8 per SIMD), and the SIMD is issue-bound on 8 VALU + 2 DS instructions per step.
constexpr int SV_B = 256;                  // decoded bytes per chunk (host picks 128 when there are too few chunks to fill the slots twice)
constexpr int SV_WARM = 256;               // warm-up steps in front of a chunk
constexpr int SV_LOOK = 128;               // steps decoded behind a chunk before its traceback starts, at least (the reference's depth, 8 ntraceback, is
                                           // 40 steps at rate 1/2: the output DELAY stays the reference's, the decision depth need not)
constexpr int SV_WAVES = 4;                // wavefronts per workgroup
constexpr int SV_GRID = 2048;              // workgroups: 256 CUs x 8 waves per SIMD
constexpr int SV_MAXBLK = (SV_WARM + 8 * SV_B + 8 * 24 + 63) / 64 + 1;      // 64-step blocks of a chunk
constexpr size_t SV_SCRATCH_WORDS = (size_t)SV_GRID * SV_WAVES * SV_MAXBLK * 64;

// gather table of A5 + A6 on soft values (symbol_inner_interleaver_impl.cc:197-209, bit_inner_deinterleaver_impl.cc:120-184): tab[odd][x] = index, within the
// symbol's P * v soft values as the demapper wrote them, of soft value x of the symbol after both de-interleavers.  x = v q + k is bit k (MSB first) of word
// q = 126 b + i; after both de-interleavers that is bit e = perm(v, v i + k) of the word w = (i - off[e]) mod 126 of block b after the symbol de-interleaver,
// which is carrier H(126 b + w) (even symbols) / H^-1 (odd symbols) of the demapper's output
__global__ __launch_bounds__(256) void soft_tab_kernel(InnerParams ip, const uint16_t *__restrict__ H, const uint16_t *__restrict__ Hinv, int *__restrict__ tab)
{
  const int P = ip.payload, v = ip.m;
  const int off[6] = {0, 63, 105, 42, 21, 84};
  for (int x = blockIdx.x * 256 + threadIdx.x; x < 2 * P * v; x += gridDim.x * 256) {
    const int odd = x >= P * v, xx = odd ? x - P * v : x;
    const int q = xx / v, k = xx - q * v, b = q / 126, i = q - b * 126;
    const int e = ((v * i + k) % v) / (v / 2) + 2 * ((v * i + k) % (v / 2));
    int w = i - off[e]; w += w < 0 ? 126 : 0; w += w < 0 ? 126 : 0;
    const int qq = b * 126 + w;
    tab[x] = (odd ? Hinv[qq] : H[qq]) * v + e;
  }
}

__global__ __launch_bounds__(64 * SV_WAVES) __attribute__((amdgpu_waves_per_eu(8, 8)))
void viterbi_soft_kernel(const int8_t *__restrict__ soft, const int *__restrict__ tab, const int *__restrict__ sym_index, uint8_t *__restrict__ out,
                         const RxState *st, VitParams vp, unsigned long long *__restrict__ scratch, int B)
{
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // told to be wave-uniform: chunk geometry and the traceback state go scalar
  if (st->first_out < 0) return;
  const long long total_steps = st->n_vit_steps, total_out = total_steps / 8 - vp.ntb;
  const long long n_soft = st->n_vit_in * vp.m;                   // soft values that exist
  const int PM = vp.payload * vp.m;
  const int par0 = sym_index[st->first_out] & 1;                  // delivered symbols are consecutive: the parity of symbol os is (par0 + os) & 1
  unsigned *dec = (unsigned *)(scratch + ((size_t)blockIdx.x * SV_WAVES + wv) * (SV_MAXBLK * 64));
  // add-compare-select: lane = new state ns; predecessors p0 = 2 (ns & 31), p1 = p0 + 1; input bit = ns >> 5.  r1 = r0 | 1 and bit 0 is a tap of both
  // generators: the branch metric of predecessor 1 is minus that of predecessor 0
  const int p0 = (lane & 31) << 1, bin = lane >> 5;
  const int r0 = p0 | (bin << 6);
  const int sx0 = 1 - 2 * (__popc((unsigned)(r0 & 0x79)) & 1), sy0 = 1 - 2 * (__popc((unsigned)(r0 & 0x5b)) & 1);
  const int a0 = p0 << 2, a1 = a0 + 4;
  const long long nslots = (long long)gridDim.x * SV_WAVES;
  for (long long chunk = (long long)blockIdx.x * SV_WAVES + wv; chunk * B < total_out; chunk += nslots) {
    const long long b0 = chunk * B, b1 = b0 + B < total_out ? b0 + B : total_out;
    long long t0 = 8 * b0 - SV_WARM; if (t0 < 0) t0 = 0;          // first step of this decoder (a multiple of 64)
    long long t1 = 8 * b1 + (8 * vp.ntb > SV_LOOK ? 8 * vp.ntb : SV_LOOK); if (t1 > total_steps) t1 = total_steps;   // look-ahead behind the chunk
    const int T = (int)(t1 - t0), nblk = (T + 63) >> 6;
    // chunk-relative 32-bit arithmetic for the depuncturing (viterbi_decoder_impl.cc:241-256 with 0 for the erasures) and the symbol / position split
    const long long qb = (2 * t0) / vp.plen; const int phb = (int)(2 * t0 - qb * vp.plen);
    const long long rb = qb * vp.n, osb = rb / PM; const int xb = (int)(rb - osb * PM);
    auto fetch = [&](int blk) -> int {
      const int i = blk * 64 + lane;
      if (i >= T) return 0;
      const unsigned x = (unsigned)(phb + 2 * i), dq = x / (unsigned)vp.plen; const int ph = (int)(x - dq * vp.plen);
      int val[2];
#pragma unroll
      for (int h = 0; h < 2; h++) {
        val[h] = 0;
        if ((vp.punct_mask >> (ph + h)) & 1) {
          const int pf = (int)((vp.prefix_nib >> (4 * (ph + h))) & 15);
          const int rr = xb + (int)dq * vp.n + pf;               // relative to soft value osb * PM
          if (rb - xb + rr < n_soft) {
            const int dos = rr / PM, xx = rr - dos * PM;
            const long long os = osb + dos;
            val[h] = soft[os * PM + tab[(((int)os + par0) & 1) * PM + xx]];
          }
        }
      }
      return (val[0] & 0xff) | (val[1] << 8);
    };
    int M = 0;                                                    // all states equal at the start (a segment begins at a superframe start of a running stream:
                                                                  // the encoder's state there is not known, as for the hard decoder)
    int cur = fetch(0);
    for (int blk = 0; blk < nblk; blk++) {
      const int nxt = blk + 1 < nblk ? fetch(blk + 1) : 0;
      unsigned d0 = 0, d1 = 0;                                   // this state's decisions of steps 0..31 / 32..63 of the block, first step in the MSB
#pragma unroll
      for (int j = 0; j < 64; j++) {
        const int pr = __builtin_amdgcn_readlane(cur, j);
        const int sx = (int)(signed char)(pr & 0xff), sy = pr >> 8;
        const int bm = sx * sx0 + sy * sy0;
        const int m0 = __builtin_amdgcn_ds_bpermute(a0, M), m1 = __builtin_amdgcn_ds_bpermute(a1, M);
        const int c0 = m0 + bm, c1 = m1 - bm;
        M = c1 > c0 ? c1 : c0;
        // decision (c1 > c0) shifted into the register as the carry of an add: compare + add-with-carry, kept together (left to itself the compiler
        // parks c0 and c1 of all 64 steps in scratch and compares after the loop)
        if (j < 32) asm volatile("v_cmp_gt_i32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(d0) : "v"(c1), "v"(c0) : "vcc");
        else asm volatile("v_cmp_gt_i32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(d1) : "v"(c1), "v"(c0) : "vcc");
      }
      if (t0 + 64 * blk + 63 >= 8 * b0) {                         // warm-up blocks are never traced
        dec[(blk * 2) * 64 + lane] = d0; dec[(blk * 2 + 1) * 64 + lane] = d1;
      }
      cur = nxt;
    }
    // best end state (steps past T are erasures: every state then carries a best metric of step T - 1 or a later one)
    int best = M, bl = lane;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int om = __shfl_xor(best, o), ol = __shfl_xor(bl, o); if (om > best || (om == best && ol < bl)) { best = om; bl = ol; } }
    int s = __builtin_amdgcn_readfirstlane(bl);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    // traceback, scalar: the input bit of step t is bit 5 of the state after it; bytes MSB first, a block = 8 bytes
    for (int blk = nblk - 1; blk >= 0; blk--) {
      const long long tb = t0 + 64 * blk;
      if (tb + 63 < 8 * b0) break;
      const int w0 = (int)__hip_atomic_load(&dec[(blk * 2) * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int w1 = (int)__hip_atomic_load(&dec[(blk * 2 + 1) * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned acc_lo = 0, acc_hi = 0;
#pragma unroll
      for (int j = 63; j >= 0; j--) {
        const unsigned bit = (unsigned)s >> 5;
        if (j < 32) acc_lo |= bit << ((j >> 3) * 8 + 7 - (j & 7)); else acc_hi |= bit << (((j - 32) >> 3) * 8 + 7 - (j & 7));
        const unsigned ww = (unsigned)__builtin_amdgcn_readlane(j < 32 ? w0 : w1, s);
        s = ((s & 31) << 1) | (int)((ww >> (31 - (j & 31))) & 1u);
      }
      const long long ob = (tb >> 3) + lane;
      if (lane < 8 && ob >= b0 && ob < b1) out[ob] = (uint8_t)((lane < 4 ? acc_lo : acc_hi) >> (8 * (lane & 3)));
    }
  }
}

}  // namespace dvbt
