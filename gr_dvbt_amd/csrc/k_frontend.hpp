// k_frontend.hpp -- gfx950 kernels for A1 (ofdm_sym_acquisition), A2 (forward FFT) and A3
// (demod_reference_signals).  Float stages: parity with the reference is "within the stated
// tolerance at the equalised-carrier tap" (SURVEY 8a); the integer decisions they feed
// (cp position, integer CFO, symbol index mod 4, TPS bits, superframe start) must be identical.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dvbt {

// ---------------------------------------------------------------- device-resident run state
struct RxState {
  int status;            // bit0 init-acq failed, bit1 tracking lost, bit2 no superframe start
  int call0;             // general_work call (window index) in which initial acquisition succeeded
  int cp_start0;         // d_cp_start returned by the initial ml_sync
  int n_symbols;         // items produced by A1
  int first_out;         // symbol at which superframe_start fired (-1: none)
  int n_out_symbols;     // items passed downstream of A3
  float avg;             // peak detector IIR state d_avg after the initial search
  float eps_init;
  long long n_vit_in;    // bytes entering A7
  long long n_vit_steps; // trellis steps (real)
  long long n_vit_bytes; // bytes leaving A7
  long long n_rs_items;
  long long n_ts_bytes;
  int descr_base, descr_index;
  int rs_fail, rs_corr;
  float avg_lost;        // d_avg after the call that lost the lock (what a re-acquisition starts from)
  int rs_list_n;         // RS words handed to the second pass (rs_fix_kernel) by deint_rs_kernel
  long long sym_off;         // cut stream: OFDM symbols between the stream's first superframe start and this segment's (0: stream start)
  long long n_rs_words;      // RS words decoded (= 8 n_rs_items unless the segment continues a cut stream)
  long long stream_rs_items; // items the byte de-interleaver of a chain over the whole stream has produced up to this segment's end
  long long ts_first_packet; // RS word index (of this segment) of the first packet of the TS tap
  unsigned long long tps_bits; // TPS word of a BCH-valid frame, bit i = s_i, with the fields that change from frame to frame (sync word s1-s16,
                               // frame number s23-s24, parity s54-s67) cleared: identical for every frame of a stream; 0 = none seen
  int small_viol;              // acq_small_kernel met a phase-increment switch outside its call: the period goes through the general kernels
  int drift_known_off;         // acq_small_kernel has established that the float-accumulator model (k_drift.hpp) does not apply to this period (increments of both
                               // signs / too small): the host, which reads this block back before it decodes the period, launches none of the drift kernels
  int descr_unclean;           // a piece of a cut stream (descramble_scan_kernel, sym_off > 0): 1 = not every 16-packet call of the whole-stream descrambler's phase
                               // (dvbt_rx_cut.descr_phase16) finds its NSYNC inside this piece -- the stream's descrambler re-searches here (energy_descramble_impl.cc:121-141)
                               // and the host follows it call by call instead of taking the piece's packets in whole groups
};
// fields of a TPS word that are the same in every frame (reference_signals_impl.cc:883-916): s17-s22 length, s25-s53 parameters
constexpr unsigned long long TPS_STATIC_MASK = ((1ull << 54) - 1) & ~((1ull << 17) - 1) & ~(3ull << 23);

struct FrontParams {
  int N, cp, K, zl, payload, n_cp, n_tps, fi_start;
  int ncalls;            // windows available: (ncalls-1)*(N+cp) + 2N+cp+16 <= nsamples
  int R;                 // half-width of the precomputed lag range around the predicted CP position of a call (centre[call])
  float half_rho;        // (float)(rho/2)
  int keep_last;         // 1: the last acquired symbol is demodulated too.  The reference's demod needs the NEXT item to process one
                         // (demod_reference_signals_impl.cc:88-94), so the last item of a stream never leaves it; the last item in front of
                         // a lost lock does, as soon as the re-acquired stream delivers its first item
  int si_start;          // symbol of the frame at which the superframe hunt fires (with fi_start: the frame).  0 = the reference's test (symbol_index % 68 == 0 in frame
                         // d_fi_start, demod_reference_signals_impl.cc:122).  Non-zero only for a piece of a cut stream whose epoch began at a superframe start that the
                         // reference declared on stale counters (dvbt_rx_cut.start_delay_symbols): the piece delivers from that many symbols behind the true start
  int hunt_known;        // 1 (with si_start / a shifted fi_start: a piece of an epoch on a shifted grid): the hunt fires only once a frame end with a valid TPS word has
                         // set the counters -- a chain that starts from blank counters holds frame_index 0, which a shifted target frame may be
  int pad2;
  long long avail;       // samples in memory from the segment's first one on: a tracking window that has crept beyond its call's 2N + cp + 16 samples (the reference then
                         // reads and WRITES past its d_norm / d_corr arrays, ofdm_sym_acquisition_impl.cc:166-186,416-419: undefined there) reads the stream's own
                         // samples here, and zeros at or beyond this bound -- as oracle/o_acq.c does
  long long hist;        // samples of the stream that lie BEFORE the segment's first sample in memory (a restart inside a segment): a tracking
                         // window at the left edge of its call reads them, as the reference reads the history of its input buffer
};

// persistent members of the acquisition block between work() calls (block API only)
struct AcqState { int acquired; int cp_start; float avg; float phase; double phaseinc, nextphaseinc; int nextpos; int lost; };

constexpr int ACQ_R = 16;
constexpr int ACQ_INIT_TRIES = 4;
constexpr int ACQ_INIT_TRIES_MAX = 64;         // windows one search launch can examine (the lock-period walk asks for more after searches that found nothing)
constexpr int ACQ_CP_MAX = 2048;               // longest guard interval (8k, 1/4)

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) { return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }  // a*conj(b)
__device__ __forceinline__ float2 cdiv(float2 a, float2 b)
{ float den = b.x * b.x + b.y * b.y; return make_float2((a.x * b.x + a.y * b.y) / den, (a.y * b.x - a.x * b.y) / den); }

// ---------------------------------------------------------------- A1: CP correlation metric
// ml_sync (ofdm_sym_acquisition_impl.cc:148-250): gamma(i) = sum_{j<cp} x[i-j] conj(x[i-j-N]),
// phi(i) = sum |x[i-j]|^2 + |x[i-j-N]|^2, lambda = |gamma| - rho/2 * phi.
// One thread per lag: the initial search, lags N+cp-1 .. 2N+cp-2 of window `try`.  A workgroup owns 256 consecutive lags: the cp + 255 products
// x[i] conj(x[i-N]) and energy pairs they share are formed once (coalesced loads) and kept in LDS; a thread then sums its lag's cp terms in the
// reference's order (j ascending, i.e. sample index descending) -- the same float expressions as summing from global memory, ~5 us per launch instead of
// ~21 (a lock period's initial search was a quarter of config 5's time at 8 dB).
__global__ __launch_bounds__(256) void acq_metric_kernel(const float2 *__restrict__ iq, FrontParams p, const RxState *st,
                                                        int mode, float2 *__restrict__ gamma, float *__restrict__ lambda, int t_begin)
{
  __shared__ __attribute__((aligned(16))) float2 s_c[ACQ_CP_MAX + 256];
  __shared__ __attribute__((aligned(16))) float s_e[ACQ_CP_MAX + 256];
  const int N = p.N, cp = p.cp, tid = threadIdx.x;
  (void)mode;
  const int t = blockIdx.y + t_begin;
  if (t >= p.ncalls) return;
  if (t_begin > 0 && !(st->status & 1)) return;                 // later windows are searched only if the first one had no peak
  const int q0 = blockIdx.x * 256, q = q0 + tid;
  const float2 *w = iq + (long long)t * (N + cp) + q0;          // the workgroup's lags read samples N + q0 .. N + q0 + cp + 254 of the window and the same N earlier
  const int span = cp + 255 < N + cp - 1 - q0 ? cp + 255 : N + cp - 1 - q0;   // (the window's last lag is 2N+cp-2)
  for (int i = tid; i < span; i += 256) {
    const float2 a = w[N + i], b = w[i];
    s_c[i] = make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
    s_e[i] = (a.x * a.x + a.y * a.y) + (b.x * b.x + b.y * b.y);
  }
  __syncthreads();
  if (q >= N) return;
  const int oidx = t * N + q;
  const float2 *c = s_c + tid + cp - 1; const float *e = s_e + tid + cp - 1;   // the lag's own sample; tap j sits j below
  float gr = 0.f, gi = 0.f, phi = 0.f;
  int j = 0;
  for (; j + 8 <= cp; j += 8) {
    float2 cv[8]; float ev[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { cv[k] = c[-(j + k)]; ev[k] = e[-(j + k)]; }
#pragma unroll
    for (int k = 0; k < 8; k++) { gr += cv[k].x; gi += cv[k].y; phi += ev[k]; }
  }
  for (; j < cp; j++) { gr += c[-j].x; gi += c[-j].y; phi += e[-j]; }
  gamma[oidx] = make_float2(gr, gi);
  lambda[oidx] = sqrtf(gr * gr + gi * gi) - phi * p.half_rho;
}

// ---- following a sample-clock offset.  The reference's tracking window is centred on the previous call's peak
// (ofdm_sym_acquisition_impl.cc:516), so it follows a drifting CP position for as long as that stays inside the call's window.
// Here the tracking metric of ALL calls is computed up front for 2R lags per call, so the lags must be placed before the tracker
// has run: a coarse, FSM-free estimate of the CP position (arg max of lambda over the whole window, sliding sums) is taken at
// every ACQ_ANCHOR-th call ("anchors"), outliers are rejected against the previous accepted anchor, and centre[call] is the linear
// interpolation.  The tracker itself is unchanged and decides everything; the centres only say which 2R lags are available to it.
// If it ever needs a lag outside them (prediction off by more than R-8), the sequential tracker computes that call's metric on the
// spot (acq_track_kernel), so the result is always the reference's.
constexpr int ACQ_ANCHOR = 256;                // calls between anchors
constexpr int ACQ_ANCHOR_LAGS = 8;             // lags per thread in acq_anchor_kernel

// One workgroup per anchor.  The products x[i] conj(x[i-N]) and the energies of the window are formed once (coalesced loads) and kept
// in LDS together with their sums over blocks of 8; a thread then owns ACQ_ANCHOR_LAGS = 8 consecutive lags: cp / 8 block sums for its first lag,
// sliding sum for the rest.  An anchor is a hint (above): its sums need not be formed in the reference's order.  Product i lives at LDS index
// i + i / 8, so that the threads' reads (8 products apart) fall on distinct banks (unpadded, 32 lanes of a wavefront shared one bank pair: 39 us per launch).
__device__ __forceinline__ int acq_anchor_pad(int i) { return i + (i >> 3); }
__global__ __launch_bounds__(1024) void acq_anchor_kernel(const float2 *__restrict__ iq, FrontParams p, const RxState *st, int *__restrict__ anchor_pos)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int k = blockIdx.x + 1, tid = threadIdx.x, N = p.N, cp = p.cp;          // anchor 0 is the initial acquisition itself
  const int np = N + cp, nb = np / 8, P = acq_anchor_pad(np) + 8;                // N, cp: multiples of 8
  float2 *s_c = reinterpret_cast<float2 *>(smem_raw);            // [P]   product at sample N + i
  float2 *sb_c = s_c + P;                                        // [nb]  sums of 8 products
  float *s_e = reinterpret_cast<float *>(sb_c + nb);             // [P]   energy pair
  float *sb_e = s_e + P;                                         // [nb]
  __shared__ float s_best[16]; __shared__ int s_arg[16];
  if (st->status & 1) return;
  const int call = st->call0 + k * ACQ_ANCHOR;
  if (call >= p.ncalls) { if (tid == 0) anchor_pos[k] = -1; return; }
  const float2 *w = iq + (long long)call * (N + cp);
  for (int i = tid; i < np; i += 1024) {                          // samples N .. 2N+cp-2 of the window (product N+cp-1 does not exist: zero)
    float2 c = make_float2(0.f, 0.f); float e = 0.f;
    if (i < np - 1) {
      const float2 a = w[N + i], b = w[i];
      c = make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
      e = (a.x * a.x + a.y * a.y) + (b.x * b.x + b.y * b.y);
    }
    s_c[acq_anchor_pad(i)] = c; s_e[acq_anchor_pad(i)] = e;
  }
  __syncthreads();
  for (int bq = tid; bq < nb; bq += 1024) {
    float gr = 0.f, gi = 0.f, phi = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j++) { const float2 c = s_c[9 * bq + j]; gr += c.x; gi += c.y; phi += s_e[9 * bq + j]; }
    sb_c[bq] = make_float2(gr, gi); sb_e[bq] = phi;
  }
  __syncthreads();
  float best = -3.0e38f; int arg = 0;
  for (int t = tid; t * ACQ_ANCHOR_LAGS < N; t += 1024) {
    // lag q (sample index N+cp-1+q) sums the products at i = q .. q+cp-1 (i counted from sample N); q0 = 8 t: blocks t .. t + cp/8 - 1
    const int q0 = t * ACQ_ANCHOR_LAGS;
    float gr = 0.f, gi = 0.f, phi = 0.f;
    for (int m = 0; m < cp / 8; m++) { const float2 c = sb_c[t + m]; gr += c.x; gi += c.y; phi += sb_e[t + m]; }
    const int lo = 9 * t, hi = 9 * (t + cp / 8);                 // padded indices of products q0 and q0 + cp
#pragma unroll
    for (int u = 0; u < ACQ_ANCHOR_LAGS; u++) {
      const float lam = sqrtf(gr * gr + gi * gi) - phi * p.half_rho;
      if (lam > best) { best = lam; arg = q0 + u; }
      if (u + 1 < ACQ_ANCHOR_LAGS) {
        const float2 a = s_c[hi + u], c = s_c[lo + u];             // enters / leaves
        gr += a.x - c.x; gi += a.y - c.y; phi += s_e[hi + u] - s_e[lo + u];
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o); const int oa = __shfl_xor(arg, o);
    if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
  }
  if ((tid & 63) == 0) { s_best[tid >> 6] = best; s_arg[tid >> 6] = arg; }
  __syncthreads();
  if (tid == 0) {
    for (int wv = 1; wv < 16; wv++) if (s_best[wv] > best || (s_best[wv] == best && s_arg[wv] < arg)) { best = s_best[wv]; arg = s_arg[wv]; }
    anchor_pos[k] = arg + N + cp - 1;
  }
}
inline size_t acq_anchor_lds_bytes(int N, int cp) { const size_t np = (size_t)(N + cp); return (np + np / 8 + 8 + np / 8) * 12 + 64; }

// centre[call] for every call from call0 on.  n_anchors: anchors computed (slots 1..n_anchors of anchor_pos); 0 = none (block API: the
// window of one work() call is short, every call is centred on the carried CP position).  Any grid: a workgroup takes 1024 calls at a time and filters
// the anchors for itself in LDS (a few dozen values; one workgroup walking 17,680 calls after a serial pass over the anchors in global memory was 23 us).
constexpr int ACQ_CENTRE_MAX_ANCHORS = 4096;
__global__ __launch_bounds__(1024) void acq_centre_kernel(FrontParams p, const RxState *st, const int *__restrict__ anchor_pos, int n_anchors, int *__restrict__ centre)
{
  __shared__ int s_a[ACQ_CENTRE_MAX_ANCHORS + 1];
  if (st->status & 1) return;
  const int tid = threadIdx.x, c0 = st->cp_start0, call0 = st->call0;
  if ((long long)call0 + (long long)blockIdx.x * 1024 >= p.ncalls) return;
  if (n_anchors > ACQ_CENTRE_MAX_ANCHORS) n_anchors = ACQ_CENTRE_MAX_ANCHORS;
  for (int k = 1 + tid; k <= n_anchors; k += 1024) s_a[k] = anchor_pos[k];
  __syncthreads();
  if (tid == 0) {
    s_a[0] = c0;
    int acc = c0, kacc = 0;
    for (int k = 1; k <= n_anchors; k++) {
      const int v = s_a[k];
      // a sample clock off by more than ~100 ppm moves the peak by more than one sample per symbol: anything faster is not drift
      if (v >= 0 && abs(v - acc) <= (k - kacc) * ACQ_ANCHOR) { acc = v; kacc = k; } else s_a[k] = -1;
    }
  }
  __syncthreads();
  for (long long cl = (long long)call0 + (long long)blockIdx.x * 1024 + tid; cl < p.ncalls; cl += (long long)gridDim.x * 1024) {
    const int call = (int)cl;
    const int s = call - call0, k = s / ACQ_ANCHOR, f = s - k * ACQ_ANCHOR;
    int k0 = k; while (k0 > 0 && (k0 > n_anchors || s_a[k0] < 0)) k0--;                 // last accepted anchor at or before the call
    int k1 = k + 1; while (k1 <= n_anchors && s_a[k1] < 0) k1++;                          // next accepted anchor after it
    const int p0 = s_a[k0];
    int c = p0;
    if (k1 <= n_anchors) {
      const long long num = (long long)(s_a[k1] - p0) * ((long long)(k - k0) * ACQ_ANCHOR + f), den = (long long)(k1 - k0) * ACQ_ANCHOR;
      c = p0 + (int)((num >= 0 ? num + den / 2 : num - den / 2) / den);
    } else if (k0 > 0) {                                                                  // beyond the last anchor: keep the last slope
      int kp = k0 - 1; while (kp > 0 && s_a[kp] < 0) kp--;
      const long long num = (long long)(p0 - s_a[kp]) * ((long long)(k - k0) * ACQ_ANCHOR + f), den = (long long)(k0 - kp) * ACQ_ANCHOR;
      c = p0 + (int)((num >= 0 ? num + den / 2 : num - den / 2) / den);
    }
    centre[call] = c;
  }
}

// tracking metric (mode 1 of acq_metric_kernel) with the samples' products staged through LDS and register blocking: a workgroup
// owns 32 calls x 32 lags; a thread owns 4 consecutive lags of a call.  Per tile of 64 correlation taps the workgroup
// loads the 95 samples (and their partners N earlier) each call needs, once and coalesced; a thread then walks the tile's
// samples downwards: sample i is tap (q + 63 - i) of lag q, so every sample is read once per thread and feeds up to 4
// lags, each lag still seeing its taps in ascending order.  The sums use the same expressions in the same order as
// acq_metric_kernel, so gamma/lambda are bit-identical.
constexpr int ACQ_TM_CALLS = 32, ACQ_TM_TILE = 64, ACQ_TM_SPAN = ACQ_TM_TILE + 2 * ACQ_R - 1;
__global__ __launch_bounds__(256) void acq_track_metric_kernel(const float2 *__restrict__ iq, FrontParams p, const RxState *st, const int *__restrict__ centre,
                                                              float2 *__restrict__ gamma, float *__restrict__ lambda)
{
  // what the taps sum: the product x[i] conj(x[i - N]) and the energy pair of a sample, formed once while staging (a sample feeds up to 4 lags of up to 4
  // threads; the expressions are the per-tap ones of acq_metric_kernel and the build has no FP contraction, so gamma / lambda stay bit-identical)
  __shared__ float2 sC[ACQ_TM_CALLS][ACQ_TM_SPAN]; __shared__ float sE[ACQ_TM_CALLS][ACQ_TM_SPAN];
  if (st->status & 1) return;
  const int N = p.N, cp = p.cp, tid = threadIdx.x, c = tid >> 3, g4 = (tid & 7) * 4;
  const int call0 = blockIdx.x * ACQ_TM_CALLS, call = call0 + c;
  if (call0 + ACQ_TM_CALLS <= st->call0 || call0 >= p.ncalls) return;
  const bool active = call < p.ncalls && call >= st->call0;
  const int lag0 = (active ? centre[call] : 0) - p.R;             // first lag of this thread's call
  float gr[4] = {0.f, 0.f, 0.f, 0.f}, gi[4] = {0.f, 0.f, 0.f, 0.f}, phi[4] = {0.f, 0.f, 0.f, 0.f};
  constexpr int T = ACQ_TM_TILE;
  for (int j0 = 0; j0 < cp; j0 += T) {
    const int jn = cp - j0 < T ? cp - j0 : T;
    __syncthreads();
    for (int e0 = tid; e0 < ACQ_TM_CALLS * ACQ_TM_SPAN; e0 += 4 * 256) {       // 8 clamped loads per lane in flight
      float2 av[4], bv[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int e = e0 + k * 256, ec = e < ACQ_TM_CALLS * ACQ_TM_SPAN ? e : 0;
        const int cc = ec / ACQ_TM_SPAN, i = ec - cc * ACQ_TM_SPAN, cl = call0 + cc;
        const bool okc = cl < p.ncalls && cl >= st->call0;
        const int lag0c = centre[okc ? cl : st->call0] - p.R;
        const long long idx = (long long)cl * (N + cp) + lag0c - j0 - (T - 1) + i;   // sample x[lag0 + q - j] for q - (j - j0) = i - (T - 1)
        const bool oka = okc && idx >= -p.hist && idx < p.avail, okb = okc && idx - N >= -p.hist && idx - N < p.avail;
        av[k] = iq[oka ? idx : 0]; bv[k] = iq[okb ? idx - N : 0];
        if (!oka) av[k] = make_float2(0.f, 0.f);
        if (!okb) bv[k] = make_float2(0.f, 0.f);
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int e = e0 + k * 256;
        if (e < ACQ_TM_CALLS * ACQ_TM_SPAN) {
          const int cc = e / ACQ_TM_SPAN, i = e - cc * ACQ_TM_SPAN;
          const float2 a = av[k], b = bv[k];
          sC[cc][i] = make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); sE[cc][i] = (a.x * a.x + a.y * a.y) + (b.x * b.x + b.y * b.y);
        }
      }
    }
    __syncthreads();
    // step s reads sample i = g4 + 3 + (T - 1) - s, which is tap (s - 3 + u) of lag g4 + u
    auto step = [&](int s, bool all) {
      const int i = g4 + 3 + (T - 1) - s;
      const float2 pr = sC[c][i];
      const float cr = pr.x, ci = pr.y, en = sE[c][i];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int jj = s - 3 + u;
        if (all || (jj >= 0 && jj < jn)) { gr[u] += cr; gi[u] += ci; phi[u] += en; }
      }
    };
    step(0, false); step(1, false); step(2, false);
    for (int s = 3; s < jn; s++) step(s, true);
    step(jn, false); step(jn + 1, false); step(jn + 2, false);
  }
  if (!active) return;
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int q = g4 + u, oidx = call * 2 * p.R + q;
    if ((long long)call * (N + cp) + lag0 + q - cp + 1 - N < -p.hist) { gamma[oidx] = make_float2(0.f, 0.f); lambda[oidx] = -3.0e38f; continue; }   // before the stream's first sample
    gamma[oidx] = make_float2(gr[u], gi[u]);
    lambda[oidx] = sqrtf(gr[u] * gr[u] + gi[u] * gi[u]) - phi[u] * p.half_rho;
  }
}

// peak_detect_process (ofdm_sym_acquisition_impl.cc:72-146); avg persists across calls.
// Same state machine written one sample per iteration (the reference's non-consuming transitions
// -- rise detection, peak completion -- are folded into the sample that triggers them), so the
// loop body is straight-line and the metric reads can be pipelined:
//   state 1: new maximum -> keep going; else still above avg*fall -> keep going; else the peak is
//            complete (largest one wins, first wins ties) and the sample is re-examined in state 0;
//   state 0: above avg*rise -> enter state 1 with this sample as the running maximum;
//   every sample then updates the IIR average exactly once.
__device__ __forceinline__ void peak_step(float v, int i, int &state, float &peak_val, int &peak_index, int &npk, float &best_val,
                                          int &best_pos, float &avg)
{
  const float rise = 0.8f, fall = 0.9f, alpha = 0.9f;
  if (state == 1) {
    if (v > peak_val) { peak_val = v; peak_index = i; }
    else if (!(v > avg * fall)) {
      if (npk == 0 || peak_val > best_val) { best_val = peak_val; best_pos = peak_index; }
      npk++; state = 0; peak_val = -INFINITY;
    }
  }
  if (state == 0 && v > avg * rise) { state = 1; peak_val = v; peak_index = i; }
  avg = alpha * v + (1 - alpha) * avg;
}

__device__ inline int peak_detect(const float *d, int n, float &avg, int &best_pos)
{
  int state = 0, peak_index = 0, npk = 0;
  float peak_val = -INFINITY, best_val = 0.f;
  int i = 0;
  for (; i + 8 <= n; i += 8) {            // 8 metric values per trip in registers: the recurrence never waits on a load
    float x[8];
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = d[i + k];
#pragma unroll
    for (int k = 0; k < 8; k++) peak_step(x[k], i + k, state, peak_val, peak_index, npk, best_val, best_pos, avg);
  }
  for (; i < n; i++) peak_step(d[i], i, state, peak_val, peak_index, npk, best_val, best_pos, avg);
  return npk;
}

// peak_detect over the 16 lags of a tracking window by a whole wavefront (all lanes in lockstep; lane i < 16 holds lambda[i]).  peak_step walks a state machine
// sample by sample: ~45 instructions each, dependent, ~1.8 us per call for a lone wavefront -- the sequential trackers' whole budget.  Here only what IS a
// recurrence stays one: the IIR average (16 x multiply-add, the reference's expression, lane i keeps the value sample i sees).  The two threshold tests of
// every sample are then one compare per lane and two ballots; the state machine runs on those 16-bit masks: state 0 skips to the next rise bit; state 1 from
// rise sample r ends at the first sample that neither exceeds the running maximum max(v_r .. v_{i-1}) (a prefix maximum over the row: four DPP shifts) nor
// passes the keep test; the completed peak's value and first position come from one more ballot; the sample that completed it is re-examined in state 0
// (its rise bit).  A peak still open at the end of the window is not counted (peak_detect_process records completed peaks only).  Same results as
// peak_detect, bit for bit (tests/test_gpu_blocks.py::test_wave_peak_detector drives both through dvbt_debug_peak_detect).
template <int SH> __device__ __forceinline__ float row_shr_fill(float x, float fill)
{ return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, x), 0x110 | SH, 0xf, 0xf, false)); }
__device__ inline int peak_detect16_wave(float v, int lane, float &avg, int &best_pos)
{
  const float rise = 0.8f, fall = 0.9f, alpha = 0.9f;
  const float tv = alpha * v;
  float a = avg, mine = avg;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    mine = lane == i ? a : mine;                                   // d_avg as sample i sees it
    const float ti = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tv), i));
    a = ti + (1 - alpha) * a;
  }
  avg = a;
  const bool ok = lane < 16;
  const unsigned R = (unsigned)__ballot(ok && v > mine * rise), K = (unsigned)__ballot(ok && v > mine * fall);
  const float NINF = -INFINITY;
  int npk = 0, pos = 0; float best_val = 0.f;
  for (;;) {
    const unsigned m = pos < 16 ? (R & (0xffffu << pos)) : 0u;
    if (!m) break;
    const int r = __ffs((int)m) - 1;
    float x = (ok && lane >= r) ? v : NINF;
    x = fmaxf(x, row_shr_fill<1>(x, NINF)); x = fmaxf(x, row_shr_fill<2>(x, NINF)); x = fmaxf(x, row_shr_fill<4>(x, NINF)); x = fmaxf(x, row_shr_fill<8>(x, NINF));
    const float ex = row_shr_fill<1>(x, NINF);                      // max(v_r .. v_{i-1})
    const unsigned D = (unsigned)__ballot(ok && lane > r && !(v > ex) && !((K >> lane) & 1u));
    if (!D) break;
    const int c = __ffs((int)D) - 1;
    const float pv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), c - 1));
    const unsigned E = (unsigned)__ballot(ok && lane >= r && lane < c && v == pv);
    const int pi = __ffs((int)E) - 1;
    if (npk == 0 || pv > best_val) { best_val = pv; best_pos = pi; }
    npk++; pos = c;
  }
  return npk;
}

// test hook (dvbt_debug_peak_detect): a wavefront per case runs both detectors on the same 16 values and the same carried average
__global__ __launch_bounds__(64) void peak_selftest_kernel(const float *__restrict__ lam, const float *__restrict__ avg_in, int n, int *__restrict__ out, float *__restrict__ avg_out)
{
  const int k = blockIdx.x, lane = threadIdx.x;
  if (k >= n) return;
  float a0 = avg_in[k], a1 = a0; int p0 = 0, p1 = 0;
  const int n0 = peak_detect(lam + (size_t)k * 16, 16, a0, p0);
  const int n1 = peak_detect16_wave(lane < 16 ? lam[(size_t)k * 16 + lane] : 0.f, lane, a1, p1);
  if (lane == 0) { out[4 * k] = n0; out[4 * k + 1] = n0 ? p0 : -1; out[4 * k + 2] = n1; out[4 * k + 3] = n1 ? p1 : -1; avg_out[2 * k] = a0; avg_out[2 * k + 1] = a1; }
}

__device__ __forceinline__ float wrap_pi(double ph)
{
  const double twopi = 6.283185307179586;
  ph = ph - twopi * floor((ph + 3.141592653589793) / twopi);
  return (float)ph;
}

// initial acquisition FSM (general_work :498-510).  Two phases per try:
//  (1) all lanes: the IIR average seen by every sample.  Every sample updates d_avg with the same expression in
//      every FSM state, so avg_i is a plain recursion over lambda; restarted 32 samples back its older history
//      weighs 1e-32, i.e. it reproduces the float value (the first 32 samples start from the carried average and
//      are exact by construction).  From it the two threshold tests of sample i (rise: > 0.8 avg, keep: > 0.9 avg).
//  (2) one lane walks the state machine on the precomputed flags; its only remaining recurrence is the running
//      maximum of the open peak, so a step costs a few cycles instead of a dependent float chain.
// reset (segment path, first launch of a lock period): the flag words of the trackers (trk_flags[0..15]; [8], [9] = first superframe-start candidate, need_seq of the
// TPS bookkeeping), the symbol kernel's ticket and -- for a period that starts the pilot engine afresh -- its state: three memset / copy launches less
struct AcqReset { int *trk_flags; int *ticket; int *tps_state; int tps_state_words; float carry_avg; int has_carry; };   // carry_avg: d_avg carried in from the call that lost the previous lock
__global__ __launch_bounds__(1024) void acq_init_fsm_kernel(FrontParams p, RxState *st, const float2 *gamma, const float *lambda, const AcqState *as,
                                                          int t_begin, int t_end, AcqReset rz = AcqReset{nullptr, nullptr, nullptr, 0, 0.f, 0})
{
  if (t_begin == 0) {
    if (rz.trk_flags && threadIdx.x < 16) rz.trk_flags[threadIdx.x] = threadIdx.x == 8 ? 0x7fffffff : 0;
    if (rz.ticket && threadIdx.x == 16) rz.ticket[0] = 0;
    if (rz.tps_state) for (int i = threadIdx.x; i < rz.tps_state_words; i += blockDim.x) rz.tps_state[i] = 0;
  }
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float *lam = reinterpret_cast<float *>(smem_raw);
  unsigned char *flg = smem_raw + (size_t)p.N * 4;
  __shared__ int s_done, s_pos;
  __shared__ float s_avg;
  __shared__ unsigned long long s_rise[8192 / 64];                  // rise flags of the window, a bit per sample (the walk skips from peak to peak)
  const int tid = threadIdx.x, N = p.N, nthr = blockDim.x;     // 256 (block API) or 1024 threads (segment path: the averages are 32-deep chains, 8 instead of 32 per thread)
  int tries = p.ncalls < t_end ? p.ncalls : t_end;
  if (tid == 0 && t_begin > 0) { s_done = (st->status & 1) ? 0 : 2; s_avg = st->avg; }   // continuation: only if the earlier windows had no peak
  if (tid == 0 && t_begin == 0) {
    st->status = 1; st->call0 = 0; st->cp_start0 = 0; st->n_symbols = 0; st->first_out = -1; st->n_out_symbols = 0;
    st->n_vit_in = st->n_vit_steps = st->n_vit_bytes = st->n_rs_items = st->n_ts_bytes = 0; st->rs_fail = st->rs_corr = 0; st->rs_list_n = 0;
    st->n_rs_words = st->stream_rs_items = st->ts_first_packet = 0; st->tps_bits = 0; st->small_viol = 0; st->drift_known_off = 0;
    s_done = 0; s_avg = as ? as->avg : (rz.has_carry ? rz.carry_avg : 0.f);
    if (as && as->acquired) {          // block API: still locked from the previous work() call, nothing to search
      st->status = 0; st->cp_start0 = as->cp_start; st->eps_init = 0.f; s_done = 2;
    }
  }
  __syncthreads();
  if (s_done == 2) { if (tid == 0) st->avg = s_avg; return; }
  const float rise = 0.8f, fall = 0.9f, alpha = 0.9f;
  constexpr int HIST = 32;
  for (int t = t_begin; t < tries; t++) {
    for (int i = tid; i < N; i += nthr) lam[i] = lambda[(size_t)t * N + i];
    __syncthreads();
    const float avg0 = s_avg;
    for (int i = tid; i <= N; i += nthr) {                         // avg before sample i (i == N: the carried value)
      float avg;
      if (i > HIST) {                                              // fixed-length history: its LDS reads are issued together
        float hbuf[HIST];
#pragma unroll
        for (int k = 0; k < HIST; k++) hbuf[k] = lam[i - HIST + k];
        avg = 0.f;
#pragma unroll
        for (int k = 0; k < HIST; k++) avg = alpha * hbuf[k] + (1 - alpha) * avg;
      } else {
        avg = avg0;
        for (int j = 0; j < i; j++) avg = alpha * lam[j] + (1 - alpha) * avg;
      }
      unsigned f = 0;
      if (i < N) { const float v = lam[i]; f = (v > avg * rise ? 1u : 0u) | (v > avg * fall ? 2u : 0u); flg[i] = (unsigned char)f; }
      else s_avg = avg;
      const unsigned long long rb = __ballot(i < N && (f & 1u));    // the wavefront's 64 samples are consecutive and 64-aligned
      if ((tid & 63) == 0 && i < N) s_rise[i >> 6] = rb;
    }
    __syncthreads();
    if (tid < 64) {
      // the first wavefront walks the state machine 64 samples at a time.  State 0: skip to the next sample with
      // the rise flag.  State 1: the running maximum of the open peak is a prefix maximum over the chunk (carried in
      // from the previous chunk); the peak is complete at the first sample that neither raises the maximum nor
      // passes the keep test; that sample is then re-examined in state 0, exactly as peak_step does.
      int state = 0, peak_index = 0, npk = 0, best_pos = 0, pos = 0;
      float peak_val = -INFINITY, best_val = 0.f;
      while (pos < N) {
        const int i = pos + tid;
        const bool in = i < N;
        const float v = in ? lam[i] : -INFINITY;
        const unsigned f = in ? flg[i] : 2u;
        if (state == 0) {
          // the next sample at or behind pos with the rise flag: straight from the bit words (a window has a handful of peaks; walking its 128 chunks of
          // silence one ballot at a time was most of this kernel's time)
          int cand = 0x7fffffff;
          for (int w0 = 0; w0 < (N >> 6); w0 += 64) {
            const int w = w0 + tid;
            unsigned long long m = w < (N >> 6) ? s_rise[w] : 0ull;
            if ((w << 6) + 63 < pos) m = 0ull; else if ((w << 6) < pos) m &= ~0ull << (pos - (w << 6));
            const unsigned long long any = __ballot(m != 0ull);
            if (any) {
              const int l = __ffsll((long long)any) - 1;
              if (tid == l) s_pos = (w << 6) + __ffsll((long long)m) - 1;
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
              cand = s_pos;
              break;
            }
          }
          if (cand < N) { state = 1; peak_index = cand; peak_val = lam[peak_index]; pos = peak_index + 1; }
          else pos = N;
        } else {
          float pm = v;                                            // inclusive prefix maximum over the lanes
          for (int o = 1; o < 64; o <<= 1) { const float u = __shfl_up(pm, o); if (tid >= o) pm = fmaxf(pm, u); }
          float ex = __shfl_up(pm, 1); if (tid == 0) ex = -INFINITY;
          ex = fmaxf(ex, peak_val);                                // maximum of the open peak before this lane's sample
          const unsigned long long done = __ballot(in && !(v > ex) && !(f & 2u));
          const int l = done ? __ffsll((long long)done) - 1 : 64;  // lanes < l extend the peak
          const float mx = __shfl(l < 64 ? ex : fmaxf(pm, peak_val), l < 64 ? l : 63);
          if (mx > peak_val) {                                     // a new maximum inside the chunk: its first occurrence
            const unsigned long long at = __ballot(tid < l && v == mx);
            peak_index = pos + __ffsll((long long)at) - 1; peak_val = mx;
          }
          if (l < 64) {
            if (npk == 0 || peak_val > best_val) { best_val = peak_val; best_pos = peak_index; }
            npk++; state = 0; peak_val = -INFINITY; pos += l;
          } else pos += 64;
        }
      }
      if (tid == 0 && npk) {
        float2 g = gamma[(size_t)t * N + best_pos];
        st->status = 0; st->call0 = t; st->cp_start0 = best_pos + N + p.cp - 1;
        st->eps_init = atan2f(g.y, g.x); s_done = 1;
      }
    }
    __syncthreads();
    if (s_done) break;
  }
  if (tid == 0) st->avg = s_avg;
}

// tracking FSM over all windows (general_work :512-560 + the phase bookkeeping of ml_sync :285-313).
// Sequential reference version: one thread walks the calls.
struct SymMeta { int cp_start; int sw; float eps; float ph_base; double incA, incB; };
}  // namespace dvbt
#include "k_drift_math.hpp"
namespace dvbt {

// d_phase after one call's N + cp additions: the reference's FLOAT accumulator, reproduced binade by binade (k_drift_math.hpp); the increment switches at
// step `nextpos` when that lies inside the call (:285-309).  The carried value is the accumulator's own (a float), so a call's entry phase is exact.
// The sequential tracker below is ONE wavefront whose lanes all walk the same calls with the same values (uniform control flow, lane 0 stores); the lanes
// split only what is wide: the 15 regions of an increment's table (a division each) and the 16 lags of a call whose window has left the precomputed ones.
// Two tables are kept in LDS (the running increment's and the next one's; a call normally brings one new increment).
__global__ __launch_bounds__(64) void acq_track_kernel(FrontParams p, RxState *st, const float2 *gamma, const float *lambda, SymMeta *meta, const int *need_seq,
                                                       AcqState *as, const int *__restrict__ centre, const float2 *__restrict__ iq)
{
  __shared__ __attribute__((aligned(16))) double s_q[2][16], s_tc[2][18], s_len[16];
  __shared__ __attribute__((aligned(16))) float s_lam[16]; __shared__ __attribute__((aligned(16))) float2 s_gam[16];
  __shared__ __attribute__((aligned(16))) float2 s_xa[ACQ_CP_MAX + 16], s_xb[ACQ_CP_MAX + 16];   // a direct call's samples (guard interval up to 1/4 of 8k)
  const int lane = threadIdx.x;
  if (blockIdx.x != 0) return;
  if (st->status & 1) { if (as && lane == 0) { as->avg = st->avg; as->lost = 0; } return; }
  if (need_seq && *need_seq == 0) return;
  double tab_inc0 = 0.0, tab_inc1 = 0.0; int tab_last = 1;        // which increments the two tables hold, which one was used last (an increment of 0 never asks for one)
  // drift_advance_safe on the shared tables: the unwrapped float phase n steps after phi0
  auto steps = [&](double inc, double phi0, double n) -> double {
    if (inc == 0.0 || n <= 0.0) return phi0;
    if (fabs(inc) < DRIFT_MIN_INC) return phi0 + n * inc;
    int k;
    if (inc == tab_inc0) k = 0;
    else if (inc == tab_inc1) k = 1;
    else {
      k = 1 - tab_last;
      __syncthreads();                                             // readers of the table being replaced are done
      if (lane < DRIFT_NR) {                                       // drift_build, a region per lane: the same expressions
        const double a = fabs(inc), u = drift_ulp(lane);
        const double qq = u > 0.0 ? rint(a / u) * u : a;
        s_q[k][lane] = qq; s_len[lane] = (drift_bnd(lane + 1) - drift_bnd(lane)) / qq;
      }
      __syncthreads();
      if (lane == 0) { double t = 0.0; for (int r = 0; r < DRIFT_NR; r++) { s_tc[k][r] = t; t += s_len[r]; } s_tc[k][DRIFT_NR] = t; }   // summed in drift_build's order
      __syncthreads();
      if (k == 0) tab_inc0 = inc; else tab_inc1 = inc;
    }
    tab_last = k;
    return k == 0 ? drift_Tinv(s_q[0], s_tc[0], inc < 0, drift_T(s_q[0], s_tc[0], inc < 0, phi0) + n)
                  : drift_Tinv(s_q[1], s_tc[1], inc < 0, drift_T(s_q[1], s_tc[1], inc < 0, phi0) + n);
  };
  // d_phase after one call's L additions; the increment switches at step `nextpos` when that lies inside the call (:285-309)
  auto advance = [&](float phase, double inc, double next_inc, int nextpos, int L) -> float {
    double x = (double)phase;
    if (nextpos >= 0 && nextpos < L) x = steps(next_inc, steps(inc, x, (double)nextpos), (double)(L - nextpos));
    else x = steps(inc, x, (double)L);
    return (float)drift_wrap(x);
  };
  const int N = p.N, cp = p.cp, R = p.R, c0 = st->cp_start0, call0 = st->call0;
  const float eps_init = st->eps_init;
  float avg = st->avg, phase = 0.f;
  double phaseinc = 0.0, nextphaseinc = (-1.0 / (double)N) * (double)eps_init;
  int nextpos = c0 - (N + cp), cur = c0, s = 0;
  if (as) {
    const AcqState a0 = *as;
    phase = a0.phase; phaseinc = a0.phaseinc;
    if (a0.acquired) { nextphaseinc = a0.nextphaseinc; nextpos = a0.nextpos; }
    else {
      // the initial ml_sync of this call ran the phase loop once with the carried increments (:285-313)
      phase = advance(phase, a0.phaseinc, a0.nextphaseinc, a0.nextpos, N + cp);
      if (a0.nextpos >= 0 && a0.nextpos < N + cp) phaseinc = a0.nextphaseinc;
    }
  }
  bool lost = false;
  // the 2 R precomputed lags of a call (metric and correlation) and its window centre come through LDS, fetched by the lanes one call ahead: the walk itself
  // then waits for no global load (three dependent ones per call before: 5 us per call, 40 % of a stream's time where the lock is lost every few dozen symbols)
  __shared__ __attribute__((aligned(16))) float s_rl[2][2 * ACQ_R]; __shared__ __attribute__((aligned(16))) float2 s_rg[2][2 * ACQ_R];
  float nl = 0.f; float2 ng = make_float2(0.f, 0.f); int ncen = 0;
  auto fetch_row = [&](int call) {
    if (call < p.ncalls) { ncen = centre[call]; if (lane < 2 * R) { nl = lambda[(size_t)call * 2 * R + lane]; ng = gamma[(size_t)call * 2 * R + lane]; } }
  };
  fetch_row(call0);
  for (int call = call0; call < p.ncalls; call++, s++) {
    const int buf = (call - call0) & 1, cen = ncen;
    __syncthreads();
    if (lane < 2 * R) { s_rl[buf][lane] = nl; s_rg[buf][lane] = ng; }
    __syncthreads();
    fetch_row(call + 1);
    int rel0 = (cur - 8) - (cen - R);
    const bool direct = rel0 < 0 || rel0 + 16 > 2 * R;            // the window left the precomputed lags: this call's metric on the spot, a lag per lane
    if (direct) {
      // the cp + 15 samples the 16 lags share, and the same N samples earlier, through LDS (coalesced loads by all lanes); then a lag per lane, summed in
      // acq_track_metric_kernel's order
      const long long lo = (long long)call * (N + cp) + cur - 8 - (cp - 1);
      __syncthreads();
      for (int t = lane; t < cp + 15; t += 64) {
        s_xa[t] = lo + t >= -p.hist && lo + t < p.avail ? iq[lo + t] : make_float2(0.f, 0.f);
        s_xb[t] = lo - N + t >= -p.hist && lo - N + t < p.avail ? iq[lo - N + t] : make_float2(0.f, 0.f);
      }
      __syncthreads();
      if (lane < 16) {
        const int lag = cur - 8 + lane;
        float gr = 0.f, gi = 0.f, phi = 0.f;
        if ((long long)call * (N + cp) + lag - cp + 1 - N < -p.hist) { s_lam[lane] = -3.0e38f; s_gam[lane] = make_float2(0.f, 0.f); }
        else {
          const float2 *xa = s_xa + (cp - 1) + lane, *xb = s_xb + (cp - 1) + lane;
          for (int j = 0; j < cp; j++) {                            // the expressions and the order of acq_track_metric_kernel
            const float2 a = xa[-j], b = xb[-j];
            gr += a.x * b.x + a.y * b.y; gi += a.y * b.x - a.x * b.y; phi += (a.x * a.x + a.y * a.y) + (b.x * b.x + b.y * b.y);
          }
          s_gam[lane] = make_float2(gr, gi); s_lam[lane] = sqrtf(gr * gr + gi * gi) - phi * p.half_rho;
        }
      }
      __syncthreads();
      rel0 = 0;
    }
    const float *lam = direct ? s_lam : &s_rl[buf][rel0];
    int pos = 0;
    int npk = peak_detect(lam, 16, avg, pos);
    if (!npk) { lost = true; break; }                               // the reference drops lock and re-acquires (:545-559)
    float2 g = direct ? s_gam[pos] : s_rg[buf][rel0 + pos];
    float eps = atan2f(g.y, g.x);
    int peak = pos + cur - 8;
    if (lane == 0) { SymMeta m; m.cp_start = peak; m.eps = eps; m.ph_base = phase; m.incA = phaseinc; m.incB = nextphaseinc; m.sw = nextpos; meta[s] = m; }
    phase = advance(phase, phaseinc, nextphaseinc, nextpos, N + cp);
    if (nextpos >= 0 && nextpos < N + cp) phaseinc = nextphaseinc;
    nextphaseinc = (-1.0 / (double)N) * (double)eps;
    nextpos = peak - (N + cp);
    cur = peak;
  }
  if (as && lost) phase = advance(phase, phaseinc, phaseinc, -1, N + cp);   // the failing call still advanced the phase by N+cp steps without switching (:336-345)
  if (lane != 0) return;
  if (lost) st->status |= 2;
  st->n_symbols = s;
  st->avg_lost = lost ? avg : st->avg;                            // d_avg after the call that lost the lock
  if (as) {
    as->acquired = lost ? 0 : 1; as->lost = lost ? 1 : 0; as->cp_start = cur; as->avg = avg; as->phase = phase;
    as->phaseinc = phaseinc; as->nextphaseinc = nextphaseinc; as->nextpos = nextpos;
  }
}

// The sequential tracker of the SEGMENT path (lock periods on which the Jacobi placement below does not settle: a lock that is being lost, noise that moves the
// peak by several samples from call to call).  One wavefront in lockstep like acq_track_kernel, but it walks POSITIONS AND EPSILON ONLY: peak detector and atan2
// per call, the derotation phase as the exact line in double (sw * incA + (L - sw) * incB per call: acq_finalize_kernel's closed form, accumulated on the way).
// The float accumulator's wander on top of that line is the drift kernels' business (k_drift.hpp), which treat the period exactly as they treat one that the
// parallel placement has settled -- no table of 15 regions per call, no double-precision division in the chain: ~0.5 us per call instead of ~4 (config 5 at
// 8 dB: 18 of 59 ms were this chain).  A phase-increment switch outside its call (never for a tracked peak, see acq_finalize_kernel) hands the period to
// acq_track_kernel through *need_heavy.
__global__ __launch_bounds__(64) void acq_track_light_kernel(FrontParams p, RxState *st, const float2 *gamma, const float *lambda, SymMeta *meta, const int *need_seq,
                                                             int *need_heavy, const int *__restrict__ centre, const float2 *__restrict__ iq)
{
  __shared__ __attribute__((aligned(16))) float s_lam[16]; __shared__ __attribute__((aligned(16))) float2 s_gam[16];
  __shared__ __attribute__((aligned(16))) float2 s_xa[ACQ_CP_MAX + 16], s_xb[ACQ_CP_MAX + 16];
  __shared__ __attribute__((aligned(16))) float s_rl[2][2 * ACQ_R]; __shared__ __attribute__((aligned(16))) float2 s_rg[2][2 * ACQ_R];
  const int lane = threadIdx.x;
  if (blockIdx.x != 0 || (st->status & 1) || *need_seq == 0) return;
  const int N = p.N, cp = p.cp, R = p.R, L = N + cp, c0 = st->cp_start0, call0 = st->call0;
  float avg = st->avg;
  double incA = 0.0, incB = (-1.0 / (double)N) * (double)st->eps_init, base = 0.0;
  int sw = c0 - L, cur = c0, s = 0;
  bool lost = false, viol = false;
  float nl = 0.f; float2 ng = make_float2(0.f, 0.f); int ncen = 0;
  auto fetch_row = [&](int call) {
    if (call < p.ncalls) { ncen = centre[call]; if (lane < 2 * R) { nl = lambda[(size_t)call * 2 * R + lane]; ng = gamma[(size_t)call * 2 * R + lane]; } }
  };
  fetch_row(call0);
  for (int call = call0; call < p.ncalls; call++, s++) {
    const int buf = (call - call0) & 1, cen = ncen;
    __syncthreads();
    if (lane < 2 * R) { s_rl[buf][lane] = nl; s_rg[buf][lane] = ng; }
    __syncthreads();
    fetch_row(call + 1);
    int rel0 = (cur - 8) - (cen - R);
    const bool direct = rel0 < 0 || rel0 + 16 > 2 * R;            // the window left the precomputed lags: this call's metric on the spot, a lag per lane
    if (direct) {
      const long long lo = (long long)call * L + cur - 8 - (cp - 1);
      __syncthreads();
      for (int t = lane; t < cp + 15; t += 64) {
        s_xa[t] = lo + t >= -p.hist && lo + t < p.avail ? iq[lo + t] : make_float2(0.f, 0.f);
        s_xb[t] = lo - N + t >= -p.hist && lo - N + t < p.avail ? iq[lo - N + t] : make_float2(0.f, 0.f);
      }
      __syncthreads();
      if (lane < 16) {
        const int lag = cur - 8 + lane;
        float gr = 0.f, gi = 0.f, phi = 0.f;
        if ((long long)call * L + lag - cp + 1 - N < -p.hist) { s_lam[lane] = -3.0e38f; s_gam[lane] = make_float2(0.f, 0.f); }
        else {
          const float2 *xa = s_xa + (cp - 1) + lane, *xb = s_xb + (cp - 1) + lane;
          for (int j = 0; j < cp; j++) {                            // the expressions and the order of acq_track_metric_kernel
            const float2 a = xa[-j], b = xb[-j];
            gr += a.x * b.x + a.y * b.y; gi += a.y * b.x - a.x * b.y; phi += (a.x * a.x + a.y * a.y) + (b.x * b.x + b.y * b.y);
          }
          s_gam[lane] = make_float2(gr, gi); s_lam[lane] = sqrtf(gr * gr + gi * gi) - phi * p.half_rho;
        }
      }
      __syncthreads();
      rel0 = 0;
    }
    const float *lam = direct ? s_lam : &s_rl[buf][rel0];
    int pos = 0;
    const int npk = peak_detect16_wave(lane < 16 ? lam[lane] : 0.f, lane, avg, pos);
    if (!npk) { lost = true; break; }                               // the reference drops lock and re-acquires (:545-559)
    if (sw < 0 || sw >= L) { viol = true; break; }                  // outside the closed form: the float-faithful tracker takes the period
    const float2 g = direct ? s_gam[pos] : s_rg[buf][rel0 + pos];
    const float eps = atan2f(g.y, g.x);
    const int peak = pos + cur - 8;
    if (lane == 0) { SymMeta m; m.cp_start = peak; m.eps = eps; m.ph_base = wrap_pi(base); m.incA = incA; m.incB = incB; m.sw = sw; meta[s] = m; }
    base += sw * incA + (L - sw) * incB;
    incA = incB; incB = (-1.0 / (double)N) * (double)eps;
    sw = peak - L; cur = peak;
  }
  if (lane != 0) return;
  if (viol) { *need_heavy = 1; return; }
  if (lost) st->status |= 2;
  st->n_symbols = s;
  st->avg_lost = lost ? avg : st->avg;                            // d_avg after the call that lost the lock
}

// ---- the tracker of a SHORT lock period in one launch (the lock-period walk of the synchronous entries: a stream on which the reference's detector drops the
// lock every few dozen symbols, BASELINE config 5 at 8-9 dB).  Everything behind the initial search that the general path spreads over eight launches sized
// for tens of thousands of calls (anchors, centres, the tracking metric of the whole look-ahead window, the Jacobi placement, its bookkeeping, the sequential walk,
// the lost-lock average): ONE workgroup, work proportional to the symbols the lock actually holds.
//   * waves 1 .. cpc/2 compute the tracking metric of the next chunk of `cpc` calls, two calls per wave (half a wave = the 2R lags of a call, a lag per
//     lane): the cp + 2R - 1 samples the lags share and the same N earlier through LDS, the sums in acq_track_metric_kernel's expressions and order
//     (bit-identical lambda / gamma).  The chunk's lags are centred on the peak the walk had reached when the chunk was started: a drifting CP position is
//     followed chunk by chunk, no anchors;
//   * wave 0 walks the chunk computed before (acq_track_light_kernel's walk: peak detector, atan2, the phase as the exact line), the two overlap;
//   * a window that has left its chunk's lags is computed on the spot by wave 0 (a lag per lane, from global memory).
// Writes what the general path writes: meta[], st->n_symbols / status bit 1 / avg_lost.  st->small_viol = 1: a switch position outside its call -- the host
// runs the general path on the period instead.
constexpr int ACQ_SMALL_MAX_CALLS = 768;
constexpr size_t ACQ_SMALL_LDS = 96 * 1024;
inline int acq_small_cpc(int cp) { int c = (int)(ACQ_SMALL_LDS / ((size_t)(cp + 2 * ACQ_R) * 16)); c &= ~1; return c > 16 ? 16 : (c < 2 ? 2 : c); }
__global__ __launch_bounds__(576) void acq_small_kernel(const float2 *__restrict__ iq, FrontParams p, RxState *st, SymMeta *__restrict__ meta, int cpc, int *drift_flags)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ __attribute__((aligned(16))) float s_lam[2][16][2 * ACQ_R]; __shared__ __attribute__((aligned(16))) float2 s_gam[2][16][2 * ACQ_R];
  __shared__ __attribute__((aligned(16))) float s_dl[16]; __shared__ __attribute__((aligned(16))) float2 s_dg[16];
  __shared__ int s_cen[2], s_next_cen, s_stop;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (st->status & 1) return;
  const int N = p.N, cp = p.cp, R = p.R, L = N + cp, c0 = st->cp_start0, call0 = st->call0;
  const int span = cp + 2 * R - 1, stride = cp + 2 * R;             // samples a call's lags share
  float2 *stage = reinterpret_cast<float2 *>(smem_raw);             // [cpc][2][stride]
  if (tid == 0) { s_next_cen = c0; s_stop = 0; st->small_viol = 0; }
  __syncthreads();
  // metric of chunk k (calls call0 + k cpc ...) into buffer k & 1, lags cen - R .. cen + R - 1 of every call.  The products x[i] conj(x[i-N]) and the energy pairs
  // of the cp + 2R - 1 samples a call's lags share are formed ONCE while staging (the same float expressions acq_track_metric_kernel evaluates per tap: the
  // sums below add the same values in the same order); a lane then adds its lag's cp terms.  (Forming them per lag and tap made this the kernel's
  // critical path: 14 operations per tap and lane instead of 3.)
  auto metric_chunk = [&](int k, int cen) {
    const int slot = 2 * (wave - 1) + (lane >> 5), q = lane & 31, call = call0 + k * cpc + slot;
    if (wave < 1 || slot >= cpc || call >= p.ncalls) return;
    float2 *sc = stage + (size_t)slot * 2 * stride; float *se = reinterpret_cast<float *>(sc + stride);
    const long long lo = (long long)call * L + cen - R - (cp - 1);  // sample of sc[0]
    for (int t0 = q; t0 < span; t0 += 32 * 5) {                     // ten loads per lane in flight (cp = 256: two trips)
      float2 ra[5], rb[5];
#pragma unroll
      for (int u = 0; u < 5; u++) {
        const int t = t0 + 32 * u; const bool in = t < span;
        ra[u] = in && lo + t >= -p.hist && lo + t < p.avail ? iq[lo + t] : make_float2(0.f, 0.f);
        rb[u] = in && lo - N + t >= -p.hist && lo - N + t < p.avail ? iq[lo - N + t] : make_float2(0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 5; u++) {
        const int t = t0 + 32 * u;
        if (t < span) {
          const float2 a = ra[u], b = rb[u];
          sc[t] = make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
          se[t] = (a.x * a.x + a.y * a.y) + (b.x * b.x + b.y * b.y);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // the half wave's own staging only
    float gr = 0.f, gi = 0.f, phi = 0.f;
    const float2 *xc = sc + (cp - 1) + q; const float *xe = se + (cp - 1) + q;
    int j = 0;
    for (; j + 8 <= cp; j += 8) {
      float2 cv[8]; float ev[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { cv[u] = xc[-(j + u)]; ev[u] = xe[-(j + u)]; }
#pragma unroll
      for (int u = 0; u < 8; u++) { gr += cv[u].x; gi += cv[u].y; phi += ev[u]; }
    }
    for (; j < cp; j++) { gr += xc[-j].x; gi += xc[-j].y; phi += xe[-j]; }
    const bool before = (long long)call * L + (cen - R + q) - cp + 1 - N < -p.hist;   // the lag reaches in front of the stream's first sample
    s_lam[k & 1][slot][q] = before ? -3.0e38f : sqrtf(gr * gr + gi * gi) - phi * p.half_rho;
    s_gam[k & 1][slot][q] = before ? make_float2(0.f, 0.f) : make_float2(gr, gi);
  };
  if (tid == 0) s_cen[0] = c0;
  metric_chunk(0, c0);
  __syncthreads();
  // walk state (wave 0, all lanes alike)
  float avg = st->avg;
  double incA = 0.0, incB = (-1.0 / (double)N) * (double)st->eps_init, base = 0.0;
  int sw = c0 - L, cur = c0, s = 0, dfl = 0;                        // dfl: drift_prep_kernel's eligibility bits over the period's increments
  bool lost = false, viol = false;
  for (int k = 0;; k++) {
    const int cb = call0 + k * cpc;
    if (cb >= p.ncalls) break;
    if (wave == 0) {
      if (lane == 0) { s_next_cen = cur; s_cen[(k + 1) & 1] = cur; }                // the next chunk's lags: around the peak reached so far
    }
    __syncthreads();
    if (wave > 0) metric_chunk(k + 1, s_next_cen);
    else {
      const int cen = s_cen[k & 1];
      for (int c = 0; c < cpc && cb + c < p.ncalls; c++, s++) {
        const int call = cb + c;
        int rel0 = (cur - 8) - (cen - R);
        const bool direct = rel0 < 0 || rel0 + 16 > 2 * R;
        if (direct) {
          if (lane < 16) {
            const int lag = cur - 8 + lane;
            const long long at = (long long)call * L + lag;
            float gr = 0.f, gi = 0.f, phi = 0.f;
            if (at - cp + 1 - N < -p.hist) { s_dl[lane] = -3.0e38f; s_dg[lane] = make_float2(0.f, 0.f); }
            else {
              for (int j = 0; j < cp; j++) {
                const float2 a = at - j < p.avail ? iq[at - j] : make_float2(0.f, 0.f), b = at - j - N < p.avail ? iq[at - j - N] : make_float2(0.f, 0.f);
                gr += a.x * b.x + a.y * b.y; gi += a.y * b.x - a.x * b.y; phi += (a.x * a.x + a.y * a.y) + (b.x * b.x + b.y * b.y);
              }
              s_dg[lane] = make_float2(gr, gi); s_dl[lane] = sqrtf(gr * gr + gi * gi) - phi * p.half_rho;
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          rel0 = 0;
        }
        const float *lam = direct ? s_dl : &s_lam[k & 1][c][rel0];
        int pos = 0;
        const int npk = peak_detect16_wave(lane < 16 ? lam[lane] : 0.f, lane, avg, pos);
        if (!npk) { lost = true; break; }
        if (sw < 0 || sw >= L) { viol = true; break; }
        const float2 g = direct ? s_dg[pos] : s_gam[k & 1][c][rel0 + pos];
        const float eps = atan2f(g.y, g.x);
        const int peak = pos + cur - 8;
        if (lane == 0) { SymMeta m; m.cp_start = peak; m.eps = eps; m.ph_base = wrap_pi(base); m.incA = incA; m.incB = incB; m.sw = sw; meta[s] = m; }
        dfl |= (incB > 0 ? 1 : (incB < 0 ? 2 : 4)) | (fabs(incB) < DRIFT_MIN_INC ? 4 : 0);
        base += sw * incA + (L - sw) * incB;
        incA = incB; incB = (-1.0 / (double)N) * (double)eps;
        sw = peak - L; cur = peak;
        if (direct) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }   // s_dl / s_dg are rewritten by the next direct call
      }
      if ((lost || viol) && lane == 0) s_stop = 1;
    }
    __syncthreads();
    if (s_stop) break;
  }
  if (tid != 0) return;
  if (viol) { st->small_viol = 1; return; }
  if (lost) st->status |= 2;
  st->n_symbols = s;
  st->avg_lost = lost ? avg : st->avg;
  // drift_exact_kernel's decision, taken here for the host: `on` needs >= 2 symbols and increments of one sign, none below the model's floor
  if (!(s >= 2 && (dfl == 1 || dfl == 2))) { st->drift_known_off = 1; drift_flags[1] = 0; }
}

// ---- parallel tracking.  Two facts make the per-call FSM independent of its predecessors:
//  (1) every sample of a window updates d_avg with the same expression whatever the FSM state,
//      so d_avg entering call j is a plain IIR (x0.1 per sample) over the earlier windows; after
//      the 32 samples of the two previous calls the older history weighs 1e-32 -- far below half
//      an ulp -- so restarting the IIR two calls back reproduces the float value;
//  (2) the window position only depends on the previous call's peak.
// Jacobi iteration on (2): iteration k places call j's window with iteration k-1's peak of call
// j-1 (iteration 0: cp_start0 everywhere) until nothing changes; the fixed point is the
// sequential trajectory by induction from call0.  If it has not converged after the last
// iteration the sequential kernel above is run instead (flag checked on the device).
struct TrackWork { int *cp_a; int *cp_b; float *eps; int *changed; /* [iters+1] */ };

__global__ __launch_bounds__(256) void acq_track_par_kernel(FrontParams p, const RxState *st, const float2 *__restrict__ gamma,
                                                           const float *__restrict__ lambda, const int *__restrict__ cp_in,
                                                           int *__restrict__ cp_out, float *__restrict__ eps_out, int *changed, int iter,
                                                           const int *__restrict__ centre)
{
  if (st->status & 1) return;
  if (iter > 0 && changed[iter - 1] == 0) return;               // already at the fixed point
  const int s = blockIdx.x * 256 + threadIdx.x;
  const int call = st->call0 + s;
  if (call >= p.ncalls) return;
  const int R = p.R, c0 = st->cp_start0;
  auto prev_cp = [&](int ss) -> int { return ss <= 0 ? c0 : (iter == 0 ? c0 : cp_in[ss - 1]); };   // peak of call ss-1
  float avg;
  int first_warm;
  if (s <= 2) { avg = st->avg; first_warm = 0; } else { avg = 0.f; first_warm = s - 2; }
  bool bad = false;
  for (int ws = first_warm; ws < s; ws++) {                      // IIR over the earlier windows (fact 1)
    int cur = prev_cp(ws);
    if (cur < 0) { bad = true; break; }
    int rel0 = (cur - 8) - (centre[st->call0 + ws] - R);
    if (rel0 < 0 || rel0 + 16 > 2 * R) { bad = true; break; }
    const float *lam = lambda + (size_t)(st->call0 + ws) * 2 * R + rel0;
    for (int i = 0; i < 16; i++) avg = 0.9f * lam[i] + (1 - 0.9f) * avg;
  }
  int cur = prev_cp(s), res = bad ? -2 : -1; float eps = 0.f;
  if (!bad && cur >= 0) {
    int rel0 = (cur - 8) - (centre[call] - R);
    if (rel0 < 0 || rel0 + 16 > 2 * R) res = -2;                // left the precomputed lags: the sequential tracker takes over
    else {
      float lam[16];
      const float *lp = lambda + (size_t)call * 2 * R + rel0;
      for (int i = 0; i < 16; i++) lam[i] = lp[i];
      int pos = 0;
      int npk = peak_detect(lam, 16, avg, pos);
      if (npk) { res = pos + cur - 8; float2 g = gamma[(size_t)call * 2 * R + rel0 + pos]; eps = atan2f(g.y, g.x); }
    }
  }
  cp_out[s] = res; eps_out[s] = eps;
  int old = iter == 0 ? c0 : cp_in[s];
  if (res != old) atomicOr(&changed[iter], 1);
}

// The four Jacobi rounds in ONE launch.  A round reads the previous round's peaks of the three calls in front of a call, so a workgroup that recomputes
// 12 calls in front of its own 244 (TRK_HALO = 3 per round) needs nothing from its neighbours: thread t of workgroup b works on call b * 244 - 12 + t, its
// round-k value is the global one whenever t >= 3 k.  Same results and the same `changed` flags as four launches of acq_track_par_kernel (rounds behind
// the fixed point reproduce it); the last round's peaks go to cp_out.
constexpr int TRK_HALO = 12, TRK_OWN = 256 - TRK_HALO, TRK_ROUNDS = 4;
__global__ __launch_bounds__(256) void acq_track_fused_kernel(FrontParams p, const RxState *st, const float2 *__restrict__ gamma,
                                                             const float *__restrict__ lambda, int *__restrict__ cp_out, float *__restrict__ eps_out,
                                                             int *changed, const int *__restrict__ centre)
{
  __shared__ int s_cp[2][256];
  if (st->status & 1) return;
  const int t = threadIdx.x, s = (int)blockIdx.x * TRK_OWN - TRK_HALO + t;
  const int call0 = st->call0, call = call0 + s, R = p.R, c0 = st->cp_start0;
  const bool live = s >= 0 && call < p.ncalls, own = live && (t >= TRK_HALO);
  int res = c0; float eps = 0.f;
  for (int iter = 0; iter < TRK_ROUNDS; iter++) {
    const int *cin = s_cp[(iter + 1) & 1];
    auto prev_cp = [&](int ss) -> int {                            // peak of call ss - 1 in the previous round
      if (ss <= 0 || iter == 0) return c0;
      const int tt = ss - 1 - ((int)blockIdx.x * TRK_OWN - TRK_HALO);
      return tt >= 0 ? cin[tt] : c0;                               // (tt < 0 only for threads outside the trusted zone)
    };
    const int old = res;
    if (live) {
      float avg; int first_warm;
      if (s <= 2) { avg = st->avg; first_warm = 0; } else { avg = 0.f; first_warm = s - 2; }
      bool bad = false;
      for (int ws = first_warm; ws < s; ws++) {                    // IIR over the earlier windows (fact 1)
        const int cur = prev_cp(ws);
        if (cur < 0) { bad = true; break; }
        const int rel0 = (cur - 8) - (centre[call0 + ws] - R);
        if (rel0 < 0 || rel0 + 16 > 2 * R) { bad = true; break; }
        const float *lam = lambda + (size_t)(call0 + ws) * 2 * R + rel0;
        for (int i = 0; i < 16; i++) avg = 0.9f * lam[i] + (1 - 0.9f) * avg;
      }
      const int cur = prev_cp(s);
      res = bad ? -2 : -1; eps = 0.f;
      if (!bad && cur >= 0) {
        const int rel0 = (cur - 8) - (centre[call] - R);
        if (rel0 < 0 || rel0 + 16 > 2 * R) res = -2;              // left the precomputed lags: the sequential tracker takes over
        else {
          float lam[16];
          const float *lp = lambda + (size_t)call * 2 * R + rel0;
          for (int i = 0; i < 16; i++) lam[i] = lp[i];
          int pos = 0;
          const int npk = peak_detect(lam, 16, avg, pos);
          if (npk) { res = pos + cur - 8; const float2 g = gamma[(size_t)call * 2 * R + rel0 + pos]; eps = atan2f(g.y, g.x); }
        }
      }
    }
    s_cp[iter & 1][t] = res;
    if (own && res != old) atomicOr(&changed[iter], 1);
    __syncthreads();
  }
  if (own) { cp_out[s] = res; eps_out[s] = eps; }
}

// d_avg after the call in which the tracker lost the lock: the reference re-acquires in the next call with this value
// (ofdm_sym_acquisition_impl.cc:545-559; d_avg persists).  IIR over the last two windows (see acq_track_par_kernel).
__device__ inline void acq_lost_avg_body(FrontParams p, RxState *st, const int *__restrict__ cp, const float *__restrict__ lambda, const int *__restrict__ centre)
{
  st->avg_lost = st->avg;
  if (!(st->status & 2) || (st->status & 1)) return;
  const int f = st->n_symbols, R = p.R, c0 = st->cp_start0;      // f: first call (relative to call0) without a peak
  float avg; int w0;
  if (f <= 1) { avg = st->avg; w0 = 0; } else { avg = 0.f; w0 = f - 1; }
  for (int ws = w0; ws <= f; ws++) {
    const int cur = ws <= 0 ? c0 : cp[ws - 1];
    if (cur < 0) return;
    const int rel0 = (cur - 8) - (centre[st->call0 + ws] - R);
    if (rel0 < 0 || rel0 + 16 > 2 * R) return;
    const float *lam = lambda + (size_t)(st->call0 + ws) * 2 * R + rel0;
    for (int i = 0; i < 16; i++) avg = 0.9f * lam[i] + (1 - 0.9f) * avg;
  }
  st->avg_lost = avg;
}

// bookkeeping after the fixed point: n_symbols = calls before the first miss; derotation phase
// parameters per symbol (ml_sync :285-313).  Falls back to nothing when the Jacobi iteration did not converge
// (acq_track_kernel then overwrites everything).  A workgroup owns 1024 consecutive calls (a call per thread: coalesced loads, SymMeta
// records stored side by side); the checks and the phase accumulated in front of its calls (a sum in double over the per-call advance) it forms
// for itself from ALL calls -- a few hundred KB out of L2 -- so the workgroups need nothing from each other.  (One workgroup with a thread per
// run of consecutive calls: 51 us on 17,680 calls, most of it strided loads and stores on one compute unit.)  Workgroup 0 reports, and when the
// placement stands it also leaves d_avg as the reference holds it after a lost lock (acq_lost_avg_body; the sequential trackers write it themselves).
__global__ __launch_bounds__(1024) void acq_finalize_kernel(FrontParams p, RxState *st, const int *__restrict__ cp, const float *__restrict__ eps,
                                                           const int *changed, int last_iter, SymMeta *__restrict__ meta, int *need_seq,
                                                           const float *__restrict__ lambda, const int *__restrict__ centre)
{
  __shared__ double s_w[2][16];
  __shared__ int s_first_bad, s_viol;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const bool lead = blockIdx.x == 0 && tid == 0;
  if (st->status & 1) { if (lead) { *need_seq = 0; acq_lost_avg_body(p, st, cp, lambda, centre); } return; }
  bool converged = false;
  for (int i = 0; i <= last_iter; i++) if (changed[i] == 0) converged = true;
  if (!converged) { if (lead) *need_seq = 1; return; }
  const int ntot = p.ncalls - st->call0, N = p.N, L = p.N + p.cp, c0 = st->cp_start0, base = blockIdx.x * 1024;
  if (base >= ntot && blockIdx.x != 0) return;
  const float eps_init = st->eps_init;
  if (tid == 0) { s_first_bad = ntot; s_viol = ntot; }
  __syncthreads();
  // per-call quantities: entering call s: nextpos = cp[s-1]-(N+cp), incB = -eps[s-1]/N, incA = -eps[s-2]/N (cp[-1] = the initial position, eps[-1] = the
  // initial estimate, eps[-2] = 0); the call advances the phase by sw * incA + (L - sw) * incB
  // (valid when every switch position lies inside the call, i.e. N+cp <= cp < 2N+2cp: always true
  //  for a tracked peak, which lives in [N+cp-1+8, 2N+cp-2]; anything else hands the period to the sequential tracker)
  auto advance = [&](int s, int cprev, float e2, float e1, int &sw, double &A, double &B) -> double {
    sw = cprev - L;
    A = s == 0 ? 0.0 : (-1.0 / N) * (double)e2; B = (-1.0 / N) * (double)e1;
    return (sw >= 0 && sw < L) ? sw * A + (L - sw) * B : (double)L * A;
  };
  // one pass over all calls, 4 clamped loads of each array per lane in flight: first call without a peak, first switch outside its call, phase in front of `base`
  double front = 0.0;
  for (int s0 = tid; s0 < ntot; s0 += 4 * 1024) {
    int cc[4], cm[4]; float e1[4], e2[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int s = s0 + k * 1024, sc = s < ntot ? s : ntot - 1;
      cc[k] = cp[sc]; cm[k] = sc >= 1 ? cp[sc - 1] : c0; e1[k] = sc >= 1 ? eps[sc - 1] : eps_init; e2[k] = sc >= 2 ? eps[sc - 2] : (sc == 1 ? eps_init : 0.f);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int s = s0 + k * 1024;
      if (s < ntot) {
        if (cc[k] < 0) atomicMin(&s_first_bad, s);
        int sw; double A, B;
        const double adv = advance(s, cm[k], e2[k], e1[k], sw, A, B);
        if (sw < 0 || sw >= L) atomicMin(&s_viol, s);
        if (s < base) front += adv;                               // only used when base < n_symbols: every call in front of it is a tracked one
      }
    }
  }
  __syncthreads();
  const int nsym = s_first_bad;
  if (nsym < ntot && cp[nsym] == -2) { if (lead) *need_seq = 1; return; }      // the placement ran out of precomputed lags: sequential tracker
  if (s_viol < nsym) { if (lead) *need_seq = 1; return; }                      // the closed form needs every phase-increment switch to fall inside its call
  if (lead) { *need_seq = 0; st->n_symbols = nsym; if (nsym < ntot) st->status |= 2; acq_lost_avg_body(p, st, cp, lambda, centre); }
  if (base >= nsym) return;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) front += __shfl_xor(front, o);
  if (lane == 0) s_w[0][wv] = front;
  // this workgroup's calls
  const int s = base + tid;
  const bool own = s < nsym;
  int sw = 0, cs = 0; float es = 0.f; double A = 0.0, B = 0.0, adv = 0.0;
  if (own) {
    cs = cp[s]; es = eps[s];
    adv = advance(s, s >= 1 ? cp[s - 1] : c0, s >= 2 ? eps[s - 2] : (s == 1 ? eps_init : 0.f), s >= 1 ? eps[s - 1] : eps_init, sw, A, B);
  }
  double incl = adv;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const double u = __shfl_up(incl, o); if (lane >= o) incl += u; }
  double excl = __shfl_up(incl, 1); if (lane == 0) excl = 0.0;
  if (lane == 63) s_w[1][wv] = incl;
  __syncthreads();
  double before = 0.0;
  for (int w = 0; w < 16; w++) before += s_w[0][w];
  for (int w = 0; w < wv; w++) before += s_w[1][w];
  if (own) {
    SymMeta m; m.cp_start = cs; m.eps = es; m.sw = sw; m.incA = A; m.incB = B; m.ph_base = wrap_pi(before + excl);
    meta[s] = m;
  }
}


// LDS image of a symbol: one float2 of padding after every 32 keeps the stride-4/16/64 accesses of the
// late radix-4 stages conflict-free (bank = float2 index mod 32 for ds_read_b64)
#ifndef DVBT_FFT_THREADS
#define DVBT_FFT_THREADS 512
#endif
constexpr int FFT_THREADS = DVBT_FFT_THREADS;   // workgroup size of the FFT kernels (a build-time experiment knob)
__device__ __forceinline__ int fpad(int a) { return a + (a >> 5); }
// twiddle W_N^t from a two-level table held in LDS: coarse[t >> 7] * fine[t & 127] (one complex multiply instead of
// an L2 round trip per twiddle; relative error ~1.2e-7)
__device__ __forceinline__ float2 twid(const float2 *coarse, const float2 *fine, int t) { return cmul(coarse[t >> 7], fine[t & 127]); }

// LDS position that holds bin k after fft_dif_lds: digit reversal over the pass radices (16.. then 4 and/or 2)
__device__ __forceinline__ int fft_pos_of_bin(int k, int N)
{
  int L = N, p = 0;
  while (L >= 16) { L >>= 4; p += (k & 15) * L; k >>= 4; }
  if (L >= 4) { L >>= 2; p += (k & 3) * L; k >>= 2; }
  if (L == 2) p += k & 1;
  return p;
}

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// forward radix-4 butterfly: y_k = sum_j a_j (-i)^(jk)
__device__ __forceinline__ void bfly4(float2 a0, float2 a1, float2 a2, float2 a3, float2 &y0, float2 &y1, float2 &y2, float2 &y3)
{
  const float2 s02 = cadd(a0, a2), d02 = csub(a0, a2), s13 = cadd(a1, a3), d13 = csub(a1, a3);
  y0 = cadd(s02, s13); y2 = csub(s02, s13);
  y1 = make_float2(d02.x + d13.y, d02.y - d13.x);               // d02 - i*d13
  y3 = make_float2(d02.x - d13.y, d02.y + d13.x);               // d02 + i*d13
}

// 16-point forward DFT in registers (two radix-4 levels): a[j], j = time index, becomes A[k], k = frequency index
__device__ __forceinline__ void dft16(float2 (&a)[16])
{
  // W16^m = exp(-2 pi i m / 16), m = 0..9 (products r'*k1 with r',k1 <= 3)
  const float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, h = 0.70710678118654752f;
  const float2 w16[10] = {{1.f, 0.f}, {c1, -s1}, {h, -h}, {s1, -c1}, {0.f, -1.f}, {-s1, -c1}, {-h, -h}, {-c1, -s1}, {-1.f, 0.f}, {-c1, s1}};
  float2 t[16];
#pragma unroll
  for (int rp = 0; rp < 4; rp++) {                              // level 1: span 16
    float2 y0, y1, y2, y3;
    bfly4(a[rp], a[rp + 4], a[rp + 8], a[rp + 12], y0, y1, y2, y3);
    t[0 * 4 + rp] = y0;
    t[1 * 4 + rp] = rp ? cmul(y1, w16[rp]) : y1;
    t[2 * 4 + rp] = rp ? cmul(y2, w16[2 * rp]) : y2;
    t[3 * 4 + rp] = rp ? cmul(y3, w16[3 * rp]) : y3;
  }
#pragma unroll
  for (int k1 = 0; k1 < 4; k1++) {                              // level 2: span 4 -> A[k1 + 4*k2]
    float2 y0, y1, y2, y3;
    bfly4(t[k1 * 4], t[k1 * 4 + 1], t[k1 * 4 + 2], t[k1 * 4 + 3], y0, y1, y2, y3);
    a[k1] = y0; a[k1 + 4] = y1; a[k1 + 8] = y2; a[k1 + 12] = y3;
  }
}
// a[k] *= w1^k, k = 1..15: the powers by products, at most 4 multiplications deep
__device__ __forceinline__ void twiddle16(float2 (&a)[16], float2 w1)
{
  float2 w[16];
  w[1] = w1;
  w[2] = cmul(w[1], w[1]); w[3] = cmul(w[2], w[1]); w[4] = cmul(w[2], w[2]); w[5] = cmul(w[4], w[1]); w[6] = cmul(w[3], w[3]);
  w[7] = cmul(w[4], w[3]); w[8] = cmul(w[4], w[4]); w[9] = cmul(w[8], w[1]); w[10] = cmul(w[5], w[5]); w[11] = cmul(w[8], w[3]);
  w[12] = cmul(w[6], w[6]); w[13] = cmul(w[8], w[5]); w[14] = cmul(w[7], w[7]); w[15] = cmul(w[8], w[7]);
#pragma unroll
  for (int k = 1; k < 16; k++) a[k] = cmul(a[k], w[k]);
}

// In-place decimation-in-frequency FFT of the padded LDS image x (N points, natural order in, digit-reversed
// out; the host-built permutation undoes the reversal).  Radix-16 passes with the 16-point transform held in
// registers (two radix-4 levels), then a radix-4 and/or radix-2 tail: 8192 = 16.16.16.2, 2048 = 16.16.4.2
// (dvbt_tables.hpp::fft_radices lists the same sequence).  One __syncthreads per pass.
__device__ __forceinline__ void fft_dif_lds(float2 *x, int N, const float2 *tw_c, const float2 *tw_f, int tid)
{
  int L = N;
  while (L >= 16) {
    const int Q = L >> 4, tstep = N / L, nblk = (N >> 4) / Q;
    for (int bf = tid; bf < (N >> 4); bf += FFT_THREADS) {
      // butterfly bf = (block, r): consecutive lanes take consecutive r (conflict-free LDS columns); when only two r exist
      // (last radix-16 pass of 8k) consecutive lanes take consecutive blocks instead (stride 33 in the padded image).
      const int r = Q <= 2 ? bf / nblk : bf % Q, base = (Q <= 2 ? bf % nblk : bf / Q) * L + r;
      // padded address of element base + j Q: fpad is affine in j here (base % 32 = r < Q for Q < 32, and j Q % 32 = 0 otherwise)
      const int pb = fpad(base);
      float2 a[16];
#pragma unroll
      for (int j = 0; j < 16; j++) a[j] = x[pb + j * Q + ((j * Q) >> 5)];
      dft16(a);
      twiddle16(a, twid(tw_c, tw_f, r * tstep));       // inter-pass twiddles W^(r k tstep), k = 1..15
#pragma unroll
      for (int k = 0; k < 16; k++) x[pb + k * Q + ((k * Q) >> 5)] = a[k];
    }
    __syncthreads();
    L = Q;
  }
  if (L >= 4) {
    const int Q = L >> 2, tstep = N / L;
    for (int bf = tid; bf < (N >> 2); bf += FFT_THREADS) {
      const int r = bf % Q, base = (bf / Q) * L + r;
      const int i0 = fpad(base), i1 = fpad(base + Q), i2 = fpad(base + 2 * Q), i3 = fpad(base + 3 * Q);
      float2 y0, y1, y2, y3;
      bfly4(x[i0], x[i1], x[i2], x[i3], y0, y1, y2, y3);
      x[i0] = y0;
      if (r == 0) { x[i1] = y1; x[i2] = y2; x[i3] = y3; }
      else { x[i1] = cmul(y1, twid(tw_c, tw_f, r * tstep)); x[i2] = cmul(y2, twid(tw_c, tw_f, 2 * r * tstep)); x[i3] = cmul(y3, twid(tw_c, tw_f, 3 * r * tstep)); }
    }
    __syncthreads();
    L = Q;
  }
  if (L == 2) {
    for (int bf = tid; bf < (N >> 1); bf += FFT_THREADS) {
      const int j0 = fpad(2 * bf), j1 = fpad(2 * bf + 1);
      const float2 a0 = x[j0], a1 = x[j1];
      x[j0] = cadd(a0, a1); x[j1] = csub(a0, a1);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------- A1 tail + A2: derotate, strip CP, forward FFT with shift
// One workgroup per OFDM symbol; the symbol lives in LDS (N*8 bytes) through all stages.
// DIF radix-4 stages (+ one radix-2 when log2 N is odd), in place; the digit-reversed result is
// written out in natural, fft-shifted order through the host-built permutation.
__global__ __launch_bounds__(FFT_THREADS) void derot_fft_kernel(const float2 *__restrict__ iq, FrontParams p, const RxState *st,
                                                       const SymMeta *__restrict__ meta, const float2 *__restrict__ tw,
                                                       const uint16_t *__restrict__ perm, float2 *__restrict__ acq_tap,
                                                       float2 *__restrict__ out, const float *__restrict__ drift = nullptr, const int *__restrict__ drift_flags = nullptr)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float2 *x = reinterpret_cast<float2 *>(smem_raw);
  const int s = blockIdx.x;
  if (s >= st->n_symbols) return;
  const int N = p.N, cp = p.cp, tid = threadIdx.x;
  float2 *tw_c = x + (N + N / 32), *tw_f = tw_c + N / 128;
  if (out) { for (int i = tid; i < N / 128; i += FFT_THREADS) tw_c[i] = tw[i * 128]; if (tid < 128) tw_f[tid] = tw[tid]; }   // N >= 2048 here
  const SymMeta m = meta[s];
  const long long low = (long long)(st->call0 + s) * (N + cp) + m.cp_start - N + 1;
  const bool rot = (m.incA != 0.0) || (m.incB != 0.0) || (m.ph_base != 0.f);
  for (int n = tid; n < N; n += FFT_THREADS) {
    float2 v = iq[low + n];
    if (rot) {                                   // derot[n] = expj(phase after n+1 increments)  (:285-309,:527-534)
      int a = n + 1, b = 0;
      if (m.sw >= 0 && m.sw < N + cp && a > m.sw) { b = a - m.sw; a = m.sw; }
      float ph = wrap_pi((double)m.ph_base + a * m.incA + b * m.incB);
      float sn, cs; sincosf(ph, &sn, &cs);
      v = cmul(make_float2(cs, sn), v);
      if (drift && drift_flags[1]) { const float dl = drift[(size_t)s * (N / 32) + (n >> 5)]; v = make_float2(v.x - dl * v.y, v.y + dl * v.x); }   // k_drift.hpp
    }
    x[fpad(n)] = v;
    if (acq_tap) acq_tap[(size_t)s * N + n] = v;
  }
  if (!out) return;                              // A1 alone (block API): derotated, CP-stripped item only
  __syncthreads();
  fft_dif_lds(x, N, tw_c, tw_f, tid);
  float2 *o = out + (size_t)s * N;
  for (int b = tid; b < N; b += FFT_THREADS) o[b] = x[fpad(fft_pos_of_bin((b + (N >> 1)) & (N - 1), N))];   // shifted: out[b] = X[(b - N/2) mod N]
}

// plain FFT for the standalone A2 block: items already CP-stripped
__global__ __launch_bounds__(FFT_THREADS) void fft_items_kernel(const float2 *__restrict__ in, int N, int nitems,
                                                       const float2 *__restrict__ tw, const uint16_t *__restrict__ perm,
                                                       float2 *__restrict__ out)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float2 *x = reinterpret_cast<float2 *>(smem_raw);
  const int s = blockIdx.x, tid = threadIdx.x;
  if (s >= nitems) return;
  const int nc = N >= 128 ? N / 128 : 1;
  float2 *tw_c = x + (N + N / 32), *tw_f = tw_c + nc;
  for (int i = tid; i < nc; i += FFT_THREADS) tw_c[i] = tw[i * 128 < N ? i * 128 : 0];
  if (tid < 128 && tid < N) tw_f[tid] = tw[tid];
  for (int n = tid; n < N; n += FFT_THREADS) x[fpad(n)] = in[(size_t)s * N + n];
  __syncthreads();
  fft_dif_lds(x, N, tw_c, tw_f, tid);
  for (int b = tid; b < N; b += FFT_THREADS) out[(size_t)s * N + b] = x[fpad(fft_pos_of_bin((b + (N >> 1)) & (N - 1), N))];
}

// ---------------------------------------------------------------- A3: pilot engine, one workgroup per OFDM symbol
struct DemodTables {
  const int16_t *cpilot;        // n_cp
  const float *known_diff;      // n_cp-1  |ref(c[j+1])-ref(c[j])|^2  (reference_signals_impl.cc:224-228)
  const int16_t *tps;           // n_tps
  const float *pilot_ref;       // K  (+-4/3)
  const uint16_t *pay_c, *pay_L, *pay_R;   // [4][payload]
  const uint16_t *tps_L, *tps_R;           // [4][n_tps]
  // rank form for the fused kernel: gains are computed once per estimation carrier (rank r <-> carrier pil_k[r])
  const uint16_t *pil_k;                   // [4][DEMOD_NP]
  int np[4];                               // estimation carriers per pattern
  const uint16_t *pay_Li, *pay_Ri; const uint8_t *pay_d;   // [4][payload]
  const uint16_t *tps_Li, *tps_Ri; const uint8_t *tps_d;   // [4][n_tps]
  const uint32_t *pay_pack;                // [4][payload]: pay_c | pay_Li << 13 | pay_d << 23 (pay_Ri = pay_Li + 1 always)
};
constexpr int DEMOD_NP = 768;              // >= estimation carriers of a symbol (569 scattered + 177 continual - shared)
struct SymInfo { int freq_offset; int mod_index; float cfc; int pad; };

__device__ __forceinline__ float wave_sum(float v)
{ for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }

// parse_input (reference_signals_impl.cc:1189-1248) minus the sequential TPS/frame bookkeeping,
// which only needs mod_index and the per-carrier TPS values produced here.
// in: fft items [n_symbols][N]; symbol s needs item s+1 too (compute_oneshot_csft :747-790).
__global__ __launch_bounds__(256) void demod_kernel(const float2 *__restrict__ fft, FrontParams p, const RxState *st, int nitems_fixed,
                                                   DemodTables T, float2 *__restrict__ eq, float2 *__restrict__ tpsval,
                                                   SymInfo *__restrict__ info)
{
  __shared__ float s_sum[16];
  __shared__ float s_red[8][4];
  __shared__ int s_fo, s_mod;
  __shared__ float2 s_ph;
  const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nsym = st ? st->n_symbols : nitems_fixed;
  if (s + 1 >= nsym) return;
  const int N = p.N, zl = p.zl;
  const float2 *in = fft + (size_t)s * N;

  // integer CFO: process_cpilot_data :715-744 -- 16 candidate shifts x (n_cp-1) pilot pairs
  {
    int cand = tid >> 4, sub = tid & 15;
    int i = zl - 8 + cand;
    float sum = 0.f;
    for (int j = sub; j < p.n_cp - 1; j += 16) {
      float2 a = in[i + T.cpilot[j + 1]], b = in[i + T.cpilot[j]];
      float dx = a.x - b.x, dy = a.y - b.y;
      sum += T.known_diff[j] * (dx * dx + dy * dy);
    }
    for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (sub == 0) s_sum[cand] = sum;
  }
  __syncthreads();
  if (tid == 0) {
    float mx = 0.f; int start = 0;
    for (int c = 0; c < 16; c++) if (s_sum[c] > mx) { mx = s_sum[c]; start = zl - 8 + c; }
    s_fo = start - zl;
  }
  __syncthreads();
  const int fo = s_fo;

  // compute_oneshot_csft :747-790 -- left/right continual-pilot correlation with the next symbol
  {
    int half = (p.n_cp - 1) / 2;
    float lr = 0.f, li = 0.f, rr = 0.f, ri = 0.f;
    for (int j = tid; j < p.n_cp; j += 256) {
      if (j == half) continue;
      int idx = fo + zl + T.cpilot[j];
      float2 c = cmulc(in[idx], in[idx + N]);
      if (j < half) { lr += c.x; li += c.y; } else { rr += c.x; ri += c.y; }
    }
    lr = wave_sum(lr); li = wave_sum(li); rr = wave_sum(rr); ri = wave_sum(ri);
    if (lane == 0) { s_red[wave][0] = lr; s_red[wave][1] = li; s_red[wave][2] = rr; s_red[wave][3] = ri; }
  }
  __syncthreads();
  if (tid == 0) {
    float lr = 0, li = 0, rr = 0, ri = 0;
    for (int w = 0; w < 4; w++) { lr += s_red[w][0]; li += s_red[w][1]; rr += s_red[w][2]; ri += s_red[w][3]; }
    float la = atan2f(li, lr), ra = atan2f(ri, rr);
    float carrier_coeff = (float)(1.0 / (2 * M_PI * (1 + (float)p.cp / (float)N) * 2));
    float cfc = (ra + la) * carrier_coeff;
    // frequency_correction :793-819: one constant phasor for the whole symbol
    float correction = (float)fo + cfc;
    float ph = (float)(-2 * M_PI * correction * (N + p.cp) / N);
    float sn, cs; sincosf(ph, &sn, &cs);
    s_ph = make_float2(cs, sn);
    info[s].freq_offset = fo; info[s].cfc = cfc;
  }
  __syncthreads();
  const float2 cph = s_ph;
  const float2 *xin = in + zl + fo;               // derot_in[zl + k] = cph * in[zl + k + fo]

  // symbol index mod 4: process_spilot_data :549-582 -- first 10 scattered pilots of each pattern
  if (tid < 64) {
    int pat = tid >> 4, j = tid & 15;
    float cr = 0.f, ci = 0.f;
    if (j < 10) {
      int k = 3 * pat + 12 * j;
      float2 v = cmul(cph, xin[k]);
      float r = T.pilot_ref[k];
      cr = r * v.x; ci = -r * v.y;                // ref * conj(v)
    }
    for (int o = 8; o > 0; o >>= 1) { cr += __shfl_xor(cr, o); ci += __shfl_xor(ci, o); }
    if (j == 0) s_sum[pat] = cr * cr + ci * ci;
  }
  __syncthreads();
  if (tid == 0) {
    // `max` starts at 0 and d_mod_symbol_index keeps its previous value when nothing exceeds it;
    // with any signal present one pattern always does, so the previous value is never needed.
    float mx = 0.f; int mod = 0;
    for (int c = 0; c < 4; c++) if (s_sum[c] > mx) { mx = s_sum[c]; mod = c; }
    s_mod = mod; info[s].mod_index = mod;
  }
  __syncthreads();
  const int mod = s_mod;

  // channel gains (set_channel_gain :486-490, interpolation :617-642) + equalise (:1111-1114)
  auto gain_at = [&](int c, int L, int Rr) -> float2 {
    float2 gl = cdiv(make_float2(T.pilot_ref[L], 0.f), cmul(cph, xin[L]));
    if (L == c) return gl;
    float2 gr = cdiv(make_float2(T.pilot_ref[Rr], 0.f), cmul(cph, xin[Rr]));
    float j = (float)(c - L);
    float tx = (gr.x - gl.x) / 11.0f, ty = (gr.y - gl.y) / 11.0f;       // the constant 11 (:625)
    return make_float2(gl.x + tx * j, gl.y + ty * j);
  };
  const uint16_t *pc = T.pay_c + (size_t)mod * p.payload, *pL = T.pay_L + (size_t)mod * p.payload, *pR = T.pay_R + (size_t)mod * p.payload;
  float2 *o = eq + (size_t)s * p.payload;
  for (int i = tid; i < p.payload; i += 256) {
    int c = pc[i];
    float2 g = gain_at(c, pL[i], pR[i]);
    o[i] = cmul(cmul(cph, xin[c]), g);
  }
  // equalised TPS carriers (process_tps_data :929-931)
  if (tid < p.n_tps) {
    int c = T.tps[tid];
    float2 g = gain_at(c, T.tps_L[mod * p.n_tps + tid], T.tps_R[mod * p.n_tps + tid]);
    tpsval[(size_t)s * p.n_tps + tid] = cmul(cmul(cph, xin[c]), g);
  }
}

// DBPSK majority vote per symbol (process_tps_data :929-950): parallel over symbols
__global__ __launch_bounds__(256) void tps_vote_kernel(const float2 *__restrict__ tpsval, int n_tps, const RxState *st, int nitems_fixed,
                                                      const float2 *__restrict__ prev0, int *__restrict__ maj, int keep_last = 0)
{
  // four lanes per symbol, each a quarter of the TPS carriers
  const int s = blockIdx.x * 64 + (threadIdx.x >> 2), part = threadIdx.x & 3;
  const int nsym = st ? st->n_symbols : nitems_fixed;
  const bool act = s + (keep_last ? 0 : 1) < nsym;
  const int per = (n_tps + 3) / 4, k0 = part * per, k1 = k0 + per < n_tps ? k0 + per : n_tps;
  int m = 0;
  if (act)
    for (int k = k0; k < k1; k++) {
      const float2 v = tpsval[(size_t)s * n_tps + k];
      const float2 pv = s > 0 ? tpsval[(size_t)(s - 1) * n_tps + k] : (prev0 ? prev0[k] : make_float2(0.f, 0.f));
      const float re = v.x * pv.x + v.y * pv.y;
      m += (re >= 0.0f) ? 1 : -1;
    }
  m += __shfl_xor(m, 1); m += __shfl_xor(m, 2);
  if (act && part == 0) maj[s] = m;
}

// persistent TPS / frame-sync state (pilot_gen members + demod block members)
struct TpsState {
  unsigned long long fifo_lo;   // fifo[0..63], bit i = fifo[i]
  unsigned fifo_hi;             // fifo[64..67]
  int symbol_index, symbol_index_known, frame_index, prev_mod, d_init;
};

__device__ inline int bch_check(unsigned long long lo, unsigned hi)
{ // verify_bch_code :385-425
  auto bit = [&](int i) -> unsigned { return i < 64 ? (unsigned)((lo >> i) & 1ull) : ((hi >> (i - 64)) & 1u); };
  unsigned reg = 0;
  for (int i = 0; i < 113; i++) {
    unsigned d = i < 60 ? 0u : bit(1 + (i - 60));
    unsigned fb = 1u & (d ^ reg);
    reg >>= 1; reg |= fb << 13;
    reg ^= (fb << 12) ^ (fb << 11) ^ (fb << 9) ^ (fb << 8) ^ (fb << 7) ^ (fb << 5) ^ (fb << 4);
  }
  for (int i = 0; i < 14; i++) if (bit(54 + i) != (1u & (reg >> i))) return -1;
  return 0;
}

// symbol/frame bookkeeping: parse_input :1228-1241, process_tps_data :952-1028,
// demod_reference_signals_impl.cc:108-143 (one item per call regime).  Sequential over symbols by
// nature (68-bit FIFO + counters); the per-symbol inputs are staged through LDS in tiles by the
// whole workgroup so the walking lane never waits on HBM.
constexpr int TPS_TILE = 2048;
__device__ __forceinline__ void tps_fsm_body(FrontParams p, RxState *st, int nitems_fixed, const SymInfo *info, const int *maj,
                               TpsState *ts, int *sym_index, int *superframe_flag, const unsigned char *sync_flags, int clear_d_init)
{
  __shared__ signed char s_mod[TPS_TILE];
  __shared__ short s_maj[TPS_TILE];
  __shared__ unsigned char s_sync[TPS_TILE], s_si[TPS_TILE], s_flag[TPS_TILE];
  __shared__ TpsState s_t;
  __shared__ int s_first_out;
  const int tid = threadIdx.x;
  const int nsym = st ? st->n_symbols : nitems_fixed;
  const int ntot = p.keep_last ? nsym : (nsym > 0 ? nsym - 1 : 0);
  if (tid == 0) { s_t = *ts; s_first_out = -1; if (clear_d_init) s_t.d_init = 0; }   // clear_d_init: the period's first item bears the sync_start tag
  // sync words s1..s15 as fifo bits 1..15 (only 15 of the 16 are compared: B-11)
  unsigned mask_even = 0, mask_odd = 0;
  {
    const unsigned char se[15] = {0, 0, 1, 1, 0, 1, 0, 1, 1, 1, 1, 0, 1, 1, 1};
    for (int i = 0; i < 15; i++) { mask_even |= (unsigned)se[i] << (1 + i); mask_odd |= (unsigned)(1 - se[i]) << (1 + i); }
  }
  for (int base = 0; base < ntot; base += TPS_TILE) {
    const int n = ntot - base < TPS_TILE ? ntot - base : TPS_TILE;
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
      s_mod[i] = (signed char)info[base + i].mod_index; s_maj[i] = (short)maj[base + i];
      s_sync[i] = sync_flags ? sync_flags[base + i] : 0;
    }
    __syncthreads();
    if (tid == 0) {
      TpsState t = s_t;
      for (int i = 0; i < n; i++) {
        int mod = s_mod[i];
        int diff = (mod - t.prev_mod + 4) % 4;
        t.prev_mod = mod;
        t.symbol_index = (t.symbol_index + diff) % 68;
        int si = t.symbol_index, fi = t.frame_index;
        int use = (!t.symbol_index_known || t.symbol_index != 0);
        unsigned bitv = use ? (s_maj[i] >= 0 ? 0u : 1u) : 0u;
        for (int k = 0; k < diff; k++) {
          t.fifo_lo = (t.fifo_lo >> 1) | ((unsigned long long)(t.fifo_hi & 1u) << 63);
          t.fifo_hi = (t.fifo_hi >> 1) | (bitv << 3);
        }
        unsigned low16 = (unsigned)(t.fifo_lo & 0xFFFEull);
        if (low16 == mask_even || low16 == mask_odd) {
          if (bch_check(t.fifo_lo, t.fifo_hi) == 0) {
            t.frame_index = (int)(((t.fifo_lo >> 23) & 1ull) << 1 | ((t.fifo_lo >> 24) & 1ull));
            t.symbol_index_known = 1; t.symbol_index = 67;
            if (st) st->tps_bits = (t.fifo_lo & TPS_STATIC_MASK) | (1ull << 63);
          } else t.symbol_index_known = 0;
          t.fifo_lo = 0; t.fifo_hi = 0;
        }
        s_si[i] = (unsigned char)si;
        if (s_sync[i]) t.d_init = 0;                             // sync_start tag: hunt the superframe start again (:115-116)
        int sf = 0;
        if (!t.d_init && (si % 68) == p.si_start && (fi % 4) == p.fi_start && (!p.hunt_known || t.symbol_index_known)) { t.d_init = 1; sf = 1; if (s_first_out < 0) s_first_out = base + i; }
        s_flag[i] = t.d_init ? (sf ? 2 : 1) : 0;                 // 0 dropped, 1 produced, 2 produced + superframe_start
      }
      s_t = t;
    }
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
      sym_index[base + i] = s_si[i];
      if (superframe_flag) superframe_flag[base + i] = s_flag[i];
    }
  }
  __syncthreads();
  if (tid == 0) {
    *ts = s_t;
    if (st) {
      st->first_out = s_first_out;
      if (s_first_out < 0) { st->status |= 4; st->n_out_symbols = 0; }
      else st->n_out_symbols = ntot - s_first_out;
    }
  }
}
__global__ __launch_bounds__(256) void tps_fsm_kernel(FrontParams p, RxState *st, int nitems_fixed, const SymInfo *info, const int *maj,
                               TpsState *ts, int *sym_index, int *superframe_flag, const unsigned char *sync_flags, const int *need_seq, int clear_d_init = 0)
{
  if (need_seq && *need_seq == 0) return;
  tps_fsm_body(p, st, nitems_fixed, info, maj, ts, sym_index, superframe_flag, sync_flags, clear_d_init);
}


// ---- the same bookkeeping, segment-parallel (segment path only: no sync_start tags, fresh state).
// The state is fully re-derived at every frame end (symbol_index forced to 67, frame number read
// from the TPS bits, FIFO cleared: :978-1028,1240-1241), so a lane that starts TPS_WARM symbols
// early from a blank state holds the sequential state when it reaches its own segment, provided
// a frame end with an intact TPS word lies in the warm-up.  Every lane records its state at the
// start and at the end of its segment; tps_tail_kernel (tps_finalize_body) checks that neighbours agree and
// otherwise requests the sequential kernel (need_seq), so the result is always the sequential one.
constexpr int TPS_SEG = 32;           // symbols per lane
constexpr int TPS_THREADS = 256;      // lanes per workgroup (a workgroup covers TPS_THREADS * TPS_SEG symbols)
constexpr int TPS_WARM = 204;         // three frames
struct TpsEdge { TpsState start, end; };

// verify_bch_code (:385-425) as a linear map: the LFSR register after the 53 data bits (fifo bits 1..53; the 60
// leading zero clocks leave it at zero) is the XOR of one 14-bit response per set bit, looked up bytewise in a
// table built on the host (tps_bch_table_host) that the kernel copies into LDS; the word is valid when it equals fifo bits 54..67.
__device__ __forceinline__ int bch_check_tab(const unsigned short *T, unsigned long long lo, unsigned hi)
{
  const unsigned long long data = (lo >> 1) & ((1ull << 53) - 1);
  unsigned reg = 0;
#pragma unroll
  for (int b = 0; b < 7; b++) reg ^= T[b * 256 + (int)((data >> (8 * b)) & 255ull)];
  const unsigned parity = (unsigned)((lo >> 54) | ((unsigned long long)hi << 10)) & 0x3fffu;
  return reg == parity ? 0 : -1;
}
// built once on the host (the kernel used to build it in LDS at every launch: 56 x 53 dependent register steps, ~10 us of its 64)
inline std::vector<uint16_t> tps_bch_table_host()
{
  unsigned R[56];
  for (int i = 0; i < 56; i++) {
    unsigned reg = 0;
    for (int it = 0; it < 53; it++) {                             // response to a single 1 at data bit i
      const unsigned d = it == i ? 1u : 0u, fb = 1u & (d ^ reg);
      reg >>= 1; reg |= fb << 13;
      reg ^= (fb << 12) ^ (fb << 11) ^ (fb << 9) ^ (fb << 8) ^ (fb << 7) ^ (fb << 5) ^ (fb << 4);
    }
    R[i] = i < 53 ? reg : 0;
  }
  std::vector<uint16_t> T(7 * 256);
  for (int e = 0; e < 7 * 256; e++) {
    unsigned r = 0;
    for (int j = 0; j < 8; j++) if ((e >> j) & 1) r ^= R[(e >> 8) * 8 + j];
    T[e] = (uint16_t)r;
  }
  return T;
}

// One symbol of the bookkeeping on registers: the 68-bit FIFO as three 32-bit words (v_alignbit shifts), no branch outside the sync-word match.
// (The first version walked a TpsState with 64-bit variable shifts and fetched every symbol's two values from LDS inside the dependent chain:
// ~600 cycles per symbol, 236 symbols per lane = 61 us.)
struct TpsRegs { unsigned f0, f1, f2; int symbol_index, known, frame_index, prev_mod; };
__device__ __forceinline__ void tps_advance(TpsRegs &t, int mod, unsigned neg, int fi_start, int si_start, int hunt_known, unsigned mask_even, unsigned mask_odd,
                                            int &si_out, int &cand, const unsigned short *T, unsigned long long *tps_bits)
{
  const int diff = (mod - t.prev_mod) & 3;
  t.prev_mod = mod;
  int si = t.symbol_index + diff; if (si >= 68) si -= 68;
  t.symbol_index = si;
  const int fi = t.frame_index;
  const unsigned bitv = (!t.known || si != 0) ? neg : 0u;
  // `diff` shifts of the 68-bit FIFO, each inserting bitv at the top (process_tps_data :952-960), in closed form: bits 3 .. 4 - diff of the top word
  t.f0 = __builtin_amdgcn_alignbit(t.f1, t.f0, diff);
  t.f1 = __builtin_amdgcn_alignbit(t.f2, t.f1, diff);
  t.f2 = (t.f2 >> diff) | (((0xEC80u >> (4 * diff)) & 0xFu) & (0u - bitv));
  const unsigned x16 = (t.f0 ^ mask_even) & 0xFFFEu;             // mask_odd is mask_even's complement on bits 1..15
  (void)mask_odd;
  if (((x16 + 2u) & 0xFFFCu) == 0u) {                            // x16 (even) is 0 or 0xFFFE: the even or the odd frame's sync word
    const unsigned long long lo = (unsigned long long)t.f0 | ((unsigned long long)t.f1 << 32);
    if (bch_check_tab(T, lo, t.f2) == 0) {
      t.frame_index = (int)(((lo >> 23) & 1ull) << 1 | ((lo >> 24) & 1ull));
      t.known = 1; t.symbol_index = 67;
      if (tps_bits) *tps_bits = (lo & TPS_STATIC_MASK) | (1ull << 63);   // every valid frame of a stream stores the same word
    } else t.known = 0;
    t.f0 = 0; t.f1 = 0; t.f2 = 0;
  }
  si_out = si;
  cand = (si == si_start) && ((fi & 3) == fi_start) && (!hunt_known || t.known);
}
__device__ __forceinline__ TpsState tps_pack(const TpsRegs &r)
{
  TpsState t; t.fifo_lo = (unsigned long long)r.f0 | ((unsigned long long)r.f1 << 32); t.fifo_hi = r.f2; t.symbol_index = r.symbol_index;
  t.symbol_index_known = r.known; t.frame_index = r.frame_index; t.prev_mod = r.prev_mod; t.d_init = 0;
  return t;
}

__global__ __launch_bounds__(TPS_THREADS) void tps_fsm_par_kernel(FrontParams p, const RxState *st, const SymInfo *__restrict__ info, const int *__restrict__ maj,
                                                                 int *__restrict__ sym_index, TpsEdge *__restrict__ edges, int *first_cand, unsigned long long *tps_bits,
                                                                 const unsigned short *__restrict__ bch_tab, const TpsState *__restrict__ init = nullptr /* the members in front of
                                                                 symbol 0 when the chain does not start from blank ones (a piece of a stream whose counters are known) */)
{
  // the symbol stream in LDS, a byte per symbol: pattern index | (TPS vote negative) << 2; four bytes of padding per TPS_SEG entries keep the lanes
  // (TPS_SEG symbols apart) on distinct banks.  A lane's first symbol is a multiple of four symbols from `lo`, so it reads its stream a dword at a time.
  constexpr int NS = TPS_THREADS * TPS_SEG + TPS_WARM;
  __shared__ __attribute__((aligned(16))) unsigned char s_pk[NS + (NS / TPS_SEG) * 4 + 8];
  __shared__ unsigned short s_T[7 * 256];
  auto pm = [](int i) { return (int)((unsigned)i + ((unsigned)i / (unsigned)TPS_SEG) * 4u); };
  static_assert(TPS_WARM % 4 == 0 && TPS_SEG % 4 == 0, "a lane reads whole dwords");
  const int tid = threadIdx.x;
  const int nsym = st->n_symbols, ntot = p.keep_last ? nsym : (nsym > 0 ? nsym - 1 : 0);
  const int blk0 = blockIdx.x * TPS_THREADS * TPS_SEG;
  if (blk0 >= ntot) return;
  const int lo = blk0 - TPS_WARM < 0 ? 0 : blk0 - TPS_WARM;
  const int hi = blk0 + TPS_THREADS * TPS_SEG < ntot ? blk0 + TPS_THREADS * TPS_SEG : ntot;
  for (int i0 = lo + tid; i0 < hi; i0 += 8 * TPS_THREADS) {       // 8 symbols per lane in flight (clamped, unconditional loads)
    int mv[8], jv[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { const int i = i0 + k * TPS_THREADS, ic = i < hi ? i : hi - 1; mv[k] = info[ic].mod_index; jv[k] = maj[ic]; }
#pragma unroll
    for (int k = 0; k < 8; k++) { const int i = i0 + k * TPS_THREADS; if (i < hi) s_pk[pm(i - lo)] = (unsigned char)((mv[k] & 3) | (jv[k] < 0 ? 4 : 0)); }
  }
  for (int i = tid; i < 7 * 128; i += TPS_THREADS) reinterpret_cast<unsigned *>(s_T)[i] = reinterpret_cast<const unsigned *>(bch_tab)[i];
  __syncthreads();
  unsigned mask_even = 0, mask_odd = 0;
  {
    const unsigned char se[15] = {0, 0, 1, 1, 0, 1, 0, 1, 1, 1, 1, 0, 1, 1, 1};
    for (int i = 0; i < 15; i++) { mask_even |= (unsigned)se[i] << (1 + i); mask_odd |= (unsigned)(1 - se[i]) << (1 + i); }
  }
  const int s0 = blk0 + tid * TPS_SEG;
  if (s0 >= ntot) return;
  const int s1 = s0 + TPS_SEG < ntot ? s0 + TPS_SEG : ntot;
  int sw = s0 - TPS_WARM; if (sw < 0) sw = 0;
  TpsRegs t; t.f0 = 0; t.f1 = 0; t.f2 = 0; t.symbol_index = 0; t.known = 0; t.frame_index = 0; t.prev_mod = 0;
  if (init && sw == 0) {                                           // a lane whose warm-up begins with the stream's first symbol starts from what the chain holds there
    const TpsState i0 = *init;
    t.f0 = (unsigned)i0.fifo_lo; t.f1 = (unsigned)(i0.fifo_lo >> 32); t.f2 = i0.fifo_hi; t.symbol_index = i0.symbol_index; t.known = i0.symbol_index_known;
    t.frame_index = i0.frame_index; t.prev_mod = i0.prev_mod;
  }
  int si, cand;
  auto word = [&](int s) -> unsigned { return *reinterpret_cast<const unsigned *>(&s_pk[pm(s - lo)]); };   // symbols s .. s + 3 (s - lo: a multiple of 4)
  unsigned nxt = word(sw);
  for (int g = sw; g < s0; g += 4) {                               // warm-up: whole dwords (s0 - sw is a multiple of 4)
    const unsigned w = nxt;
    nxt = word(g + 4);                                             // the next dword is on its way while this one is walked (the array has 8 spare bytes)
#pragma unroll
    for (int k = 0; k < 4; k++) tps_advance(t, (int)((w >> (8 * k)) & 3u), (w >> (8 * k + 2)) & 1u, p.fi_start, p.si_start, p.hunt_known, mask_even, mask_odd, si, cand, s_T, nullptr);
  }
  const int seg = s0 / TPS_SEG;
  edges[seg].start = tps_pack(t);
  int first = 0x7fffffff;
  for (int g = s0; g < s1; g += 4) {
    const unsigned w = nxt;
    nxt = word(g + 4 < s1 ? g + 4 : g);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int s = g + k;
      if (s < s1) {
        tps_advance(t, (int)((w >> (8 * k)) & 3u), (w >> (8 * k + 2)) & 1u, p.fi_start, p.si_start, p.hunt_known, mask_even, mask_odd, si, cand, s_T, tps_bits);
        sym_index[s] = si;
        if (cand && first == 0x7fffffff) first = s;
      }
    }
  }
  edges[seg].end = tps_pack(t);
  if (first != 0x7fffffff) atomicMin(first_cand, first);
}

// (all 256 threads of the workgroup; returns whether the sequential bookkeeping has to run -- the same value in every thread)
__device__ __forceinline__ int tps_finalize_body(RxState *st, const TpsEdge *edges, const int *first_cand, int *need_seq, int keep_last, TpsState *ts)
{
  __shared__ int s_bad;
  const int tid = threadIdx.x;
  const int nsym = st->n_symbols, ntot = keep_last ? nsym : (nsym > 0 ? nsym - 1 : 0);
  const int nseg = (ntot + TPS_SEG - 1) / TPS_SEG;
  if (tid == 0) s_bad = 0;
  __syncthreads();
  for (int k = tid; k + 1 < nseg; k += 256) {
    const TpsState &a = edges[k].end, &b = edges[k + 1].start;
    if (a.fifo_lo != b.fifo_lo || a.fifo_hi != b.fifo_hi || a.symbol_index != b.symbol_index || a.symbol_index_known != b.symbol_index_known ||
        a.frame_index != b.frame_index || a.prev_mod != b.prev_mod) s_bad = 1;
  }
  __syncthreads();
  if (tid == 0) {
    *need_seq = s_bad;
    if (!s_bad) {
      int fo = *first_cand;
      if (fo == 0x7fffffff) { st->first_out = -1; st->status |= 4; st->n_out_symbols = 0; }
      else { st->first_out = fo; st->n_out_symbols = ntot - fo; }
      if (ts && nseg > 0) { TpsState e = edges[nseg - 1].end; e.d_init = fo != 0x7fffffff; *ts = e; }   // the members a later lock period starts from
    }
  }
  __syncthreads();
  return s_bad;
}

}  // namespace dvbt
