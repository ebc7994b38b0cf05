// k_drift.hpp -- the reference's FLOAT phase accumulator, reproduced in closed form.
//
// ofdm_sym_acquisition keeps the derotation phase in a float and adds the (double) increment once per sample, N + cp times per call
// (ofdm_sym_acquisition_impl.cc:285-309: `d_phase += d_phaseinc`, wrapped to [-pi, pi] with float constants).  A float sum rounds to the grid of
// the RESULT's binade, and the accumulator is always a multiple of that grid, so inside a binade every step advances the phase by the same amount
// q = rint(inc / ulp) * ulp instead of inc: the phase runs at a slightly wrong rate that changes from binade to binade (ulp = 2^-22 for |phase| in
// [2, pi], 2^-23 in [1, 2), ...).  Against the exact line that is a wander of up to 5e-4 rad inside one 8k symbol (1e-4 at 2k) whenever the carrier
// offset has a fractional part -- 2.4e-3 .. 3.7e-3 of the constellation spacing at the equalised-carrier tap of 8k QAM64, above the stated tolerance
// of 1e-3 (tests/test_gpu_channel.py measured it; tests/test_phase_accumulator_model.py replays the accumulator on the CPU).  Reproducing it sample by
// sample is a recurrence over the whole stream; this file does it in parallel:
//
//  * "time potential": with the per-step advance q_j of region j (a binade of one sign), T(phi) = integral of dphi / q over the unwrapped float phase
//    is piecewise linear, and n steps from phi0 end at T^-1(T(phi0) + n): one table of 15 regions per increment (|phase| < 2^-5 is treated as exact:
//    its grid is finer than 2^-28);
//  * the increment changes once per call ("run" r = the samples between the switch positions of calls r and r+1, increment -eps_r / N), so the phase at
//    the start of run s obeys  T_{s-1}(phi_s) = T_0(phi_0) + (n_s - n_0) - sum_{r=1}^{s-1} [T_{r-1}(phi_r) - T_r(phi_r)] : the sum's terms depend on phi_r
//    only through the small difference of two neighbouring tables, so a fixed-point iteration that evaluates them at the previous iterate (start: the exact
//    line) and takes a prefix sum converges in two or three rounds, all runs in parallel;
//  * from a call's entry phase every thread of a small kernel evaluates the deviation at one 32-sample block; the symbol kernels multiply the derotation
//    phasor of a sample by (1 + i delta) of its block.
//
// Validity: all increments of the lock period have one sign and |inc| >= 2 ulp(2) (|epsilon| >= 4e-3 rad at 8k), every switch position lies inside its
// call; otherwise (clean loopbacks, offsets that jitter around zero: the accumulator then lives near zero, where its grid is fine, and wanders < 1e-4)
// nothing is applied.  The first lock period of a segment is reproduced from the reference's start state (phase 0, increment 0); later periods start
// from phase 0 as well (the reference carries d_phase through the search calls in between; its value only sets where the binade crossings fall).
// The pure arithmetic is host-callable so that tests/test_drift_model.py can check it against the literal accumulator on the CPU.
#pragma once
#include "k_drift_math.hpp"

namespace dvbt {

#if defined(__HIPCC__)
// ------------------------------------------------------------------------------------------------ kernels (segment path)
// scratch layout (doubles): tabs[C][DRIFT_TAB] | ex_run[C] | ex_entry[C] | d[C] | S[C] | A0[1]
struct DriftBufs { double *tabs, *ex_run, *ex_entry, *d, *S, *A0; float *delta; int *flags; };   // flags[0]: sign / validity bits (collected by drift_prep_kernel, cleared by drift_exact_kernel), flags[1]: 1 = applied, flags[2]: 1 = negative increments

__device__ __forceinline__ void drift_load(const double *tabs, int r, double *q, double *tc)
{
  const double *t = tabs + (size_t)r * DRIFT_TAB;
#pragma unroll
  for (int j = 0; j < DRIFT_NR; j++) q[j] = t[j];
#pragma unroll
  for (int j = 0; j <= DRIFT_NR; j++) tc[j] = t[16 + j];
}

// tables of all runs + eligibility.  flags[0] collects: 1 some increment > 0, 2 some < 0, 4 something outside the model (tiny increment, switch outside its call)
__global__ __launch_bounds__(256) void drift_prep_kernel(FrontParams p, const RxState *st, const SymMeta *__restrict__ meta, DriftBufs B)
{
  const int r = blockIdx.x * 256 + threadIdx.x, nsym = st->n_symbols;
  if ((st->status & 1) || r >= nsym) return;
  const SymMeta m = meta[r];
  const double inc = m.incB;
  int f = inc > 0 ? 1 : (inc < 0 ? 2 : 4);
  if (fabs(inc) < DRIFT_MIN_INC) f |= 4;
  if (m.sw < 0 || m.sw >= p.N + p.cp) f |= 4;
  if (r == 0 && m.incA != 0.0) f |= 4;                         // a period that starts with a carried increment (block API): not modelled here
  if (f & 4) { atomicOr(&B.flags[0], f); return; }
  double q[DRIFT_NR], tc[DRIFT_NR + 1];
  drift_build(inc, q, tc);
  double *t = B.tabs + (size_t)r * DRIFT_TAB;
#pragma unroll
  for (int j = 0; j < DRIFT_NR; j++) t[j] = q[j];
#pragma unroll
  for (int j = 0; j <= DRIFT_NR; j++) t[16 + j] = tc[j];
  atomicOr(&B.flags[0], f);
}

// the float phase at the start of run r from the current prefix sums: T_{r-1}(phi_r) = A0 + (n_r - n_0) - S_{r-1}; first round: the exact line
__device__ __forceinline__ double drift_phi_run(const DriftBufs &B, const SymMeta *meta, int r, int L, bool neg, bool first)
{
  if (r == 0) return 0.0;
  if (first) return B.ex_run[r];
  double q[DRIFT_NR], tc[DRIFT_NR + 1];
  drift_load(B.tabs, r - 1, q, tc);
  const double steps = (double)r * L + meta[r].sw - meta[0].sw;
  return drift_Tinv(q, tc, neg, B.A0[0] + steps - B.S[r - 1]);
}

// One workgroup: the exact (unrounded) accumulated phase at every call entry and run start, a prefix sum in double over the calls (a thread owns a run of
// consecutive calls); decides whether the period is inside the model and says so to the kernels behind (flags[1], flags[2]).
__global__ __launch_bounds__(1024) void drift_exact_kernel(FrontParams p, const RxState *st, const SymMeta *__restrict__ meta, DriftBufs B)
{
  __shared__ double s_tot[1024];
  const int tid = threadIdx.x, nsym = st->n_symbols, L = p.N + p.cp;
  const int fl = B.flags[0];
  const bool on = !(st->status & 1) && nsym >= 2 && (fl == 1 || fl == 2);
  __syncthreads();
  if (tid == 0) { B.flags[1] = on ? 1 : 0; B.flags[2] = fl == 2 ? 1 : 0; B.flags[0] = 0; }     // flags[0] is left clear for the next lock period's drift_prep_kernel
  if (!on) return;
  const bool neg = fl == 2;
  const int per = (nsym + 1023) / 1024, sbeg = tid * per, cnt = sbeg >= nsym ? 0 : (nsym - sbeg < per ? nsym - sbeg : per);
  double local = 0.0;
  for (int i = 0; i < cnt; i++) { const SymMeta m = meta[sbeg + i]; local += m.sw * m.incA + (L - m.sw) * m.incB; }
  s_tot[tid] = local;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) { const double v = tid >= off ? s_tot[tid - off] : 0.0; __syncthreads(); s_tot[tid] += v; __syncthreads(); }
  double base = tid == 0 ? 0.0 : s_tot[tid - 1];
  for (int i = 0; i < cnt; i++) {
    const SymMeta m = meta[sbeg + i];
    B.ex_entry[sbeg + i] = base; B.ex_run[sbeg + i] = base + m.sw * m.incA;
    base += m.sw * m.incA + (L - m.sw) * m.incB;
  }
  if (tid == 0) { double q[DRIFT_NR], tc[DRIFT_NR + 1]; drift_load(B.tabs, 0, q, tc); B.A0[0] = drift_T(q, tc, neg, 0.0); }   // run 0 starts at phase 0 (increment 0 before it)
}

// One round of the fixed point, a thread per run, 256 runs per workgroup (the single-workgroup version of round 3's first build spent 0.4 ms on 4,624
// calls -- 1.5 ms on a 65-superframe segment -- loading three 31-double tables per run and round on ONE compute unit):
//   d_r = T_{r-1}(phi_r) - T_r(phi_r) at the previous round's phases (MODE 0: the exact line), S_r = d_1 + ... + d_r.
// The prefix S of the previous round's d is formed here: the workgroup sums everything in front of its 256 runs (at most nsym doubles from L2) and scans
// its own.  MODE 2: that prefix alone, written out for drift_table_kernel.  d_prev / d_next ping-pong between B.d and B.S.
template <int MODE> __global__ __launch_bounds__(256) void drift_round_kernel(FrontParams p, const RxState *st, const SymMeta *__restrict__ meta, DriftBufs B,
                                                                              const double *__restrict__ d_prev, double *__restrict__ d_next)
{
  __shared__ double s_red[256];
  if (!B.flags[1]) return;
  const int tid = threadIdx.x, nsym = st->n_symbols, L = p.N + p.cp, base = blockIdx.x * 256, r = base + tid;
  if (base >= nsym) return;
  const bool neg = B.flags[2] != 0;
  double s_before = 0.0;                                          // S_{r-1} of the previous round
  if (MODE != 0) {
    double acc = 0.0;
    for (int i = tid; i < base; i += 256) acc += d_prev[i];
    s_red[tid] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) s_red[tid] += s_red[tid + o]; __syncthreads(); }
    const double front = s_red[0];
    __syncthreads();
    const double v = r < nsym ? d_prev[r] : 0.0;
    s_red[tid] = v;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) { const double u = tid >= off ? s_red[tid - off] : 0.0; __syncthreads(); s_red[tid] += u; __syncthreads(); }
    const double incl = front + s_red[tid];
    s_before = incl - v;
    if (MODE == 2) { if (r < nsym) d_next[r] = incl; return; }
  }
  if (r >= nsym) return;
  double dr = 0.0;
  if (r >= 1) {
    double q[DRIFT_NR], tc[DRIFT_NR + 1];
    drift_load(B.tabs, r - 1, q, tc);
    const double phi = MODE == 0 ? B.ex_run[r] : drift_Tinv(q, tc, neg, B.A0[0] + ((double)r * L + meta[r].sw - meta[0].sw) - s_before);
    const double a = drift_T(q, tc, neg, phi);
    drift_load(B.tabs, r, q, tc);
    dr = a - drift_T(q, tc, neg, phi);
  }
  d_next[r] = dr;
}

// one workgroup per call, one thread per 32-sample block of the item: delta[s][k] = (float phase - exact line) after step 32 k + 17, relative to the call's entry
// (derot[j] carries the phase after j + 1 steps, :307; the block's middle sample stands for the block: the deviation moves < 4e-6 rad over 32 samples)
__global__ __launch_bounds__(256) void drift_table_kernel(FrontParams p, const RxState *st, const SymMeta *__restrict__ meta, DriftBufs B)
{
  __shared__ double s_ent, s_sw;
  const int s = blockIdx.x, k = threadIdx.x, nsym = st->n_symbols, L = p.N + p.cp, nb = p.N / 32;
  if (!B.flags[1] || s >= nsym) return;
  const bool neg = B.flags[2] != 0;
  const SymMeta m = meta[s];
  if (k == 0) {
    double ent = 0.0;
    if (s > 0) {   // the entry lies in run s - 1, (s L - n_{s-1}) steps after its start
      const double phr = drift_phi_run(B, meta, s - 1, L, neg, false);
      double q[DRIFT_NR], tc[DRIFT_NR + 1];
      drift_load(B.tabs, s - 1, q, tc);
      ent = drift_Tinv(q, tc, neg, drift_T(q, tc, neg, phr) + (double)(L - meta[s - 1].sw));
    }
    s_ent = ent;
    s_sw = drift_advance(m.incA, ent, (double)m.sw);
  }
  __syncthreads();
  for (int kk = k; kk < nb; kk += 256) {
    const int n = 32 * kk + 17;
    double q[DRIFT_NR], tc[DRIFT_NR + 1], phi, exact;
    if (n <= m.sw) {
      if (s == 0) phi = s_ent;                                   // increment 0 in front of the first switch
      else { drift_load(B.tabs, s - 1, q, tc); phi = drift_Tinv(q, tc, neg, drift_T(q, tc, neg, s_ent) + n); }
      exact = n * m.incA;
    } else {
      drift_load(B.tabs, s, q, tc);
      phi = drift_Tinv(q, tc, neg, drift_T(q, tc, neg, s_sw) + (n - m.sw));
      exact = m.sw * m.incA + (n - m.sw) * m.incB;
    }
    B.delta[(size_t)s * nb + kk] = (float)((phi - s_ent) - exact);
  }
}
// Block API (ofdm_sym_acquisition called item by item, acq_track_kernel walks the calls one after the other and carries the EMULATED float phase in
// AcqState / SymMeta.ph_base): the deviations of a call from its entry phase, any increments (sign changes included: every stretch has its own table)
__global__ __launch_bounds__(256) void drift_table_entry_kernel(FrontParams p, const RxState *st, const SymMeta *__restrict__ meta, float *__restrict__ delta, int *flags)
{
  const int s = blockIdx.x, nsym = st->n_symbols, nb = p.N / 32, L = p.N + p.cp;
  if (s == 0 && threadIdx.x == 0) flags[1] = 1;
  if (s >= nsym) return;
  const SymMeta m = meta[s];
  const double ent = (double)m.ph_base;
  const int sw = (m.sw >= 0 && m.sw < L) ? m.sw : L;
  __shared__ double s_sw;
  if (threadIdx.x == 0) s_sw = drift_advance_safe(m.incA, ent, (double)sw);
  __syncthreads();
  for (int kk = threadIdx.x; kk < nb; kk += 256) {
    const int n = 32 * kk + 17;
    double phi, exact;
    if (n <= sw) { phi = drift_advance_safe(m.incA, ent, (double)n); exact = n * m.incA; }
    else { phi = drift_advance_safe(m.incB, s_sw, (double)(n - sw)); exact = sw * m.incA + (n - sw) * m.incB; }
    delta[(size_t)s * nb + kk] = (float)((phi - ent) - exact);
  }
}
#endif  // __HIPCC__

}  // namespace dvbt
