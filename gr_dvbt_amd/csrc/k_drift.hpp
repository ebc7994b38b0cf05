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
#include <cmath>
#include <cstdint>

#ifndef DRIFT_HD
#if defined(__HIPCC__)
#define DRIFT_HD __host__ __device__ __forceinline__
#else
#define DRIFT_HD inline
#endif
#endif

namespace dvbt {

constexpr int DRIFT_NR = 15;                       // regions: 7 binades per sign (2^-5 .. pi) and the zone around zero
constexpr int DRIFT_TAB = 32;                      // doubles per table in memory: q[15], pad, tc[16]
constexpr double DRIFT_PI_F = 3.1415927410125732;  // (float)M_PI: the wrap limits of the reference (:297-300)
constexpr double DRIFT_2PI_F = 6.2831854820251465; // (float)(2.0 * M_PI)
constexpr double DRIFT_MIN_INC = 2.0 * 2.384185791015625e-07;   // 2 ulp of [2, 4): below that the accumulator's steps are granular (rint = 0 or 1)

DRIFT_HD double drift_bnd(int j)                   // lower boundary of region j (j = DRIFT_NR: the upper end)
{
  if (j <= 0) return -DRIFT_PI_F;
  if (j >= DRIFT_NR) return DRIFT_PI_F;
  if (j <= 7) return -ldexp(1.0, 2 - j);          // -2, -1, ..., -2^-5
  return ldexp(1.0, j - 13);                      // 2^-5 (j = 8) ... 2 (j = 14)
}
DRIFT_HD double drift_ulp(int j)                   // float grid inside region j (0: the zone around zero, treated as exact)
{
  if (j == 7) return 0.0;
  const int e = j < 7 ? 1 - j : j - 13;           // binade exponent: |phase| in [2^e, 2^(e+1))
  return ldexp(1.0, e - 23);
}
DRIFT_HD int drift_region(double p)                // p in [-PI_F, PI_F]
{
  const double a = fabs(p);
  if (a < 0.03125) return 7;
  int e = ilogb(a); if (e > 1) e = 1;
  return p > 0 ? 13 + e : 1 - e;
}

// one table: q[j] = what a step adds in region j (mirrored coordinate: the increment is taken positive), tc[j] = steps from -PI_F to the region's lower boundary
struct DriftTab { double q[DRIFT_NR]; double tc[DRIFT_NR + 1]; };

DRIFT_HD void drift_build(double inc, double *q /* [DRIFT_NR] */, double *tc /* [DRIFT_NR + 1] */)
{
  const double a = fabs(inc);
  double t = 0.0;
  for (int j = 0; j < DRIFT_NR; j++) {
    const double u = drift_ulp(j);
    const double qq = u > 0.0 ? rint(a / u) * u : a;   // round-half-even like the float addition (the accumulator is an even multiple at a tie's second step)
    q[j] = qq; tc[j] = t;
    t += (drift_bnd(j + 1) - drift_bnd(j)) / qq;
  }
  tc[DRIFT_NR] = t;
}
// T(phi): steps needed from the (virtual) phase -PI_F of cycle 0 to the unwrapped float phase phi.  neg: the increment is negative (mirror)
DRIFT_HD double drift_T(const double *q, const double *tc, bool neg, double phi)
{
  const double x = neg ? -phi : phi;
  const double k = floor((x + DRIFT_PI_F) / DRIFT_2PI_F);
  const double p = x - k * DRIFT_2PI_F;
  const int j = drift_region(p);
  return k * tc[DRIFT_NR] + tc[j] + (p - drift_bnd(j)) / q[j];
}
DRIFT_HD double drift_Tinv(const double *q, const double *tc, bool neg, double t)
{
  const double k = floor(t / tc[DRIFT_NR]);
  const double tt = t - k * tc[DRIFT_NR];
  int j = 0;
  for (int i = 1; i < DRIFT_NR; i++) if (tc[i] <= tt) j = i;
  const double x = drift_bnd(j) + (tt - tc[j]) * q[j] + k * DRIFT_2PI_F;
  return neg ? -x : x;
}
// the unwrapped float phase n steps after phi0 (inc == 0: nothing moves)
DRIFT_HD double drift_advance(double inc, double phi0, double n)
{
  if (inc == 0.0 || n <= 0.0) return phi0;
  double q[DRIFT_NR], tc[DRIFT_NR + 1];
  drift_build(inc, q, tc);
  return drift_Tinv(q, tc, inc < 0, drift_T(q, tc, inc < 0, phi0) + n);
}

#if defined(__HIPCC__)
// ------------------------------------------------------------------------------------------------ kernels (segment path)
// scratch layout (doubles): tabs[C][DRIFT_TAB] | ex_run[C] | ex_entry[C] | d[C] | S[C] | A0[1]
struct DriftBufs { double *tabs, *ex_run, *ex_entry, *d, *S, *A0; float *delta; int *flags; };   // flags[0]: sign / validity bits (collected by drift_prep_kernel, cleared by drift_solve_kernel), flags[1]: 1 = applied, flags[2]: 1 = negative increments

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

// One workgroup: (1) the exact (unrounded) accumulated phase at every call entry and run start, a prefix sum in double over the calls; (2) three rounds of
// the fixed point: d_r = T_{r-1}(phi_r) - T_r(phi_r) at the previous round's phases (r >= 1), S_r = d_1 + ... + d_r.  A thread owns a run of consecutive calls.
__global__ __launch_bounds__(1024) void drift_solve_kernel(FrontParams p, const RxState *st, const SymMeta *__restrict__ meta, DriftBufs B)
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
  auto scan = [&](double local) -> double {   // exclusive prefix of the threads' sums
    s_tot[tid] = local;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) { const double v = tid >= off ? s_tot[tid - off] : 0.0; __syncthreads(); s_tot[tid] += v; __syncthreads(); }
    const double base = tid == 0 ? 0.0 : s_tot[tid - 1];
    __syncthreads();
    return base;
  };
  {
    double local = 0.0;
    for (int i = 0; i < cnt; i++) { const SymMeta m = meta[sbeg + i]; local += m.sw * m.incA + (L - m.sw) * m.incB; }
    double base = scan(local);
    for (int i = 0; i < cnt; i++) {
      const SymMeta m = meta[sbeg + i];
      B.ex_entry[sbeg + i] = base; B.ex_run[sbeg + i] = base + m.sw * m.incA;
      base += m.sw * m.incA + (L - m.sw) * m.incB;
    }
    if (tid == 0) { double q[DRIFT_NR], tc[DRIFT_NR + 1]; drift_load(B.tabs, 0, q, tc); B.A0[0] = drift_T(q, tc, neg, 0.0); }   // run 0 starts at phase 0 (increment 0 before it)
  }
  __threadfence_block(); __syncthreads();
  for (int it = 0; it < 3; it++) {
    double local = 0.0;
    for (int i = 0; i < cnt; i++) {
      const int r = sbeg + i;
      double dr = 0.0;
      if (r >= 1) {
        const double phi = drift_phi_run(B, meta, r, L, neg, it == 0);
        double q[DRIFT_NR], tc[DRIFT_NR + 1];
        drift_load(B.tabs, r - 1, q, tc);
        const double a = drift_T(q, tc, neg, phi);
        drift_load(B.tabs, r, q, tc);
        dr = a - drift_T(q, tc, neg, phi);
      }
      B.d[r] = dr; local += dr;
    }
    __syncthreads();                                              // every S of the previous round has been read
    double base = scan(local);
    for (int i = 0; i < cnt; i++) { base += B.d[sbeg + i]; B.S[sbeg + i] = base; }
    __threadfence_block(); __syncthreads();
  }
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
#endif  // __HIPCC__

}  // namespace dvbt
