// k_drift_math.hpp -- the host-callable arithmetic of k_drift.hpp (the closed form of the reference's float phase accumulator); see k_drift.hpp for the method.
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

// the same, usable for any increment: below DRIFT_MIN_INC (the accumulator's steps are granular there) the exact line is taken
DRIFT_HD double drift_advance_safe(double inc, double phi0, double n)
{
  if (fabs(inc) < DRIFT_MIN_INC) return phi0 + n * inc;
  return drift_advance(inc, phi0, n);
}
// the float accumulator's value of an unwrapped phase: wrapped into [-PI_F, PI_F) with the reference's float constants (:297-300)
DRIFT_HD double drift_wrap(double x) { return x - floor((x + DRIFT_PI_F) / DRIFT_2PI_F) * DRIFT_2PI_F; }


}  // namespace dvbt
