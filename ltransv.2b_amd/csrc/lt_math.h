// lt_math.h -- device-side numerics of the LTRANS hot path for gfx950.
//
// Register-resident restatements (one wavefront lane = one particle) of
//   INT_MOD::polintd/linint            interpolation_module.f90:25-107
//   TENSION_MOD (TSPACK, TOMS 716)     tension_module.f90:231-1354
//   GRIDCELL_MOD::gridcell             gridcell_module.f90:26-257
//   HYDRO_MOD interpolation weights    hydrodynamic_module.f90:1680-1740,2527-2563
// The arithmetic order of every formula follows the Fortran so that results
// agree with the CPU oracle bit for bit (this file is compiled with -ffp-contract=off and the
// transcendental functions are the host-libm-exact ones of lt_libm.h).  What is different from the
// reference is the shape: fixed N=4 splines live in registers and are fitted
// one interval at a time by a single loop body (keeps the instruction
// footprint of the 36 fits per particle-step small), the interval search is a
// comparison chain, and nothing is allocated.
//
// Provenance and licences.  The TENSION_MOD section restates the ALGORITHM of TSPACK (R. J. Renka, ACM TOMS
// Algorithm 716: YPC1, SIGS, SNHCSH, HVAL, HPVAL, INTRVL) as the reference's tension_module.f90 carries it --
// operation by operation, because the reference's results branch on the last bit of these formulas, which is
// also why the identifiers of the published algorithm (SBIG, RTOL, D1D2, TP1, EMS, SSM, COSHMM, FMAX, ...) are
// kept.  TSPACK is copyrighted by the ACM and distributed under the ACM Software Copyright and License
// Agreement (royalty-free for academic, research and other noncommercial use; commercial use needs a licence
// from ACM, permissions@acm.org) -- see the reference's README.md ("TSPACK: tension spline curve-fitting
// package"); that agreement, not this repository's licence, governs this section.  MT19937 (lt_mt.h) is the
// Matsumoto-Nishimura generator (3-clause BSD).  lt_libm_impl.h / lt_libm_tables.h: see their headers.
#pragma once
#include <hip/hip_runtime.h>

namespace ltd {

#define LT_DEV __device__ __forceinline__
#define LT_HD __host__ __device__ __forceinline__   // (also callable from the host: the test build checks such code without a GPU)
#define LT_DEV_NOINLINE __device__ __noinline__

// RTOL = 200*eps computed by the STORE loop, tension_module.f90:433-439:
// the loop halves RTOL until 1+RTOL <= 1, i.e. RTOL = 2^-53, then *200.
#define LT_SIGS_RTOL (200.0 * 1.1102230246251565e-16)
#define LT_SBIG 85.0

// Fortran MIN/MAX as compare-and-select (NaN falls through to the second
// argument, like the oracle's dmin/dmax), not IEEE fmin/fmax.
LT_DEV double dmin(double a, double b) { return (a < b) ? a : b; }
LT_DEV double dmax(double a, double b) { return (a > b) ? a : b; }

// a / b for a divisor b that is the same for every particle, given 1/b as a double-double from the host:
// c.h + c.l = 1/b with |c.l| <= ulp(c.h), both of b's sign (the sign of a zero quotient then comes out right).
//   q = RN(a * c.h + RN(a * c.l))            -- a multiplication and an FMA (Brisebarre & Muller, "Correctly rounded
//                                               multiplication by arbitrary precision constants", IEEE TC 2008, algorithm 1)
// The value under the final rounding differs from a/b by less than 2^-104 |a/b| = 2^-51 ulp.  The host offers the
// constants only for a divisor b = B * 2^e with an odd integer B < 2^30 (4*ws, 6, differences of the external
// times): a quotient A/B of a 53-bit integer A is never closer than ulp/(2B) > 2^-31 ulp to the midpoint of two
// doubles (A * 2^j - B * (2m+1) is an odd integer), so the rounding is that of the exact quotient: q == a / b for
// every a of ordinary magnitude (|a/b| > 2^-960), bit for bit.  2 full-rate operations in a chain of 2; the
// reciprocal-and-two-corrections form this replaces (Markstein) was 5 in a chain of 5, the compiler's a / b ~14.
// tests/test_host.py checks the host's constants and the algorithm in exact rational arithmetic on numerators built
// to land next to midpoints, tests/test_gpu_div.py the device code against `/`.
struct DivConst { double h, l; };
LT_DEV double div_by(double a, double b, const DivConst& c, bool ok) {
  if (!ok) return a / b;
  return __builtin_fma(a, c.h, a * c.l);
}

// Division by a divisor that is used several times.  The compiler expands a / b into
//   s,t = div_scale(b), div_scale(a); y = rcp(s); two Newton steps on y (4 fma);
//   q = t*y; r = fma(-s,q,t); div_fmas(r,y,q); div_fixup
// where the scale / fixup steps only act when an exponent is near the ends of the range
// (|b| outside ~[2^-1021, 2^1021], |a| < 2^-969 and nonzero, quotient near overflow /
// underflow) or an operand is 0 / inf / NaN.  For operands of ordinary magnitude the
// quotient is therefore exactly  fma(fma(-b, a*y, a), y, a*y)  with the refined
// reciprocal y, which depends on b alone: prep_div() does the reciprocal part once and
// div_prep() is the three remaining operations.  Used only where the operands are depths,
// level spacings (metres, > 0) and differences of fp32 field values (or exactly 0):
// bit-identical to `a / b` there (checked on the device by tests/test_gpu_div.py).
struct Divisor { double b, y; };
LT_DEV Divisor prep_div(double b) {
  Divisor d;
  d.b = b;
  double y = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-b, y, 1.0);
  d.y = __builtin_fma(y, e, y);
  return d;
}
LT_DEV double div_prep(double a, const Divisor& d) {
  const double q = a * d.y;
  const double r = __builtin_fma(-d.b, q, a);
  // The corrected quotient has the sign of q (the correction is a fraction of an ulp of q) unless the
  // numerator is a zero: then q already is the zero of the right sign and the correction would lose
  // it.  Taking the sign of q covers both (one v_bfi instead of a compare and two selects).
  return __builtin_copysign(__builtin_fma(r, d.y, q), q);
}
// the divisor -b: every step of prep_div is odd in b, so this is prep_div(-b) bit for bit
LT_DEV Divisor neg_div(const Divisor& d) {
  Divisor n;
  n.b = -d.b; n.y = -d.y;
  return n;
}

// exp, log, log10, pow(x, 0.5), sin, cos, sincos evaluated exactly as the host libm of the CPU run
// does (the tension-spline code branches on the last bit of exp, the random-walk displacements
// carry log / cos / pow): lt_libm.h, shared with the host-side check of tests/.
#include "lt_libm.h"

// SIGS' convexity Newton iteration (tension_module.f90:531-580) gives up after
// 10000 evaluations (SigErr).  Each iterate is a function of the previous one
// only, so once an iterate REPEATS the sequence is periodic, none of the states
// on the cycle passed a stopping test, and the reference is certain to run into
// its cap: the outcome (SigErr) is known at the first repeat.  Brent's cycle
// detection finds it within ~2x(tail + period) iterations instead of 10000
// (the usual case is a 2-cycle in the last ulp).  A NaN iterate never passes a
// stopping test either.
struct NewtonCycle {
  double mark;
  int power, lam;
  __device__ __forceinline__ explicit NewtonCycle(double s0) : mark(s0), power(1), lam(0) {}
  __device__ __forceinline__ bool repeats(double s) {
    if (s == mark || s != s) return true;
    if (++lam == power) { mark = s; power <<= 1; lam = 0; }
    return false;
  }
};

// ---------------------------------------------------------------- INT_MOD
// polintd, interpolation_module.f90:70-107 (n = 3)
LT_DEV double polintd3(double xa1, double xa2, double xa3, double ya1, double ya2, double ya3,
                       double x) {
  int ns = 1;
  double dif = fabs(x - xa1);
  double dift = fabs(x - xa2);
  if (dift < dif) { ns = 2; dif = dift; }
  dift = fabs(x - xa3);
  if (dift < dif) { ns = 3; dif = dift; }
  double c = (xa2 - x) * ((ya3 - ya2) / (xa2 - xa3));
  c = c - (xa2 - x) * ((ya2 - ya1) / (xa1 - xa2));
  c = c / (xa1 - xa3);
  double a, b;
  if (ns == 3) {
    a = (ya3 - ya2) / (xa2 - xa3);
    b = xa3 - x;
  } else {
    a = (ya2 - ya1) / (xa1 - xa2);
    b = xa1 - x;
  }
  double yns = (ns == 1) ? ya1 : (ns == 2) ? ya2 : ya3;
  double xns = (ns == 1) ? xa1 : (ns == 2) ? xa2 : xa3;
  return yns + (xns - x) * a + b * c;
}

// linint for n = 4, interpolation_module.f90:25-59.  The bisection visits
// k=2 then k=3 or stops: jlo/jhi end as (1,2), (2,3) or (3,4).
LT_DEV double linint4(const double (&xa)[4], const double (&ya)[4], double x) {
  // k = (4+1)/2 = 2
  int jlo = 1, jhi = 4;
  if (xa[1] > x) jhi = 2; else jlo = 2;
  if (jhi - jlo != 1) {  // (2,4): k = 3
    if (xa[2] > x) jhi = 3; else jlo = 3;
  }
  double xl = (jlo == 1) ? xa[0] : (jlo == 2) ? xa[1] : xa[2];
  double yl = (jlo == 1) ? ya[0] : (jlo == 2) ? ya[1] : ya[2];
  double xh = (jhi == 2) ? xa[1] : (jhi == 3) ? xa[2] : xa[3];
  double yh = (jhi == 2) ? ya[1] : (jhi == 3) ? ya[2] : ya[3];
  double m = (yl - yh) / (xl - xh);
  double b = yl - m * xl;
  return m * x + b;
}

// linint over an array in memory (VTurb fallback), returns y and slope m
LT_DEV void linint_n(const double* xa, const double* ya, int n, int stride, double x, double& y,
                     double& m) {
  int jlo = 1, jhi = n;
  for (;;) {
    int k = (jhi + jlo) / 2;
    if (xa[(size_t)(k - 1) * stride] > x) jhi = k; else jlo = k;
    if (jhi - jlo == 1) break;
  }
  double xl = xa[(size_t)(jlo - 1) * stride], xh = xa[(size_t)(jhi - 1) * stride];
  double yl = ya[(size_t)(jlo - 1) * stride], yh = ya[(size_t)(jhi - 1) * stride];
  m = (yl - yh) / (xl - xh);
  double b = yl - m * xl;
  y = m * x + b;
}

// ------------------------------------------------------------ TENSION_MOD
// SNHCSH, tension_module.f90:784-849
LT_DEV void snhcsh(double X, double& SINHM, double& COSHM, double& COSHMM) {
  const double P1 = -3.51754964808151394800e5, P2 = -1.15614435765005216044e4,
               P3 = -1.63725857525983828727e2, P4 = -7.89474443963537015605e-1;
  const double Q1 = -2.11052978884890840399e6, Q2 = 3.61578279834431989373e4,
               Q3 = -2.77711081420602794433e2, Q4 = 1.0;
  double AX = fabs(X), XS = AX * AX;
  if (AX <= .5) {
    double XC = X * XS;
    double P = ((P4 * XS + P3) * XS + P2) * XS + P1;
    double Q = ((Q4 * XS + Q3) * XS + Q2) * XS + Q1;
    SINHM = XC * (P / Q);
    double XSD4 = .25 * XS;
    double XSD2 = XSD4 + XSD4;
    P = ((P4 * XSD4 + P3) * XSD4 + P2) * XSD4 + P1;
    Q = ((Q4 * XSD4 + Q3) * XSD4 + Q2) * XSD4 + Q1;
    double F = XSD4 * (P / Q);
    COSHMM = XSD2 * F * (F + 2.0);
    COSHM = COSHMM + XSD2;
  } else {
    double EXPX = exp_core(AX);
    SINHM = -(((1.0 / EXPX + AX) + AX) - EXPX) / 2.0;
    if (X < 0.0) SINHM = -SINHM;
    COSHM = ((1.0 / EXPX - 2.0) + EXPX) / 2.0;
    COSHMM = COSHM - XS / 2.0;
  }
}

LT_DEV double fsign1(double b) {  // SIGN(1.D0, b)
  return copysign(1.0, b);
}

// One interval of SIGS (tension_module.f90:314-781 with SIGMA initialised to
// zero as TSPSI's caller does): returns the tension factor of the interval
// with knot spacing DX, end values Y1,Y2 and end slopes S1,S2.  sigerr is
// incremented when the convexity Newton iteration exceeds 10000 steps (the
// caller must then stop, as the reference RETURNs from SIGS).
// The CONT flag is .TRUE. on the SIG<=.5 branch of the secant loop (original
// TSPACK semantics; the F90 text leaves it unset there) -- same choice as the
// oracle, checked against the flang build of the reference.
LT_DEV double sigs_interval(double DX, double Y1, double Y2, double S1, double S2, int& sigerr) {
  const double SBIG = LT_SBIG, RTOL = LT_SIGS_RTOL, FTOL = 0.0;
  double S = (Y2 - Y1) / DX;
  double D1 = S - S1;
  double D2 = S2 - S;
  double D1D2 = D1 * D2;
  // infinite tension :470-485
  if ((D1D2 == 0.0 && S1 != S2) || (S == 0.0 && S1 * S2 > 0.0)) return SBIG;
  double SIG = 0.0;
  if (D1D2 >= 0.0) {
    if (D1D2 == 0.0) return 0.0;
    double T = dmax(D1 / D2, D2 / D1);
    if (T <= 2.0) return 0.0;
    double TP1 = T + 1.0;
    // convexity: Newton on F(SIG) = SIG*COSHM/SINHM - TP1  :531-580
    SIG = sqrt(10.0 * T - 20.0);
    int NIT = 0;
    NewtonCycle cyc(SIG);
    for (;;) {
      double T1, FP;
      if (SIG <= .5) {
        double SINHM, COSHM, COSHMM;
        snhcsh(SIG, SINHM, COSHM, COSHMM);
        T1 = COSHM / SINHM;
        FP = T1 + SIG * (SIG / SINHM - T1 * T1 + 1.0);
      } else {
        double EMS = lt_exp(-SIG);
        double SSM = 1.0 - EMS * (EMS + SIG + SIG);
        T1 = (1.0 - EMS) * (1.0 - EMS) / SSM;
        FP = T1 + SIG * (2.0 * SIG * EMS / SSM - T1 * T1 + 1.0);
      }
      double F = SIG * T1 - TP1;
      NIT = NIT + 1;
      if (NIT > 10000) {
        sigerr = sigerr + 1;
        return 0.0;
      }
      if (FP <= 0.0) break;
      double DSIG = -F / FP;
      if (fabs(DSIG) <= RTOL * SIG || (F >= 0.0 && F <= FTOL) || fabs(F) <= RTOL) break;
      SIG = SIG + DSIG;
      if (cyc.repeats(SIG)) {
        sigerr = sigerr + 1;
        return 0.0;
      }
    }
    SIG = dmin(SIG, SBIG);
    return (SIG > 0.0) ? SIG : 0.0;
  }
  // convexity cannot be satisfied :597-624
  if (S1 * S < 0.0 || S2 * S < 0.0) return 0.0;
  double T0 = 3.0 * S - S1 - S2;
  double D0 = T0 * T0 - S1 * S2;
  if (D0 <= 0.0 || S * T0 >= 0.0) return 0.0;
  // monotonicity: secant iteration :638-760
  double SGN = fsign1(S);
  SIG = SBIG;
  double FMAX = SGN * (SIG * S - S1 - S2) / (SIG - 2.0);
  if (FMAX <= 0.0) return SBIG;
  double STOL = RTOL * SIG;
  double F = FMAX;
  double F0 = SGN * D0 / (3.0 * (D1 - D2));
  double FNEG = F0;
  double DSIG = SIG;
  double DMAX = SIG;
  double D1PD2 = D1 + D2;
  double A = 0.0, E = 0.0, C1, C2;
  for (;;) {
    DSIG = -F * DSIG / (F - F0);
    if (fabs(DSIG) > fabs(DMAX) || DSIG * DMAX > 0.) {
      DSIG = DMAX;
      F0 = FNEG;
      continue;
    }
    if (fabs(DSIG) < STOL / 2.0) DSIG = -copysign(STOL / 2.0, DMAX);
    SIG = SIG + DSIG;
    F0 = F;
    bool CONT = true;
    if (SIG <= .5) {
      double SINHM, COSHM, COSHMM;
      snhcsh(SIG, SINHM, COSHM, COSHMM);
      C1 = SIG * COSHM * D2 - SINHM * D1PD2;
      C2 = SIG * (SINHM + SIG) * D2 - COSHM * D1PD2;
      A = C2 - C1;
      E = SIG * SINHM - COSHMM - COSHMM;
    } else {
      double EMS = exp_core(-SIG);
      double EMS2 = EMS + EMS;
      double TM = 1.0 - EMS;
      double SSINH = TM * (1.0 + EMS);
      double SSM = SSINH - SIG * EMS2;
      double SCM = TM * TM;
      C1 = SIG * SCM * D2 - SSM * D1PD2;
      C2 = SIG * SSINH * D2 - SCM * D1PD2;
      F = FMAX;
      if (C1 * (SIG * SCM * D1 - SSM * D1PD2) >= 0.0) CONT = false;
      if (CONT) A = EMS2 * (SIG * TM * D2 + (TM - SIG) * D1PD2);
      if (A * (C2 + C1) < 0.0) CONT = false;
      if (CONT) E = SIG * SSINH - SCM - SCM;
    }
    if (CONT) F = (SGN * (E * S2 - C2) + sqrt(A * (C2 + C1))) / E;
    STOL = RTOL * SIG;
    if (fabs(DMAX) <= STOL || (F >= 0.0 && F <= FTOL) || fabs(F) <= RTOL) break;
    DMAX = DMAX + DSIG;
    if (F0 * F > 0.0 && fabs(F) >= fabs(F0)) {
      DSIG = DMAX;
      F0 = FNEG;
      continue;
    }
    if (F0 * F <= 0.0) {
      double T1 = DMAX, T2 = FNEG;
      DMAX = DSIG;
      FNEG = F0;
      if (fabs(DSIG) > fabs(T1) && fabs(F) < fabs(T2)) {
        DSIG = T1;
        F0 = T2;
      }
    }
  }
  SIG = dmin(SIG, SBIG);
  return (SIG > 0.0) ? SIG : 0.0;
}

// YPC1 end / interior formulas, tension_module.f90:852-977
LT_DEV double ypc1_end(double SI, double T) {
  return (SI >= 0.0) ? dmin(dmax(0.0, T), 3.0 * SI) : dmax(dmin(0.0, T), 3.0 * SI);
}
LT_DEV double ypc1_mid(double DXIM1, double DXI, double SIM1, double SI) {
  double T = (DXIM1 * SI + DXI * SIM1) / (DXIM1 + DXI);
  double ASIM1 = fabs(SIM1), ASI = fabs(SI);
  double SGN = fsign1(SI);
  if (ASIM1 > ASI) SGN = fsign1(SIM1);
  return (SGN > 0.0) ? dmin(dmax(0.0, T), 3.0 * dmin(ASIM1, ASI))
                     : dmax(dmin(0.0, T), -3.0 * dmin(ASIM1, ASI));
}

LT_DEV double ypc1_mid_d(double DXIM1, double DXI, const Divisor& dsum, double SIM1, double SI) {
  double T = div_prep(DXIM1 * SI + DXI * SIM1, dsum);
  double ASIM1 = fabs(SIM1), ASI = fabs(SI);
  double SGN = fsign1(SI);
  if (ASIM1 > ASI) SGN = fsign1(SIM1);
  return (SGN > 0.0) ? dmin(dmax(0.0, T), 3.0 * dmin(ASIM1, ASI))
                     : dmax(dmin(0.0, T), -3.0 * dmin(ASIM1, ASI));
}

// The same two formulas with the hardware's min / max (one instruction instead of a compare
// and two selects each), for the fast path of k_rk: identical values for finite operands;
// only the SIGN of a zero result can differ from the compare-and-select forms above, and no
// later operation of the path looks at the sign of a zero derivative (they are subtracted
// from slopes, multiplied and compared with 0).  NaN operands do not reach it (degenerate
// level windows are taken out before).
LT_DEV double ypc1_end_f(double SI, double T) {
  const double lim = 3.0 * SI;
  return (SI >= 0.0) ? __builtin_fmin(__builtin_fmax(0.0, T), lim) : __builtin_fmax(__builtin_fmin(0.0, T), lim);
}
LT_DEV double ypc1_mid_f(double DXIM1, double DXI, const Divisor& dsum, double SIM1, double SI) {
  const double T = div_prep(DXIM1 * SI + DXI * SIM1, dsum);
  const double ASIM1 = fabs(SIM1), ASI = fabs(SI);
  double SGN = fsign1(SI);
  if (ASIM1 > ASI) SGN = fsign1(SIM1);
  const double lim = 3.0 * __builtin_fmin(ASIM1, ASI);
  return (SGN > 0.0) ? __builtin_fmin(__builtin_fmax(0.0, T), lim) : __builtin_fmax(__builtin_fmin(0.0, T), -lim);
}

// The same formulas without a branch on the sign: with sb the sign bit of the slope that decides,
//   SGN > 0:  min(max(0, T), lim)            SGN < 0:  max(min(0, T), -lim) = -min(max(0, -T), lim)
// i.e. flip T by sb, clamp to [0, lim], flip back (a sign flip is exact).  The hardware min / max
// are issued directly: the operands are results of arithmetic (never signalling NaNs), so the
// canonicalising v_max x,x the compiler puts in front of llvm.minnum / maxnum is dead weight.
// Same values as the forms above except, again, for the sign of a zero result.
LT_DEV double vmax0(double x) {
  double r;
  asm("v_max_f64 %0, %1, 0" : "=v"(r) : "v"(x));
  return r;
}
LT_DEV double vmin2(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
LT_DEV double vmax2(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
LT_DEV double flipsign(double x, unsigned sb) {  // sb = 0 or 0x80000000
  return __hiloint2double(__double2hiint(x) ^ (int)sb, __double2loint(x));
}
LT_DEV double ypc1_end_b(double SI, double T) {
  const unsigned sb = (unsigned)__double2hiint(SI) & 0x80000000u;
  const double lim = 3.0 * fabs(SI);
  return flipsign(vmin2(vmax0(flipsign(T, sb)), lim), sb);
}
LT_DEV double ypc1_mid_b(double DXIM1, double DXI, const Divisor& dsum, double SIM1, double SI) {
  const double T = div_prep(DXIM1 * SI + DXI * SIM1, dsum);
  const double ASIM1 = fabs(SIM1), ASI = fabs(SI);
  const unsigned sb = (unsigned)((ASIM1 > ASI) ? __double2hiint(SIM1) : __double2hiint(SI)) & 0x80000000u;
  const double lim = 3.0 * vmin2(ASIM1, ASI);
  return flipsign(vmin2(vmax0(flipsign(T, sb)), lim), sb);
}

// YPC1 for N = 4.  ier != 0 when the abscissae are not strictly increasing.
LT_DEV void ypc1_4(const double (&X)[4], const double (&Y)[4], double (&YP)[4], int& ier) {
  double DX1 = X[1] - X[0], DX2 = X[2] - X[1], DX3 = X[3] - X[2];
  ier = 0;
  if (DX1 <= 0.0) { ier = 2; return; }
  if (DX2 <= 0.0) { ier = 3; return; }
  if (DX3 <= 0.0) { ier = 4; return; }
  double SL1 = (Y[1] - Y[0]) / DX1;
  double SL2 = (Y[2] - Y[1]) / DX2;
  double SL3 = (Y[3] - Y[2]) / DX3;
  YP[0] = ypc1_end(SL1, SL1 + DX1 * (SL1 - SL2) / (DX1 + DX2));
  YP[1] = ypc1_mid(DX1, DX2, SL1, SL2);
  YP[2] = ypc1_mid(DX2, DX3, SL2, SL3);
  YP[3] = ypc1_end(SL3, SL3 + DX3 * (SL3 - SL2) / (DX2 + DX3));
}

// value of the Hermite tension spline on one interval, HVAL
// tension_module.f90:1052-1117 (after the interval search)
LT_DEV double hval_interval(double T, double X1, double X2, double Y1, double Y2, double YP1,
                            double YP2, double SIGMA) {
  double DX = X2 - X1;
  double U = T - X1;
  double B2 = U / DX;
  double B1 = 1.0 - B2;
  double S1 = YP1;
  double S = (Y2 - Y1) / DX;
  double D1 = S - S1;
  double D2 = YP2 - S;
  double SIG = fabs(SIGMA);
  if (SIG < 1.e-9) {
    return Y1 + U * (S1 + B2 * (D1 + B1 * (D1 - D2)));
  } else if (SIG <= .5) {
    double SB2 = SIG * B2, SM, CM, CMM, SM2, CM2, DUMMY;
    snhcsh(SIG, SM, CM, CMM);
    snhcsh(SB2, SM2, CM2, DUMMY);
    double E = SIG * SM - CMM - CMM;
    return Y1 + S1 * U +
           DX * ((CM * SM2 - SM * CM2) * (D1 + D2) + SIG * (CM * CM2 - (SM + SIG) * SM2) * D1) /
               (SIG * E);
  } else {
    double SB1 = SIG * B1;
    double SB2 = SIG - SB1;
    if (-SB1 > LT_SBIG || -SB2 > LT_SBIG) return Y1 + S * U;
    double E1 = exp_core(-SB1), E2 = exp_core(-SB2);
    double EMS = E1 * E2;
    double TM = 1.0 - EMS;
    double TS = TM * TM;
    double TP = 1.0 + EMS;
    double E = TM * (SIG * TP - TM - TM);
    return Y1 + S * U +
           DX * (TM * (TP - E1 - E2) * (D1 + D2) +
                 SIG * ((E2 + EMS * (E1 - 2.0) - B1 * TS) * D1 +
                        (E1 + EMS * (E2 - 2.0) - B2 * TS) * D2)) /
               (SIG * E);
  }
}

// derivative on one interval, HPVAL tension_module.f90:1196-1250
LT_DEV double hpval_interval(double T, double X1, double X2, double Y1, double Y2, double YP1,
                             double YP2, double SIGMA) {
  double DX = X2 - X1;
  double B1 = (X2 - T) / DX;
  double B2 = 1.0 - B1;
  double S1 = YP1;
  double S = (Y2 - Y1) / DX;
  double D1 = S - S1;
  double D2 = YP2 - S;
  double SIG = fabs(SIGMA);
  if (SIG < 1.e-9) {
    return S1 + B2 * (D1 + D2 - 3.0 * B1 * (D2 - D1));
  } else if (SIG <= .5) {
    double SB2 = SIG * B2, SM, CM, CMM, SM2, CM2, DUMMY;
    snhcsh(SIG, SM, CM, CMM);
    snhcsh(SB2, SM2, CM2, DUMMY);
    double SINH2 = SM2 + SB2;
    double E = SIG * SM - CMM - CMM;
    return S1 + ((CM * CM2 - SM * SINH2) * (D1 + D2) + SIG * (CM * SINH2 - (SM + SIG) * CM2) * D1) / E;
  } else {
    double SB1 = SIG * B1;
    double SB2 = SIG - SB1;
    if (-SB1 > LT_SBIG || -SB2 > LT_SBIG) return S;
    double E1 = exp_core(-SB1), E2 = exp_core(-SB2);
    double EMS = E1 * E2;
    double TM = 1.0 - EMS;
    double E = TM * (SIG * (1.0 + EMS) - TM - TM);
    return S + (TM * ((E2 - E1) * (D1 + D2) + TM * (D1 - D2)) +
                SIG * ((E1 * EMS - E2) * D1 + (E1 - E2 * EMS) * D2)) /
                   E;
  }
}

// TSPSI(4) + HVAL at T, or the linint fallback when SIGS reports SigErr:
// the body of one slot of WCTS_ITPI, hydrodynamic_module.f90:2619-2644.
// sigerr_count (diagnostic) counts fits that fell back.
LT_DEV double spline4_eval(const double (&X)[4], const double (&Y)[4], double T,
                            int& sigerr_count) {
  double YP[4];
  int ier;
  ypc1_4(X, Y, YP, ier);
  // TSPSI: IER = -4 when YPC1 fails; SIGMA is then whatever the caller's array
  // held.  s-levels are strictly increasing for any valid ROMS grid, so this
  // only guards against h <= 0: treat as zero tension with zero slopes.
  double sig0 = 0.0, sig1 = 0.0, sig2 = 0.0;
  int sigerr = 0;
  if (ier == 0) {
    // SIGS interval loop (tension_module.f90:448-776): ONE copy of the
    // interval body, fed through selects; stops at the first SigErr as the
    // reference RETURNs.
#pragma unroll 1
    for (int iv = 0; iv < 3; iv++) {
      double dx = (iv == 0) ? X[1] - X[0] : (iv == 1) ? X[2] - X[1] : X[3] - X[2];
      double ya = (iv == 0) ? Y[0] : (iv == 1) ? Y[1] : Y[2];
      double yb = (iv == 0) ? Y[1] : (iv == 1) ? Y[2] : Y[3];
      double pa = (iv == 0) ? YP[0] : (iv == 1) ? YP[1] : YP[2];
      double pb = (iv == 0) ? YP[1] : (iv == 1) ? YP[2] : YP[3];
      double sg = sigs_interval(dx, ya, yb, pa, pb, sigerr);
      if (sigerr != 0) break;
      if (iv == 0) sig0 = sg; else if (iv == 1) sig1 = sg; else sig2 = sg;
    }
  } else {
    // abscissae not strictly increasing (degenerate water column): slopes and
    // tensions are defined as zero here and in the oracle; HVAL then returns 0
    // on an interval of non-positive width (IER = -2, tension_module.f90:1060).
    YP[0] = YP[1] = YP[2] = YP[3] = 0.0;
  }
  if (sigerr != 0) {
    sigerr_count += 1;
    return linint4(X, Y, T);
  }
  // interval search: extrapolate with the end intervals, else INTRVL
  // (tension_module.f90:1044-1053,1287-1354) = comparison chain for N = 4
  int I = (T < X[1]) ? 1 : (T < X[2]) ? 2 : 3;
  if (T > X[3]) I = 3;
  double x1 = (I == 1) ? X[0] : (I == 2) ? X[1] : X[2];
  double x2 = (I == 1) ? X[1] : (I == 2) ? X[2] : X[3];
  double y1 = (I == 1) ? Y[0] : (I == 2) ? Y[1] : Y[2];
  double y2 = (I == 1) ? Y[1] : (I == 2) ? Y[2] : Y[3];
  double p1 = (I == 1) ? YP[0] : (I == 2) ? YP[1] : YP[2];
  double p2 = (I == 1) ? YP[1] : (I == 2) ? YP[2] : YP[3];
  double sg = (I == 1) ? sig0 : (I == 2) ? sig1 : sig2;
  if (x2 - x1 <= 0.0) return 0.0;
  return hval_interval(T, x1, x2, y1, y2, p1, p2, sg);
}

// ---------------------------------------------- batched SIGS pieces
// The fused kernel does not iterate inside the per-lane spline code: each
// interval is first CLASSIFIED (no loops), the few that need the convexity
// Newton iteration are handed to a wave-cooperative pass (lt_kernels.h), and
// the monotonicity secant iteration (rare) runs out of line.  The arithmetic of
// every piece is that of SIGS (tension_module.f90:314-781).
enum { SIGK_DONE = 0, SIGK_NEWTON = 1, SIGK_SECANT = 2 };

// Classify one interval: S = (Y2-Y1)/DX as computed by the caller (the same
// expression SIGS evaluates), end slopes S1,S2.  Returns SIGK_DONE with sigma,
// SIGK_NEWTON with T (= MAX(D1/D2,D2/D1) > 2), or SIGK_SECANT.
// T <= 2 is decided without dividing when |big| <= 2|small| (then the rounded
// quotient cannot exceed 2); otherwise the one quotient that is the maximum is
// formed exactly as the reference forms it.
LT_DEV int sigs_classify(double S, double S1, double S2, double& sigma, double& T) {
  double D1 = S - S1;
  double D2 = S2 - S;
  double D1D2 = D1 * D2;
  sigma = 0.0;
  T = 0.0;
  if ((D1D2 == 0.0 && S1 != S2) || (S == 0.0 && S1 * S2 > 0.0)) {
    sigma = LT_SBIG;
    return SIGK_DONE;
  }
  if (D1D2 >= 0.0) {
    if (D1D2 == 0.0) return SIGK_DONE;
    double a = fabs(D1), b = fabs(D2);
    double big = (a > b) ? a : b, small = (a > b) ? b : a;
    if (big <= 2.0 * small) return SIGK_DONE;
    T = (a > b) ? D1 / D2 : D2 / D1;
    if (T <= 2.0) return SIGK_DONE;
    return SIGK_NEWTON;
  }
  if (S1 * S < 0.0 || S2 * S < 0.0) return SIGK_DONE;
  double T0 = 3.0 * S - S1 - S2;
  double D0 = T0 * T0 - S1 * S2;
  if (D0 <= 0.0 || S * T0 >= 0.0) return SIGK_DONE;
  return SIGK_SECANT;
}

// The same classification without its quotient: want_T = the interval is convex with |big| > 2 |small|, and
// num / den are the operands of T = MAX(D1/D2, D2/D1); the caller divides (pass B of k_rk forms the quotients of
// its evaluation and check intervals with ONE division) and decides T <= 2 -> SIGK_DONE, else SIGK_NEWTON.
// Without want_T the returned kind is final.  Straight-line: every comparison is evaluated.
LT_DEV int sigs_classify_nd(double S, double S1, double S2, double& sigma, double& num, double& den, bool& want_T) {
  const double D1 = S - S1, D2 = S2 - S, D1D2 = D1 * D2;
  const bool sbig = ((D1D2 == 0.0) & (S1 != S2)) | ((S == 0.0) & (S1 * S2 > 0.0));
  const bool convex = D1D2 >= 0.0;
  const double a = fabs(D1), b = fabs(D2);
  const bool ab = a > b;
  const double big = ab ? a : b, small = ab ? b : a;
  want_T = !sbig & convex & !(D1D2 == 0.0) & !(big <= 2.0 * small);
  num = ab ? D1 : D2;
  den = ab ? D2 : D1;
  const double T0 = 3.0 * S - S1 - S2;
  const double D0 = T0 * T0 - S1 * S2;
  const bool nc_done = (S1 * S < 0.0) | (S2 * S < 0.0) | (D0 <= 0.0) | (S * T0 >= 0.0);
  sigma = sbig ? LT_SBIG : 0.0;
  return (sbig | convex | nc_done) ? SIGK_DONE : SIGK_SECANT;
}

// Division-free screening of one SIGS interval.  Newton CANDIDATE: the data
// are convex on the interval (D1*D2 > 0) and the larger second difference
// exceeds twice the smaller, i.e. T = MAX(D1/D2,D2/D1) is not certainly <= 2.
// num/den are the operands of the one quotient that equals T.
LT_DEV bool sigs_candidate(double S, double S1, double S2, double& num, double& den) {
  const double D1 = S - S1, D2 = S2 - S;
  const double a = fabs(D1), b = fabs(D2);
  const bool ab = a > b;
  num = ab ? D1 : D2;
  den = ab ? D2 : D1;
  return (D1 * D2 > 0.0) && ((ab ? a : b) > 2.0 * (ab ? b : a));
}

// Lane conditions as 64-bit lane masks -- what a vector compare writes -- combined with scalar instructions
// (k_rk's pass A).  lm() must be evaluated with every lane of the wave active; lm_lane() is the lane's own bit.
typedef unsigned long long LaneMask;
LT_DEV LaneMask lm(bool b) { return __builtin_amdgcn_ballot_w64(b); }
LT_DEV bool lm_lane(LaneMask m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

// exchange of two vector registers
LT_DEV void reg_swap(float& a, float& b) { asm volatile("v_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
LT_DEV void reg_swap(unsigned& a, unsigned& b) { asm volatile("v_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
LT_DEV void reg_swap(double& a, double& b) { const double t = a; a = b; b = t; }

// the test alone (the operands of the quotient are formed only for the few candidates)
LT_DEV bool sigs_candidate_test(double S, double S1, double S2) {
  const double D1 = S - S1, D2 = S2 - S;
  const double a = fabs(D1), b = fabs(D2);
  return (D1 * D2 > 0.0) & (__builtin_fmax(a, b) > 2.0 * __builtin_fmin(a, b));
}

LT_DEV LaneMask sigs_candidate_mask(double S, double S1, double S2) {
  const double D1 = S - S1, D2 = S2 - S;
  const double a = fabs(D1), b = fabs(D2);
  return lm(D1 * D2 > 0.0) & lm(__builtin_fmax(a, b) > 2.0 * __builtin_fmin(a, b));
}

// Screening of the EVALUATION interval: true = its tension factor is known
// without iterating (sigma = 0 or SBIG); false = needs sigs_classify + an
// iteration (done by the batched pass).
LT_DEV bool sigs_eval_known(double S, double S1, double S2, double& sigma) {
  const double D1 = S - S1, D2 = S2 - S, D1D2 = D1 * D2;
  sigma = 0.0;
  if ((D1D2 == 0.0 && S1 != S2) || (S == 0.0 && S1 * S2 > 0.0)) {
    sigma = LT_SBIG;
    return true;
  }
  if (D1D2 >= 0.0) {
    if (D1D2 == 0.0) return true;
    const double a = fabs(D1), b = fabs(D2);
    return ((a > b) ? a : b) <= 2.0 * ((a > b) ? b : a);
  }
  if (S1 * S < 0.0 || S2 * S < 0.0) return true;
  const double T0 = 3.0 * S - S1 - S2;
  const double D0 = T0 * T0 - S1 * S2;
  return (D0 <= 0.0 || S * T0 >= 0.0);
}

// The same screening as one boolean expression without branches (every comparison is evaluated):
// true = the evaluation interval's tension factor is known to be ZERO, the only case pass A of
// k_rk evaluates at once.  (Known SBIG -- rare -- and everything that iterates go to pass B.)
LT_DEV bool sigs_eval_zero(double S, double S1, double S2) {
  const double D1 = S - S1, D2 = S2 - S, D1D2 = D1 * D2;
  const bool inf_t = ((D1D2 == 0.0) & (S1 != S2)) | ((S == 0.0) & (S1 * S2 > 0.0));
  const double a = fabs(D1), b = fabs(D2);
  const bool conv_zero = (D1D2 == 0.0) | (__builtin_fmax(a, b) <= 2.0 * __builtin_fmin(a, b));
  const double T0 = 3.0 * S - S1 - S2;
  const double D0 = T0 * T0 - S1 * S2;
  const bool nconv_zero = (S1 * S < 0.0) | (S2 * S < 0.0) | (D0 <= 0.0) | (S * T0 >= 0.0);
  const bool conv = D1D2 >= 0.0;
  return !inf_t & ((conv & conv_zero) | (!conv & nconv_zero));
}

// (the same as lane masks)
LT_DEV LaneMask sigs_eval_zero_mask(double S, double S1, double S2) {
  const double D1 = S - S1, D2 = S2 - S, D1D2 = D1 * D2;
  const LaneMask d0 = lm(D1D2 == 0.0);
  const LaneMask inf_t = (d0 & lm(S1 != S2)) | (lm(S == 0.0) & lm(S1 * S2 > 0.0));
  const double a = fabs(D1), b = fabs(D2);
  const LaneMask conv_zero = d0 | lm(__builtin_fmax(a, b) <= 2.0 * __builtin_fmin(a, b));
  const double T0 = 3.0 * S - S1 - S2;
  const double D0 = T0 * T0 - S1 * S2;
  const LaneMask nconv_zero = lm(S1 * S < 0.0) | lm(S2 * S < 0.0) | lm(D0 <= 0.0) | lm(S * T0 >= 0.0);
  const LaneMask conv = lm(D1D2 >= 0.0);
  return ~inf_t & ((conv & conv_zero) | (~conv & nconv_zero));
}

// convexity Newton iteration, tension_module.f90:531-580.  Returns the tension
// factor; failed = true when the iteration count exceeds 10000 (SigErr).
LT_DEV double sigs_newton(double T, bool& failed, int* nit_out = nullptr) {
  const double SBIG = LT_SBIG, RTOL = LT_SIGS_RTOL, FTOL = 0.0;
  double TP1 = T + 1.0;
  double SIG = sqrt(10.0 * T - 20.0);
  int NIT = 0;
  failed = false;
  NewtonCycle cyc(SIG);
  for (;;) {
    if (nit_out) *nit_out = NIT + 1;
    double T1, FP;
    if (SIG <= .5) {
      double SINHM, COSHM, COSHMM;
      snhcsh(SIG, SINHM, COSHM, COSHMM);
      const Divisor ds = prep_div(SINHM);  // two quotients over SINHM
      T1 = div_prep(COSHM, ds);
      FP = T1 + SIG * (div_prep(SIG, ds) - T1 * T1 + 1.0);
    } else {
      double EMS = lt_exp(-SIG);
      double SSM = 1.0 - EMS * (EMS + SIG + SIG);
      const Divisor ds = prep_div(SSM);  // two quotients over SSM
      T1 = div_prep((1.0 - EMS) * (1.0 - EMS), ds);
      FP = T1 + SIG * (div_prep(2.0 * SIG * EMS, ds) - T1 * T1 + 1.0);
    }
    double F = SIG * T1 - TP1;
    NIT = NIT + 1;
    if (NIT > 10000) {
      failed = true;
      return 0.0;
    }
    if (FP <= 0.0) break;
    double DSIG = -F / FP;
    if (fabs(DSIG) <= RTOL * SIG || (F >= 0.0 && F <= FTOL) || fabs(F) <= RTOL) break;
    SIG = SIG + DSIG;
    if (cyc.repeats(SIG)) {
      failed = true;
      return 0.0;
    }
  }
  SIG = dmin(SIG, SBIG);
  return (SIG > 0.0) ? SIG : 0.0;
}

// K Newton iterations of sigs_newton advanced in lockstep by one lane.  The arithmetic of each is
// sigs_newton's operation for operation; what changes is the schedule: every step is evaluated in BOTH
// forms (SIG <= .5: SNHCSH's rational form, tension_module.f90:822-834; else the exponential form) and the
// operands of the two quotients are selected, so the 2K chains of a step interleave in the instruction
// stream instead of running one after the other under exec masks -- the iteration is a serial chain
// sigma -> exp / rational -> reciprocal -> sigma whose latency, not its issue slots, is what a round of the
// cooperative pass costs.  A task that has stopped keeps its SIG; the lane leaves when all K have.
#define LT_NEWTON_CYC_START 8
template <int K>
LT_DEV void sigs_newton_k(const double (&T)[K], const bool (&need)[K], double (&sg)[K], bool (&failed)[K]) {
  const double SBIG = LT_SBIG, RTOL = LT_SIGS_RTOL, FTOL = 0.0;
  const double P1 = -3.51754964808151394800e5, P2 = -1.15614435765005216044e4,
               P3 = -1.63725857525983828727e2, P4 = -7.89474443963537015605e-1;
  const double Q1 = -2.11052978884890840399e6, Q2 = 3.61578279834431989373e4,
               Q3 = -2.77711081420602794433e2, Q4 = 1.0;
  double SIG[K], TP1[K], mark[K];
  int power[K], lam[K];
  bool live[K];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    TP1[j] = T[j] + 1.0;
    SIG[j] = need[j] ? sqrt(10.0 * T[j] - 20.0) : 1.0;
    mark[j] = 0.0; power[j] = 1; lam[j] = 0;
    live[j] = need[j];
    failed[j] = false;
  }
  int NIT = 0;
  for (;;) {
#ifdef LT_RK_MARK
    asm volatile("; MARK 30");
#endif
    NIT = NIT + 1;
    bool any = false;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const double s = SIG[j];
      const bool small = s <= .5;
      // SNHCSH, |X| <= .5 (skipped by the whole wave when none of its live tasks is there)
      double SINHM = 1.0, COSHM = 1.0;
#ifndef LT_NEWTON_NO_SKIP
      if (__builtin_amdgcn_ballot_w64(small & live[j]) != 0ull)
#endif
      {
        const double XS = s * s, XC = s * XS;
        double P = ((P4 * XS + P3) * XS + P2) * XS + P1;
        double Q = ((Q4 * XS + Q3) * XS + Q2) * XS + Q1;
        SINHM = XC * (P / Q);
        const double XSD4 = .25 * XS, XSD2 = XSD4 + XSD4;
        P = ((P4 * XSD4 + P3) * XSD4 + P2) * XSD4 + P1;
        Q = ((Q4 * XSD4 + Q3) * XSD4 + Q2) * XSD4 + Q1;
        const double Fc = XSD4 * (P / Q);
        const double COSHMM = XSD2 * Fc * (Fc + 2.0);
        COSHM = COSHMM + XSD2;
      }
      // the exponential form
      double EMS = exp_core((s < 512.0) ? -s : 0.0);
      if (s >= 512.0) EMS = lt_exp(-s);  // T > 26000: rare
      const double SSM = 1.0 - EMS * (EMS + s + s);
      const double TM = (1.0 - EMS) * (1.0 - EMS);
      const double N2 = 2.0 * s * EMS;
      const Divisor ds = prep_div(small ? SINHM : SSM);
      const double T1 = div_prep(small ? COSHM : TM, ds);
      const double FP = T1 + s * (div_prep(small ? s : N2, ds) - T1 * T1 + 1.0);
      const double F = s * T1 - TP1[j];
      const bool capped = NIT > 10000;
#ifdef LT_NEWTON_TRUE_DIV
      const double DSIG = -F / FP;
#else
      // (FP is of order one where the step is used -- FP <= 0 stops the iteration -- and F is a rounded
      // difference of such numbers or zero: the prepared-divisor quotient is the value of -F / FP, lt_math.h above)
      const double DSIG = div_prep(-F, prep_div(FP));
#endif
      const bool stop = (FP <= 0.0) | (fabs(DSIG) <= RTOL * s) | ((F >= 0.0) & (F <= FTOL)) | (fabs(F) <= RTOL);
      const bool adv = live[j] & !capped & !stop;
      const double nx = s + DSIG;
      // Cycle detection (NewtonCycle) for a task that moves on.  A sequence that is going to repeat can be
      // caught from any of its points, so the bookkeeping starts at the 8th step (the iteration takes 3.7
      // on average; NIT is the same for the whole wave: a scalar branch) with that iterate as the mark.
      bool rep = adv & (nx != nx);
      if (NIT == LT_NEWTON_CYC_START) {
        mark[j] = nx; power[j] = 1; lam[j] = 0;
      } else if (NIT > LT_NEWTON_CYC_START) {
        rep = rep | (adv & (nx == mark[j]));
        const bool roll = adv & !rep & (lam[j] + 1 == power[j]);
        lam[j] = roll ? 0 : (adv ? lam[j] + 1 : lam[j]);
        mark[j] = roll ? nx : mark[j];
        power[j] = roll ? (power[j] << 1) : power[j];
      }
      failed[j] = failed[j] | (live[j] & capped) | rep;
      SIG[j] = adv ? nx : s;
      live[j] = adv & !rep;
      any = any | live[j];
    }
#ifdef LT_RK_MARK
    asm volatile("; MARK 31");
#endif
    if (!any) break;
  }
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const double v = dmin(SIG[j], SBIG);
    sg[j] = (failed[j] | !(v > 0.0)) ? 0.0 : v;
  }
}

// the three addend constants of exp (lt_libm_impl.h: exp_parts_k) held in vector registers over a loop: opaque to
// the compiler, which would otherwise re-materialise each with two moves per evaluation
struct ExpAcc { double Shift, C2, C4; };
LT_DEV ExpAcc exp_acc() {
  ExpAcc a;
  a.Shift = 0x1.8p+52; a.C2 = 0x1.ffffffffffdbdp-2; a.C4 = 0x1.55555cf172b91p-5;
  asm volatile("" : "+v"(a.Shift), "+v"(a.C2), "+v"(a.C4));
  return a;
}

// sigs_newton_k<1> for a whole wave, one task per lane, with the lane conditions (live, failed, the stopping
// tests) as lane masks: loop-carried booleans are otherwise kept as 0 / 1 in vector registers and rebuilt with a
// select and a compare in every trip.  Same operations on the same operands, the same cycle detection.  NEED: the
// lanes with a task.  The operands of the two quotients are selected only in the trips in which a live task is at
// SIG <= .5 (a minority): otherwise they are the exponential form's.
LT_DEV void sigs_newton_w(double T, LaneMask NEED, double& sg, LaneMask& FAILED, const ExpAcc& ea) {
  const double SBIG = LT_SBIG, RTOL = LT_SIGS_RTOL, FTOL = 0.0;
  const double P1 = -3.51754964808151394800e5, P2 = -1.15614435765005216044e4,
               P3 = -1.63725857525983828727e2, P4 = -7.89474443963537015605e-1;
  const double Q1 = -2.11052978884890840399e6, Q2 = 3.61578279834431989373e4,
               Q3 = -2.77711081420602794433e2, Q4 = 1.0;
  const double TP1 = T + 1.0;
  double SIG = lm_lane(NEED) ? sqrt(10.0 * T - 20.0) : 1.0;
  double mark = 0.0;
  int power = 1, lam = 0;
  LaneMask LIVE = NEED;
  FAILED = 0ull;
  int NIT = 0;
  for (;;) {
    NIT = NIT + 1;
    const double s = SIG;
    const LaneMask SMALL = lm(s <= .5);
    const bool any_small = (SMALL & LIVE) != 0ull;
    // the exponential form
    double EMS = exp_core_k((s < 512.0) ? -s : 0.0, ea.Shift, ea.C2, ea.C4);
    if (s >= 512.0) EMS = lt_exp(-s);  // T > 26000: rare
    double den = 1.0 - EMS * (EMS + s + s);        // SSM
    double n1 = (1.0 - EMS) * (1.0 - EMS);         // TM
    double n2 = 2.0 * s * EMS;                     // N2
    if (any_small) {
      // SNHCSH, |X| <= .5 (tension_module.f90:822-834)
      const double XS = s * s, XC = s * XS;
      double Pp = ((P4 * XS + P3) * XS + P2) * XS + P1;
      double Qq = ((Q4 * XS + Q3) * XS + Q2) * XS + Q1;
      const double SINHM = XC * (Pp / Qq);
      const double XSD4 = .25 * XS, XSD2 = XSD4 + XSD4;
      Pp = ((P4 * XSD4 + P3) * XSD4 + P2) * XSD4 + P1;
      Qq = ((Q4 * XSD4 + Q3) * XSD4 + Q2) * XSD4 + Q1;
      const double Fc = XSD4 * (Pp / Qq);
      const double COSHMM = XSD2 * Fc * (Fc + 2.0);
      const double COSHM = COSHMM + XSD2;
      const bool small = lm_lane(SMALL);
      den = small ? SINHM : den;
      n1 = small ? COSHM : n1;
      n2 = small ? s : n2;
    }
    const Divisor ds = prep_div(den);
    const double T1 = div_prep(n1, ds);
    const double FP = T1 + s * (div_prep(n2, ds) - T1 * T1 + 1.0);
    const double F = s * T1 - TP1;
    const bool capped = NIT > 10000;
    // (FP is of order one where the step is used -- FP <= 0 stops the iteration -- and F is a rounded
    // difference of such numbers or zero: the prepared-divisor quotient is the value of -F / FP, lt_math.h above)
    const double DSIG = div_prep(-F, prep_div(FP));
    const LaneMask STOP = lm(FP <= 0.0) | lm(fabs(DSIG) <= RTOL * s) | (lm(F >= 0.0) & lm(F <= FTOL)) | lm(fabs(F) <= RTOL);
    const LaneMask ADV = capped ? 0ull : (LIVE & ~STOP);
    const double nx = s + DSIG;
    // Cycle detection (NewtonCycle) for a task that moves on, from the 8th step (sigs_newton_k)
    LaneMask REP = ADV & lm(nx != nx);
    if (NIT == LT_NEWTON_CYC_START) {
      mark = nx; power = 1; lam = 0;
    } else if (NIT > LT_NEWTON_CYC_START) {
      REP = REP | (ADV & lm(nx == mark));
      const LaneMask ROLL = ADV & ~REP & lm(lam + 1 == power);
      const bool roll = lm_lane(ROLL);
      lam = roll ? 0 : (lm_lane(ADV) ? lam + 1 : lam);
      mark = roll ? nx : mark;
      power = roll ? (power << 1) : power;
    }
    FAILED = FAILED | (capped ? LIVE : 0ull) | REP;
    SIG = lm_lane(ADV) ? nx : s;
    LIVE = ADV & ~REP;
    if (LIVE == 0ull) break;
  }
  const double v = dmin(SIG, SBIG);
  sg = (lm_lane(FAILED) | !(v > 0.0)) ? 0.0 : v;
}

// monotonicity secant iteration, tension_module.f90:638-760 (out of line: rare)
__device__ __noinline__ double sigs_secant(double S, double S1, double S2) {
  const double SBIG = LT_SBIG, RTOL = LT_SIGS_RTOL, FTOL = 0.0;
  double D1 = S - S1;
  double D2 = S2 - S;
  double T0 = 3.0 * S - S1 - S2;
  double D0 = T0 * T0 - S1 * S2;
  double SGN = fsign1(S);
  double SIG = SBIG;
  double FMAX = SGN * (SIG * S - S1 - S2) / (SIG - 2.0);
  if (FMAX <= 0.0) return SBIG;
  double STOL = RTOL * SIG;
  double F = FMAX;
  double F0 = SGN * D0 / (3.0 * (D1 - D2));
  double FNEG = F0;
  double DSIG = SIG;
  double DMAX = SIG;
  double D1PD2 = D1 + D2;
  double A = 0.0, E = 0.0, C1, C2;
  for (;;) {
    DSIG = -F * DSIG / (F - F0);
    if (fabs(DSIG) > fabs(DMAX) || DSIG * DMAX > 0.) {
      DSIG = DMAX;
      F0 = FNEG;
      continue;
    }
    if (fabs(DSIG) < STOL / 2.0) DSIG = -copysign(STOL / 2.0, DMAX);
    SIG = SIG + DSIG;
    F0 = F;
    bool CONT = true;
    if (SIG <= .5) {
      double SINHM, COSHM, COSHMM;
      snhcsh(SIG, SINHM, COSHM, COSHMM);
      C1 = SIG * COSHM * D2 - SINHM * D1PD2;
      C2 = SIG * (SINHM + SIG) * D2 - COSHM * D1PD2;
      A = C2 - C1;
      E = SIG * SINHM - COSHMM - COSHMM;
    } else {
      double EMS = exp_core(-SIG);
      double EMS2 = EMS + EMS;
      double TM = 1.0 - EMS;
      double SSINH = TM * (1.0 + EMS);
      double SSM = SSINH - SIG * EMS2;
      double SCM = TM * TM;
      C1 = SIG * SCM * D2 - SSM * D1PD2;
      C2 = SIG * SSINH * D2 - SCM * D1PD2;
      F = FMAX;
      if (C1 * (SIG * SCM * D1 - SSM * D1PD2) >= 0.0) CONT = false;
      if (CONT) A = EMS2 * (SIG * TM * D2 + (TM - SIG) * D1PD2);
      if (A * (C2 + C1) < 0.0) CONT = false;
      if (CONT) E = SIG * SSINH - SCM - SCM;
    }
    if (CONT) F = (SGN * (E * S2 - C2) + sqrt(A * (C2 + C1))) / E;
    STOL = RTOL * SIG;
    if (fabs(DMAX) <= STOL || (F >= 0.0 && F <= FTOL) || fabs(F) <= RTOL) break;
    DMAX = DMAX + DSIG;
    if (F0 * F > 0.0 && fabs(F) >= fabs(F0)) {
      DSIG = DMAX;
      F0 = FNEG;
      continue;
    }
    if (F0 * F <= 0.0) {
      double T1 = DMAX, T2 = FNEG;
      DMAX = DSIG;
      FNEG = F0;
      if (fabs(DSIG) > fabs(T1) && fabs(F) < fabs(T2)) {
        DSIG = T1;
        F0 = T2;
      }
    }
  }
  SIG = dmin(SIG, SBIG);
  return (SIG > 0.0) ? SIG : 0.0;
}

// HVAL on one interval with the chord slope S = (Y2-Y1)/DX supplied by the
// caller (same expression as tension_module.f90:1074).
LT_DEV double hval_interval_s(double T, double X1, double X2, double Y1, double YP1, double YP2,
                              double S, double SIGMA, const ExpAcc* ea = nullptr) {
  double DX = X2 - X1;
  double U = T - X1;
  double B2 = U / DX;
  double B1 = 1.0 - B2;
  double S1 = YP1;
  double D1 = S - S1;
  double D2 = YP2 - S;
  double SIG = fabs(SIGMA);
  if (SIG < 1.e-9) {
    return Y1 + U * (S1 + B2 * (D1 + B1 * (D1 - D2)));
  } else if (SIG <= .5) {
    double SB2 = SIG * B2, SM, CM, CMM, SM2, CM2, DUMMY;
    snhcsh(SIG, SM, CM, CMM);
    snhcsh(SB2, SM2, CM2, DUMMY);
    double E = SIG * SM - CMM - CMM;
    return Y1 + S1 * U +
           DX * ((CM * SM2 - SM * CM2) * (D1 + D2) + SIG * (CM * CM2 - (SM + SIG) * SM2) * D1) /
               (SIG * E);
  } else {
    double SB1 = SIG * B1;
    double SB2 = SIG - SB1;
    if (-SB1 > LT_SBIG || -SB2 > LT_SBIG) return Y1 + S * U;
    const double E1 = ea ? exp_core_k(-SB1, ea->Shift, ea->C2, ea->C4) : exp_core(-SB1);
    const double E2 = ea ? exp_core_k(-SB2, ea->Shift, ea->C2, ea->C4) : exp_core(-SB2);
    double EMS = E1 * E2;
    double TM = 1.0 - EMS;
    double TS = TM * TM;
    double TP = 1.0 + EMS;
    double E = TM * (SIG * TP - TM - TM);
    return Y1 + S * U +
           DX * (TM * (TP - E1 - E2) * (D1 + D2) +
                 SIG * ((E2 + EMS * (E1 - 2.0) - B1 * TS) * D1 +
                        (E1 + EMS * (E2 - 2.0) - B2 * TS) * D2)) /
               (SIG * E);
  }
}

// The zero-tension (cubic) form of HVAL alone, with U / DX through the interval's prepared divisor:
// what pass A of k_rk evaluates at once when the interval's tension is known to be zero.
LT_DEV double hval_cubic_d(double T, double X1, const Divisor& dDX, double Y1, double YP1, double YP2, double S) {
  const double U = T - X1;
  const double B2 = div_prep(U, dDX);
  const double B1 = 1.0 - B2;
  const double D1 = S - YP1;
  const double D2 = YP2 - S;
  return Y1 + U * (YP1 + B2 * (D1 + B1 * (D1 - D2)));
}

// ------------------------------------------------------------ GRIDCELL_MOD
// gridcell for ONE element (checkele form), gridcell_module.f90:26-257.
// qx/qy: the four corners in element order.  Returns true when the reference
// sets triangle = 1.
LT_DEV bool gc_horiz(double ya, double yb, double xa, double xb, double Xpos, double Ypos,
                     bool& hit) {
  if (ya == yb && Ypos == ya) {
    hit = (xa > xb && Xpos > xb && Xpos < xa) || (xb > xa && Xpos > xa && Xpos < xb);
    return true;  // decided (hit, or "cycle")
  }
  return false;
}

// One element of gridcell (gridcell_module.f90:46-254).  0: the point is not in the element;
// 1: it is, and the scan over elements ends here (EXIT of the element loop); 2: it lies on an
// edge found by the crossing loop with an even crossing count so far -- P_ele and triangle are
// set but only the EDGE loop is left (:224,:238), so a scan over all elements goes on and a
// later element may overwrite P_ele.  With checkele (one element) 1 and 2 are both "found".
LT_DEV int gridcell_kind(const double (&qx)[4], const double (&qy)[4], double Xpos, double Ypos) {
  // 1, 2: bounding tests :49-59
  if ((Ypos < qy[0] && Ypos < qy[1] && Ypos < qy[2] && Ypos < qy[3]) ||
      (Ypos > qy[0] && Ypos > qy[1] && Ypos > qy[2] && Ypos > qy[3]))
    return 0;
  if ((Xpos < qx[0] && Xpos < qx[1] && Xpos < qx[2] && Xpos < qx[3]) ||
      (Xpos > qx[0] && Xpos > qx[1] && Xpos > qx[2] && Xpos > qx[3]))
    return 0;
  // 3: on a node :61-69
  if ((Xpos == qx[0] && Ypos == qy[0]) || (Xpos == qx[1] && Ypos == qy[1]) ||
      (Xpos == qx[2] && Ypos == qy[2]) || (Xpos == qx[3] && Ypos == qy[3]))
    return 1;
  // 4: horizontal segments :71-149 (pairs in the source order)
  if (qy[0] == qy[1] || qy[0] == qy[2] || qy[0] == qy[3] || qy[1] == qy[2] || qy[1] == qy[3] ||
      qy[2] == qy[3]) {
    bool hit;
    if (gc_horiz(qy[0], qy[1], qx[0], qx[1], Xpos, Ypos, hit)) return hit ? 1 : 0;
    if (gc_horiz(qy[0], qy[2], qx[0], qx[2], Xpos, Ypos, hit)) return hit ? 1 : 0;
    if (gc_horiz(qy[0], qy[3], qx[0], qx[3], Xpos, Ypos, hit)) return hit ? 1 : 0;
    if (gc_horiz(qy[1], qy[2], qx[1], qx[2], Xpos, Ypos, hit)) return hit ? 1 : 0;
    if (gc_horiz(qy[1], qy[3], qx[1], qx[3], Xpos, Ypos, hit)) return hit ? 1 : 0;
    if (gc_horiz(qy[2], qy[3], qx[2], qx[3], Xpos, Ypos, hit)) return hit ? 1 : 0;
  }
  // 5: y equal to a corner y :151-183
  if (Ypos == qy[0] || Ypos == qy[1] || Ypos == qy[2] || Ypos == qy[3]) {
    double bhighy = fmax(fmax(qy[0], qy[1]), fmax(qy[2], qy[3]));
    if (Ypos == bhighy) return 0;
    double blowy = fmin(fmin(qy[0], qy[1]), fmin(qy[2], qy[3]));
    if (Ypos == blowy) return 0;
  }
  // 6: crossing count :185-254.  An on-edge hit sets triangle=1 and leaves the
  // edge loop; the parity test that follows cannot unset it.
  int total = 0;
  bool on_edge = false;
#pragma unroll
  for (int p = 0; p < 4; p++) {
    if (on_edge) break;
    int q = (p + 1) & 3;
    double bx1 = qx[p], by1 = qy[p], bx2 = qx[q], by2 = qy[q];
    if (Xpos <= bx1 || Xpos <= bx2) {
      if ((by1 > by2 && Ypos >= by2 && Ypos <= by1) || (by2 > by1 && Ypos >= by1 && Ypos <= by2)) {
        if (bx1 == bx2) {
          if (Xpos == bx1) {
            on_edge = true;
          } else if (Ypos != by2) {
            total += 1;
          }
        } else {
          double slope = ((by1 - by2) / (bx1 - bx2));
          double xintersect = (Ypos - by1 + (slope * bx1)) / slope;
          if (xintersect > Xpos && Ypos != by2) total += 1;
          if (xintersect == Xpos) on_edge = true;
        }
      }
    }
  }
  return ((total & 1) != 0) ? 1 : (on_edge ? 2 : 0);
}
LT_DEV bool gridcell_one(const double (&qx)[4], const double (&qy)[4], double Xpos, double Ypos) {
  return gridcell_kind(qx, qy, Xpos, Ypos) != 0;
}

// ---------------------------------------------------- HYDRO_MOD weights
// Interpolation weights inside a quadrilateral element with the reference's
// three regimes (hydrodynamic_module.f90:1707-1738 = :2527-2563).  Expressed
// as four node weights w1..w4 is NOT what the reference does (it evaluates
// v1+(v2-v1)*t+(v3-v1)*u), so the regime and (t,u)/(W) are kept and applied by
// apply() in the reference's own operation order.
struct QuadWeights {
  int regime;  // 1: first triangle, 2: second triangle, 3: inverse distance, 4: on a node
  double t, u, W1, W2, W3, W4;
  int node_hit;  // regime 4: last matching node (1..4), as :2543-2546 overwrite in order
};

// the point is in neither triangle (hydrodynamic_module.f90:1718-1738 = :2540-2563): on a node, or
// inverse-distance weights
LT_DEV void quad_weights_outside(QuadWeights& w, double xp, double yp, double x1, double x2, double x3, double x4,
                                 double y1, double y2, double y3, double y4) {
  bool n1 = (xp == x1 && yp == y1), n2 = (xp == x2 && yp == y2), n3 = (xp == x3 && yp == y3),
       n4 = (xp == x4 && yp == y4);
  if (n1 || n2 || n3 || n4) {
    w.regime = 4;
    if (n1) { w.W1 = 1.0; w.node_hit = 1; }
    if (n2) { w.W2 = 1.0; w.node_hit = 2; }
    if (n3) { w.W3 = 1.0; w.node_hit = 3; }
    if (n4) { w.W4 = 1.0; w.node_hit = 4; }
  } else {
    double Dis1 = 1. / (sqrt((x1 - xp) * (x1 - xp) + (y1 - yp) * (y1 - yp)));
    double Dis2 = 1. / (sqrt((x2 - xp) * (x2 - xp) + (y2 - yp) * (y2 - yp)));
    double Dis3 = 1. / (sqrt((x3 - xp) * (x3 - xp) + (y3 - yp) * (y3 - yp)));
    double Dis4 = 1. / (sqrt((x4 - xp) * (x4 - xp) + (y4 - yp) * (y4 - yp)));
    double TDis = Dis1 + Dis2 + Dis3 + Dis4;
    w.W1 = Dis1 / TDis;
    w.W2 = Dis2 / TDis;
    w.W3 = Dis3 / TDis;
    w.W4 = Dis4 / TDis;
    w.regime = 3;
  }
}

// the same for k_rk's fast path: the four weights divide by the same sum and the four distances are inverted --
// twelve divisions' worth of scaling, reciprocal and fix-up steps, of which prep_div / div_prep keep the part that
// acts on operands of ordinary magnitude (metres and their reciprocals: bit-identical to `/` there)
LT_DEV void quad_weights_outside_f(QuadWeights& w, double xp, double yp, double x1, double x2, double x3, double x4,
                                   double y1, double y2, double y3, double y4) {
  const bool n1 = (xp == x1) & (yp == y1), n2 = (xp == x2) & (yp == y2), n3 = (xp == x3) & (yp == y3),
             n4 = (xp == x4) & (yp == y4);
  if (n1 || n2 || n3 || n4) {
    w.regime = 4;
    if (n1) { w.W1 = 1.0; w.node_hit = 1; }
    if (n2) { w.W2 = 1.0; w.node_hit = 2; }
    if (n3) { w.W3 = 1.0; w.node_hit = 3; }
    if (n4) { w.W4 = 1.0; w.node_hit = 4; }
  } else {
    const double Dis1 = div_prep(1., prep_div(sqrt((x1 - xp) * (x1 - xp) + (y1 - yp) * (y1 - yp))));
    const double Dis2 = div_prep(1., prep_div(sqrt((x2 - xp) * (x2 - xp) + (y2 - yp) * (y2 - yp))));
    const double Dis3 = div_prep(1., prep_div(sqrt((x3 - xp) * (x3 - xp) + (y3 - yp) * (y3 - yp))));
    const double Dis4 = div_prep(1., prep_div(sqrt((x4 - xp) * (x4 - xp) + (y4 - yp) * (y4 - yp))));
    const Divisor dT = prep_div(Dis1 + Dis2 + Dis3 + Dis4);
    w.W1 = div_prep(Dis1, dT);
    w.W2 = div_prep(Dis2, dT);
    w.W3 = div_prep(Dis3, dT);
    w.W4 = div_prep(Dis4, dT);
    w.regime = 3;
  }
}

LT_DEV QuadWeights quad_weights(double xp, double yp, const double (&x)[4], const double (&y)[4]) {
  QuadWeights w;
  w.W1 = w.W2 = w.W3 = w.W4 = 0.0;
  w.node_hit = 0;
  double x1 = x[0], x2 = x[1], x3 = x[2], x4 = x[3], y1 = y[0], y2 = y[1], y3 = y[2], y4 = y[3];
  w.t = ((xp - x1) * (y3 - y1) + (y1 - yp) * (x3 - x1)) / ((x2 - x1) * (y3 - y1) - (y2 - y1) * (x3 - x1));
  w.u = ((xp - x1) * (y2 - y1) + (y1 - yp) * (x2 - x1)) / ((x3 - x1) * (y2 - y1) - (y3 - y1) * (x2 - x1));
  w.regime = 1;
  if (w.t < 0. || w.u < 0. || (w.t + w.u) > 1.0) {
    w.t = ((xp - x3) * (y1 - y3) + (y3 - yp) * (x1 - x3)) / ((x4 - x3) * (y1 - y3) - (y4 - y3) * (x1 - x3));
    w.u = ((xp - x3) * (y4 - y3) + (y3 - yp) * (x4 - x3)) / ((x1 - x3) * (y4 - y3) - (y1 - y3) * (x4 - x3));
    w.regime = 2;
    if (w.t < 0. || w.u < 0. || (w.t + w.u) > 1.0) quad_weights_outside(w, xp, yp, x1, x2, x3, x4, y1, y2, y3, y4);
  }
  return w;
}

// quad_weights for k_rk's fast path: the two denominators of a triangle are (x2-x1)(y3-y1) - (y2-y1)(x3-x1)
// and (x3-x1)(y2-y1) - (y3-y1)(x2-x1) -- the same two rounded products subtracted in the other order,
// i.e. exactly the negative of each other -- so one refined reciprocal serves both quotients
// (prep_div / div_prep: bit-identical to `/` for operands of ordinary magnitude, here metres).
LT_DEV QuadWeights quad_weights_f(double xp, double yp, const double (&x)[4], const double (&y)[4]) {
  QuadWeights w;
  w.W1 = w.W2 = w.W3 = w.W4 = 0.0;
  w.node_hit = 0;
  const double x1 = x[0], x2 = x[1], x3 = x[2], x4 = x[3], y1 = y[0], y2 = y[1], y3 = y[2], y4 = y[3];
  {
    const Divisor d = prep_div((x2 - x1) * (y3 - y1) - (y2 - y1) * (x3 - x1));
    w.t = div_prep((xp - x1) * (y3 - y1) + (y1 - yp) * (x3 - x1), d);
    w.u = div_prep((xp - x1) * (y2 - y1) + (y1 - yp) * (x2 - x1), neg_div(d));
  }
  w.regime = 1;
  // (the three comparisons as lane masks: written with || the compiler forms min(t, u) first, quieting both operands)
  if (lm_lane(lm(w.t < 0.) | lm(w.u < 0.) | lm((w.t + w.u) > 1.0))) {
    const Divisor d = prep_div((x4 - x3) * (y1 - y3) - (y4 - y3) * (x1 - x3));
    w.t = div_prep((xp - x3) * (y1 - y3) + (y3 - yp) * (x1 - x3), d);
    w.u = div_prep((xp - x3) * (y4 - y3) + (y3 - yp) * (x4 - x3), neg_div(d));
    w.regime = 2;
    if (lm_lane(lm(w.t < 0.) | lm(w.u < 0.) | lm((w.t + w.u) > 1.0))) quad_weights_outside_f(w, xp, yp, x1, x2, x3, x4, y1, y2, y3, y4);
  }
  return w;
}

// The free-slip substitution of getInterp (hydrodynamic_module.f90:1934-1994) and interp (:2329-2520):
// values on land nodes of the element are replaced by values of its water nodes before the
// interpolation.  m: bit k = node k+1 is water.  kind 0: rho grid.  kind 1: interp's u-grid branch,
// which tests umasksum.EQ.1 where "one land node" is meant (:2408).  kind 2: interp's v-grid branch,
// whose two diagonal cases read v_mask at the U element's node ids (:2500-2503; d: those four masks).
LT_DEV void freeslip_sub(int kind, unsigned m, unsigned d, double& v1, double& v2, double& v3, double& v4) {
  m &= 15u;
  if (m == 15u) return;
  const int sum = __popc(m);
  const bool l1 = !(m & 1u), l2 = !(m & 2u), l3 = !(m & 4u), l4 = !(m & 8u);
  if (sum == ((kind == 1) ? 1 : 3)) {
    if (l1) v1 = 0.5 * (v2 + v4);
    else if (l2) v2 = 0.5 * (v1 + v3);
    else if (l3) v3 = 0.5 * (v2 + v4);
    else if (l4) v4 = 0.5 * (v1 + v3);
  } else if (sum == 2) {
    if (l1 && l2) { v1 = v4; v2 = v3; }
    else if (l2 && l3) { v2 = v1; v3 = v4; }
    else if (l3 && l4) { v3 = v2; v4 = v1; }
    else if (l4 && l1) { v4 = v3; v1 = v2; }
    else {
      const unsigned a = (kind == 2) ? (d & 15u) : m;
      if (!(a & 1u) && !(a & 4u)) { v1 = v4; v3 = v2; }
      else if (!(a & 8u) && !(a & 2u)) { v4 = v1; v2 = v3; }
    }
  } else if (sum == 1 && kind != 1) {
    if (m & 1u) { v2 = v1; v3 = v1; v4 = v1; }
    else if (m & 2u) { v1 = v2; v3 = v2; v4 = v2; }
    else if (m & 4u) { v1 = v3; v2 = v3; v4 = v3; }
    else { v1 = v4; v2 = v4; v3 = v4; }
  }
}

// interp() value formula, hydrodynamic_module.f90:2530-2561
// apply_interp for values whose node order was already rotated to (3,4,1,2) when the weights
// are those of the second triangle (regime 2): both triangle forms are v1+(v2-v1)t+(v3-v1)u
LT_DEV double apply_interp_rot(const QuadWeights& w, double v1, double v2, double v3, double v4) {
  double r = v1 + (v2 - v1) * w.t + (v3 - v1) * w.u;
  if (w.regime >= 3) {
    if (w.regime == 3) r = w.W1 * v1 + w.W2 * v2 + w.W3 * v3 + w.W4 * v4;
    else r = (w.node_hit == 1) ? v1 : (w.node_hit == 2) ? v2 : (w.node_hit == 3) ? v3 : v4;
  }
  return r;
}
LT_DEV double apply_interp(const QuadWeights& w, double v1, double v2, double v3, double v4) {
  // regimes 1 and 2 share one expression with permuted operands
  // (v1+(v2-v1)*t+(v3-v1)*u  |  v3+(v4-v3)*t+(v1-v3)*u): select, then evaluate once
  const bool r2 = (w.regime == 2);
  const double base = r2 ? v3 : v1, a = r2 ? v4 : v2, b = r2 ? v1 : v3;
  double r = base + (a - base) * w.t + (b - base) * w.u;
  if (w.regime >= 3) {
    if (w.regime == 3) r = w.W1 * v1 + w.W2 * v2 + w.W3 * v3 + w.W4 * v4;
    else r = (w.node_hit == 1) ? v1 : (w.node_hit == 2) ? v2 : (w.node_hit == 3) ? v3 : v4;
  }
  return r;
}

// getInterp() value formula, hydrodynamic_module.f90:1997-2003: the on-node
// case keeps tOK = 2, i.e. the second-triangle formula (copied as is).
LT_DEV double apply_getinterp(const QuadWeights& w, double v1, double v2, double v3, double v4) {
  if (w.regime == 1) return v1 + (v2 - v1) * w.t + (v3 - v1) * w.u;
  if (w.regime == 2 || w.regime == 4) return v3 + (v4 - v3) * w.t + (v1 - v3) * w.u;
  return w.W1 * v1 + w.W2 * v2 + w.W3 * v3 + w.W4 * v4;
}

}  // namespace ltd
