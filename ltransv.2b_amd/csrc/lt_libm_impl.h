// lt_libm_impl.h -- the function bodies of lt_libm.h.  Included by lt_libm.h once per table
// placement with LT_LM_NAME(f) and the LT_*_TAB macros set (no include guard on purpose).
//
// Provenance and licences.  These are restatements, operation for operation (the reference's results branch
// on the last bit of what libm returns), of the double-precision algorithms the GNU C
// Library 2.28+ executes on an FMA-capable x86-64: exp / log / pow from the Arm Optimized Routines (MIT licence;
// in glibc under the LGPL 2.1+), sin / cos / sincos / atan from the IBM Accurate Mathematical Library (LGPL 2.1+
// in glibc), log10 in fdlibm's form (Sun Microsystems: "permission to use, copy, modify, and distribute this
// software is freely granted, provided that this notice is preserved").  No glibc source text is included;
// the polynomial coefficients and tables (lt_libm_tables.h) are the algorithms' published constants.
// ------------------------------------------------------------------------------------ exp
// exp(x) = scale * (1 + tmp): the polynomial part and the bits of the scale 2^(k/128); xtail is
// added to the reduced argument (pow's low part; 0 for exp)
// (Shift, C2, C4 -- the three constants that enter as ADDENDS of a fused multiply-add -- as arguments: a caller
// that evaluates exp in a loop keeps them in registers, lt_math.h: ExpAcc; as literals each costs two register
// moves per evaluation, because the accumulating form of the instruction overwrites its addend)
LT_LM void LT_LM_NAME(exp_parts_k)(double x, double xtail, double* tmp, unsigned long long* sbits, double Shift, double C2,
                                   double C4) {
  const double InvLn2N = 0x1.71547652b82fep+7;
  const double NegLn2hiN = -0x1.62e42fefa0000p-8, NegLn2loN = -0x1.cf79abc9e3b3ap-47;
  const double C3 = 0x1.555555555543cp-3, C5 = 0x1.1111167a4d017p-7;
  const double z = LT_LM_FMA(x, InvLn2N, Shift);
  const unsigned long long ki = LT_LM_ASU(z);
  const double kd = z - Shift;
  double r = LT_LM_FMA(kd, NegLn2hiN, x);
  r = LT_LM_FMA(kd, NegLn2loN, r);
  r = xtail + r;
  const double tail = LT_LM_ASD(LT_EXP_TAB[ki & 127u].tail);
  *sbits = LT_EXP_TAB[ki & 127u].sbits + (ki << 45);
  const double p23 = LT_LM_FMA(C3, r, C2);
  const double t3 = r + tail;
  const double r2 = r * r;
  const double p45 = LT_LM_FMA(r, C5, C4);
  double t = LT_LM_FMA(p23, r2, t3);
  const double r4 = r2 * r2;
  *tmp = LT_LM_FMA(r4, p45, t);
}
LT_LM void LT_LM_NAME(exp_parts)(double x, double xtail, double* tmp, unsigned long long* sbits) {
  LT_LM_NAME(exp_parts_k)(x, xtail, tmp, sbits, 0x1.8p+52, 0x1.ffffffffffdbdp-2, 0x1.55555cf172b91p-5);
}
// exp_core with the three addend constants from the caller's registers (|x| < 512)
LT_LM double LT_LM_NAME(exp_core_k)(double x, double Shift, double C2, double C4) {
  double tmp;
  unsigned long long sbits;
  LT_LM_NAME(exp_parts_k)(x, 0.0, &tmp, &sbits, Shift, C2, C4);
  const double scale = LT_LM_ASD(sbits);
  return LT_LM_FMA(scale, tmp, scale);
}
// |x| < 512 (HVAL/HPVAL: |SIG*B| <= 170; SNHCSH; the secant loop, SIG <= 85; x2lon / y2lat)
LT_LM double LT_LM_NAME(exp_core)(double x) {
  double tmp;
  unsigned long long sbits;
  LT_LM_NAME(exp_parts)(x, 0.0, &tmp, &sbits);
  const double scale = LT_LM_ASD(sbits);
  return LT_LM_FMA(scale, tmp, scale);
}
// any x <= 0 the way libm's out-of-line path does it (unfused scale + scale*tmp on a 2^1022-scaled
// value, the double-double fix-up of subnormal results, 0 below -1024): the Newton loop starts at
// SIG = sqrt(10 T - 20) with T unbounded.  x >= 512, which no caller produces: the platform's exp.
LT_LM double LT_LM_NAME(lt_exp)(double x) {
  if (fabs(x) < 512.0) return LT_LM_NAME(exp_core)(x);
  if (!(x < 0.0)) return exp(x);  // x >= 512 or NaN
  if (x <= -1024.0) return 0.0;
  double tmp;
  unsigned long long sbits;
  LT_LM_NAME(exp_parts)(x, 0.0, &tmp, &sbits);
  const double scale = LT_LM_ASD(sbits + (1022ull << 52));
  const double st = scale * tmp;
  double y = scale + st;
  if (y < 1.0) {  // subnormal result: round once, at the final scaling
    double lo = scale - y + st;
    const double hi = 1.0 + y;
    lo = 1.0 - hi + y + lo;
    y = (hi + lo) - 1.0;
    if (y == 0.0) y = 0.0;
  }
  return 0x1p-1022 * y;
}

// ------------------------------------------------------------------------------------ log
// x > 0, finite and normal (the callers: a uniform deviate in (0,1); the mantissa in [1,2) from log10)
LT_LM double LT_LM_NAME(lt_log)(double x) {
  const double A[5] = LT_LOG_A_INIT;
  const unsigned long long ix = LT_LM_ASU(x);
  if (ix - 0x3fee000000000000ull < 0x3090000000000ull) {  // 1 - 2^-4 <= x < 1 + 0x1.09p-4
    const double B[11] = LT_LOG_B_INIT;
    if (ix == 0x3ff0000000000000ull) return 0.0;
    const double r = x - 1.0;
    double p2 = LT_LM_FMA(r, B[2], B[1]);
    double p3 = LT_LM_FMA(r, B[5], B[4]);
    const double r2 = r * r;
    const double p5 = LT_LM_FMA(r, B[8], B[7]);
    p2 = LT_LM_FMA(r2, B[3], p2);
    p3 = LT_LM_FMA(r2, B[6], p3);
    const double r3 = r * r2;
    double p1 = LT_LM_FMA(r2, B[9], p5);
    p1 = LT_LM_FMA(r3, B[10], p1);
    p1 = LT_LM_FMA(p1, r3, p3);
    p1 = LT_LM_FMA(p1, r3, p2);
    const double w = LT_LM_FMA(r, 0x1p27, r);
    const double rhi = LT_LM_FMA(-0x1p27, r, w);
    const double rhi2 = rhi * rhi;
    const double rlo = r - rhi;
    const double hi = LT_LM_FMA(rhi2, B[0], r);
    const double t8 = r - hi;
    const double rr = r + rhi;
    double lo = LT_LM_FMA(rhi2, B[0], t8);
    const double t2 = B[0] * rlo;
    lo = LT_LM_FMA(t2, rr, lo);
    const double y = LT_LM_FMA(p1, r3, lo);
    return hi + y;
  }
  const unsigned long long tmp = ix - 0x3fe6000000000000ull;
  const int i = (int)((tmp >> 45) & 127u);
  const int k = (int)((long long)tmp >> 52);
  const double z = LT_LM_ASD(ix - (tmp & 0xfff0000000000000ull)), kd = (double)k;
  const double invc = LT_LOG_TAB[i].invc, logc = LT_LOG_TAB[i].logc;
  const double r = LT_LM_FMA(z, invc, -1.0);
  const double w = LT_LM_FMA(kd, LT_LOG_LN2HI, logc);
  const double p5 = LT_LM_FMA(r, A[2], A[1]);
  const double hi = r + w;
  const double r2 = r * r;
  double lo = w - hi;
  lo = lo + r;
  lo = LT_LM_FMA(kd, LT_LOG_LN2LO, lo);
  const double r3 = r * r2;
  double q = LT_LM_FMA(r, A[4], A[3]);
  lo = LT_LM_FMA(r2, A[0], lo);
  q = LT_LM_FMA(q, r2, p5);
  const double y = LT_LM_FMA(r3, q, lo);
  return y + hi;
}
// log10 of a positive normal number (fdlibm's e_log10 on the log above)
LT_LM double LT_LM_NAME(lt_log10)(double x) {
  unsigned long long ix = LT_LM_ASU(x);
  int k = (int)(ix >> 52) - 1023;
  const int i = (k < 0) ? 1 : 0;
  k += i;
  const double y = (double)k;
  ix = (ix & 0x000fffffffffffffull) | ((unsigned long long)(0x3ff - i) << 52);
  const double l = LT_LM_NAME(lt_log)(LT_LM_ASD(ix));
  const double a = y * LT_LOG10_2LO;
  const double b = LT_LOG10_IVLN10 * l;
  const double c = y * LT_LOG10_2HI;
  return (a + b) + c;
}

// ------------------------------------------------------------------------------------ pow(x, 0.5)
// x > 0, finite and normal; the platform's sqrt for anything else (0 -> 0 as pow does)
LT_LM double LT_LM_NAME(lt_pow_half)(double x) {
  const unsigned long long ix = LT_LM_ASU(x);
  if (ix - 0x0010000000000000ull >= 0x7fe0000000000000ull) return sqrt(x);  // 0, subnormal, negative, inf, NaN
  const double A[7] = LT_POW_A_INIT;
  const double y = 0.5;
  const unsigned long long tmp = ix - 0x3fe6955500000000ull;
  const int i = (int)((tmp >> 45) & 127u);
  const int k = (int)((long long)tmp >> 52);
  const double z = LT_LM_ASD(ix - (tmp & 0xfff0000000000000ull)), kd = (double)k;
  const double invc = LT_POW_TAB[i].invc, logc = LT_POW_TAB[i].logc, logctail = LT_POW_TAB[i].logctail;
  const double r = LT_LM_FMA(z, invc, -1.0);
  const double t1 = LT_LM_FMA(kd, LT_LOG_LN2HI, logc);
  const double t2 = t1 + r;
  const double lo1 = LT_LM_FMA(kd, LT_LOG_LN2LO, logctail);
  const double lo2 = t1 - t2 + r;
  const double ar = A[0] * r;
  const double ar2 = r * ar;
  const double ar3 = r * ar2;
  const double hi = t2 + ar2;
  const double lo3 = LT_LM_FMA(ar, r, -ar2);
  const double lo4 = t2 - hi + ar2;
  const double p12 = LT_LM_FMA(r, A[2], A[1]);
  const double p34 = LT_LM_FMA(r, A[4], A[3]);
  double p56 = LT_LM_FMA(r, A[6], A[5]);
  p56 = LT_LM_FMA(p56, ar2, p34);
  const double inner = LT_LM_FMA(ar2, p56, p12);
  double lo = lo1 + lo2;
  lo = lo + lo3;
  lo = lo + lo4;
  lo = LT_LM_FMA(ar3, inner, lo);
  const double lhi = hi + lo;
  const double llo = hi - lhi + lo;
  const double ehi = y * lhi;
  double elo = LT_LM_FMA(lhi, y, -ehi);
  elo = LT_LM_FMA(y, llo, elo);
  // exp(ehi + elo): |ehi| <= 355 < 512, so only the tiny-argument case is special
  const unsigned abstop = (unsigned)(LT_LM_ASU(ehi) >> 52) & 0x7ffu;
  if (abstop - 0x3c9u >= 0x3fu) return 1.0 + ehi;  // |ehi| < 2^-54
  double tm;
  unsigned long long sbits;
  LT_LM_NAME(exp_parts)(ehi, elo, &tm, &sbits);
  const double scale = LT_LM_ASD(sbits);
  return LT_LM_FMA(scale, tm, scale);
}

// ------------------------------------------------------------------------------------ sin, cos
// FMA = true: glibc's cos / sin (FMA build).  FMA = false: the same source as sincos is built (no
// contraction).  |x| < 105414350; beyond that the platform's functions.
LT_LM double LT_LM_NAME(lt_sc_taylor)(double xx, double x, double dx, int fma_) {
  if (fma_) {
    double p = LT_LM_FMA(xx, LT_SC_S5, LT_SC_S4);
    p = LT_LM_FMA(xx, p, LT_SC_S3);
    p = LT_LM_FMA(xx, p, LT_SC_S2);
    p = LT_LM_FMA(xx, p, LT_SC_S1);
    double t = LT_LM_FMA(p, x, -(0.5 * dx));
    t = LT_LM_FMA(xx, t, dx);
    return x + t;
  }
  const double p2 = ((((LT_SC_S5 * xx + LT_SC_S4) * xx + LT_SC_S3) * xx + LT_SC_S2) * xx);
  const double p = p2 + LT_SC_S1;
  const double t = ((p * x - 0.5 * dx) * xx + dx);
  return x + t;
}
LT_LM double LT_LM_NAME(lt_sc_do_cos)(double x, double dx, int fma_) {
  if (x < 0) dx = -dx;
  const double u = LT_SC_BIG + fabs(x);
  const int k = (int)(unsigned)LT_LM_ASU(u) * 4;
  x = fabs(x) - (u - LT_SC_BIG) + dx;
  const double xx = x * x;
  const double sn = LT_SINCOS_TAB[k], ssn = LT_SINCOS_TAB[k + 1], cs = LT_SINCOS_TAB[k + 2], ccs = LT_SINCOS_TAB[k + 3];
  if (fma_) {
    const double s = LT_LM_FMA(x * xx, LT_LM_FMA(xx, LT_SC_SN5, LT_SC_SN3), x);
    const double c = xx * LT_LM_FMA(xx, LT_LM_FMA(xx, LT_SC_CS6, LT_SC_CS4), LT_SC_CS2);
    double cor = LT_LM_FMA(-s, ssn, ccs);
    cor = LT_LM_FMA(-c, cs, cor);
    cor = LT_LM_FMA(-s, sn, cor);
    return cs + cor;
  }
  const double s = x + x * xx * (LT_SC_SN3 + xx * LT_SC_SN5);
  const double c = xx * (LT_SC_CS2 + xx * (LT_SC_CS4 + xx * LT_SC_CS6));
  const double cor = (ccs - s * ssn - cs * c) - sn * s;
  return cs + cor;
}
LT_LM double LT_LM_NAME(lt_sc_do_sin)(double x, double dx, int fma_) {
  const double xold = x;
  if (fabs(x) < LT_SC_T126) return LT_LM_NAME(lt_sc_taylor)(x * x, x, dx, fma_);
  if (x <= 0) dx = -dx;
  const double u = LT_SC_BIG + fabs(x);
  const int k = (int)(unsigned)LT_LM_ASU(u) * 4;
  x = fabs(x) - (u - LT_SC_BIG);
  const double xx = x * x;
  const double sn = LT_SINCOS_TAB[k], ssn = LT_SINCOS_TAB[k + 1], cs = LT_SINCOS_TAB[k + 2], ccs = LT_SINCOS_TAB[k + 3];
  double cor;
  if (fma_) {
    const double t = LT_LM_FMA(x * xx, LT_LM_FMA(xx, LT_SC_SN5, LT_SC_SN3), dx);
    const double s = x + t;
    const double c = LT_LM_FMA(x, dx, xx * LT_LM_FMA(xx, LT_LM_FMA(xx, LT_SC_CS6, LT_SC_CS4), LT_SC_CS2));
    cor = LT_LM_FMA(s, ccs, ssn);
    cor = LT_LM_FMA(-c, sn, cor);
    cor = LT_LM_FMA(s, cs, cor);
  } else {
    const double s = x + (dx + x * xx * (LT_SC_SN3 + xx * LT_SC_SN5));
    const double c = x * dx + xx * (LT_SC_CS2 + xx * (LT_SC_CS4 + xx * LT_SC_CS6));
    cor = (ssn + s * ccs - sn * c) + cs * s;
  }
  return copysign(sn + cor, xold);
}
// x - n*pi/2 as a + da, returns n mod 4
LT_LM int LT_LM_NAME(lt_sc_reduce)(double x, double* a, double* da, int fma_) {
  if (fma_) {
    const double t = LT_LM_FMA(x, LT_SC_HPINV, LT_SC_TOINT);
    const double xn = t - LT_SC_TOINT;
    double y = LT_LM_FMA(-xn, LT_SC_MP1, x);
    y = LT_LM_FMA(-xn, LT_SC_MP2, y);
    const double t2 = LT_LM_FMA(-xn, LT_SC_PP3, y);
    double db = LT_LM_FMA(-LT_SC_PP3, xn, y - t2);
    const double b = LT_LM_FMA(-xn, LT_SC_PP4, t2);
    db = db + LT_LM_FMA(-xn, LT_SC_PP4, t2 - b);
    *a = b;
    *da = db;
    return (int)(LT_LM_ASU(t) & 3u);
  }
  const double t = (x * LT_SC_HPINV + LT_SC_TOINT);
  const double xn = t - LT_SC_TOINT;
  const double y = (x - xn * LT_SC_MP1) - xn * LT_SC_MP2;
  double t1 = xn * LT_SC_PP3;
  const double t2 = y - t1;
  double db = (y - t2) - t1;
  t1 = xn * LT_SC_PP4;
  const double b = t2 - t1;
  db += (t2 - b) - t1;
  *a = b;
  *da = db;
  return (int)(LT_LM_ASU(t) & 3u);
}
// do_cos(a, da) or do_sin(a, da) of the FMA build in one body: the table look-up, the reduced
// argument and the two polynomials are common, only the last few operations differ (lanes of one
// wave need both kinds at almost every call)
LT_LM double LT_LM_NAME(lt_sc_eval)(double x, double dx, int is_cos) {
  const double xold = x;
  if (!is_cos && fabs(x) < LT_SC_T126) return LT_LM_NAME(lt_sc_taylor)(x * x, x, dx, 1);
  if (is_cos ? (x < 0) : (x <= 0)) dx = -dx;
  const double u = LT_SC_BIG + fabs(x);
  const int k = (int)(unsigned)LT_LM_ASU(u) * 4;
  x = fabs(x) - (u - LT_SC_BIG);
  if (is_cos) x = x + dx;
  const double xx = x * x;
  const double sn = LT_SINCOS_TAB[k], ssn = LT_SINCOS_TAB[k + 1], cs = LT_SINCOS_TAB[k + 2], ccs = LT_SINCOS_TAB[k + 3];
  const double x3 = x * xx;
  const double ps = LT_LM_FMA(xx, LT_SC_SN5, LT_SC_SN3);
  const double xxpc = xx * LT_LM_FMA(xx, LT_LM_FMA(xx, LT_SC_CS6, LT_SC_CS4), LT_SC_CS2);
  if (is_cos) {
    const double s = LT_LM_FMA(x3, ps, x);
    double cor = LT_LM_FMA(-s, ssn, ccs);
    cor = LT_LM_FMA(-xxpc, cs, cor);
    cor = LT_LM_FMA(-s, sn, cor);
    return cs + cor;
  }
  const double s = x + LT_LM_FMA(x3, ps, dx);
  const double c = LT_LM_FMA(x, dx, xxpc);
  double cor = LT_LM_FMA(s, ccs, ssn);
  cor = LT_LM_FMA(-c, sn, cor);
  cor = LT_LM_FMA(s, cs, cor);
  return copysign(sn + cor, xold);
}
// The four argument ranges of __cos / __sin end in the same two evaluations, do_cos(a, da) or
// do_sin(a, da), with (a, da) = (x, 0), the complement to pi/2, or the reduced argument: the range
// decides (a, da, which one, sign) first, so that lanes of different ranges share the two
// evaluations instead of running three copies of them one after the other.
LT_LM double LT_LM_NAME(lt_cos)(double x) {
  const unsigned k = (unsigned)(LT_LM_ASU(x) >> 32) & 0x7fffffffu;
  if (k >= 0x419921FBu) return cos(x);
  double a = x, da = 0.0;
  int n = 1;  // odd: do_cos; bit 1: negate
  if (k >= 0x400368fdu) {
    n = LT_LM_NAME(lt_sc_reduce)(x, &a, &da, 1) + 1;
  } else if (k >= 0x3feb6000u) {
    const double y = LT_SC_HP0 - fabs(x);
    a = y + LT_SC_HP1;
    da = (y - a) + LT_SC_HP1;
    n = 0;
  }
  double r = LT_LM_NAME(lt_sc_eval)(a, da, n & 1);
  if (n & 2) r = -r;
  return (k < 0x3e400000u) ? 1.0 : r;
}
LT_LM double LT_LM_NAME(lt_sin)(double x) {
  const unsigned k = (unsigned)(LT_LM_ASU(x) >> 32) & 0x7fffffffu;
  if (k >= 0x419921FBu) return sin(x);
  double a = x, da = 0.0;
  int n = 0;  // even: do_sin; bit 1: negate
  int sign_of_x = 0;
  if (k >= 0x400368fdu) {
    n = LT_LM_NAME(lt_sc_reduce)(x, &a, &da, 1);
  } else if (k >= 0x3feb6000u) {
    a = LT_SC_HP0 - fabs(x);
    da = LT_SC_HP1;
    n = 1;
    sign_of_x = 1;
  }
  double r = LT_LM_NAME(lt_sc_eval)(a, da, n & 1);
  if (n & 2) r = -r;
  if (sign_of_x) r = copysign(r, x);
  return (k < 0x3e500000u) ? x : r;
}
// sin and cos of one argument as glibc's sincos (one non-FMA build)
LT_LM void LT_LM_NAME(lt_sincos)(double x, double* sinx, double* cosx) {
  const unsigned k = (unsigned)(LT_LM_ASU(x) >> 32) & 0x7fffffffu;
  if (k < 0x400368fdu) {
    if (k < 0x3e400000u) { *sinx = x; *cosx = 1.0; return; }
    if (k < 0x3feb6000u) { *sinx = LT_LM_NAME(lt_sc_do_sin)(x, 0.0, 0); *cosx = LT_LM_NAME(lt_sc_do_cos)(x, 0.0, 0); return; }
    const double y = LT_SC_HP0 - fabs(x), a = y + LT_SC_HP1, da = (y - a) + LT_SC_HP1;
    *sinx = copysign(LT_LM_NAME(lt_sc_do_cos)(a, da, 0), x);
    *cosx = LT_LM_NAME(lt_sc_do_sin)(a, da, 0);
    return;
  }
  if (k < 0x419921FBu) {
    double a, da;
    const int n = LT_LM_NAME(lt_sc_reduce)(x, &a, &da, 0) & 3;
    if (n == 1 || n == 2) { a = -a; da = -da; }
    const double s = LT_LM_NAME(lt_sc_do_sin)(a, da, 0), c0 = LT_LM_NAME(lt_sc_do_cos)(a, da, 0);
    const double c = (n & 2) ? -c0 : c0;
    if (n & 1) { *cosx = s; *sinx = c; } else { *sinx = s; *cosx = c; }
    return;
  }
  *sinx = sin(x);
  *cosx = cos(x);
}

// ------------------------------------------------------------------------------------ atan
LT_LM double LT_LM_NAME(lt_atan)(double x) {
  const double u = (x < 0) ? -x : x;
  const double TWO52 = 0x1p52, TWO8 = 256.0;
  if (!(u == u)) return x + x;
  if (u < 1.0) {
    if (u < LT_AT_B) {
      if (u < LT_AT_A) return x;
      const double v = x * x;
      double yy = LT_LM_FMA(v, LT_AT_D13, LT_AT_D11);
      yy = LT_LM_FMA(v, yy, LT_AT_D9);
      yy = LT_LM_FMA(v, yy, LT_AT_D7);
      yy = LT_LM_FMA(v, yy, LT_AT_D5);
      yy = LT_LM_FMA(v, yy, LT_AT_D3);
      return LT_LM_FMA(x * v, yy, x);
    }
    const int i = (int)(LT_LM_FMA(u, TWO8, TWO52) - TWO52) - 16;
    const double z = u - LT_ATAN_TAB[i][0];
    double yy = LT_LM_FMA(z, LT_ATAN_TAB[i][6], LT_ATAN_TAB[i][5]);
    yy = LT_LM_FMA(z, yy, LT_ATAN_TAB[i][4]);
    yy = LT_LM_FMA(z, yy, LT_ATAN_TAB[i][3]);
    yy = LT_LM_FMA(z, yy, LT_ATAN_TAB[i][2]);
    return copysign(fabs(LT_LM_FMA(yy, z, LT_ATAN_TAB[i][1])), x);
  }
  if (u < LT_AT_D) {
    const double w = 1.0 / u;
    const double t1 = w * u;
    const double t2 = LT_LM_FMA(u, w, -t1);
    const int i = (int)(LT_LM_FMA(w, TWO8, TWO52) - TWO52) - 16;
    const double e = (1.0 - t1) - t2;
    const double z = LT_LM_FMA(e, w, w - LT_ATAN_TAB[i][0]);
    double yy = LT_LM_FMA(z, LT_ATAN_TAB[i][6], LT_ATAN_TAB[i][5]);
    yy = LT_LM_FMA(z, yy, LT_ATAN_TAB[i][4]);
    yy = LT_LM_FMA(z, yy, LT_ATAN_TAB[i][3]);
    yy = LT_LM_FMA(z, yy, LT_ATAN_TAB[i][2]);
    const double a = LT_LM_FMA(-z, yy, LT_SC_HP1);
    return copysign(fabs((LT_SC_HP0 - LT_ATAN_TAB[i][1]) + a), x);
  }
  if (u < LT_AT_E) {
    const double w = 1.0 / u;
    const double v = w * w;
    const double t1 = u * w;
    double yy = LT_LM_FMA(v, LT_AT_D13, LT_AT_D11);
    yy = LT_LM_FMA(v, yy, LT_AT_D9);
    yy = LT_LM_FMA(v, yy, LT_AT_D7);
    yy = LT_LM_FMA(v, yy, LT_AT_D5);
    yy = LT_LM_FMA(v, yy, LT_AT_D3);
    const double wv = w * v;
    const double t2 = LT_LM_FMA(u, w, -t1);
    yy = wv * yy;
    const double e = (1.0 - t1) - t2;
    const double t3 = LT_SC_HP0 - w;
    const double ww = e * w;
    const double cor = (LT_SC_HP0 > fabs(w)) ? ((LT_SC_HP0 - t3) - w) : (LT_SC_HP0 - (w + t3));
    double r = cor + LT_SC_HP1;
    r = r - ww;
    r = r - yy;
    return copysign(fabs(r + t3), x);
  }
  return (x > 0) ? LT_SC_HP0 : -LT_SC_HP0;
}
