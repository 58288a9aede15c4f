// lt_vturb.h -- HTurb / VTurb helpers and the VTurb kernels (table, SIGS, deviates, the two walks).
// Part of lt_kernels.h (included inside namespace ltd); see the notes there.
// The kernels are not all in one translation unit: LT_VT_DEFINE_A (k_vt_table, k_vt_norm, k_vt_walk_fast) is set by
// lt_rk_kernel.hip, which is compiled without the machine-level loop-invariant code motion (these three gain from
// that: fewer registers held by hoisted constants), LT_VT_DEFINE_B (k_vt_sigs, k_vt_walk) by lt_engine.hip (these lose).
#pragma once

// ---------------------------------------------------------------- turbulence
// one uniform deviate of genrand_real3 (random_module.f90:239-246) from the
// materialised stream
LT_DEV double draw_real3(const StepParams& P, unsigned long long idx) {
  const uint32_t y = ltmt_temper(P.rng_ring[idx & P.rng_mask]);
  return ((double)y + 0.5) / 4294967296.0;
}
// norm(), norm_module.f90:25-39: two uniforms per deviate, cos branch only
// genrand_real1 (random_module.f90:213-219): [0,1]
LT_DEV double draw_real1(const StepParams& P, unsigned long long idx) {
  const uint32_t y = ltmt_temper(P.rng_ring[idx & P.rng_mask]);
  return (double)y / 4294967295.0;
}
LT_DEV double draw_norm(const StepParams& P, unsigned long long idx) {
  const double dev1 = draw_real3(P, idx), dev2 = draw_real3(P, idx + 1);
  return sqrt(-2.0 * lt_log_g(dev1)) * lt_cos_g(2.0 * P.PI * dev2);  // (k_post: two deviates per particle-step)
}

// HVAL / HPVAL on a spline table in memory (element k of array a at a[k*stride]),
// tension_module.f90:980-1253.  `il` caches the last interval like INTRVL's SAVE IL.
// (tab: the records {x, y, yp} of one particle, knot k at tab[k * VT_TS].  Used where TSPSI failed before SIGS ran:
// every tension factor is zero)
#define VT_TS 64  // table layout [wave][knot][lane]: a wave's records of one knot are 1.5 KB contiguous
// The tension factor of an interval is not in the record (r05: nine in ten are zero, and a quarter of the table's bytes
// were theirs).  k_vt_table sets bit k of the particle's 256 (vt_bits) for every interval (k, k+1) it LISTS for
// k_vt_sigs, which writes the factor to vt_sig[same index as the record]: k_vt_walk_fast -- which may run beside
// k_vt_sigs -- stops at a listed interval whether or not its factor has been written yet, k_vt_walk reads the factor
// where the bit is set and takes zero elsewhere.  (A listed interval's factor comes out zero only when T rounds to
// <= 2 in sigs_classify; such a walk then goes to k_vt_walk, whose cubic form is the same expression: same sums
// either way.)
struct VtBits { unsigned long long w0, w1, w2, w3; };
LT_DEV VtBits vt_bits_load(const StepParams& P, int n) {
  const unsigned long long* b = P.vt_bits + (size_t)(n >> 6) * 4 * 64 + (size_t)(n & 63);
  VtBits r;
  r.w0 = b[0]; r.w1 = b[64]; r.w2 = b[128]; r.w3 = b[192];
  return r;
}
LT_DEV bool vt_bit(const VtBits& b, int k) {   // k = 0-based knot, < 256 (ws <= 64: lt_create)
  const unsigned long long w = (k < 128) ? ((k < 64) ? b.w0 : b.w1) : ((k < 192) ? b.w2 : b.w3);
  return ((w >> (k & 63)) & 1ull) != 0ull;
}
template <bool DERIV>
LT_DEV double hval_table(double T, int n, const VtRec* tab, int& il) {
  int I;
  if (T < tab[0].x) I = 1;
  else if (T > tab[(size_t)(n - 1) * VT_TS].x) I = n - 1;
  else {
    // INTRVL :1287-1354: try the previous interval, else bisection
    bool hit = false;
    if (il >= 1 && il < n) hit = (tab[(size_t)(il - 1) * VT_TS].x <= T) && (T < tab[(size_t)il * VT_TS].x);
    if (!hit) {
      int lo = 1, hi = n;
      while (hi > lo + 1) {
        int kk = (lo + hi) >> 1;
        if (T < tab[(size_t)(kk - 1) * VT_TS].x) hi = kk; else lo = kk;
      }
      il = lo;
    }
    I = il;
  }
  const VtRec ra = tab[(size_t)(I - 1) * VT_TS], rb = tab[(size_t)I * VT_TS];
  const double X1 = ra.x, X2 = rb.x, Y1 = ra.y, Y2 = rb.y, P1 = ra.yp, P2 = rb.yp, SG = 0.0;
  if (X2 - X1 <= 0.0) return 0.0;  // IER = -2
  if (DERIV) return hpval_interval(T, X1, X2, Y1, Y2, P1, P2, SG);
  return hval_interval(T, X1, X2, Y1, Y2, P1, P2, SG);
}

// polintd at the three internal times ix(1..3) of one (ex, ey) triple
// (ver_turb_module.f90:236-238,258-260): the two inner quotients of polintd do
// not depend on x, so they are formed once (same values, 5 divisions instead of 9).
// (ns, the node of the time axis closest to x, is the same for every particle: the host
// finds it once per step with polintd's own comparisons, StepParams::pns)
LT_DEV double polintd_shared(double xa1, double xa2, double xa3, double ya1, double ya2, double ya3,
                             double q32, double q21, const DivConst& r13, bool fd, double x, int ns) {
  double c = (xa2 - x) * q32;
  c = c - (xa2 - x) * q21;
  c = div_by(c, xa1 - xa3, r13, fd);
  const double a = (ns == 3) ? q32 : q21;
  // xa(ns) - x picked among the three differences (selects on computed values only)
  const double d1 = xa1 - x, d2 = xa2 - x, d3 = xa3 - x;
  const double bb = (ns == 3) ? d3 : d1;
  const double yns = (ns == 1) ? ya1 : (ns == 2) ? ya2 : ya3;
  const double dns = (ns == 1) ? d1 : (ns == 2) ? d2 : d3;
  return yns + dns * a + bb * c;
}
// (NS > 0: the node is known at compile time and the selects go -- within an external step the three internal
// times usually share their closest record, so polintd3x3 asks once per call, a scalar branch)
template <bool FD, int NS>
LT_DEV void polintd3x3_ns(const StepParams& P, double ya1, double ya2, double ya3, double q32, double q21,
                          double& rb, double& rc, double& rf) {
  const double xa1 = P.ex0, xa2 = P.ex1, xa3 = P.ex2;
  constexpr bool fd = FD;
  rb = polintd_shared(xa1, xa2, xa3, ya1, ya2, ya3, q32, q21, DivConst{P.c_d13[0], P.c_d13[1]}, fd, P.ix0, NS > 0 ? NS : P.pns0);
  rc = polintd_shared(xa1, xa2, xa3, ya1, ya2, ya3, q32, q21, DivConst{P.c_d13[0], P.c_d13[1]}, fd, P.ix1, NS > 0 ? NS : P.pns1);
  rf = polintd_shared(xa1, xa2, xa3, ya1, ya2, ya3, q32, q21, DivConst{P.c_d13[0], P.c_d13[1]}, fd, P.ix2, NS > 0 ? NS : P.pns2);
}
template <bool FD>
LT_DEV void polintd3x3(const StepParams& P, double ya1, double ya2, double ya3,
                       double& rb, double& rc, double& rf) {
  const double xa1 = P.ex0, xa2 = P.ex1, xa3 = P.ex2;
  constexpr bool fd = FD;
  const double q32 = div_by(ya3 - ya2, xa2 - xa3, DivConst{P.c_d23[0], P.c_d23[1]}, fd);
  const double q21 = div_by(ya2 - ya1, xa1 - xa2, DivConst{P.c_d12[0], P.c_d12[1]}, fd);
  const int ns = (P.pns0 == P.pns1 && P.pns1 == P.pns2) ? P.pns0 : 0;
  if (ns == 2) polintd3x3_ns<FD, 2>(P, ya1, ya2, ya3, q32, q21, rb, rc, rf);
  else if (ns == 3) polintd3x3_ns<FD, 3>(P, ya1, ya2, ya3, q32, q21, rb, rc, rf);
  else polintd3x3_ns<FD, 0>(P, ya1, ya2, ya3, q32, q21, rb, rc, rf);
}

// The interval of the VTurb spline table a walk is in, kept in registers: HVAL and
// HPVAL (tension_module.f90:980-1253) reload from HBM only when the depth leaves it.
// For a strictly increasing abscissa table INTRVL's answer (:1287-1354) is the unique
// I with X(I) <= T < X(I+1), clamped to 1..N-1, whatever its saved IL was, so
// stepping from the cached interval finds the same I as its bisection.
// Beside the knot data it holds what HVAL / HPVAL compute from the interval alone (chord
// slope, second differences, the prepared divisor DX and, for 1e-9 <= sigma <= .5, SNHCSH of
// sigma), formed once when the interval is entered: the same operations, hoisted.
// the chord slope of an interval through the reciprocal prepared for its width (a difference of diffusivities over a
// knot spacing > 0: the same value as `/`, lt_math.h) -- LT_VT_SLOPE_DIV: with `/`, for comparison
#ifdef LT_VT_SLOPE_DIV
#define LT_VT_SLOPE(num, den, dv) ((num) / (den))
#else
#define LT_VT_SLOPE(num, den, dv) div_prep((num), (dv))
#endif
struct VtInterval {
  int I;
  double X1, X2, Y1, Y2, P1, P2, SG;
  double S, D1, D2, SM, CM, CMM;
  Divisor dx;
};
LT_DEV void vt_prepare(VtInterval& c) {
  const double DX = c.X2 - c.X1;
  c.dx = prep_div(DX);
  c.S = LT_VT_SLOPE(c.Y2 - c.Y1, DX, c.dx);
  c.D1 = c.S - c.P1;
  c.D2 = c.P2 - c.S;
  const double SIG = fabs(c.SG);
  c.SM = c.CM = c.CMM = 0.0;
  if (SIG >= 1.e-9 && SIG <= .5) snhcsh(SIG, c.SM, c.CM, c.CMM);
}
// HVAL (tension_module.f90:1052-1117) and HPVAL (:1196-1250) on the prepared interval
LT_DEV double hval_prepared(double T, const VtInterval& c) {
  const double DX = c.X2 - c.X1, U = T - c.X1;
  const double B2 = div_prep(U, c.dx), B1 = 1.0 - B2;
  const double S1 = c.P1, S = c.S, D1 = c.D1, D2 = c.D2, SIG = fabs(c.SG);
  if (SIG < 1.e-9) {
    return c.Y1 + U * (S1 + B2 * (D1 + B1 * (D1 - D2)));
  } else if (SIG <= .5) {
    double SB2 = SIG * B2, SM2, CM2, DUMMY;
    const double SM = c.SM, CM = c.CM, CMM = c.CMM;
    snhcsh(SB2, SM2, CM2, DUMMY);
    const double E = SIG * SM - CMM - CMM;
    return c.Y1 + S1 * U +
           DX * ((CM * SM2 - SM * CM2) * (D1 + D2) + SIG * (CM * CM2 - (SM + SIG) * SM2) * D1) / (SIG * E);
  } else {
    const double SB1 = SIG * B1, SB2 = SIG - SB1;
    if (-SB1 > LT_SBIG || -SB2 > LT_SBIG) return c.Y1 + S * U;
    const double E1 = exp_core(-SB1), E2 = exp_core(-SB2);
    const double EMS = E1 * E2, TM = 1.0 - EMS, TS = TM * TM, TP = 1.0 + EMS;
    const double E = TM * (SIG * TP - TM - TM);
    return c.Y1 + S * U +
           DX * (TM * (TP - E1 - E2) * (D1 + D2) +
                 SIG * ((E2 + EMS * (E1 - 2.0) - B1 * TS) * D1 + (E1 + EMS * (E2 - 2.0) - B2 * TS) * D2)) /
               (SIG * E);
  }
}
LT_DEV double hpval_prepared(double T, const VtInterval& c) {
  const double B1 = div_prep(c.X2 - T, c.dx), B2 = 1.0 - B1;
  const double S1 = c.P1, S = c.S, D1 = c.D1, D2 = c.D2, SIG = fabs(c.SG);
  if (SIG < 1.e-9) {
    return S1 + B2 * (D1 + D2 - 3.0 * B1 * (D2 - D1));
  } else if (SIG <= .5) {
    double SB2 = SIG * B2, SM2, CM2, DUMMY;
    const double SM = c.SM, CM = c.CM, CMM = c.CMM;
    snhcsh(SB2, SM2, CM2, DUMMY);
    const double SINH2 = SM2 + SB2;
    const double E = SIG * SM - CMM - CMM;
    return S1 + ((CM * CM2 - SM * SINH2) * (D1 + D2) + SIG * (CM * SINH2 - (SM + SIG) * CM2) * D1) / E;
  } else {
    const double SB1 = SIG * B1, SB2 = SIG - SB1;
    if (-SB1 > LT_SBIG || -SB2 > LT_SBIG) return S;
    const double E1 = exp_core(-SB1), E2 = exp_core(-SB2);
    const double EMS = E1 * E2, TM = 1.0 - EMS;
    const double E = TM * (SIG * (1.0 + EMS) - TM - TM);
    return S + (TM * ((E2 - E1) * (D1 + D2) + TM * (D1 - D2)) +
                SIG * ((E1 * EMS - E2) * D1 + (E1 - E2 * EMS) * D2)) / E;
  }
}
// (a move to the neighbouring interval loads ONE 24-byte record -- the new knot; the knot shared with
// the old interval is kept)
// (SIG: with the interval's tension factor -- sig: the particle's column of vt_sig, indexed like tab, read where the
// interval's bit is set, in the same round trip as the record)
template <bool SIG>
LT_DEV bool vt_seek_move(VtInterval& c, double T, int n, const VtRec* tab, const double* sig, const VtBits& bits) {
  bool moved = false;
  if (c.I == 0) {  // first use: the start of vt_seek_lite (the table is strictly increasing wherever this is called)
    const double xa = tab[0].x, xb = tab[(size_t)(n - 1) * VT_TS].x;
    const double f = (T - xa) / (xb - xa) * (double)(n - 1);
    int lo = (f >= 0.0 && f < (double)(n - 1)) ? (int)f + 1 : ((f >= 0.0) ? n - 1 : 1);
    if (lo < 1) lo = 1;
    if (lo > n - 1) lo = n - 1;
    c.I = lo;
    const VtRec ra = tab[(size_t)(lo - 1) * VT_TS], rb = tab[(size_t)lo * VT_TS];
    if (SIG) c.SG = vt_bit(bits, lo - 1) ? sig[(size_t)(lo - 1) * VT_TS] : 0.0;
    c.X1 = ra.x; c.Y1 = ra.y; c.P1 = ra.yp;
    c.X2 = rb.x; c.Y2 = rb.y; c.P2 = rb.yp;
    moved = true;
  }
  while (T < c.X1 && c.I > 1) {
    c.I -= 1;
    const VtRec r = tab[(size_t)(c.I - 1) * VT_TS];
    if (SIG) c.SG = vt_bit(bits, c.I - 1) ? sig[(size_t)(c.I - 1) * VT_TS] : 0.0;
    c.X2 = c.X1; c.Y2 = c.Y1; c.P2 = c.P1;
    c.X1 = r.x; c.Y1 = r.y; c.P1 = r.yp;
    moved = true;
  }
  while (T >= c.X2 && c.I < n - 1) {
    c.I += 1;
    const VtRec r = tab[(size_t)c.I * VT_TS];
    if (SIG) c.SG = vt_bit(bits, c.I - 1) ? sig[(size_t)(c.I - 1) * VT_TS] : 0.0;
    c.X1 = c.X2; c.Y1 = c.Y2; c.P1 = c.P2;
    c.X2 = r.x; c.Y2 = r.y; c.P2 = r.yp;
    moved = true;
  }
  return moved;
}
LT_DEV void vt_seek(VtInterval& c, double T, int n, const VtRec* tab, const double* sig, const VtBits& bits) {
  if (vt_seek_move<true>(c, T, n, tab, sig, bits)) vt_prepare(c);
}
// The walk of a particle whose SIGS failed (SigErr: linint instead of HVAL / HPVAL, ver_turb_module.f90:296-331) on
// the same cached interval.  linint's bisection (interpolation_module: the j with xa(j) <= x < xa(j+1), the end
// intervals beyond the ends) and the stepping find the same interval of a strictly increasing table -- and the table is
// one whenever YPC1 went through, which it did if SIGS ran.  (A bisection from scratch is seven dependent loads, twice
// per sub-step: the 0.03 % of the particles that take this path were the longest waves of k_vt_walk, 0.2 ms of the step.)
LT_DEV void vt_seek_plain(VtInterval& c, double T, int n, const VtRec* tab) {
  (void)vt_seek_move<false>(c, T, n, tab, nullptr, VtBits{0ull, 0ull, 0ull, 0ull});
}
LT_DEV void linint_interval(const VtInterval& c, double x, double& y, double& m) {
  m = (c.Y1 - c.Y2) / (c.X1 - c.X2);
  const double b = c.Y1 - m * c.X1;
  y = m * x + b;
}

// The spline intervals of the VTurb table whose tension factor needs an iteration (SIGS,
// tension_module.f90:531-760; ~9 % of them) are written by the table kernel to a task list per
// wave in HBM (S, S1, S2, knot, lane: 26 bytes) and iterated by k_vt_sigs, 64 at a time, at the
// occupancy that kernel's few registers allow.  (Iterating them inside the table kernel -- 255
// VGPRs, two waves per SIMD -- cost 1.2 ms per step; a kernel that re-read the whole table to
// find them 0.95 ms.)  A list has room for every interval of its wave.

// VTurb, ver_turb_module.f90:30-380, as four kernels:
//   k_vt_table  one lane per particle: builds the spline table of the time-interpolated, smoothed Kh
//               profile (TSPSI's YPC1) and settles every tension factor that needs no iteration in the
//               same stream over the knots;
//   k_vt_sigs   one wave per task list: the SIGS iterations of the intervals that need one;
//   k_vt_walk_fast / k_vt_walk   one lane per particle: the idt/2 random-walk sub-steps on the table (the
//               walks on zero-tension intervals; the rest of the walks that met a tension factor).
// The ~30 work arrays of the reference are never materialised: the proliferated /
// smoothed profile is produced as a stream over k (8-point window in an LDS ring),
// the YPC1 derivative of knot k-1 is formed as soon as knot k exists, the interval (k-2, k-1)
// is classified as soon as both its end derivatives exist -- its tension factor is zero (the
// usual case), or it goes to the wave's task list -- and knot k-2 goes to
// HBM as ONE 24-byte record {x, y, yp}, [wave][knot][lane] (r02-r04: 32 bytes with the tension factor; r05: the
// listed intervals as a bit per knot, their factors in a plane of their own, above): the table is written once, in
// 1.5 KB rows per wave and knot (per-lane runs of records, [particle][knot], cost 3.6x their
// size in HBM writes: profiles/r02_calib), and never read back by the build; the walk loads one
// record per move to a neighbouring interval.  (Four separate arrays, re-read by a SIGS kernel and
// gathered 8 bytes at a time by the walk, moved 14.9 GB per step for 3.5 GB of algorithmic traffic.)
// SIGS' "RETURN at the first SigErr" leaves later tensions at 0 in the reference;
// since SigErr switches every later evaluation to linint, which ignores the
// tensions, computing all of them here changes nothing.
// FD: the per-step divisors (4*ws, 6, the ex differences) have usable reciprocals (div_by);
// a template parameter so that the ~20 quotients per knot carry no branch
#ifdef LT_VT_DEFINE_A
template <typename SlabT, int VT, bool FD>
__global__ __launch_bounds__(256, 2) void k_vt_table(const StepParams P) {
  __shared__ double lev[6 * LT_LEV_STRIDE];
  __shared__ double wring[3][8][256];
  __shared__ unsigned long long fbits[4][256];   // the lane's listed intervals (vt_bits), gathered here: no registers
  for (int i = threadIdx.x; i < 6 * LT_LEV_STRIDE; i += blockDim.x) lev[i] = P.lev[i];
#pragma unroll
  for (int w = 0; w < 4; w++) fbits[w][threadIdx.x] = 0ull;
  exp_table_load();
  if (blockIdx.x == 0 && threadIdx.x == 0) *P.vt_wl_count = 0;  // the walk kernels' list (they run after this one)
#ifndef LT_VT_NO_XCD_MAP
  // workgroups go round-robin to the 8 XCDs: XCD x takes the x-th contiguous eighth of the particles (any grid size),
  // so that the Kh columns its waves stream up are shared in ITS L2 (a shard of configs[3] 7.00 -> 6.98 ms; nothing
  // at 7.6 particles per element)
  const unsigned xq_ = gridDim.x >> 3, xr_ = gridDim.x & 7u, xx_ = blockIdx.x & 7u;
  const unsigned vb_ = xx_ * xq_ + (xx_ < xr_ ? xx_ : xr_) + (blockIdx.x >> 3);
  const int n = vb_ * blockDim.x + threadIdx.x;
#else
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
#endif
  const int lane = threadIdx.x & 63;
  const int wave = n >> 6;
  // (every lane of the wave stays to the end: the task list is filled with wave-wide ballots)
  const int fl = (n < P.numpar) ? P.ws_i[n] : 0;
  const bool act = (fl & FL_ACT) != 0;
  const size_t N = (size_t)P.numpar;
  const int na = act ? n : 0;
  double* wsd = P.ws_d + na;
  const double* wtab = lev + 3 * LT_LEV_STRIDE;
  const int ws = P.ws, p2 = 4 * P.ws;
  double P_depth = -1.0, h = 1.0;
  double zeta[3] = {0.0, 0.0, 0.0};
  int4 rn = make_int4(1, 1, 1, 1);
  QuadWeights w0;
  w0.regime = 1; w0.node_hit = 0; w0.t = w0.u = w0.W1 = w0.W2 = w0.W3 = w0.W4 = 0.0;
  if (act) {
    P_depth = wsd[(size_t)WS_PDEPTH * N]; h = -1.0 * P_depth;
    zeta[0] = wsd[(size_t)WS_ZETA0 * N]; zeta[1] = wsd[(size_t)WS_ZETA1 * N]; zeta[2] = wsd[(size_t)WS_ZETA2 * N];
    // getInterp weights of the start position in the rho element (setInterp :882)
    const double X = P.par[(size_t)pX * N + n], Y = P.par[(size_t)pY * N + n];
    rn = P.RE[P.P_r_element[n] - 1];
    Quad qq = load_quad_rho(P.rnode, rn);
    w0 = quad_weights(X, Y, qq.x, qq.y);
  }
  const SlabT* K = (const SlabT*)P.Kh;
  const int pslot[3] = {P.tb, P.tc, P.tf};
  const unsigned fsm = (P.FreeSlip && act) ? P.fsR[P.P_r_element[na] - 1] : 15u;  // free slip, :1934-1994
  // getInterp (hydrodynamic_module.f90:1997-2003) is v1+(v2-v1)t+(v3-v1)u in the first triangle and the same form on
  // the nodes (3,4,1) in the second (and on a node): the node ids of such a lane are rotated once, here, so that the
  // ~400 interpolations of the profile are one expression without a branch on the lane's regime.  A lane whose
  // element has a land node under free slip (the substitution works on the nodes in element order) or that takes
  // the inverse-distance weights keeps the order and the general form (`plain` false: rare).
  const bool plain = (fsm == 15u) && (w0.regime != 3);
  if (plain && (w0.regime == 2 || w0.regime == 4)) {
    const int tx = rn.x, ty = rn.y;
    rn.x = rn.z; rn.y = rn.w; rn.z = tx; rn.w = ty;
  }
  auto GI = [&](double f1, double f2, double f3, double f4) -> double {
    double r = f1 + (f2 - f1) * w0.t + (f3 - f1) * w0.u;
    if (!plain) {
      if (fsm != 15u) freeslip_sub(0, fsm, 15u, f1, f2, f3, f4);
      r = apply_getinterp(w0, f1, f2, f3, f4);
    }
    return r;
  };
  auto KH = [&](int s, int i) -> double {  // Pwc_KH{b,c,f}(i) :102-108
    const double f1 = fld(K, rn.x, ws, i, pslot[s]), f2 = fld(K, rn.y, ws, i, pslot[s]);
    const double f3 = fld(K, rn.z, ws, i, pslot[s]), f4 = fld(K, rn.w, ws, i, pslot[s]);
    return GI(f1, f2, f3, f4);
  };
  const Divisor dvh = prep_div((VT == 2) ? P.hc + h : h);  // every level depth divides by it
  auto WZ = [&](int s, int i) -> double {  // Pwc_wz{b,c,f}(i)
    return level_at(level_coef_d(wtab, i, h, P.hc, VT, dvh), zeta[s], h, VT);
  };
  double wz1[3], wzN[3], kh1[3], khN[3], rng_[3];
#pragma unroll
  for (int s = 0; s < 3; s++) {
    wz1[s] = WZ(s, 1); wzN[s] = WZ(s, ws);
    kh1[s] = KH(s, 1); khN[s] = KH(s, ws);
    rng_[s] = wzN[s] - wz1[s];
  }
  auto NEWX = [&](int s, int j) -> double {  // :121-123
    return wz1[s] + div_by(((double)(float)(j - 4)) * rng_[s], (double)p2, DivConst{P.c_p2[0], P.c_p2[1]}, FD);
  };
  // piecewise-linear profile segment per slot :126-163.  Moving up one level reuses
  // the upper end of the previous segment (same values), and the Kh of the level after
  // that was asked for one segment earlier, so its latency is off the critical path.
  int jlo[3] = {1, 1, 1};
  double slope[3], icpt[3], wzhi[3], khhi[3];
  SlabT rawn[3][4];
  // (32-bit byte offsets from the slab's base: lt_set_grid refuses a slab of 4 GiB)
  const unsigned esz = (unsigned)sizeof(SlabT), L3b = (unsigned)(ws * LT_SLOTS) * esz;
  const unsigned nb0 = (unsigned)(rn.x - 1) * L3b, nb1 = (unsigned)(rn.y - 1) * L3b, nb2 = (unsigned)(rn.z - 1) * L3b,
                 nb3 = (unsigned)(rn.w - 1) * L3b;
  const char* Kb = (const char*)K;
  auto RAW = [&](int s, int i) {
    const unsigned o = (unsigned)((i - 1) * LT_SLOTS + pslot[s]) * esz;
    rawn[s][0] = *(const SlabT*)(Kb + (nb0 + o)); rawn[s][1] = *(const SlabT*)(Kb + (nb1 + o));
    rawn[s][2] = *(const SlabT*)(Kb + (nb2 + o)); rawn[s][3] = *(const SlabT*)(Kb + (nb3 + o));
  };
  auto LINE = [&](int s, double a, double b, double ka, double kb) {
    slope[s] = (ka - kb) / (a - b);
    icpt[s] = ka - slope[s] * a;
    wzhi[s] = b;
    khhi[s] = kb;
  };
#pragma unroll
  for (int s = 0; s < 3; s++) {
    LINE(s, wz1[s], WZ(s, 2), kh1[s], KH(s, 2));
    RAW(s, 3);  // ws >= 6
  }
#ifdef LT_VT_ADV_STATS
  int adv_exec = 0, adv_lanes = 0;   // development aid: executions of the segment advance by this wave, lanes in them
#endif
  auto NEWY_AT = [&](int s, double xj) -> double {  // newy{b,c,f}(j) at its abscissa xj = newx(j), j = 5..p2+3
    while (!(wzhi[s] > xj) && jlo[s] < ws - 1) {
#ifdef LT_VT_ADV_STATS
      adv_exec += 1; adv_lanes += __popcll(__ballot(1));
#endif
      jlo[s] += 1;
      const double kb = GI((double)rawn[s][0], (double)rawn[s][1], (double)rawn[s][2], (double)rawn[s][3]);
      LINE(s, wzhi[s], WZ(s, jlo[s] + 1), khhi[s], kb);
      if (jlo[s] + 2 <= ws) RAW(s, jlo[s] + 2);
    }
    return slope[s] * xj + icpt[s];
  };
  // newy(1..8): the low pads (:170-173, all three use Pwc_KHb(1)) and the first 4 points.
  // The 8-point window lives in LDS as a ring: newy(j) of slot s at wring[s][j & 7][thread].
#pragma unroll
  for (int s = 0; s < 3; s++) {
#pragma unroll
    for (int j = 1; j <= 4; j++) wring[s][j & 7][threadIdx.x] = kh1[0];
#pragma unroll
    for (int j = 5; j <= 8; j++) wring[s][j & 7][threadIdx.x] = NEWY_AT(s, NEWX(s, j));
  }
  // table layout: [wave][knot][lane] records {x, y, yp}; knot k-2 is written at iteration k
  VtRec* tab = P.vt_tab + (size_t)(na >> 6) * (size_t)p2 * VT_TS + (size_t)(na & 63);
  // streaming YPC1 / SIGS state: the two previous knots, the two previous chord slopes, YP(k-2)
  double xm1 = 0.0, ym1 = 0.0, xm2 = 0.0, ym2 = 0.0, Sp = 0.0, DXp = 0.0, YPp = 0.0;
  bool ypc_fail = false;
  int ntask = 0;  // wave-uniform: intervals on this wave's task list
  const VtQueue Q = P.vt_q;
  const size_t qb = (size_t)wave * (size_t)Q.cap;
  // one interval of SIGS (tension_module.f90:448-776) as soon as its end derivatives exist: tension
  // zero, or onto the task list (the known-SBIG case goes through the list too: rare)
  // (the screening as lane masks, lt_math.h: LaneMask; every lane of the wave is here)
  auto SIGMA = [&](bool on, double S, double S1, double S2, int where) -> bool {
    const LaneMask m = lm(on) & ~sigs_eval_zero_mask(S, S1, S2);
    if (lm_lane(m)) {   // (the list has room for every interval of the wave: lt_set_particles)
      const int idx = ntask + __popcll(m & ((1ull << lane) - 1ull));
      Q.S[qb + idx] = S; Q.S1[qb + idx] = S1; Q.S2[qb + idx] = S2;
      Q.wl[qb + idx] = (unsigned short)((where << 6) | lane);
    }
    ntask += __popcll(m);
    if (lm_lane(m)) fbits[where >> 6][threadIdx.x] |= 1ull << (where & 63);   // (k_vt_sigs writes the factor to vt_sig)
    return lm_lane(m);
  };
  // newx(k+7) of the knot in hand is newx(k+8) of the one before: kept (the reference recomputes each newx
  // three times: as the end of the 8-point window, as its last-but-one point, and as its start)
  double x7[3] = {NEWX(0, 8), NEWX(1, 8), NEWX(2, 8)};
  // One knot.  MID = a knot in 4 .. p2-5 (all but seven of them): none of the end cases -- pinned end points,
  // YP(1) / YP(N), the last interval, the pads of the window -- applies, and the compiler drops their tests,
  // selects and the second screening from the loop that runs 120 times.
#ifdef LT_RK_TIMING
  // development aid (tools/dev/vt_sections.py): wave cycles per section of a knot, summed into g_tim[8..13]
  unsigned long long vt_acc[6] = {0, 0, 0, 0, 0, 0}, vt_prev = __builtin_readcyclecounter();
#define VTICK(sec) do { const unsigned long long t_ = __builtin_readcyclecounter(); vt_acc[sec] += t_ - vt_prev; vt_prev = __builtin_readcyclecounter(); } while (0)
#else
#define VTICK(sec) do { } while (0)
#endif
  auto knot = [&](const int k, auto mid_tag) {
    constexpr bool MID = decltype(mid_tag)::value;
    VTICK(5);
    double xk = 0.0, yk = 0.0;
    if (act) {
      double mx[3], my[3];
#pragma unroll
      for (int s = 0; s < 3; s++) {
        // :184-194 8-point average (formed at every k, selected below: no branch)
        const int t = threadIdx.x;
        const double av = (wring[s][k & 7][t] + wring[s][(k + 1) & 7][t] + wring[s][(k + 2) & 7][t] + wring[s][(k + 3) & 7][t] +
                           wring[s][(k + 4) & 7][t] + wring[s][(k + 5) & 7][t] + wring[s][(k + 6) & 7][t] + wring[s][(k + 7) & 7][t]) / 8.0;
        const double xa = NEWX(s, k);
        const double xm = xa + (x7[s] - xa) / 2.0;
        // :197-210 pinned ends
        mx[s] = MID ? xm : (k == 1) ? wz1[s] : (k == p2) ? wzN[s] : xm;
        my[s] = MID ? av : (k == 1) ? kh1[s] : (k == p2) ? khN[s] : av;
      }
      VTICK(0);
      // :220-275 interpolation to the internal times, clamp, Simpson in time
      // (selects on values, not on array elements: those would turn into stack indexing)
      const double mx0 = mx[0], mx1 = mx[1], mx2 = mx[2], my0 = my[0], my1 = my[1], my2 = my[2];
      double ex1 = mx0, ex2 = mx1, ex3 = mx2, ey1 = my0, ey2 = my1, ey3 = my2;
      if (P.p == 1) { ex2 = mx0; ex3 = mx1; ey2 = my0; ey3 = my1; }
      double fxb, fxc, fxf, fyb, fyc, fyf;
      polintd3x3<FD>(P, ex1, ex2, ex3, fxb, fxc, fxf);
      polintd3x3<FD>(P, ey1, ey2, ey3, fyb, fyc, fyf);
      if (fyb < 0.0) fyb = 0.0;
      if (fyc < 0.0) fyc = 0.0;
      if (fyf < 0.0) fyf = 0.0;
      yk = div_by(fyb + 4.0 * fyc + fyf, 6.0, DivConst{P.c_6[0], P.c_6[1]}, FD);
      xk = div_by(fxb + 4.0 * fxc + fxf, 6.0, DivConst{P.c_6[0], P.c_6[1]}, FD);
    }
    VTICK(1);
    // TSPSI(p2) as a stream: YPC1 (tension_module.f90:852-977), SIGS (:314-781).  YPC1 stops at the
    // first non-increasing abscissa (IER = -4 of TSPSI, SIGS never runs): the derivatives found so
    // far stay, the others and every tension factor are 0 (the walk ignores the stored tensions then).
    bool fail_now = false;
    double S = 0.0, DX = 0.0, YPc = 0.0, YPn = 0.0;
    const bool live = act && !ypc_fail;
    if (live && (MID || k >= 2)) {
      DX = xk - xm1;  // DX_{k-1}
      if (DX <= 0.0) {
        fail_now = true;
      } else {
        // (quotients through prepared reciprocals, sign cases as a flip / clamp / flip: lt_math.h, the forms of k_rk)
        S = div_prep(yk - ym1, prep_div(DX));  // S_{k-1}
        const Divisor dsum = prep_div(DXp + DX);
        if (!MID && k == 3) YPp = ypc1_end_b(Sp, Sp + div_prep(DXp * (Sp - S), dsum));  // YP(1)
        if (MID || k >= 3) YPc = ypc1_mid_b(DXp, DX, dsum, Sp, S);                      // YP(k-1)
        if (!MID && k == p2) YPn = ypc1_end_b(S, S + div_prep(DX * (S - Sp), dsum));     // YP(N)
      }
    }
    const bool ok = live && !fail_now;
    VTICK(2);
    // interval (k-2, k-1): chord slope S_{k-2} (still in Sp), derivatives YP(k-2), YP(k-1)
    (void)SIGMA(ok && (MID || k >= 3), Sp, YPp, YPc, k - 3);
    // (the last interval (N-1, N) with YP(N-1), YP(N))
    if (!MID) (void)SIGMA(ok && k == p2, S, YPc, YPn, p2 - 2);
    VTICK(3);
    if (act) {
      if (ok) {
        if (MID || k >= 3) tab[(size_t)(k - 3) * VT_TS] = VtRec{xm2, ym2, YPp};           // knot k-2, interval (k-2, k-1)
        if (!MID && k == p2) {
          tab[(size_t)(p2 - 2) * VT_TS] = VtRec{xm1, ym1, YPc};
          tab[(size_t)(p2 - 1) * VT_TS] = VtRec{xk, yk, YPn};
        }
      } else if (fail_now) {
        if (MID || k >= 3) tab[(size_t)(k - 3) * VT_TS] = VtRec{xm2, ym2, (MID || k >= 4) ? YPp : 0.0};
        if (MID || k >= 2) tab[(size_t)(k - 2) * VT_TS] = VtRec{xm1, ym1, 0.0};
        tab[(size_t)(k - 1) * VT_TS] = VtRec{xk, yk, 0.0};
        ypc_fail = true;
      } else {  // after a failure: abscissa and ordinate only
        tab[(size_t)(k - 1) * VT_TS] = VtRec{xk, yk, 0.0};
      }
      if (ok && (MID || k >= 2)) {
        if (MID || k >= 3) YPp = YPc;
        Sp = S; DXp = DX;
      }
      xm2 = xm1; ym2 = ym1;
      xm1 = xk; ym1 = yk;
      VTICK(4);
      // slide the 8-point window: drop newy(k), append newy(k+8)
      if (MID || k + 8 <= p2 + 7) {
#pragma unroll
        for (int s = 0; s < 3; s++) {
          const double x8 = NEWX(s, k + 8);
          wring[s][k & 7][threadIdx.x] = (!MID && k + 8 >= p2 + 4) ? khN[s] : NEWY_AT(s, x8);  // high pads
          x7[s] = x8;
        }
      }
    }
  };
#pragma unroll 1
  for (int k = 1; k <= 3; k++) knot(k, std::false_type{});
#pragma unroll 1
  for (int k = 4; k <= p2 - 5; k++) knot(k, std::true_type{});
#pragma unroll 1
  for (int k = p2 - 4; k <= p2; k++) knot(k, std::false_type{});
#ifdef LT_RK_TIMING
#ifndef LT_VT_EXCURSION
  if (lane == 0)
    for (int q = 0; q < 6; q++) atomicAdd(&g_tim[8 + q], vt_acc[q]);
#endif
#endif
#ifdef LT_VT_ADV_STATS
  if (lane == 0) { atomicAdd(&g_tim[6], (unsigned long long)adv_exec); atomicAdd(&g_tim[7], (unsigned long long)adv_lanes); atomicAdd(&g_tim[5], 1ull); }
#endif
  if ((size_t)wave * 64 < (size_t)P.numpar) {   // the lane's listed intervals (a YPC1 failure: the walks ignore them)
    unsigned long long* vb = P.vt_bits + (size_t)wave * 4 * 64 + (size_t)lane;
#pragma unroll
    for (int w = 0; w < 4; w++) vb[(size_t)w * 64] = fbits[w][threadIdx.x];
  }
  if (lane == 0 && (size_t)wave * 64 < (size_t)P.numpar) {   // (the last workgroup may hold waves beyond the particles)
    Q.count[wave] = (ntask < Q.cap) ? ntask : Q.cap;
    if (ntask) atomicAdd(&P.counters[CNT_VT_QUEUED], (unsigned long long)ntask);
  }
  if (act) {
    int f2 = fl & ~(FL_VT_YPCFAIL | FL_VT_SIGERR);
    if (ypc_fail) f2 |= FL_VT_YPCFAIL;
    if (f2 != fl) P.ws_i[n] = f2;
  }
}
#endif  // LT_VT_DEFINE_A

// The SIGS iterations (tension_module.f90:531-760) of the listed intervals, one wave per list, 64 at a
// time; the tension factor goes to vt_sig at the knot's index, a failed convexity iteration (SigErr) into the
// particle's flag word.
#ifdef LT_VT_DEFINE_B
__global__ __launch_bounds__(256) void k_vt_sigs(const StepParams P) {
  exp_table_load();
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = gid >> 6;
  if ((size_t)wave * 64 >= (size_t)P.numpar) return;
  const VtQueue Q = P.vt_q;
  const int count = Q.count[wave], p2 = 4 * P.ws;
  const size_t qb = (size_t)wave * (size_t)Q.cap;
  unsigned rounds = 0;
  // (the next round's task is requested before the current one is iterated: a round is a load's round trip and then
  // a chain of dependent operations, and with one after the other the wave sat through both)
  double nS = 0.0, nS1 = 0.0, nS2 = 0.0;
  unsigned nwl = 0;
  {
    const size_t ti = qb + ((lane < count) ? lane : 0);
    nS = Q.S[ti]; nS1 = Q.S1[ti]; nS2 = Q.S2[ti]; nwl = Q.wl[ti];
  }
#pragma unroll 1
  for (int t = lane; t - lane < count; t += 64) {
    rounds += 1;
    const bool have = t < count;
    const double S = nS, S1 = nS1, S2 = nS2;
    const unsigned wl = nwl;
    if (t - lane + 64 < count) {
      const size_t tn = qb + ((t + 64 < count) ? t + 64 : 0);
      nS = Q.S[tn]; nS1 = Q.S1[tn]; nS2 = Q.S2[tn]; nwl = Q.wl[tn];
    }
    double sg = 0.0;
    double T = 0.0, sgn;
    const int kind = have ? sigs_classify(S, S1, S2, sg, T) : SIGK_DONE;
    const bool need = kind == SIGK_NEWTON;
    const LaneMask NEED = lm(need);
    LaneMask FAILED;
    sigs_newton_w(T, NEED, sgn, FAILED, exp_acc());      // the whole wave, lanes without a Newton task idle in it
    if (need) sg = sgn;
    else if (kind == SIGK_SECANT) sg = sigs_secant(S, S1, S2);
    if (have) {
      P.vt_sig[((size_t)wave * (size_t)p2 + (wl >> 6)) * VT_TS + (wl & 63u)] = sg;
      if (lm_lane(NEED & FAILED)) {
        // SigErr: every evaluation of this particle's walk is linint's (ver_turb_module.f90:296-331).  k_vt_walk_fast
        // may be running beside this kernel and may not see the flag: the particle goes onto k_vt_walk's list from
        // here (once: the first failed interval of the particle), marked as such, and is walked from the first
        // sub-step whatever the fast kernel did with it.
        const int pn = (int)((size_t)wave * 64 + (wl & 63u));
        const int old = atomicOr(&P.ws_i[pn], (int)FL_VT_SIGERR);
        if (!(old & FL_VT_SIGERR)) P.vt_wl_list[atomicAdd(P.vt_wl_count, 1)] = pn | (int)0x80000000;
      }
    }
  }
  if (lane == 0 && rounds) atomicAdd(&P.counters[CNT_VT_ROUNDS], (unsigned long long)rounds);
}
#endif  // LT_VT_DEFINE_B

// rows of deviates per group of 64 particles: two for HTurb's X and Y (hor_turb_module.f90:29-50; formed here when
// both kinds of turbulence are on, read by k_post), then the idt/2 of the walk
#define VT_DEV_HROWS 2
#define VT_DEV_BASE(P, n, loop) ((P).vt_dev + (size_t)((n) >> 6) * (size_t)((loop) + VT_DEV_HROWS) * 64 + (size_t)((n) & 63))
struct __attribute__((packed, aligned(4))) RawWords4 { uint32_t a, b, c, d; };
// ---- the normal deviates of the walk
// norm() (norm_module.f90:25-39) of a sub-step is a function of the particle's two words of the stream alone, not
// of where the walk has got to: the idt/2 deviates of every tracked particle are formed here, ahead of the walk and
// out of its chain of dependent operations (a sub-step of the walk is a serial chain HPVAL -> HVAL -> displacement;
// log, cos and sqrt -- the host-libm-exact forms, 240 of its 386 instructions -- sat in that chain although they
// depend on nothing in it).  One lane per particle, a line of raw words (8 sub-steps) at a time as the walk took
// them; the deviates go to vt_dev[group of 64 particles][sub-step][lane], so that a wave of the walk reads one
// 512-byte row per sub-step.  Same functions, same operands, same product: the walk's sums are unchanged.
#ifdef LT_VT_DEFINE_A
__global__ __launch_bounds__(256) void k_vt_norm(const StepParams P) {
  __shared__ uint2 rbuf[8][256];
  libm_tables_load<LM_LOG | LM_SINCOS>();
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= P.numpar) return;
  if (!(P.ws_i[n] & FL_ACT)) return;
  const int loop = P.idt / 2;
  unsigned long long di = P.rng_base + (unsigned long long)(unsigned)P.rank[P.orig[n]];
  double* out = VT_DEV_BASE(P, n, loop);
  if (P.HTurbOn) {   // HTurb's two deviates (its four words come first, LTRANS.f90:1087,1098): norm() as below
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const double dev1 = ((double)ltmt_temper(P.rng_ring[(di + 2 * j) & P.rng_mask]) + 0.5) / 4294967296.0;
      const double dev2 = ((double)ltmt_temper(P.rng_ring[(di + 2 * j + 1) & P.rng_mask]) + 0.5) / 4294967296.0;
      out[(size_t)j * 64] = sqrt(-2.0 * lt_log(dev1)) * lt_cos(2.0 * P.PI * dev2);
    }
    di += 4ull;
  }
  out += (size_t)VT_DEV_HROWS * 64;
#pragma unroll 1
  for (int i0 = 0; i0 < loop; i0 += 8) {
    const unsigned long long o = di & P.rng_mask;
    if (o + 16 <= P.rng_mask + 1ull) {
      const RawWords4* src = (const RawWords4*)(P.rng_ring + o);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const RawWords4 w = src[j];
        rbuf[2 * j][threadIdx.x] = make_uint2(w.a, w.b);
        rbuf[2 * j + 1][threadIdx.x] = make_uint2(w.c, w.d);
      }
    } else {  // the window wraps around the end of the ring
#pragma unroll 1
      for (int j = 0; j < 8; j++)
        rbuf[j][threadIdx.x] = make_uint2(P.rng_ring[(di + 2 * j) & P.rng_mask],
                                          P.rng_ring[(di + 2 * j + 1) & P.rng_mask]);
    }
    di += 16;
    const int m = (loop - i0 < 8) ? (loop - i0) : 8;
#pragma unroll 2
    for (int j = 0; j < m; j++) {
      const uint2 rw = rbuf[j][threadIdx.x];
      const double dev1 = ((double)ltmt_temper(rw.x) + 0.5) / 4294967296.0;
      const double dev2 = ((double)ltmt_temper(rw.y) + 0.5) / 4294967296.0;
      out[(size_t)(i0 + j) * 64] = sqrt(-2.0 * lt_log(dev1)) * lt_cos(2.0 * P.PI * dev2);
    }
  }
}
#endif  // LT_VT_DEFINE_A

// ---- the walk, in two kernels
// 80 % of the walks never leave the zero-tension intervals (~9 % of the intervals carry a tension factor and
// a walk stays within a knot or two of its start), but a wave of 64 walks meets a tension factor at nearly
// every evaluation, and the exponential / rational forms of HVAL and HPVAL -- with SNHCSH -- are 60 % of the
// instructions of a sub-step executed for five or six lanes.  k_vt_walk_fast therefore knows the cubic form
// only: a walk that enters an interval with a tension factor (or whose table is not a proper spline: SigErr,
// YPC1's failure) stops at the beginning of that sub-step and is listed; k_vt_walk takes the listed walks,
// one lane each, from the sub-step they stopped at, with every form.  Same operations per walk either way.
#ifdef LT_VT_EXCURSION
#define EX_NOTE() do { if (ci.I < ex_lo) ex_lo = ci.I; if (ci.I > ex_hi) ex_hi = ci.I; } while (0)
#else
#define EX_NOTE() do { } while (0)
#endif
struct VtLite {          // the cached interval without the tension-dependent parts
  int I;
  double X1, X2, Y1, Y2, P1, P2;
  double S, D1, D2;
  Divisor dx;
};
LT_DEV void vt_seek_lite(VtLite& c, double T, int n, const VtRec* tab) {
  bool moved = false;
  if (c.I == 0) {
    // first use.  The table is strictly increasing here (k_vt_walk_fast does not take a table YPC1 rejected), so
    // the interval INTRVL's bisection finds is the unique one with X(I) <= T < X(I+1) (clamped to 1 .. N-1), and
    // the stepping below reaches it from any start: start where T lies between the end knots in proportion --
    // the knots are nearly evenly spaced -- two or three loads instead of a chain of seven.
    const double xa = tab[0].x, xb = tab[(size_t)(n - 1) * VT_TS].x;
    const double f = (T - xa) / (xb - xa) * (double)(n - 1);
    int lo = (f >= 0.0 && f < (double)(n - 1)) ? (int)f + 1 : ((f >= 0.0) ? n - 1 : 1);   // (NaN: 1)
    if (lo < 1) lo = 1;
    if (lo > n - 1) lo = n - 1;
    c.I = lo;
    const VtRec ra = tab[(size_t)(lo - 1) * VT_TS], rb = tab[(size_t)lo * VT_TS];
    c.X1 = ra.x; c.Y1 = ra.y; c.P1 = ra.yp;
    c.X2 = rb.x; c.Y2 = rb.y; c.P2 = rb.yp;
    moved = true;
  }
  while (T < c.X1 && c.I > 1) {
    c.I -= 1;
    const VtRec r = tab[(size_t)(c.I - 1) * VT_TS];
    c.X2 = c.X1; c.Y2 = c.Y1; c.P2 = c.P1;
    c.X1 = r.x; c.Y1 = r.y; c.P1 = r.yp;
    moved = true;
  }
  while (T >= c.X2 && c.I < n - 1) {
    c.I += 1;
    const VtRec r = tab[(size_t)c.I * VT_TS];
    c.X1 = c.X2; c.Y1 = c.Y2; c.P1 = c.P2;
    c.X2 = r.x; c.Y2 = r.y; c.P2 = r.yp;
    moved = true;
  }
  if (moved) {  // vt_prepare without SNHCSH
    const double DX = c.X2 - c.X1;
    c.dx = prep_div(DX);
    c.S = LT_VT_SLOPE(c.Y2 - c.Y1, DX, c.dx);
    c.D1 = c.S - c.P1;
    c.D2 = c.P2 - c.S;
  }
}
#ifndef LT_VT_WALKFAST_BLOCK
#define LT_VT_WALKFAST_BLOCK 256
#endif
#ifndef LT_VT_WALK_BLOCK
#define LT_VT_WALK_BLOCK 256
#endif
#ifdef LT_VT_DEFINE_A
__global__ __launch_bounds__(LT_VT_WALKFAST_BLOCK) void k_vt_walk_fast(const StepParams P) {
  libm_tables_load<LM_EXP | LM_POW>();  // (pow takes the exp table too)
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= P.numpar) return;
  const int fl = P.ws_i[n];
  if (!(fl & FL_ACT)) return;
  const int lane = threadIdx.x & 63;
  const size_t N = (size_t)P.numpar;
  double* wsd = P.ws_d + n;
  const int p2 = 4 * P.ws;
  const VtRec* tab = P.vt_tab + (size_t)(n >> 6) * (size_t)p2 * VT_TS + (size_t)(n & 63);
  const VtBits bits = vt_bits_load(P, n);   // the listed intervals of this particle: a walk stops there
  const double P_depth = wsd[(size_t)WS_PDEPTH * N], P_zc = wsd[(size_t)WS_PZ1 * N];
  const double P_zetac = wsd[(size_t)WS_ZETA1 * N];
  const double background = (double)1.0E-6f, deltat = 2.0;
  const int loop = P.idt / (int)deltat;
  const double* dev = VT_DEV_BASE(P, n, loop) + (size_t)VT_DEV_HROWS * 64;
  double ParZc = P_zc;
  VtLite ci;
  ci.I = 0; ci.X1 = ci.X2 = ci.Y1 = ci.Y2 = ci.P1 = ci.P2 = 0.0;
  ci.S = ci.D1 = ci.D2 = 0.0; ci.dx.b = 1.0; ci.dx.y = 1.0;
  // (a table that is not a proper spline is walked by k_vt_walk from the first sub-step)
  bool stopped = (fl & (FL_VT_YPCFAIL | FL_VT_SIGERR)) != 0;
  int stop_i = 1;
#ifdef LT_VT_EXCURSION
  int ex_lo = 1 << 30, ex_hi = 0;   // development aid (tools/dev/vt_excursion.py): the intervals this walk visits
#endif
#pragma unroll 1
  for (int i = 1; i <= loop; i++) {
    if (stopped) break;
    const double DEV = dev[(size_t)(i - 1) * 64];   // norm() of this sub-step (k_vt_norm)
    double Kprimec = 0.0;
    bool tens = false;
    if (!(ParZc < P_depth || ParZc > P_zetac)) {
      vt_seek_lite(ci, ParZc, p2, tab);
      EX_NOTE();
      tens = vt_bit(bits, ci.I - 1);
      const double B1 = div_prep(ci.X2 - ParZc, ci.dx), B2 = 1.0 - B1;  // HPVAL's cubic form, :1222
      Kprimec = ci.P1 + B2 * (ci.D1 + ci.D2 - 3.0 * B1 * (ci.D2 - ci.D1));
    }
    const double KprimeZc = -1.0 * Kprimec * deltat;
    const double Z3rdc = ParZc + 0.5 * KprimeZc;
    double KH3rdc = background;
    if (!tens && !(Z3rdc < P_depth || Z3rdc > P_zetac)) {
      vt_seek_lite(ci, Z3rdc, p2, tab);
      EX_NOTE();
      tens = vt_bit(bits, ci.I - 1);
      const double U = Z3rdc - ci.X1;                                   // HVAL's cubic form, :1083
      const double B2 = div_prep(U, ci.dx), B1 = 1.0 - B2;
      KH3rdc = ci.Y1 + U * (ci.P1 + B2 * (ci.D1 + B1 * (ci.D1 - ci.D2)));
      if (KH3rdc < background) KH3rdc = background;
    }
    if (tens) {  // this sub-step is k_vt_walk's
      stopped = true;
      stop_i = i;
      break;
    }
    // (2/r * KH3rd * deltat)**0.5 with r = 1
    ParZc = ParZc + KprimeZc + DEV * lt_pow_half(2.0 / 1.0 * KH3rdc * deltat);
  }
  if (!stopped) {
    wsd[(size_t)WS_TURBV * N] = P_zc - ParZc;  // :342
  } else {
    wsd[(size_t)WS_TURBV * N] = ParZc;         // the depth at the beginning of sub-step stop_i
    P.vt_wl_resume[n] = stop_i;
  }
#ifdef LT_VT_EXCURSION
  if (ex_hi >= ex_lo) {
    // how far from the interval of its start depth (estimated as vt_seek_lite's first use does) the walk went
    const double xa = tab[0].x, xb = tab[(size_t)(p2 - 1) * VT_TS].x;
    int k0 = (int)((P_zc - xa) / (xb - xa) * (double)(p2 - 1)) + 1;
    if (k0 < 1) k0 = 1;
    if (k0 > p2 - 1) k0 = p2 - 1;
    int ex = max(ex_hi - k0, k0 - ex_lo);
    if (ex < 0) ex = 0;
    atomicAdd(&g_tim[ex < 15 ? ex : 15], 1ull);
  }
#endif
  // the list of the stopped walks (one atomic per wave)
  const unsigned long long m = __ballot(stopped);
  if (m) {
    const int leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(P.vt_wl_count, __popcll(m));
    base = __shfl(base, leader, 64);
    if (stopped) P.vt_wl_list[base + __popcll(m & ((1ull << lane) - 1ull))] = n;
  }
}
#endif  // LT_VT_DEFINE_A

// :282-335 random displacement model, idt/2 sub-steps of 2 s: the listed walks, every form of HVAL / HPVAL
#ifdef LT_VT_DEFINE_B
__global__ __launch_bounds__(LT_VT_WALK_BLOCK) void k_vt_walk(const StepParams P) {
  const int nlisted = *P.vt_wl_count;
  if ((int)(blockIdx.x * blockDim.x) >= nlisted) return;   // (the whole workgroup: launched for the worst case)
  libm_tables_load<LM_EXP | LM_POW>();
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nlisted) return;
  // an entry is a particle slot; bit 31: listed by k_vt_sigs (SigErr).  A SigErr particle is walked through ITS entry
  // of k_vt_sigs, from the first sub-step; an entry k_vt_walk_fast made for it (a stop, or the flag seen at its start)
  // is skipped.
  const int ent = P.vt_wl_list[t];
  const int n = ent & 0x7fffffff;
  const int fl = P.ws_i[n];
  const bool ypc_fail = (fl & FL_VT_YPCFAIL) != 0;
  const int sigerr = (fl & FL_VT_SIGERR) ? 1 : 0;
  if (sigerr && ent >= 0) return;
  const int i0 = sigerr ? 1 : P.vt_wl_resume[n];     // the sub-step the walk stopped at (1: nothing done yet)
  const size_t N = (size_t)P.numpar;
  double* wsd = P.ws_d + n;
  const int p2 = 4 * P.ws;
  const VtRec* tab = P.vt_tab + (size_t)(n >> 6) * (size_t)p2 * VT_TS + (size_t)(n & 63);
  const double* sig = P.vt_sig + (size_t)(n >> 6) * (size_t)p2 * VT_TS + (size_t)(n & 63);
  const VtBits bits = vt_bits_load(P, n);
  const double P_depth = wsd[(size_t)WS_PDEPTH * N], P_zc = wsd[(size_t)WS_PZ1 * N];
  const double P_zetac = wsd[(size_t)WS_ZETA1 * N];
  // :282-335 random displacement model, idt/2 sub-steps of 2 s
  const double background = (double)1.0E-6f, deltat = 2.0;
  const int loop = P.idt / (int)deltat;
  const double* dev = VT_DEV_BASE(P, n, loop) + (size_t)VT_DEV_HROWS * 64;   // norm() per sub-step
  double ParZc = sigerr ? P_zc : wsd[(size_t)WS_TURBV * N];   // left there by k_vt_walk_fast
  int il = 1;
  VtInterval ci;
  ci.I = 0; ci.X1 = ci.X2 = ci.Y1 = ci.Y2 = ci.P1 = ci.P2 = ci.SG = 0.0;
  ci.S = ci.D1 = ci.D2 = ci.SM = ci.CM = ci.CMM = 0.0; ci.dx.b = 1.0; ci.dx.y = 1.0;
  const bool cached = (sigerr == 0) && !ypc_fail;
#ifdef LT_VT_EXCURSION
  int ex_lo = 1 << 30, ex_hi = 0;
#endif
#ifdef LT_RK_TIMING
  // development aid (tools/dev/vt_walk_stats.py): what the waves of this kernel spend their sub-steps on
  unsigned long long st_t0 = __builtin_readcyclecounter();
  int st_iter = 0, st_lane_iter = 0, st_lane_T = 0, st_anyT = 0, st_anyR = 0, st_anyE = 0;
#endif
#pragma unroll 1
  for (int i = i0; i <= loop; i++) {
    const double DEV = dev[(size_t)(i - 1) * 64];
    double Kprimec = 0.0;
#ifdef LT_RK_TIMING
    bool st_T = false, st_R = false, st_E = false;
#define ST_NOTE() do { const double a_ = fabs(ci.SG); if (a_ >= 1.e-9) { st_T = true; if (a_ <= .5) st_R = true; else st_E = true; } } while (0)
#else
#define ST_NOTE() do { } while (0)
#endif
    if (!(ParZc < P_depth || ParZc > P_zetac)) {
      if (cached) {
        vt_seek(ci, ParZc, p2, tab, sig, bits);
        EX_NOTE();
        ST_NOTE();
        Kprimec = hpval_prepared(ParZc, ci);
      } else if (sigerr == 0) {
        Kprimec = hval_table<true>(ParZc, p2, tab, il);      // (ypc_fail: SIGS never ran)
      } else if (!ypc_fail) {
        vt_seek_plain(ci, ParZc, p2, tab);
        double yy;
        linint_interval(ci, ParZc, yy, Kprimec);
      } else { double yy; linint_n(&tab[0].x, &tab[0].y, p2, 3 * VT_TS, ParZc, yy, Kprimec); }
    }
    const double KprimeZc = -1.0 * Kprimec * deltat;
    const double Z3rdc = ParZc + 0.5 * KprimeZc;
    double KH3rdc;
    if (Z3rdc < P_depth || Z3rdc > P_zetac) {
      KH3rdc = background;
    } else {
      if (cached) {
        vt_seek(ci, Z3rdc, p2, tab, sig, bits);
        EX_NOTE();
        ST_NOTE();
        KH3rdc = hval_prepared(Z3rdc, ci);
      } else if (sigerr == 0) {
        KH3rdc = hval_table<false>(Z3rdc, p2, tab, il);
      } else if (!ypc_fail) {
        vt_seek_plain(ci, Z3rdc, p2, tab);
        double mm;
        linint_interval(ci, Z3rdc, KH3rdc, mm);
      } else { double mm; linint_n(&tab[0].x, &tab[0].y, p2, 3 * VT_TS, Z3rdc, KH3rdc, mm); }
      if (KH3rdc < background) KH3rdc = background;
    }
    // (2/r * KH3rd * deltat)**0.5 with r = 1
    ParZc = ParZc + KprimeZc + DEV * lt_pow_half(2.0 / 1.0 * KH3rdc * deltat);
#ifdef LT_RK_TIMING
    st_iter += 1; st_lane_iter += 1; st_lane_T += st_T ? 1 : 0;
    st_anyT += __ballot(st_T) ? 1 : 0; st_anyR += __ballot(st_R) ? 1 : 0; st_anyE += __ballot(st_E) ? 1 : 0;
#endif
  }
#ifdef LT_RK_TIMING
  {
    // per wave: iterations of the longest lane, lane-iterations, lane-iterations on a tension interval, iterations
    // in which any lane was on one (any / rational form / exponential form), the most such sub-steps of one lane
    int mx_iter = st_iter, mx_T = st_lane_T, s_li = st_lane_iter, s_lt = st_lane_T, aT = st_anyT, aR = st_anyR, aE = st_anyE;
    for (int o = 32; o; o >>= 1) {
      mx_iter = max(mx_iter, __shfl_xor(mx_iter, o, 64)); mx_T = max(mx_T, __shfl_xor(mx_T, o, 64));
      s_li += __shfl_xor(s_li, o, 64); s_lt += __shfl_xor(s_lt, o, 64);
      aT = max(aT, __shfl_xor(aT, o, 64)); aR = max(aR, __shfl_xor(aR, o, 64)); aE = max(aE, __shfl_xor(aE, o, 64));
    }
#ifndef LT_VT_EXCURSION
    if ((threadIdx.x & 63) == (__ffsll((long long)__ballot(1)) - 1)) {
      atomicAdd(&g_tim[0], 1ull); atomicAdd(&g_tim[1], (unsigned long long)mx_iter);
      atomicAdd(&g_tim[2], (unsigned long long)s_li); atomicAdd(&g_tim[3], (unsigned long long)s_lt);
#ifndef LT_VT_ADV_STATS
      atomicAdd(&g_tim[4], (unsigned long long)aT); atomicAdd(&g_tim[5], (unsigned long long)aR);
      atomicAdd(&g_tim[6], (unsigned long long)aE); atomicAdd(&g_tim[7], (unsigned long long)mx_T);
#endif
      atomicAdd(&g_tim[14], __builtin_readcyclecounter() - st_t0);
      atomicMax(&g_tim[15], __builtin_readcyclecounter() - st_t0);
    }
#endif
  }
#endif
#ifdef LT_VT_EXCURSION
  if (ex_hi >= ex_lo) {
    const double xa = tab[0].x, xb = tab[(size_t)(p2 - 1) * VT_TS].x;
    int k0 = (int)((P_zc - xa) / (xb - xa) * (double)(p2 - 1)) + 1;
    if (k0 < 1) k0 = 1;
    if (k0 > p2 - 1) k0 = p2 - 1;
    int ex = max(ex_hi - k0, k0 - ex_lo);
    if (ex < 0) ex = 0;
    atomicAdd(&g_tim[ex < 15 ? ex : 15], 1ull);
  }
#endif
  wsd[(size_t)WS_TURBV * N] = P_zc - ParZc;  // :342
  if (sigerr != 0) atomicAdd(&P.counters[CNT_SIGERR], 1ull);
}
#endif  // LT_VT_DEFINE_B
