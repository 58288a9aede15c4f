// lt_rk.h -- field access, the wave-cooperative find_currents (pass A / B / C) and the kernels
// k_pre, k_beh_pre, k_rk, k_rk_slow.
// Part of lt_kernels.h (included inside namespace ltd); see the notes there.
#pragma once

// ---------------------------------------------------- field access
// log10 of the log layer at the bottom (hydrodynamic_module.f90: the water-column profile below the
// deepest level): rare, and out of line so that its registers are not k_rk's
LT_DEV_NOINLINE double log10_call(double x) { return lt_log10_g(x); }

template <typename SlabT>
LT_DEV double fld(const SlabT* F, int node, int L, int k, int ps) {
  return (double)F[((size_t)(node - 1) * L + (k - 1)) * LT_SLOTS + ps];
}
// the velocity fields U, V, W: [node][slot][level] (lt_device.h)
template <typename SlabT>
LT_DEV double fldv(const SlabT* F, int node, int L, int k, int ps) {
  return (double)F[((size_t)(node - 1) * LT_SLOTS + ps) * L + (k - 1)];
}
// four consecutive levels of one (node, slot) of a velocity field: one 16-byte access for
// fp32 storage (aligned to the element only: a window starts at any level)
template <typename SlabT> struct Win4;
template <> struct __attribute__((packed, aligned(4))) Win4<float> { float v[4]; };
template <> struct __attribute__((packed, aligned(8))) Win4<double> { double v[4]; };

// what a stage of k_rk needs first and what the step to the next stage needs: requested before the
// last round of pass B of the stage before, so that the round covers their latency
struct StageIn { double zb, zc, zf; int4 nd0, nd1; int ii, iii; };   // zetas, node ids of the u and v elements, last windows
struct StageTail { double X, Y, Zpar, cosA, sinA, minpd, maxpd; };
// the per-particle context that find_currents needs (registers)
struct Column {
  double zeta[3];   // zetab, zetac, zetaf at the particle (roles b,c,f)
  double h;         // -P_depth
  double Pz[3];     // P_zb, P_zc, P_zf
  double wz1[3];    // Pwc_wz{b,c,f}(1)
  double z1[3];     // Pwc_z{b,c,f}(1)
};

// s-level depth split into its zeta-independent part (one division per level
// instead of one per level and slot): the operations and their order are those
// of getSlevel/getWlevel (hydrodynamic_module.f90:2691-2777).
struct LevCoef { double a, b, c; };
LT_DEV LevCoef level_coef(const double* tab, int k, double h, double hc, int Vtransform) {
  LevCoef r;
  double A = tab[k - 1];
  double B = tab[LT_LEV_STRIDE + k - 1];
  if (Vtransform == 1) {
    double S = A + (h - hc) * B;
    r.a = S; r.b = 1.0 + S / h; r.c = 0.0;
  } else if (Vtransform == 2) {
    r.a = (A + h * B) / (hc + h); r.b = 0.0; r.c = 0.0;
  } else {
    r.a = A; r.b = 1.0 + tab[2 * LT_LEV_STRIDE + k - 1]; r.c = (h - hc) * B;
  }
  return r;
}
// the same with the per-particle divisor prepared once: dv = h (Vtransform 1) or hc + h (2)
LT_DEV LevCoef level_coef_d(const double* tab, int k, double h, double hc, int Vtransform, const Divisor& dv) {
  LevCoef r;
  double A = tab[k - 1];
  double B = tab[LT_LEV_STRIDE + k - 1];
  if (Vtransform == 1) {
    double S = A + (h - hc) * B;
    r.a = S; r.b = 1.0 + div_prep(S, dv); r.c = 0.0;
  } else if (Vtransform == 2) {
    r.a = div_prep(A + h * B, dv); r.b = 0.0; r.c = 0.0;
  } else {
    r.a = A; r.b = 1.0 + tab[2 * LT_LEV_STRIDE + k - 1]; r.c = (h - hc) * B;
  }
  return r;
}
LT_DEV double level_at(const LevCoef& q, double zeta, double h, int Vtransform) {
  if (Vtransform == 1) return q.a + zeta * q.b;
  if (Vtransform == 2) return zeta + (zeta + h) * q.a;
  return zeta * q.b + q.a + q.c;
}


// per-wave LDS scratch of the batched passes.  Two task kinds:
//   check: a non-evaluation interval that may need the convexity Newton
//          iteration; only its convergence matters (SigErr): (num, den) of T;
//   eval:  the evaluation interval's tension is not known without iterating:
//          the task carries the interval, the lane that iterates evaluates HVAL.
// Results of all 9 (variable, slot) fits of a stage are collected in res[][lane].
#ifdef LT_RK_MARK
#define TICKB(slot) asm volatile("; MARK " #slot)
#else
#define TICKB(slot) do { } while (0)
#endif
#ifdef LT_RK_WAVETIME
// development aid (tools/dev/rk_wavetime.py): start / end of every wave on the 100 MHz clock, its shader cycles
__device__ unsigned long long g_wavetime[40000];
__device__ unsigned long long g_wavecyc[20000];
#endif
#ifdef LT_RK_TIMING
// development aid: wave time per section, accumulated in the wave's LDS scratch and
// added to g_tim once at the end of the kernel
__device__ unsigned long long g_tim[16];
#define TICK(slot) do { const unsigned long long t_ = __builtin_readcyclecounter(); if ((threadIdx.x & 63) == 0) ws.tim[slot] += t_ - tprev_; tprev_ = __builtin_readcyclecounter(); } while (0)
#define TICK_INIT unsigned long long tprev_ = __builtin_readcyclecounter()
#else
#ifdef LT_RK_MARK
#define TICK(slot) asm volatile("; MARK " #slot)
#else
#define TICK(slot) do { } while (0)
#endif
#define TICK_INIT do { } while (0)
#endif
// The nine (variable, slot) fits of a STAGE share the queues: whenever 64 evaluation intervals
// are waiting (and the next slot might not fit) one full round of pass B takes the last 64;
// what is left at the end of the stage goes with the check intervals into the last round(s).
// ~23 % of the fits queue an evaluation interval, i.e. ~147 per stage and wave: two full
// rounds and a partial one, where a pass B per variable ran 4 rounds at 57 % of the lanes.
// A check interval that finds its queue full (~1 % of them are candidates: 15 per stage and
// wave; the queue is emptied with every full round and whenever it is half full) sends its
// particle-step to k_rk_slow.
#ifndef LT_CHK_CAP
#define LT_CHK_CAP 128
#endif
#ifndef LT_EV_CAP
#define LT_EV_CAP 128
#endif
struct WaveScratch {
  double chk[LT_CHK_CAP][2];         // num, den of T = MAX(D1/D2, D2/D1)
  unsigned short chk_owner[LT_CHK_CAP];  // lane | slot9 << 6
  double ev[LT_EV_CAP][7];           // x1,x2,y1,p1,p2,S,Pz
  unsigned short ev_owner[LT_EV_CAP];
  double res[9][64];                 // H(P_z) per (variable * 3 + slot, lane) of the current stage
  unsigned int errmask[64];          // bit (variable*3+slot): SigErr / degenerate window
#ifdef LT_RK_TIMING
  unsigned long long tim[16];
#endif
};

// pass B: the queued iterations, 64 at a time, all lanes of the wave.  Evaluation
// intervals and check intervals share the batches, so the wave runs ONE Newton loop per
// batch whatever the mix (a separate loop for the few check intervals cost as much as the
// one for the evaluation intervals).
template <int K>
LT_DEV void run_task_batch(WaveScratch& ws, int nev, int total, int t0, const ExpAcc& ea) {
  bool is_ev[K], valid[K], need[K], failed[K];
  int tq[K], kind[K];
  unsigned short o[K];
  double sg[K], T[K];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const int t = t0 + 64 * j;
    valid[j] = t < total;
    is_ev[j] = t < nev;
    tq[j] = is_ev[j] ? t : (valid[j] ? t - nev : 0);
    o[j] = is_ev[j] ? ws.ev_owner[tq[j]] : ws.chk_owner[tq[j]];
    sg[j] = 0.0;
    kind[j] = SIGK_DONE;
    // the quotient T = MAX(D1/D2, D2/D1) of an evaluation interval (full SIGS decision) and of a check
    // interval (only the convergence of its Newton iteration matters): one division for both kinds
    double num = 0.0, den = 1.0;
    bool want_T = false;
    if (is_ev[j]) {
      const double* e = ws.ev[tq[j]];
      kind[j] = sigs_classify_nd(e[5], e[3], e[4], sg[j], num, den, want_T);
    } else if (valid[j]) {
      num = ws.chk[tq[j]][0]; den = ws.chk[tq[j]][1];
      want_T = true;
    }
    if (!want_T) { num = 0.0; den = 1.0; }
    T[j] = num / den;
    // (SIGS' own test is T <= 2 -> done, the check interval's was queued as T > 2 -> iterate: a NaN differs)
    if (want_T) kind[j] = (is_ev[j] ? !(T[j] <= 2.0) : (T[j] > 2.0)) ? SIGK_NEWTON : SIGK_DONE;
    need[j] = kind[j] == SIGK_NEWTON;
  }
  TICKB(20);
  if (K == 1) {
    // one task per lane: the wave version with the lane conditions as lane masks (lt_math.h)
    double sgn;
    LaneMask FAILED;
    const LaneMask NEED = lm(need[0]);
    sigs_newton_w(T[0], NEED, sgn, FAILED, ea);
    TICKB(21);
    if (need[0]) sg[0] = sgn;
    else if (kind[0] == SIGK_SECANT) sg[0] = sigs_secant(ws.ev[tq[0]][5], ws.ev[tq[0]][3], ws.ev[tq[0]][4]);  // rare
    if (is_ev[0]) {
      const double* e = ws.ev[tq[0]];
      ws.res[o[0] >> 6][o[0] & 63] = hval_interval_s(e[6], e[0], e[1], e[2], e[3], e[4], e[5], sg[0], &ea);
    }
    if (lm_lane(NEED & FAILED)) atomicOr(&ws.errmask[o[0] & 63], 1u << (o[0] >> 6));
  } else {
    double sgn[K];
    sigs_newton_k<K>(T, need, sgn, failed);
    TICKB(21);
#pragma unroll
    for (int j = 0; j < K; ++j) {
      if (need[j]) sg[j] = sgn[j];
      else if (kind[j] == SIGK_SECANT) sg[j] = sigs_secant(ws.ev[tq[j]][5], ws.ev[tq[j]][3], ws.ev[tq[j]][4]);  // rare
      if (is_ev[j]) {
        const double* e = ws.ev[tq[j]];
        ws.res[o[j] >> 6][o[j] & 63] = hval_interval_s(e[6], e[0], e[1], e[2], e[3], e[4], e[5], sg[j]);
      }
      if (need[j] && failed[j]) atomicOr(&ws.errmask[o[j] & 63], 1u << (o[j] >> 6));
    }
  }
  TICKB(22);
}

LT_DEV void run_tasks(WaveScratch& ws, int nchk, int nev, int lane, int first) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int total = nev + nchk;
  int t0 = first;
  const ExpAcc ea = exp_acc();   // (exp's addend constants in registers for the Newton trips and HVAL of every round)
#ifdef LT_RK_PAIRS
  // two tasks per lane and trip (measured: 3 % slower than one -- the rounds are bound by issue slots once
  // the two forms of a Newton step interleave, and the second task costs registers)
#pragma unroll 1
  for (; total - t0 > 64; t0 += 128) run_task_batch<2>(ws, nev, total, t0 + lane, ea);
#endif
#pragma unroll 1
  for (; t0 < total; t0 += 64)
    if (t0 + lane < total) run_task_batch<1>(ws, nev, total, t0 + lane, ea);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// find_currents, LTRANS.f90:1422-1614, executed by the WHOLE wave: lanes whose
// particle is not tracked (act = false) only take part in the cooperative pass.
// Per-particle constants of the water column are read from the step workspace
// (coalesced, L2-resident) where needed instead of living in registers.
// FS (free slip, hydrodynamic_module.f90:1934-1994, 2329-2520; an instantiation of its own, so that the kernel of every
// other run is the code it was): fsbits = the lane's element masks fsU | fsV << 4 | fsVU << 8 | fsR << 12 -- bit k of a
// mask: node k+1 of the element is water.  A lane whose element has a land node substitutes its values before every
// interpolation, on the element's node order, and takes the general form of the interpolation (r05; before, such a
// particle-step went to k_rk_slow: 5.2 ns each against 0.8 -- a run with a fifth of its particles in coastal elements
// took 2.0 ms per step instead of 1.0).
template <typename SlabT, int VT, bool FS>
LT_DEV void find_currents(const StepParams& P, const double* lev, WaveScratch& ws, bool act, int n,
                          int er, int eu, int ev_, double h, const Divisor& dv, double Xs, double Ys, double Zs, int version,
                          double& Uad, double& Vad, double& Wad, int& sigerr_count, bool& slow,
                          StageIn& SI, StageTail& ST, bool first_stage, bool last_stage, unsigned fsbits) {
  const double* rtab = lev;
  const double* wtab = lev + 3 * LT_LEV_STRIDE;
  const int lane = threadIdx.x & 63;
  const size_t N = (size_t)P.numpar;
  const double* wsd = P.ws_d + (act ? n : 0);
  TICK_INIT;
  Uad = 0.0; Vad = 0.0; Wad = 0.0;
  // regime tests :1481-1491
  bool zero = false, loglayer = false;
  int ii = 1, iii = 1;
  double wzb1 = 0.0;
  if (act) {
    const double zb = SI.zb, zc = SI.zc, zf = SI.zf;
    const LevCoef w1 = level_coef_d(wtab, 1, h, P.hc, VT, dv);
    const LevCoef r1 = level_coef_d(rtab, 1, h, P.hc, VT, dv);
    wzb1 = level_at(w1, zb, h, VT);
    const double wzc1 = level_at(w1, zc, h, VT), wzf1 = level_at(w1, zf, h, VT);
    zero = (Zs < wzb1 + P.z0) || (Zs < wzc1 + P.z0) || (Zs < wzf1 + P.z0);
    loglayer = !zero && ((Zs < level_at(r1, zb, h, VT)) || (Zs < level_at(r1, zc, h, VT)) ||
                         (Zs < level_at(r1, zf, h, VT)));
    if (!zero && !loglayer) {
      // lowest 4-level windows :1451-1467: lo = the lowest level in [3, L-1] with Zs below it at any of the
      // three times (the level depths increase with the level), found by bisection at the first stage and,
      // at the later stages, by walking from the window of the stage before (the stage depths of a step
      // differ by centimetres: almost always the same window, i.e. two probes instead of five per table)
      auto below = [&](const double* tab, int k) -> bool {
        const LevCoef q = level_coef_d(tab, k, h, P.hc, VT, dv);
        return (Zs < level_at(q, zb, h, VT)) | (Zs < level_at(q, zc, h, VT)) | (Zs < level_at(q, zf, h, VT));
      };
      int lo, lw;
      if (first_stage) {
        lo = 3;
        int hi = P.us - 1;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (below(rtab, mid)) hi = mid; else lo = mid + 1;
        }
        lw = 3; hi = P.ws - 1;
        while (lw < hi) {
          const int mid = (lw + hi) >> 1;
          if (below(wtab, mid)) hi = mid; else lw = mid + 1;
        }
      } else {
        lo = SI.ii + 2; lw = SI.iii + 2;
        while (lo > 3 && below(rtab, lo - 1)) lo--;
        while (lo < P.us - 1 && !below(rtab, lo)) lo++;
        while (lw > 3 && below(wtab, lw - 1)) lw--;
        while (lw < P.ws - 1 && !below(wtab, lw)) lw++;
      }
      ii = lo - 2; iii = lw - 2;
      SI.ii = ii; SI.iii = iii;
    }
  }
  const bool spl = act && !zero && !loglayer;  // this lane fits splines
  const LaneMask SPL = lm(spl);
  unsigned degenerate = 0;
  const double ixv = (version == 1) ? P.ix0 : (version == 2) ? P.ix1 : P.ix2;
  ws.errmask[lane] = 0;
  TICK(0);
  // ---------------- pass A: fit without iterating, queue what must iterate.
  // Written as straight-line code over ALL lanes: a lane without a spline to fit (not tracked,
  // below z0, log layer) runs the same arithmetic on safe dummy operands (element 1, levels
  // 1..4, h = 1) and is masked only where something leaves the lane -- the queues and the
  // result.  Divergent branches cost this kernel more than the arithmetic they skip: every
  // `if` on a lane condition is an exec-mask save / restore and a branch in the scalar stream
  // of a wave that has one other wave to overlap with.
  // Memory latency is taken off the critical path by asking early: the node ids of the
  // NEXT variable's element, and the 4x4 field values + zeta + depth of the NEXT time
  // slot, are requested before the arithmetic of the current one (same registers: the
  // raw values are dead once the horizontal interpolation has produced Y).
  int nchk = 0, nev = 0;  // wave-uniform task counts of this stage
  double zt = 0.0, pz = 0.0;
  SlabT raw[16];
  LevCoef lc[4];
  const unsigned esz = (unsigned)sizeof(SlabT);
  // one (variable, time slot) of a stage: the 16 field values (4 nodes x 4 levels: one 16-byte access per node in
  // the level-fastest layout, at 32-bit byte offsets e[] from the slab's base -- lt_set_grid refuses a slab of
  // 4 GiB) are requested one fit ahead of their use, into the registers the previous fit's values just left
  auto request = [&](const char* Fb, const unsigned (&e)[4], int L, int s) {
    const int ps = (s == 0) ? P.tb : (s == 1) ? P.tc : P.tf;
    const unsigned o = (unsigned)(ps * L) * esz;
    const Win4<SlabT> q0 = *(const Win4<SlabT>*)(Fb + (e[0] + o)), q1 = *(const Win4<SlabT>*)(Fb + (e[1] + o));
    const Win4<SlabT> q2 = *(const Win4<SlabT>*)(Fb + (e[2] + o)), q3 = *(const Win4<SlabT>*)(Fb + (e[3] + o));
#pragma unroll
    for (int l = 0; l < 4; l++) {
      raw[4 * l + 0] = q0.v[l]; raw[4 * l + 1] = q1.v[l];
      raw[4 * l + 2] = q2.v[l]; raw[4 * l + 3] = q3.v[l];
    }
  };
  auto request_zp = [&](int s) {   // zeta and P_z of a time slot (the same for the three variables)
    zt = wsd[(size_t)(WS_ZETA0 + s) * N];
    pz = wsd[(size_t)(WS_PZ0 + s) * N];
  };
  auto offsets = [&](const int4& nd, int L, int k0, unsigned (&e)[4]) {
    const unsigned rowb = (unsigned)(LT_SLOTS * L) * esz, kb = (unsigned)(k0 - 1) * esz;
    e[0] = (unsigned)(nd.x - 1) * rowb + kb; e[1] = (unsigned)(nd.y - 1) * rowb + kb;
    e[2] = (unsigned)(nd.z - 1) * rowb + kb; e[3] = (unsigned)(nd.w - 1) * rowb + kb;
  };
  // The second-triangle form (regime 2) is the first-triangle form on the nodes (3,4,1): v3+(v4-v3)t+(v1-v3)u.
  // Instead of selecting the operands at every one of the 12 interpolations of a variable, the node order itself
  // is rotated once (register swaps: v_swap_b32, which the compiler does not use, is one instruction per pair).
  // Regimes 3 / 4 (inverse distance, on a node) keep (1,2,3,4).
  auto rotate_offsets = [&](const QuadWeights& w, unsigned (&e)[4]) {
    if (w.regime == 2) { reg_swap(e[0], e[2]); reg_swap(e[1], e[3]); }
  };
  auto rotate_raw = [&](const QuadWeights& w) {
    if (w.regime == 2) {
#pragma unroll
      for (int l = 0; l < 4; l++) { reg_swap(raw[4 * l + 0], raw[4 * l + 2]); reg_swap(raw[4 * l + 1], raw[4 * l + 3]); }
    }
  };
  // the level geometry of a time slot -- the four level depths, their spacings, the five prepared divisors of YPC1 --
  // is the same for U and V (the same four rho levels, the same zeta): formed once per slot for the two
  double X[4], DX1 = 1.0, DX2 = 1.0, DX3 = 1.0;
  Divisor d1, d2, d3, d12, d23;
  LaneMask OK = 0ull;
  auto geometry = [&](double zeta) {
#pragma unroll
    for (int l = 0; l < 4; l++) X[l] = level_at(lc[l], zeta, h, VT);
    DX1 = X[1] - X[0]; DX2 = X[2] - X[1]; DX3 = X[3] - X[2];
    // Lane conditions are kept as 64-bit lane masks (LaneMask: the scalar result of a vector compare) and
    // combined with scalar instructions; written as `bool`s the compiler turns every and-or of them into a
    // select, which it builds in a vector register and compares back: ~18 vector instructions per slot.
    OK = SPL & lm(DX1 > 0.0) & lm(DX2 > 0.0) & lm(DX3 > 0.0);
    // (seven quotients of YPC1 over five divisors: the reciprocals are prepared once)
    d1 = prep_div(DX1); d2 = prep_div(DX2); d3 = prep_div(DX3);
    d12 = prep_div(DX1 + DX2); d23 = prep_div(DX2 + DX3);
  };
  // horizontal interpolation of the 4 levels of the slot in `raw`
  auto interp = [&](const QuadWeights& w, double (&Y)[4], int fskind, unsigned fsm, unsigned fsd) {
#pragma unroll
    for (int l = 0; l < 4; l++) {
      const double v1 = (double)raw[4 * l + 0], v2 = (double)raw[4 * l + 1], v3 = (double)raw[4 * l + 2];
      Y[l] = v1 + (v2 - v1) * w.t + (v3 - v1) * w.u;
    }
    if (FS && (fsm & 15u) != 15u) {   // free slip, an element with a land node: substitute, then the general form
#pragma unroll
      for (int l = 0; l < 4; l++) {
        double a = (double)raw[4 * l + 0], b = (double)raw[4 * l + 1], c = (double)raw[4 * l + 2], d = (double)raw[4 * l + 3];
        if (w.regime == 2) { double t_ = a; a = c; c = t_; t_ = b; b = d; d = t_; }   // (rotate_raw: back to the element's order)
        freeslip_sub(fskind, fsm, fsd, a, b, c, d);
        Y[l] = apply_interp(w, a, b, c, d);
      }
      return;
    }
    if (w.regime >= 3) {  // the stage position has left the start element (a few % of the lanes)
#pragma unroll
      for (int l = 0; l < 4; l++) {
        const double v1 = (double)raw[4 * l + 0], v2 = (double)raw[4 * l + 1], v3 = (double)raw[4 * l + 2],
                     v4 = (double)raw[4 * l + 3];
        if (w.regime == 3) Y[l] = w.W1 * v1 + w.W2 * v2 + w.W3 * v3 + w.W4 * v4;
        else Y[l] = (w.node_hit == 1) ? v1 : (w.node_hit == 2) ? v2 : (w.node_hit == 3) ? v3 : v4;
      }
    }
  };
  // the queues must have room for a fit (up to 64 evaluation and 128 check intervals): else one full round of
  // pass B first -- the check intervals waiting and the last 64 - nchk evaluation intervals (a second round if the
  // check intervals alone filled the first).  Wave-uniform.
  auto make_room = [&]() {
    while (nev + 64 > LT_EV_CAP || nchk + 64 > LT_CHK_CAP) {
      const int nc = (nchk < LT_CHK_CAP) ? nchk : LT_CHK_CAP;  // (entries beyond the capacity were not stored)
      const int room = (nc < 64) ? 64 - nc : 0;
      const int take = (nev < room) ? nev : room;
      run_tasks(ws, nc, nev, lane, nev - take);
      nev -= take;
      nchk = 0;
      TICK(4);
    }
  };
  // YPC1, the screening of the three intervals, the queues, HVAL of the zero-tension case: the fit of one
  // (variable, slot) on the geometry in hand
  auto fit = [&](const double (&Y)[4], const int slot9, const double Pzs) {
    degenerate |= lm_lane(SPL & ~OK) ? (1u << slot9) : 0u;
    // YPC1 (tension_module.f90:852-977) with the chord slopes kept for SIGS/HVAL
    double YP[4], SL[3];
    SL[0] = div_prep(Y[1] - Y[0], d1);
    SL[1] = div_prep(Y[2] - Y[1], d2);
    SL[2] = div_prep(Y[3] - Y[2], d3);
    YP[0] = ypc1_end_b(SL[0], SL[0] + div_prep(DX1 * (SL[0] - SL[1]), d12));
    YP[1] = ypc1_mid_b(DX1, DX2, d12, SL[0], SL[1]);
    YP[2] = ypc1_mid_b(DX2, DX3, d23, SL[1], SL[2]);
    YP[3] = ypc1_end_b(SL[2], SL[2] + div_prep(DX3 * (SL[2] - SL[1]), d23));
    // the evaluation interval :1044-1053 (extrapolates with the end intervals)
    // (for strictly increasing knots "T > X(4) -> interval 3" is implied; other windows are redone by k_rk_slow)
    const LaneMask I1 = lm(Pzs < X[1]), I2 = ~I1 & lm(Pzs < X[2]), I3 = ~I1 & ~I2;
    const bool i1 = lm_lane(I1), i2 = lm_lane(I2);
    const double x1 = i1 ? X[0] : i2 ? X[1] : X[2];
    const double x2 = i1 ? X[1] : i2 ? X[2] : X[3];
    const double y1 = i1 ? Y[0] : i2 ? Y[1] : Y[2];
    const double p1 = i1 ? YP[0] : i2 ? YP[1] : YP[2];
    const double p2 = i1 ? YP[1] : i2 ? YP[2] : YP[3];
    const double Sv = i1 ? SL[0] : i2 ? SL[1] : SL[2];
    Divisor dI;
    dI.b = i1 ? d1.b : i2 ? d2.b : d3.b;
    dI.y = i1 ? d1.y : i2 ? d2.y : d3.y;
    TICK(2);
    // SIGS (tension_module.f90:448-776), screened without divisions or loops.
    // Only the evaluation interval's tension is used by HVAL; the other two
    // matter only through SigErr, which only the convexity Newton iteration
    // can raise, so they are queued just when they are Newton candidates.
    const LaneMask C0 = sigs_candidate_mask(SL[0], YP[0], YP[1]), C1 = sigs_candidate_mask(SL[1], YP[1], YP[2]),
                   C2 = sigs_candidate_mask(SL[2], YP[2], YP[3]);
    // the two non-evaluation intervals: I=1 -> (2,3), I=2 -> (1,3), I=3 -> (1,2)
    const LaneMask ma = OK & ((I1 & C1) | (~I1 & C0)), mb = OK & ((I3 & C1) | (~I3 & C2));
    // the evaluation interval: done here when its tension is known to be ZERO (the cubic form);
    // everything else (an iteration, or the rare known SBIG) goes to pass B
    const LaneMask me = OK & ~sigs_eval_zero_mask(Sv, p1, p2);
    const LaneMask mc = ma | mb;
    const unsigned long long below = (1ull << lane) - 1ull;
    if (mc) {  // ~1 % of the intervals are candidates: the operands of T = MAX(D1/D2, D2/D1)
      if (lm_lane(ma)) {
        double na, da;
        if (i1) (void)sigs_candidate(SL[1], YP[1], YP[2], na, da);
        else (void)sigs_candidate(SL[0], YP[0], YP[1], na, da);
        const int idx = nchk + __popcll(ma & below);
        if (idx < LT_CHK_CAP) {
          ws.chk[idx][0] = na; ws.chk[idx][1] = da;
          ws.chk_owner[idx] = (unsigned short)(lane | (slot9 << 6));
        } else degenerate |= (1u << slot9);
      }
      nchk += __popcll(ma);
      if (lm_lane(mb)) {
        double nb, db;
        if (lm_lane(I3)) (void)sigs_candidate(SL[1], YP[1], YP[2], nb, db);
        else (void)sigs_candidate(SL[2], YP[2], YP[3], nb, db);
        const int idx = nchk + __popcll(mb & below);
        if (idx < LT_CHK_CAP) {
          ws.chk[idx][0] = nb; ws.chk[idx][1] = db;
          ws.chk_owner[idx] = (unsigned short)(lane | (slot9 << 6));
        } else degenerate |= (1u << slot9);
      }
      nchk += __popcll(mb);
    }
    if (lm_lane(me)) {
      const int idx = nev + __popcll(me & below);
      double* e = ws.ev[idx];
      e[0] = x1; e[1] = x2; e[2] = y1; e[3] = p1; e[4] = p2; e[5] = Sv; e[6] = Pzs;
      ws.ev_owner[idx] = (unsigned short)(lane | (slot9 << 6));
    }
    nev += __popcll(me);
#ifdef LT_RK_TIMING
    if (lane == 0) ws.tim[6] += __popcll(me);
#endif
    // HVAL of the zero-tension case, stored by every lane (pass B overwrites the queued ones,
    // the log layer below the others that do not fit a spline)
    ws.res[slot9][lane] = hval_cubic_d(Pzs, x1, dI, y1, p1, p2, Sv);
    TICK(3);
  };
  // the log layer :1493-1522 (level 1 of u, v; level 2 of w) for the lanes below the deepest rho level: after the
  // fits of the variable, whose stores it overwrites
  auto log_layer = [&](int var, const int4& nd, const SlabT* F, int L) {
    const int k = (var == 2) ? 2 : 1;
    const Quad q = (var == 2) ? load_quad16(P.rnode, nd, (unsigned)sizeof(RhoNode))
                 : load_quad16((var == 0) ? (const void*)P.unode : (const void*)P.vnode, nd, (unsigned)sizeof(XYNode));
    const LevCoef qt = level_coef_d((var == 2) ? wtab : rtab, k, h, P.hc, VT, dv);
    const QuadWeights wl = quad_weights(Xs, Ys, q.x, q.y);
#pragma unroll
    for (int s = 0; s < 3; s++) {
      const int ps = (s == 0) ? P.tb : (s == 1) ? P.tc : P.tf;
      const double zeta = wsd[(size_t)(WS_ZETA0 + s) * N];
      double f1 = fldv(F, nd.x, L, k, ps), f2 = fldv(F, nd.y, L, k, ps), f3 = fldv(F, nd.z, L, k, ps), f4 = fldv(F, nd.w, L, k, ps);
      if (FS) {
        const unsigned fsm = (var == 0) ? (fsbits & 15u) : (var == 1) ? ((fsbits >> 4) & 15u) : ((fsbits >> 12) & 15u);
        if (fsm != 15u) freeslip_sub((var == 0) ? 1 : (var == 1) ? 2 : 0, fsm, (fsbits >> 8) & 15u, f1, f2, f3, f4);
      }
      double val = apply_interp(wl, f1, f2, f3, f4);
      double ztop = level_at(qt, zeta, h, VT);
      ws.res[var * 3 + s][lane] = val * log10_call((Zs - wzb1) / P.z0) / log10_call((ztop - wzb1) / P.z0);
    }
  };
  // ---- U and V: the two set-ups, then slot by slot the shared geometry and the two fits
  {
    const int4 ndU = SI.nd0, ndV = SI.nd1;
    const int L = P.us;
    const char* FbU = (const char*)P.U;
    const char* FbV = (const char*)P.V;
    unsigned eU[4], eV[4];
    offsets(ndU, L, ii, eU);
    offsets(ndV, L, ii, eV);
    request(FbU, eU, L, 0);
    request_zp(0);
    const Quad qU = load_quad16(P.unode, ndU, (unsigned)sizeof(XYNode));
    const QuadWeights wU = quad_weights_f(Xs, Ys, qU.x, qU.y);
    rotate_offsets(wU, eU);
    rotate_raw(wU);
    const Quad qV = load_quad16(P.vnode, ndV, (unsigned)sizeof(XYNode));
    const QuadWeights wV = quad_weights_f(Xs, Ys, qV.x, qV.y);
    rotate_offsets(wV, eV);
#ifdef LT_RK_TIMING
    { const unsigned long long m3 = __ballot(spl && (wU.regime >= 3 || wV.regime >= 3)); if (lane == 0) ws.tim[7] += __popcll(m3) + ((m3 != 0) ? (1ull << 32) : 0ull); }
#endif
    TICK(1);
#pragma unroll
    for (int l = 0; l < 4; l++) lc[l] = level_coef_d(rtab, ii + l, h, P.hc, VT, dv);
#pragma unroll 1
    for (int s = 0; s < 3; s++) {
      make_room();
      const double Pzs = pz;
      geometry(zt);
      double Y[4];
      interp(wU, Y, 1, fsbits & 15u, 15u);
      request(FbV, eV, L, s);            // V's values of this slot arrive while U is fitted
      fit(Y, s, Pzs);
      make_room();
      interp(wV, Y, 2, (fsbits >> 4) & 15u, (fsbits >> 8) & 15u);
      if (s < 2) { request(FbU, eU, L, s + 1); request_zp(s + 1); }
      fit(Y, 3 + s, Pzs);
    }
    if (loglayer) {
      log_layer(0, ndU, (const SlabT*)P.U, L);
      log_layer(1, ndV, (const SlabT*)P.V, L);
    }
  }
  // ---- W
  {
    const int4 nd = P.RE[er - 1];
    const int L = P.ws;
    const char* Fb = (const char*)P.W;
    unsigned e[4];
    offsets(nd, L, iii, e);
    request(Fb, e, L, 0);
    request_zp(0);
    const Quad q = load_quad16(P.rnode, nd, (unsigned)sizeof(RhoNode));
    const QuadWeights w = quad_weights_f(Xs, Ys, q.x, q.y);
    rotate_offsets(w, e);
    rotate_raw(w);
#ifdef LT_RK_TIMING
    { const unsigned long long m3 = __ballot(spl && w.regime >= 3); if (lane == 0) ws.tim[7] += __popcll(m3) + ((m3 != 0) ? (1ull << 32) : 0ull); }
#endif
    TICK(1);
#pragma unroll
    for (int l = 0; l < 4; l++) lc[l] = level_coef_d(wtab, iii + l, h, P.hc, VT, dv);
#pragma unroll 1
    for (int s = 0; s < 3; s++) {
      make_room();
      const double Pzs = pz;
      geometry(zt);
      double Y[4];
      interp(w, Y, 0, (fsbits >> 12) & 15u, 15u);
      if (s < 2) { request(Fb, e, L, s + 1); request_zp(s + 1); }
      fit(Y, 6 + s, Pzs);
    }
    if (loglayer) log_layer(2, nd, (const SlabT*)P.W, L);
  }
  // what the step to the next stage and the next stage itself need first (coalesced per-particle
  // constants, the node ids of the u element) is requested here: the round below covers the latency
  // (also at the last stage, where nothing uses them: a conditional assignment would keep the old
  // values alive -- 26 registers -- through all of pass A)
  {
    ST.X = P.par[(size_t)pX * N + (act ? n : 0)]; ST.Y = P.par[(size_t)pY * N + (act ? n : 0)];
    ST.Zpar = wsd[(size_t)WS_ZPAR * N];
    ST.cosA = wsd[(size_t)WS_COSA * N]; ST.sinA = wsd[(size_t)WS_SINA * N];
    ST.minpd = wsd[(size_t)WS_MINPD * N]; ST.maxpd = wsd[(size_t)WS_MAXPD * N];
    SI.zb = wsd[(size_t)WS_ZETA0 * N]; SI.zc = wsd[(size_t)WS_ZETA1 * N]; SI.zf = wsd[(size_t)WS_ZETA2 * N];
    SI.nd0 = P.UE[eu - 1];
    SI.nd1 = P.VE[ev_ - 1];
  }
  // ---------------- pass B: what is left of this stage
  run_tasks(ws, (nchk < LT_CHK_CAP) ? nchk : LT_CHK_CAP, nev, lane, 0);
  TICK(4);
  // ---------------- pass C: time interpolation :1534-1598 / hydrodynamic_module.f90:2653-2670
  if (act && !zero) {
#pragma unroll
    for (int var = 0; var < 3; var++) {
      const double Pv0 = ws.res[var * 3 + 0][lane], Pv1 = ws.res[var * 3 + 1][lane], Pv2 = ws.res[var * 3 + 2][lane];
      double e1 = Pv0, e2 = Pv1, e3 = Pv2;
      if (P.p == 1) { e2 = Pv0; e3 = Pv1; }
      const double val = polintd3(P.ex0, P.ex1, P.ex2, e1, e2, e3, ixv);
      if (var == 0) Uad = val; else if (var == 1) Vad = val; else Wad = val;
    }
  }
  TICK(5);
  if (act && !zero) {
    const unsigned em = ws.errmask[lane];
    if (em) sigerr_count += __popc(em);
    if (spl && (degenerate | em)) slow = true;  // this particle-step is redone by k_rk_slow
  }
  // the next stage reuses the scratch: all lanes must have read their results
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  TICK(5);
}

// ------------------------------------------------------------ the step
// The kernels of this file exchange a per-particle workspace in HBM (WS_* columns + one flag
// word, SoA, stride numpar) with k_post and the VTurb kernels, so that the register-hungry
// RK4 part does not carry the state of the element search and boundary code (lt_kernels.h).

template <typename SlabT>
#ifdef LT_PRE_WAVES
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LT_PRE_WAVES, LT_PRE_WAVES))) void k_pre(const StepParams P) {
#else
__global__ __launch_bounds__(256) void k_pre(const StepParams P) {
#endif
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  // the step's diagnostic counters (SigErr, errors, ...): later kernels of the step add to them
  if (blockIdx.x == 0 && threadIdx.x < CNT_COUNT) P.counters[threadIdx.x] = 0ull;
  unsigned c_active = 0;
  if (n < P.numpar) {
    const size_t N = (size_t)P.numpar;
    double* par = P.par + n;
    double* wsd = P.ws_d + n;
#define PARV(k) par[(size_t)(k) * N]
    // Everything the gates may look at is requested at once (the kernel is a chain of
    // dependent loads, not arithmetic): state, status flags, the stored elements and,
    // through them, the nodes of the rho element.
    const double X = PARV(pX), Y = PARV(pY);
    double Z = PARV(pZ);
    const double dob = PARV(pDOB), age0 = PARV(pAge);
    const int settle_n = P.settlementon ? P.settle[n] : 0;
    const double deadage_n = P.mortality ? P.P_deadage[n] : 0.0;
    int dead_n = P.mortality ? P.dead[n] : 0;
    const int oob_n = P.OpenOceanBoundary ? P.oob[n] : 0;
    const int ele_ok = P.ele_ok[n];
    int er = P.P_r_element[n], eu = P.P_u_element[n], ev = P.P_v_element[n];
    int4 rn = P.RE[er - 1], un, vn;  // stored element ids are always valid ids
    int flags = 0, err = 0;
    // the gates of LTRANS.f90:790-816
    const bool released = !(P.ix2 <= dob);
    bool track = released;
    if (released) {
      double age = age0 + (double)(float)P.idt;
      PARV(pAge) = age;
      // updateStatus, behavior_module.f90:162-179
      bool settled = P.settlementon ? (settle_n != 0) : false;
      if (P.mortality && age >= deadage_n) {
        if (!(P.settlementon && settled)) { P.dead[n] = 1; dead_n = 1; }
      }
      if (P.settlementon && settled) track = false;
      if (track && P.mortality && dead_n != 0) track = false;
      if (track && P.OpenOceanBoundary && oob_n != 0) track = false;
      if (!track) flags |= FL_KEEP_PN;
    }
    if (track) {
      // setEle :830 (all three grids are searched even after a miss).  When the
      // previous step's closing setEle (:1317) found these very elements for this
      // very position, the first probe of each grid is that element and succeeds
      // again: the search is skipped and only the rho nodes are used.
      bool okr = true, oku = true, okv = true;
      if (!ele_ok) {
        find_elements3(P, X, Y, er, eu, ev, rn, un, vn, okr, oku, okv);
        P.P_r_element[n] = er; P.P_u_element[n] = eu; P.P_v_element[n] = ev;
      }
      if (!okv) err = 6; else if (!oku) err = 5; else if (!okr) err = 4;  // last assignment wins
      if (err == 0) {
        flags |= FL_ACT;
        c_active = 1;
        // setInterp + getInterp :882-896
        RhoNode r1 = P.rnode[rn.x - 1], r2 = P.rnode[rn.y - 1], r3 = P.rnode[rn.z - 1],
                r4 = P.rnode[rn.w - 1];
        double qx[4] = {r1.x, r2.x, r3.x, r4.x}, qy[4] = {r1.y, r2.y, r3.y, r4.y};
        const QuadWeights w0 = quad_weights(X, Y, qx, qy);
        // (free slip: the values on the element's land nodes are replaced first, :1934-1994)
        const unsigned fsm = P.FreeSlip ? P.fsR[er - 1] : 15u;
        auto gi = [&](double a, double b, double c, double d) -> double {
          if (fsm != 15u) freeslip_sub(0, fsm, 15u, a, b, c, d);
          return apply_getinterp(w0, a, b, c, d);
        };
        const double P_depth = -1.0 * gi(r1.depth, r2.depth, r3.depth, r4.depth);
        const double P_angle = gi(r1.angle, r2.angle, r3.angle, r4.angle);
        using Z4 = typename SlabTraits<SlabT>::zeta4;
        const Z4* zt = (const Z4*)P.zeta;
        Z4 z1 = zt[rn.x - 1], z2 = zt[rn.y - 1], z3 = zt[rn.z - 1], z4 = zt[rn.w - 1];
        auto pick = [](const Z4& v, int s) -> double {
          return (double)((s == 0) ? v.x : (s == 1) ? v.y : (s == 2) ? v.z : v.w);
        };
        double zeta[3], Pz[3];
        zeta[0] = gi(pick(z1, P.tb), pick(z2, P.tb), pick(z3, P.tb), pick(z4, P.tb));
        zeta[1] = gi(pick(z1, P.tc), pick(z2, P.tc), pick(z3, P.tc), pick(z4, P.tc));
        zeta[2] = gi(pick(z1, P.tf), pick(z2, P.tf), pick(z3, P.tf), pick(z4, P.tf));
        // vertical clamps :900-922
        if (Z < P_depth) {
          Z = P_depth + (double)0.001f;
          if (P.TrackCollisions) P.hitBottom[n] += 1;
        }
        Pz[0] = Pz[1] = Pz[2] = Z;
        if (Z > zeta[0]) Pz[0] = zeta[0] - (double)0.001f;
        if (Z > zeta[1]) Pz[1] = zeta[1] - (double)0.001f;
        if (Z > zeta[2]) Pz[2] = zeta[2] - (double)0.001f;
        const double Zpar = polintd3(P.ex0, P.ex1, P.ex2, Pz[0], Pz[1], Pz[2], P.ix1);
        wsd[(size_t)WS_PDEPTH * N] = P_depth;
        double sa, ca;  // one sincos call in a GCC-compiled host program
        lt_sincos_g(P_angle, &sa, &ca);
        wsd[(size_t)WS_COSA * N] = ca;
        wsd[(size_t)WS_SINA * N] = sa;
        wsd[(size_t)WS_ZETA0 * N] = zeta[0];
        wsd[(size_t)WS_ZETA1 * N] = zeta[1];
        wsd[(size_t)WS_ZETA2 * N] = zeta[2];
        wsd[(size_t)WS_PZ0 * N] = Pz[0];
        wsd[(size_t)WS_PZ1 * N] = Pz[1];
        wsd[(size_t)WS_PZ2 * N] = Pz[2];
        wsd[(size_t)WS_ZPAR * N] = Zpar;  // = par(n,pZ) after :922
        {  // maxpartdepth / minpartdepth :981-987: the limits of every RK stage depth
          const double* wtab = P.lev + 3 * LT_LEV_STRIDE;
          const double hh = -1.0 * P_depth;
          double mx = level_z(wtab, 1, zeta[0], hh, P.hc, P.Vtransform);
          double t1 = level_z(wtab, 1, zeta[1], hh, P.hc, P.Vtransform), t2 = level_z(wtab, 1, zeta[2], hh, P.hc, P.Vtransform);
          if (t1 > mx) mx = t1;
          if (t2 > mx) mx = t2;
          double mn = level_z(wtab, P.ws, zeta[0], hh, P.hc, P.Vtransform);
          t1 = level_z(wtab, P.ws, zeta[1], hh, P.hc, P.Vtransform); t2 = level_z(wtab, P.ws, zeta[2], hh, P.hc, P.Vtransform);
          if (t1 < mn) mn = t1;
          if (t2 < mn) mn = t2;
          wsd[(size_t)WS_MINPD * N] = mn;
          wsd[(size_t)WS_MAXPD * N] = mx;
        }
      }
    }
    P.ws_i[n] = flags | (err << FL_ERR_SHIFT);
#undef PARV
  }
  // tracked particles of this block -> block_active[block]; k_post's block 0 adds them up
  // (one atomic per wave on a single counter serialised at the L2: 160 us per step)
  __shared__ unsigned wsum[4];
  unsigned v = c_active;
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) P.block_active[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// WCTS_ITPI (hydrodynamic_module.f90:2577-2688) of a rho-grid field with us levels (salinity,
// temperature) at the start-of-step position, version 4 = (vb + 4 vc + vf)/6: straight
// transcription (used once or twice per particle-step, only with behaviours 4,5,7 / SaltTempOn)
template <typename SlabT>
LT_DEV double wcts_itpi_rho(const StepParams& P, const double* lev, const Column& C, int4 rn,
                            const QuadWeights& w, const SlabT* F, int deplvl, unsigned fsm) {
  double Pv[3];
#pragma unroll 1
  for (int s = 0; s < 3; s++) {
    const int ps = (s == 0) ? P.tb : (s == 1) ? P.tc : P.tf;
    double az[4], av[4];
#pragma unroll
    for (int l = 0; l < 4; l++) {
      az[l] = level_z(lev, deplvl + l, C.zeta[s], C.h, P.hc, P.Vtransform);
      double f1 = fld(F, rn.x, P.us, deplvl + l, ps), f2 = fld(F, rn.y, P.us, deplvl + l, ps);
      double f3 = fld(F, rn.z, P.us, deplvl + l, ps), f4 = fld(F, rn.w, P.us, deplvl + l, ps);
      if (fsm != 15u) freeslip_sub(0, fsm, 15u, f1, f2, f3, f4);
      av[l] = apply_interp(w, f1, f2, f3, f4);
    }
    int dummy = 0;
    Pv[s] = spline4_eval(az, av, C.Pz[s], dummy);
  }
  double e1 = Pv[0], e2 = Pv[1], e3 = Pv[2];
  if (P.p == 1) { e2 = Pv[0]; e3 = Pv[1]; }
  const double vb = polintd3(P.ex0, P.ex1, P.ex2, e1, e2, e3, P.ix0);
  const double vc = polintd3(P.ex0, P.ex1, P.ex2, e1, e2, e3, P.ix1);
  const double vf = polintd3(P.ex0, P.ex1, P.ex2, e1, e2, e3, P.ix2);
  return (vb + vc * 4 + vf) / 6.0;
}

// The same for a whole wave (r05): the salinity / temperature fits of k_beh_pre.  The straight transcription above runs
// SIGS' convexity Newton iteration inside each of its nine intervals (three time slots x three intervals), every time for
// as many trips as the slowest lane needs, with a tenth to a quarter of the lanes in it: 80 % of the kernel.  Here the
// intervals that need the iteration are first LISTED (their T, in LDS, packed by wave ballots), the list is iterated 64
// at a time by the wave-level loop k_rk and k_vt_sigs use (sigs_newton_w: same operations, same cycle detection), and a
// second pass over the slots picks the results up: the tension factor of the evaluation interval, and SigErr (linint
// instead of HVAL, :2636-2644) if ANY iteration of the fit failed -- SIGS returns at the first failure, and whichever
// interval it was the fit falls back.  Every lane of the wave is here; `want` says which lanes have a fit.
struct BehScratch {
  double T[9 * 64];            // the listed intervals' T = MAX(D1/D2, D2/D1)
  double sg[9][64];            // tension factor per (slot * 3 + interval, lane)
  unsigned short own[9 * 64];  // (slot * 3 + interval) << 6 | lane
  unsigned int fail[64];       // bit slot * 3 + interval: the iteration failed (SigErr)
};
template <typename SlabT> struct __attribute__((aligned(4 * sizeof(SlabT)))) Slots4 { SlabT v[LT_SLOTS]; };
template <typename SlabT>
LT_DEV double wcts_itpi_rho_w(const StepParams& P, const double* lev, const Column& C, int4 rn, const QuadWeights& w,
                              const SlabT* F, int deplvl, unsigned fsm, bool want, BehScratch& bs) {
  static_assert(LT_SLOTS == 4, "one aligned access per (node, level) holds the four time slots");
  const int lane = threadIdx.x & 63;
  // the fits' operands, once: X = the level depths, Y = the field at the particle, YP = YPC1's derivatives, per slot.
  // (the four time slots of a (node, level) are 16 contiguous bytes in this layout, [node][level][slot]: one access
  // instead of three)
  double X[3][4], Y[3][4], YP[3][4];
  int ier[3] = {0, 0, 0};
  if (want) {
    const int ps[3] = {P.tb, P.tc, P.tf};
#pragma unroll
    for (int l = 0; l < 4; l++) {
      const int k = deplvl + l;
      const Slots4<SlabT> q1 = *(const Slots4<SlabT>*)(F + ((size_t)(rn.x - 1) * P.us + (k - 1)) * LT_SLOTS);
      const Slots4<SlabT> q2 = *(const Slots4<SlabT>*)(F + ((size_t)(rn.y - 1) * P.us + (k - 1)) * LT_SLOTS);
      const Slots4<SlabT> q3 = *(const Slots4<SlabT>*)(F + ((size_t)(rn.z - 1) * P.us + (k - 1)) * LT_SLOTS);
      const Slots4<SlabT> q4 = *(const Slots4<SlabT>*)(F + ((size_t)(rn.w - 1) * P.us + (k - 1)) * LT_SLOTS);
#pragma unroll
      for (int s = 0; s < 3; s++) {
        const int c = ps[s];   // (uniform)
        auto pick = [c](const Slots4<SlabT>& q) -> double {
          return (double)((c == 0) ? q.v[0] : (c == 1) ? q.v[1] : (c == 2) ? q.v[2] : q.v[3]);
        };
        double f1 = pick(q1), f2 = pick(q2), f3 = pick(q3), f4 = pick(q4);
        if (fsm != 15u) freeslip_sub(0, fsm, 15u, f1, f2, f3, f4);
        Y[s][l] = apply_interp(w, f1, f2, f3, f4);
        X[s][l] = level_z(lev, k, C.zeta[s], C.h, P.hc, P.Vtransform);
      }
    }
#pragma unroll
    for (int s = 0; s < 3; s++) ypc1_4(X[s], Y[s], YP[s], ier[s]);
  }
  bs.fail[lane] = 0u;
  int ntask = 0;   // wave-uniform
  // ---- first pass: list the intervals that need the Newton iteration
#pragma unroll
  for (int s = 0; s < 3; s++) {
#pragma unroll
    for (int iv = 0; iv < 3; iv++) {
      double T = 0.0;
      bool need = false;
      if (want && ier[s] == 0) {
        double sg0;
        need = sigs_classify((Y[s][iv + 1] - Y[s][iv]) / (X[s][iv + 1] - X[s][iv]), YP[s][iv], YP[s][iv + 1], sg0, T) == SIGK_NEWTON;
      }
      const LaneMask m = lm(need);
      if (need) {
        const int idx = ntask + __popcll(m & ((1ull << lane) - 1ull));
        bs.T[idx] = T;
        bs.own[idx] = (unsigned short)(((s * 3 + iv) << 6) | lane);
      }
      ntask += __popcll(m);
    }
  }
  // ---- the listed iterations, 64 at a time
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const ExpAcc ea = exp_acc();
#pragma unroll 1
  for (int t0 = 0; t0 < ntask; t0 += 64) {
    const bool have = t0 + lane < ntask;
    const double T = have ? bs.T[t0 + lane] : 0.0;
    const unsigned o = have ? bs.own[t0 + lane] : 0u;
    double sgn;
    LaneMask FAILED;
    const LaneMask NEED = lm(have);
    sigs_newton_w(T, NEED, sgn, FAILED, ea);
    if (have) {
      bs.sg[o >> 6][o & 63u] = sgn;
      if (lm_lane(FAILED)) atomicOr(&bs.fail[o & 63u], 1u << (o >> 6));
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // ---- second pass: the value of each slot's fit (spline4_eval with the tension factors in hand)
  if (!want) return 0.0;
  const unsigned failbits = bs.fail[lane];
  double Pv[3];
#pragma unroll
  for (int s = 0; s < 3; s++) {
    const double T = C.Pz[s];
    if (ier[s] != 0) YP[s][0] = YP[s][1] = YP[s][2] = YP[s][3] = 0.0;     // (as spline4_eval: a degenerate column)
    if (ier[s] == 0 && ((failbits >> (3 * s)) & 7u) != 0u) { Pv[s] = linint4(X[s], Y[s], T); continue; }
    int I = (T < X[s][1]) ? 1 : (T < X[s][2]) ? 2 : 3;
    if (T > X[s][3]) I = 3;
    const double x1 = (I == 1) ? X[s][0] : (I == 2) ? X[s][1] : X[s][2];
    const double x2 = (I == 1) ? X[s][1] : (I == 2) ? X[s][2] : X[s][3];
    const double y1 = (I == 1) ? Y[s][0] : (I == 2) ? Y[s][1] : Y[s][2];
    const double y2 = (I == 1) ? Y[s][1] : (I == 2) ? Y[s][2] : Y[s][3];
    const double p1 = (I == 1) ? YP[s][0] : (I == 2) ? YP[s][1] : YP[s][2];
    const double p2 = (I == 1) ? YP[s][1] : (I == 2) ? YP[s][2] : YP[s][3];
    double sg = 0.0;
    if (ier[s] == 0) {
      double Tq;
      const int kind = sigs_classify((y2 - y1) / (x2 - x1), p1, p2, sg, Tq);
      if (kind == SIGK_NEWTON) sg = bs.sg[s * 3 + (I - 1)][lane];
      else if (kind == SIGK_SECANT) sg = sigs_secant((y2 - y1) / (x2 - x1), p1, p2);
    }
    Pv[s] = (x2 - x1 <= 0.0) ? 0.0 : hval_interval(T, x1, x2, y1, y2, p1, p2, sg);
  }
  double e1 = Pv[0], e2 = Pv[1], e3 = Pv[2];
  if (P.p == 1) { e2 = Pv[0]; e3 = Pv[1]; }
  const double vb = polintd3(P.ex0, P.ex1, P.ex2, e1, e2, e3, P.ix0);
  const double vc = polintd3(P.ex0, P.ex1, P.ex2, e1, e2, e3, P.ix1);
  const double vf = polintd3(P.ex0, P.ex1, P.ex2, e1, e2, e3, P.ix2);
  return (vb + vc * 4 + vf) / 6.0;
}

// Everything of behave() (behavior_module.f90:181-551) and of the SaltTempOn block
// (LTRANS.f90:1062-1076) that does not depend on the random deviates or on the RK4 result:
// swimming speed, stage changes, the C. ariakensis timer, the salinity at the particle, the
// salinity-gradient cue, and with them WHAT the behaviour will draw (none / one directed
// deviate / two random-swimming deviates / both), which the draw offsets of all later
// particles depend on.  k_post consumes the deviates and finishes the behaviour.
template <typename SlabT>
__global__ __launch_bounds__(256) void k_beh_pre(const StepParams P) {
  __shared__ double lev[6 * LT_LEV_STRIDE];
  __shared__ BehScratch bscr[4];
  for (int i = threadIdx.x; i < 6 * LT_LEV_STRIDE; i += blockDim.x) lev[i] = P.lev[i];
  exp_table_load();
  // (every lane of a wave stays to the end of the salinity fits: they are done by the wave, wcts_itpi_rho_w; a lane
  // without a tracked particle works on particle 0's numbers and stores nothing)
  const int n0 = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = n0 < P.numpar && (P.ws_i[n0 < P.numpar ? n0 : 0] & FL_ACT) != 0;
  const int n = live ? n0 : 0;
  int fl = P.ws_i[n];
  const size_t N = (size_t)P.numpar;
  double* wsd = P.ws_d + n;
  const double X = P.par[(size_t)pX * N + n], Y = P.par[(size_t)pY * N + n];
  const double P_age = P.par[(size_t)pAge * N + n];
  const double Zpar = wsd[(size_t)WS_ZPAR * N];
  Column C;
  C.h = -1.0 * wsd[(size_t)WS_PDEPTH * N];
#pragma unroll
  for (int s = 0; s < 3; s++) {
    C.zeta[s] = wsd[(size_t)(WS_ZETA0 + s) * N];
    C.Pz[s] = wsd[(size_t)(WS_PZ0 + s) * N];
  }
  const double P_depth = -C.h, P_zc = C.Pz[1], P_zetac = C.zeta[1];
  int beh = (live && P.Behavior != 0) ? P.P_behave[n] : 0;
  double tim = 0.0;
  if (live && P.Behavior != 0) {
    // :214-215 swimming speed at the current age
    double sw3 = P.P_swim[n + 2 * N];
    bool dirty = false;
    if (P_age >= P.swimstart) { sw3 = P.P_swim[n] * P_age + P.P_swim[n + N]; dirty = true; }
    if (P_age >= P.P_pediage[n]) { sw3 = P.swimfast; dirty = true; }
    if (dirty) P.P_swim[n + 2 * N] = sw3;
    // :220-229 oyster larvae: pediveliger stage, timer
    if (beh == 4 || beh == 5) {
      if (P_age >= P.P_pediage[n] && P_age < P.P_deadage[n]) { beh = 2; P.P_behave[n] = 2; }
      tim = P.timer[n];
      tim = (0.0 > tim - (double)P.dt) ? 0.0 : tim - (double)P.dt;
      P.timer[n] = tim;
    }
  }
  // salinity (and temperature) at the particle :233-246, LTRANS.f90:1062-1076
  const bool want_S = live && ((beh == 4) || (beh == 5 && tim == 0.0) || (beh == 7) || P.SaltTempOn);
  double P_S = 0.0;
  {
    BehScratch& bs = bscr[threadIdx.x >> 6];
    int4 rn = make_int4(1, 1, 1, 1);
    QuadWeights w;
    w.regime = 1; w.node_hit = 0; w.t = w.u = w.W1 = w.W2 = w.W3 = w.W4 = 0.0;
    int deplvl = 1;
    unsigned fsm = 15u;
    if (want_S) {
      rn = P.RE[P.P_r_element[n] - 1];
      const Quad q = load_quad_rho(P.rnode, rn);
      w = quad_weights(X, Y, q.x, q.y);
      deplvl = level_window(lev, P.us, Zpar, C.zeta[0], C.zeta[1], C.zeta[2], C.h, P.hc, P.Vtransform);
      fsm = P.FreeSlip ? P.fsR[P.P_r_element[n] - 1] : 15u;
    }
    if (__ballot(want_S)) P_S = wcts_itpi_rho_w<SlabT>(P, lev, C, rn, w, (const SlabT*)P.S, deplvl, fsm, want_S, bs);
    if (P.SaltTempOn) {      // (uniform)
      const double P_T = wcts_itpi_rho_w<SlabT>(P, lev, C, rn, w, (const SlabT*)P.T, deplvl, fsm, want_S, bs);
      if (live) { P.P_Salt[n] = P_S; P.P_Temp[n] = P_T; }
    }
  }
  if (!live) return;
  bool directed = false, random = false;
  double thr1 = 0.0, thr2 = 0.0;
  // TYPE 1 :256-283
  if (beh == 1) {
    random = true;
    thr2 = (P_zc < (P_zetac - 1.0)) ? (double)0.80f : 0.5;
  }
  // TYPE 2 :287-313 (also C. ariakensis while its timer runs)
  if (beh == 2 || (beh == 5 && tim > 0.0)) {
    random = true;
    thr2 = (P_zc > (P_depth + 1.0)) ? (double)0.20f : 0.5;
  }
  // TYPE 3 :316-353
  if (beh == 3) {
    const double daytime = P.ix2 / 86400.0;
    const double dtime = (daytime - trunc(daytime)) * 24.0;
    double E0 = 0.0;
    if (dtime > P.twistart && dtime < P.twiend) {
      const double tst = (dtime - P.twistart) * 3600.0;
      E0 = P.Em * lt_sin_g(P.PI * tst / (P.daylength * 3600.0)) * lt_sin_g(P.PI * tst / (P.daylength * 3600.0));
    }
    const double P_light = E0 * lt_exp(P.Kd * P_zc);
    if (P_light < P.thresh) { random = true; thr2 = 0.5; }
    if (P_light > P.thresh) { random = true; thr2 = (double)0.20f; }
  }
  // TYPES 4 and 5 :357-461: the salinity-gradient cue
  if (beh == 4 || (beh == 5 && tim == 0.0)) {
    double Sprev = P.P_Sprev[n], zprev = P.P_zprev[n];
    if (P.it == 1) { Sprev = P_S; zprev = P_zc; }
    double Sslope = 0.0;
    const double deltaS = Sprev - P_S, deltaz = zprev - P_zc;
    if (P.it > 1) Sslope = deltaS / deltaz;
    bool btest = false;
    if (fabs(Sslope) > P.Sgradient) {
      directed = true;
      btest = true;
      if (beh == 4) {
        thr1 = (double)0.80f;
      } else {
        thr1 = (double)0.20f;
        tim = 2.0 * 3600.0;
        if (P_age < (double)(3.5f * 24.f * 3600.f)) { btest = false; tim = 0.; }
        P.timer[n] = tim;
      }
    }
    if (!btest) {
      random = true;
      double sw;
      if (beh == 4) {
        if (P_age < (double)(1.5f * 24.f * 3600.f)) sw = (double)0.1f;
        else if (P_age < (double)(5.f * 24.f * 3600.f)) sw = (double)0.49f;
        else if (P_age < (double)(8.f * 24.f * 3600.f)) sw = (double)0.50f;
        else {
          const double switchslope = ((double)0.50f - (double)0.517f) / (8.0 * 24.0 * 3600.0 - P.P_pediage[n]);
          sw = switchslope * P_age + (double)0.50f - switchslope * 8.0 * 24.0 * 3600.0;
          if (P_zc < P_depth + 1.) sw = 0.5;
        }
        thr2 = 1 - sw;  // :393 dev1 .GT. (1-switch)
      } else {
        sw = (double)0.495f;
        if (P_age < (double)(1.5f * 24.f * 3600.f)) sw = (double)0.9f;
        if (P_age > (double)(2.0f * 24.f * 3600.f) && P_age < (double)(3.5f * 24.f * 3600.f)) {
          const double switchslope =
              ((double)0.3f - (double)0.495f) / (2.0 * 24.0 * 3600.0 - (double)3.5f * 24.0 * 3600.0);
          sw = switchslope * P_age + (double)0.3f - switchslope * 2.0 * 24.0 * 3600.0;
        }
        thr2 = sw;
      }
    }
    P.P_Sprev[n] = P_S;  // :404-405, :461-462
    P.P_zprev[n] = P_zc;
  }
  // TYPE 7 :497-500 first iteration (the rest needs the RK4 currents: k_post)
  if (beh == 7 && P.it == 1) P.P_Sprev[n] = P_S;
  wsd[(size_t)WS_PS * N] = P_S;
  wsd[(size_t)WS_BTHR1 * N] = thr1;
  wsd[(size_t)WS_BTHR2 * N] = thr2;
  fl |= (directed ? FL_BEH_DIRECTED : 0) | (random ? FL_BEH_RANDOM : 0);
  P.ws_i[n] = fl;
}

#ifndef LT_RK_BLOCK
#define LT_RK_BLOCK 64   // one wave per block: a finished wave frees its slot at once (256-lane blocks: +3 % kernel time)
#endif
template <typename SlabT, int VT, int WAVES_PER_SIMD, bool FS = false>
__global__ __launch_bounds__(LT_RK_BLOCK) __attribute__((amdgpu_waves_per_eu(WAVES_PER_SIMD, WAVES_PER_SIMD)))
void k_rk(const StepParams P) {
  __shared__ double lev[6 * LT_LEV_STRIDE];
  __shared__ WaveScratch wsc[LT_RK_BLOCK / 64];
  // XCD-aware order: workgroups go round-robin to the 8 XCDs (each with its own L2), so workgroup b takes
  // the chunk  (b mod 8) * (grid / 8) + b / 8  of the particles: every XCD works through ONE contiguous
  // eighth of the (element-sorted) particles, and the waves in flight on it share the node columns of
  // a few neighbouring rows of elements in its L2.  (The grid is a multiple of 8 workgroups.)
#if defined(LT_RK_XCD_CHUNK) && LT_RK_XCD_CHUNK > 0
  // (the same, in chunks: XCD x takes the chunks x, x+8, x+16 ... of LT_RK_XCD_CHUNK workgroups each, so that a
  // region of the domain whose particles are more work than the others' is shared by all eight; the grid is a
  // multiple of 8 * LT_RK_XCD_CHUNK workgroups, the ones beyond the particles leave at once)
  const unsigned xs_ = blockIdx.x >> 3;
  const unsigned vb = ((xs_ / LT_RK_XCD_CHUNK) * 8u + (blockIdx.x & 7u)) * LT_RK_XCD_CHUNK + (xs_ % LT_RK_XCD_CHUNK);
#elif !defined(LT_RK_NO_XCD_MAP)
  const unsigned vb = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
#else
  const unsigned vb = blockIdx.x;
#endif
#ifdef LT_RK_WAVETIME
  const unsigned long long wt0_ = wall_clock64(), wc0_ = __builtin_readcyclecounter();
#endif
  const int n = (int)(vb * blockDim.x + threadIdx.x);
  const size_t N = (size_t)P.numpar;
  // The particle's state is asked for before the tables are staged and without waiting for its flag word (a lane
  // beyond the particles, or one that is not tracked, reads particle 0's / its own stale values and selects them
  // away): one round trip for flag, position, depth and elements beside the tables' instead of three in a row.
  const int nn = (n < P.numpar) ? n : 0;
  const int fl0 = P.ws_i[nn];
  const double* wsd = P.ws_d + nn;
  const double h0 = wsd[(size_t)WS_PDEPTH * N], X0 = P.par[(size_t)pX * N + nn], Y0 = P.par[(size_t)pY * N + nn];
  const double Z0 = wsd[(size_t)WS_ZPAR * N];
  const int er0 = P.P_r_element[nn], eu0 = P.P_u_element[nn], ev0 = P.P_v_element[nn];
  for (int i = threadIdx.x; i < 6 * LT_LEV_STRIDE; i += blockDim.x) lev[i] = P.lev[i];
  exp_table_load();
  const bool act = (n < P.numpar) && ((fl0 & FL_ACT) != 0);
  if (LT_RK_BLOCK == 64 && !__ballot(act)) return;   // (a wave beyond the particles, or of 64 that are not tracked)
  double h = 1.0, Xs = 0.0, Ys = 0.0, Zs = 0.0;
  if (act) {
    h = -1.0 * h0;
    Xs = X0;
    Ys = Y0;
    Zs = Z0;
  }
  // RK4 :990-1053, executed by the whole wave
  double sumU = 0.0, sumV = 0.0, sumW = 0.0;
  int sigerr_count = 0;
  bool slow = false;
  WaveScratch& ws = wsc[threadIdx.x >> 6];
#ifdef LT_RK_TIMING
  if ((threadIdx.x & 63) < 16) ws.tim[threadIdx.x & 63] = 0;
#endif
  const Divisor dv = prep_div((VT == 2) ? P.hc + h : h);  // every level depth divides by it
  // the RK stages keep the elements of the start position (LTRANS.f90:990-1053)
  int er = 1, eu = 1, ev_ = 1;
  if (act) { er = er0; eu = eu0; ev_ = ev0; }
  // free slip: an element with a land node has its values substituted before every interpolation
  // (:1934-1994, 2329-2520): by the FS instantiation in place (find_currents); the plain one, should it be launched
  // for such a run, hands those particles to the straight transcription of k_rk_slow
  bool act_fast = act;
  unsigned fsbits = 0xffffu;
  if (FS) {
    if (act) fsbits = (unsigned)P.fsU[eu - 1] | ((unsigned)P.fsV[ev_ - 1] << 4) | ((unsigned)P.fsVU[eu - 1] << 8) | ((unsigned)P.fsR[er - 1] << 12);
  } else if (P.FreeSlip && act && ((P.fsR[er - 1] & P.fsU[eu - 1] & P.fsV[ev_ - 1]) != 15u)) {
    slow = true;
    act_fast = false;
  }
  StageIn SI;
  StageTail ST;
  SI.zb = wsd[(size_t)WS_ZETA0 * N]; SI.zc = wsd[(size_t)WS_ZETA1 * N]; SI.zf = wsd[(size_t)WS_ZETA2 * N];
  SI.nd0 = P.UE[eu - 1];
  SI.nd1 = P.VE[ev_ - 1];
  SI.ii = 1; SI.iii = 1;
  ST.X = ST.Y = ST.Zpar = ST.cosA = ST.sinA = ST.minpd = ST.maxpd = 0.0;
#pragma unroll 1
  for (int st = 0; st < 4; st++) {
    const int version = (st == 0) ? 1 : (st == 3) ? 3 : 2;
    double Uad, Vad, Wad;
    // (h made opaque per stage: otherwise every h-dependent level coefficient is hoisted
    // out of the stage loop and held in registers for the whole kernel)
    asm volatile("" : "+v"(h));
    find_currents<SlabT, VT, FS>(P, lev, ws, act_fast, n, er, eu, ev_, h, dv, Xs, Ys, Zs, version, Uad, Vad, Wad, sigerr_count, slow,
                                 SI, ST, st == 0, st == 3, fsbits);
    if (st == 0) { sumU = Uad; sumV = Vad; sumW = Wad; }
    else if (st == 3) { sumU = sumU + Uad; sumV = sumV + Vad; sumW = sumW + Wad; }
    else { sumU = sumU + 2.0 * Uad; sumV = sumV + 2.0 * Vad; sumW = sumW + 2.0 * Wad; }
    if (st < 3) {
      // next RK position :999-1035 (the per-particle constants were requested before the last round of
      // the stage); maxpartdepth / minpartdepth :981-987 come from k_pre
      const double dtf = (double)P.idt;
      if (st < 2) {
        Xs = ST.X + (Uad * ST.cosA - Vad * ST.sinA) * dtf / 2.0;
        Ys = ST.Y + (Uad * ST.sinA + Vad * ST.cosA) * dtf / 2.0;
        Zs = ST.Zpar + Wad * dtf / 2.0;
      } else {
        Xs = ST.X + (Uad * ST.cosA - Vad * ST.sinA) * dtf;
        Ys = ST.Y + (Uad * ST.sinA + Vad * ST.cosA) * dtf;
        Zs = ST.Zpar + Wad * dtf;
      }
      if (Zs > ST.minpd) Zs = ST.minpd - (double)0.000001f;
      if (Zs < ST.maxpd) Zs = ST.maxpd + (double)0.000001f;
    }
  }
#if defined(LT_RK_TIMING) && !defined(LT_VT_EXCURSION)
  if ((threadIdx.x & 63) < 16) atomicAdd(&g_tim[threadIdx.x & 63], ws.tim[threadIdx.x & 63]);
#endif
  if (act) {
    double* wso = P.ws_d + n;
    wso[(size_t)WS_SUMU * N] = sumU;
    wso[(size_t)WS_SUMV * N] = sumV;
    wso[(size_t)WS_SUMW * N] = sumW;
  }
  {  // the particle-steps k_rk_slow redoes: listed (one atomic per wave that has any), so that the kernel behind
     // this one is a few workgroups over the list instead of a pass over every particle's flag word
    const unsigned long long ms = __ballot(act && slow);
    if (ms) {
      const int lane = threadIdx.x & 63, leader = __ffsll((long long)ms) - 1;
      unsigned long long base = 0;
      if (lane == leader) base = atomicAdd(&P.counters[CNT_SLOW], (unsigned long long)__popcll(ms));
      base = __shfl(base, leader, 64);
      if (act && slow) P.slow_list[base + __popcll(ms & ((1ull << lane) - 1ull))] = n;
    }
  }
  unsigned v = (unsigned)sigerr_count;
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0 && v) atomicAdd(&P.counters[CNT_SIGERR], (unsigned long long)v);
#ifdef LT_RK_WAVETIME
  if (threadIdx.x == 0 && vb < 20000u) {
    g_wavetime[2 * vb] = wt0_; g_wavetime[2 * vb + 1] = wall_clock64();
    g_wavecyc[vb] = __builtin_readcyclecounter() - wc0_;
  }
#endif
}

// ---- k_rk_slow: the particle-steps k_rk could not finish on its fast path
// (a SigErr in some spline fit, or a degenerate level window) are redone here
// by a straight per-lane transcription of find_currents / WCTS_ITPI
// (LTRANS.f90:1422-1614, hydrodynamic_module.f90:2577-2689): YPC1, the SIGS
// interval loop with its early RETURN, HVAL or the linint fallback.
template <typename SlabT>
LT_DEV void find_currents_ref(const StepParams& P, const double* lev, const Column& C, int4 rn,
                              int4 un, int4 vn, double Xs, double Ys, double Zs, int version,
                              double& Uad, double& Vad, double& Wad, unsigned fsr, unsigned fsu,
                              unsigned fsv, unsigned fsvu) {
  const double* rtab = lev;
  const double* wtab = lev + 3 * LT_LEV_STRIDE;
  const double ixv = (version == 1) ? P.ix0 : (version == 2) ? P.ix1 : P.ix2;
  if ((Zs < C.wz1[0] + P.z0) || (Zs < C.wz1[1] + P.z0) || (Zs < C.wz1[2] + P.z0)) {
    Uad = 0.0; Vad = 0.0; Wad = 0.0;
    return;
  }
  const bool loglayer = (Zs < C.z1[0]) || (Zs < C.z1[1]) || (Zs < C.z1[2]);
  const int ii = level_window(rtab, P.us, Zs, C.zeta[0], C.zeta[1], C.zeta[2], C.h, P.hc, P.Vtransform);
  const int iii = level_window(wtab, P.ws, Zs, C.zeta[0], C.zeta[1], C.zeta[2], C.h, P.hc, P.Vtransform);
#pragma unroll 1
  for (int var = 0; var < 3; var++) {
    const int4 nd = (var == 0) ? un : (var == 1) ? vn : rn;
    Quad q = (var == 2) ? load_quad_rho(P.rnode, nd)
                        : load_quad_xy((var == 0) ? P.unode : P.vnode, nd);
    const QuadWeights w = quad_weights(Xs, Ys, q.x, q.y);
    const SlabT* F = (const SlabT*)((var == 0) ? P.U : (var == 1) ? P.V : P.W);
    const int L = (var == 2) ? P.ws : P.us;
    const double* tab = (var == 2) ? wtab : rtab;
    const unsigned fsm = (var == 0) ? fsu : (var == 1) ? fsv : fsr;  // free slip: water nodes of the element
    const int fskind = (var == 0) ? 1 : (var == 1) ? 2 : 0;
    double Pv0 = 0.0, Pv1 = 0.0, Pv2 = 0.0;
#pragma unroll 1
    for (int s = 0; s < 3; s++) {
      const int ps = (s == 0) ? P.tb : (s == 1) ? P.tc : P.tf;
      const double zeta = (s == 0) ? C.zeta[0] : (s == 1) ? C.zeta[1] : C.zeta[2];
      double r;
      if (loglayer) {
        const int k = (var == 2) ? 2 : 1;
        double f1 = fldv(F, nd.x, L, k, ps), f2 = fldv(F, nd.y, L, k, ps), f3 = fldv(F, nd.z, L, k, ps),
               f4 = fldv(F, nd.w, L, k, ps);
        if (fsm != 15u) freeslip_sub(fskind, fsm, fsvu, f1, f2, f3, f4);
        double val = apply_interp(w, f1, f2, f3, f4);
        const double z1s = (s == 0) ? C.z1[0] : (s == 1) ? C.z1[1] : C.z1[2];
        double ztop = (var == 2) ? level_z(wtab, 2, zeta, C.h, P.hc, P.Vtransform) : z1s;
        r = val * log10_call((Zs - C.wz1[0]) / P.z0) / log10_call((ztop - C.wz1[0]) / P.z0);
      } else {
        const int k0 = (var == 2) ? iii : ii;
        const double Pzs = (s == 0) ? C.Pz[0] : (s == 1) ? C.Pz[1] : C.Pz[2];
        double az[4], av[4];
#pragma unroll
        for (int l = 0; l < 4; l++) {
          az[l] = level_z(tab, k0 + l, zeta, C.h, P.hc, P.Vtransform);
          double f1 = fldv(F, nd.x, L, k0 + l, ps), f2 = fldv(F, nd.y, L, k0 + l, ps),
                 f3 = fldv(F, nd.z, L, k0 + l, ps), f4 = fldv(F, nd.w, L, k0 + l, ps);
          if (fsm != 15u) freeslip_sub(fskind, fsm, fsvu, f1, f2, f3, f4);
          av[l] = apply_interp(w, f1, f2, f3, f4);
        }
        int dummy = 0;
        r = spline4_eval(az, av, Pzs, dummy);
      }
      if (s == 0) Pv0 = r; else if (s == 1) Pv1 = r; else Pv2 = r;
    }
    double e1 = Pv0, e2 = (P.p == 1) ? Pv0 : Pv1, e3 = (P.p == 1) ? Pv1 : Pv2;
    double val = polintd3(P.ex0, P.ex1, P.ex2, e1, e2, e3, ixv);
    if (var == 0) Uad = val; else if (var == 1) Vad = val; else Wad = val;
  }
}

template <typename SlabT>
__global__ __launch_bounds__(256) void k_rk_slow(const StepParams P) {
  __shared__ double lev[6 * LT_LEV_STRIDE];
  // k_rk's list of the particle-steps to redo; a fixed, small grid strides over it (usually a few hundred
  // entries or none: the workgroups beyond the list leave before staging the tables)
  const long long nslow = (long long)P.counters[CNT_SLOW];
  if ((long long)blockIdx.x * blockDim.x >= nslow) return;
  for (int i = threadIdx.x; i < 6 * LT_LEV_STRIDE; i += blockDim.x) lev[i] = P.lev[i];
  exp_table_load();
  const size_t N = (size_t)P.numpar;
#pragma unroll 1
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < nslow; t += (long long)gridDim.x * blockDim.x) {
  const int n = P.slow_list[t];
  double* wsd = P.ws_d + n;
  const double X = P.par[(size_t)pX * N + n], Y = P.par[(size_t)pY * N + n];
  const double Zpar = wsd[(size_t)WS_ZPAR * N];
  const double cosA = wsd[(size_t)WS_COSA * N], sinA = wsd[(size_t)WS_SINA * N];
  const int4 rn = P.RE[P.P_r_element[n] - 1], un = P.UE[P.P_u_element[n] - 1],
             vn = P.VE[P.P_v_element[n] - 1];
  unsigned fsr = 15u, fsu = 15u, fsv = 15u, fsvu = 15u;
  if (P.FreeSlip) {
    fsr = P.fsR[P.P_r_element[n] - 1]; fsu = P.fsU[P.P_u_element[n] - 1];
    fsv = P.fsV[P.P_v_element[n] - 1]; fsvu = P.fsVU[P.P_u_element[n] - 1];
  }
  Column C;
  C.h = -1.0 * wsd[(size_t)WS_PDEPTH * N];
#pragma unroll
  for (int s = 0; s < 3; s++) {
    C.zeta[s] = wsd[(size_t)(WS_ZETA0 + s) * N];
    C.Pz[s] = wsd[(size_t)(WS_PZ0 + s) * N];
    C.wz1[s] = level_z(lev + 3 * LT_LEV_STRIDE, 1, C.zeta[s], C.h, P.hc, P.Vtransform);
    C.z1[s] = level_z(lev, 1, C.zeta[s], C.h, P.hc, P.Vtransform);
  }
  double maxpartdepth = C.wz1[0];
  if (C.wz1[1] > maxpartdepth) maxpartdepth = C.wz1[1];
  if (C.wz1[2] > maxpartdepth) maxpartdepth = C.wz1[2];
  double minpartdepth = level_z(lev + 3 * LT_LEV_STRIDE, P.ws, C.zeta[0], C.h, P.hc, P.Vtransform);
  {
    double t1 = level_z(lev + 3 * LT_LEV_STRIDE, P.ws, C.zeta[1], C.h, P.hc, P.Vtransform);
    double t2 = level_z(lev + 3 * LT_LEV_STRIDE, P.ws, C.zeta[2], C.h, P.hc, P.Vtransform);
    if (t1 < minpartdepth) minpartdepth = t1;
    if (t2 < minpartdepth) minpartdepth = t2;
  }
  double sumU = 0.0, sumV = 0.0, sumW = 0.0, Xs = X, Ys = Y, Zs = Zpar;
#pragma unroll 1
  for (int st = 0; st < 4; st++) {
    const int version = (st == 0) ? 1 : (st == 3) ? 3 : 2;
    double Uad, Vad, Wad;
    find_currents_ref<SlabT>(P, lev, C, rn, un, vn, Xs, Ys, Zs, version, Uad, Vad, Wad, fsr, fsu, fsv, fsvu);
    if (st == 0) { sumU = Uad; sumV = Vad; sumW = Wad; }
    else if (st == 3) { sumU = sumU + Uad; sumV = sumV + Vad; sumW = sumW + Wad; }
    else { sumU = sumU + 2.0 * Uad; sumV = sumV + 2.0 * Vad; sumW = sumW + 2.0 * Wad; }
    if (st < 3) {
      const double dtf = (double)P.idt;
      if (st < 2) {
        Xs = X + (Uad * cosA - Vad * sinA) * dtf / 2.0;
        Ys = Y + (Uad * sinA + Vad * cosA) * dtf / 2.0;
        Zs = Zpar + Wad * dtf / 2.0;
      } else {
        Xs = X + (Uad * cosA - Vad * sinA) * dtf;
        Ys = Y + (Uad * sinA + Vad * cosA) * dtf;
        Zs = Zpar + Wad * dtf;
      }
      if (Zs > minpartdepth) Zs = minpartdepth - (double)0.000001f;
      if (Zs < maxpartdepth) Zs = maxpartdepth + (double)0.000001f;
    }
  }
  wsd[(size_t)WS_SUMU * N] = sumU;
  wsd[(size_t)WS_SUMV * N] = sumV;
  wsd[(size_t)WS_SUMW * N] = sumW;
  }
}
