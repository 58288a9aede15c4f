// lt_post.h -- k_post: behaviour, new position, boundaries, closing setEle, settlement, commit.
// Part of lt_kernels.h (included inside namespace ltd); see the notes there.
#pragma once

#ifdef LT_POST_WAVES
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LT_POST_WAVES, LT_POST_WAVES))) void k_post(const StepParams P) {
#else
__global__ __launch_bounds__(256) void k_post(const StepParams P) {
#endif
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned c_err = 0, c_brute = 0, c_pipslow = 0;
#if defined(LT_POST_TIMING) && defined(LT_RK_TIMING)
  // development aid (tools/dev/post_sections.py): wave cycles per part of the kernel, summed into g_tim[0..6]
  unsigned long long pt_acc[7] = {0, 0, 0, 0, 0, 0, 0}, pt_prev = __builtin_readcyclecounter();
#define PTICK(sec) do { const unsigned long long t_ = __builtin_readcyclecounter(); pt_acc[sec] += t_ - pt_prev; pt_prev = t_; } while (0)
#else
#define PTICK(sec) do { } while (0)
#endif
  __shared__ EleScratch esc[4];
  __shared__ double s_hamp;   // HTurb's amplitude (2/r*KM*idt)**0.5 -- the same for every particle: libm's pow once per workgroup
  if (P.HTurbOn) {
    if (threadIdx.x == 0) s_hamp = lt_pow_half_g(2.0 / 1.0 * P.ConstantHTurb * P.idt);
    __syncthreads();
  }
  // Part 1 (per lane): behaviour, new position, boundaries.  Part 2 (whole wave): setEle on
  // the new position.  Part 3 (per lane): settlement, error handling, commit.
  const bool inrange = n < P.numpar;
  const size_t N = (size_t)P.numpar;
  double* par = P.par + (inrange ? n : 0);
  const double* wsd = P.ws_d + (inrange ? n : 0);
#define PARV(k) par[(size_t)(k) * N]
  const int fl = inrange ? P.ws_i[n] : 0;
  const bool act = (fl & FL_ACT) != 0;
  bool keep_pn = (fl & FL_KEEP_PN) != 0;
  int err = fl >> FL_ERR_SHIFT;
  double X = 0.0, Y = 0.0, Z = 0.0;
  if (inrange) { X = PARV(pX); Y = PARV(pY); Z = act ? wsd[(size_t)WS_ZPAR * N] : PARV(pZ); }
  double nX = X, nY = Y, nZ = Z, lifespan = 0.0, age = 0.0, P_depth = 0.0;
  bool set_lifespan = false, verified = false, want_ele = false;
  double eX = 0.0, eY = 0.0;
  // the stored elements and their node ids are what the closing setEle probes first: asked
  // for now, they arrive while part 1 runs
  int er = 1, eu = 1, ev = 1;
  int4 rn0 = make_int4(1, 1, 1, 1), un0 = rn0, vn0 = rn0;
  if (act) {
    er = P.P_r_element[n]; eu = P.P_u_element[n]; ev = P.P_v_element[n];
    rn0 = P.RE[er - 1]; un0 = P.UE[eu - 1]; vn0 = P.VE[ev - 1];
  }
  {
    if (act) {
      age = PARV(pAge);
      P_depth = wsd[(size_t)WS_PDEPTH * N];
      const double P_zetac = wsd[(size_t)WS_ZETA1 * N];
      const double cosA = wsd[(size_t)WS_COSA * N], sinA = wsd[(size_t)WS_SINA * N];
      const double P_U = wsd[(size_t)WS_SUMU * N] / 6.0, P_V = wsd[(size_t)WS_SUMV * N] / 6.0,
                   P_W = wsd[(size_t)WS_SUMW * N] / 6.0;
      const double AdvectX = P.idt * (P_U * cosA - P_V * sinA);
      const double AdvectY = P.idt * (P_U * sinA + P_V * cosA);
      const double AdvectZ = P.idt * P_W;
      // HTurb, hor_turb_module.f90:29-50: X deviate before Y; VTurb's TurbV from k_vt_walk
      double TurbHx = 0.0, TurbHy = 0.0, TurbV = 0.0;
      if (P.HTurbOn) {
        double devX, devY;
        if (P.VTurbOn) {   // formed by k_vt_norm with the walk's deviates (the tables are in LDS there)
          const double* hrow = VT_DEV_BASE(P, n, P.idt / 2);
          devX = hrow[0]; devY = hrow[64];
        } else {
          const unsigned long long di = P.rng_base + (unsigned long long)(unsigned)P.rank[P.orig[n]];
          devX = draw_norm(P, di); devY = draw_norm(P, di + 2);
        }
        const double amp = s_hamp;
        TurbHx = devX * amp;
        TurbHy = devY * amp;
      }
      if (P.VTurbOn) TurbV = wsd[(size_t)WS_TURBV * N];
      // behave, behavior_module.f90:181-551 (the part that needs the deviates and the currents)
      double ZBehav = 0.0, XBehav = 0.0, YBehav = 0.0;
      bool bott = false;
      if (P.Behavior != 0) {
        const size_t np = (size_t)P.numpar;
        double sw3 = P.P_swim[n + 2 * np];
        bool sw_dirty = false;
        if (age >= P.swimstart) { sw3 = P.P_swim[n] * age + P.P_swim[n + np]; sw_dirty = true; }
        if (age >= P.P_pediage[n]) { sw3 = P.swimfast; sw_dirty = true; }
        if (sw_dirty) P.P_swim[n + 2 * np] = sw3;
        double parBehav = 0.0;
        const int beh = P.P_behave[n];
        // types 1-5: the deviates k_beh_pre announced, after this particle's turbulence draws
        if (fl & (FL_BEH_DIRECTED | FL_BEH_RANDOM)) {
          unsigned long long db = P.rng_base + (unsigned long long)(unsigned)P.rank[P.orig[n]] +
                                  (unsigned long long)P.rng_draws;
          if (fl & FL_BEH_DIRECTED) {  // :371-377, :425-437
            const double dev1 = draw_real1(P, db);
            db += 1;
            const double negpos = (dev1 > wsd[(size_t)WS_BTHR1 * N]) ? -1.0 : 1.0;
            parBehav = negpos * sw3;
          }
          if (fl & FL_BEH_RANDOM) {
            const double dev1 = draw_real1(P, db), devB = draw_real1(P, db + 1);
            const double negpos = (dev1 > wsd[(size_t)WS_BTHR2 * N]) ? -1.0 : 1.0;
            parBehav = negpos * devB * sw3;
          }
        }
        if (beh == 6) parBehav = (age >= P.swimstart) ? P.sink : sw3;
        ZBehav = parBehav * P.idt;
        // TYPE 7 :495-547 tidal stream transport
        if (beh == 7) {
          const double P_S = wsd[(size_t)WS_PS * N];
          const double cu = P_U * cosA - P_V * sinA, cv = P_U * sinA + P_V * cosA;
          const double currentspeed = sqrt(cu * cu + cv * cv);
          int onb = P.bottom[n];
          if (onb) {
            if (P.P_Sprev[n] < P_S) {
              onb = 0;
              ZBehav = P_depth + P.Swimdepth;
            } else {
              ZBehav = -9999;
            }
          } else {
            if (currentspeed > (double)0.05f) {
              const double Hdistance = P.Hswimspeed * P.idt;
              const double theta = lt_atan_g(cv / cu);
              double sth, cth;
              lt_sincos_g(theta, &sth, &cth);
              if (cu > 0.0) { XBehav = Hdistance * cth; YBehav = Hdistance * sth; }
              if (cu < 0.0) { XBehav = -1.0 * Hdistance * cth; YBehav = -1.0 * Hdistance * sth; }
              if (cu == 0 && cv >= 0.0) { XBehav = 0.0; YBehav = Hdistance; }
              if (cu == 0 && cv <= 0.0) { XBehav = 0.0; YBehav = -1.0 * Hdistance; }
              ZBehav = P_depth + P.Swimdepth;
            } else {
              ZBehav = -9999;
              onb = 1;
            }
          }
          P.bottom[n] = onb;
          bott = onb != 0;
        }
      }
      // new position, vertical reflection :1128-1176
      double newXpos = X + AdvectX + TurbHx;
      double newYpos = Y + AdvectY + TurbHy;
      double newZpos = Z + AdvectZ + TurbV;
      if (newZpos > P_zetac) {
        double reflect = P_zetac - newZpos;
        newZpos = P_zetac + reflect;
      }
      if (newZpos < P_depth) {
        double reflect = P_depth - newZpos;
        newZpos = P_depth + reflect;
        if (P.TrackCollisions) P.hitBottom[n] += 1;
      }
      newZpos = newZpos + ZBehav;
      if (P.Behavior == 7) {  // :1148-1164
        if (bott) {
          newXpos = X; newYpos = Y; newZpos = P_depth;
        } else {
          newXpos = newXpos + XBehav; newYpos = newYpos + YBehav; newZpos = P_depth + P.Swimdepth;
        }
      }
      if (newZpos > P_zetac) newZpos = P_zetac - (double)0.000001f;
      if (newZpos < P_depth) {
        newZpos = P_depth + (double)0.000001f;
        if (P.TrackCollisions) P.hitBottom[n] += 1;
      }
      PTICK(0);
      // horizontal boundaries :1180-1232
      double Xpos = X, Ypos = Y, nXpos = newXpos, nYpos = newYpos;
      int skipbound = -1, reflects = 0;
      bool done = false;  // waterFlag
      for (;;) {
        Hit H;
        intersect_reflect(P, Xpos, Ypos, nXpos, nYpos, skipbound, H, c_brute);
        skipbound = H.seg;
        if (!H.found) break;
        if (P.TrackCollisions) P.hitLand[n] += 1;
        if (P.OpenOceanBoundary && H.water) {
          nX = H.fix; nY = H.fiy; nZ = newZpos;
          P.oob[n] = 1;
          done = true;
          break;
        }
        reflects += 1;
        if (reflects > 3) {
          err = 24;
          done = true;
          break;
        }
        Xpos = H.fix; Ypos = H.fiy; nXpos = H.frx; nYpos = H.fry;
      }
      PTICK(1);
      if (!done) {
        newXpos = nXpos;
        newYpos = nYpos;
        // mbounds :1240, ibounds :1275
        bool slow = false, inb;
        if (P.use_pip_index) {
          inb = mbounds_indexed(P, newXpos, newYpos, slow);
        } else {
          slow = true;
          inb = false;
        }
        if (slow) {
          c_pipslow += 1;
          inb = inpoly_full(newXpos, newYpos, P.maxbound, P.bx, P.by, false);
        }
        PTICK(2);
        if (!inb) {
          err = 25;
        } else {
          bool inisl = false;
          if (P.numislands > 0) {
            slow = false;
            if (P.use_pip_index) inisl = ibounds_indexed(P, newXpos, newYpos, slow);
            else slow = true;
            if (slow) {
              c_pipslow += 1;
              inisl = ibounds_full(P.hx, P.hy, P.hid, P.maxisland, P.numislands, newXpos, newYpos);
            }
          }
          if (inisl) {
            err = 26;
          } else {
            nX = newXpos; nY = newYpos; nZ = newZpos;  // :1312-1314
            want_ele = true;  // setEle on the new position :1317, by the whole wave below
            eX = nXpos; eY = nYpos;
          }
        }
      }
    }
  }
  PTICK(3);
  // Part 2: setEle :1317
  bool okr2, oku2, okv2;
  wave_set_ele(P, esc[threadIdx.x >> 6], want_ele, eX, eY, er, eu, ev, rn0, un0, vn0, okr2, oku2, okv2);
  PTICK(4);
  // Part 3
  if (inrange) {
    if (want_ele) {
      P.P_r_element[n] = er; P.P_u_element[n] = eu; P.P_v_element[n] = ev;
      if (!okv2) err = 29; else if (!oku2) err = 28; else if (!okr2) err = 27;
      verified = (err == 0);  // the committed position lies in the stored elements
      if (err == 0 && P.settlementon) {
        // testSettlement with the OLD position and the NEW element :1375
        if (age >= P.settletime[n]) {
          int inpoly = test_settlement(P, X, Y, er);
          if (inpoly > 0) {
            P.settle[n] = 1;
            nZ = P_depth;
            P.endpoly[n] = inpoly;
            lifespan = age;
            set_lifespan = true;
          }
        }
      }
    }
    // ErrorFlag handling :858-877 etc.  (ErrorFlag 0 = the host stops; the
    // particle is returned to its previous location as under ErrorFlag 1)
    if (err != 0) {
      c_err = 1;
      // An error of the closing setEle (27-29) is raised AFTER par(n,pnX..pnZ) took the new position
      // (:1312-1317): die() / setOut() leave it there and the commit loop moves the particle to it.  The
      // earlier errors (4-6, 24-26) `cycle` before that assignment: pn keeps what the last step left.
      const bool pn_assigned = (err >= 27);
      if (P.ErrorFlag == 2) { P.dead[n] = 1; keep_pn = !pn_assigned; }
      else if (P.ErrorFlag == 3) { P.oob[n] = 1; keep_pn = !pn_assigned; }
      else { nX = X; nY = Y; nZ = Z; keep_pn = false; }
    }
    P.err_code[n] = err;
    P.ele_ok[n] = verified ? 1 : 0;
    // commit loop :1407-1414 fused: pp <- p, p <- pn
    if (keep_pn) {
      nX = PARV(pnX); nY = PARV(pnY); nZ = PARV(pnZ);
    } else {
      PARV(pnX) = nX; PARV(pnY) = nY; PARV(pnZ) = nZ;
    }
    PARV(ppX) = X; PARV(ppY) = Y; PARV(ppZ) = Z;
    PARV(pX) = nX; PARV(pY) = nY; PARV(pZ) = nZ;
    if (set_lifespan) PARV(pLifespan) = lifespan;
#undef PARV
  }
  PTICK(5);
#if defined(LT_POST_TIMING) && defined(LT_RK_TIMING)
  if ((threadIdx.x & 63) == 0) {
    for (int q = 0; q < 6; q++) atomicAdd(&g_tim[q], pt_acc[q]);
    atomicAdd(&g_tim[7], 1ull);
  }
#endif
  __shared__ unsigned red[3];
  if (threadIdx.x < 3) red[threadIdx.x] = 0;
  if (blockIdx.x == 0) {  // total of k_pre's per-block counts of tracked particles
    __shared__ unsigned long long tot;
    if (threadIdx.x == 0) tot = 0;
    __syncthreads();
    unsigned long long s = 0;
    for (unsigned b = threadIdx.x; b < gridDim.x; b += blockDim.x) s += P.block_active[b];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&tot, s);
    __syncthreads();
    if (threadIdx.x == 0) {
      P.counters[CNT_ACTIVE] = tot;
      if (P.expect_own >= 0 && (long long)P.rank[P.numpar] != P.expect_own) P.counters[CNT_STATIC_MISS] += 1ull;
    }
  }
  __syncthreads();
  unsigned vals[3] = {c_err, c_brute, c_pipslow};
#pragma unroll
  for (int k = 0; k < 3; k++) {
    unsigned v = vals[k];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&red[k], v);
  }
  __syncthreads();
  const int slot[3] = {CNT_ERRORS, CNT_BRUTE, CNT_PIP_SLOW};
  if (threadIdx.x < 3 && red[threadIdx.x])
    atomicAdd(&P.counters[slot[threadIdx.x]], (unsigned long long)red[threadIdx.x]);
}
