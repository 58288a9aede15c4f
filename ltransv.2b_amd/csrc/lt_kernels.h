// lt_kernels.h -- HIP kernels of the LTRANS hot path for gfx950 (CDNA4).
//
// k_step: one wavefront lane per particle does the whole of
// update_particles() (LTRANS.f90:707-1419) for its particle:
//   gates -> setEle -> setInterp/getInterp -> z clamps -> RK4 over
//   find_currents/WCTS_ITPI -> behaviour -> vertical reflect -> horizontal
//   boundary reflection -> in-bounds tests -> setEle -> settlement -> commit.
// No MFMA: the path is gather + scalar fp64 spline arithmetic.
//
// Shape decisions that differ from the reference (results do not):
//   * interpolation weights are computed once per (stage, grid) instead of at
//     each of the 144 interp() calls (hydrodynamic_module.f90:2527-2563 is a
//     pure function of position and element);
//   * s-level columns (LTRANS.f90:934-951) are never materialised: the level
//     windows are found by bisection on the (monotone) level depths, and the
//     4-level windows are recomputed from hc*SC / CS tables held in LDS;
//   * boundary segments are visited through a uniform-grid index in reference
//     order (ascending segment id) instead of scanning all nbounds; the
//     point-in-polygon tests read a y-slab edge index and fall back to the
//     full crossings algorithm whenever a polygon vertex lies exactly on the
//     particle's ray (the only case where the two differ);
//   * the 36 TSPSI(4)+HVAL fits per particle-step run through ONE loop body
//     (stage x variable x slot) so the instruction footprint stays in the
//     instruction cache.
#pragma once
#include "lt_device.h"
#include "lt_math.h"

namespace ltd {
#if defined(LT_POST_TIMING) && !defined(LT_RK_TIMING)
__device__ unsigned long long g_tim[8];  // development aid
#endif

LT_DEV uint32_t ltmt_temper(uint32_t y) {  // random_module.f90:190-193
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

// par(n,k) column indices (LTRANS.f90:106-118), 0-based
enum { pX = 0, pY, pZ, pnX, pnY, pnZ, ppX, ppY, ppZ, pStatus, pDOB, pAge, pLifespan };

#define LT_LEV_STRIDE 64

// per-step workspace columns (doubles, SoA stride numpar) and the flag word
enum { WS_PDEPTH = 0, WS_COSA, WS_SINA, WS_ZETA0, WS_ZETA1, WS_ZETA2, WS_PZ0, WS_PZ1, WS_PZ2,
       WS_ZPAR, WS_SUMU, WS_SUMV, WS_SUMW, WS_TURBV,
       WS_PS, WS_BTHR1, WS_BTHR2,  // behaviour: salinity at the particle, thresholds of the two draws
       WS_NDOUBLE };
// FL_BEH_DIRECTED / FL_BEH_RANDOM: what k_beh_pre decided the behaviour will draw (one deviate
// compared with WS_BTHR1; then two deviates, the first compared with WS_BTHR2)
enum { FL_ACT = 1, FL_KEEP_PN = 2, FL_SLOW = 4, FL_VT_YPCFAIL = 16, FL_VT_SIGERR = 32,
       FL_BEH_DIRECTED = 64, FL_BEH_RANDOM = 128, FL_ERR_SHIFT = 8 };

template <typename SlabT> struct SlabTraits;
template <> struct SlabTraits<float> { using zeta4 = float4; };
template <> struct SlabTraits<double> { using zeta4 = double4; };

// -------------------------------------------------------------- s-levels
// getSlevel / getWlevel, hydrodynamic_module.f90:2691-2777.  tab points at the
// three tables {hc*SC, CS, SC} of the rho or the w grid in LDS; k is 1-based;
// h = -P_depth > 0.
LT_DEV double level_z(const double* tab, int k, double zeta, double h, double hc, int Vtransform) {
  double A = tab[k - 1];
  double B = tab[LT_LEV_STRIDE + k - 1];
  if (Vtransform == 1) {
    double S = A + (h - hc) * B;
    return S + zeta * (1.0 + S / h);
  } else if (Vtransform == 2) {
    double S = (A + h * B) / (hc + h);
    return zeta + (zeta + h) * S;
  } else {
    double C = tab[2 * LT_LEV_STRIDE + k - 1];
    return zeta * (1.0 + C) + A + (h - hc) * B;
  }
}

// Lowest level i in [3, n-2] with Z below any of the three level depths,
// else n-1: the loops of find_currents LTRANS.f90:1451-1467 (the level depths
// increase with i, so the first hit is found by bisection).  Returns i-2.
LT_DEV int level_window(const double* tab, int n, double Z, double zb, double zc, double zf,
                        double h, double hc, int Vtransform) {
  int lo = 3, hi = n - 1;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    bool below = (Z < level_z(tab, mid, zb, h, hc, Vtransform)) ||
                 (Z < level_z(tab, mid, zc, h, hc, Vtransform)) ||
                 (Z < level_z(tab, mid, zf, h, hc, Vtransform));
    if (below) hi = mid; else lo = mid + 1;
  }
  return lo - 2;
}

// -------------------------------------------------------- element search
struct Quad {
  double x[4], y[4];
};

LT_DEV Quad load_quad_xy(const XYNode* nodes, int4 n) {
  Quad q;
  XYNode a = nodes[n.x - 1], b = nodes[n.y - 1], c = nodes[n.z - 1], d = nodes[n.w - 1];
  q.x[0] = a.x; q.y[0] = a.y; q.x[1] = b.x; q.y[1] = b.y;
  q.x[2] = c.x; q.y[2] = c.y; q.x[3] = d.x; q.y[3] = d.y;
  return q;
}
LT_DEV Quad load_quad_rho(const RhoNode* nodes, int4 n) {
  Quad q;
  q.x[0] = nodes[n.x - 1].x; q.y[0] = nodes[n.x - 1].y;
  q.x[1] = nodes[n.y - 1].x; q.y[1] = nodes[n.y - 1].y;
  q.x[2] = nodes[n.z - 1].x; q.y[2] = nodes[n.z - 1].y;
  q.x[3] = nodes[n.w - 1].x; q.y[3] = nodes[n.w - 1].y;
  return q;
}

// one grid of setEle's neighbour walk, hydrodynamic_module.f90:1462-1475.
// Adjacency rows start with the element itself (checked at upload), so the
// first probe needs no adjacency read.  Returns false when no listed element
// contains the point (the reference then reports error 4/5/6).
template <bool RHO>
LT_DEV bool find_element(const void* nodes, const int4* EL, const int* Adj, double X, double Y,
                         int& ele, int4& nd, int first = 0) {
  int cand = ele;
#pragma unroll 1
  for (int i = first; i < 10; i++) {
    if (i > 0) {
      cand = Adj[(size_t)(ele - 1) * 10 + i];
      if (cand == 0) break;  // zero-filled tail: nothing further to probe
    }
    int4 n = EL[cand - 1];
    Quad q = RHO ? load_quad_rho((const RhoNode*)nodes, n) : load_quad_xy((const XYNode*)nodes, n);
    if (gridcell_one(q.x, q.y, X, Y)) {
      ele = cand;
      nd = n;
      return true;
    }
  }
  nd = EL[ele - 1];
  return false;
}

// The neighbour part of the same walk (entries 2..10 of the adjacency row) with the probes
// of three neighbours in flight at a time: the row, then per chunk the three node-id
// records, then their twelve corner records, tested in the row's order -- the result is
// the reference's (first listed element that contains the point; a zero entry ends the
// list), in at most 1 + 3x2 dependent loads instead of up to 9x3.
template <bool RHO>
LT_DEV bool find_element_neighbours(const void* nodes, const int4* EL, const int* Adj, double X, double Y,
                                    int& ele, int4& nd) {
  const int* row = Adj + (size_t)(ele - 1) * 10;
  int cand[9];
#pragma unroll
  for (int i = 0; i < 9; i++) cand[i] = row[i + 1];
#pragma unroll 1
  for (int c = 0; c < 9; c += 3) {
    int4 n[3];
    Quad q[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int cj = (c == 0) ? cand[j] : (c == 3) ? cand[3 + j] : cand[6 + j];
      n[j] = EL[(cj > 0 ? cj : ele) - 1];
    }
#pragma unroll
    for (int j = 0; j < 3; j++)
      q[j] = RHO ? load_quad_rho((const RhoNode*)nodes, n[j]) : load_quad_xy((const XYNode*)nodes, n[j]);
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int cj = (c == 0) ? cand[j] : (c == 3) ? cand[3 + j] : cand[6 + j];
      if (cj == 0) { nd = EL[ele - 1]; return false; }  // zero-filled tail: nothing further to probe
      if (gridcell_one(q[j].x, q[j].y, X, Y)) {
        ele = cj;
        nd = n[j];
        return true;
      }
    }
  }
  nd = EL[ele - 1];
  return false;
}

// setEle for the three grids (hydrodynamic_module.f90:1462-1510) with the FIRST
// probe of each grid (the particle's current element, almost always the hit)
// issued together: the three dependent load chains element -> node ids -> node
// coordinates overlap instead of running one after the other.  A grid whose
// first probe misses continues with the neighbour walk.
LT_DEV void find_elements3(const StepParams& P, double X, double Y, int& er, int& eu, int& ev,
                           int4& rn, int4& un, int4& vn, bool& okr, bool& oku, bool& okv) {
  rn = P.RE[er - 1];
  un = P.UE[eu - 1];
  vn = P.VE[ev - 1];
  const Quad qr = load_quad_rho(P.rnode, rn);
  const Quad qu = load_quad_xy(P.unode, un);
  const Quad qv = load_quad_xy(P.vnode, vn);
  okr = gridcell_one(qr.x, qr.y, X, Y);
  oku = gridcell_one(qu.x, qu.y, X, Y);
  okv = gridcell_one(qv.x, qv.y, X, Y);
#ifdef LT_POST_TIMING
  {
    const unsigned long long m1 = __ballot(!okr), m2 = __ballot(!oku), m3 = __ballot(!okv);
    if ((threadIdx.x & 63) == 0) {
      atomicAdd(&g_tim[5], (unsigned long long)(__popcll(m1) + __popcll(m2) + __popcll(m3)));
      atomicAdd(&g_tim[6], (unsigned long long)((m1 != 0) + (m2 != 0) + (m3 != 0)));
      atomicAdd(&g_tim[7], 1ull);
    }
  }
#endif
  if (!okr) okr = find_element_neighbours<true>(P.rnode, P.RE, P.rAdj, X, Y, er, rn);
  if (!oku) oku = find_element_neighbours<false>(P.unode, P.UE, P.uAdj, X, Y, eu, un);
  if (!okv) okv = find_element_neighbours<false>(P.vnode, P.VE, P.vAdj, X, Y, ev, vn);
}

// setEle for the three grids, executed by the WHOLE wave.  The first probe of each grid
// (the stored element) is per lane as in find_elements3.  Few lanes need the neighbour
// walk (a particle that changed cell: ~3 % of the lane-grids per step), but almost every
// wave has some, and a per-lane walk makes the wave sit through up to nine dependent
// probes per grid for two or three busy lanes.  Here the walks are TASKS (lane, grid) in
// the wave's LDS scratch and each of 63 lanes takes ONE probe of one task (7 tasks x the
// 9 neighbour entries of the adjacency row per round): three dependent loads and one
// gridcell test for all of them; the first hit in row order wins and a zero entry ends
// the row, as hydrodynamic_module.f90:1462-1475.
struct EleScratch {
  double X[192], Y[192];
  int ele[192];
  unsigned char owner[192], grid[192];
  int found[3][64];
};

LT_DEV void wave_set_ele(const StepParams& P, EleScratch& S, bool want, double X, double Y, int& er,
                         int& eu, int& ev, int4 rn, int4 un, int4 vn, bool& okr, bool& oku, bool& okv) {
  const int lane = threadIdx.x & 63;
  okr = oku = okv = true;
  if (want) {
    const Quad qr = load_quad_rho(P.rnode, rn);
    const Quad qu = load_quad_xy(P.unode, un);
    const Quad qv = load_quad_xy(P.vnode, vn);
    okr = gridcell_one(qr.x, qr.y, X, Y);
    oku = gridcell_one(qu.x, qu.y, X, Y);
    okv = gridcell_one(qv.x, qv.y, X, Y);
  }
  int T = 0;  // wave-uniform
  const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
  for (int g = 0; g < 3; g++) {
    const bool need = !((g == 0) ? okr : (g == 1) ? oku : okv);
    const unsigned long long m = __ballot(need);
    if (need) {
      const int idx = T + __popcll(m & below);
      S.X[idx] = X; S.Y[idx] = Y;
      S.ele[idx] = (g == 0) ? er : (g == 1) ? eu : ev;
      S.owner[idx] = (unsigned char)lane; S.grid[idx] = (unsigned char)g;
      S.found[g][lane] = 0;
    }
    T += __popcll(m);
  }
  if (T == 0) return;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 1
  for (int base = 0; base < T; base += 7) {
    const int grp = lane / 9, pr = lane - grp * 9;  // probe = entry pr+2 of the row (1-based)
    const int t = base + grp;
    const bool valid = lane < 63 && t < T;
    bool hit = false;
    int cand = 0;
    if (valid) {
      const int g = S.grid[t];
      const int* Adj = (g == 0) ? P.rAdj : (g == 1) ? P.uAdj : P.vAdj;
      cand = Adj[(size_t)(S.ele[t] - 1) * 10 + pr + 1];
      if (cand != 0) {
        const int4* EL = (g == 0) ? P.RE : (g == 1) ? P.UE : P.VE;
        const int4 nn = EL[cand - 1];
        const char* nb = (g == 0) ? (const char*)P.rnode : (g == 1) ? (const char*)P.unode : (const char*)P.vnode;
        const size_t st = (g == 0) ? sizeof(RhoNode) : sizeof(XYNode);
        const XYNode a = *(const XYNode*)(nb + (size_t)(nn.x - 1) * st), b = *(const XYNode*)(nb + (size_t)(nn.y - 1) * st),
                     c = *(const XYNode*)(nb + (size_t)(nn.z - 1) * st), d = *(const XYNode*)(nb + (size_t)(nn.w - 1) * st);
        const double qx[4] = {a.x, b.x, c.x, d.x}, qy[4] = {a.y, b.y, c.y, d.y};
        hit = gridcell_one(qx, qy, S.X[t], S.Y[t]);
      }
    }
    const unsigned long long mh = __ballot(hit), mz = __ballot(valid && cand == 0);
    if (valid && hit) {
      const unsigned gh = (unsigned)(mh >> (9 * grp)) & 0x1ffu, gz = (unsigned)(mz >> (9 * grp)) & 0x1ffu;
      const unsigned before = (1u << pr) - 1u;
      // the first hit of the row, and no zero entry (= end of the list) before it
      if ((gh & before) == 0 && (gz & before) == 0) S.found[S.grid[t]][S.owner[t]] = cand;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (!okr) { const int c = S.found[0][lane]; if (c) { er = c; okr = true; } }
  if (!oku) { const int c = S.found[1][lane]; if (c) { eu = c; oku = true; } }
  if (!okv) { const int c = S.found[2][lane]; if (c) { ev = c; okv = true; } }
}

// --------------------------------------------------- point in polygon
// inpoly, point_in_polygon_module.f90:25-167, for a polygon whose points are
// ex[i], ey[i], i = 0..n-1 (closed: last point repeats the first).  The hilo
// array of the reference is recomputed on the fly.
__device__ __noinline__ bool inpoly_full(double x, double y, int n, const double* ex,
                                         const double* ey, bool onout) {
  bool on = false;
  for (int i = 0; i < n; i++) {
    double py = ey[i], px = ex[i];
    if (py == y && px > x) on = true;
    if (px == x && py == y) return !onout;
  }
  int crossed = 0;
#define LT_HILO(i1) ((ey[(i1)-1] > y) ? 1 : ((ey[(i1)-1] < y) ? -1 : 0))
  if (on) {
    bool first = true;
    int i = 1;
    for (;;) {
      if (i > n) break;
      if (LT_HILO(i) == 0 && ex[i - 1] > x) {
        if (first) {
          i = i + 1;
          continue;
        }
        if (LT_HILO(i - 1) == 0) return !onout;
        int j = 1;
        for (;;) {
          if ((i + j) == (n + 1)) j = 2 - i;
          if (LT_HILO(i + j) != 0) break;
          if (ex[i + j - 1] < x) return !onout;
          j = j + 1;
        }
        if ((LT_HILO(i - 1) + LT_HILO(i + j)) == 0) crossed = crossed + 1;
        if (j < 0) break;
        i = i + j;
      }
      first = false;
      i = i + 1;
    }
  }
#undef LT_HILO
  for (int i = 0; i < n - 1; i++) {
    double x1 = ex[i], y1 = ey[i], x2 = ex[i + 1], y2 = ey[i + 1];
    if (x1 <= x && x2 <= x) continue;
    if (y1 <= y && y2 <= y) continue;
    if (y1 >= y && y2 >= y) continue;
    if (x1 > x && x2 > x) {
      crossed = crossed + 1;
      continue;
    }
    double m = (y2 - y1) / (x2 - x1);
    double b = y1 - m * x1;
    double ix = (y - b) / m;
    if (ix == x) return !onout;
    if (ix > x) crossed = crossed + 1;
  }
  return (crossed & 1) != 0;
}

// crossings test of ONE edge (point_in_polygon_module.f90:125-160).
// returns 0 no crossing, 1 crossing, 2 point lies on the edge
LT_DEV int pip_edge(double x, double y, double x1, double y1, double x2, double y2) {
  if (x1 <= x && x2 <= x) return 0;
  if (y1 <= y && y2 <= y) return 0;
  if (y1 >= y && y2 >= y) return 0;
  if (x1 > x && x2 > x) return 1;
  double m = (y2 - y1) / (x2 - x1);
  double b = y1 - m * x1;
  double ix = (y - b) / m;
  if (ix == x) return 2;
  return (ix > x) ? 1 : 0;
}

// mbounds (boundary_module.f90:1515-1540) through the y-slab index; `slow` is
// set when a vertex shares the particle's y exactly (then the full algorithm
// must decide).
LT_DEV bool mbounds_indexed(const StepParams& P, double x, double y, bool& slow) {
  const PipIndex& ix = P.pip_main;
  double fb = floor((y - ix.y0) * ix.inv_h);
  int b = (fb < 0.0) ? 0 : (fb >= (double)ix.nbins ? ix.nbins - 1 : (int)fb);
  int crossed = 0;
  for (int q = ix.ptr[b]; q < ix.ptr[b + 1]; q++) {
    int i = ix.ids[q];
    double x1 = P.bx[i - 1], y1 = P.by[i - 1], x2 = P.bx[i], y2 = P.by[i];
    if (y1 == y || y2 == y) {
      slow = true;
      return false;
    }
    int r = pip_edge(x, y, x1, y1, x2, y2);
    if (r == 2) return true;
    crossed += r;
  }
  return (crossed & 1) != 0;
}

// ibounds (boundary_module.f90:1548-1614): inside ANY island?
LT_DEV bool ibounds_indexed(const StepParams& P, double x, double y, bool& slow) {
  const PipIndex& ix = P.pip_isl;
  double fb = floor((y - ix.y0) * ix.inv_h);
  int b = (fb < 0.0) ? 0 : (fb >= (double)ix.nbins ? ix.nbins - 1 : (int)fb);
  int cur = -1, crossed = 0;
  for (int q = ix.ptr[b]; q < ix.ptr[b + 1]; q++) {
    int i = ix.ids[q];
    int id = P.hid[i - 1];
    if (id != cur) {
      if (crossed & 1) return true;
      cur = id;
      crossed = 0;
    }
    double x1 = P.hx[i - 1], y1 = P.hy[i - 1], x2 = P.hx[i], y2 = P.hy[i];
    if (y1 == y || y2 == y) {
      slow = true;
      return false;
    }
    int r = pip_edge(x, y, x1, y1, x2, y2);
    if (r == 2) return true;
    crossed += r;
  }
  return (crossed & 1) != 0;
}

// ibounds by the reference's own island loop (slow path)
__device__ __noinline__ bool ibounds_full(const double* hx, const double* hy, const int* hid,
                                          int maxisland, int numislands, double x, double y) {
  if (numislands <= 0) return false;
  int i = 1, start = 0;
  int isle = hid[0];
  for (;;) {
    i = i + 1;
    bool endIsle = false;
    if (i == maxisland) endIsle = true;
    else if (hid[i] != isle) endIsle = true;  // hid(i+1)
    if (endIsle) {
      int count = i - start;
      if (inpoly_full(x, y, count, hx + start, hy + start, false)) return true;
      if (i == maxisland) break;
      start = i;
      isle = hid[i];
    }
  }
  return false;
}

// ------------------------------------------------ intersect_reflect
struct Hit {
  double fix, fiy, frx, fry;
  double dtest;
  double rPx, rPy;  // rPxyzX/Y: carried across segments within one call
  int seg;          // skipboundi
  int found;
  int water;
};

// body of the segment loop of intersect_reflect, boundary_module.f90:1664-1898
LT_DEV void segment_test(const StepParams& P, int i, double Xpos, double Ypos, double nXpos,
                         double nYpos, double xlow, double xhigh, double ylow, double yhigh,
                         Hit& H) {
  const double* s = P.bnd + (size_t)(i - 1) * 4;
  double bcx1 = s[0], bcy1 = s[1], bcx2 = s[2], bcy2 = s[3];
  if (((bcx1 > xhigh) && (bcx2 > xhigh)) || ((bcx1 < xlow) && (bcx2 < xlow)) ||
      ((bcy1 > yhigh) && (bcy2 > yhigh)) || ((bcy1 < ylow) && (bcy2 < ylow)))
    return;
  double bxhigh, bxlow, byhigh, bylow;
  if (bcx1 >= bcx2) { bxhigh = bcx1; bxlow = bcx2; } else { bxhigh = bcx2; bxlow = bcx1; }
  if (bcy1 >= bcy2) { byhigh = bcy1; bylow = bcy2; } else { byhigh = bcy2; bylow = bcy1; }
  double intersctx = 0.0, interscty = 0.0, rx1 = 0.0, ry1 = 0.0, rx2 = 0.0, ry2 = 0.0;
  bool candidate = false;  // passed the in-box test: reflection points are defined
  // kind of reflection geometry: 0 mirror in x, 1 mirror in y, 2 general
  int kind = -1;
  double dPBC = 0.0, Mbc = 0.0;
  if (bcx1 == bcx2 || nXpos == Xpos) {
    if (bcx1 == bcx2 && nXpos == Xpos) return;
    if (bcx1 == bcx2 && nYpos == Ypos) {
      intersctx = bcx1;
      interscty = nYpos;
      kind = 10;
    } else if (nXpos == Xpos && bcy1 == bcy2) {
      intersctx = nXpos;
      interscty = bcy1;
      kind = 11;
    } else if (bcx1 == bcx2 && nYpos != Ypos) {
      double Mp = (nYpos - Ypos) / (nXpos - Xpos);
      double Bp = Ypos - Mp * Xpos;
      intersctx = bcx1;
      interscty = Mp * intersctx + Bp;
      kind = 12;
    } else if (nXpos == Xpos && bcy1 != bcy2) {
      Mbc = (bcy2 - bcy1) / (bcx2 - bcx1);
      double Bbc = bcy2 - Mbc * bcx2;
      intersctx = nXpos;
      interscty = Mbc * intersctx + Bbc;
      kind = 2;
    } else {
      return;  // no branch of the source applies (e.g. vertical move, horizontal... handled above)
    }
  } else {
    Mbc = (bcy2 - bcy1) / (bcx2 - bcx1);
    double Bbc = bcy2 - Mbc * bcx2;
    double Mp = (nYpos - Ypos) / (nXpos - Xpos);
    double Bp = Ypos - Mp * Xpos;
    intersctx = (Bbc - Bp) / (Mp - Mbc);
    interscty = Mp * intersctx + Bp;
    if (Mbc == 0.0) interscty = byhigh;
    kind = (Mbc == 0.0) ? 13 : 2;
  }
  candidate = intersctx <= xhigh && intersctx >= xlow && interscty <= yhigh && interscty >= ylow &&
              intersctx <= bxhigh && intersctx >= bxlow && interscty <= byhigh && interscty >= bylow;
  if (!candidate) return;
  if (kind == 10) {
    dPBC = sqrt((intersctx - nXpos) * (intersctx - nXpos) + (interscty - nYpos) * (interscty - nYpos));
    rx1 = nXpos + (2.0 * dPBC); ry1 = nYpos;
    rx2 = nXpos - (2.0 * dPBC); ry2 = nYpos;
  } else if (kind == 11) {
    dPBC = sqrt((intersctx - nXpos) * (intersctx - nXpos) + (interscty - nYpos) * (interscty - nYpos));
    rx1 = nXpos; ry1 = nYpos + (2.0 * dPBC);
    rx2 = nXpos; ry2 = nYpos - (2.0 * dPBC);
  } else if (kind == 12) {
    dPBC = nXpos - intersctx;
    rx1 = nXpos + (2.0 * dPBC); ry1 = nYpos;
    rx2 = nXpos - (2.0 * dPBC); ry2 = nYpos;
  } else if (kind == 13) {
    dPBC = nYpos - bcy1;
    rx1 = nXpos; ry1 = nYpos + (2.0 * dPBC);
    rx2 = nXpos; ry2 = nYpos - (2.0 * dPBC);
  } else {
    double distBC = sqrt((bcx1 - bcx2) * (bcx1 - bcx2) + (bcy1 - bcy2) * (bcy1 - bcy2));
    double crossk = ((nXpos - bcx1) * (bcy2 - bcy1)) - ((bcx2 - bcx1) * (nYpos - bcy1));
    dPBC = sqrt(crossk * crossk) / distBC;
    double mBCperp = -1.0 / Mbc;
    double bBCperp = nYpos - mBCperp * nXpos;
    rx1 = sqrt(((2.0 * dPBC) * (2.0 * dPBC)) / (1.0 + mBCperp * mBCperp)) + nXpos;
    ry1 = mBCperp * rx1 + bBCperp;
    rx2 = sqrt(((2.0 * dPBC) * (2.0 * dPBC)) / (1.0 + mBCperp * mBCperp)) * -1.0 + nXpos;
    ry2 = mBCperp * rx2 + bBCperp;
  }
  double dist1 = sqrt((intersctx - rx1) * (intersctx - rx1) + (interscty - ry1) * (interscty - ry1));
  double dist2 = sqrt((intersctx - rx2) * (intersctx - rx2) + (interscty - ry2) * (interscty - ry2));
  if (dist1 < dist2) { H.rPx = rx1; H.rPy = ry1; }
  else if (dist1 > dist2) { H.rPx = rx2; H.rPy = ry2; }
  double d_Pinter = sqrt((Xpos - intersctx) * (Xpos - intersctx) + (Ypos - interscty) * (Ypos - interscty));
  if (d_Pinter < H.dtest) {
    H.fix = intersctx;
    H.fiy = interscty;
    H.frx = H.rPx;
    H.fry = H.rPy;
    H.found = 1;
    H.dtest = d_Pinter;
    H.seg = i;
    H.water = !P.land[i - 1];
  }
}

// intersect_reflect, boundary_module.f90:1620-1902
LT_DEV void intersect_reflect(const StepParams& P, double Xpos, double Ypos, double nXpos,
                              double nYpos, int skipbound, Hit& H, unsigned& brute_count) {
  H.fix = H.fiy = H.frx = H.fry = -999999.;
  H.rPx = H.rPy = -999999.;
  H.dtest = 999999.;
  H.seg = skipbound;
  H.found = 0;
  H.water = 0;
  double xhigh, xlow, yhigh, ylow;
  if (Xpos >= nXpos) { xhigh = Xpos; xlow = nXpos; } else { xhigh = nXpos; xlow = Xpos; }
  if (Ypos >= nYpos) { yhigh = Ypos; ylow = nYpos; } else { yhigh = nYpos; ylow = Ypos; }
  // candidate list: one cell list of the index (reference order), or all segments
  int q0 = 0, q1 = P.nbounds;
  bool indexed = false;
  if (P.use_seg_index) {
    const SegIndex& S = P.seg;
    double fx0 = floor((xlow - S.x0) * S.inv_cw), fx1 = floor((xhigh - S.x0) * S.inv_cw);
    double fy0 = floor((ylow - S.y0) * S.inv_ch), fy1 = floor((yhigh - S.y0) * S.inv_ch);
    // clamp to the grid (monotone, so overlap is preserved); NaN fails every test
    double mx = (double)(S.ncx - 1), my = (double)(S.ncy - 1);
    fx0 = fx0 < 0.0 ? 0.0 : (fx0 > mx ? mx : fx0);
    fx1 = fx1 < 0.0 ? 0.0 : (fx1 > mx ? mx : fx1);
    fy0 = fy0 < 0.0 ? 0.0 : (fy0 > my ? my : fy0);
    fy1 = fy1 < 0.0 ? 0.0 : (fy1 > my ? my : fy1);
    if (fx1 - fx0 <= 1.0 && fy1 - fy0 <= 1.0) {
      indexed = true;
      int c = (int)fx0 + (int)fy0 * S.ncx;
      q0 = S.ptr[c];
      q1 = S.ptr[c + 1];
    }
  }
  if (!indexed) brute_count += 1;
#pragma unroll 1
  for (int q = q0; q < q1; q++) {
    int i = indexed ? P.seg.ids[q] : q + 1;
    if (i == skipbound) continue;
    segment_test(P, i, Xpos, Ypos, nXpos, nYpos, xlow, xhigh, ylow, yhigh, H);
  }
}

// ------------------------------------------------------- settlement
// testSettlement / psettle / hsettle, settlement_module.f90:485-624
LT_DEV int test_settlement(const StepParams& P, double Px, double Py, int R_ele) {
  int polyin = 0;
  for (int q = P.elepolys_ptr[R_ele - 1]; q < P.elepolys_ptr[R_ele]; q++) {
    int pid = P.elepolys_ids[q];
    int start = P.polyspecs[pid - P.minpolyid];
    int size = P.polyspecs[(pid - P.minpolyid) + P.npoly];
    const double* base = P.polys + (start - 1);
    double cx = base[(size_t)P.pedges * 1], cy = base[(size_t)P.pedges * 2];
    double dis = sqrt((Px - cx) * (Px - cx) + (Py - cy) * (Py - cy));
    int idnum = (int)llrint(base[0]);
    if (dis > P.maxbdis[idnum - P.minpolyid]) continue;
    if (inpoly_full(Px, Py, size, base + (size_t)P.pedges * 3, base + (size_t)P.pedges * 4, false)) {
      polyin = idnum;
      break;
    }
  }
  if (polyin > 0 && P.holesExist) {
    int pi = polyin - P.minpolyid;
    for (int q = P.polyholes_ptr[pi]; q < P.polyholes_ptr[pi + 1]; q++) {
      int hidx = P.polyholes_ids[q];
      int start = P.holespecs[hidx - P.minholeid];
      int size = P.holespecs[(hidx - P.minholeid) + P.nhole];
      const double* base = P.holes + (start - 1);
      double cx = base[(size_t)P.hedges * 1], cy = base[(size_t)P.hedges * 2];
      double dis = sqrt((Px - cx) * (Px - cx) + (Py - cy) * (Py - cy));
      int idnum = (int)llrint(base[0]);
      if (dis > P.maxhdis[idnum - P.minholeid]) continue;
      // onin = .FALSE.: a particle on the edge of a hole still settles (:609)
      if (inpoly_full(Px, Py, size, base + (size_t)P.hedges * 3, base + (size_t)P.hedges * 4, true)) {
        return 0;
      }
    }
  }
  return polyin;
}

// ---------------------------------------------------- field access
template <typename SlabT>
LT_DEV double fld(const SlabT* F, int node, int L, int k, int ps) {
  return (double)F[((size_t)(node - 1) * L + (k - 1)) * LT_SLOTS + ps];
}

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
#ifdef LT_RK_TIMING
// development aid: wave time per section, accumulated in the wave's LDS scratch and
// added to g_tim once at the end of the kernel
__device__ unsigned long long g_tim[8];
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
// One variable's three time slots are queued, then iterated (pass B), then combined in
// time (pass C): at most 3 x 64 evaluation intervals and 6 x 64 check intervals.  The
// evaluation queue holds the worst case; a check interval that finds the queue full
// (never seen: ~1 % of them are candidates) sends its particle-step to k_rk_slow.
#define LT_CHK_CAP 256
#define LT_EV_CAP 192
struct WaveScratch {
  double chk[LT_CHK_CAP][2];         // num, den of T = MAX(D1/D2, D2/D1)
  unsigned short chk_owner[LT_CHK_CAP];  // lane | slot9 << 6
  double ev[LT_EV_CAP][7];           // x1,x2,y1,p1,p2,S,Pz
  unsigned short ev_owner[LT_EV_CAP];
  double res[3][64];                 // H(P_z) per (slot, lane) of the current variable
  unsigned int errmask[64];          // bit (variable*3+slot): SigErr / degenerate window
#ifdef LT_RK_TIMING
  unsigned long long tim[8];
#endif
};

// pass B: the queued iterations, 64 at a time, all lanes of the wave.  Evaluation
// intervals and check intervals share the batches, so the wave runs ONE Newton loop per
// batch whatever the mix (a separate loop for the few check intervals cost as much as the
// one for the evaluation intervals).
LT_DEV void run_tasks(WaveScratch& ws, int nchk, int nev, int lane) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int total = nev + nchk;
#pragma unroll 1
  for (int t = lane; t < total; t += 64) {
    const bool is_ev = t < nev;
    const double* e = ws.ev[is_ev ? t : 0];
    const int tc = is_ev ? 0 : (t - nev);
    const unsigned short o = is_ev ? ws.ev_owner[t] : ws.chk_owner[tc];
    double sg = 0.0, T = 0.0;
    int kind;
    if (is_ev) {
      // the evaluation interval: full SIGS decision
      kind = sigs_classify(e[5], e[3], e[4], sg, T);
    } else {
      // a non-evaluation interval: only the convergence of its Newton iteration matters
      T = ws.chk[tc][0] / ws.chk[tc][1];
      kind = (T > 2.0) ? SIGK_NEWTON : SIGK_DONE;
    }
    bool failed = false;
    if (kind == SIGK_NEWTON) sg = sigs_newton(T, failed);
    else if (kind == SIGK_SECANT) sg = sigs_secant(e[5], e[3], e[4]);  // rare
    if (is_ev) ws.res[(o >> 6) % 3][o & 63] = hval_interval_s(e[6], e[0], e[1], e[2], e[3], e[4], e[5], sg);
    if (failed) atomicOr(&ws.errmask[o & 63], 1u << (o >> 6));
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// find_currents, LTRANS.f90:1422-1614, executed by the WHOLE wave: lanes whose
// particle is not tracked (act = false) only take part in the cooperative pass.
// Per-particle constants of the water column are read from the step workspace
// (coalesced, L2-resident) where needed instead of living in registers.
template <typename SlabT, int VT>
LT_DEV void find_currents(const StepParams& P, const double* lev, WaveScratch& ws, bool act, int n,
                          int er, int eu, int ev_, double h, const Divisor& dv, double Xs, double Ys, double Zs, int version,
                          double& Uad, double& Vad, double& Wad, int& sigerr_count, bool& slow) {
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
    const double zb = wsd[(size_t)WS_ZETA0 * N], zc = wsd[(size_t)WS_ZETA1 * N], zf = wsd[(size_t)WS_ZETA2 * N];
    const LevCoef w1 = level_coef_d(wtab, 1, h, P.hc, VT, dv);
    const LevCoef r1 = level_coef_d(rtab, 1, h, P.hc, VT, dv);
    wzb1 = level_at(w1, zb, h, VT);
    const double wzc1 = level_at(w1, zc, h, VT), wzf1 = level_at(w1, zf, h, VT);
    zero = (Zs < wzb1 + P.z0) || (Zs < wzc1 + P.z0) || (Zs < wzf1 + P.z0);
    loglayer = !zero && ((Zs < level_at(r1, zb, h, VT)) || (Zs < level_at(r1, zc, h, VT)) ||
                         (Zs < level_at(r1, zf, h, VT)));
    if (!zero && !loglayer) {
      // lowest 4-level windows :1451-1467 by bisection on the monotone level depths
      int lo = 3, hi = P.us - 1;
      while (lo < hi) {
        int mid = (lo + hi) >> 1;
        LevCoef q = level_coef_d(rtab, mid, h, P.hc, VT, dv);
        bool below = (Zs < level_at(q, zb, h, VT)) || (Zs < level_at(q, zc, h, VT)) ||
                     (Zs < level_at(q, zf, h, VT));
        if (below) hi = mid; else lo = mid + 1;
      }
      ii = lo - 2;
      lo = 3; hi = P.ws - 1;
      while (lo < hi) {
        int mid = (lo + hi) >> 1;
        LevCoef q = level_coef_d(wtab, mid, h, P.hc, VT, dv);
        bool below = (Zs < level_at(q, zb, h, VT)) || (Zs < level_at(q, zc, h, VT)) ||
                     (Zs < level_at(q, zf, h, VT));
        if (below) hi = mid; else lo = mid + 1;
      }
      iii = lo - 2;
    }
  }
  const bool spl = act && !zero && !loglayer;  // this lane fits splines
  unsigned degenerate = 0;
  const double ixv = (version == 1) ? P.ix0 : (version == 2) ? P.ix1 : P.ix2;
  ws.errmask[lane] = 0;
  TICK(0);
  // ---------------- pass A: fit without iterating, queue what must iterate.
  // Memory latency is taken off the critical path by asking early: the node ids of the
  // NEXT variable's element, and the 4x4 field values + zeta + depth of the NEXT time
  // slot, are requested before the arithmetic of the current one (same registers: the
  // raw values are dead once the horizontal interpolation has produced Y).
  int4 nd_next = make_int4(1, 1, 1, 1);
  if (act && !zero) nd_next = P.UE[eu - 1];
  double zt = 0.0, pz = 0.0;
  SlabT raw[16];
#pragma unroll
  for (int i = 0; i < 16; i++) raw[i] = (SlabT)0;
#pragma unroll 1
  for (int var = 0; var < 3; var++) {
    const int4 nd = nd_next;
    const SlabT* F = (const SlabT*)((var == 0) ? P.U : (var == 1) ? P.V : P.W);
    const int L = (var == 2) ? P.ws : P.us;
    const double* tab = (var == 2) ? wtab : rtab;
    const int k0 = (var == 2) ? iii : ii;
    int nchk = 0, nev = 0;  // wave-uniform task counts of this variable
    // The 16 field values (4 nodes x 4 levels) of a time slot, its zeta and its depth are
    // requested one slot ahead of their use, into the registers the previous slot's values
    // just left (they are dead once the horizontal interpolation has produced Y).
    const SlabT *b0 = F, *b1 = F, *b2 = F, *b3 = F;
    if (spl) {
      b0 = F + ((size_t)(nd.x - 1) * L + (k0 - 1)) * LT_SLOTS;
      b1 = F + ((size_t)(nd.y - 1) * L + (k0 - 1)) * LT_SLOTS;
      b2 = F + ((size_t)(nd.z - 1) * L + (k0 - 1)) * LT_SLOTS;
      b3 = F + ((size_t)(nd.w - 1) * L + (k0 - 1)) * LT_SLOTS;
    }
    auto request = [&](int s) {
      const int ps = (s == 0) ? P.tb : (s == 1) ? P.tc : P.tf;
#pragma unroll
      for (int l = 0; l < 4; l++) {
        raw[4 * l + 0] = b0[LT_SLOTS * l + ps]; raw[4 * l + 1] = b1[LT_SLOTS * l + ps];
        raw[4 * l + 2] = b2[LT_SLOTS * l + ps]; raw[4 * l + 3] = b3[LT_SLOTS * l + ps];
      }
      zt = wsd[(size_t)(WS_ZETA0 + s) * N];
      pz = wsd[(size_t)(WS_PZ0 + s) * N];
    };
    if (spl) request(0);
    QuadWeights w;
    w.regime = 1; w.node_hit = 0; w.t = w.u = w.W1 = w.W2 = w.W3 = w.W4 = 0.0;
    if (act && !zero) {
      Quad q;
      if (var == 2) q = load_quad_rho(P.rnode, nd);
      else if (var == 0) q = load_quad_xy(P.unode, nd);
      else q = load_quad_xy(P.vnode, nd);
      if (var == 0) nd_next = P.VE[ev_ - 1];
      else if (var == 1) nd_next = P.RE[er - 1];
      w = quad_weights(Xs, Ys, q.x, q.y);
    }
    TICK(1);
    if (loglayer) {
      // log layer :1493-1522: level 1 of u,v; level 2 of w
      const int k = (var == 2) ? 2 : 1;
      const LevCoef qt = level_coef_d((var == 2) ? wtab : rtab, k, h, P.hc, VT, dv);
#pragma unroll
      for (int s = 0; s < 3; s++) {
        const int ps = (s == 0) ? P.tb : (s == 1) ? P.tc : P.tf;
        const double zeta = wsd[(size_t)(WS_ZETA0 + s) * N];
        double val = apply_interp(w, fld(F, nd.x, L, k, ps), fld(F, nd.y, L, k, ps),
                                  fld(F, nd.z, L, k, ps), fld(F, nd.w, L, k, ps));
        double ztop = level_at(qt, zeta, h, VT);
        ws.res[s][lane] = val * log10((Zs - wzb1) / P.z0) / log10((ztop - wzb1) / P.z0);
      }
    }
    LevCoef lc[4];
    if (spl) {
#pragma unroll
      for (int l = 0; l < 4; l++) lc[l] = level_coef_d(tab, k0 + l, h, P.hc, VT, dv);
    }
#pragma unroll 1
    for (int s = 0; s < 3; s++) {
      const int slot9 = var * 3 + s;
      double YP[4], SL[3];
      double x1 = 0.0, x2 = 0.0, y1 = 0.0, p1 = 0.0, p2 = 0.0, Sv = 0.0, Pzs = 0.0;
      int I = 1;
      bool ok = spl;
      if (spl) {
        const double zeta = zt;
        Pzs = pz;
        double X[4], Y[4];
#pragma unroll
        for (int l = 0; l < 4; l++) {
          X[l] = level_at(lc[l], zeta, h, VT);
          Y[l] = apply_interp(w, (double)raw[4 * l + 0], (double)raw[4 * l + 1],
                              (double)raw[4 * l + 2], (double)raw[4 * l + 3]);
        }
        if (s < 2) request(s + 1);
        const double DX1 = X[1] - X[0], DX2 = X[2] - X[1], DX3 = X[3] - X[2];
        ok = (DX1 > 0.0) && (DX2 > 0.0) && (DX3 > 0.0);
        if (!ok) degenerate |= (1u << slot9);
        // YPC1 (tension_module.f90:852-977) with the chord slopes kept for SIGS/HVAL
        // (seven quotients over five divisors: the reciprocals are prepared once)
        const Divisor d1 = prep_div(DX1), d2 = prep_div(DX2), d3 = prep_div(DX3);
        const Divisor d12 = prep_div(DX1 + DX2), d23 = prep_div(DX2 + DX3);
        SL[0] = div_prep(Y[1] - Y[0], d1);
        SL[1] = div_prep(Y[2] - Y[1], d2);
        SL[2] = div_prep(Y[3] - Y[2], d3);
        YP[0] = ypc1_end(SL[0], SL[0] + div_prep(DX1 * (SL[0] - SL[1]), d12));
        YP[1] = ypc1_mid_d(DX1, DX2, d12, SL[0], SL[1]);
        YP[2] = ypc1_mid_d(DX2, DX3, d23, SL[1], SL[2]);
        YP[3] = ypc1_end(SL[2], SL[2] + div_prep(DX3 * (SL[2] - SL[1]), d23));
        I = (Pzs < X[1]) ? 1 : (Pzs < X[2]) ? 2 : 3;
        if (Pzs > X[3]) I = 3;
        x1 = (I == 1) ? X[0] : (I == 2) ? X[1] : X[2];
        x2 = (I == 1) ? X[1] : (I == 2) ? X[2] : X[3];
        y1 = (I == 1) ? Y[0] : (I == 2) ? Y[1] : Y[2];
        p1 = (I == 1) ? YP[0] : (I == 2) ? YP[1] : YP[2];
        p2 = (I == 1) ? YP[1] : (I == 2) ? YP[2] : YP[3];
        Sv = (I == 1) ? SL[0] : (I == 2) ? SL[1] : SL[2];
      }
      TICK(2);
      // SIGS (tension_module.f90:448-776), screened without divisions or loops.
      // Only the evaluation interval's tension is used by HVAL; the other two
      // matter only through SigErr, which only the convexity Newton iteration
      // can raise, so they are queued just when they are Newton candidates.
      bool evdone = false;
      double sigI = 0.0;
      {
        double n0, d0, n1, d1, n2, d2;
        const bool c0 = ok && sigs_candidate(SL[0], YP[0], YP[1], n0, d0);
        const bool c1 = ok && sigs_candidate(SL[1], YP[1], YP[2], n1, d1);
        const bool c2 = ok && sigs_candidate(SL[2], YP[2], YP[3], n2, d2);
        // the two non-evaluation intervals: I=1 -> (2,3), I=2 -> (1,3), I=3 -> (1,2)
        const bool ca = (I == 1) ? c1 : c0, cb = (I == 3) ? c1 : c2;
        const double na = (I == 1) ? n1 : n0, da = (I == 1) ? d1 : d0;
        const double nb = (I == 3) ? n1 : n2, db = (I == 3) ? d1 : d2;
        if (ok) evdone = sigs_eval_known(Sv, p1, p2, sigI);
        const bool neede = ok && !evdone;
        const unsigned long long ma = __ballot(ca), mb = __ballot(cb), me = __ballot(neede);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (ca) {
          const int idx = nchk + __popcll(ma & below);
          if (idx < LT_CHK_CAP) {
            ws.chk[idx][0] = na; ws.chk[idx][1] = da;
            ws.chk_owner[idx] = (unsigned short)(lane | (slot9 << 6));
          } else degenerate |= (1u << slot9);
        }
        nchk += __popcll(ma);
        if (cb) {
          const int idx = nchk + __popcll(mb & below);
          if (idx < LT_CHK_CAP) {
            ws.chk[idx][0] = nb; ws.chk[idx][1] = db;
            ws.chk_owner[idx] = (unsigned short)(lane | (slot9 << 6));
          } else degenerate |= (1u << slot9);
        }
        nchk += __popcll(mb);
        if (neede) {
          const int idx = nev + __popcll(me & below);
          double* e = ws.ev[idx];
          e[0] = x1; e[1] = x2; e[2] = y1; e[3] = p1; e[4] = p2; e[5] = Sv; e[6] = Pzs;
          ws.ev_owner[idx] = (unsigned short)(lane | (slot9 << 6));
        }
        nev += __popcll(me);
#ifdef LT_RK_TIMING
        { const unsigned long long mk = __ballot(evdone && sigI != 0.0);
          if (lane == 0) { ws.tim[6] += __popcll(me); ws.tim[7] += __popcll(mk); } }
#endif
      }
      // HVAL now when the evaluation interval's tension is already known
      if (evdone) ws.res[s][lane] = hval_interval_s(Pzs, x1, x2, y1, p1, p2, Sv, sigI);
      TICK(3);
    }
    // ---------------- pass B: everything of this variable that iterates
    run_tasks(ws, (nchk < LT_CHK_CAP) ? nchk : LT_CHK_CAP, nev, lane);
    TICK(4);
    // ---------------- pass C: time interpolation :1534-1598 / hydrodynamic_module.f90:2653-2670
    if (act && !zero) {
      const double Pv0 = ws.res[0][lane], Pv1 = ws.res[1][lane], Pv2 = ws.res[2][lane];
      double e1 = Pv0, e2 = Pv1, e3 = Pv2;
      if (P.p == 1) { e2 = Pv0; e3 = Pv1; }
      const double val = polintd3(P.ex0, P.ex1, P.ex2, e1, e2, e3, ixv);
      if (var == 0) Uad = val; else if (var == 1) Vad = val; else Wad = val;
    }
    TICK(5);
  }
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
// update_particles() is run as three kernels over the same particle index so
// that the register-hungry RK4 part does not carry the state of the element
// search and boundary code (occupancy), exchanging 13 doubles + one flag word
// per particle through a workspace in HBM (SoA, stride numpar):
//   k_pre   gates, setEle, setInterp/getInterp, vertical clamps   (per lane)
//   k_rk    RK4 over find_currents/WCTS_ITPI                      (wave-cooperative)
//   k_post  behaviour, reflection, boundaries, setEle, settlement, commit

template <typename SlabT>
__global__ __launch_bounds__(256) void k_pre(const StepParams P) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
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
        const double P_depth = -1.0 * apply_getinterp(w0, r1.depth, r2.depth, r3.depth, r4.depth);
        const double P_angle = apply_getinterp(w0, r1.angle, r2.angle, r3.angle, r4.angle);
        using Z4 = typename SlabTraits<SlabT>::zeta4;
        const Z4* zt = (const Z4*)P.zeta;
        Z4 z1 = zt[rn.x - 1], z2 = zt[rn.y - 1], z3 = zt[rn.z - 1], z4 = zt[rn.w - 1];
        auto pick = [](const Z4& v, int s) -> double {
          return (double)((s == 0) ? v.x : (s == 1) ? v.y : (s == 2) ? v.z : v.w);
        };
        double zeta[3], Pz[3];
        zeta[0] = apply_getinterp(w0, pick(z1, P.tb), pick(z2, P.tb), pick(z3, P.tb), pick(z4, P.tb));
        zeta[1] = apply_getinterp(w0, pick(z1, P.tc), pick(z2, P.tc), pick(z3, P.tc), pick(z4, P.tc));
        zeta[2] = apply_getinterp(w0, pick(z1, P.tf), pick(z2, P.tf), pick(z3, P.tf), pick(z4, P.tf));
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
        wsd[(size_t)WS_COSA * N] = cos(P_angle);
        wsd[(size_t)WS_SINA * N] = sin(P_angle);
        wsd[(size_t)WS_ZETA0 * N] = zeta[0];
        wsd[(size_t)WS_ZETA1 * N] = zeta[1];
        wsd[(size_t)WS_ZETA2 * N] = zeta[2];
        wsd[(size_t)WS_PZ0 * N] = Pz[0];
        wsd[(size_t)WS_PZ1 * N] = Pz[1];
        wsd[(size_t)WS_PZ2 * N] = Pz[2];
        wsd[(size_t)WS_ZPAR * N] = Zpar;  // = par(n,pZ) after :922
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
                            const QuadWeights& w, const SlabT* F, int deplvl) {
  double Pv[3];
#pragma unroll 1
  for (int s = 0; s < 3; s++) {
    const int ps = (s == 0) ? P.tb : (s == 1) ? P.tc : P.tf;
    double az[4], av[4];
#pragma unroll
    for (int l = 0; l < 4; l++) {
      az[l] = level_z(lev, deplvl + l, C.zeta[s], C.h, P.hc, P.Vtransform);
      av[l] = apply_interp(w, fld(F, rn.x, P.us, deplvl + l, ps), fld(F, rn.y, P.us, deplvl + l, ps),
                           fld(F, rn.z, P.us, deplvl + l, ps), fld(F, rn.w, P.us, deplvl + l, ps));
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

// Everything of behave() (behavior_module.f90:181-551) and of the SaltTempOn block
// (LTRANS.f90:1062-1076) that does not depend on the random deviates or on the RK4 result:
// swimming speed, stage changes, the C. ariakensis timer, the salinity at the particle, the
// salinity-gradient cue, and with them WHAT the behaviour will draw (none / one directed
// deviate / two random-swimming deviates / both), which the draw offsets of all later
// particles depend on.  k_post consumes the deviates and finishes the behaviour.
template <typename SlabT>
__global__ __launch_bounds__(256) void k_beh_pre(const StepParams P) {
  __shared__ double lev[6 * LT_LEV_STRIDE];
  for (int i = threadIdx.x; i < 6 * LT_LEV_STRIDE; i += blockDim.x) lev[i] = P.lev[i];
  __syncthreads();
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= P.numpar) return;
  int fl = P.ws_i[n];
  if (!(fl & FL_ACT)) return;
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
  int beh = (P.Behavior != 0) ? P.P_behave[n] : 0;
  double tim = 0.0;
  if (P.Behavior != 0) {
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
  const bool want_S = (beh == 4) || (beh == 5 && tim == 0.0) || (beh == 7);
  double P_S = 0.0;
  if (want_S || P.SaltTempOn) {
    const int4 rn = P.RE[P.P_r_element[n] - 1];
    const Quad q = load_quad_rho(P.rnode, rn);
    const QuadWeights w = quad_weights(X, Y, q.x, q.y);
    const int deplvl = level_window(lev, P.us, Zpar, C.zeta[0], C.zeta[1], C.zeta[2], C.h, P.hc, P.Vtransform);
    P_S = wcts_itpi_rho<SlabT>(P, lev, C, rn, w, (const SlabT*)P.S, deplvl);
    if (P.SaltTempOn) {
      P.P_Salt[n] = P_S;
      P.P_Temp[n] = wcts_itpi_rho<SlabT>(P, lev, C, rn, w, (const SlabT*)P.T, deplvl);
    }
  }
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
      E0 = P.Em * sin(P.PI * tst / (P.daylength * 3600.0)) * sin(P.PI * tst / (P.daylength * 3600.0));
    }
    const double P_light = E0 * exp(P.Kd * P_zc);
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

template <typename SlabT, int VT, int WAVES_PER_SIMD>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void k_rk(const StepParams P) {
  __shared__ double lev[6 * LT_LEV_STRIDE];
  __shared__ WaveScratch wsc[4];
  for (int i = threadIdx.x; i < 6 * LT_LEV_STRIDE; i += blockDim.x) lev[i] = P.lev[i];
  __syncthreads();
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const size_t N = (size_t)P.numpar;
  const bool act = (n < P.numpar) && ((P.ws_i[n] & FL_ACT) != 0);
  const double* wsd = P.ws_d + (act ? n : 0);
  double h = 1.0, Xs = 0.0, Ys = 0.0, Zs = 0.0;
  if (act) {
    h = -1.0 * wsd[(size_t)WS_PDEPTH * N];
    Xs = P.par[(size_t)pX * N + n];
    Ys = P.par[(size_t)pY * N + n];
    Zs = wsd[(size_t)WS_ZPAR * N];
  }
  // RK4 :990-1053, executed by the whole wave
  double sumU = 0.0, sumV = 0.0, sumW = 0.0;
  int sigerr_count = 0;
  bool slow = false;
  WaveScratch& ws = wsc[threadIdx.x >> 6];
#ifdef LT_RK_TIMING
  if ((threadIdx.x & 63) < 8) ws.tim[threadIdx.x & 63] = 0;
#endif
  const Divisor dv = prep_div((VT == 2) ? P.hc + h : h);  // every level depth divides by it
  // the RK stages keep the elements of the start position (LTRANS.f90:990-1053)
  int er = 1, eu = 1, ev_ = 1;
  if (act) { er = P.P_r_element[n]; eu = P.P_u_element[n]; ev_ = P.P_v_element[n]; }
#pragma unroll 1
  for (int st = 0; st < 4; st++) {
    const int version = (st == 0) ? 1 : (st == 3) ? 3 : 2;
    double Uad, Vad, Wad;
    // (h made opaque per stage: otherwise every h-dependent level coefficient is hoisted
    // out of the stage loop and held in registers for the whole kernel)
    asm volatile("" : "+v"(h));
    find_currents<SlabT, VT>(P, lev, ws, act, n, er, eu, ev_, h, dv, Xs, Ys, Zs, version, Uad, Vad, Wad, sigerr_count, slow);
    if (st == 0) { sumU = Uad; sumV = Vad; sumW = Wad; }
    else if (st == 3) { sumU = sumU + Uad; sumV = sumV + Vad; sumW = sumW + Wad; }
    else { sumU = sumU + 2.0 * Uad; sumV = sumV + 2.0 * Vad; sumW = sumW + 2.0 * Wad; }
    if (st < 3 && act) {
      // next RK position :999-1035; the per-particle constants come back from
      // the workspace / par (coalesced) instead of staying in registers
      const double X = P.par[(size_t)pX * N + n], Y = P.par[(size_t)pY * N + n];
      const double Zpar = wsd[(size_t)WS_ZPAR * N];
      const double cosA = wsd[(size_t)WS_COSA * N], sinA = wsd[(size_t)WS_SINA * N];
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
      // maxpartdepth / minpartdepth :981-987
      const double* wtab = lev + 3 * LT_LEV_STRIDE;
      const LevCoef w1 = level_coef_d(wtab, 1, h, P.hc, VT, dv);
      const LevCoef wN = level_coef_d(wtab, P.ws, h, P.hc, VT, dv);
      const double zb = wsd[(size_t)WS_ZETA0 * N], zc = wsd[(size_t)WS_ZETA1 * N], zf = wsd[(size_t)WS_ZETA2 * N];
      double maxpartdepth = level_at(w1, zb, h, VT);
      double t1 = level_at(w1, zc, h, VT), t2 = level_at(w1, zf, h, VT);
      if (t1 > maxpartdepth) maxpartdepth = t1;
      if (t2 > maxpartdepth) maxpartdepth = t2;
      double minpartdepth = level_at(wN, zb, h, VT);
      t1 = level_at(wN, zc, h, VT); t2 = level_at(wN, zf, h, VT);
      if (t1 < minpartdepth) minpartdepth = t1;
      if (t2 < minpartdepth) minpartdepth = t2;
      if (Zs > minpartdepth) Zs = minpartdepth - (double)0.000001f;
      if (Zs < maxpartdepth) Zs = maxpartdepth + (double)0.000001f;
    }
  }
#ifdef LT_RK_TIMING
  if ((threadIdx.x & 63) < 8) atomicAdd(&g_tim[threadIdx.x & 63], ws.tim[threadIdx.x & 63]);
#endif
  if (act) {
    double* wso = P.ws_d + n;
    wso[(size_t)WS_SUMU * N] = sumU;
    wso[(size_t)WS_SUMV * N] = sumV;
    wso[(size_t)WS_SUMW * N] = sumW;
    if (slow) P.ws_i[n] |= FL_SLOW;
  }
  unsigned v = (unsigned)sigerr_count;
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0 && v) atomicAdd(&P.counters[CNT_SIGERR], (unsigned long long)v);
}

// ---- k_rk_slow: the particle-steps k_rk could not finish on its fast path
// (a SigErr in some spline fit, or a degenerate level window) are redone here
// by a straight per-lane transcription of find_currents / WCTS_ITPI
// (LTRANS.f90:1422-1614, hydrodynamic_module.f90:2577-2689): YPC1, the SIGS
// interval loop with its early RETURN, HVAL or the linint fallback.
template <typename SlabT>
LT_DEV void find_currents_ref(const StepParams& P, const double* lev, const Column& C, int4 rn,
                              int4 un, int4 vn, double Xs, double Ys, double Zs, int version,
                              double& Uad, double& Vad, double& Wad) {
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
    double Pv0 = 0.0, Pv1 = 0.0, Pv2 = 0.0;
#pragma unroll 1
    for (int s = 0; s < 3; s++) {
      const int ps = (s == 0) ? P.tb : (s == 1) ? P.tc : P.tf;
      const double zeta = (s == 0) ? C.zeta[0] : (s == 1) ? C.zeta[1] : C.zeta[2];
      double r;
      if (loglayer) {
        const int k = (var == 2) ? 2 : 1;
        double val = apply_interp(w, fld(F, nd.x, L, k, ps), fld(F, nd.y, L, k, ps),
                                  fld(F, nd.z, L, k, ps), fld(F, nd.w, L, k, ps));
        const double z1s = (s == 0) ? C.z1[0] : (s == 1) ? C.z1[1] : C.z1[2];
        double ztop = (var == 2) ? level_z(wtab, 2, zeta, C.h, P.hc, P.Vtransform) : z1s;
        r = val * log10((Zs - C.wz1[0]) / P.z0) / log10((ztop - C.wz1[0]) / P.z0);
      } else {
        const int k0 = (var == 2) ? iii : ii;
        const double Pzs = (s == 0) ? C.Pz[0] : (s == 1) ? C.Pz[1] : C.Pz[2];
        double az[4], av[4];
#pragma unroll
        for (int l = 0; l < 4; l++) {
          az[l] = level_z(tab, k0 + l, zeta, C.h, P.hc, P.Vtransform);
          av[l] = apply_interp(w, fld(F, nd.x, L, k0 + l, ps), fld(F, nd.y, L, k0 + l, ps),
                               fld(F, nd.z, L, k0 + l, ps), fld(F, nd.w, L, k0 + l, ps));
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
  for (int i = threadIdx.x; i < 6 * LT_LEV_STRIDE; i += blockDim.x) lev[i] = P.lev[i];
  __syncthreads();
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= P.numpar) return;
  if (!(P.ws_i[n] & FL_SLOW)) return;
  const size_t N = (size_t)P.numpar;
  double* wsd = P.ws_d + n;
  const double X = P.par[(size_t)pX * N + n], Y = P.par[(size_t)pY * N + n];
  const double Zpar = wsd[(size_t)WS_ZPAR * N];
  const double cosA = wsd[(size_t)WS_COSA * N], sinA = wsd[(size_t)WS_SINA * N];
  const int4 rn = P.RE[P.P_r_element[n] - 1], un = P.UE[P.P_u_element[n] - 1],
             vn = P.VE[P.P_v_element[n] - 1];
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
    find_currents_ref<SlabT>(P, lev, C, rn, un, vn, Xs, Ys, Zs, version, Uad, Vad, Wad);
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
  return sqrt(-2.0 * log(dev1)) * cos(2.0 * P.PI * dev2);
}

// HVAL / HPVAL on a spline table in memory (element k of array a at a[k*stride]),
// tension_module.f90:980-1253.  `il` caches the last interval like INTRVL's SAVE IL.
template <bool DERIV>
LT_DEV double hval_table(double T, int n, const double* x, const double* y, const double* yp,
                         const double* sig, size_t stride, int& il) {
  int I;
  if (T < x[0]) I = 1;
  else if (T > x[(size_t)(n - 1) * stride]) I = n - 1;
  else {
    // INTRVL :1287-1354: try the previous interval, else bisection
    bool hit = false;
    if (il >= 1 && il < n) hit = (x[(size_t)(il - 1) * stride] <= T) && (T < x[(size_t)il * stride]);
    if (!hit) {
      int lo = 1, hi = n;
      while (hi > lo + 1) {
        int kk = (lo + hi) >> 1;
        if (T < x[(size_t)(kk - 1) * stride]) hi = kk; else lo = kk;
      }
      il = lo;
    }
    I = il;
  }
  const size_t a = (size_t)(I - 1) * stride, b = (size_t)I * stride;
  const double X1 = x[a], X2 = x[b], Y1 = y[a], Y2 = y[b], P1 = yp[a], P2 = yp[b], SG = sig[a];
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
                             double q32, double q21, double r13, bool fd, double x, int ns) {
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
LT_DEV void polintd3x3(const StepParams& P, double ya1, double ya2, double ya3,
                       double& rb, double& rc, double& rf) {
  const double xa1 = P.ex0, xa2 = P.ex1, xa3 = P.ex2;
  const bool fd = P.fastdiv != 0;
  const double q32 = div_by(ya3 - ya2, xa2 - xa3, P.r_d23, fd);
  const double q21 = div_by(ya2 - ya1, xa1 - xa2, P.r_d12, fd);
  rb = polintd_shared(xa1, xa2, xa3, ya1, ya2, ya3, q32, q21, P.r_d13, fd, P.ix0, P.pns0);
  rc = polintd_shared(xa1, xa2, xa3, ya1, ya2, ya3, q32, q21, P.r_d13, fd, P.ix1, P.pns1);
  rf = polintd_shared(xa1, xa2, xa3, ya1, ya2, ya3, q32, q21, P.r_d13, fd, P.ix2, P.pns2);
}

// The interval of the VTurb spline table a walk is in, kept in registers: HVAL and
// HPVAL (tension_module.f90:980-1253) reload from HBM only when the depth leaves it.
// For a strictly increasing abscissa table INTRVL's answer (:1287-1354) is the unique
// I with X(I) <= T < X(I+1), clamped to 1..N-1, whatever its saved IL was, so
// stepping from the cached interval finds the same I as its bisection.
struct VtInterval { int I; double X1, X2, Y1, Y2, P1, P2, SG; };
LT_DEV void vt_seek(VtInterval& c, double T, int n, const double* x, const double* y,
                    const double* yp, const double* sig, size_t stride) {
  bool moved = false;
  if (c.I == 0) {  // first use: bisection
    int lo = 1, hi = n;
    while (hi > lo + 1) {
      const int kk = (lo + hi) >> 1;
      if (T < x[(size_t)(kk - 1) * stride]) hi = kk; else lo = kk;
    }
    c.I = lo;
    c.X1 = x[(size_t)(lo - 1) * stride];
    c.X2 = x[(size_t)lo * stride];
    moved = true;
  }
  while (T < c.X1 && c.I > 1) { c.I -= 1; c.X2 = c.X1; c.X1 = x[(size_t)(c.I - 1) * stride]; moved = true; }
  while (T >= c.X2 && c.I < n - 1) { c.I += 1; c.X1 = c.X2; c.X2 = x[(size_t)c.I * stride]; moved = true; }
  if (moved) {
    const size_t a = (size_t)(c.I - 1) * stride, b = (size_t)c.I * stride;
    c.Y1 = y[a]; c.Y2 = y[b]; c.P1 = yp[a]; c.P2 = yp[b]; c.SG = sig[a];
  }
}

// per-wave LDS queue of the spline intervals of the VTurb table whose tension
// factor needs an iteration (SIGS, tension_module.f90:531-760)
#define LT_VT_CAP 256
struct VtScratch {
  double S[LT_VT_CAP], S1[LT_VT_CAP], S2[LT_VT_CAP];
  unsigned short where[LT_VT_CAP];  // interval index k (0-based)
  unsigned char lane[LT_VT_CAP];
  unsigned int failed[64];
};

LT_DEV void vt_run_tasks(VtScratch& q, int ntask, int lane, int wave_first, double* vt_sig,
                         int p2) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 1
  for (int t = lane; t < ntask; t += 64) {
    double sg, T;
    bool failed = false;
    const int kind = sigs_classify(q.S[t], q.S1[t], q.S2[t], sg, T);
    if (kind == SIGK_NEWTON) sg = sigs_newton(T, failed);
    else if (kind == SIGK_SECANT) sg = sigs_secant(q.S[t], q.S1[t], q.S2[t]);
    vt_sig[((size_t)(wave_first >> 6) * (size_t)p2 + q.where[t]) * 64 + q.lane[t]] = sg;
    if (failed) atomicOr(&q.failed[q.lane[t]], 1u);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// VTurb, ver_turb_module.f90:30-380, as three kernels, one lane per particle:
//   k_vt_table  builds the spline table of the time-interpolated, smoothed Kh profile
//               (the register-heavy, arithmetic part);
//   k_vt_sigs   chooses the tension factors (SIGS), iterating only the intervals that
//               need it, 64 at a time per wave;
//   k_vt_walk   runs the idt/2 random-walk sub-steps on the table.
// The ~30 work arrays of the reference are never materialised: the proliferated /
// smoothed profile is produced as a stream over k (8-point window in registers),
// the YPC1 derivative of point k-1 is formed as soon as point k exists, and only the
// final spline table (x, y, yp, sigma)(k) goes to HBM, [wave][k][lane], so every table
// access of the build phases is coalesced and a wave streams one contiguous block.
// SIGS' "RETURN at the first SigErr" leaves later tensions at 0 in the reference;
// since SigErr switches every later evaluation to linint, which ignores the
// tensions, computing all of them here changes nothing.
template <typename SlabT, int VT>
__global__ __launch_bounds__(256, 2) void k_vt_table(const StepParams P) {
  __shared__ double lev[6 * LT_LEV_STRIDE];
  __shared__ double wring[3][8][256];
  for (int i = threadIdx.x; i < 6 * LT_LEV_STRIDE; i += blockDim.x) lev[i] = P.lev[i];
  __syncthreads();
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= P.numpar || !(P.ws_i[n] & FL_ACT)) return;
  const bool act = true;
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
  auto KH = [&](int s, int i) -> double {  // Pwc_KH{b,c,f}(i) :102-108
    return apply_getinterp(w0, fld(K, rn.x, ws, i, pslot[s]), fld(K, rn.y, ws, i, pslot[s]),
                           fld(K, rn.z, ws, i, pslot[s]), fld(K, rn.w, ws, i, pslot[s]));
  };
  auto WZ = [&](int s, int i) -> double {  // Pwc_wz{b,c,f}(i)
    return level_at(level_coef(wtab, i, h, P.hc, VT), zeta[s], h, VT);
  };
  double wz1[3], wzN[3], kh1[3], khN[3], rng_[3];
#pragma unroll
  for (int s = 0; s < 3; s++) {
    wz1[s] = WZ(s, 1); wzN[s] = WZ(s, ws);
    kh1[s] = KH(s, 1); khN[s] = KH(s, ws);
    rng_[s] = wzN[s] - wz1[s];
  }
  auto NEWX = [&](int s, int j) -> double {  // :121-123
    return wz1[s] + div_by(((double)(float)(j - 4)) * rng_[s], (double)p2, P.r_p2, P.fastdiv != 0);
  };
  // piecewise-linear profile segment per slot :126-163.  Moving up one level reuses
  // the upper end of the previous segment (same values), and the Kh of the level after
  // that was asked for one segment earlier, so its latency is off the critical path.
  int jlo[3] = {1, 1, 1};
  double slope[3], icpt[3], wzhi[3], khhi[3];
  SlabT rawn[3][4];
  auto RAW = [&](int s, int i) {
    const size_t o = (size_t)(i - 1) * LT_SLOTS + pslot[s], L3 = (size_t)ws * LT_SLOTS;
    rawn[s][0] = K[(size_t)(rn.x - 1) * L3 + o]; rawn[s][1] = K[(size_t)(rn.y - 1) * L3 + o];
    rawn[s][2] = K[(size_t)(rn.z - 1) * L3 + o]; rawn[s][3] = K[(size_t)(rn.w - 1) * L3 + o];
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
  auto NEWY = [&](int s, int j) -> double {  // newy{b,c,f}(j), j = 5..p2+3
    const double xj = NEWX(s, j);
    while (!(wzhi[s] > xj) && jlo[s] < ws - 1) {
      jlo[s] += 1;
      const double kb = apply_getinterp(w0, (double)rawn[s][0], (double)rawn[s][1], (double)rawn[s][2],
                                        (double)rawn[s][3]);
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
    for (int j = 5; j <= 8; j++) wring[s][j & 7][threadIdx.x] = NEWY(s, j);
  }
  // table layout: [wave][k][lane], a wave streams one contiguous 4*ws*512 B block per array
  const size_t TS = 64, toff = (size_t)(na >> 6) * (size_t)p2 * 64 + (size_t)(na & 63);
  double* tx = P.vt_x + toff; double* ty = P.vt_y + toff; double* typ = P.vt_yp + toff;
  // streaming YPC1 / SIGS state: previous point, previous interval, YP(k-2)
  double xm1 = 0.0, ym1 = 0.0, Sp = 0.0, DXp = 0.0, YPp = 0.0;
  bool ypc_fail = false;
#pragma unroll 1
  for (int k = 1; k <= p2; k++) {
    if (act) {
      double mx[3], my[3];
#pragma unroll
      for (int s = 0; s < 3; s++) {
        if (k == 1) { mx[s] = wz1[s]; my[s] = kh1[s]; }            // :197-210 pinned ends
        else if (k == p2) { mx[s] = wzN[s]; my[s] = khN[s]; }
        else {                                                      // :184-194 8-point average
          const int t = threadIdx.x;
          my[s] = (wring[s][k & 7][t] + wring[s][(k + 1) & 7][t] + wring[s][(k + 2) & 7][t] + wring[s][(k + 3) & 7][t] +
                   wring[s][(k + 4) & 7][t] + wring[s][(k + 5) & 7][t] + wring[s][(k + 6) & 7][t] + wring[s][(k + 7) & 7][t]) / 8.0;
          const double xa = NEWX(s, k);
          mx[s] = xa + (NEWX(s, k + 7) - xa) / 2.0;
        }
      }
      // :220-275 interpolation to the internal times, clamp, Simpson in time
      // (selects on values, not on array elements: those would turn into stack indexing)
      const double mx0 = mx[0], mx1 = mx[1], mx2 = mx[2], my0 = my[0], my1 = my[1], my2 = my[2];
      double ex1 = mx0, ex2 = mx1, ex3 = mx2, ey1 = my0, ey2 = my1, ey3 = my2;
      if (P.p == 1) { ex2 = mx0; ex3 = mx1; ey2 = my0; ey3 = my1; }
      double fxb, fxc, fxf, fyb, fyc, fyf;
      polintd3x3(P, ex1, ex2, ex3, fxb, fxc, fxf);
      polintd3x3(P, ey1, ey2, ey3, fyb, fyc, fyf);
      if (fyb < 0.0) fyb = 0.0;
      if (fyc < 0.0) fyc = 0.0;
      if (fyf < 0.0) fyf = 0.0;
      const double yk = div_by(fyb + 4.0 * fyc + fyf, 6.0, P.r_6, P.fastdiv != 0);
      const double xk = div_by(fxb + 4.0 * fxc + fxf, 6.0, P.r_6, P.fastdiv != 0);
      tx[(size_t)(k - 1) * TS] = xk;
      ty[(size_t)(k - 1) * TS] = yk;
      // TSPSI(p2) as a stream: YPC1 (tension_module.f90:852-977), SIGS (:314-781)
      if (k >= 2 && !ypc_fail) {
        const double DX = xk - xm1;  // DX_{k-1}
        if (DX <= 0.0) {
          ypc_fail = true;  // YPC1 fails -> TSPSI returns IER = -4 and SIGS never runs
          typ[(size_t)(k - 2) * TS] = 0.0;  // derivatives not reached stay 0
        } else {
          const double S = (yk - ym1) / DX;  // S_{k-1}
          if (k == 3) {
            YPp = ypc1_end(Sp, Sp + DXp * (Sp - S) / (DXp + DX));  // YP(1)
            typ[0] = YPp;
          }
          if (k >= 3) {
            const double YPc = ypc1_mid(DXp, DX, Sp, S);  // YP(k-1)
            typ[(size_t)(k - 2) * TS] = YPc;
            YPp = YPc;
          }
          if (k == p2) {
            const double YPn = ypc1_end(S, S + DX * (S - Sp) / (DXp + DX));  // YP(N)
            typ[(size_t)(k - 1) * TS] = YPn;
          }
          Sp = S;
          DXp = DX;
        }
      }
      if (ypc_fail) typ[(size_t)(k - 1) * TS] = 0.0;
      xm1 = xk;
      ym1 = yk;
      // slide the 8-point window: drop newy(k), append newy(k+8)
      if (k + 8 <= p2 + 7) {
#pragma unroll
        for (int s = 0; s < 3; s++) {
          wring[s][k & 7][threadIdx.x] = (k + 8 >= p2 + 4) ? khN[s] : NEWY(s, k + 8);  // high pads
        }
      }
    }
  }
  if (ypc_fail) P.ws_i[n] |= FL_VT_YPCFAIL;
}

// SIGS over the finished tables (tension_module.f90:314-781): intervals whose tension
// is known without iterating are stored at once, the others are queued for the wave.
// Whole waves stay converged (ballots); a particle whose YPC1 failed is skipped: TSPSI
// returned IER = -4 before SIGS, so its tensions stay 0 and SigErr = 0.
__global__ __launch_bounds__(256) void k_vt_sigs(const StepParams P) {
  __shared__ VtScratch vsc[4];
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int wave_first = n - lane;
  VtScratch& q = vsc[threadIdx.x >> 6];
  q.failed[lane] = 0;
  const int fl = (n < P.numpar) ? P.ws_i[n] : 0;
  const bool act = (fl & FL_ACT) != 0;
  const bool ypc_fail = (fl & FL_VT_YPCFAIL) != 0;
  const int na = act ? n : 0;
  const int p2 = 4 * P.ws;
  const size_t TS = 64, toff = (size_t)(na >> 6) * (size_t)p2 * 64 + (size_t)(na & 63);
  const double* tx = P.vt_x + toff; const double* ty = P.vt_y + toff; const double* typ = P.vt_yp + toff;
  double* tsg = P.vt_sig + toff;
  int ntask = 0;  // wave-uniform
  {
    double x0 = 0.0, y0 = 0.0, p0 = 0.0;
    const bool run = act && !ypc_fail;
    if (act && ypc_fail)
      for (int k = 0; k < p2 - 1; k++) tsg[(size_t)k * TS] = 0.0;
    if (run) { x0 = tx[0]; y0 = ty[0]; p0 = typ[0]; }
#pragma unroll 1
    for (int k0 = 0; k0 < p2 - 1; k0 += 4) {
      // the table comes back from HBM: 12 loads in flight per lane, then the arithmetic
      double xs[4], ys[4], ps[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int kk = (k0 + j + 1 < p2) ? (k0 + j + 1) : (p2 - 1);
        xs[j] = ys[j] = ps[j] = 0.0;
        if (run) { xs[j] = tx[(size_t)kk * TS]; ys[j] = ty[(size_t)kk * TS]; ps[j] = typ[(size_t)kk * TS]; }
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (k0 + j < p2 - 1) {
          if (ntask > LT_VT_CAP - 64) {
            vt_run_tasks(q, ntask, lane, wave_first, P.vt_sig, p2);
            ntask = 0;
          }
          double S = 0.0, sg = 0.0;
          bool need = false;
          if (run) {
            S = (ys[j] - y0) / (xs[j] - x0);
            need = !sigs_eval_known(S, p0, ps[j], sg);
            if (!need) tsg[(size_t)(k0 + j) * TS] = sg;
          }
          const unsigned long long m = __ballot(need);
          if (need) {
            const int idx = ntask + __popcll(m & ((1ull << lane) - 1ull));
            q.S[idx] = S; q.S1[idx] = p0; q.S2[idx] = ps[j];
            q.where[idx] = (unsigned short)(k0 + j);
            q.lane[idx] = (unsigned char)lane;
          }
          ntask += __popcll(m);
          x0 = xs[j]; y0 = ys[j]; p0 = ps[j];
        }
      }
    }
  }
  vt_run_tasks(q, ntask, lane, wave_first, P.vt_sig, p2);
  if (act && q.failed[lane]) P.ws_i[n] = fl | FL_VT_SIGERR;
}

// :282-335 random displacement model, idt/2 sub-steps of 2 s
struct __attribute__((packed, aligned(4))) RawWords4 { uint32_t a, b, c, d; };
__global__ __launch_bounds__(256) void k_vt_walk(const StepParams P) {
  __shared__ uint2 rbuf[8][256];
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= P.numpar) return;
  const int fl = P.ws_i[n];
  if (!(fl & FL_ACT)) return;
  const bool ypc_fail = (fl & FL_VT_YPCFAIL) != 0;
  const int sigerr = (fl & FL_VT_SIGERR) ? 1 : 0;
  const size_t N = (size_t)P.numpar;
  double* wsd = P.ws_d + n;
  const int p2 = 4 * P.ws;
  const size_t TS = 64, toff = (size_t)(n >> 6) * (size_t)p2 * 64 + (size_t)(n & 63);
  const double* tx = P.vt_x + toff; const double* ty = P.vt_y + toff; const double* typ = P.vt_yp + toff;
  const double* tsg = P.vt_sig + toff;
  const double P_depth = wsd[(size_t)WS_PDEPTH * N], P_zc = wsd[(size_t)WS_PZ1 * N];
  const double P_zetac = wsd[(size_t)WS_ZETA1 * N];
  // :282-335 random displacement model, idt/2 sub-steps of 2 s
  const double background = (double)1.0E-6f, deltat = 2.0;
  const int loop = P.idt / (int)deltat;
  // this particle's window of the serial stream: HTurb's 4 draws come first (:1087,1098)
  unsigned long long di = P.rng_base + (unsigned long long)(unsigned)P.rank[P.orig[n]] +
                          (P.HTurbOn ? 4ull : 0ull);
  double ParZc = P_zc;
  int il = 1;
  VtInterval ci;
  ci.I = 0; ci.X1 = ci.X2 = ci.Y1 = ci.Y2 = ci.P1 = ci.P2 = ci.SG = 0.0;
  const bool cached = (sigerr == 0) && !ypc_fail;
#pragma unroll 1
  for (int i = 1; i <= loop; i++) {
    // The particle's raw words are consecutive in the ring: 16 of them (8 sub-steps,
    // one 64-B line) are fetched at a time with four 16-byte loads and parked in this
    // thread's LDS column; word-by-word loads re-fetched the line for every sub-step.
    const int sub = (i - 1) & 7;
    if (sub == 0) {
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
    }
    const uint2 rw = rbuf[sub][threadIdx.x];
    const uint32_t raw1 = rw.x, raw2 = rw.y;
    di += 2;
    double Kprimec = 0.0;
    if (!(ParZc < P_depth || ParZc > P_zetac)) {
      if (cached) {
        vt_seek(ci, ParZc, p2, tx, ty, typ, tsg, TS);
        Kprimec = hpval_interval(ParZc, ci.X1, ci.X2, ci.Y1, ci.Y2, ci.P1, ci.P2, ci.SG);
      } else if (sigerr == 0) {
        Kprimec = hval_table<true>(ParZc, p2, tx, ty, typ, tsg, TS, il);
      } else { double yy; linint_n(tx, ty, p2, (int)TS, ParZc, yy, Kprimec); }
    }
    const double KprimeZc = -1.0 * Kprimec * deltat;
    const double Z3rdc = ParZc + 0.5 * KprimeZc;
    double KH3rdc;
    if (Z3rdc < P_depth || Z3rdc > P_zetac) {
      KH3rdc = background;
    } else {
      if (cached) {
        vt_seek(ci, Z3rdc, p2, tx, ty, typ, tsg, TS);
        KH3rdc = hval_interval(Z3rdc, ci.X1, ci.X2, ci.Y1, ci.Y2, ci.P1, ci.P2, ci.SG);
      } else if (sigerr == 0) {
        KH3rdc = hval_table<false>(Z3rdc, p2, tx, ty, typ, tsg, TS, il);
      } else { double mm; linint_n(tx, ty, p2, (int)TS, Z3rdc, KH3rdc, mm); }
      if (KH3rdc < background) KH3rdc = background;
    }
    // norm(), norm_module.f90:25-39
    const double dev1 = ((double)ltmt_temper(raw1) + 0.5) / 4294967296.0;
    const double dev2 = ((double)ltmt_temper(raw2) + 0.5) / 4294967296.0;
    const double DEV = sqrt(-2.0 * log(dev1)) * cos(2.0 * P.PI * dev2);
    // (2/r * KH3rd * deltat)**0.5 with r = 1
    ParZc = ParZc + KprimeZc + DEV * sqrt(2.0 / 1.0 * KH3rdc * deltat);
  }
  wsd[(size_t)WS_TURBV * N] = P_zc - ParZc;  // :342
  if (sigerr != 0) atomicAdd(&P.counters[CNT_SIGERR], 1ull);
}

__global__ __launch_bounds__(256) void k_post(const StepParams P) {
#ifdef LT_POST_TIMING
  __shared__ unsigned long long ptim[32];
  if (threadIdx.x < 32) ptim[threadIdx.x] = 0;
  __syncthreads();
  unsigned long long tprev_ = __builtin_readcyclecounter();
#endif
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned c_err = 0, c_brute = 0, c_pipslow = 0;
  __shared__ EleScratch esc[4];
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
      // HTurb, hor_turb_module.f90:29-50: X deviate before Y; VTurb's TurbV from k_vturb
      double TurbHx = 0.0, TurbHy = 0.0, TurbV = 0.0;
      if (P.HTurbOn) {
        const unsigned long long di = P.rng_base + (unsigned long long)(unsigned)P.rank[P.orig[n]];
        const double devX = draw_norm(P, di), devY = draw_norm(P, di + 2);
        const double amp = sqrt(2.0 / 1.0 * P.ConstantHTurb * P.idt);  // (2/r*KM*idt)**0.5
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
              const double theta = atan(cv / cu);
              if (cu > 0.0) { XBehav = Hdistance * cos(theta); YBehav = Hdistance * sin(theta); }
              if (cu < 0.0) { XBehav = -1.0 * Hdistance * cos(theta); YBehav = -1.0 * Hdistance * sin(theta); }
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
#ifdef LT_POST_TIMING
      { const unsigned long long t_=__builtin_readcyclecounter(); if ((threadIdx.x&63)==0) ptim[(threadIdx.x>>6)*8+0] += t_-tprev_; tprev_=__builtin_readcyclecounter(); }
#endif
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
#ifdef LT_POST_TIMING
      { const unsigned long long t_=__builtin_readcyclecounter(); if ((threadIdx.x&63)==0) ptim[(threadIdx.x>>6)*8+1] += t_-tprev_; tprev_=__builtin_readcyclecounter(); }
#endif
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
  // Part 2: setEle :1317
  bool okr2, oku2, okv2;
  wave_set_ele(P, esc[threadIdx.x >> 6], want_ele, eX, eY, er, eu, ev, rn0, un0, vn0, okr2, oku2, okv2);
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
      if (P.ErrorFlag == 2) { P.dead[n] = 1; keep_pn = true; }
      else if (P.ErrorFlag == 3) { P.oob[n] = 1; keep_pn = true; }
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
#ifdef LT_POST_TIMING
  { const unsigned long long t_=__builtin_readcyclecounter(); if ((threadIdx.x&63)==0) ptim[(threadIdx.x>>6)*8+4] += t_-tprev_; }
  __syncthreads();
  if (threadIdx.x < 8) atomicAdd(&g_tim[threadIdx.x], ptim[threadIdx.x] + ptim[8+threadIdx.x] + ptim[16+threadIdx.x] + ptim[24+threadIdx.x]);
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
    if (threadIdx.x == 0) P.counters[CNT_ACTIVE] = tot;
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

// ------------------------------------------------------ slab upload
// staging (node,level) column-major doubles -> [node][level][slot] SlabT.
// flags[0] is set when a value is not exactly representable in SlabT.
template <typename SlabT>
__global__ void k_slab_scatter(const double* __restrict__ src, SlabT* __restrict__ dst, int nodes,
                               int L, int slot, int* flags) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)nodes * L;
  if (idx >= total) return;
  // idx enumerates the destination (node-major, level fastest) so that the
  // strided writes of one slot stay within a few cache lines per wave
  size_t node = idx / L;
  int k = (int)(idx - node * L);
  double v = src[node + (size_t)nodes * k];
  SlabT c = (SlabT)v;
  if ((double)c != v && !(v != v)) atomicOr(flags, 1);
  dst[idx * LT_SLOTS + slot] = c;
}

template <typename SlabT>
__global__ void k_zeta_scatter(const double* __restrict__ src, SlabT* __restrict__ dst, int nodes,
                               int slot, int* flags) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)nodes) return;
  double v = src[idx];
  SlabT c = (SlabT)v;
  if ((double)c != v && !(v != v)) atomicOr(flags, 1);
  dst[idx * 4 + slot] = c;
}

// One field of a ROMS record (packed window [level][j][i], host type HostT) into a
// physical slot of the slab: mask multiply in double as hydrodynamic_module.f90:1371-1403,
// conversion to the storage type (checked lossless), node = (j-1)*n1 + i.
template <typename SlabT, typename HostT>
__global__ void k_rom_scatter(const HostT* __restrict__ src, SlabT* __restrict__ dst,
                              const int* __restrict__ mask, int n1, int i1, int wi, int j1, int wj, int L,
                              int pslot, int* flags) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t cells = (size_t)wi * wj;
  if (idx >= cells * L) return;
  // idx enumerates the destination order (cell-major, level fastest)
  const size_t cell = idx / L;
  const int k = (int)(idx - cell * L);
  const int jj = (int)(cell / wi), ii = (int)(cell - (size_t)jj * wi);
  const size_t node = (size_t)(j1 - 1 + jj) * n1 + (size_t)(i1 - 1 + ii);
  double v = (double)src[cell + cells * k];
  if (mask) v = v * (double)mask[node];
  const SlabT c = (SlabT)v;
  if ((double)c != v && !(v != v)) atomicOr(flags, 1);
  dst[(node * L + k) * LT_SLOTS + pslot] = c;
}

// xyz(numpar,3) column-major in the REFERENCE particle order <- par(:,1:3)
__global__ void k_pack_positions(const double* __restrict__ par, const int* __restrict__ orig,
                                 double* __restrict__ dst, int numpar) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= numpar) return;
  const size_t N = (size_t)numpar;
  const int o = orig[i];
  dst[o] = par[i];
  dst[N + o] = par[N + i];
  dst[2 * N + o] = par[2 * N + i];
}

// act flag of every particle in REFERENCE order (input of the rank scan)
// The conversion loop of writeOutput (LTRANS.f90:1719-1723): dx2lon / dy2lat
// (conversion_module.f90:322-378), getStatus (behavior_module.f90:554-574), and the columns
// the writers take, in the REFERENCE particle order.  cols: lon, lat, z, age, status
// (doubles, stride numpar); hits: hitBottom, hitLand (ints, stride numpar).
__global__ void k_output(const StepParams P, double Earth_Radius, int Spherical, double lonmin, double latmin,
                         double* __restrict__ cols, int* __restrict__ hits, int reset_hits) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.numpar) return;
  const size_t N = (size_t)P.numpar;
  const int o = P.orig[i];
  const double x = P.par[(size_t)pX * N + i], y = P.par[(size_t)pY * N + i];
  const double RCF = 180.0 / P.PI;
  double lon, lat;
  if (Spherical) {
    const double la = y * 180.0 / (Earth_Radius * P.PI) + latmin;
    lon = x * 180.0 / (Earth_Radius * P.PI * cos(la / RCF)) + lonmin;
    lat = y * RCF / Earth_Radius + latmin;
  } else {
    lon = x / Earth_Radius * RCF;
    lat = 2.0 * RCF * (atan(exp(y / Earth_Radius)) - P.PI / 4.0);
  }
  int status = P.P_behave[i];
  if (P.dead[i]) status = -1;
  if (P.settlementon && P.settle[i]) status = -2;
  if (P.OpenOceanBoundary && P.oob[i]) status = -3;
  cols[o] = lon;
  cols[N + o] = lat;
  cols[2 * N + o] = P.par[(size_t)pZ * N + i];
  cols[3 * N + o] = P.par[(size_t)pAge * N + i];
  cols[4 * N + o] = (double)status;
  hits[o] = P.hitBottom[i];
  hits[N + o] = P.hitLand[i];
  if (reset_hits) { P.hitBottom[i] = 0; P.hitLand[i] = 0; }
}

// (the value scanned is the number of deviates the particle will draw: the turbulence draws
// plus what k_beh_pre decided for the behaviour)
__global__ void k_flag_scatter(const int* __restrict__ ws_i, const int* __restrict__ orig,
                               int* __restrict__ flag, int numpar, int turb_draws) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= numpar) return;
  const int f = ws_i[i];
  flag[orig[i]] = (f & FL_ACT) ? turb_draws + ((f & FL_BEH_DIRECTED) ? 1 : 0) + ((f & FL_BEH_RANDOM) ? 2 : 0) : 0;
}

// number of particles that pass the gates and the first setEle (= draw RNG)
__global__ void k_count_active(const StepParams P, unsigned long long* out) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned c = 0;
  if (n < P.numpar) {
    const size_t N = (size_t)P.numpar;
    const double* par = P.par + n;
    bool track = !(P.ix2 <= par[(size_t)pDOB * N]);
    if (track) {
      double age = par[(size_t)pAge * N] + (double)(float)P.idt;
      bool settled = P.settlementon ? (P.settle[n] != 0) : false;
      bool deadn = P.dead[n] != 0;
      if (P.mortality && age >= P.P_deadage[n] && !(P.settlementon && settled)) deadn = true;
      if (P.settlementon && settled) track = false;
      if (track && P.mortality && deadn) track = false;
      if (track && P.OpenOceanBoundary && P.oob[n] != 0) track = false;
    }
    if (track) {
      double X = par[(size_t)pX * N], Y = par[(size_t)pY * N];
      int er = P.P_r_element[n], eu = P.P_u_element[n], ev = P.P_v_element[n];
      int4 nd;
      bool ok = find_element<true>(P.rnode, P.RE, P.rAdj, X, Y, er, nd);
      ok = find_element<false>(P.unode, P.UE, P.uAdj, X, Y, eu, nd) && ok;
      ok = find_element<false>(P.vnode, P.VE, P.vAdj, X, Y, ev, nd) && ok;
      c = ok ? 1u : 0u;
    }
  }
  // per block partial sums, then few atomics spread over 64 slots (one counter hit by
  // every wave serialises at the L2); the host adds the slots
  __shared__ unsigned wsum[4];
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned s = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (s) atomicAdd(out + (blockIdx.x & 63) * 8, (unsigned long long)s);
  }
}

}  // namespace ltd
