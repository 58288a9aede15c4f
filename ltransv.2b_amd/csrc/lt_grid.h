// lt_grid.h -- s-levels, element search (per lane and by the whole wave), point in polygon, boundary
// reflection, settlement test: the geometry the step kernels share.
// Part of lt_kernels.h (included inside namespace ltd); see the notes there.
#pragma once

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
// the same for k_rk: one 16-byte load per node at a 32-bit offset from the (uniform) base of the table, where
// the 8-byte-aligned struct reads above are two loads each at a 64-bit address formed per lane
LT_DEV Quad load_quad16(const void* nodes, int4 n, unsigned stride) {
  const char* b = (const char*)nodes;
  const double2 p0 = *(const double2*)(b + (unsigned)(n.x - 1) * stride), p1 = *(const double2*)(b + (unsigned)(n.y - 1) * stride);
  const double2 p2 = *(const double2*)(b + (unsigned)(n.z - 1) * stride), p3 = *(const double2*)(b + (unsigned)(n.w - 1) * stride);
  Quad q;
  q.x[0] = p0.x; q.y[0] = p0.y; q.x[1] = p1.x; q.y[1] = p1.y;
  q.x[2] = p2.x; q.y[2] = p2.y; q.x[3] = p3.x; q.y[3] = p3.y;
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
LT_HD int pip_edge(double x, double y, double x1, double y1, double x2, double y2) {
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

// the slab of y in a PipIndex (lt_device.h): k with ys[k] < y < ys[k+1], or -1 when y lies outside every edge's
// y-range (no edge can be crossed); `slow` when y equals a vertex ordinate
LT_HD int pip_slab(const PipIndex& ix, double y, bool& slow) {
  if (ix.m < 2 || !(y >= ix.ys[0])) return -1;          // (below the polygon, or NaN)
  double fc = floor((y - ix.y0) * ix.inv_h);
  const int c = (fc < 0.0) ? 0 : (fc >= (double)ix.ncell ? ix.ncell - 1 : (int)fc);
  // the largest k with ys[k] <= y lies in [gidx[c] - 1, gidx[c + 1] - 1]
  int lo = ix.gidx[c] - 1, hi = ix.gidx[c + 1] - 1;
  if (lo < 0) lo = 0;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (ix.ys[mid] <= y) lo = mid; else hi = mid - 1;
  }
  if (ix.ys[lo] == y) { slow = true; return -1; }
  return (lo < ix.m - 1) ? lo : -1;                      // (above the last vertex: outside)
}

// mbounds (boundary_module.f90:1515-1540) through the slab index; `slow` is
// set when a vertex shares the particle's y exactly (then the full algorithm
// must decide).
LT_HD bool pip_one_polygon(const PipIndex& ix, double x, double y, bool& slow) {
  const int k = pip_slab(ix, y, slow);
  if (k < 0) return false;
  int crossed = 0;
  for (int q = ix.ptr[k]; q < ix.ptr[k + 1]; q++) {
    const double4 e = ix.rec[q];
    int r = pip_edge(x, y, e.x, e.y, e.z, e.w);
    if (r == 2) return true;
    crossed += r;
  }
  return (crossed & 1) != 0;
}

LT_DEV bool mbounds_indexed(const StepParams& P, double x, double y, bool& slow) { return pip_one_polygon(P.pip_main, x, y, slow); }

// ibounds (boundary_module.f90:1548-1614): inside ANY island?
LT_HD bool pip_any_polygon(const PipIndex& ix, double x, double y, bool& slow) {
  const int k = pip_slab(ix, y, slow);
  if (k < 0) return false;
  int cur = -1, crossed = 0;
  for (int q = ix.ptr[k]; q < ix.ptr[k + 1]; q++) {
    const int id = ix.pid[q];
    if (id != cur) {
      if (crossed & 1) return true;
      cur = id;
      crossed = 0;
    }
    const double4 e = ix.rec[q];
    int r = pip_edge(x, y, e.x, e.y, e.z, e.w);
    if (r == 2) return true;
    crossed += r;
  }
  return (crossed & 1) != 0;
}

LT_DEV bool ibounds_indexed(const StepParams& P, double x, double y, bool& slow) { return pip_any_polygon(P.pip_isl, x, y, slow); }

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
