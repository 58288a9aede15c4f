// lt_init_host.h -- the host side of the init-time geometry calls (lt_locate_particles, lt_create_polyspecs):
// plain C++ index arithmetic with no device call in it, kept apart from lt_engine.hip so that it can also be
// compiled by g++ with -fsanitize=address,undefined (tests/csrc/host_san.cpp, tests/test_sanitizers.py).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace ltih {

#ifndef LT_LOC_CAP
#define LT_LOC_CAP 8   // (lt_init.h: the candidate elements k_locate lists per particle)
#endif

// uniform bins over the elements' bounding boxes (lt_init.h: EleBins), ids ascending in every bin
struct BinGeom { double x0, y0, inv; int ncx, ncy; };
inline void build_ele_bins(const std::vector<double>& nx, const std::vector<double>& ny, const std::vector<int32_t>& E,
                           int elements, BinGeom& B, std::vector<int>& ptr, std::vector<int>& ids) {
  double xmin = 1e300, xmax = -1e300, ymin = 1e300, ymax = -1e300;
  std::vector<double> bb((size_t)elements * 4);
  for (int i = 0; i < elements; i++) {
    double x0 = 1e300, x1 = -1e300, y0 = 1e300, y1 = -1e300;
    for (int k = 0; k < 4; k++) {
      const int nd = E[(size_t)i * 4 + k] - 1;
      x0 = std::min(x0, nx[nd]); x1 = std::max(x1, nx[nd]);
      y0 = std::min(y0, ny[nd]); y1 = std::max(y1, ny[nd]);
    }
    bb[(size_t)i * 4] = x0; bb[(size_t)i * 4 + 1] = x1; bb[(size_t)i * 4 + 2] = y0; bb[(size_t)i * 4 + 3] = y1;
    xmin = std::min(xmin, x0); xmax = std::max(xmax, x1); ymin = std::min(ymin, y0); ymax = std::max(ymax, y1);
  }
  const double wx = std::max(xmax - xmin, 1e-300), wy = std::max(ymax - ymin, 1e-300);
  double cell = std::sqrt(wx * wy / std::max(elements, 1));
  if (!(cell > 0.0)) cell = std::max(wx, wy);
  while ((wx / cell + 1.0) * (wy / cell + 1.0) > 4.0e6) cell *= 2.0;
  B.x0 = xmin; B.y0 = ymin; B.inv = 1.0 / cell;
  B.ncx = (int)std::floor(wx / cell) + 1;
  B.ncy = (int)std::floor(wy / cell) + 1;
  // the same expression as k_locate's, so that bin(min) <= bin(x) <= bin(max) for min <= x <= max
  auto cx = [&](double x) { double f = std::floor((x - B.x0) * B.inv); return (int)std::min(std::max(f, 0.0), (double)(B.ncx - 1)); };
  auto cy = [&](double y) { double f = std::floor((y - B.y0) * B.inv); return (int)std::min(std::max(f, 0.0), (double)(B.ncy - 1)); };
  const size_t ncell = (size_t)B.ncx * B.ncy;
  ptr.assign(ncell + 1, 0);
  for (int pass = 0; pass < 2; pass++) {
    std::vector<int> cur;
    if (pass) {
      for (size_t c = 0; c < ncell; c++) ptr[c + 1] += ptr[c];
      ids.assign((size_t)ptr[ncell], 0);
      cur.assign(ptr.begin(), ptr.end() - 1);
    }
    for (int i = 0; i < elements; i++) {
      const int a0 = cx(bb[(size_t)i * 4]), a1 = cx(bb[(size_t)i * 4 + 1]);
      const int b0 = cy(bb[(size_t)i * 4 + 2]), b1 = cy(bb[(size_t)i * 4 + 3]);
      for (int b = b0; b <= b1; b++)
        for (int a = a0; a <= a1; a++) {
          const size_t c = (size_t)a + (size_t)b * B.ncx;
          if (pass) ids[(size_t)cur[c]++] = i + 1; else ptr[c + 1]++;
        }
    }
  }
}

// a scan of all elements (gridcell without checkele): the first element that ends the scan, else
// the last one that set P_ele (gridcell_kind 2), 0 if none
inline int full_scan_result(const int* codes, int c) {
  int res = 0;
  for (int t = 0; t < c; t++) {
    if ((codes[t] & 3) == 1) return codes[t] >> 2;
    res = codes[t] >> 2;
  }
  return res;
}

// The particle loop of setEle_all (hydrodynamic_module.f90:1536-1676) over the candidate lists k_locate made
// (codes[g][k*LT_LOC_CAP + t] = element << 2 | gridcell's kind, cnt[g][k] of them, ascending): only the ORDER of
// the tests depends on the previous particle.  out[g][k] = the element; error / p as the reference's optional
// arguments (0 / 0: every particle was found).
inline void replay_set_ele_all(int n, const std::vector<int> (&codes)[3], const std::vector<int> (&cnt)[3],
                               const std::vector<int> (&adj)[3], std::vector<int32_t> (&out)[3], int& error, int& p) {
  for (int g = 0; g < 3; g++) out[g].assign((size_t)n, 0);
  error = 0; p = 0;
  if (n >= 1) {
    for (int g = 0; g < 3; g++) {  // :1553-1583, no early exit between the three grids
      out[g][0] = full_scan_result(&codes[g][0], cnt[g][0]);
      if (out[g][0] == 0) { error = g + 1; p = 1; }
    }
  }
  for (int k = 1; k < n && error == 0; k++) {
    for (int g = 0; g < 3; g++) {
      const int* cd = &codes[g][(size_t)k * LT_LOC_CAP];
      const int c = cnt[g][k];
      int res = 0;
      if (c == 1) {
        res = cd[0] >> 2;  // the only element that contains it, whichever test finds it
      } else if (c > 1) {
        const int* row = &adj[g][(size_t)(out[g][k - 1] - 1) * 10];
        bool done = false;
        for (int i = 0; i < 10 && !done; i++) {
          const int a = row[i];
          if (a == 0) { res = full_scan_result(cd, c); done = true; break; }
          for (int t = 0; t < c; t++)
            if ((cd[t] >> 2) == a) { res = a; done = true; break; }
        }
      }
      if (res == 0) { error = g + 1; p = k + 1; break; }
      out[g][k] = res;
    }
  }
}

// The serial scans of createPolySpecs (settlement_module.f90:277-289, 393-405, 410-468): first row and number of
// edge points of every polygon / hole id, and the holes of every polygon.  polys(pedges,5), holes(hedges,6)
// column-major.  Returns 0, or 1 / 2 with bad_id set when a polygon / hole id lies outside its declared range.
inline int polyspecs_host(const double* polys, int pedges, int minpolyid, int maxpolyid, const double* holes, int hedges,
                          int minholeid, int maxholeid, std::vector<int32_t>& ps, std::vector<int32_t>& hs,
                          std::vector<int32_t>& php, std::vector<int32_t>& phi, long& bad_id) {
  const int npoly = maxpolyid - minpolyid + 1;
  const int nhole = (hedges > 0) ? maxholeid - minholeid + 1 : 0;
  auto POLYS = [&](int i, int k) { return polys[((size_t)i - 1) + (size_t)pedges * (k - 1)]; };
  auto HOLES = [&](int i, int k) { return holes[((size_t)i - 1) + (size_t)hedges * (k - 1)]; };
  for (int i = 1; i <= pedges; i++) {
    const long id = lround(POLYS(i, 1));
    if (id < minpolyid || id > maxpolyid) { bad_id = id; return 1; }
  }
  for (int i = 1; i <= hedges; i++) {
    const long id = lround(HOLES(i, 1));
    if (id < minholeid || id > maxholeid) { bad_id = id; return 2; }
  }
  ps.assign((size_t)npoly * 2, 0);
  hs.assign((size_t)std::max(nhole, 1) * 2, 0);
  auto specs = [](std::vector<int32_t>& sp, int n, int nrow, int minid, auto ID) {
    int count = 1;
    sp[(size_t)(lround(ID(1)) - minid)] = 1;
    for (int i = 2; i <= nrow; i++) {
      if (ID(i) != ID(i - 1)) {
        sp[(size_t)(lround(ID(i)) - minid)] = i;
        sp[(size_t)(lround(ID(i - 1)) - minid) + n] = count;
        count = 1;
      } else if (i == nrow) {
        sp[(size_t)(lround(ID(i)) - minid) + n] = count + 1;
      } else {
        count = count + 1;
      }
    }
  };
  specs(ps, npoly, pedges, minpolyid, [&](int i) { return POLYS(i, 1); });
  if (hedges > 0) specs(hs, nhole, hedges, minholeid, [&](int i) { return HOLES(i, 1); });
  // holes of every polygon :410-468
  php.assign((size_t)npoly + 1, 0);
  phi.clear();
  for (int id = minpolyid; id <= maxpolyid; id++) {
    int count = 0;
    const int first = ps[(size_t)(id - minpolyid)];
    if (hedges > 0 && first != 0) {
      for (int j = 1; j <= hedges; j++)
        if ((j == 1 || HOLES(j, 1) != HOLES(j - 1, 1)) && HOLES(j, 6) == POLYS(first, 1)) {
          count++;
          phi.push_back((int32_t)lround(HOLES(j, 1)));
        }
    }
    php[(size_t)(id - minpolyid) + 1] = php[(size_t)(id - minpolyid)] + count;
  }
  return 0;
}

}  // namespace ltih
