// host_san.cpp -- the host-only index arithmetic of the library (createBounds: lt_bounds_host.h; the host sides of
// lt_locate_particles and lt_create_polyspecs: lt_init_host.h) in a translation unit of its own, compiled by
// tests/test_sanitizers.py with g++ -fsanitize=address,undefined.  TEST INFRASTRUCTURE.
//
//   host_san bounds <in> <out>   in:  int32 nmask, vi, uj; then nmask x { int32 mask(vi*uj) }
//                                out: per mask int32 rc, nbounds, maxbound, maxisland, numislands, then (rc == 0) the
//                                     arrays bx by (maxbound) hx hy (maxisland) as doubles, hid (maxisland),
//                                     land (nbounds) as int32, bnd_x bnd_y (2*nbounds) as doubles
//                                (node coordinates: x_u(i,j) = i + 0.5, y_u = j; x_v = i, y_v = j + 0.5)
//   host_san init                a rectilinear element grid: bins, the setEle_all replay over candidate lists with
//                                particles on shared nodes and edges, the polyspecs scans; prints a checksum
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "lt_bounds_host.h"
#include "lt_init_host.h"

static int run_bounds(const char* in, const char* out) {
  FILE* f = fopen(in, "rb");
  FILE* g = fopen(out, "wb");
  if (!f || !g) return 2;
  int32_t hdr[3];
  if (fread(hdr, 4, 3, f) != 3) return 2;
  const int nmask = hdr[0], vi = hdr[1], uj = hdr[2], ui = vi - 1, vj = uj - 1;
  std::vector<double> xu((size_t)ui * uj), yu((size_t)ui * uj), xv((size_t)vi * vj), yv((size_t)vi * vj);
  for (int j = 0; j < uj; j++) for (int i = 0; i < ui; i++) { xu[(size_t)i + (size_t)ui * j] = i + 0.5; yu[(size_t)i + (size_t)ui * j] = j; }
  for (int j = 0; j < vj; j++) for (int i = 0; i < vi; i++) { xv[(size_t)i + (size_t)vi * j] = i; yv[(size_t)i + (size_t)vi * j] = j + 0.5; }
  const int cap = 2 * vi * uj;
  std::vector<int32_t> mask((size_t)vi * uj), hid(cap), land(cap);
  std::vector<double> bx(cap), by(cap), hx(cap), hy(cap), bnx(2 * (size_t)cap), bny(2 * (size_t)cap);
  for (int m = 0; m < nmask; m++) {
    if (fread(mask.data(), 4, mask.size(), f) != mask.size()) return 2;
    int32_t o[5] = {0, 0, 0, 0, 0};
    o[0] = ltb::create_bounds_arrays(vi, uj, mask.data(), xu.data(), yu.data(), xv.data(), yv.data(), cap, &o[1], &o[2], &o[3],
                                     &o[4], bx.data(), by.data(), hx.data(), hy.data(), hid.data(), bnx.data(), bny.data(),
                                     land.data());
    fwrite(o, 4, 5, g);
    if (o[0] != 0) continue;
    fwrite(bx.data(), 8, o[2], g); fwrite(by.data(), 8, o[2], g);
    fwrite(hx.data(), 8, o[3], g); fwrite(hy.data(), 8, o[3], g);
    fwrite(hid.data(), 4, o[3], g); fwrite(land.data(), 4, o[1], g);
    fwrite(bnx.data(), 8, 2 * (size_t)o[1], g); fwrite(bny.data(), 8, 2 * (size_t)o[1], g);
  }
  fclose(f);
  fclose(g);
  return 0;
}

static int run_init() {
  // a 9 x 7 rectilinear grid of unit elements, node (i, j) = i + 10 * j (1-based ids)
  const int ni = 9, nj = 7, nn = 10 * 8;
  std::vector<double> nx(nn), ny(nn);
  for (int j = 0; j < 8; j++) for (int i = 0; i < 10; i++) { nx[(size_t)i + 10 * j] = i; ny[(size_t)i + 10 * j] = j; }
  std::vector<int32_t> E;
  for (int j = 0; j < nj; j++)
    for (int i = 0; i < ni; i++) {
      const int a = i + 10 * j + 1;
      E.push_back(a); E.push_back(a + 1); E.push_back(a + 11); E.push_back(a + 10);
    }
  const int ne = ni * nj;
  ltih::BinGeom B;
  std::vector<int> ptr, ids;
  ltih::build_ele_bins(nx, ny, E, ne, B, ptr, ids);
  unsigned long long sum = (unsigned long long)B.ncx * 1000003ull + (unsigned long long)B.ncy;
  for (int v : ptr) sum = sum * 31 + (unsigned)v;
  for (int v : ids) sum = sum * 31 + (unsigned)v;
  // adjacency rows: the element itself, then its neighbours in ascending order, zero-filled
  std::vector<int> adj[3];
  for (int g = 0; g < 3; g++) {
    adj[g].assign((size_t)ne * 10, 0);
    for (int j = 0; j < nj; j++)
      for (int i = 0; i < ni; i++) {
        int* row = &adj[g][(size_t)(i + ni * j) * 10];
        int c = 0;
        row[c++] = i + ni * j + 1;
        for (int dj = -1; dj <= 1; dj++)
          for (int di = -1; di <= 1; di++)
            if ((di || dj) && i + di >= 0 && i + di < ni && j + dj >= 0 && j + dj < nj) row[c++] = (i + di) + ni * (j + dj) + 1;
      }
  }
  // candidate lists: particle k lies in element (k mod ne) + 1; every third one on an edge (two candidates), every
  // seventh on a node (four), one nowhere (count 0: the error path)
  const int n = 500;
  std::vector<int> codes[3], cnt[3];
  for (int g = 0; g < 3; g++) {
    codes[g].assign((size_t)n * LT_LOC_CAP, 0);
    cnt[g].assign((size_t)n, 0);
    for (int k = 0; k < n; k++) {
      const int e0 = (k * 7) % ne, i = e0 % ni, j = e0 / ni;
      int c = 0;
      auto add = [&](int ii, int jj, int kind) { if (ii >= 0 && ii < ni && jj >= 0 && jj < nj) codes[g][(size_t)k * LT_LOC_CAP + c++] = ((ii + ni * jj + 1) << 2) | kind; };
      if (k % 7 == 0) { add(i - 1, j - 1, 2); add(i, j - 1, 2); add(i - 1, j, 2); add(i, j, 1); }
      else if (k % 3 == 0) { add(i - 1, j, 2); add(i, j, 2); }
      else add(i, j, 1);
      cnt[g][k] = (g == 2 && k == n - 1) ? 0 : c;
    }
  }
  std::vector<int32_t> out[3];
  int error = 0, p = 0;
  ltih::replay_set_ele_all(n, codes, cnt, adj, out, error, p);
  for (int g = 0; g < 3; g++) for (int32_t v : out[g]) sum = sum * 31 + (unsigned)v;
  sum = sum * 31 + (unsigned)error; sum = sum * 31 + (unsigned)p;
  // polyspecs: four polygons (ids 3..6, one without edge points), two holes
  const int pedges = 11, hedges = 7;
  std::vector<double> polys((size_t)pedges * 5, 0.0), holes((size_t)hedges * 6, 0.0);
  const int pid[pedges] = {3, 3, 3, 3, 4, 4, 4, 6, 6, 6, 6};
  const int hidv[hedges] = {10, 10, 10, 11, 11, 11, 11}, hpoly[hedges] = {3, 3, 3, 6, 6, 6, 6};
  for (int i = 0; i < pedges; i++) polys[i] = pid[i];
  for (int i = 0; i < hedges; i++) { holes[i] = hidv[i]; holes[(size_t)i + (size_t)hedges * 5] = hpoly[i]; }
  std::vector<int32_t> ps, hs, php, phi;
  long bad = 0;
  int rc = ltih::polyspecs_host(polys.data(), pedges, 3, 6, holes.data(), hedges, 10, 11, ps, hs, php, phi, bad);
  for (auto* v : {&ps, &hs, &php, &phi}) for (int32_t x : *v) sum = sum * 31 + (unsigned)x;
  sum = sum * 31 + (unsigned)rc;
  polys[0] = 99.0;   // an id outside the declared range
  rc = ltih::polyspecs_host(polys.data(), pedges, 3, 6, holes.data(), hedges, 10, 11, ps, hs, php, phi, bad);
  sum = sum * 31 + (unsigned)rc + (unsigned long long)bad;
  printf("init checksum %llu error %d at %d polyspecs", sum, error, p);
  polys[0] = 3.0;
  ltih::polyspecs_host(polys.data(), pedges, 3, 6, holes.data(), hedges, 10, 11, ps, hs, php, phi, bad);
  for (int32_t x : ps) printf(" %d", x);
  printf(" holes");
  for (int32_t x : php) printf(" %d", x);
  printf("\n");
  return 0;
}

int main(int argc, char** argv) {
  if (argc == 4 && !strcmp(argv[1], "bounds")) return run_bounds(argv[2], argv[3]);
  if (argc == 2 && !strcmp(argv[1], "init")) return run_init();
  fprintf(stderr, "usage: host_san bounds <in> <out> | host_san init\n");
  return 2;
}
