// lt_init.h -- init-time geometry on the device.
//
// What LTRANS computes once, serially, before the first step and what costs minutes there
// at the 1024x512 grid / millions of particles (a scan of all elements with gridcell for
// every particle that is not next to its predecessor, rho_elements x pedges gridcell calls
// for the habitat lists):
//   the adjacency lists of initGrid   hydrodynamic_module.f90:588-645   k_adjacency
//   setEle_all                        hydrodynamic_module.f90:1536-1676 k_locate (+ host chain)
//   createPolySpecs                   settlement_module.f90:293-386     k_elepolys
// All three are independent tests of one element against one point / polygon, so they are
// one thread per element or per particle; the results are the reference's, element for
// element (tests/test_gpu_init_geometry.py against the oracle's literal loops).
// createBounds (boundary_module.f90:82-1215) is a serial walk along the land/sea edge whose
// OUTPUT ORDER is the walk order, O(boundary points) work: it stays on the host.
// Part of lt_kernels.h (included inside namespace ltd).
#pragma once

// ------------------------------------------------------------------ adjacency
// E: node ids of the elements; adj: (elements,10) column-major, zeroed by the caller.
__global__ __launch_bounds__(256) void k_adjacency(const int4* __restrict__ E, int elements, int row,
                                                   int* __restrict__ adj, int* __restrict__ overflow) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= elements) return;
  const int4 a = E[i];
  const long long lo64 = (long long)i - ((long long)row + 2), hi64 = (long long)i + row + 2;
  const int lo = (int)(lo64 < 0 ? 0 : lo64), hi = (int)(hi64 > elements - 1 ? elements - 1 : hi64);
  int m = 1;
  adj[i] = i + 1;
  for (int j = lo; j <= hi; j++) {
    if (j == i) continue;
    const int4 b = E[j];
    const bool share = a.x == b.x || a.x == b.y || a.x == b.z || a.x == b.w ||
                       a.y == b.x || a.y == b.y || a.y == b.z || a.y == b.w ||
                       a.z == b.x || a.z == b.y || a.z == b.z || a.z == b.w ||
                       a.w == b.x || a.w == b.y || a.w == b.z || a.w == b.w;
    if (share) {
      if (m >= 10) {  // the reference would write r_Adjacent(i,11)
        atomicExch(overflow, i + 1);
        return;
      }
      adj[(size_t)i + (size_t)elements * m] = j + 1;
      m++;
    }
  }
}

// ------------------------------------------------------------------ setEle_all
// Uniform bins over the elements' bounding boxes (built on the host, ids ascending in every
// bin).  gridcell rejects a point outside the element's bounding box before anything else
// (:49-59), and the bin of a coordinate is a monotone function of it, so the bin of the
// point lists every element gridcell can accept.
struct EleBins {
  const int* ptr;  // (ncx*ncy+1)
  const int* ids;  // 1-based element ids
  double x0, y0, inv;
  int ncx, ncy;
};
#define LT_LOC_CAP 8

// For every particle: the elements that contain it, ascending, as (id << 2) | kind
// (gridcell_kind), and their number.
template <bool RHO>
__global__ __launch_bounds__(256) void k_locate(const void* __restrict__ nodes, const int4* __restrict__ EL,
                                                const EleBins B, const double* __restrict__ X,
                                                const double* __restrict__ Y, int n, int* __restrict__ codes,
                                                int* __restrict__ cnt) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const double x = X[k], y = Y[k];
  const double fx = floor((x - B.x0) * B.inv), fy = floor((y - B.y0) * B.inv);
  const int cx = (int)fmin(fmax(fx, 0.0), (double)(B.ncx - 1));
  const int cy = (int)fmin(fmax(fy, 0.0), (double)(B.ncy - 1));
  const size_t cell = (size_t)cx + (size_t)cy * B.ncx;
  const int beg = B.ptr[cell], end = B.ptr[cell + 1];
  int c = 0;
  for (int t = beg; t < end; t++) {
    const int ele = B.ids[t];
    const int4 nd = EL[ele - 1];
    const Quad q = RHO ? load_quad_rho((const RhoNode*)nodes, nd) : load_quad_xy((const XYNode*)nodes, nd);
    const int kind = gridcell_kind(q.x, q.y, x, y);
    if (kind != 0) {
      if (c < LT_LOC_CAP) codes[(size_t)k * LT_LOC_CAP + c] = (ele << 2) | kind;
      c++;
    }
  }
  cnt[k] = c;
}

// ------------------------------------------------------------------ createPolySpecs
// The element loop of settlement_module.f90:293-386, one thread per rho element; the scan
// over the pedges edge points keeps the reference's order and its "same id as the last one
// added" skip, so the list of an element is the reference's list.  pass 0 counts, pass 1
// writes the ids at ptr[i].  polys is (pedges,5) column-major: id, centre x,y, edge x,y.
__global__ __launch_bounds__(256) void k_elepolys(const RhoNode* __restrict__ nodes, const int4* __restrict__ RE,
                                                  int rho_elements, const double* __restrict__ polys, int pedges,
                                                  const int* __restrict__ polyspecs, const double* __restrict__ maxbdis,
                                                  int minpolyid, int npoly, const int* __restrict__ ptr,
                                                  int* __restrict__ out, int pass) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rho_elements) return;
  const Quad q = load_quad_rho(nodes, RE[i]);
  const double* pid = polys;
  const double* pcx = polys + (size_t)pedges;
  const double* pcy = polys + (size_t)pedges * 2;
  const double* pex = polys + (size_t)pedges * 3;
  const double* pey = polys + (size_t)pedges * 4;
  int count = 0;
  double tail = 0.0;
  int* dst = pass ? out + ptr[i] : nullptr;
  for (int j = 0; j < pedges; j++) {
    const double idj = pid[j];
    if (count != 0 && idj == tail) continue;
    bool add = gridcell_kind(q.x, q.y, pex[j], pey[j]) != 0;
    if (!add && (j == pedges - 1 || idj != pid[j + 1])) {
      const int id = (int)llrint(idj);
      const double mb = maxbdis[id - minpolyid];
      bool near = false;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const double ddx = q.x[k] - pcx[j], ddy = q.y[k] - pcy[j];
        if (sqrt(ddx * ddx + ddy * ddy) < mb) near = true;
      }
      if (near) {
        const int first = polyspecs[id - minpolyid], np = polyspecs[(id - minpolyid) + npoly];
        for (int k = 0; k < 4 && !add; k++)
          add = inpoly_full(q.x[k], q.y[k], np, pex + (first - 1), pey + (first - 1), false);
      }
    }
    if (add) {
      tail = (double)(int)llrint(idj);
      if (pass) dst[count] = (int)llrint(idj);
      count++;
    }
  }
  if (!pass) out[i] = count;
}

