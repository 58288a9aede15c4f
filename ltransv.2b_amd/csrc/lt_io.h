// lt_io.h -- slab upload / scatter, packing for the all-gather and for the output writers, draw counts.
// Part of lt_kernels.h (included inside namespace ltd); see the notes there.
#pragma once

// ------------------------------------------------------ slab upload
// staging (node,level) column-major doubles -> [node][level][slot] SlabT (or [node][slot][level]
// when levfast: the velocity fields, lt_device.h).
// flags[0] is set when a value is not exactly representable in SlabT.
template <typename SlabT>
__global__ void k_slab_scatter(const double* __restrict__ src, SlabT* __restrict__ dst, int nodes,
                               int L, int slot, int* flags, int levfast) {
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
  if (levfast) dst[(node * LT_SLOTS + slot) * L + k] = c;
  else dst[idx * LT_SLOTS + slot] = c;
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
                              int pslot, int* flags, int levfast) {
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
  if (levfast) dst[(node * LT_SLOTS + pslot) * L + k] = c;
  else dst[(node * L + k) * LT_SLOTS + pslot] = c;
}

// xyz(numpar,3) column-major in the REFERENCE particle order <- par(:,1:3)
// (ld = distance between the columns of dst: numpar, or the padded shard length of the all-gather)
__global__ void k_pack_positions(const double* __restrict__ par, const int* __restrict__ orig,
                                 double* __restrict__ dst, int numpar, size_t ld) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= numpar) return;
  const size_t N = (size_t)numpar;
  const int o = orig[i];
  dst[o] = par[i];
  dst[ld + o] = par[N + i];
  dst[2 * ld + o] = par[2 * N + i];
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
    lon = x * 180.0 / (Earth_Radius * P.PI * lt_cos_g(la / RCF)) + lonmin;
    lat = y * RCF / Earth_Radius + latmin;
  } else {
    lon = x / Earth_Radius * RCF;
    lat = 2.0 * RCF * (lt_atan_g(exp_core_g(y / Earth_Radius)) - P.PI / 4.0);
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

// ------------------------------------------------------ re-sort of the internal particle order
// dst[c*n + i] = src[c*n + idx[i]] for the columns of one per-particle array
template <typename T>
__global__ void k_permute(const T* __restrict__ src, T* __restrict__ dst, const int* __restrict__ idx, int n,
                          int cols) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int j = idx[i];
  for (int c = 0; c < cols; c++) dst[(size_t)c * n + i] = src[(size_t)c * n + j];
}
// the inverse, for handing an array out in the reference's particle order: dst[c*n + orig[i]] = src[c*n + i]
// (orig[i] = reference index of internal slot i)
// every per-particle array of the engine through the same index in ONE launch (the re-sort: 23 arrays were 23
// launches of k_permute and 23 copies back), into one staging buffer; k_copy_back_multi returns them.
#define LT_PERM_MAX 32
struct PermDesc { const void* src; void* dst; int elem, cols; };   // elem: 4 or 8 bytes
struct PermList { int nd; PermDesc d[LT_PERM_MAX]; };
__global__ __launch_bounds__(256) void k_permute_multi(const PermList L, const int* __restrict__ idx, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int j = idx[i];
  for (int k = 0; k < L.nd; k++) {
    const PermDesc D = L.d[k];
    if (D.elem == 8) {
      const double* s = (const double*)D.src; double* t = (double*)D.dst;
      for (int c = 0; c < D.cols; c++) t[(size_t)c * n + i] = s[(size_t)c * n + j];
    } else {
      const int* s = (const int*)D.src; int* t = (int*)D.dst;
      for (int c = 0; c < D.cols; c++) t[(size_t)c * n + i] = s[(size_t)c * n + j];
    }
  }
}
// (dst of the list -> src of the list, element for element)
__global__ __launch_bounds__(256) void k_copy_back_multi(const PermList L, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int k = 0; k < L.nd; k++) {
    const PermDesc D = L.d[k];
    if (D.elem == 8) {
      double* s = (double*)D.src; const double* t = (const double*)D.dst;
      for (int c = 0; c < D.cols; c++) s[(size_t)c * n + i] = t[(size_t)c * n + i];
    } else {
      int* s = (int*)D.src; const int* t = (const int*)D.dst;
      for (int c = 0; c < D.cols; c++) s[(size_t)c * n + i] = t[(size_t)c * n + i];
    }
  }
}
template <typename T>
__global__ void k_unpermute(const T* __restrict__ src, T* __restrict__ dst, const int* __restrict__ orig, int n,
                            int cols) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int j = orig[i];
  for (int c = 0; c < cols; c++) dst[(size_t)c * n + j] = src[(size_t)c * n + i];
}
// sort key of every particle: the tile key of its rho element (lt_set_grid)
__global__ void k_tile_keys(const int* __restrict__ pr, const int* __restrict__ tile_key, int* __restrict__ keys, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) keys[i] = tile_key[pr[i] - 1];
}
__global__ void k_iota(int* __restrict__ v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = i;
}
