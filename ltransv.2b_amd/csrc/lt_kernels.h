// lt_kernels.h -- HIP kernels of the LTRANS hot path for gfx950 (CDNA4).
//
// update_particles() (LTRANS.f90:707-1419) is run as a short sequence of kernels over the
// same particle index, one wavefront lane per particle, exchanging a per-particle workspace
// (SoA, stride numpar) in HBM so that each part gets its own register budget:
//   k_pre       gates, setEle, setInterp / getInterp, vertical clamps            (lt_rk.h)
//   k_beh_pre   the deterministic part of behave(), salinity cue, draw counts    (lt_rk.h)
//   k_rk        RK4 over find_currents / WCTS_ITPI, wave-cooperative             (lt_rk.h)
//   k_rk_slow   reference-shaped redo of the rare SigErr / degenerate cases      (lt_rk.h)
//   k_vt_table / k_vt_sigs / k_vt_walk_fast / k_vt_walk   VTurb                   (lt_vturb.h)
//   k_post      HTurb, behaviour, reflection, boundaries, setEle, settlement, commit (lt_post.h)
// plus the upload / download kernels of lt_io.h and the init-time geometry kernels of lt_init.h.  No MFMA: the path is gather + scalar fp64
// spline arithmetic with data-dependent iteration.
//
// Shape decisions that differ from the reference (results do not):
//   * interpolation weights are computed once per (stage, grid) instead of at each of the
//     144 interp() calls (hydrodynamic_module.f90:2527-2563 is a pure function of position
//     and element);
//   * s-level columns (LTRANS.f90:934-951) are never materialised: the level windows are
//     found by bisection on the (monotone) level depths, and the 4-level windows are
//     recomputed from hc*SC / CS tables held in LDS;
//   * boundary segments are visited through a uniform-grid index in reference order
//     (ascending segment id) instead of scanning all nbounds; the point-in-polygon tests
//     read a y-slab edge index and fall back to the full crossings algorithm whenever a
//     polygon vertex lies exactly on the particle's ray (the only case where the two differ);
//   * data-dependent iterations (SIGS' Newton / secant, the setEle neighbour walk) are
//     queued in per-wave LDS and executed by all lanes of the wave together;
//   * divisions that share a divisor share its refined reciprocal (lt_math.h: prep_div),
//     bit-identical to `/`.
#pragma once
#include "lt_device.h"
#include "lt_math.h"

namespace ltd {

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
       WS_MINPD, WS_MAXPD,         // minpartdepth / maxpartdepth of the RK stages (LTRANS.f90:981-987)
       WS_NDOUBLE };
// FL_BEH_DIRECTED / FL_BEH_RANDOM: what k_beh_pre decided the behaviour will draw (one deviate
// compared with WS_BTHR1; then two deviates, the first compared with WS_BTHR2)
enum { FL_ACT = 1, FL_KEEP_PN = 2, FL_SLOW = 4, FL_VT_YPCFAIL = 16, FL_VT_SIGERR = 32,
       FL_BEH_DIRECTED = 64, FL_BEH_RANDOM = 128, FL_ERR_SHIFT = 8 };

template <typename SlabT> struct SlabTraits;
template <> struct SlabTraits<float> { using zeta4 = float4; };
template <> struct SlabTraits<double> { using zeta4 = double4; };

#include "lt_grid.h"
#include "lt_rk.h"
#include "lt_vturb.h"   // (its kernels are split between the two translation units: LT_VT_DEFINE_A / _B)
#ifndef LT_KERNELS_RK_ONLY  // (lt_rk_kernel.hip: the kernels below are not templates and belong to lt_engine.hip)
#include "lt_post.h"
#include "lt_io.h"
#include "lt_init.h"
#endif

}  // namespace ltd
