// lt_libm.h -- exp, log, log10, pow(x, 0.5), sin, cos, sincos and atan as the CPU run of LTRANS gets them.
//
// The reference calls the host libm for these, and its results branch on them: SIGS' Newton
// iteration stops or falls into a cycle (-> SigErr -> the whole profile is read with linint) on the
// last bit of exp (tension_module.f90:531-580); a vertical random-walk displacement that is 1 ulp off
// moves a particle of a few metres depth by an ulp, which moves the knots of the next step's
// splines.  A faithful device-library routine is right to < 1 ulp but differs from the host's
// result in 0.1-3 % of the evaluations, so a long run drifts away from the CPU run bit by bit.
// The functions here restate, operation for operation, what the host libm of the reference's
// platform executes (glibc >= 2.28 on an FMA-capable x86-64, read from its published algorithms and
// checked against the library itself on millions of arguments, on the host and on the device):
//   lt_exp / exp_core   table-driven exp, N = 128, degree-5 polynomial, FMA variant
//   lt_log              table-driven log, N = 128, FMA variant (with its separate path near 1)
//   lt_log10            the fdlibm form y*log10_2lo + ivln10*log(m) + y*log10_2hi on that log (no FMA)
//   lt_pow_half         pow(x, 0.5) = exp(0.5 * log(x)) with the log in double-double, FMA variant --
//                       what the reference's (..)**0.5 calls; it is NOT always sqrt(x) (0.08 % differ)
//   lt_cos / lt_sin     the table-driven sin / cos of the IBM Accurate Mathematical Library, FMA variant
//   lt_sincos           the same algorithm compiled without FMA: a C or Fortran compiler of the GCC
//                       family merges cos(a) and sin(a) of one argument into one sincos call, which
//                       glibc provides in a single, non-FMA build
//   lt_atan             the table-driven atan of the same library, FMA variant (13.5 KB table, always
//                       read from global memory: behaviour 7's heading and y2lat only)
// Tables: lt_exp_table.h (generated from its definition) and lt_libm_tables.h.  On the device they
// are copied to LDS by the kernels that evaluate the functions (libm_tables_load<MASK>()); on the
// host (tests only: the same source compiled with gcc) they are static arrays.
#pragma once
#include "lt_exp_table.h"
#include "lt_libm_tables.h"

#if defined(__HIPCC__)
#define LT_LM __device__ __forceinline__
#define LT_LM_ASU(d) ((unsigned long long)__double_as_longlong(d))
#define LT_LM_ASD(u) (__longlong_as_double((long long)(u)))
#else
#include <math.h>
#include <string.h>
#define LT_LM static inline
static inline unsigned long long lt_lm_asu(double d) { unsigned long long u; memcpy(&u, &d, 8); return u; }
static inline double lt_lm_asd(unsigned long long u) { double d; memcpy(&d, &u, 8); return d; }
#define LT_LM_ASU(d) lt_lm_asu(d)
#define LT_LM_ASD(u) lt_lm_asd(u)
#endif
#define LT_LM_FMA(a, b, c) __builtin_fma((a), (b), (c))
// every fused operation of these functions is written as LT_LM_FMA; nothing else may be contracted
// (the build passes -ffp-contract=off as well)
#if defined(__clang__)
#pragma clang fp contract(off)
#endif

struct ExpPair { unsigned long long tail, sbits; };
struct LogPair { double invc, logc; };
struct PowLog { double invc, logc, logctail; };

#if defined(__HIPCC__)
__device__ const ExpPair g_exp_tab[128] __attribute__((aligned(16))) = LT_EXP_TABLE_INIT;
__device__ const LogPair g_log_tab[128] __attribute__((aligned(16))) = LT_LOG_TAB_INIT;
__device__ const PowLog g_pow_tab[128] = LT_POW_TAB_INIT;
__device__ const double g_sincos_tab[440] __attribute__((aligned(16))) = LT_SINCOS_TAB_INIT;
__device__ const double g_atan_tab[241][7] = LT_ATAN_TAB_INIT;
#define LT_ATAN_TAB g_atan_tab
__shared__ ExpPair s_exp_tab[128];
__shared__ LogPair s_log_tab[128];
__shared__ PowLog s_pow_tab[128];
__shared__ double s_sincos_tab[440];
// which tables a kernel needs (it must load every table whose functions it evaluates)
enum { LM_EXP = 1, LM_LOG = 2, LM_POW = 4, LM_SINCOS = 8 };
template <unsigned MASK>
__device__ __forceinline__ void libm_tables_load() {
  if (MASK & LM_EXP)
    for (int i = threadIdx.x; i < 128; i += blockDim.x) s_exp_tab[i] = g_exp_tab[i];
  if (MASK & LM_LOG)
    for (int i = threadIdx.x; i < 128; i += blockDim.x) s_log_tab[i] = g_log_tab[i];
  if (MASK & LM_POW)
    for (int i = threadIdx.x; i < 128; i += blockDim.x) s_pow_tab[i] = g_pow_tab[i];
  if (MASK & LM_SINCOS)
    for (int i = threadIdx.x; i < 440; i += blockDim.x) s_sincos_tab[i] = g_sincos_tab[i];
  __syncthreads();
}
__device__ __forceinline__ void exp_table_load() { libm_tables_load<LM_EXP>(); }
#else
static const struct ExpPair lt_h_exp_tab[128] = LT_EXP_TABLE_INIT;
static const struct LogPair lt_h_log_tab[128] = LT_LOG_TAB_INIT;
static const struct PowLog lt_h_pow_tab[128] = LT_POW_TAB_INIT;
static const double lt_h_sincos_tab[440] = LT_SINCOS_TAB_INIT;
static const double lt_h_atan_tab[241][7] = LT_ATAN_TAB_INIT;
#define LT_ATAN_TAB lt_h_atan_tab
#endif

#if defined(__HIPCC__)
// f(): tables in LDS (the kernels that evaluate a function many times per particle: the Newton
// iterations of the splines, the sub-steps of the vertical random walk); f_g(): tables read from
// global memory through the caches (a few evaluations per particle-step: no LDS, no copy)
#define LT_LM_NAME(f) f
#define LT_EXP_TAB s_exp_tab
#define LT_LOG_TAB s_log_tab
#define LT_POW_TAB s_pow_tab
#define LT_SINCOS_TAB s_sincos_tab
#include "lt_libm_impl.h"
#undef LT_LM_NAME
#undef LT_EXP_TAB
#undef LT_LOG_TAB
#undef LT_POW_TAB
#undef LT_SINCOS_TAB
#define LT_LM_NAME(f) f##_g
#define LT_EXP_TAB g_exp_tab
#define LT_LOG_TAB g_log_tab
#define LT_POW_TAB g_pow_tab
#define LT_SINCOS_TAB g_sincos_tab
#include "lt_libm_impl.h"
#else
#define LT_LM_NAME(f) f
#define LT_EXP_TAB lt_h_exp_tab
#define LT_LOG_TAB lt_h_log_tab
#define LT_POW_TAB lt_h_pow_tab
#define LT_SINCOS_TAB lt_h_sincos_tab
#include "lt_libm_impl.h"
#endif

