// lt_rk_kernel.hip -- k_rk (lt_rk.h) and three of the VTurb kernels (lt_vturb.h: k_vt_table, k_vt_norm,
// k_vt_walk_fast) in a translation unit of their own, because it is compiled with `-mllvm -disable-machine-licm` and
// the rest of the library is not.
//
// Pass B of k_rk (the Newton iterations of SIGS, lt_math.h: sigs_newton_w, exp_core, hval_interval_s) works with
// some twenty double-precision constants -- polynomial coefficients, the constants of libm's exp.  A double that
// is not an inline constant occupies a pair of scalar registers, and the machine-level loop-invariant code
// motion of the compiler moves their `s_mov`s out of every loop of the kernel: ~45 scalar registers held from
// the first instruction to the last.  Pass A then has no scalar registers left for its own lane masks and loop
// counters, which the register allocator parks in lanes of a vector register and fetches back with
// v_readlane -- a vector-ALU issue slot each, inside the slot loop the kernel spends its time in.  With the
// constants materialised where they are used (scalar-ALU instructions, which issue beside the other wave's
// vector instructions) k_rk issues 24.2k instead of 25.1k vector instructions per wave and step at 212
// instead of 256 vector registers, and takes 0.946 instead of 0.971 ms.
// The option was then measured on every kernel of the library: k_vt_walk_fast 526 -> 491 us (96 -> 80 registers, a
// sixth wave per SIMD), k_vt_norm 444 -> 437 (85 -> 38), k_vt_table -0.5 %: these three are defined here
// (LT_VT_DEFINE_A); k_vt_sigs +3 %, k_vt_walk +2.5 %, k_post 106 -> 128 us, k_pre slower: they stay in lt_engine.hip.
#include <hip/hip_runtime.h>

#include "../../include/ltrans_hip.h"
#include "lt_rk_launch.h"    // (first: LT_RK_BLOCK, LT_RK_XCD_CHUNK are used by the kernels)
#define LT_KERNELS_RK_ONLY
#define LT_VT_DEFINE_A
#include "lt_kernels.h"

namespace ltd {

template <typename SlabT, bool FS>
static void launch_k_rk_t(int Vtransform, unsigned grid, unsigned dynlds, hipStream_t stream, const StepParams& P) {
  switch (Vtransform) {
    case 1: hipLaunchKernelGGL((k_rk<SlabT, 1, LT_RK_WAVES, FS>), dim3(grid), dim3(LT_RK_BLOCK), dynlds, stream, P); break;
    case 2: hipLaunchKernelGGL((k_rk<SlabT, 2, LT_RK_WAVES, FS>), dim3(grid), dim3(LT_RK_BLOCK), dynlds, stream, P); break;
    default: hipLaunchKernelGGL((k_rk<SlabT, 3, LT_RK_WAVES, FS>), dim3(grid), dim3(LT_RK_BLOCK), dynlds, stream, P); break;
  }
}

// (FreeSlip runs take the instantiation with the substitution in its interpolations; every other run the plain one)
void launch_k_rk(bool f32, int Vtransform, bool freeslip, unsigned grid, unsigned dynlds, hipStream_t stream, const StepParams& P) {
  if (freeslip) {
    if (f32) launch_k_rk_t<float, true>(Vtransform, grid, dynlds, stream, P);
    else launch_k_rk_t<double, true>(Vtransform, grid, dynlds, stream, P);
  } else {
    if (f32) launch_k_rk_t<float, false>(Vtransform, grid, dynlds, stream, P);
    else launch_k_rk_t<double, false>(Vtransform, grid, dynlds, stream, P);
  }
}

template <typename SlabT>
static void launch_k_vt_table_t(int Vtransform, bool fastdiv, unsigned blocks, hipStream_t st, const StepParams& P) {
#define LT_LAUNCH_VT_TABLE(VT)                                                                         \
  do {                                                                                                \
    if (fastdiv) hipLaunchKernelGGL((k_vt_table<SlabT, VT, true>), dim3(blocks), dim3(256), 0, st, P); \
    else hipLaunchKernelGGL((k_vt_table<SlabT, VT, false>), dim3(blocks), dim3(256), 0, st, P);          \
  } while (0)
  switch (Vtransform) {
    case 1: LT_LAUNCH_VT_TABLE(1); break;
    case 2: LT_LAUNCH_VT_TABLE(2); break;
    default: LT_LAUNCH_VT_TABLE(3); break;
  }
#undef LT_LAUNCH_VT_TABLE
}
void launch_k_vt_table(bool f32, int Vtransform, bool fastdiv, unsigned blocks, hipStream_t stream, const StepParams& P) {
  if (f32) launch_k_vt_table_t<float>(Vtransform, fastdiv, blocks, stream, P);
  else launch_k_vt_table_t<double>(Vtransform, fastdiv, blocks, stream, P);
}
void launch_k_vt_norm(unsigned blocks, hipStream_t stream, const StepParams& P) {
  hipLaunchKernelGGL(k_vt_norm, dim3(blocks), dim3(256), 0, stream, P);
}
void launch_k_vt_walk_fast(unsigned blocks, hipStream_t stream, const StepParams& P) {
  hipLaunchKernelGGL(k_vt_walk_fast, dim3(blocks * (256 / LT_VT_WALKFAST_BLOCK)), dim3(LT_VT_WALKFAST_BLOCK), 0, stream, P);
}

#ifdef LT_RK_TIMING
void rk_timing_fetch(unsigned long long* out16, bool reset) {
  hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_tim), 128);
  if (reset) {
    const unsigned long long z[16] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_tim), z, 128);
  }
}
#endif

}  // namespace ltd

#ifdef LT_RK_WAVETIME
extern "C" int lt_diag_wavetime(unsigned long long* out40000) {
  hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(out40000, HIP_SYMBOL(ltd::g_wavetime), 40000 * 8);
}
extern "C" int lt_diag_wavecyc(unsigned long long* out20000) {
  hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(out20000, HIP_SYMBOL(ltd::g_wavecyc), 20000 * 8);
}
#endif
