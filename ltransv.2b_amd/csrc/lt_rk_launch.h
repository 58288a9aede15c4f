// lt_rk_launch.h -- what lt_engine.hip sees of the translation unit that holds k_rk (lt_rk_kernel.hip)
#pragma once
#include <hip/hip_runtime.h>

#ifndef LT_RK_WAVES
#define LT_RK_WAVES 2  // k_rk is held to 2 waves per SIMD (its LDS: 8 x 19.7 KB per compute unit)
#endif
#ifndef LT_RK_BLOCK
#define LT_RK_BLOCK 64   // one wave per block: a finished wave frees its slot at once (256-lane blocks: +3 % kernel time)
#endif

namespace ltd {
struct StepParams;
// k_rk<float|double, Vtransform, LT_RK_WAVES> on `grid` workgroups of LT_RK_BLOCK lanes
void launch_k_rk(bool f32, int Vtransform, unsigned grid, unsigned dynlds, hipStream_t stream, const StepParams& P);
// the VTurb kernels of that unit (lt_vturb.h, LT_VT_DEFINE_A), 256-lane workgroups over `blocks` x 256 particles
void launch_k_vt_table(bool f32, int Vtransform, bool fastdiv, unsigned blocks, hipStream_t stream, const StepParams& P);
void launch_k_vt_norm(unsigned blocks, hipStream_t stream, const StepParams& P);
void launch_k_vt_walk_fast(unsigned blocks, hipStream_t stream, const StepParams& P);
#ifdef LT_RK_TIMING
void rk_timing_fetch(unsigned long long* out16, bool reset);  // development aid (tools/dev/rk_sections.py)
#endif
}  // namespace ltd
