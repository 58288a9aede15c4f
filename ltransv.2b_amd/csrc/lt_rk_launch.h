// lt_rk_launch.h -- what lt_engine.hip sees of the translation unit that holds k_rk (lt_rk_kernel.hip)
#pragma once
#include <hip/hip_runtime.h>

#ifndef LT_RK_WAVES
#define LT_RK_WAVES 2  // k_rk is held to 2 waves per SIMD (its LDS: 8 x 19.7 KB per compute unit)
#endif
#ifndef LT_RK_BLOCK
#define LT_RK_BLOCK 64   // one wave per block: a finished wave frees its slot at once (256-lane blocks: +3 % kernel time)
#endif

#ifndef LT_RK_XCD_CHUNK
// k_rk's workgroups go round-robin to the 8 XCDs; XCD x works through the chunks x, x+8, x+16, ... of this many
// workgroups of the (tile-sorted) particle order.  One contiguous eighth per XCD (0) fetches least (0.94 GB per
// launch against 1.11 with 64 and 1.26 with 32), but a region of the domain whose particles are more work than the
// others' then keeps one XCD busy after the rest have finished: 400 steps into the synthetic run the kernel was
// 2-3 % faster with chunks of 64 (3 % with 32), at equal time in the first steps.
#define LT_RK_XCD_CHUNK 64
#endif

namespace ltd {
struct StepParams;
// k_rk<float|double, Vtransform, LT_RK_WAVES, free slip> on `grid` workgroups of LT_RK_BLOCK lanes
void launch_k_rk(bool f32, int Vtransform, bool freeslip, unsigned grid, unsigned dynlds, hipStream_t stream, const StepParams& P);
// the VTurb kernels of that unit (lt_vturb.h, LT_VT_DEFINE_A), 256-lane workgroups over `blocks` x 256 particles
void launch_k_vt_table(bool f32, int Vtransform, bool fastdiv, unsigned blocks, hipStream_t stream, const StepParams& P);
void launch_k_vt_norm(unsigned blocks, hipStream_t stream, const StepParams& P);
void launch_k_vt_walk_fast(unsigned blocks, hipStream_t stream, const StepParams& P);
#ifdef LT_RK_TIMING
void rk_timing_fetch(unsigned long long* out16, bool reset);  // development aid (tools/dev/rk_sections.py)
#endif
}  // namespace ltd
