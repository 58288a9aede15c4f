// microbenchmark: issue rate of dependent fp64 FMA chains vs independent chains (ILP) vs waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP>
__global__ void k_chain(double* out, int iters, double a, double b) {
  double x[ILP];
#pragma unroll
  for (int i = 0; i < ILP; i++) x[i] = threadIdx.x * 1e-3 + i;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
#pragma unroll
      for (int i = 0; i < ILP; i++) x[i] = __builtin_fma(x[i], a, b);
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < ILP; i++) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP>
void run(int waves_per_simd, double* d) {
  const int iters = 2000;
  // one block = 64 threads = 1 wave; grid = 256 CUs * 4 SIMDs * W  (the dispatcher spreads waves over SIMDs)
  const int blocks = 256 * 4 * waves_per_simd;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_chain<ILP>, dim3(blocks), dim3(64), 0, 0, d, iters, 0.999, 0.001);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_chain<ILP>, dim3(blocks), dim3(64), 0, 0, d, iters, 0.999, 0.001);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double n_instr = (double)iters * 16 * ILP;  // per wave
  // cycles per wave-instruction per SIMD at 2.4 GHz: time * f / (instr per wave * waves per SIMD)
  printf("ILP %d waves/SIMD %d: %.3f ms, %.2f cycles per FMA per wave (@2.4GHz), SIMD issue interval %.2f cycles\n", ILP,
         waves_per_simd, ms, ms * 1e-3 * 2.4e9 / n_instr, ms * 1e-3 * 2.4e9 / (n_instr * waves_per_simd));
}
int main() {
  double* d; hipMalloc(&d, 256 * 4 * 8 * 64 * 8);
  for (int w = 1; w <= 4; w++) { run<1>(w, d); run<2>(w, d); run<4>(w, d); }
  return 0;
}
