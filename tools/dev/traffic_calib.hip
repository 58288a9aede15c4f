// Calibration of the FETCH_SIZE / WRITE_SIZE counters (rocprofv3 --pmc) on gfx950 with kernels whose HBM-side
// bytes are known, for the access shapes the engine's kernels use.  Every kernel touches each cache line of its
// buffer at most once, buffers are 1 GiB (beyond L2 and the 256 MB Infinity Cache), so the bytes that must cross
// the L2's memory side = the bytes of the lines touched.  Run under rocprofv3 (profiles/tools/calibrate.sh) and
// compare the counters per kernel with the `expect` column this program prints.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// (a) wide coalesced reads: 16 B per lane, consecutive lanes
__global__ void cal_read_wide16(const float4* __restrict__ src, float* out, size_t n4) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.f;
  for (; i < n4; i += (size_t)gridDim.x * blockDim.x) { float4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 12345.678f) out[0] = acc;
}
// (b) coalesced 8-byte reads
__global__ void cal_read_coal8(const double* __restrict__ src, double* out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  double acc = 0.0;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) acc += src[i];
  if (acc == 12345.678) out[0] = acc;
}
// (c) gathers: every lane reads `bytes` (4, 8, 16 or 32) from its OWN 64-byte line (line index = a permutation
// of the lane's global index that keeps neighbouring lanes on neighbouring lines, like element-sorted particles)
template <int BYTES>
__global__ void cal_gather(const char* __restrict__ src, float* out, size_t nlines, int off) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.f;
  for (; i < nlines; i += (size_t)gridDim.x * blockDim.x) {
    const char* p = src + i * 64 + off;
    if (BYTES == 4) acc += *(const float*)p;
    else if (BYTES == 8) { const float2 v = *(const float2*)p; acc += v.x + v.y; }
    else if (BYTES == 16) { struct __attribute__((packed, aligned(4))) F4 { float a, b, c, d; }; const F4 v = *(const F4*)p; acc += v.a + v.b + v.c + v.d; }
    else { const double4 v = *(const double4*)p; acc += (float)(v.x + v.y + v.z + v.w); }
  }
  if (acc == 12345.678f) out[0] = acc;
}
// (d) 128-byte-line variant: one 16-byte read per 128 bytes
__global__ void cal_gather16_per128(const char* __restrict__ src, float* out, size_t nlines) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.f;
  for (; i < nlines; i += (size_t)gridDim.x * blockDim.x) { const float4 v = *(const float4*)(src + i * 128); acc += v.x + v.y + v.z + v.w; }
  if (acc == 12345.678f) out[0] = acc;
}
// (e) writes: coalesced 8 B, and 32-byte records one per 64-byte line, and 32-byte records back to back per lane
__global__ void cal_write_coal8(double* dst, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (double)i;
}
__global__ void cal_write_rec32_stream(double4* dst, size_t nrec_per_lane, size_t lanes) {
  // lane l writes its own contiguous run of records (like k_vt_table's [particle][knot] table)
  const size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= lanes) return;
  double4* p = dst + l * nrec_per_lane;
  for (size_t k = 0; k < nrec_per_lane; k++) p[k] = make_double4((double)k, 1.0, 2.0, 3.0);
}

int main() {
  const size_t BYTES = (size_t)1 << 30;
  char* buf; float* out;
  CK(hipMalloc(&buf, BYTES)); CK(hipMalloc(&out, 64));
  CK(hipMemset(buf, 0, BYTES));
  const int B = 256, G = 256 * 16;
  printf("kernel expect_read_bytes expect_write_bytes\n");
  hipLaunchKernelGGL(cal_read_wide16, dim3(G), dim3(B), 0, 0, (const float4*)buf, out, BYTES / 16);
  printf("cal_read_wide16 %zu 0\n", BYTES);
  hipLaunchKernelGGL(cal_read_coal8, dim3(G), dim3(B), 0, 0, (const double*)buf, (double*)out, BYTES / 8);
  printf("cal_read_coal8 %zu 0\n", BYTES);
  hipLaunchKernelGGL(cal_gather<4>, dim3(G), dim3(B), 0, 0, buf, out, BYTES / 64, 20);
  printf("cal_gather<4> %zu 0   (64-byte lines; %zu if the fetch granule is 32 bytes)\n", BYTES, BYTES / 2);
  hipLaunchKernelGGL(cal_gather<8>, dim3(G), dim3(B), 0, 0, buf, out, BYTES / 64, 24);
  printf("cal_gather<8> %zu 0\n", BYTES);
  hipLaunchKernelGGL(cal_gather<16>, dim3(G), dim3(B), 0, 0, buf, out, BYTES / 64, 20);
  printf("cal_gather<16> %zu 0   (element-aligned 16-byte window, inside one 64-byte line)\n", BYTES);
  hipLaunchKernelGGL(cal_gather<32>, dim3(G), dim3(B), 0, 0, buf, out, BYTES / 64, 32);
  printf("cal_gather<32> %zu 0\n", BYTES);
  hipLaunchKernelGGL(cal_gather16_per128, dim3(G), dim3(B), 0, 0, buf, out, BYTES / 128);
  printf("cal_gather16_per128 %zu 0   (one 16-byte read per 128 bytes: %zu if whole 128-byte lines are fetched)\n", BYTES / 2, BYTES);
  hipLaunchKernelGGL(cal_write_coal8, dim3(G), dim3(B), 0, 0, (double*)buf, BYTES / 8);
  printf("cal_write_coal8 0 %zu\n", BYTES);
  hipLaunchKernelGGL(cal_write_rec32_stream, dim3((unsigned)((BYTES / 32 / 128 + B - 1) / B)), dim3(B), 0, 0, (double4*)buf, (size_t)128, BYTES / 32 / 128);
  printf("cal_write_rec32_stream 0 %zu\n", BYTES);
  CK(hipDeviceSynchronize());
  return 0;
}
