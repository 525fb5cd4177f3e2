// Calibration: sustained dense f16 MFMA rate of this GPU (no memory traffic), for several launch lengths.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512, 1) void k(float* out, int iters, long long* clk) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  const long long t0 = wall_clock64();
  const long long s0 = clock64();
  for (int it = 0; it < iters; ++it) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
  }
  const long long s1 = clock64();
  const long long t1 = wall_clock64();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = s1 - s0; clk[1] = t1 - t0; }
}
int main() {
  float* out; long long* clk;
  hipMalloc(&out, 256 * 8 * 512 * 4); hipMalloc(&clk, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int its[] = {2000, 2000, 20000, 200000, 1000000, 2000, 20000};
  for (int iters : its) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256 * 4), dim3(512), 0, 0, out, iters, clk);   // 4 blocks per CU queued: 8 waves/CU resident at a time
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double flops = 256.0 * 4 * 8 * iters * 4.0 * 2 * 32 * 32 * 16;
    printf("iters %7d  %9.3f ms  %8.1f TFLOP/s   shader clocks/iter %.1f (ideal 4 MFMA x 32 = 128 x 2 waves per SIMD)  shader clock ~%.0f MHz (wall clock 100 MHz)\n",
           iters, ms, flops / ms / 1e9, (double)h[0] / iters, (double)h[0] / ((double)h[1] / 100.0));
  }
  return 0;
}
