// VALU issue rate of transcendental instructions on gfx950 (cycles per wave-instruction), measured with s_memtime around
// an unrolled dependent-free instruction stream.  hipcc --offload-arch=gfx950 -O3 trans_rate.hip -o trans_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP 64
#define STR(x) #x
#define BODY(INS)                                                                                  \
  for (int it = 0; it < iters; ++it) {                                                             \
    _Pragma("unroll") for (int r = 0; r < REP / 4; ++r) {                                          \
      asm volatile(INS " %0, %0\n" INS " %1, %1\n" INS " %2, %2\n" INS " %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); \
    }                                                                                              \
  }
template <int WHICH> __global__ void k(float* out, long long* cyc, int iters) {
  float a = threadIdx.x * 1e-3f + 1.f, b = a + 1.f, c = a + 2.f, d = a + 3.f;
  long long t0 = clock64();
  if (WHICH == 0) { BODY("v_exp_f32") }
  if (WHICH == 1) { BODY("v_rcp_f32") }
  if (WHICH == 2) { BODY("v_exp_f16") }
  if (WHICH == 3) { BODY("v_rcp_f16") }
  if (WHICH == 4) { BODY("v_mov_b32") }
  if (WHICH == 5) { BODY("v_sqrt_f32") }
  if (WHICH == 6) { BODY("v_log_f32") }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
  const char* names[] = {"v_exp_f32", "v_rcp_f32", "v_exp_f16", "v_rcp_f16", "v_mov_b32", "v_sqrt_f32", "v_log_f32"};
  const int iters = 1000;
  for (int waves = 1; waves <= 2; ++waves) {
    for (int w = 0; w < 7; ++w) {
      long long h = 0;
      dim3 g(1), b(64 * 4 * waves);   // `waves` waves per SIMD on one CU
      for (int rep = 0; rep < 2; ++rep) {
        switch (w) {
          case 0: hipLaunchKernelGGL(k<0>, g, b, 0, 0, out, cyc, iters); break;
          case 1: hipLaunchKernelGGL(k<1>, g, b, 0, 0, out, cyc, iters); break;
          case 2: hipLaunchKernelGGL(k<2>, g, b, 0, 0, out, cyc, iters); break;
          case 3: hipLaunchKernelGGL(k<3>, g, b, 0, 0, out, cyc, iters); break;
          case 4: hipLaunchKernelGGL(k<4>, g, b, 0, 0, out, cyc, iters); break;
          case 5: hipLaunchKernelGGL(k<5>, g, b, 0, 0, out, cyc, iters); break;
          case 6: hipLaunchKernelGGL(k<6>, g, b, 0, 0, out, cyc, iters); break;
        }
        hipDeviceSynchronize();
      }
      hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      printf("%d wave(s)/SIMD  %-10s  %.2f clock64 ticks per instruction per wave\n", waves, names[w], (double)h / (iters * REP) / waves);
    }
  }
  return 0;
}
