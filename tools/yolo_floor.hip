// Floor of a launch chain: the same NUMBER of launches, moving the same bytes per launch, doing nothing but load -> (SiLU) -> store.
// hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/yolo_floor.hip -o tools/libyolofloor.so     (tools/yolo_floor.py builds and drives it)
#include <hip/hip_runtime.h>
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));

template <int MODE>     // 0: copy; 1: SiLU on every element (v_exp + v_rcp per value, as the conv epilogues pay)
__global__ __launch_bounds__(256) void floor_kernel(const f16x8* __restrict__ in, long long n_in, f16x8* __restrict__ out, long long n_out) {
  const long long stride = (long long)gridDim.x * blockDim.x, t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  // every output piece is fed by ceil(n_in / n_out) input pieces (a conv reads more than it writes, or less): all bytes of both sides move once
  const long long per = (n_in + n_out - 1) / n_out;
  for (long long o = t; o < n_out; o += stride) {
    f16x8 acc = {};
    for (long long k = 0; k < per; ++k) {
      const long long i = o + k * n_out;
      if (i < n_in) { const f16x8 v = in[i];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += v[e]; }
    }
    if (MODE == 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float x = (float)acc[e]; acc[e] = (f16)(x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * x))); }
    }
    out[o] = acc;
  }
}

extern "C" int floor_chain(const long long* in_bytes, const long long* out_bytes, int n, const void* src, void* dst, int mode, int blocks_cap, hipStream_t s) {
  for (int i = 0; i < n; ++i) {
    const long long ni = in_bytes[i] / 16, no = out_bytes[i] / 16 > 0 ? out_bytes[i] / 16 : 1;
    long long blocks = (no + 255) / 256;
    if (blocks > blocks_cap) blocks = blocks_cap;
    if (mode == 1) hipLaunchKernelGGL(floor_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, (const f16x8*)src, ni, (f16x8*)dst, no);
    else hipLaunchKernelGGL(floor_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, s, (const f16x8*)src, ni, (f16x8*)dst, no);
  }
  return (int)hipGetLastError();
}
