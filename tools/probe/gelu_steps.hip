// GeluPk2Steps (common.hpp) against gelu_fast_pk on random values: four instances stepped side by side, as tok_linear16's epilogue uses them.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I circuitvision_amd/csrc tools/probe/gelu_steps.hip -o tools/probe/gelu_steps
#include "common.hpp"
#include <vector>
#include <random>
__global__ void k(const float* __restrict__ in, uint32_t* __restrict__ ref, uint32_t* __restrict__ got, int n16) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n16) return;
  float a[16];
  for (int i = 0; i < 16; ++i) a[i] = in[t * 16 + i];
  for (int p = 0; p < 8; ++p) ref[t * 8 + p] = __builtin_bit_cast(uint32_t, gelu_fast_pk(a[2 * p], a[2 * p + 1]));
  uint32_t c1v = GeluPk2Steps::GELU_C1_H;
  asm volatile("" : "+v"(c1v));
  GeluPk2Steps g[4];
  for (int q = 0; q < 4; ++q) g[q].s0(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
  for (int q = 0; q < 4; ++q) g[q].s1(c1v);
  for (int q = 0; q < 4; ++q) g[q].s2();
  for (int q = 0; q < 4; ++q) g[q].s3();
  for (int q = 0; q < 4; ++q) { got[t * 8 + 2 * q] = g[q].ra; got[t * 8 + 2 * q + 1] = g[q].rb; }
}
int main() {
  const int n16 = 1 << 16;
  std::vector<float> h(n16 * 16);
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 3.f);
  for (auto& v : h) v = nd(rng);
  float* d; uint32_t *r, *g;
  hipMalloc(&d, h.size() * 4); hipMalloc(&r, n16 * 8 * 4); hipMalloc(&g, n16 * 8 * 4);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n16 / 256), dim3(256), 0, 0, d, r, g, n16);
  std::vector<uint32_t> hr(n16 * 8), hg(n16 * 8);
  hipMemcpy(hr.data(), r, hr.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hg.data(), g, hg.size() * 4, hipMemcpyDeviceToHost);
  long bad = 0;
  for (size_t i = 0; i < hr.size(); ++i) if (hr[i] != hg[i]) { if (bad < 5) printf("pair %zu: ref %08x got %08x (in %f %f)\n", i, hr[i], hg[i], h[2 * i], h[2 * i + 1]); ++bad; }
  printf("GeluPk2Steps vs gelu_fast_pk: %ld of %zu packed pairs differ\n", bad, hr.size());
  return bad != 0;
}
