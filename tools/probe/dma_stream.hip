// What rate does an LDS-DMA weight stream reach per CU?  (r04: the token-stationary kernels of this library see 20 - 35 GB/s per CU in their chunk
// loops; MI355X_MICROARCH.md quotes 66 - 73 for an L2-resident gather.)  One workgroup per CU streams a buffer in chunks through a ring of LDS slots,
// one barrier per chunk, nothing else: variations = waves per workgroup, ring depth, chunk size, shared vs per-CU source.
// hipcc --offload-arch=gfx950 -O3 tools/probe/dma_stream.hip -o tools/probe/dma_stream && tools/probe/dma_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int NW, int SLOTS>
__global__ __launch_bounds__(NW * 64, 1) void stream_kernel(const char* __restrict__ w, long long wbytes, long long per_cu_stride, int chb, int nch, int* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const char* base = w + (long long)blockIdx.x * per_cu_stride;
  const int pieces = chb / 1024;                               // 1-KiB pieces per chunk
  auto issue = [&](int j) {
    const long long off = ((long long)j * chb) % wbytes;
    for (int f = wv; f < pieces; f += NW)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off + f * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void*)(smem + (j % SLOTS) * chb + f * 1024), 16, 0, 0);
  };
  for (int j = 0; j < SLOTS - 1 && j < nch; ++j) issue(j);
  int acc = 0;
  for (int j = 0; j < nch; ++j) {
    // wait for this wave's pieces of chunk j: everything but the younger chunks' (SLOTS - 2 chunks of ceil(pieces / NW) instructions)
    if (SLOTS == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (SLOTS == 3) { if (j + 1 < nch) { if ((pieces + NW - 1) / NW <= 5 && pieces % NW == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (j + SLOTS - 1 < nch) issue(j + SLOTS - 1);
    acc += *reinterpret_cast<const int*>(smem + (j % SLOTS) * chb + (tid * 16) % chb);     // touch the chunk (one ds_read per lane)
  }
  if (acc == 0x7fffffff) sink[0] = acc;
}

template <int NW, int SLOTS>
void run(const char* w, long long wbytes, long long stride, int chb, int nch, int* sink, const char* what) {
  const int lds = SLOTS * chb;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_kernel<NW, SLOTS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((stream_kernel<NW, SLOTS>), dim3(256), dim3(NW * 64), lds, 0, w, wbytes, stride, chb, nch, sink);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((stream_kernel<NW, SLOTS>), dim3(256), dim3(NW * 64), lds, 0, w, wbytes, stride, chb, nch, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double per_cu = (double)chb * nch;
  printf("%-46s waves %d slots %d chunk %3d KB: %7.1f us / launch, %6.1f GB/s per CU, %5.2f TB/s chip (%s)\n", what, NW, SLOTS, chb / 1024, ms * 1e3,
         per_cu / (ms * 1e-3) / 1e9, per_cu * 256 / (ms * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
}

int main() {
  const long long W = 2654208;                                  // one stage-3 fc1 weight matrix in the packed format: 72 chunks of 37 KB
  char* w; int* sink;
  hipMalloc(&w, 256ll * (4ll << 20)); hipMalloc(&sink, 64);
  hipMemset(w, 1, 256ll * (4ll << 20));
  const int nch = 72 * 8;                                       // 8 passes over the matrix
  for (int chb : {37 * 1024, 20 * 1024, 64 * 1024}) {
    const long long wb = (long long)(W / chb) * chb;
    run<8, 2>(w, wb, 0, chb, nch, sink, "every CU streams the SAME matrix (L2)");
    run<8, 3>(w, wb, 0, chb, nch, sink, "every CU streams the SAME matrix (L2)");
    run<4, 2>(w, wb, 0, chb, nch, sink, "every CU streams the SAME matrix (L2)");
    run<4, 3>(w, wb, 0, chb, nch, sink, "every CU streams the SAME matrix (L2)");
    run<8, 2>(w, wb, 4ll << 20, chb, nch, sink, "each CU its OWN 2.6 MB matrix (680 MB: HBM / MALL)");
    run<8, 3>(w, wb, 4ll << 20, chb, nch, sink, "each CU its OWN 2.6 MB matrix (680 MB: HBM / MALL)");
  }
  return 0;
}
