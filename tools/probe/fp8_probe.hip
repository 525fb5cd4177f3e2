// Probe of the gfx950 block-scaled fp8 MFMA and the fp8 conversions (run on the GPU box): which lane / byte of the A and B operands of
// v_mfma_scale_f32_32x32x64_f8f6f4 is which (row, k) / (k, col), what the e8m0 scale operand does, how v_cvt_pk_fp8_f32 and
// v_cvt_scalef32_pk_fp8_f16 encode.  Build: hipcc --offload-arch=gfx950 -O2 tools/probe/fp8_probe.hip -o tools/probe/fp8_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef short s2 __attribute__((ext_vector_type(2)));

__global__ void mm(const v8i* a, const v8i* b, v16f* c, const int* sa, const int* sb) {
  v16f acc = {0};
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0, sa[threadIdx.x], 0, sb[threadIdx.x]);
  c[threadIdx.x] = acc;
}
__global__ void cv(const float* x, int* o, const _Float16* xh, float sc) {
  int r = 0;
  r = __builtin_amdgcn_cvt_pk_fp8_f32(x[0], x[1], r, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(x[2], x[3], r, true);
  o[0] = r;
  s2 q = {0, 0};
  h2 v = {xh[0], xh[1]};
  q = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(q, v, sc, false);
  o[1] = (unsigned short)q[0] | ((unsigned)(unsigned short)q[1] << 16);
  h2 w = {xh[2], xh[3]};
  q = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(q, w, sc, true);
  o[2] = (unsigned short)q[0] | ((unsigned)(unsigned short)q[1] << 16);
}
static float e4m3(unsigned char v) {          // OCP e4m3fn
  int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float r = e == 0 ? ldexpf(m / 8.f, -6) : (e == 15 && m == 7) ? NAN : ldexpf(1.f + m / 8.f, e - 7);
  return s ? -r : r;
}
static unsigned char enc(int v) {               // small non-negative integers 0..15 exactly
  for (int c = 0; c < 256; ++c) if (e4m3((unsigned char)c) == (float)v) return (unsigned char)c;
  return 0;
}
int main() {
  unsigned char A[64][32], B[64][32];
  srand(1);
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) { A[l][j] = enc(rand() % 4); B[l][j] = enc(rand() % 4); }
  int sa[64], sb[64];
  for (int l = 0; l < 64; ++l) { sa[l] = 127; sb[l] = 127; }
  v8i *da, *db; v16f* dc; int *dsa, *dsb;
  hipMalloc(&da, 64 * 32); hipMalloc(&db, 64 * 32); hipMalloc(&dc, 64 * 64); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256);
  float C[64][16];
  auto run = [&]() {
    hipMemcpy(da, A, 2048, hipMemcpyHostToDevice); hipMemcpy(db, B, 2048, hipMemcpyHostToDevice);
    hipMemcpy(dsa, sa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mm, dim3(1), dim3(64), 0, 0, da, db, dc, dsa, dsb);
    hipMemcpy(C, dc, 4096, hipMemcpyDeviceToHost);
  };
  run();
  // hypothesis: A lane (r = l & 31, h = l >> 5) byte j <-> A[row r][k = (h, j)]; B lane (r, h) byte j <-> B[k = (h, j)][col r]; D standard
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int g = 0; g < 16; ++g) {
    const int col = l & 31, row = (g & 3) + 8 * (g >> 2) + 4 * (l >> 5);
    float ref = 0;
    for (int h = 0; h < 2; ++h) for (int j = 0; j < 32; ++j) ref += e4m3(A[row + 32 * h][j]) * e4m3(B[col + 32 * h][j]);
    if (ref != C[l][g]) { if (bad < 5) printf("mismatch lane %d reg %d: got %g want %g\n", l, g, C[l][g], ref); ++bad; }
  }
  printf("LAYOUT same-position pairing, A lane&31 = row, B lane&31 = col, D standard 32x32: %s (%d mismatches)\n", bad ? "NO" : "YES", bad);
  // scales: lanes 0..31 of A (k block 0 of every row) x 2 (e8m0 128); expectation: only the h = 0 half of each row's sum doubles
  for (int l = 0; l < 32; ++l) sa[l] = 128;
  run();
  bad = 0;
  for (int l = 0; l < 64; ++l) for (int g = 0; g < 16; ++g) {
    const int col = l & 31, row = (g & 3) + 8 * (g >> 2) + 4 * (l >> 5);
    float ref = 0;
    for (int h = 0; h < 2; ++h) for (int j = 0; j < 32; ++j) ref += (h == 0 ? 2.f : 1.f) * e4m3(A[row + 32 * h][j]) * e4m3(B[col + 32 * h][j]);
    if (ref != C[l][g]) { if (bad < 5) printf("scale mismatch lane %d reg %d: got %g want %g\n", l, g, C[l][g], ref); ++bad; }
  }
  printf("SCALE_A per lane = per (row, 32-k block), e8m0 byte 0 (127 = 1, 128 = 2): %s (%d mismatches)\n", bad ? "NO" : "YES", bad);
  for (int l = 0; l < 64; ++l) { sa[l] = 127; sb[l] = (l >= 32) ? 126 : 127; }
  run();
  bad = 0;
  for (int l = 0; l < 64; ++l) for (int g = 0; g < 16; ++g) {
    const int col = l & 31, row = (g & 3) + 8 * (g >> 2) + 4 * (l >> 5);
    float ref = 0;
    for (int h = 0; h < 2; ++h) for (int j = 0; j < 32; ++j) ref += (h == 1 ? 0.5f : 1.f) * e4m3(A[row + 32 * h][j]) * e4m3(B[col + 32 * h][j]);
    if (ref != C[l][g]) { if (bad < 5) printf("scale-b mismatch lane %d reg %d: got %g want %g\n", l, g, C[l][g], ref); ++bad; }
  }
  printf("SCALE_B per lane = per (col, 32-k block), 126 = 0.5: %s (%d mismatches)\n", bad ? "NO" : "YES", bad);
  // conversions
  float x[4] = {1.0f, 2.5f, 0.07f, 300.0f}, *dx; _Float16 xh[4] = {(_Float16)1.0f, (_Float16)2.5f, (_Float16)0.07f, (_Float16)300.0f}, *dxh; int o[3], *dout;
  hipMalloc(&dx, 16); hipMalloc(&dxh, 8); hipMalloc(&dout, 12);
  hipMemcpy(dx, x, 16, hipMemcpyHostToDevice); hipMemcpy(dxh, xh, 8, hipMemcpyHostToDevice);
  for (float sc : {1.0f, 2.0f, 0.5f}) {
    hipLaunchKernelGGL(cv, dim3(1), dim3(1), 0, 0, dx, dout, dxh, sc);
    hipMemcpy(o, dout, 12, hipMemcpyDeviceToHost);
    printf("cvt_pk_fp8_f32(1, 2.5, 0.07, 300) = %08x -> %g %g %g %g\n", o[0], e4m3(o[0] & 255), e4m3((o[0] >> 8) & 255), e4m3((o[0] >> 16) & 255), e4m3((o[0] >> 24) & 255));
    printf("cvt_scalef32_pk_fp8_f16(.., scale %g): lo-word call %08x -> %g %g ; hi-word call %08x -> %g %g %g %g\n", sc, o[1], e4m3(o[1] & 255), e4m3((o[1] >> 8) & 255), o[2],
           e4m3(o[2] & 255), e4m3((o[2] >> 8) & 255), e4m3((o[2] >> 16) & 255), e4m3((o[2] >> 24) & 255));
  }
  return 0;
}
