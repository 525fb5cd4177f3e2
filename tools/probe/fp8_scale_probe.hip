// Which lane / byte of the scale operands of v_mfma_scale_f32_32x32x64_f8f6f4 scales which (row, k-block)?  A = B = 1.0 everywhere, so
// D[row][col] = 32 * (sA(row, 0) * sB(col, 0) + sA(row, 1) * sB(col, 1)); one lane's scale register is changed at a time.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
template <int OA, int OB>
__global__ void mm(const v8i* a, const v8i* b, v16f* c, const int* sa, const int* sb) {
  v16f acc = {0};
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, OA, sa[threadIdx.x], OB, sb[threadIdx.x]);
  c[threadIdx.x] = acc;
}
int main() {
  unsigned char A[64][32]; memset(A, 0x38, sizeof(A));
  int sa[64], sb[64]; float C[64][16];
  v8i *da, *db; v16f* dc; int *dsa, *dsb;
  (void)hipMalloc(&da, 2048); (void)hipMalloc(&db, 2048); (void)hipMalloc(&dc, 4096); (void)hipMalloc(&dsa, 256); (void)hipMalloc(&dsb, 256);
  (void)hipMemcpy(da, A, 2048, hipMemcpyHostToDevice); (void)hipMemcpy(db, A, 2048, hipMemcpyHostToDevice);
  auto run = [&](int opa) {
    (void)hipMemcpy(dsa, sa, 256, hipMemcpyHostToDevice); (void)hipMemcpy(dsb, sb, 256, hipMemcpyHostToDevice);
    if (opa == 0) hipLaunchKernelGGL((mm<0, 0>), dim3(1), dim3(64), 0, 0, da, db, dc, dsa, dsb);
    else if (opa == 1) hipLaunchKernelGGL((mm<1, 0>), dim3(1), dim3(64), 0, 0, da, db, dc, dsa, dsb);
    else if (opa == 2) hipLaunchKernelGGL((mm<2, 0>), dim3(1), dim3(64), 0, 0, da, db, dc, dsa, dsb);
    else hipLaunchKernelGGL((mm<3, 0>), dim3(1), dim3(64), 0, 0, da, db, dc, dsa, dsb);
    (void)hipMemcpy(C, dc, 4096, hipMemcpyDeviceToHost);
  };
  const int base = 0x7f7f7f7f;
  for (int which = 0; which < 2; ++which)            // 0: vary scale_a, 1: vary scale_b
    for (int byte = 0; byte < 2; ++byte)
      for (int L : {0, 1, 5, 31, 32, 33, 37, 63}) {
        for (int l = 0; l < 64; ++l) { sa[l] = base; sb[l] = base; }
        (which ? sb : sa)[L] = base + (1 << (8 * byte));      // that byte 127 -> 128
        run(0);
        // D row (from A) / col (from B) whose value changed, and the new values
        int n = 0; char msg[256] = ""; int len = 0;
        for (int l = 0; l < 64 && len < 200; ++l) for (int g = 0; g < 16 && len < 200; ++g) if (C[l][g] != 64.f) {
          const int col = l & 31, row = (g & 3) + 8 * (g >> 2) + 4 * (l >> 5);
          if (n < 3) len += snprintf(msg + len, sizeof(msg) - len, " D[%d][%d]=%g", row, col, C[l][g]);
          ++n;
        }
        printf("scale_%c lane %2d byte %d (opsel 0): %4d elements changed:%s\n", which ? 'b' : 'a', L, byte, n, msg);
      }
  // opsel: byte selection
  for (int op = 0; op < 4; ++op) {
    for (int l = 0; l < 64; ++l) { sa[l] = 0x7f7f7f7f; sb[l] = 0x7f7f7f7f; }
    for (int l = 0; l < 64; ++l) sa[l] = 0x82818080 ;       // bytes: 0x80, 0x80, 0x81, 0x82 -> 2, 2, 4, 8
    run(op);
    printf("opsel_a %d with scale_a bytes (lo..hi) 2,2,4,8 in every lane: D[0][0] = %g (64 x scale expected)\n", op, C[0][0]);
  }
  for (int l = 0; l < 64; ++l) { sa[l] = 0x7f7f7f7f; sb[l] = 0x7f7f7f7f; }
  for (int l = 0; l < 32; ++l) sa[l] = 0x7f7f7f80;
  run(0);
  printf("scale_a byte0 = 128 in lanes 0..31: D[0][0] = %g D[5][7] = %g D[31][31] = %g\n", C[0][0], C[7][5 % 4 + 0], C[31 + 32][15]);
  return 0;
}
