// Fused YOLO11 stem for gfx950: model.0 (Conv 3x3 s2, 3 -> 16) + model.1 (Conv 3x3 s2, 16 -> 32) in one launch.
//
// Reference path: the first two layers of ultralytics' YOLO11 DetectionModel inside YOLO.predict
// (/root/reference/src/circuit_analyzer.py:268).  As two launches the 16-channel 320 x 320 map (105 MB at B = 32, fp16) is
// written and read back -- more than the 105 MB input and 52 MB output the pair actually needs.  Here one workgroup
// owns an 8 x 16 tile of model.1's output: it stages the 18 x 34 patch of the space-to-depth(2) image (on which model.0
// is a 2x2 / stride-1 conv over 16 channels, see Yolo11Weights.stem), runs model.0 on the 17 x 33 halo tile with MFMA
// (one 16-channel tap per k-step, weights as the A operand straight from L2), keeps SiLU(model.0) as fp16 in LDS -- zero
// outside the image, which is model.1's padding -- and runs model.1 (9 taps x 16 channels) from there.
#include "common.hpp"

namespace {

struct StemArgs {
  const char* x; const char* w0; const float* b0; const char* w1; const float* b1; char* y;
  int x_ld, y_ld, kpad0, kpad1;
  int H2, W2, OH, OW;                     // space-to-depth grid (= model.0 output grid), model.1 output grid
  int tiles_x, tiles_y;
};

constexpr int TH = 8, TW = 16;
constexpr int SH = 2 * TH + 1, SW = 2 * TW + 1;   // model.0 halo tile (17 x 33)
constexpr int PH = SH + 1, PW = SW + 1;           // s2d patch (18 x 34)
constexpr int NS = SH * SW;                       // 561 stem pixels
// LDS layouts (no padding bytes; both conflict-free for the 16-byte MFMA operand reads):
//   patch  [chunk 0..1][pixel]                      a lane group reads one 8-channel chunk of consecutive pixels -> consecutive slots
//   T      [chunk 0..1][column parity][row][17]     model.1 walks T with stride 2 in x: even and odd columns live in separate planes,
//                                                   so its 16 consecutive output columns read 16 consecutive slots
constexpr int PPLANE = PH * PW * 16;              // bytes of one patch chunk plane (612 pixels)
constexpr int TROW = (SW / 2 + 1) * 16;           // 17 slots per (parity, row)
constexpr int TPLANE = SH * TROW;                 // one (chunk, parity) plane
__device__ __forceinline__ int t_off(int ck, int sy, int sx) { return ((ck * 2 + (sx & 1)) * SH + sy) * TROW + (sx >> 1) * 16; }

__device__ __forceinline__ void mma16(const u32x4& a, const u32x4& b, f32x16& c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

__global__ __launch_bounds__(256, 4) void stem2_kernel(const StemArgs p) {
  constexpr int PATCH_B = 2 * PPLANE;               // 19584
  constexpr int T_B = 4 * TPLANE;                   // 18496
  constexpr int OSTR = 32 * 2 + 16;                 // output tile row stride
  static_assert(TH * TW * OSTR <= PATCH_B, "the output tile overlays the patch");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const patch = smem;
  char* const T = smem + PATCH_B;
  float* const b0s = reinterpret_cast<float*>(T + T_B);
  float* const b1s = b0s + 32;

  const int tid = threadIdx.x;
  int bid = blockIdx.x;
  const int tx = bid % p.tiles_x; bid /= p.tiles_x;
  const int ty = bid % p.tiles_y;
  const int b = bid / p.tiles_y;
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int Y0 = 2 * oy0 - 1, X0 = 2 * ox0 - 1;      // model.0 coordinates of halo-tile pixel (0, 0)
  const int wv = tid >> 6, lane = tid & 63;
  const int lr = lane & 31, lh = lane >> 5;

  // ---- stage the s2d patch (origin (Y0 - 1, X0 - 1), zero outside the image): all loads first, one round trip.  A patch
  //      row is 68 contiguous 16-byte chunks; a pass covers three rows (204 of 256 threads), six passes the 18 rows:
  //      per-pass addressing is "thread base + pass * 3 rows" (a generic chunk -> pixel map costs ~40 VALU per load)
  constexpr int RC = PW * 2, RPP = 3, NP = PH / RPP;
  static_assert(RC * RPP <= 256 && PH % RPP == 0, "patch staging map");
  u32x4 pv[NP];
  bool pok[NP];
  const int p_rsel = tid / RC, p_t = tid - p_rsel * RC;
  const int p_ix = X0 - 1 + (p_t >> 1);
  const bool p_xok = p_rsel < RPP && (unsigned)p_ix < (unsigned)p.W2;
  const size_t row_pitch = (size_t)p.W2 * p.x_ld * 2;
  const char* const p_src = p.x + (size_t)b * p.H2 * row_pitch + (p_xok ? (size_t)p_ix * p.x_ld * 2 + (p_t & 1) * 16 : 0);
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int iy = Y0 - 1 + i * RPP + p_rsel;
    const bool ok = p_xok && (unsigned)iy < (unsigned)p.H2;
    pv[i] = *reinterpret_cast<const u32x4*>(p_src + (size_t)(ok ? iy : 0) * row_pitch);
    pok[i] = ok;
  }
  // weight fragments (A operands) straight from L2: model.0 = 4 taps x 16 channels, model.1 = 9 taps x 16 channels
  u32x4 w0f[4], w1f[9];
#pragma unroll
  for (int t = 0; t < 4; ++t) w0f[t] = *reinterpret_cast<const u32x4*>(p.w0 + ((size_t)lr * p.kpad0 + t * 16 + lh * 8) * 2);
#pragma unroll
  for (int t = 0; t < 9; ++t) w1f[t] = *reinterpret_cast<const u32x4*>(p.w1 + ((size_t)lr * p.kpad1 + t * 16 + lh * 8) * 2);
  const float bias_v = tid < 32 ? p.b0[tid] : (tid < 64 ? p.b1[tid - 32] : 0.f);
  if (p_rsel < RPP) {
#pragma unroll
    for (int i = 0; i < NP; ++i)
      *reinterpret_cast<u32x4*>(patch + (p_t & 1) * PPLANE + ((i * RPP + p_rsel) * PW + (p_t >> 1)) * 16) = pok[i] ? pv[i] : u32x4{0u, 0u, 0u, 0u};
  }
  if (tid < 64) b0s[tid] = bias_v;                   // b0s[0..32) = model.0 bias (16 real), b1s = b0s + 32
  __syncthreads();

  // ---- model.0 on the halo tile: 18 M-tiles of 32 pixels, 4 taps each; rows 0..15 of the 32-row MFMA tile are real ---
  f32x4 b0v[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) b0v[q] = *reinterpret_cast<const f32x4*>(b0s + 8 * q + 4 * lh);
  int sy = (wv * 32 + lr) / SW, sx = (wv * 32 + lr) - sy * SW;          // halo-tile pixel of this lane, advanced by 128 pixels per round
  for (int mt = wv; mt < (NS + 31) / 32; mt += 4, sy += 3, sx += 128 - 3 * SW) {
    if (sx >= SW) { sx -= SW; ++sy; }
    const int m = mt * 32 + lr;
    if (m >= NS) { sy = SH - 1; sx = SW - 1; }          // lanes past the tile: any valid pixel for the reads; they do not write
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const u32x4 xf = *reinterpret_cast<const u32x4*>(patch + lh * PPLANE + ((sy + (t >> 1)) * PW + sx + (t & 1)) * 16);
      mma16(w0f[t], xf, acc);
    }
    const bool in_img = m < NS && (unsigned)(Y0 + sy) < (unsigned)p.H2 && (unsigned)(X0 + sx) < (unsigned)p.W2;
#pragma unroll
    for (int q = 0; q < 2; ++q) {                     // registers 4q..4q+3 = channels 8q + 4 lh + (0..3)
      const f32x4 bv = b0v[q];
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = in_img ? act_apply<true>(acc[4 * q + e] + bv[e], CVMI_ACT_SILU) : 0.f;   // outside: model.1's zero padding
      const f16x4 hv = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
      if (m < NS) *reinterpret_cast<f16x4*>(T + t_off(q, sy, sx) + lh * 8) = hv;      // chunk q = channels 8q..8q+7, this lane's half at 4 lh
    }
  }
  __syncthreads();

  // ---- model.1: wave wv owns output pixels [32 wv, 32 wv + 32) (two tile rows), 9 taps ---------------------------------
  {
    const int m = wv * 32 + lr;
    const int r = m >> 4, c = m & 15;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const u32x4 xf = *reinterpret_cast<const u32x4*>(T + t_off(lh, 2 * r + t / 3, 2 * c + t % 3));
      mma16(w1f[t], xf, acc);
    }
    char* const Ot = patch;                           // the patch is dead since the barrier above
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int nl = 8 * q + 4 * lh;
      const f32x4 bv = *reinterpret_cast<const f32x4*>(b1s + nl);
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = act_apply<true>(acc[4 * q + e] + bv[e], CVMI_ACT_SILU);
      const f16x4 hv = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
      *reinterpret_cast<f16x4*>(Ot + m * OSTR + nl * 2) = hv;
    }
  }
  __syncthreads();
  for (int idx = tid; idx < TH * TW * 4; idx += 256) {  // 4 x 16 bytes per output pixel
    const int row = idx >> 2, ch = idx & 3;
    const int oy = oy0 + (row >> 4), ox = ox0 + (row & 15);
    if (oy >= p.OH || ox >= p.OW) continue;
    *reinterpret_cast<u32x4*>(p.y + ((((size_t)b * p.OH + oy) * p.OW + ox) * p.y_ld + ch * 8) * 2) =
        *reinterpret_cast<const u32x4*>(patch + row * OSTR + ch * 16);
  }
}

}  // namespace

extern "C" int cvmi_stem2_supported(int c0, int c1, int dtype) { return dtype == CVMI_F16 && c0 == 16 && c1 == 32; }

extern "C" int cvmi_stem2(const void* x, int x_ld, const void* w0, const float* b0, int kpad0, const void* w1, const float* b1, int kpad1,
                          void* y, int y_ld, int B, int H2, int W2, int c0, int c1, int dtype, cvmi_stream_t stream_) {
  CVMI_CHECK(x && w0 && b0 && w1 && b1 && y, "stem2: null pointer");
  CVMI_CHECK(cvmi_stem2_supported(c0, c1, dtype), "stem2: configuration (c0=%d c1=%d) is not built", c0, c1);
  CVMI_CHECK(B > 0 && H2 > 0 && W2 > 0 && x_ld >= 16 && x_ld % 8 == 0 && y_ld >= c1 && y_ld % 8 == 0, "stem2: bad shape / ld");
  CVMI_CHECK(kpad0 >= 64 && kpad1 >= 144 && kpad0 % 8 == 0 && kpad1 % 8 == 0, "stem2: Kpad too small");
  CVMI_CHECK((((uintptr_t)x | (uintptr_t)y | (uintptr_t)w0 | (uintptr_t)w1) & 15) == 0, "stem2: tensors must be 16-byte aligned");
  StemArgs a;
  a.x = (const char*)x; a.w0 = (const char*)w0; a.b0 = b0; a.w1 = (const char*)w1; a.b1 = b1; a.y = (char*)y;
  a.x_ld = x_ld; a.y_ld = y_ld; a.kpad0 = kpad0; a.kpad1 = kpad1;
  a.H2 = H2; a.W2 = W2; a.OH = (H2 - 1) / 2 + 1; a.OW = (W2 - 1) / 2 + 1;
  a.tiles_x = cdiv(a.OW, TW); a.tiles_y = cdiv(a.OH, TH);
  const long long blocks = (long long)B * a.tiles_y * a.tiles_x;
  CVMI_CHECK(blocks > 0 && blocks < (1ll << 31), "stem2: bad grid");
  constexpr size_t lds = (size_t)2 * PPLANE + (size_t)4 * TPLANE + 64 * sizeof(float);
  hipLaunchKernelGGL(stem2_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream_, a);
  CVMI_LAUNCH_CHECK();
  return 0;
}
