// Fused C3k2 block (c3k = False, n = 1) of the YOLO11 backbone / neck for the small-channel, HBM-bound stages:
//
//     y = SiLU(cv1 x) -> [a | b]          1x1,  C1 -> 2c      (optional: fused when the block has ONE source, C1 <= 64)
//     t = SiLU(m.cv1 (*) b)               3x3,  c  -> h
//     m = b + SiLU(m.cv2 (*) t)           3x3,  h  -> c       (shortcut)
//     out = SiLU(cv2 [a | b | m])         1x1,  3c -> C2
//
// Unfused, the block is 4 launches that move [a|b], t, m and the 3c-wide concat through HBM / L2 (YOLO11-n model.2:
// 393 MB per batch of 32 against 157 MB of block input + output).  Here a workgroup owns an 8 x 16 tile of output
// pixels: it stages the input patch with a 2-pixel halo once (12 x 20 pixels), keeps every intermediate in LDS
// (zeroed outside the image, exactly the convs' zero padding; rounded to fp16 exactly where the unfused path stores
// fp16) and runs the four small GEMMs on MFMA with the operand roles / fragment addressing of conv_tile.hip
// (weights = A operand, pixel fragments = ds_read_b128 at "lane base + compile-time tap offset").
// Replaces, inside YOLO.predict (ultralytics C3k2.forward), the launches model.N.cv1 / m.0.cv1 / m.0.cv2 / cv2.
#include "common.hpp"

namespace {

struct C3Args {
  const char* x; const char *w0, *w1, *w2, *w3; const float *b0, *b1, *b2, *b3; char* y;
  int x_ld, y_ld, kpad0, kpad1, kpad2, kpad3;
  int H, W, tiles_x, tiles_y, shortcut, ntiles;
};

__device__ __forceinline__ void mma16(const u32x4& a, const u32x4& b, f32x16& c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ float silu(float v) { return v * fast_rcp(1.0f + __expf(-v)); }

constexpr int odd16(int bytes) { return bytes + ((bytes / 16) % 2 == 0 ? 16 : 32); }   // row stride: odd multiple of 16 B

template <int C, int HR, int C2, int C1>
struct C3Cfg {
  static_assert(C == 16 || C == 32, "hidden bottleneck width c must be 16 or 32");
  static constexpr int TH = 8, TW = 16;
  static constexpr int YH = TH + 4, YW = TW + 4, NY = YH * YW;          // [a|b] (and x) patch: halo 2
  static constexpr int TRH = TH + 2, TRW = TW + 2, NT = TRH * TRW;      // t region: halo 1
  static constexpr int NM = TH * TW;
  static constexpr int HP = HR < 16 ? 16 : HR;                          // hidden channels padded to one MFMA k-step
  static constexpr int XSTR = C1 ? odd16(C1 * 2) : 0, YSTR = odd16(2 * C * 2), TSTR = odd16(HP * 2), MSTR = odd16(C * 2);
  static constexpr int OSTR = C2 * 2 + 16;
  // LDS map.  Fused cv1: [X | Y | bias], with t and m overlaying the x patch (dead after S0); otherwise
  // [Y | t | m | bias].  The output staging tile overlays the front once every patch is dead.
  static constexpr int X_OFF = 0, Y_OFF = NY * XSTR;
  static constexpr int T_OFF = C1 ? 0 : Y_OFF + NY * YSTR, M_OFF = T_OFF + 192 * TSTR;
  static constexpr int B_OFF = C1 ? Y_OFF + NY * YSTR : M_OFF + NM * MSTR;          // biases: b0 [2C] b1 [32] b2 [32] b3 [C2] (f32)
  static constexpr int B0 = 0, B1 = 2 * C, B2 = B1 + 32, B3 = B2 + 32, NBIAS = B3 + C2;
  static constexpr int END = B_OFF + NBIAS * 4;
  static constexpr int OUT_B = NM * OSTR;
  static constexpr int LDS = END;
  static_assert(OUT_B <= B_OFF, "output staging must not reach the bias table");
  static_assert(!C1 || M_OFF + NM * MSTR <= Y_OFF, "t and m must fit inside the x patch");
};

// Persistent: a workgroup loops over tiles; every weight fragment it needs lives in registers for the whole launch
// (loaded once, straight from the packed global matrices), and all per-lane LDS / global offsets are tile-invariant.
// Per tile the only global traffic is the input patch (prefetched into registers one tile ahead) and the output.
// c = 16: 4 waves per workgroup, 2-3 workgroups per CU; c = 32: 8 waves (the per-wave share of pixel tiles, and with it
// the register footprint beside the 148 weight registers, halves), one workgroup per CU.
template <int C, int HR, int C2, int C1>
__global__ __launch_bounds__(C == 16 ? 256 : 512, C == 16 ? 2 : 1) void c3k2_kernel(const C3Args p) {
  using G = C3Cfg<C, HR, C2, C1>;
  constexpr int NW = C == 16 ? 4 : 8, NTH = NW * 64;
  constexpr int TW = G::TW, YW = G::YW, NY = G::NY, TRW = G::TRW, NT = G::NT, NM = G::NM, HP = G::HP;
  constexpr int YSTR = G::YSTR, TSTR = G::TSTR, MSTR = G::MSTR, OSTR = G::OSTR;
  constexpr bool FUSE = C1 > 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const X = smem + G::X_OFF; char* const Y = smem + G::Y_OFF;
  float* const Bs = reinterpret_cast<float*>(smem + G::B_OFF);
  char* const Ot = smem;

  const int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lr = lane & 31, lh = lane >> 5;

  // ---- one-time: biases -> LDS, weight fragments -> registers ---------------------------------------------------------
  for (int i = tid; i < G::NBIAS; i += NTH) {
    float v;
    if (i < G::B1) v = FUSE ? p.b0[i] : 0.f;
    else if (i < G::B2) v = p.b1[i - G::B1];
    else if (i < G::B3) v = p.b2[i - G::B2];
    else v = p.b3[i - G::B3];
    Bs[i] = v;
  }
  // S0: 2C/32 channel tiles x 8 pixel tiles; a wave owns ONE channel tile and PT0 pixel tiles
  constexpr int NTN0 = FUSE ? 2 * C / 32 : 1, PT0 = FUSE ? 8 * NTN0 / NW : 2, NK0 = FUSE ? C1 / 16 : 1;
  static_assert(PT0 == 2, "S0 processes pixel tiles in pairs");
  const int ct0 = wv % NTN0, pt0 = (wv / NTN0) * PT0;
  u32x4 wr0[NK0];
  if constexpr (FUSE) {
#pragma unroll
    for (int ks = 0; ks < NK0; ++ks) wr0[ks] = *reinterpret_cast<const u32x4*>(p.w0 + ((size_t)(ct0 * 32 + lr) * p.kpad0 + ks * 16 + lh * 8) * 2);
  }
  // S1: one channel tile (h <= 32), 6 pixel tiles: wave w takes tiles w, w + NW, ... below 6
  constexpr int NK1 = 9 * C / 16, NPT1 = (6 + NW - 1) / NW;
  u32x4 wr1[NK1];
#pragma unroll
  for (int ks = 0; ks < NK1; ++ks) wr1[ks] = *reinterpret_cast<const u32x4*>(p.w1 + ((size_t)lr * p.kpad1 + ks * 16 + lh * 8) * 2);
  // S2: one channel tile (c <= 32), 4 pixel tiles: one per wave.  k-step = (tap, 16 hidden channels); hidden width 8:
  // the upper lane half multiplies the zero padding of t
  constexpr int NK2 = 9 * HP / 16;
  u32x4 wr2[NK2];
#pragma unroll
  for (int ks = 0; ks < NK2; ++ks) {
    if constexpr (HP == HR) {
      wr2[ks] = *reinterpret_cast<const u32x4*>(p.w2 + ((size_t)lr * p.kpad2 + ks * 16 + lh * 8) * 2);
    } else {
      wr2[ks] = *reinterpret_cast<const u32x4*>(p.w2 + ((size_t)lr * p.kpad2 + ks * 8) * 2);
      if (lh) wr2[ks] = u32x4{0u, 0u, 0u, 0u};
    }
  }
  // S3: C2/32 channel tiles x 4 pixel tiles; a wave owns ONE channel tile and PT3 pixel tiles
  constexpr int NTN3 = C2 / 32, PT3 = 4 * NTN3 / NW, NK3 = 3 * C / 16;
  static_assert(PT3 == 1 || PT3 == 2, "S3: one or two pixel tiles per wave");
  static_assert(NTN3 == 2 || NTN3 == 4, "C2 must be 64 or 128");
  const int ct3 = wv % NTN3, pt3 = (wv / NTN3) * PT3;
  u32x4 wr3[NK3];
#pragma unroll
  for (int ks = 0; ks < NK3; ++ks) wr3[ks] = *reinterpret_cast<const u32x4*>(p.w3 + ((size_t)(ct3 * 32 + lr) * p.kpad3 + ks * 16 + lh * 8) * 2);

  // ---- tile-invariant per-lane offsets ---------------------------------------------------------------------------------
  // patch staging: chunk id -> (patch pixel, 16-byte chunk)
  constexpr int PCH = FUSE ? C1 * 2 / 16 : 2 * C * 2 / 16;
  constexpr int PSTR = FUSE ? G::XSTR : YSTR;
  constexpr int NPL = (NY * PCH + NTH - 1) / NTH;
  char* const P = FUSE ? X : Y;
  int sg_rc[NPL];                                           // pr | pc << 8 | chunk << 16: global / LDS offsets are rebuilt per tile (registers)
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int id = tid + i * NTH;
    const int pp = id / PCH, ch = id - pp * PCH;
    const int pr = pp / YW, pc = pp - pr * YW;
    sg_rc[i] = pr | (pc << 8) | (ch << 16);
  }
  // S0 pixels
  int s0_rd[PT0], s0_wr[PT0], s0_rc[PT0];
#pragma unroll
  for (int j = 0; j < PT0; ++j) {
    const int pp = (pt0 + j) * 32 + lr, pcl = pp < NY ? pp : NY - 1;
    s0_rd[j] = G::X_OFF + pcl * G::XSTR + lh * 16;
    s0_wr[j] = pp < NY ? G::Y_OFF + pp * YSTR + ct0 * 64 + lh * 8 : -1;
    s0_rc[j] = (pcl / YW) | ((pcl % YW) << 8);
  }
  // S1 pixels
  int s1_rd[NPT1], s1_wr[NPT1], s1_rc[NPT1];
#pragma unroll
  for (int j = 0; j < NPT1; ++j) {
    const int pp = (wv + NW * j) * 32 + lr, pcl = pp < NT ? pp : NT - 1;
    const int pr = pcl / TRW, pc = pcl - pr * TRW;
    s1_rd[j] = G::Y_OFF + (pr * YW + pc) * YSTR + C * 2 + lh * 16;
    s1_wr[j] = pp < NT ? G::T_OFF + pp * TSTR + lh * 8 : -1;
    s1_rc[j] = pr | (pc << 8);
  }
  // S2 pixel
  const int mp2 = (wv & 3) * 32 + lr, mr2 = mp2 / TW, mc2 = mp2 - mr2 * TW;   // waves 4..7 (c = 32) sit S2 out
  const int s2_rd = G::T_OFF + (mr2 * TRW + mc2) * TSTR + lh * 16;
  const int s2_res = G::Y_OFF + ((mr2 + 2) * YW + mc2 + 2) * YSTR + C * 2 + lh * 8;
  const int s2_wr = G::M_OFF + mp2 * MSTR + lh * 8;
  // S3 pixels
  int s3_ab[PT3], s3_m[PT3], s3_wr[PT3];
#pragma unroll
  for (int j = 0; j < PT3; ++j) {
    const int pp = (pt3 + j) * 32 + lr, r = pp / TW, c = pp - r * TW;
    s3_ab[j] = G::Y_OFF + ((r + 2) * YW + c + 2) * YSTR + lh * 16;
    s3_m[j] = G::M_OFF + pp * MSTR + lh * 16;
    s3_wr[j] = pp * OSTR + ct3 * 64 + lh * 8;
  }

  const int ntiles = p.ntiles, tpi = p.tiles_x * p.tiles_y;
  // window tests as unsigned compares: patch pixel (pr, pc) is inside the image iff pr - rlo < rn and pc - clo < cn
  auto tile_origin = [&](int tile, int& b, int& oy0, int& ox0) {
    b = tile / tpi;
    const int t = tile - b * tpi;
    const int ty = t / p.tiles_x;
    oy0 = ty * G::TH; ox0 = (t - ty * p.tiles_x) * TW;
  };
  constexpr bool PREFETCH = !(C == 32 && C1 > 0);       // that one configuration has no registers left for a patch in flight
  u32x4 pv[NPL];
  unsigned pmask = 0;                                       // bit i: chunk i of the prefetched patch is inside the image
  auto patch_load = [&](int tile) {
    int b, oy0, ox0;
    tile_origin(tile, b, oy0, ox0);
    const char* base = p.x + (((long long)b * p.H + (oy0 - 2)) * p.W + (ox0 - 2)) * p.x_ld * 2;
    const unsigned rlo = (unsigned)(2 - oy0), clo = (unsigned)(2 - ox0);            // valid rows: pr in [2 - oy0, H + 2 - oy0)
    pmask = 0;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const unsigned pr = sg_rc[i] & 255, pc = (sg_rc[i] >> 8) & 255, ch = sg_rc[i] >> 16;
      const bool ok = tid + i * NTH < NY * PCH && (unsigned)(pr - rlo) < (unsigned)p.H && (unsigned)(pc - clo) < (unsigned)p.W;
      const char* src = ok ? base + (int)(((pr * p.W + pc) * p.x_ld + ch * 8) * 2) : p.x;   // clamped address; the zero is applied at
      pv[i] = *reinterpret_cast<const u32x4*>(src);                                        // LDS-write time: nothing waits for the load
      pmask |= ok ? 1u << i : 0u;
    }
  };
  auto patch_store = [&]() {
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const unsigned pr = sg_rc[i] & 255, pc = (sg_rc[i] >> 8) & 255, ch = sg_rc[i] >> 16;
      if (tid + i * NTH < NY * PCH) *reinterpret_cast<u32x4*>(P + (pr * YW + pc) * PSTR + ch * 16) = (pmask >> i) & 1u ? pv[i] : u32x4{0u, 0u, 0u, 0u};
    }
  };

  int tile = blockIdx.x;
  if (tile < ntiles) { patch_load(tile); patch_store(); }
  __syncthreads();

  for (; tile < ntiles; tile += gridDim.x) {
    int b, oy0, ox0;
    tile_origin(tile, b, oy0, ox0);
    const int next = tile + gridDim.x;
    if (PREFETCH && next < ntiles) patch_load(next);       // in flight during the whole tile

    // ---- S0: [a|b] = SiLU(W0 x + b0) on the 12 x 20 patch, zero outside the image ----------------------------------------
    if constexpr (FUSE) {
      const unsigned rlo = (unsigned)(2 - oy0), clo = (unsigned)(2 - ox0);
#pragma unroll
      for (int j0 = 0; j0 < PT0; j0 += 2) {
        f32x16 acc[2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NK0; ++ks)
#pragma unroll
          for (int u = 0; u < 2; ++u) mma16(wr0[ks], *reinterpret_cast<const u32x4*>(smem + s0_rd[j0 + u] + ks * 32), acc[u]);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int j = j0 + u;
          const unsigned pr = s0_rc[j] & 255, pc = s0_rc[j] >> 8;
          const bool in = (unsigned)(pr - rlo) < (unsigned)p.H && (unsigned)(pc - clo) < (unsigned)p.W;
          if (s0_wr[j] >= 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f32x4 bv = *reinterpret_cast<const f32x4*>(Bs + G::B0 + ct0 * 32 + 8 * q + 4 * lh);
              f16x4 hv;
#pragma unroll
              for (int e = 0; e < 4; ++e) hv[e] = (f16)silu(acc[u][4 * q + e] + bv[e]);
              u32x2 hb = __builtin_bit_cast(u32x2, hv);
              hb[0] = in ? hb[0] : 0u; hb[1] = in ? hb[1] : 0u;             // selects, not branches
              *reinterpret_cast<u32x2*>(smem + s0_wr[j] + 16 * q) = hb;
            }
          }
        }
      }
      __syncthreads();
    }

    // ---- S1: t = SiLU(W1 (*) b + b1) on the 10 x 18 region, zero outside the image ----------------------------------------
    {
      const unsigned rlo = (unsigned)(1 - oy0), clo = (unsigned)(1 - ox0);
      const int nt1 = (6 - wv + NW - 1) / NW;                 // wave-uniform number of tiles (0, 1 or 2)
      f32x16 acc[NPT1];
#pragma unroll
      for (int u = 0; u < NPT1; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
      if (NPT1 == 2 && nt1 == 2) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
          for (int ks = 0; ks < C / 16; ++ks)
#pragma unroll
            for (int u = 0; u < NPT1; ++u)
              mma16(wr1[tap * (C / 16) + ks], *reinterpret_cast<const u32x4*>(smem + s1_rd[u] + ((tap / 3) * YW + tap % 3) * YSTR + ks * 32), acc[u]);
      } else if (nt1 >= 1) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
          for (int ks = 0; ks < C / 16; ++ks)
            mma16(wr1[tap * (C / 16) + ks], *reinterpret_cast<const u32x4*>(smem + s1_rd[0] + ((tap / 3) * YW + tap % 3) * YSTR + ks * 32), acc[0]);
      }
#pragma unroll
      for (int u = 0; u < NPT1; ++u) {
        if (u < nt1 && s1_wr[u] >= 0) {
          const unsigned pr = s1_rc[u] & 255, pc = s1_rc[u] >> 8;
          const bool in = (unsigned)(pr - rlo) < (unsigned)p.H && (unsigned)(pc - clo) < (unsigned)p.W;
#pragma unroll
          for (int q = 0; q < HP / 8; ++q) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(Bs + G::B1 + 8 * q + 4 * lh);
            f16x4 hv;
#pragma unroll
            for (int e = 0; e < 4; ++e) hv[e] = (f16)silu(acc[u][4 * q + e] + bv[e]);
            u32x2 hb = __builtin_bit_cast(u32x2, hv);
            const bool keep = in && 8 * q + 4 * lh < HR;                      // HR is a multiple of 4: a lane's 4 channels are all real or all padding
            hb[0] = keep ? hb[0] : 0u; hb[1] = keep ? hb[1] : 0u;
            *reinterpret_cast<u32x2*>(smem + s1_wr[u] + 16 * q) = hb;
          }
        }
      }
    }
    __syncthreads();

    // ---- S2: m = b + SiLU(W2 (*) t + b2) on the 8 x 16 tile ---------------------------------------------------------------
    if (NW == 4 || wv < 4) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int ks = 0; ks < HP / 16; ++ks)
          mma16(wr2[tap * (HP / 16) + ks], *reinterpret_cast<const u32x4*>(smem + s2_rd + ((tap / 3) * TRW + tap % 3) * TSTR + ks * 32), acc);
#pragma unroll
      for (int q = 0; q < C / 8; ++q) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(Bs + G::B2 + 8 * q + 4 * lh);
        const f16x4 rb = *reinterpret_cast<const f16x4*>(smem + s2_res + 16 * q);
        f16x4 hv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const f16 s = (f16)silu(acc[4 * q + e] + bv[e]);              // the unfused conv stores fp16 before the residual add
          hv[e] = p.shortcut ? (f16)((float)s + (float)rb[e]) : s;
        }
        *reinterpret_cast<f16x4*>(smem + s2_wr + 16 * q) = hv;
      }
    }
    __syncthreads();

    // ---- S3: out = SiLU(W3 [a|b|m] + b3) ----------------------------------------------------------------------------------
    f16x4 ov[PT3][4];
    {
      f32x16 acc[PT3];
#pragma unroll
      for (int u = 0; u < PT3; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < NK3; ++ks)
#pragma unroll
        for (int u = 0; u < PT3; ++u) {
          const int off = ks < 2 * C / 16 ? s3_ab[u] + ks * 32 : s3_m[u] + (ks - 2 * C / 16) * 32;
          mma16(wr3[ks], *reinterpret_cast<const u32x4*>(smem + off), acc[u]);
        }
#pragma unroll
      for (int u = 0; u < PT3; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(Bs + G::B3 + ct3 * 32 + 8 * q + 4 * lh);
#pragma unroll
          for (int e = 0; e < 4; ++e) ov[u][q][e] = (f16)silu(acc[u][4 * q + e] + bv[e]);
        }
    }
    __syncthreads();                                         // every read of the patches is done: overlay the output tile
#pragma unroll
    for (int j = 0; j < PT3; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) *reinterpret_cast<f16x4*>(Ot + s3_wr[j] + 16 * q) = ov[j][q];
    __syncthreads();
    {
      constexpr int NCH = C2 / 8;
      char* const ybase = p.y + (((long long)b * p.H + oy0) * p.W + ox0) * p.y_ld * 2;
#pragma unroll
      for (int it = 0; it < NM * NCH / NTH; ++it) {
        const int idx = tid + it * NTH;
        const int row = idx / NCH, ch = idx - row * NCH;
        const int r = row / TW, c = row - r * TW;
        if (oy0 + r < p.H && ox0 + c < p.W)
          *reinterpret_cast<u32x4*>(ybase + ((long long)(r * p.W + c) * p.y_ld + ch * 8) * 2) = *reinterpret_cast<const u32x4*>(Ot + row * OSTR + ch * 16);
      }
    }
    if (!PREFETCH && next < ntiles) patch_load(next);
    __syncthreads();                                         // output tile consumed: the next patch may land
    if (next < ntiles) patch_store();
    __syncthreads();
  }
}

template <int C, int HR, int C2, int C1>
int launch_c3k2(C3Args& a, int B, hipStream_t stream) {
  using G = C3Cfg<C, HR, C2, C1>;
  static bool attr_done = false;
  static int wgs_per_cu = 1, ncu = 256;
  if (!attr_done) {
    if (G::LDS > 64 * 1024)
      CVMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&c3k2_kernel<C, HR, C2, C1>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
    int dev = 0;
    CVMI_HIP(hipGetDevice(&dev));
    CVMI_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    CVMI_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&wgs_per_cu, c3k2_kernel<C, HR, C2, C1>, C == 16 ? 256 : 512, G::LDS));
    if (wgs_per_cu < 1) wgs_per_cu = 1;
    attr_done = true;
  }
  a.ntiles = B * a.tiles_y * a.tiles_x;
  CVMI_CHECK(a.ntiles > 0, "c3k2: bad grid");
  const int grid = a.ntiles < ncu * wgs_per_cu ? a.ntiles : ncu * wgs_per_cu;
  hipLaunchKernelGGL((c3k2_kernel<C, HR, C2, C1>), dim3((unsigned)grid), dim3(C == 16 ? 256 : 512), G::LDS, stream, a);
  CVMI_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int cvmi_c3k2_supported(int c1, int c, int h, int c2, int fuse_cv1, int dtype) {
  if (dtype != CVMI_F16) return 0;
  if (fuse_cv1) return (c1 == 32 && c == 16 && h == 8 && c2 == 64) || (c1 == 64 && c == 32 && h == 16 && c2 == 128);
  return (c == 16 && h == 8 && c2 == 64) || (c == 32 && h == 16 && (c2 == 64 || c2 == 128));
}

extern "C" int cvmi_c3k2(const cvmi_c3k2_desc* d, cvmi_stream_t stream_) {
  CVMI_CHECK(d != nullptr && d->x && d->y && d->w1 && d->w2 && d->w3 && d->b1 && d->b2 && d->b3, "c3k2: null pointer");
  CVMI_CHECK(!d->fuse_cv1 || (d->w0 && d->b0), "c3k2: fuse_cv1 needs the cv1 weights");
  CVMI_CHECK(cvmi_c3k2_supported(d->c1, d->c, d->h, d->c2, d->fuse_cv1, d->dtype), "c3k2: configuration (c1=%d c=%d h=%d c2=%d fuse=%d) is not built",
             d->c1, d->c, d->h, d->c2, d->fuse_cv1);
  const int cin = d->fuse_cv1 ? d->c1 : 2 * d->c;
  CVMI_CHECK(d->B > 0 && d->H > 0 && d->W > 0 && d->x_ld >= cin && d->x_ld % 8 == 0 && d->y_ld >= d->c2 && d->y_ld % 8 == 0, "c3k2: bad shape / ld");
  CVMI_CHECK((((uintptr_t)d->x | (uintptr_t)d->y) & 15) == 0, "c3k2: tensors must be 16-byte aligned");
  CVMI_CHECK((!d->fuse_cv1 || d->kpad0 >= d->c1) && d->kpad1 >= 9 * d->c && d->kpad2 >= 9 * d->h && d->kpad3 >= 3 * d->c, "c3k2: Kpad too small");
  C3Args a;
  a.x = (const char*)d->x; a.y = (char*)d->y; a.x_ld = d->x_ld; a.y_ld = d->y_ld;
  a.w0 = (const char*)d->w0; a.w1 = (const char*)d->w1; a.w2 = (const char*)d->w2; a.w3 = (const char*)d->w3;
  a.b0 = d->b0; a.b1 = d->b1; a.b2 = d->b2; a.b3 = d->b3;
  a.kpad0 = d->kpad0; a.kpad1 = d->kpad1; a.kpad2 = d->kpad2; a.kpad3 = d->kpad3;
  a.H = d->H; a.W = d->W; a.tiles_x = cdiv(d->W, 16); a.tiles_y = cdiv(d->H, 8); a.shortcut = d->shortcut;
  hipStream_t s = (hipStream_t)stream_;
  if (d->fuse_cv1) {
    if (d->c == 16) return launch_c3k2<16, 8, 64, 32>(a, d->B, s);
    return launch_c3k2<32, 16, 128, 64>(a, d->B, s);
  }
  if (d->c == 16) return launch_c3k2<16, 8, 64, 0>(a, d->B, s);
  if (d->c2 == 64) return launch_c3k2<32, 16, 64, 0>(a, d->B, s);
  return launch_c3k2<32, 16, 128, 0>(a, d->B, s);
}
