// Fused depthwise 3x3 + pointwise 1x1 (+ chained 1x1) for gfx950: the YOLO11 Detect class branch.
//
// Reference path: ultralytics Detect.cv3[i] = Sequential(Sequential(DWConv(x, x, 3), Conv(x, c3, 1)),
// Sequential(DWConv(c3, c3, 3), Conv(c3, c3, 1)), Conv2d(c3, nc, 1)) inside YOLO.predict
// (/root/reference/src/circuit_analyzer.py:268); as five launches the branch writes and re-reads four intermediate maps
// (at the 80x80 level of YOLO11-n, B = 32: 26-33 MB each) and pays five launch floors per level.
//
//   launch 1:  u = SiLU(pw1 * SiLU(dw1 (*) x))                      (C -> C -> c3)
//   launch 2:  y = cls * SiLU(pw2 * SiLU(dw2 (*) u))                (c3 -> c3 -> c3 -> nc, the last 1x1 without activation)
//
// One workgroup (4 waves) owns an 8 x 16 pixel tile.  Per 64/80-channel chunk: the 10 x 18 halo patch, the chunk's
// pointwise weight columns and the nine depthwise taps are staged global -> registers -> LDS in ONE round trip (the next
// chunk's loads are issued before this chunk's arithmetic); the depthwise conv runs on the VALU with fp32 accumulation
// (a thread owns a strip of 4 pixels x 8 channels: 18 LDS reads for 4 outputs) and writes its fp16 result straight into
// the MFMA operand tile [128 pixels][chunk]; the pointwise conv is mfma_f32_32x32x16_f16 with the weights as the A operand
// (each lane ends up with 4 consecutive output channels).  The chained 1x1 reads the activated pointwise tile back
// from LDS as its B operand and takes its weight fragments straight from L2.
#include "common.hpp"

namespace {

struct DwPwArgs {
  const char* x; const char* wd; const float* bd;
  const char* w1; const float* b1; const char* w2; const float* b2;
  char* y;
  int x_ld, y_ld, kpad1, kpad2;
  int H, W, N1, N2;
  int tiles_x, tiles_y;
};

constexpr int TH = 8, TW = 16, BMT = TH * TW;
constexpr int PH = TH + 2, PW = TW + 2;

__device__ __forceinline__ void mma16(const u32x4& a, const u32x4& b, f32x16& c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// CC channels per chunk (64 or 80), NCHUNK chunks (C = CC * NCHUNK), N1P / N2P = padded widths of the pointwise / chained conv
template <int CC, int NCHUNK, int N1P, int N2P>
__global__ __launch_bounds__(256, (NCHUNK > 1 ? 2 : 3)) void dwpw_kernel(const DwPwArgs p) {
  constexpr int NCK = CC / 8;                       // 16-byte channel chunks per pixel
  constexpr int PSTR = CC * 2 + 16;                 // LDS row stride of patch / operand tile / weight rows: odd multiple of 16 bytes
  constexpr int C = CC * NCHUNK;
  constexpr int PATCH_B = PH * PW * PSTR, AS_B = BMT * PSTR;
  constexpr int HSTR = N1P * 2 + 16, OSTR = (N2P > 0 ? N2P : 8) * 2 + 16;
  constexpr int WD_B = 9 * CC * 2;
  static_assert(BMT * HSTR <= (N2P > 0 ? PATCH_B : PATCH_B + AS_B), "the activated pointwise tile overlays the patch (and, unchained, the operand tile)");
  static_assert(N2P == 0 || BMT * OSTR <= AS_B, "the output tile overlays the operand tile");
  static_assert(N1P % 32 == 0 && N2P % 32 == 0 && CC % 16 == 0, "MFMA tiling");
  constexpr int TN1 = N1P / 32, TN2 = N2P / 32;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const patch = smem;
  char* const As = smem + PATCH_B;
  char* const wds = As + AS_B;
  float* const bds = reinterpret_cast<float*>(wds + WD_B);
  float* const b1s = bds + CC;
  float* const b2s = b1s + N1P;

  const int tid = threadIdx.x;
  int bid = blockIdx.x;
  const int tx = bid % p.tiles_x; bid /= p.tiles_x;
  const int ty = bid % p.tiles_y;
  const int b = bid / p.tiles_y;
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int wv = tid >> 6, lane = tid & 63;
  const int lr = lane & 31, lh = lane >> 5;

  // ---- staging (one round trip per chunk) ------------------------------------------------------------------------
  // (the pointwise weights are NOT staged: every wave needs all of them exactly once per chunk, so each lane fetches its
  //  MFMA A fragments straight from L2 with the patch -- 14 KB less LDS: three workgroups per CU instead of two)
  // patch staging map: a pass covers RPP rows of 16 columns (thread -> row-in-pass, column, channel chunk), a last pass the
  // two halo columns 16 / 17 of all rows: every pass is "base pointer + pass * row pitch" (the generic "chunk id -> pixel"
  // map cost ~45 VALU instructions per load in divisions and 64-bit multiplies)
  constexpr int TPR = 16 * NCK, RPP = 256 / TPR, NMAIN = PH / RPP, NP = NMAIN + 1;
  static_assert(PH % RPP == 0 && PH * 2 * NCK <= 256, "patch staging map");
  u32x4 pv[NP], wf1[TN1][CC / 16], wdv;
  bool pok[NP];
  float bdv;
  const int m_rsel = tid / TPR, m_t = tid - m_rsel * TPR;
  const int m_pc = m_t / NCK, m_ck = m_t - m_pc * NCK;
  const bool m_act = m_rsel < RPP;
  const int t_pr = tid / (2 * NCK), t_rem = tid - t_pr * (2 * NCK);
  const int t_pc = 16 + t_rem / NCK, t_ck = t_rem % NCK;
  const bool t_act = tid < PH * 2 * NCK;
  const size_t row_pitch = (size_t)p.W * p.x_ld * 2;
  const char* const img = p.x + (size_t)b * p.H * row_pitch;
  const bool m_xok = m_act && (unsigned)(ox0 - 1 + m_pc) < (unsigned)p.W;
  const bool t_xok = t_act && (unsigned)(ox0 - 1 + t_pc) < (unsigned)p.W;
  const char* const m_src = img + (size_t)(m_xok ? ox0 - 1 + m_pc : 0) * p.x_ld * 2 + (m_xok ? m_ck * 16 : 0);
  const char* const t_src = img + (size_t)(t_xok ? ox0 - 1 + t_pc : 0) * p.x_ld * 2 + (t_xok ? t_ck * 16 : 0);
  const int m_dst = (m_rsel * PW + m_pc) * PSTR + m_ck * 16, t_dst = (t_pr * PW + t_pc) * PSTR + t_ck * 16;
  auto load_chunk = [&](int c0) {
#pragma unroll
    for (int i = 0; i < NMAIN; ++i) {
      const int iy = oy0 - 1 + i * RPP + m_rsel;
      const bool ok = m_xok && (unsigned)iy < (unsigned)p.H;           // branch-free: clamped address + select at LDS-write time
      pv[i] = *reinterpret_cast<const u32x4*>(m_src + (size_t)(ok ? iy : 0) * row_pitch + c0 * 2);
      pok[i] = ok;
    }
    {
      const int iy = oy0 - 1 + t_pr;
      const bool ok = t_xok && (unsigned)iy < (unsigned)p.H;
      pv[NMAIN] = *reinterpret_cast<const u32x4*>(t_src + (size_t)(ok ? iy : 0) * row_pitch + c0 * 2);
      pok[NMAIN] = ok;
    }
    {
      const int tap = tid / NCK, ck = tid - tap * NCK;
      const bool ok = tid < 9 * NCK;
      wdv = *reinterpret_cast<const u32x4*>(p.wd + ((size_t)(ok ? tap : 0) * C + c0 + (ok ? ck * 8 : 0)) * 2);
    }
    bdv = p.bd[c0 + (tid < CC ? tid : 0)];
  };
  auto load_wfrag = [&](int c0) {
#pragma unroll
    for (int i = 0; i < TN1; ++i)
#pragma unroll
      for (int ks = 0; ks < CC / 16; ++ks)
        wf1[i][ks] = *reinterpret_cast<const u32x4*>(p.w1 + ((size_t)(i * 32 + lr) * p.kpad1 + c0 + ks * 16 + lh * 8) * 2);
  };
  auto store_chunk = [&]() {
    if (m_act) {
#pragma unroll
      for (int i = 0; i < NMAIN; ++i) *reinterpret_cast<u32x4*>(patch + m_dst + i * (RPP * PW * PSTR)) = pok[i] ? pv[i] : u32x4{0u, 0u, 0u, 0u};
    }
    if (t_act) *reinterpret_cast<u32x4*>(patch + t_dst) = pok[NMAIN] ? pv[NMAIN] : u32x4{0u, 0u, 0u, 0u};
    if (tid < 9 * NCK) *reinterpret_cast<u32x4*>(wds + tid * 16) = wdv;        // [tap][CC]: chunk tid = tap * NCK + ck
    if (tid < CC) bds[tid] = bdv;
  };

  const float b1v = p.b1[tid < N1P ? tid : 0];
  float b2v = 0.f;
  if constexpr (N2P > 0) b2v = p.b2[tid < N2P ? tid : 0];
  load_chunk(0);
  load_wfrag(0);

  f32x16 acc[TN1];
#pragma unroll
  for (int i = 0; i < TN1; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  for (int cc = 0; cc < NCHUNK; ++cc) {
    if (cc) __syncthreads();                          // the previous chunk's operand tile is consumed
    store_chunk();
    if (cc == 0) {
      if (tid < N1P) b1s[tid] = b1v;
      if (N2P > 0 && tid < N2P) b2s[tid] = b2v;
    }
    __syncthreads();
    if (cc + 1 < NCHUNK) load_chunk((cc + 1) * CC);     // next patch / taps: in flight during this chunk's arithmetic
    // ---- depthwise 3x3 + bias + SiLU on the VALU: strip of 4 pixels x 8 channels per item -----------------------
    for (int item = tid; item < 32 * NCK; item += 256) {
      const int strip = item / NCK, ck = item - strip * NCK;
      const int r = strip >> 2, x0 = (strip & 3) * 4;
      float a[4][8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float bv = bds[ck * 8 + e];
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q][e] = bv;
      }
#pragma unroll 1
      for (int ky = 0; ky < 3; ++ky) {                  // (not unrolled: 18 hoisted patch reads + the next chunk's staging registers spill)
        f16x8 wt[3];                                    // operands stay fp16: fma(ext(x), ext(w), acc) is one v_fma_mix_f32
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) wt[kx] = *reinterpret_cast<const f16x8*>(wds + ((ky * 3 + kx) * NCK + ck) * 16);
#pragma unroll
        for (int cx = 0; cx < 6; ++cx) {
          const f16x8 xf = *reinterpret_cast<const f16x8*>(patch + ((r + ky) * PW + x0 + cx) * PSTR + ck * 16);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int kx = cx - q;                      // same accumulation order as dwconv3x3_strip_kernel
            if (kx >= 0 && kx < 3) {
#pragma unroll
              for (int e = 0; e < 8; ++e) a[q][e] = fmaf((float)xf[e], (float)wt[kx][e], a[q][e]);
            }
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int e = 0; e < 8; ++e) a[q][e] = act_apply<true>(a[q][e], CVMI_ACT_SILU);
        *reinterpret_cast<u32x4*>(As + (r * TW + x0 + q) * PSTR + ck * 16) = pack16<f16>(a[q]);
      }
    }
    __syncthreads();
    // ---- pointwise 1x1 over this chunk: wave wv owns pixels [32 wv, 32 wv + 32), all N1P output channels --------
#pragma unroll
    for (int ks = 0; ks < CC / 16; ++ks) {
      const u32x4 xf = *reinterpret_cast<const u32x4*>(As + (wv * 32 + lr) * PSTR + ks * 32 + lh * 16);
#pragma unroll
      for (int i = 0; i < TN1; ++i) mma16(wf1[i][ks], xf, acc[i]);
    }
    if (cc + 1 < NCHUNK) load_wfrag((cc + 1) * CC);      // (after the MFMAs, which consume this chunk's fragments; they land during the next depthwise stage)
  }

  // chained 1x1: its weight fragments come straight from L2 and are requested before the activation epilogue
  u32x4 w2f[TN2 > 0 ? TN2 : 1][N1P / 16];
  if constexpr (N2P > 0) {
#pragma unroll
    for (int i = 0; i < TN2; ++i)
#pragma unroll
      for (int ks = 0; ks < N1P / 16; ++ks)
        w2f[i][ks] = *reinterpret_cast<const u32x4*>(p.w2 + ((size_t)(i * 32 + lr) * p.kpad2 + ks * 16 + lh * 8) * 2);
  }
  __syncthreads();                                    // every wave is done with patch / operand tile: the activated tile overlays them

  // ---- epilogue 1: bias + SiLU -> fp16 tile Ht [128 pixels][N1P] ---------------------------------------------------
  char* const Ht = smem;
#pragma unroll
  for (int i = 0; i < TN1; ++i) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int nl = i * 32 + 8 * q + 4 * lh;
      const f32x4 bv = *reinterpret_cast<const f32x4*>(b1s + nl);
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = act_apply<true>(acc[i][4 * q + e] + bv[e], CVMI_ACT_SILU);
      const f16x4 hv = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
      *reinterpret_cast<f16x4*>(Ht + (wv * 32 + lr) * HSTR + nl * 2) = hv;
    }
  }
  __syncthreads();

  const char* out_t = Ht;
  int ostr = HSTR, nout = p.N1;
  if constexpr (N2P > 0) {
    // ---- chained 1x1 (no activation): B operand = the wave's own 32 rows of Ht ---------------------------------
    f32x16 acc2[TN2];
#pragma unroll
    for (int i = 0; i < TN2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[i][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < N1P / 16; ++ks) {
      const u32x4 xf = *reinterpret_cast<const u32x4*>(Ht + (wv * 32 + lr) * HSTR + ks * 32 + lh * 16);
#pragma unroll
      for (int i = 0; i < TN2; ++i) mma16(w2f[i][ks], xf, acc2[i]);
    }
    char* const Ot = As;                              // dead since the last chunk's MFMAs (barrier above)
#pragma unroll
    for (int i = 0; i < TN2; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nl = i * 32 + 8 * q + 4 * lh;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(b2s + nl);
        const f16x4 hv = {(f16)(acc2[i][4 * q] + bv[0]), (f16)(acc2[i][4 * q + 1] + bv[1]), (f16)(acc2[i][4 * q + 2] + bv[2]),
                          (f16)(acc2[i][4 * q + 3] + bv[3])};
        *reinterpret_cast<f16x4*>(Ot + (wv * 32 + lr) * OSTR + nl * 2) = hv;
      }
    }
    __syncthreads();
    out_t = Ot; ostr = OSTR; nout = p.N2;
  }

  // ---- 16-byte channel-contiguous stores (element-wise on a ragged channel tail: padding lanes stay untouched) --------
  const int nch = (nout + 7) >> 3;
  for (int idx = tid; idx < BMT * nch; idx += 256) {
    const int row = idx / nch, ch = idx - row * nch;
    const int oy = oy0 + row / TW, ox = ox0 + row % TW;
    if (oy >= p.H || ox >= p.W) continue;
    const u32x4 cv = *reinterpret_cast<const u32x4*>(out_t + row * ostr + ch * 16);
    char* yp = p.y + ((((size_t)b * p.H + oy) * p.W + ox) * p.y_ld + ch * 8) * 2;
    if (ch * 8 + 8 <= nout) {
      *reinterpret_cast<u32x4*>(yp) = cv;
    } else {
      const f16x8 hv = __builtin_bit_cast(f16x8, cv);
      for (int e = 0; e < nout - ch * 8; ++e) reinterpret_cast<f16*>(yp)[e] = hv[e];
    }
  }
}

template <int CC, int NCHUNK, int N1P, int N2P>
int launch_dwpw(DwPwArgs& a, int B, hipStream_t stream) {
  constexpr int PSTR = CC * 2 + 16;
  constexpr size_t lds = (size_t)PH * PW * PSTR + (size_t)BMT * PSTR + 9 * CC * 2 + (CC + N1P + (N2P > 0 ? N2P : 4)) * sizeof(float);
  static_assert(lds <= 64 * 1024, "two to three workgroups per CU");
  static bool attr_done = false;
  if (!attr_done && lds > 64 * 1024) {
    CVMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dwpw_kernel<CC, NCHUNK, N1P, N2P>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  const long long blocks = (long long)B * a.tiles_y * a.tiles_x;
  CVMI_CHECK(blocks > 0 && blocks < (1ll << 31), "dwpw: bad grid");
  hipLaunchKernelGGL((dwpw_kernel<CC, NCHUNK, N1P, N2P>), dim3((unsigned)blocks), dim3(256), lds, stream, a);
  CVMI_LAUNCH_CHECK();
  return 0;
}

}  // namespace

// built configurations: pointwise width padded to 64 or 96 columns; the chained class conv (<= 64 outputs) for c3 = 64 / 80
static int dwpw_config(int C, int N1, int N2) {
  if (N1 <= 0 || N1 > 96 || N1 % 8 != 0 || N2 < 0 || N2 > 64) return 0;
  const int wide = N1 > 64;
  if (N2 == 0) return (C == 64 || C == 128 || C == 256) ? (wide ? 2 : 1) : 0;
  if (C == 64 && N1 <= 64) return 3;
  if (C == 80 && N1 <= 80) return 4;
  return 0;
}

extern "C" int cvmi_dwpw_supported(int C, int N1, int N2, int dtype) { return dtype == CVMI_F16 && dwpw_config(C, N1, N2) != 0; }

extern "C" int cvmi_dwpw(const cvmi_dwpw_desc* d, cvmi_stream_t stream_) {
  CVMI_CHECK(d != nullptr && d->x && d->y && d->wd && d->bd && d->w1 && d->b1, "dwpw: null pointer");
  CVMI_CHECK(d->N2 == 0 || (d->w2 && d->b2), "dwpw: chained conv needs w2 / b2");
  CVMI_CHECK(cvmi_dwpw_supported(d->C, d->N1, d->N2, d->dtype), "dwpw: configuration (C=%d N1=%d N2=%d) is not built", d->C, d->N1, d->N2);
  const int cfg = dwpw_config(d->C, d->N1, d->N2);
  const int nout = d->N2 > 0 ? d->N2 : d->N1;
  CVMI_CHECK(d->B > 0 && d->H > 0 && d->W > 0 && d->x_ld >= d->C && d->x_ld % 8 == 0 && d->y_ld >= nout && d->y_ld % 8 == 0, "dwpw: bad shape / ld");
  CVMI_CHECK((((uintptr_t)d->x | (uintptr_t)d->y | (uintptr_t)d->wd | (uintptr_t)d->w1 | (uintptr_t)d->w2) & 15) == 0, "dwpw: tensors must be 16-byte aligned");
  CVMI_CHECK(d->kpad1 >= d->C && d->kpad1 % 8 == 0 && (d->N2 == 0 || (d->kpad2 >= (cfg == 3 ? 64 : 96) && d->kpad2 % 8 == 0)), "dwpw: Kpad too small");
  DwPwArgs a;
  a.x = (const char*)d->x; a.wd = (const char*)d->wd; a.bd = d->bd; a.w1 = (const char*)d->w1; a.b1 = d->b1;
  a.w2 = (const char*)d->w2; a.b2 = d->b2; a.y = (char*)d->y;
  a.x_ld = d->x_ld; a.y_ld = d->y_ld; a.kpad1 = d->kpad1; a.kpad2 = d->kpad2;
  a.H = d->H; a.W = d->W; a.N1 = d->N1; a.N2 = d->N2;
  a.tiles_x = cdiv(d->W, TW); a.tiles_y = cdiv(d->H, TH);
  hipStream_t s = (hipStream_t)stream_;
  if (cfg == 3) return launch_dwpw<64, 1, 64, 64>(a, d->B, s);
  if (cfg == 4) return launch_dwpw<80, 1, 96, 64>(a, d->B, s);
  if (cfg == 1) {
    if (d->C == 64) return launch_dwpw<64, 1, 64, 0>(a, d->B, s);
    if (d->C == 128) return launch_dwpw<64, 2, 64, 0>(a, d->B, s);
    return launch_dwpw<64, 4, 64, 0>(a, d->B, s);
  }
  if (d->C == 64) return launch_dwpw<64, 1, 96, 0>(a, d->B, s);
  if (d->C == 128) return launch_dwpw<64, 2, 96, 0>(a, d->B, s);
  return launch_dwpw<64, 4, 96, 0>(a, d->B, s);
}
