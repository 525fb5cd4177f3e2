// Spatially tiled KxK convolution on MFMA for the HBM-bound, small-channel layers of the detector.
//
// The implicit-GEMM kernel (igemm.hip) computes a gather address per 16-byte chunk; for layers whose K is short
// (3x3 over 8..64 channels) that arithmetic, not HBM, sets the time.  Here a workgroup owns an 8 x 16 tile of
// output pixels: the input patch it needs ((8-1)*S+KS rows x (16-1)*S+KS columns x CC channels) is staged into LDS
// once with coalesced 16-byte loads, and every MFMA pixel fragment is then a ds_read_b128 at
// "lane base + compile-time tap offset" -- no per-chunk address math, and the 3x3 halo is re-read from LDS
// instead of from L2.  Weights of the current channel chunk ([BN][KS*KS*CC]) sit in LDS next to the patch.
// Operand roles, accumulator layout and the transposing epilogue are those of igemm.hip.
#include "common.hpp"
#include <stdlib.h>

namespace {

constexpr int TW = 16;                             // tile width; height TH = 8 (128 pixels) or 16 (256 pixels per workgroup)

struct TileArgs {
  const char* x; const char* w; const float* bias; const char* res; char* y;
  int x_ld, res_ld, y_ld;
  int Cin, H, W, OH, OW, N, Kpad, pad;
  int act;
  int tiles_x, tiles_y, nb_n;
  int bias_off;                                    // LDS byte offset of the parked bias row
};

template <typename T> struct MmaT;
template <> struct MmaT<f16> {
  __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};
template <> struct MmaT<float> {
  __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c) {
    const f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
#pragma unroll
    for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[e], c, 0, 0, 0);
  }
};

template <typename T, int KS, int S, int CC, int BN, int WM, int WN, int TH>
__global__ __launch_bounds__(256, (TH == 16 ? 2 : 3)) void conv_tile_kernel(const TileArgs p) {
  constexpr int BMT = TH * TW;
  constexpr int ES = sizeof(T);
  constexpr int VEC = 16 / ES;
  constexpr int PH = (TH - 1) * S + KS, PW = (TW - 1) * S + KS;
  constexpr int PSTR = CC * ES + 16;                 // patch pixel stride (bytes): odd multiple of 16
  constexpr int CCH = CC * ES / 16;                  // 16-byte chunks per patch pixel
  // PAIR: a pixel carries only 16 bytes of channels (fp16, CC = 8), half an MFMA k-step: the two lane halves then
  // take two consecutive TAPS instead of two channel halves (weights are [tap][CC] contiguous, so their fragment
  // is unchanged); an odd tap count is padded with a zero-weight tap.
  constexpr bool PAIR = (CC * ES == 16);
  constexpr int NTAP = KS * KS, NTAPP = PAIR ? (NTAP + 1) / 2 * 2 : NTAP;
  constexpr int WDATA = NTAPP * CC * ES;
  constexpr int WROW = WDATA + ((WDATA / 16) % 2 == 0 ? 16 : 32);   // weight slab row stride: odd multiple of 16
  constexpr int WCH = NTAPP * CCH;                   // 16-byte chunks per weight row
  constexpr int PATCH_B = PH * PW * PSTR;
  constexpr int WTM = BMT / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "4 waves");
  constexpr int CROWB = BN * ES + 16;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const patch = smem;
  char* const wslab = smem + PATCH_B;

  const int tid = threadIdx.x;
  int bid = blockIdx.x;
  const int bn = bid % p.nb_n; bid /= p.nb_n;
  const int tx = bid % p.tiles_x; bid /= p.tiles_x;
  const int ty = bid % p.tiles_y;
  const int b = bid / p.tiles_y;
  const int oy0 = ty * TH, ox0 = tx * TW, n0 = bn * BN;
  const int iy0 = oy0 * S - p.pad, ix0 = ox0 * S - p.pad;

  const int wv = tid >> 6, lane = tid & 63;
  const int wm = wv % WM, wn = wv / WM;
  const int lr = lane & 31, lh = lane >> 5;

  int pix_base[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int pl = wm * WTM + j * 32 + lr;
    pix_base[j] = (((pl / TW) * S) * PW + (pl % TW) * S) * PSTR + (PAIR ? 0 : lh * 16);
  }
  const char* wbase = wslab + (wn * WTN + lr) * WROW + lh * 16;

  f32x16 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nchunks = p.Cin / CC;
  // ---- staging: the input patch (zero outside the image) and the weight slab [BN][taps][CC] of a channel chunk; every
  //      global load of the chunk is issued before the first LDS write (ONE memory round trip), and the NEXT chunk's
  //      loads are issued before the MFMAs of the current one (register double buffer), so a small grid -- one workgroup
  //      per CU at the 20x20 level -- does not pay a dependent round trip per chunk.  The bias row is fetched with the
  //      first chunk and parked in LDS: the epilogue would otherwise start with one more dependent global load.
  constexpr int NP = (PH * PW * CCH + 255) / 256, NW = (BN * WCH + 255) / 256;
  u32x4 pv[NP], wv_[NW];
  bool pok[NP], wok[NW];                              // masks applied at LDS-write time (loads stay back to back)
  float* const bias_s = reinterpret_cast<float*>(smem + p.bias_off);
  const float bias_v = tid < BN ? p.bias[n0 + tid] : 0.f;     // (bias is padded to the column tile)
  auto load_chunk = [&](int c0) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int id = tid + i * 256;
      const int pp = id / CCH, ch = id - pp * CCH;
      const int pr = pp / PW, pc = pp - pr * PW;
      const int iy = iy0 + pr, ix = ix0 + pc;
      const bool ok = id < PH * PW * CCH && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const int iyc = ok ? iy : 0, ixc = ok ? ix : 0;                  // branch-free: clamped address + select
      pv[i] = *reinterpret_cast<const u32x4*>(p.x + ((((size_t)b * p.H + iyc) * p.W + ixc) * p.x_ld + c0 + (ok ? ch * VEC : 0)) * ES);
      pok[i] = ok;
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int id = tid + i * 256;
      const int n = id / WCH, r = id - n * WCH;
      const int tap = r / CCH, ch = r - tap * CCH;
      const bool ok = id < BN * WCH && tap < NTAP;
      wv_[i] = *reinterpret_cast<const u32x4*>(p.w + ((size_t)(n0 + (ok ? n : 0)) * p.Kpad + (ok ? tap * p.Cin + c0 + ch * VEC : 0)) * ES);
      wok[i] = ok;
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int id = tid + i * 256;
      const int pp = id / CCH, ch = id - pp * CCH;
      if (id < PH * PW * CCH) *reinterpret_cast<u32x4*>(patch + pp * PSTR + ch * 16) = pok[i] ? pv[i] : u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int id = tid + i * 256;
      const int n = id / WCH, r = id - n * WCH;
      if (id < BN * WCH) *reinterpret_cast<u32x4*>(wslab + n * WROW + r * 16) = wok[i] ? wv_[i] : u32x4{0u, 0u, 0u, 0u};
    }
  };
  load_chunk(0);
  for (int cc = 0; cc < nchunks; ++cc) {
    if (cc) __syncthreads();                         // previous chunk fully consumed
    store_chunk();
    if (cc == 0 && tid < BN) bias_s[tid] = bias_v;
    __syncthreads();
    if (cc + 1 < nchunks) load_chunk((cc + 1) * CC);
    // ---- MFMA over taps x channel steps ---------------------------------------------------------------
    if constexpr (PAIR) {
#pragma unroll
      for (int tp = 0; tp < NTAPP / 2; ++tp) {
        const int t0 = 2 * tp, t1 = (2 * tp + 1 < NTAP) ? 2 * tp + 1 : 2 * tp;     // padded tap: any valid address (weight is zero)
        const int off0 = ((t0 / KS) * PW + (t0 % KS)) * PSTR, off1 = ((t1 / KS) * PW + (t1 % KS)) * PSTR;
        const int off = lh ? off1 : off0;
        u32x4 wf[TN], xf[TM];
#pragma unroll
        for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const u32x4*>(wbase + i * 32 * WROW + tp * 32);
#pragma unroll
        for (int j = 0; j < TM; ++j) xf[j] = *reinterpret_cast<const u32x4*>(patch + pix_base[j] + off);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j) MmaT<T>::run(wf[i], xf[j], acc[i][j]);
      }
    } else {
#pragma unroll
      for (int tap = 0; tap < KS * KS; ++tap) {
        const int ky = tap / KS, kx = tap % KS;
#pragma unroll
        for (int c = 0; c < CC * ES / 32; ++c) {
          u32x4 wf[TN], xf[TM];
#pragma unroll
          for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const u32x4*>(wbase + i * 32 * WROW + tap * CC * ES + c * 32);
#pragma unroll
          for (int j = 0; j < TM; ++j) xf[j] = *reinterpret_cast<const u32x4*>(patch + pix_base[j] + (ky * PW + kx) * PSTR + c * 32);
#pragma unroll
          for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) MmaT<T>::run(wf[i], xf[j], acc[i][j]);
        }
      }
    }
  }
  __syncthreads();

  // ---- epilogue: bias + act -> LDS [128][BN] -> 16-byte channel-contiguous stores (+ residual) ----------
  char* const Ct = smem;
  constexpr bool FAST = FastMath<T>::value;
  with_act<FAST>(p.act, [&](auto actf) {
#pragma unroll
    for (int i = 0; i < TN; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nl = wn * WTN + i * 32 + 8 * q + 4 * lh;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_s + nl);
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          const int ml = wm * WTM + j * 32 + lr;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = actf(acc[i][j][4 * q + e] + bv[e]);
          char* dst = Ct + ml * CROWB + nl * ES;
          if constexpr (ES == 2) {
            f16x4 hv = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
            *reinterpret_cast<f16x4*>(dst) = hv;
          } else {
            f32x4 fv = {v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(dst) = fv;
          }
        }
      }
    }
  });
  __syncthreads();
  constexpr int NCH = BN / VEC;
  for (int idx = tid; idx < BMT * NCH; idx += 256) {
    const int row = idx / NCH, ch = idx - row * NCH;
    const int oy = oy0 + row / TW, ox = ox0 + row % TW;
    const int n = n0 + ch * VEC;
    if (oy >= p.OH || ox >= p.OW || n >= p.N) continue;
    const size_t opix = ((size_t)b * p.OH + oy) * p.OW + ox;
    u32x4 cv = *reinterpret_cast<const u32x4*>(Ct + row * CROWB + ch * 16);
    char* yp = p.y + (opix * p.y_ld + n) * ES;
    if (n + VEC <= p.N) {
      if (p.res) {
        float a[VEC], r[VEC];
        unpack16<T>(cv, a);
        unpack16<T>(*reinterpret_cast<const u32x4*>(p.res + (opix * p.res_ld + n) * ES), r);
#pragma unroll
        for (int e = 0; e < VEC; ++e) a[e] += r[e];
        cv = pack16<T>(a);
      }
      *reinterpret_cast<u32x4*>(yp) = cv;
    } else {
      float a[VEC];
      unpack16<T>(cv, a);
      for (int e = 0; e < p.N - n; ++e) {
        float v = a[e];
        if (p.res) v += (float)reinterpret_cast<const T*>(p.res + (opix * p.res_ld + n) * ES)[e];
        reinterpret_cast<T*>(yp)[e] = (T)v;
      }
    }
  }
}

template <typename T, int KS, int S, int CC, int BN, int WM, int WN, int TH>
int launch_tile(TileArgs& a, int B, hipStream_t stream) {
  constexpr int BMT = TH * TW;
  constexpr int ES = sizeof(T);
  constexpr int PH = (TH - 1) * S + KS, PW = (TW - 1) * S + KS;
  constexpr bool PAIR = (CC * ES == 16);
  constexpr int NTAPP = PAIR ? (KS * KS + 1) / 2 * 2 : KS * KS;
  constexpr int WDATA = NTAPP * CC * ES;
  constexpr int WROW = WDATA + ((WDATA / 16) % 2 == 0 ? 16 : 32);
  constexpr size_t stage = (size_t)PH * PW * (CC * ES + 16) + (size_t)BN * WROW;
  constexpr size_t epi = (size_t)BMT * (BN * ES + 16);
  constexpr size_t lds0 = ((stage > epi ? stage : epi) + 15) / 16 * 16;
  constexpr size_t lds = lds0 + BN * sizeof(float);           // + the parked bias row
  a.bias_off = (int)lds0;
  static bool attr_done = false;
  if (!attr_done && lds > 64 * 1024) {
    CVMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_tile_kernel<T, KS, S, CC, BN, WM, WN, TH>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  a.nb_n = cdiv(a.N, BN);
  a.tiles_y = cdiv(a.OH, TH);
  const long long blocks = (long long)B * a.tiles_y * a.tiles_x * a.nb_n;
  CVMI_CHECK(blocks > 0 && blocks < (1ll << 31), "conv_tile: bad grid");
  hipLaunchKernelGGL((conv_tile_kernel<T, KS, S, CC, BN, WM, WN, TH>), dim3((unsigned)blocks), dim3(256), lds, stream, a);
  CVMI_LAUNCH_CHECK();
  return 0;
}

template <typename T, int KS, int S, int CC>
int launch_tile_n(TileArgs& a, int B, hipStream_t stream) {
  // (16 x 16 tiles were measured: no gain over 8 x 16 on any YOLO11-n layer, so only the smaller tile is built)
  if (a.N <= 32) return launch_tile<T, KS, S, CC, 32, 4, 1, 8>(a, B, stream);
  if (a.N <= 64) {
    // small maps (the 20 x 20 level at batch 32: 192 tiles of 8 x 16 for 256 CUs): 4 x 16 tiles double the workgroups and
    // halve each one's dependent chain (stage, barrier, 9 taps of MFMAs, store)
    static const int th4 = getenv("CVMI_TILE_TH4") ? atoi(getenv("CVMI_TILE_TH4")) : 1;       // tuning experiments only
    if (th4 && sizeof(T) == 2 && (long long)B * cdiv(a.OH, 8) * a.tiles_x <= 256) return launch_tile<T, KS, S, CC, 64, 2, 2, 4>(a, B, stream);
    return launch_tile<T, KS, S, CC, 64, 2, 2, 8>(a, B, stream);
  }
  return launch_tile<T, KS, S, CC, 128, 2, 2, 8>(a, B, stream);
}

template <typename T, int KS, int S>
int launch_tile_c(TileArgs& a, int B, hipStream_t stream) {
  constexpr int ES = sizeof(T);
  constexpr int TH = 8;
  constexpr int PH = (TH - 1) * S + KS, PW = (TW - 1) * S + KS;
  const int bn = a.N <= 32 ? 32 : (a.N <= 64 ? 64 : 128);
  auto lds_for = [&](int cc) { return (size_t)PH * PW * (cc * ES + 16) + (size_t)bn * (KS * KS * cc * ES + 32); };
  // widest channel chunk that still leaves room for >= 2 workgroups per CU (160 KB LDS)
  if (a.Cin % 32 == 0 && lds_for(32) <= 72 * 1024) return launch_tile_n<T, KS, S, 32>(a, B, stream);
  if (a.Cin % 16 == 0 && lds_for(16) <= 72 * 1024) return launch_tile_n<T, KS, S, 16>(a, B, stream);
  if (a.Cin % 8 == 0 && a.Cin % 16 != 0 && lds_for(8) <= 72 * 1024) return launch_tile_n<T, KS, S, 8>(a, B, stream);
  return -1;
}

template <typename T>
int launch_tile_t(TileArgs& a, int B, int KS, int S, hipStream_t stream) {
  if (KS == 3 && S == 1) return launch_tile_c<T, 3, 1>(a, B, stream);
  if (KS == 3 && S == 2) return launch_tile_c<T, 3, 2>(a, B, stream);
  if (KS == 2 && S == 1) return launch_tile_c<T, 2, 1>(a, B, stream);
  return -1;
}

}  // namespace

// Returns 0 on success, 1 on error, -1 when the shape is not covered (caller falls back to the implicit GEMM).
int cvmi_conv_tile_try(const cvmi_conv_desc* d, hipStream_t stream) {
  if (d->c1 != 0 || d->up0 != 0 || d->scalar_gather || d->out_f32 || d->res_mod || d->act_after_res || d->shuffle_cout) return -1;
  if (d->KH != d->KW || !((d->KH == 3 && (d->stride == 1 || d->stride == 2)) || (d->KH == 2 && d->stride == 1))) return -1;
  // measured on YOLO11-n B=32: the tile kernel wins up to 128 input channels at stride 1 and 64 at stride 2; deeper
  // layers are MFMA-bound and the LDS-tiled GEMM pipeline of igemm.hip is the better fit
  static const int s2max = getenv("CVMI_TILE_S2MAX") ? atoi(getenv("CVMI_TILE_S2MAX")) : 64;       // tuning experiments only
  static const int s1max = getenv("CVMI_TILE_S1MAX") ? atoi(getenv("CVMI_TILE_S1MAX")) : 128;
  if (d->c0 > (d->stride == 1 ? s1max : s2max) || d->N > 128) return -1;
  const int es = d->dtype == CVMI_F16 ? 2 : 4;
  if (d->c0 % (16 / es) != 0) return -1;
  TileArgs a;
  a.x = (const char*)d->x0; a.w = (const char*)d->w; a.bias = d->bias; a.res = (const char*)d->res; a.y = (char*)d->y;
  a.x_ld = d->x0_ld; a.res_ld = d->res_ld; a.y_ld = d->y_ld;
  a.Cin = d->c0; a.H = d->H; a.W = d->W; a.OH = d->OH; a.OW = d->OW; a.N = d->N; a.Kpad = d->Kpad; a.pad = d->pad; a.act = d->act;
  a.tiles_x = cdiv(d->OW, TW); a.tiles_y = 0; a.nb_n = 1;
  if (d->dtype == CVMI_F16) return launch_tile_t<f16>(a, d->B, d->KH, d->stride, stream);
  return launch_tile_t<float>(a, d->B, d->KH, d->stride, stream);
}
