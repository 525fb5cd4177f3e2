// Token-stationary linear layer for the short-K GEMMs of Hiera (K = C <= 576):
//     y[r, n] = act( sum_k in[r, k] * W[n, k] + b[n] )        in = LayerNorm(x[r, :]) (f32 stream, fused) or an fp16 matrix
//     out: fp16 [rows, N]   or   f32 residual stream updated in place:  res[r, n] += y[r, n]
// (sam2 hieradet MultiScaleBlock: `qkv(norm1(x))`, `x = shortcut + proj(attn)`, `mlp.layers[0](norm2(x))` + GELU; behind
//  /root/reference/src/sam2_infer.py:226.)
//
// Why not the tiled GEMM: with K = 144 .. 576 a 256 x 256 tile runs only 3 .. 9 K-steps, so its prologue, epilogue and the
// LayerNorm pass in front of it weigh as much as the MFMAs.  Here a wave keeps its 32 rows as the B fragments of the MFMA for
// the whole launch (K/16 x 4 registers) and walks the output channels 32 at a time: per chunk K/16 + 1 MFMAs (the extra k-step
// carries the bias as a hi + lo fp16 pair against constant-1 columns), A = the weight chunk in fragment order, streamed
// L2 -> LDS by global_load_lds into a ring and read lane-linearly (conflict-free ds_read_b128, kept PF deep in flight).  Only
// the A operand comes from LDS -- half the LDS traffic of a tiled GEMM -- and the accumulator of a chunk is its epilogue:
// lane = row, registers = 4 consecutive channels per group -> 8-byte fp16 / 16-byte f32 stores.
// 8 waves per workgroup (two per SIMD: one wave's epilogue VALU and stores overlap the other's MFMAs), 256 rows per workgroup.
#include "common.hpp"
#include <stdlib.h>

namespace {

// Variant knobs (measured):
//   TT  token tiles (32 rows each) per wave: TT = 2 uses every A fragment read from LDS for two MFMAs (half the LDS traffic per
//       flop) at 8 / TT waves per workgroup -- one wave per SIMD, 2 x the resident B fragments;
//   CN  32-channel chunks per barrier interval.
template <int K, int TT, int CN> struct TlCfg {
  static constexpr int KS = K / 16, KS1 = KS + 1;
  static constexpr int NW = 8 / TT;                             // waves per workgroup (256 rows per workgroup either way)
  static constexpr int CHB = CN * KS1 * 1024;                   // bytes per barrier interval (CN weight chunks)
  static constexpr int SLOTS = (3 * CHB <= 150 * 1024) ? 3 : 2;  // ring depth
  static constexpr int LDS = SLOTS * CHB;
};

// LN = true: `in` is the f32 stream (ld in_ld), normalised with gamma / beta / eps.  LN = false: `in` is fp16 [rows, in_ld].
// RES = true: out is f32 (ld out_ld), out[r, n] += y.  RES = false: out is fp16.
template <int K, bool LN, bool RES, int TT, int CN>
__global__ __launch_bounds__(512 / TT, TT == 1 ? 2 : 1) void tok_linear_kernel(
    const void* __restrict__ in, int in_ld, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    const char* __restrict__ wp, void* __restrict__ out, int out_ld, long long rows, int N, int act) {
  using Cfg = TlCfg<K, TT, CN>;
  constexpr int KS = Cfg::KS, KS1 = Cfg::KS1, CHB = Cfg::CHB, SLOTS = Cfg::SLOTS, NW = Cfg::NW, FR = CN * KS1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int lr = lane & 31, lh = lane >> 5;
  const long long row0 = ((long long)blockIdx.x * NW + wv) * (32 * TT) + lr;      // this lane's row in token tile 0 (tile t: + 32 t)
  const int nsc = ((N + 31) / 32 + CN - 1) / CN;                                    // barrier intervals ("super-chunks")

  // super-chunk j -> ring slot j % SLOTS; wave w moves fragments w, w + NW, ...
  auto issue_chunk = [&](int j) {
    const char* src = wp + (size_t)j * CHB + lane * 16;
    char* dst = smem + (j % SLOTS) * CHB;
#pragma unroll
    for (int f = 0; f < (FR + NW - 1) / NW; ++f) {
      const int fi = f * NW + wv;
      if (fi < FR)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)fi * 1024),
                                         (__attribute__((address_space(3))) void*)(dst + fi * 1024), 16, 0, 0);
    }
  };
#pragma unroll
  for (int j = 0; j < SLOTS - 1; ++j)
    if (j < nsc) issue_chunk(j);

  // ---- B fragments: lane (row lr, half lh) holds in[row][16 s + 8 lh .. + 7]
  u32x4 xn[TT][KS1];
#pragma unroll
  for (int t = 0; t < TT; ++t) {
    const long long row = row0 + 32 * t;
    __builtin_amdgcn_sched_barrier(0);                           // one token tile's loads at a time
    if constexpr (LN) {
      // Two passes over the row instead of K/2 live f32 registers per lane (K = 576 would need 288 of the 256 available at two
      // waves per SIMD): pass 1 accumulates sum and sum of squares of (x - x0), x0 = the row's first element (a shift that keeps
      // the single-pass variance formula well conditioned: what cancels is (mean - x0)^2, bounded by the row's own spread);
      // pass 2 re-reads the row -- from L1 / L2, the workgroup's 256 rows were touched a few hundred cycles earlier -- and writes
      // the fp16 fragments.
      const float* xr = reinterpret_cast<const float*>(in) + row * (long long)in_ld;
      const float x0 = xr[0];
      float s = 0.f, q = 0.f;
#pragma unroll 6
      for (int k = 0; k < KS; ++k) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(xr + 16 * k + 8 * lh), b = *reinterpret_cast<const f32x4*>(xr + 16 * k + 8 * lh + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float da = a[e] - x0, db = b[e] - x0;
          s += da + db;
          q = fmaf(da, da, fmaf(db, db, q));
        }
      }
      s += __shfl_xor(s, 32);
      q += __shfl_xor(q, 32);
      const float dm = s / (float)K;                           // mean - x0
      const float mean = x0 + dm;
      const float var = fmaxf(q / (float)K - dm * dm, 0.f);
      const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        if (k % 3 == 0) __builtin_amdgcn_sched_barrier(0);      // at most 3 steps' loads in flight: no hoisting of all K/16 of them
        const f32x4 a = *reinterpret_cast<const f32x4*>(xr + 16 * k + 8 * lh), b = *reinterpret_cast<const f32x4*>(xr + 16 * k + 8 * lh + 4);
        const float* gp = gamma + 16 * k + 8 * lh;
        const float* bp = beta + 16 * k + 8 * lh;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp), g1 = *reinterpret_cast<const f32x4*>(gp + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 4);
        f16x8 h;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          h[e] = (f16)((a[e] - mean) * rstd * g0[e] + b0[e]);
          h[4 + e] = (f16)((b[e] - mean) * rstd * g1[e] + b1[e]);
        }
        xn[t][k] = __builtin_bit_cast(u32x4, h);
      }
    } else {
      const f16* xr = reinterpret_cast<const f16*>(in) + row * (long long)in_ld;
#pragma unroll
      for (int k = 0; k < KS; ++k) xn[t][k] = *reinterpret_cast<const u32x4*>(xr + 16 * k + 8 * lh);
    }
    const u32x4 one = {lh == 0 ? 0x3C003C00u : 0u, 0u, 0u, 0u};   // bias step: constant-1 columns k = K, K + 1
    xn[t][KS] = one;
  }

  // epilogue of one 32-channel chunk of one token tile: lane (row, half lh), register group g -> channels 32 ch + 8 g + 4 lh .. + 3
  auto epilogue = [&](const f32x16& acc, int ch, int t) {
    const long long row = row0 + 32 * t;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c0 = 32 * ch + 8 * g + 4 * lh;
      if (c0 < N) {
        if constexpr (RES) {
          float* o = reinterpret_cast<float*>(out) + row * (long long)out_ld + c0;
          f32x4 r4 = *reinterpret_cast<const f32x4*>(o);
#pragma unroll
          for (int e = 0; e < 4; ++e) r4[e] += acc[4 * g + e];
          *reinterpret_cast<f32x4*>(o) = r4;
        } else {
          f16x4 h4;
          if (act == CVMI_ACT_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) h4[e] = (f16)gelu_fast(acc[4 * g + e]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) h4[e] = (f16)acc[4 * g + e];
          }
          *reinterpret_cast<f16x4*>(reinterpret_cast<f16*>(out) + row * (long long)out_ld + c0) = h4;
        }
      }
    }
  };

  constexpr int PF = (K >= 576 && TT == 1) ? 6 : 8;             // ds_read_b128 kept in flight ahead of their MFMAs
  f32x16 prev[CN][TT];
#pragma unroll
  for (int c = 0; c < CN; ++c)
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) prev[c][t][r] = 0.f;
#pragma unroll 1
  for (int j = 0; j < nsc; ++j) {
    // Every wave waits for its OWN DMA pieces (explicitly: hipcc puts no vmcnt wait in front of a barrier for LDS-DMA writes),
    // then the barrier publishes super-chunk j and frees slot (j - 1) % SLOTS.  vmcnt(0) also covers the wave's own stores, which
    // is why the epilogue of super-chunk j - 1 is issued AFTER this barrier: its stores then have a whole interval to complete in.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (j + SLOTS - 1 < nsc) issue_chunk(j + SLOTS - 1);
    const char* const buf = smem + (j % SLOTS) * CHB + lane * 16;
    u32x4 ring[PF];
#pragma unroll
    for (int f = 0; f < PF; ++f) ring[f] = *reinterpret_cast<const u32x4*>(buf + f * 1024);
    if (j > 0) {
#pragma unroll
      for (int c = 0; c < CN; ++c)
#pragma unroll
        for (int t = 0; t < TT; ++t) epilogue(prev[c][t], (j - 1) * CN + c, t);
    }
    f32x16 acc[CN][TT];
#pragma unroll
    for (int c = 0; c < CN; ++c)
#pragma unroll
      for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][t][r] = 0.f;
#pragma unroll
    for (int f = 0; f < FR; ++f) {
      const f16x8 a = __builtin_bit_cast(f16x8, ring[f % PF]);
      if (f + PF < FR) ring[f % PF] = *reinterpret_cast<const u32x4*>(buf + (f + PF) * 1024);
#pragma unroll
      for (int t = 0; t < TT; ++t)
        acc[f / KS1][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(f16x8, xn[t][f % KS1]), acc[f / KS1][t], 0, 0, 0);
    }
#pragma unroll
    for (int c = 0; c < CN; ++c)
#pragma unroll
      for (int t = 0; t < TT; ++t) prev[c][t] = acc[c][t];
  }
#pragma unroll
  for (int c = 0; c < CN; ++c)
#pragma unroll
    for (int t = 0; t < TT; ++t) epilogue(prev[c][t], (nsc - 1) * CN + c, t);
}

template <int K, bool LN, bool RES, int TT, int CN>
int launch_tl(const void* in, int in_ld, const float* gamma, const float* beta, float eps, const void* wp, void* out, int out_ld, long long rows,
              int N, int act, hipStream_t s) {
  using Cfg = TlCfg<K, TT, CN>;
  static hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&tok_linear_kernel<K, LN, RES, TT, CN>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
  CVMI_HIP(attr);
  hipLaunchKernelGGL((tok_linear_kernel<K, LN, RES, TT, CN>), dim3((unsigned)(rows / 256)), dim3(Cfg::NW * 64), Cfg::LDS, s, in, in_ld, gamma, beta,
                     eps, (const char*)wp, out, out_ld, rows, N, act);
  CVMI_LAUNCH_CHECK();
  return 0;
}

// variant: 0 = (TT 1, CN 1), 1 = (TT 1, CN 2), 2 = (TT 2, CN 1), 3 = (TT 2, CN 2).  TT = 2 is built for the fp16-input form only: with
// the LayerNorm prologue hipcc spills it (80 .. 630 registers), so LN launches map variant 2 -> 0 and 3 -> 1.
template <int K, bool LN, bool RES>
int dispatch_var(int variant, const void* in, int in_ld, const float* gamma, const float* beta, float eps, const void* wp, void* out, int out_ld,
                 long long rows, int N, int act, hipStream_t s) {
  if constexpr (LN) {
    if (variant & 1) return launch_tl<K, LN, RES, 1, 2>(in, in_ld, gamma, beta, eps, wp, out, out_ld, rows, N, act, s);
    return launch_tl<K, LN, RES, 1, 1>(in, in_ld, gamma, beta, eps, wp, out, out_ld, rows, N, act, s);
  } else {
    switch (variant) {
      case 1: return launch_tl<K, LN, RES, 1, 2>(in, in_ld, gamma, beta, eps, wp, out, out_ld, rows, N, act, s);
      case 2: return launch_tl<K, LN, RES, 2, 1>(in, in_ld, gamma, beta, eps, wp, out, out_ld, rows, N, act, s);
      case 3: if constexpr (K <= 288) return launch_tl<K, LN, RES, 2, 2>(in, in_ld, gamma, beta, eps, wp, out, out_ld, rows, N, act, s);
              return launch_tl<K, LN, RES, 2, 1>(in, in_ld, gamma, beta, eps, wp, out, out_ld, rows, N, act, s);
      default: return launch_tl<K, LN, RES, 1, 1>(in, in_ld, gamma, beta, eps, wp, out, out_ld, rows, N, act, s);
    }
  }
}

template <int K>
int dispatch_tl(int variant, bool ln, bool res, const void* in, int in_ld, const float* gamma, const float* beta, float eps, const void* wp, void* out,
                int out_ld, long long rows, int N, int act, hipStream_t s) {
  if (ln && !res) return dispatch_var<K, true, false>(variant, in, in_ld, gamma, beta, eps, wp, out, out_ld, rows, N, act, s);
  if (!ln && res) return dispatch_var<K, false, true>(variant, in, in_ld, gamma, beta, eps, wp, out, out_ld, rows, N, act, s);
  if (!ln && !res) return dispatch_var<K, false, false>(variant, in, in_ld, gamma, beta, eps, wp, out, out_ld, rows, N, act, s);
  return dispatch_var<K, true, true>(variant, in, in_ld, gamma, beta, eps, wp, out, out_ld, rows, N, act, s);
}

// which (TT, CN) variant a shape runs; CVMI_TOKLIN_VARIANT overrides it (measurements only)
int pick_variant(int K, int N) {
  static const char* env = getenv("CVMI_TOKLIN_VARIANT");
  if (env && env[0] >= '0' && env[0] <= '3') return env[0] - '0';
  (void)K; (void)N;
  return 0;
}

}  // namespace

extern "C" int cvmi_tok_linear_supported(int K) { return K == 144 || K == 288 || K == 576; }

extern "C" size_t cvmi_tok_linear_packed_bytes(int K, int N) {
  if (!cvmi_tok_linear_supported(K) || N <= 0) return 0;
  return (size_t)(((N + 31) / 32 + 1) / 2 * 2) * (size_t)(K / 16 + 1) * 1024;        // chunk count padded to even
}

extern "C" int cvmi_tok_linear(const void* in, int in_ld, int in_f32_layernorm, const float* gamma, const float* beta, float eps,
                               const void* w_packed, void* out, int out_ld, int out_f32_residual, long long rows, int K, int N, int act,
                               cvmi_stream_t stream_) {
  CVMI_CHECK(in && w_packed && out && rows > 0 && rows % 256 == 0 && N > 0, "tok_linear: bad arguments (rows must be a multiple of 256)");
  CVMI_CHECK(cvmi_tok_linear_supported(K), "tok_linear: K=%d is not built (144, 288, 576)", K);
  CVMI_CHECK(!in_f32_layernorm || (gamma && beta), "tok_linear: LayerNorm input needs gamma / beta");
  CVMI_CHECK(act == CVMI_ACT_NONE || (act == CVMI_ACT_GELU && !out_f32_residual), "tok_linear: act must be NONE, or GELU with fp16 output");
  CVMI_CHECK(in_ld >= K && in_ld % (in_f32_layernorm ? 4 : 8) == 0 && out_ld >= N && out_ld % 4 == 0 && N % 4 == 0 &&
                 (((uintptr_t)in | (uintptr_t)w_packed | (uintptr_t)out) & 15) == 0 &&
                 (!in_f32_layernorm || (((uintptr_t)gamma | (uintptr_t)beta) & 15) == 0),
             "tok_linear: pointers / ld not aligned (in_ld=%d out_ld=%d N=%d)", in_ld, out_ld, N);
  hipStream_t s = (hipStream_t)stream_;
  const bool ln = in_f32_layernorm != 0, res = out_f32_residual != 0;
  const int var = pick_variant(K, N);
  switch (K) {
    case 144: return dispatch_tl<144>(var, ln, res, in, in_ld, gamma, beta, eps, w_packed, out, out_ld, rows, N, act, s);
    case 288: return dispatch_tl<288>(var, ln, res, in, in_ld, gamma, beta, eps, w_packed, out, out_ld, rows, N, act, s);
    default: return dispatch_tl<576>(var, ln, res, in, in_ld, gamma, beta, eps, w_packed, out, out_ld, rows, N, act, s);
  }
}
