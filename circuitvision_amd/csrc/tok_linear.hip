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

constexpr int TL_NW = 8;                                        // waves per workgroup: 256 rows
constexpr int TL_STG = 32 * 144 + 256;                          // per-wave transposition stage: 32 rows x (128 B + 16 B pad) + a (mean, rstd) table

template <int K> struct TlCfg {
  static constexpr int KS = K / 16, KS1 = KS + 1;
  static constexpr int CHB = KS1 * 1024;                        // bytes per 32-channel weight chunk
  static constexpr int SLOTS = 3;                                // ring depth
  static constexpr int LDS = SLOTS * CHB + TL_NW * TL_STG;
};

__device__ __forceinline__ float red8(float v) {                // sum over the 8 lanes that share a row in the coalesced pattern
  v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
  return v;
}

// LN = true: `in` is the f32 stream (ld in_ld), normalised with gamma / beta / eps.  LN = false: `in` is fp16 [rows, in_ld].
// RES = true: out is f32 (ld out_ld), out[r, n] += y.  RES = false: out is fp16.
//
// Global traffic is issued in the COALESCED pattern -- 8 (f32) or 4 (fp16) lanes per 128-/64-byte row piece, 8 / 16 rows per
// wave-instruction -- and transposed to / from the MFMA's "lane = row" layout through a 4.5 KB wave-private LDS stage.  Issued
// directly in the fragment layout every lane of a load / store touches a different cache line (64 lines per instruction: measured,
// the texture-address path was then busy for ~40 % of the launch and the MFMA pipe for 30 %).
template <int K, bool LN, bool RES>
__global__ __launch_bounds__(TL_NW * 64, 2) void tok_linear_kernel(const void* __restrict__ in, int in_ld, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, float eps, const char* __restrict__ wp,
                                                                   void* __restrict__ out, int out_ld, long long rows, int N, int act) {
  using Cfg = TlCfg<K>;
  constexpr int KS = Cfg::KS, KS1 = Cfg::KS1, CHB = Cfg::CHB, SLOTS = Cfg::SLOTS, NW = TL_NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int lr = lane & 31, lh = lane >> 5;
  const long long wrow0 = ((long long)blockIdx.x * NW + wv) * 32;                  // first row of this wave (rows % 256 == 0: all valid)
  char* const stage = smem + SLOTS * CHB + wv * TL_STG;
  const int nch = (N + 31) / 32;

  // chunk j -> ring slot j % SLOTS; wave w moves fragments w, w + 8, ...
  auto issue_chunk = [&](int j) {
    const char* src = wp + (size_t)j * CHB + lane * 16;
    char* dst = smem + (j % SLOTS) * CHB;
#pragma unroll
    for (int f = 0; f < (KS1 + NW - 1) / NW; ++f) {
      const int fi = f * NW + wv;
      if (fi < KS1)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)fi * 1024),
                                         (__attribute__((address_space(3))) void*)(dst + fi * 1024), 16, 0, 0);
    }
  };
#pragma unroll
  for (int j = 0; j < SLOTS - 1; ++j)
    if (j < nch) issue_chunk(j);

  // ---- B fragments: lane (row lr, half lh) holds in[row][16 s + 8 lh .. + 7]
  u32x4 xn[KS1];
  if constexpr (LN) {
    // Coalesced pattern: lane -> (sub-row sr = lane >> 3, 16-byte piece pc = lane & 7); instruction i of a 32-channel slab covers rows
    // 8 i + sr.  Pass 1: row means.  Pass 2: row variances about the mean.  Pass 3: each slab is parked in the stage and read back in the
    // fragment layout, normalised and rounded to fp16 once.  (Passes 2 and 3 re-read the rows from L2.)
    constexpr int NSL = (K + 31) / 32;                          // slabs (the last one of K = 144 is half a slab)
    const int sr = lane >> 3, pc = lane & 7;
    const float* xb = reinterpret_cast<const float*>(in) + (wrow0 + sr) * (long long)in_ld + pc * 4;
    float mean[4], rstd[4];
    {
      float sm[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 3
      for (int sl = 0; sl < NSL; ++sl) {
        if (sl * 32 + pc * 4 < K) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xb + (long long)(8 * i) * in_ld + sl * 32);
            sm[i] += (v[0] + v[1]) + (v[2] + v[3]);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) mean[i] = red8(sm[i]) / (float)K;
      float sq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 3
      for (int sl = 0; sl < NSL; ++sl) {
        if (sl * 32 + pc * 4 < K) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xb + (long long)(8 * i) * in_ld + sl * 32);
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[e] - mean[i]; sq[i] = fmaf(d, d, sq[i]); }
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) rstd[i] = 1.0f / sqrtf(red8(sq[i]) / (float)K + eps);
    }
    // (mean, rstd) of row lr for the fragment layout: through a 32-entry table at the end of the stage
    float* const tab = reinterpret_cast<float*>(stage + 32 * 144);
    if (pc == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { tab[2 * (8 * i + sr)] = mean[i]; tab[2 * (8 * i + sr) + 1] = rstd[i]; }
    }
    const float my_mean = tab[2 * lr], my_rstd = tab[2 * lr + 1];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl) {
      __builtin_amdgcn_sched_barrier(0);                       // one slab's loads in flight at a time (register budget)
      const bool full = sl * 32 + 32 <= K;                     // compile-time after unrolling
      if (sl * 32 + pc * 4 < K) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(xb + (long long)(8 * i) * in_ld + sl * 32);
          *reinterpret_cast<f32x4*>(stage + (8 * i + sr) * 144 + pc * 16) = v;
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < (full ? 2 : 1); ++s2) {
        const int k = 2 * sl + s2;
        const f32x4 a = *reinterpret_cast<const f32x4*>(stage + lr * 144 + (16 * s2 + 8 * lh) * 4);
        const f32x4 b = *reinterpret_cast<const f32x4*>(stage + lr * 144 + (16 * s2 + 8 * lh) * 4 + 16);
        const float* gp = gamma + 16 * k + 8 * lh;
        const float* bp = beta + 16 * k + 8 * lh;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp), g1 = *reinterpret_cast<const f32x4*>(gp + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 4);
        f16x8 h;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          h[e] = (f16)((a[e] - my_mean) * my_rstd * g0[e] + b0[e]);
          h[4 + e] = (f16)((b[e] - my_mean) * my_rstd * g1[e] + b1[e]);
        }
        xn[k] = __builtin_bit_cast(u32x4, h);
      }
    }
  } else {
    const f16* xr = reinterpret_cast<const f16*>(in) + (wrow0 + lr) * (long long)in_ld;
#pragma unroll
    for (int k = 0; k < KS; ++k) xn[k] = *reinterpret_cast<const u32x4*>(xr + 16 * k + 8 * lh);
  }
  {
    const u32x4 one = {lh == 0 ? 0x3C003C00u : 0u, 0u, 0u, 0u};   // bias step: constant-1 columns k = K, K + 1
    xn[KS] = one;
  }

  // epilogue of chunk `ch`: the accumulator (lane = row lr, register group g -> channels 8 g + 4 lh .. + 3) goes through the stage and
  // leaves as whole 64-byte (fp16) / 128-byte (f32) row pieces
  auto epilogue = [&](const f32x16& acc, int ch) {
    if constexpr (RES) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        *reinterpret_cast<f32x4*>(stage + lr * 144 + (8 * g + 4 * lh) * 4) = v;
      }
      const int sr = lane >> 3, pc = lane & 7;
      const int c0 = 32 * ch + pc * 4;
      if (c0 < N) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(stage + (8 * i + sr) * 144 + pc * 16);
          float* o = reinterpret_cast<float*>(out) + (wrow0 + 8 * i + sr) * (long long)out_ld + c0;
          f32x4 r4 = *reinterpret_cast<const f32x4*>(o);
#pragma unroll
          for (int e = 0; e < 4; ++e) r4[e] += v[e];
          *reinterpret_cast<f32x4*>(o) = r4;
        }
      }
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f16x4 h4;
        if (act == CVMI_ACT_GELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) h4[e] = (f16)gelu_fast(acc[4 * g + e]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) h4[e] = (f16)acc[4 * g + e];
        }
        *reinterpret_cast<f16x4*>(stage + lr * 80 + (8 * g + 4 * lh) * 2) = h4;
      }
      const int sr = lane >> 2, pc = lane & 3;
      const int c0 = 32 * ch + pc * 8;
      if (c0 < N) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(stage + (16 * i + sr) * 80 + pc * 16);
          *reinterpret_cast<u32x4*>(reinterpret_cast<f16*>(out) + (wrow0 + 16 * i + sr) * (long long)out_ld + c0) = v;
        }
      }
    }
  };

  constexpr int PF = K >= 576 ? 6 : 8;                          // ds_read_b128 kept in flight ahead of their MFMA
  f32x16 prev;
#pragma unroll
  for (int r = 0; r < 16; ++r) prev[r] = 0.f;
#pragma unroll 1
  for (int j = 0; j < nch; ++j) {
    // Every wave waits for its OWN DMA pieces (explicitly: hipcc puts no vmcnt wait in front of a barrier for LDS-DMA writes),
    // then the barrier publishes chunk j and frees slot (j - 1) % SLOTS.  vmcnt(0) also covers the wave's own stores, which is
    // why the epilogue of chunk j - 1 is issued AFTER this barrier: its stores then have a whole chunk of MFMAs to complete in.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (j + SLOTS - 1 < nch) issue_chunk(j + SLOTS - 1);
    const char* const buf = smem + (j % SLOTS) * CHB + lane * 16;
    u32x4 ring[PF];
#pragma unroll
    for (int f = 0; f < PF; ++f) ring[f] = *reinterpret_cast<const u32x4*>(buf + f * 1024);
    if (j > 0) epilogue(prev, j - 1);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int f = 0; f < KS1; ++f) {
      const f16x8 a = __builtin_bit_cast(f16x8, ring[f % PF]);
      if (f + PF < KS1) ring[f % PF] = *reinterpret_cast<const u32x4*>(buf + (f + PF) * 1024);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(f16x8, xn[f]), acc, 0, 0, 0);
    }
    prev = acc;
  }
  epilogue(prev, nch - 1);
}

template <int K, bool LN, bool RES>
int launch_tl(const void* in, int in_ld, const float* gamma, const float* beta, float eps, const void* wp, void* out, int out_ld, long long rows,
              int N, int act, hipStream_t s) {
  using Cfg = TlCfg<K>;
  static hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&tok_linear_kernel<K, LN, RES>), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
  CVMI_HIP(attr);
  hipLaunchKernelGGL((tok_linear_kernel<K, LN, RES>), dim3((unsigned)(rows / 256)), dim3(TL_NW * 64), Cfg::LDS, s, in, in_ld, gamma, beta, eps,
                     (const char*)wp, out, out_ld, rows, N, act);
  CVMI_LAUNCH_CHECK();
  return 0;
}

template <int K>
int dispatch_tl(bool ln, bool res, const void* in, int in_ld, const float* gamma, const float* beta, float eps, const void* wp, void* out, int out_ld,
                long long rows, int N, int act, hipStream_t s) {
  if (ln && !res) return launch_tl<K, true, false>(in, in_ld, gamma, beta, eps, wp, out, out_ld, rows, N, act, s);
  if (!ln && res) return launch_tl<K, false, true>(in, in_ld, gamma, beta, eps, wp, out, out_ld, rows, N, act, s);
  if (!ln && !res) return launch_tl<K, false, false>(in, in_ld, gamma, beta, eps, wp, out, out_ld, rows, N, act, s);
  return launch_tl<K, true, true>(in, in_ld, gamma, beta, eps, wp, out, out_ld, rows, N, act, s);
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
  CVMI_CHECK(out_f32_residual || (N % 8 == 0 && out_ld % 8 == 0), "tok_linear: fp16 output needs N and out_ld to be multiples of 8");
  switch (K) {
    case 144: return dispatch_tl<144>(ln, res, in, in_ld, gamma, beta, eps, w_packed, out, out_ld, rows, N, act, s);
    case 288: return dispatch_tl<288>(ln, res, in, in_ld, gamma, beta, eps, w_packed, out, out_ld, rows, N, act, s);
    default: return dispatch_tl<576>(ln, res, in, in_ld, gamma, beta, eps, w_packed, out, out_ld, rows, N, act, s);
  }
}
