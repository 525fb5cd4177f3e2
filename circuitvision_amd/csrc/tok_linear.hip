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
// 8 waves per workgroup, 256 rows; the two waves of a SIMD run half a chunk interval apart (ping-pong schedule, below): one in its
// MFMAs beside the other in its epilogue VALU, stores and weight prefetch.
#include "common.hpp"
#include <type_traits>
#include <stdlib.h>

namespace {

constexpr int TL_NW = 8;

template <int K> struct TlCfg {
  static constexpr int KS = K / 16, KS1 = KS + 1;
  static constexpr int CHB = KS1 * 1024;                       // bytes per 32-channel weight chunk
  static constexpr int SLOTS = K <= 288 ? 4 : 3;                // ring depth
  static constexpr int STG = 32 * 144;                          // per-wave transposition stage of the epilogue: 32 rows x (128 + 16) bytes
  static constexpr int LDS = SLOTS * CHB + TL_NW * STG;
};

// LN = 1: `in` is the f32 stream (ld in_ld), normalised with gamma / beta / eps.  LN = 0: `in` is fp16 [rows, in_ld].  LN = 2: f32 rows
// converted as they are (the neck's lateral convs read the stage outputs of the f32 stream: no cast pass).
// RES = true: out is f32 (ld out_ld), out[r, n] += y.  RES = false: out is fp16.
// STAMP: diagnostic build (CVMI_TOKLIN_STAMP=1): wave 0 of workgroup 0 accumulates s_memtime differences of the loop's segments into
// g_tl_stamp (read by cvmi_debug_stamps).  Never used for timing runs: the stamps serialise what the real kernel overlaps.
__device__ unsigned long long g_tl_stamp[24];

// Optional extras of a launch.  pool_*: the POOL form's token grid.  stats_in: LN = 1 only -- per-row (mean, rstd) of the LayerNorm, written by the
// launch that produced the rows (the prologue then reads every row ONCE instead of twice).  stats_out: RES only -- after the in-place update, the
// (mean, rstd) over the N updated values of every row, for the LayerNorm (eps = stats_eps) of the NEXT launch.
struct TlExtra {
  int pool_w, pool_hw2;
  const float* stats_in;
  float* stats_out;
  float stats_eps;
  int stats_parts;          // 0: stats_in holds (mean, rstd) per row; P > 0: P raw (sum, sum of squares) partials per row (cvmi_conv_desc.row_stats)
};

template <int K, int LN, bool RES, bool GELU, bool TSTORE, bool STAMP = false, bool PP = false, bool POOL = false>
__global__ __launch_bounds__(TL_NW * 64, 2) void tok_linear_kernel(const void* __restrict__ in, int in_ld, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, float eps, const char* __restrict__ wp,
                                                                   void* __restrict__ out, int out_ld, long long rows, int N, const TlExtra ex) {
  using Cfg = TlCfg<K>;
  const int pool_w = ex.pool_w, pool_hw2 = ex.pool_hw2;
  constexpr int KS = Cfg::KS, KS1 = Cfg::KS1, CHB = Cfg::CHB, SLOTS = Cfg::SLOTS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int lr = lane & 31, lh = lane >> 5;
  // rows % 256 == 0 (checked by the host): every row exists.  POOL: lane quad q = lr / 4 holds the four tokens (dy, dx) = ((lr / 2) & 1, lr & 1)
  // of 2 x 2 block `prow` of the [B, H, W] token grid (pool_w = W, pool_hw2 = (H / 2)(W / 2)); out is the pooled [B, H/2, W/2, N] f32 map.
  long long row = ((long long)blockIdx.x * TL_NW + wv) * 32 + lr;
  long long prow = 0;
  if constexpr (POOL) {
    prow = row >> 2;
    const long long b = prow / pool_hw2;
    const int r = (int)(prow - b * pool_hw2), w2 = pool_w >> 1;
    const int py = r / w2, px = r - py * w2;
    row = b * 4 * pool_hw2 + (long long)(2 * py + ((lr >> 1) & 1)) * pool_w + 2 * px + (lr & 1);
  }
  const int nch = (N + 31) / 32;

  // chunk j -> ring slot j % SLOTS; wave w moves fragments w, w + 8, ...
  auto issue_chunk = [&](int j) {
    const char* src = wp + (size_t)j * CHB + lane * 16;
    char* dst = smem + (j % SLOTS) * CHB;
#pragma unroll
    for (int f = 0; f < (KS1 + TL_NW - 1) / TL_NW; ++f) {
      const int fi = f * TL_NW + wv;
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
  if constexpr (LN == 2) {
    const float* xr = reinterpret_cast<const float*>(in) + row * (long long)in_ld;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      if (k % 3 == 0) __builtin_amdgcn_sched_barrier(0);
      const f32x4 a = *reinterpret_cast<const f32x4*>(xr + 16 * k + 8 * lh), b = *reinterpret_cast<const f32x4*>(xr + 16 * k + 8 * lh + 4);
      f16x8 h;
#pragma unroll
      for (int e = 0; e < 4; ++e) { h[e] = (f16)a[e]; h[4 + e] = (f16)b[e]; }
      xn[k] = __builtin_bit_cast(u32x4, h);
    }
  } else if constexpr (LN == 1) {
    // Two passes over the row instead of K/2 live f32 registers per lane (K = 576 would need 288 of the 256 available at two
    // waves per SIMD): pass 1 accumulates sum and sum of squares of (x - x0), x0 = the row's first element (a shift that keeps
    // the single-pass variance formula well conditioned: what cancels is (mean - x0)^2, bounded by the row's own spread);
    // pass 2 re-reads the row -- from L1 / L2, the workgroup's 256 rows were touched a few hundred cycles earlier -- and writes
    // the fp16 fragments.
    const float* xr = reinterpret_cast<const float*>(in) + row * (long long)in_ld;
    float mean, rstd;
    if (ex.stats_in && ex.stats_parts == 0) {                // forwarded by the producer of these rows: pass 1 disappears
      const float2 st = *reinterpret_cast<const float2*>(ex.stats_in + 2 * row);
      mean = st.x; rstd = st.y;
    } else if (ex.stats_in) {                                // per column slice (mean, sum of squared deviations) of a tiled GEMM's epilogue, combined
      float ms = 0.f, m2 = 0.f;                              // as Chan et al. do (equal slice sizes K / P, fixed order): see tok_linear16.hip
      for (int t = 0; t < ex.stats_parts; ++t) {
        const float2 st = *reinterpret_cast<const float2*>(ex.stats_in + (row * ex.stats_parts + t) * 2);
        ms += st.x; m2 += st.y;
      }
      const float inv_p = 1.0f / (float)ex.stats_parts;
      mean = ms * inv_p;
      float dev = 0.f;
      for (int t = 0; t < ex.stats_parts; ++t) {
        const float d = ex.stats_in[(row * ex.stats_parts + t) * 2] - mean;
        dev = fmaf(d, d, dev);
      }
      rstd = 1.0f / sqrtf(fmaxf((m2 + ((float)K * inv_p) * dev) / (float)K, 0.f) + eps);
    } else {
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
      mean = x0 + dm;
      const float var = fmaxf(q / (float)K - dm * dm, 0.f);
      rstd = 1.0f / sqrtf(var + eps);
    }
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
      xn[k] = __builtin_bit_cast(u32x4, h);
    }
  } else {
    const f16* xr = reinterpret_cast<const f16*>(in) + row * (long long)in_ld;
#pragma unroll
    for (int k = 0; k < KS; ++k) xn[k] = *reinterpret_cast<const u32x4*>(xr + 16 * k + 8 * lh);
  }
  {
    const u32x4 one = {lh == 0 ? CVMI_ONE16X2 : 0u, 0u, 0u, 0u};   // bias step: constant-1 columns k = K, K + 1
    xn[KS] = one;
  }

  // Epilogue of chunk j: lane (row lr, half lh), register group g -> channels 32 j + 8 g + 4 lh .. + 3.
  // RES: the four residual loads of chunk j - 1 are ordinary loads issued right after the barrier of interval j; their first use sits in
  // the MIDDLE of the interval's MFMA sequence.  Beside in-flight LDS-DMA hipcc answers an ordinary load's first use with vmcnt(0):
  // placed there the wait finds the loads and the interval's weight prefetch long landed, and the stores that follow have the second
  // half of the MFMAs to complete in before the next barrier's vmcnt(0).
  f32x4 r4[4];
  auto res_load = [&](int j) {
    const float* o = reinterpret_cast<const float*>(out) + row * (long long)out_ld + 32 * j + 4 * lh;
#pragma unroll
    for (int g = 0; g < 4; ++g) {                              // (the last chunk of N = 144 is half a chunk: no read past the row's end --
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};                     //  for the buffer's last row that would be a read past the allocation)
      r4[g] = 32 * j + 8 * g + 4 * lh < N ? *reinterpret_cast<const f32x4*>(o + 8 * g) : z;
    }
  };
  // The accumulator has the row on the lane: written out directly, every lane of a store touches a different cache line (8-byte / 16-byte
  // pieces, 64 lines per instruction).  TSTORE: the chunk goes through a wave-private LDS stage and leaves as whole 64-byte (16-bit) /
  // 128-byte (f32) row pieces -- 4 / 8 lanes per row, 16 / 8 rows per instruction.
  float st_shift = 0.f, st_s = 0.f, st_q = 0.f;               // RES + stats_out: shifted sums over the row's updated values (this lane's half)
  char* const stage = smem + SLOTS * CHB + wv * Cfg::STG;
  const long long wrow0 = ((long long)blockIdx.x * TL_NW + wv) * 32;
  auto epilogue = [&](const f32x16& acc, int j) {
    if constexpr (POOL) {
      // 2 x 2 max over the lane quad (two DPP quad permutes per value: lanes ^ 1, lanes ^ 2), then the quad's first lane stores
      float* o = reinterpret_cast<float*>(out) + prow * (long long)out_ld + 32 * j + 4 * lh;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float a = acc[4 * g + e];
          a = fmaxf(a, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a), 0xB1, 0xF, 0xF, true)));   // quad_perm [1,0,3,2]
          a = fmaxf(a, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a), 0x4E, 0xF, 0xF, true)));   // quad_perm [2,3,0,1]
          v[e] = a;
        }
        if ((lr & 3) == 0 && 32 * j + 8 * g + 4 * lh < N) *reinterpret_cast<f32x4*>(o + 8 * g) = v;
      }
    } else if constexpr (RES) {
      float* o = reinterpret_cast<float*>(out) + row * (long long)out_ld + 32 * j + 4 * lh;
      const bool stats = ex.stats_out != nullptr;              // (uniform)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v = r4[g];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += acc[4 * g + e];
        if (stats && j == 0 && g == 0) st_shift = __shfl(v[0], lr);      // the row's first updated value (held by the lane of half 0): the variance shift
        if (32 * j + 8 * g + 4 * lh < N) {
          *reinterpret_cast<f32x4*>(o + 8 * g) = v;
          if (stats) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float dv = v[e] - st_shift; st_s += dv; st_q = fmaf(dv, dv, st_q); }
          }
        }
      }
    } else if constexpr (TSTORE) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f16x4 h4;
        if constexpr (GELU) {
          const f16x2 lo2 = gelu_fast_pk(acc[4 * g], acc[4 * g + 1]), hi2 = gelu_fast_pk(acc[4 * g + 2], acc[4 * g + 3]);
          h4 = (f16x4){lo2[0], lo2[1], hi2[0], hi2[1]};
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) h4[e] = (f16)acc[4 * g + e];
        }
        *reinterpret_cast<f16x4*>(stage + lr * 80 + (8 * g + 4 * lh) * 2) = h4;
      }
      const int sr = lane >> 2, pc = lane & 3;
      const int c0 = 32 * j + pc * 8;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(stage + (16 * i + sr) * 80 + pc * 16);
        if (c0 < N) *reinterpret_cast<u32x4*>(reinterpret_cast<f16*>(out) + (wrow0 + 16 * i + sr) * (long long)out_ld + c0) = v;
      }
    } else {
      f16* o = reinterpret_cast<f16*>(out) + row * (long long)out_ld + 32 * j + 4 * lh;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f16x4 h4;
        if constexpr (GELU) {
          const f16x2 lo2 = gelu_fast_pk(acc[4 * g], acc[4 * g + 1]), hi2 = gelu_fast_pk(acc[4 * g + 2], acc[4 * g + 3]);
          h4 = (f16x4){lo2[0], lo2[1], hi2[0], hi2[1]};
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) h4[e] = (f16)acc[4 * g + e];
        }
        if (32 * j + 8 * g + 4 * lh < N) *reinterpret_cast<f16x4*>(o + 8 * g) = h4;
      }
    }
  };

  auto stats_tail = [&]() {                                    // after the last chunk's epilogue
    if constexpr (RES) {
      if (ex.stats_out) {
        const float ss = st_s + __shfl_xor(st_s, 32), qq = st_q + __shfl_xor(st_q, 32);
        const float dm = ss / (float)N;
        const float var = fmaxf(qq / (float)N - dm * dm, 0.f);
        if (lh == 0) *reinterpret_cast<float2*>(ex.stats_out + 2 * row) = make_float2(st_shift + dm, 1.0f / sqrtf(var + ex.stats_eps));
      }
    }
  };
  constexpr int PF = K >= 576 ? 6 : 8;          // ring depth (K = 576: the 148 Xn registers leave less room)
  // The K/16 + 1 MFMAs of chunk j.  A-fragment ring: PF ds_read_b128 stay in flight ahead of the MFMA that consumes them.  The reads and
  // their COUNTED waits are inline asm: left to hipcc the same source becomes read -> lgkmcnt(0) -> MFMA (every MFMA then waits a full LDS
  // round trip, and the matrix pipe idles two thirds of the time).  LDS returns data in issue order, so before MFMA f at most
  // min(PF - 1, KS1 - 1 - f) younger reads may still be outstanding; nothing else of the wave may touch LDS inside the sequence, and no
  // run-time branch may sit between a read and its wait (hipcc may copy values that live across a block boundary, in-flight or not).
  // `head()` runs after the ring's first reads are issued, `mid()` after MFMA KS1 / 2.
  auto mfma_seq = [&](int j, auto&& head, auto&& mid) -> f32x16 {
    const char* const buf = smem + (j % SLOTS) * CHB + lane * 16;
    u32x4 ring[PF];
    const unsigned lbase = (unsigned)(size_t)((const __attribute__((address_space(3))) char*)buf);
#pragma unroll
    for (int f = 0; f < PF; ++f) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[f]) : "v"(lbase), "i"(f * 1024));
    head();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int f = 0; f < KS1; ++f) {
      const int young = (KS1 - 1 - f) < (PF - 1) ? (KS1 - 1 - f) : (PF - 1);
      switch (young) {                                         // (f is a compile-time constant after unrolling)
        case 0: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ring[f % PF])); break;
        case 1: asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(ring[f % PF])); break;
        case 2: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ring[f % PF])); break;
        case 3: asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(ring[f % PF])); break;
        case 4: asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ring[f % PF])); break;
        case 5: asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(ring[f % PF])); break;
        case 6: asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(ring[f % PF])); break;
        default: asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(ring[f % PF])); break;
      }
      const f16x8 a = __builtin_bit_cast(f16x8, ring[f % PF]);
      acc = CVMI_MFMA_32X32X16(a, __builtin_bit_cast(f16x8, xn[f]), acc, 0, 0, 0);
      if (f + PF < KS1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[f % PF]) : "v"(lbase), "i"((f + PF) * 1024));
      if (f == KS1 / 2) mid();
    }
    return acc;
  };
  auto nothing = [] {};
  unsigned long long seg[6] = {0, 0, 0, 0, 0, 0};
  auto stamp = [&]() -> unsigned long long {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
  };
  const unsigned long long t_begin = STAMP ? stamp() : 0ull;

  if constexpr (PP) {
    // ---- ping-pong schedule ------------------------------------------------------------------------------------------------------
    // The two waves of a SIMD (w and w + 4) share its matrix pipe and its VALU issue.  Under one barrier per chunk both run the same
    // program in phase -- epilogue beside epilogue, MFMAs beside MFMAs -- and a stamped build showed what that costs: per chunk 1325
    // cycles in the 37 MFMAs, 1254 in epilogue + prefetch issue, and 1827 + 655 waiting (the partner's MFMAs): 5061 cycles for 2 x 1184
    // cycles of matrix work per SIMD.  Here every chunk interval has TWO barriers and the halves run half an interval apart:
    //     waves 0-3:  b1 | MFMAs(j)                | b2 | prefetch, epilogue(j)     |
    //     waves 4-7:  b1 | prefetch, epilogue(j-1) | b2 | MFMAs(j)                  |
    // so a SIMD always holds one wave in its matrix phase beside one in its VALU / memory phase, and the accumulator of a chunk is
    // consumed by the phase right after it (no copy).  Ring invariants: chunk c is written to slot c % SLOTS after b1 of interval
    // c - SLOTS + 1 -- the last reads of chunk c - SLOTS (trailing half, second phase of interval c - SLOTS) ended before that barrier
    // -- and every wave waits for its own pieces (vmcnt(0), explicit: hipcc puts no wait in front of a barrier for LDS-DMA writes) at
    // the end of its NEXT matrix phase, at least one barrier before b1 of interval c.  That wait also covers the wave's epilogue stores
    // and residual loads, all issued a full phase earlier.
    auto bar = [] {
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // chunks 0 .. SLOTS - 2
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // STAMP: segment sums of wave 0 (leading half) -> g_tl_stamp[0 .. 11], of wave 4 (trailing half) -> [12 .. 23]
    auto tick = [&](int k, unsigned long long& t) {
      if constexpr (STAMP) { const unsigned long long n = stamp(); seg[k] += n - t; t = n; }
    };
    unsigned long long t = t_begin;
    if (wv < TL_NW / 2) {
#pragma unroll 1
      for (int j = 0; j < nch; ++j) {
        bar();
        tick(0, t);
        if constexpr (RES) res_load(j);
        acc = mfma_seq(j, nothing, nothing);
        if constexpr (STAMP) { asm volatile("" : "+v"(acc)); tick(1, t); }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(acc) :: "memory");
        tick(2, t);
        bar();
        tick(3, t);
        if (j + SLOTS - 1 < nch) issue_chunk(j + SLOTS - 1);
        tick(5, t);
        epilogue(acc, j);
        tick(4, t);
      }
    } else {
#pragma unroll 1
      for (int j = 0; j < nch; ++j) {
        bar();
        tick(0, t);
        if (j + SLOTS - 1 < nch) issue_chunk(j + SLOTS - 1);
        tick(5, t);
        if (j > 0) epilogue(acc, j - 1);
        if constexpr (RES) res_load(j);
        tick(1, t);
        bar();
        tick(2, t);
        acc = mfma_seq(j, nothing, nothing);
        if constexpr (STAMP) { asm volatile("" : "+v"(acc)); tick(3, t); }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(acc) :: "memory");
        tick(4, t);
      }
      epilogue(acc, nch - 1);
    }
    stats_tail();
    if constexpr (STAMP) {
      if (blockIdx.x == 0 && (tid == 0 || tid == 256)) {
        unsigned long long* g = g_tl_stamp + (tid ? 12 : 0);
        for (int k = 0; k < 6; ++k) g[k] += seg[k];
        g[6] += stamp() - t_begin; g[7] += (unsigned long long)nch; g[8] += 1;
      }
    }
    return;
  }

  f32x16 prev;
#pragma unroll
  for (int r = 0; r < 16; ++r) prev[r] = 0.f;
#pragma unroll 1
  for (int j = 0; j < nch; ++j) {
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    if constexpr (STAMP) t0 = stamp();
    // Every wave waits for its OWN DMA pieces (explicitly: hipcc puts no vmcnt wait in front of a barrier for LDS-DMA writes),
    // then the barrier publishes chunk j and frees slot (j - 1) % SLOTS.  vmcnt(0) also covers the wave's own stores, which is
    // why the epilogue of chunk j - 1 is issued AFTER this barrier: its stores then have a whole chunk of MFMAs to complete in.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (STAMP) t1 = stamp();
    __syncthreads();
    if constexpr (STAMP) t2 = stamp();
    if constexpr (RES) { if (j > 0) res_load(j - 1); }
    if constexpr (!RES && TSTORE) { if (j > 0) epilogue(prev, j - 1); }      // (its LDS round trip ends before the ring's counted waits begin)
    if (j + SLOTS - 1 < nch) issue_chunk(j + SLOTS - 1);
    if constexpr (STAMP) t3 = stamp();
    prev = mfma_seq(
        j, [&] { if constexpr (!RES && !TSTORE) { if (j > 0) epilogue(prev, j - 1); } },
        [&] { if constexpr (RES) { if (j > 0) epilogue(prev, j - 1); } });
    if constexpr (STAMP) {
      asm volatile("" : "+v"(prev));
      const unsigned long long t4 = stamp();
      seg[0] += t1 - t0; seg[1] += t2 - t1; seg[2] += t3 - t2; seg[3] += t4 - t3;
    }
  }
  if constexpr (STAMP) {
    const unsigned long long t_end = stamp();
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      g_tl_stamp[0] += seg[0]; g_tl_stamp[1] += seg[1]; g_tl_stamp[2] += seg[2]; g_tl_stamp[3] += seg[3];
      g_tl_stamp[4] += t_end - t_begin; g_tl_stamp[5] += (unsigned long long)nch; g_tl_stamp[6] += 1;
    }
  }
  if constexpr (RES) res_load(nch - 1);
  epilogue(prev, nch - 1);
  stats_tail();
}

template <int K, int LN, bool RES, bool GELU, bool TSTORE, bool STAMP = false, bool PP = false, bool POOL = false>
int launch_tl1(const void* in, int in_ld, const float* gamma, const float* beta, float eps, const void* wp, void* out, int out_ld, long long rows,
              int N, hipStream_t s, const TlExtra& ex) {
  using Cfg = TlCfg<K>;
  static hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&tok_linear_kernel<K, LN, RES, GELU, TSTORE, STAMP, PP, POOL>), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
  CVMI_HIP(attr);
  cvmi_note_kernel("tok_linear_kernel<%d, %d, %s, %s, %s, %s, %s, %s>", K, LN, CVMI_BOOLNAME(RES), CVMI_BOOLNAME(GELU), CVMI_BOOLNAME(TSTORE), CVMI_BOOLNAME(STAMP), CVMI_BOOLNAME(PP), CVMI_BOOLNAME(POOL));
  hipLaunchKernelGGL((tok_linear_kernel<K, LN, RES, GELU, TSTORE, STAMP, PP, POOL>), dim3((unsigned)(rows / 256)), dim3(TL_NW * 64), Cfg::LDS, s, in, in_ld, gamma, beta, eps,
                     (const char*)wp, out, out_ld, rows, N, ex);
  CVMI_LAUNCH_CHECK();
  return 0;
}

template <int K, int LN, bool RES, bool GELU>
int launch_tl(const void* in, int in_ld, const float* gamma, const float* beta, float eps, const void* wp, void* out, int out_ld, long long rows,
              int N, hipStream_t s, const TlExtra& ex) {
  static const int pp = getenv("CVMI_TOKLIN_PP") ? atoi(getenv("CVMI_TOKLIN_PP")) : 1;                 // 0 = one barrier per chunk (A/B measurements)
  static const int ts = getenv("CVMI_TOKLIN_TSTORE") ? atoi(getenv("CVMI_TOKLIN_TSTORE")) : 1;         // 0 = direct stores (A/B measurements)
  const bool tstore = !RES && ts && N % 8 == 0 && out_ld % 8 == 0;
#ifndef CVMI_OPERAND_BF16
  static const int st = getenv("CVMI_TOKLIN_STAMP") ? atoi(getenv("CVMI_TOKLIN_STAMP")) : 0;           // diagnostic build, never for timing
  if constexpr (K == 576 && LN == 1 && !RES) {
    if (st && tstore && (st == 1 || (st == 2) == GELU)) {       // 2: only the GELU launches (fc1), 3: only the plain ones (qkv)
      if (pp) return launch_tl1<K, LN, RES, GELU, true, true, true>(in, in_ld, gamma, beta, eps, wp, out, out_ld, rows, N, s, ex);
      return launch_tl1<K, LN, RES, GELU, true, true>(in, in_ld, gamma, beta, eps, wp, out, out_ld, rows, N, s, ex);
    }
  }
#endif
  if constexpr (!RES) {
    if (tstore) {
      if (pp) return launch_tl1<K, LN, RES, GELU, true, false, true>(in, in_ld, gamma, beta, eps, wp, out, out_ld, rows, N, s, ex);
      return launch_tl1<K, LN, RES, GELU, true>(in, in_ld, gamma, beta, eps, wp, out, out_ld, rows, N, s, ex);
    }
  }
  // The one-barrier schedule runs the RES epilogue in the MIDDLE of the MFMA sequence, where hipcc re-loads a kernel argument (s_load:
  // the same counter as the ring's counted lgkmcnt waits -- tools/check_ring_asm.py found it): the residual form always takes the ping-pong
  // schedule, whose epilogue sits outside the ring window.
  if constexpr (RES) {
    return launch_tl1<K, LN, RES, GELU, false, false, true>(in, in_ld, gamma, beta, eps, wp, out, out_ld, rows, N, s, ex);
  } else {
    if (pp) return launch_tl1<K, LN, RES, GELU, false, false, true>(in, in_ld, gamma, beta, eps, wp, out, out_ld, rows, N, s, ex);
    return launch_tl1<K, LN, RES, GELU, false>(in, in_ld, gamma, beta, eps, wp, out, out_ld, rows, N, s, ex);
  }
}

template <int K>
int dispatch_tl(int ln, bool res, int act, const void* in, int in_ld, const float* gamma, const float* beta, float eps, const void* wp, void* out,
                int out_ld, long long rows, int N, hipStream_t s, const TlExtra& ex) {
  const bool gelu = act == CVMI_ACT_GELU;
  if (ln == 2) {
    if constexpr (K <= 288) return launch_tl<K, 2, false, false>(in, in_ld, gamma, beta, eps, wp, out, out_ld, rows, N, s, ex);
    CVMI_FAIL("tok_linear: plain f32 input is built for K = 144 and 288");
  }
  if (res) {
    if (ln) return launch_tl<K, 1, true, false>(in, in_ld, gamma, beta, eps, wp, out, out_ld, rows, N, s, ex);
    return launch_tl<K, 0, true, false>(in, in_ld, gamma, beta, eps, wp, out, out_ld, rows, N, s, ex);
  }
  if (ln) {
    if (gelu) return launch_tl<K, 1, false, true>(in, in_ld, gamma, beta, eps, wp, out, out_ld, rows, N, s, ex);
    return launch_tl<K, 1, false, false>(in, in_ld, gamma, beta, eps, wp, out, out_ld, rows, N, s, ex);
  }
  if (gelu) return launch_tl<K, 0, false, true>(in, in_ld, gamma, beta, eps, wp, out, out_ld, rows, N, s, ex);
  return launch_tl<K, 0, false, false>(in, in_ld, gamma, beta, eps, wp, out, out_ld, rows, N, s, ex);
}

}  // namespace

// K = 576 is served by tok_linear16.hip (16x16x32 MFMA shape, its own packed-weight format) unless CVMI_TOKLIN_M16=0 (A/B runs)
int CVMI_ENTRY(cvmi_tok_linear16_launch)(int K, int ln, bool res, bool gelu, const void* in, int in_ld, const float* gamma, const float* beta, float eps,
                                         const void* wp, void* out, int out_ld, long long rows, int N, int pool_w, int pool_hw2, const float* stats_in,
                                         int stats_parts, float* stats_out, float stats_eps, hipStream_t s);
static int tl_format(int K) {
  static const int m16 = getenv("CVMI_TOKLIN_M16") ? atoi(getenv("CVMI_TOKLIN_M16")) : 1;
  return (m16 && K == 576) ? 16 : 32;
}

#ifndef CVMI_OPERAND_BF16
extern "C" int cvmi_tok_linear_format(int K) { return tl_format(K); }
int CVMI_ENTRY(cvmi_tok_linear16_splits)(long long rows, int N);
extern "C" int cvmi_tok_linear_stats_parts(long long rows, int K, int N) {
  if (rows <= 0 || rows % 256 || N <= 0 || tl_format(K) != 16) return 0;
  const int ns = CVMI_ENTRY(cvmi_tok_linear16_splits)(rows, N);
  return ns > 1 ? ns : 0;
}

// diagnostic: read and clear the segment sums of the CVMI_TOKLIN_STAMP build.  Ping-pong schedule: [0 .. 8] = wave 0 {b1 wait, MFMAs, vmcnt wait,
// b2 wait, epilogue, prefetch issue, total, chunks, launches}, [12 .. 20] = wave 4 {b1 wait, epilogue, b2 wait, MFMAs, vmcnt wait, prefetch issue,
// total, chunks, launches}.  One-barrier schedule (CVMI_TOKLIN_PP=0): [0 .. 6] = {vmcnt wait, barrier, issue, MFMAs, total, chunks, launches}.
extern "C" int cvmi_debug_stamps(unsigned long long* out24) {
  CVMI_HIP(hipMemcpyFromSymbol(out24, HIP_SYMBOL(g_tl_stamp), 24 * sizeof(unsigned long long)));
  unsigned long long z[24] = {};
  CVMI_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_tl_stamp), z, sizeof(z)));
  return 0;
}

extern "C" int cvmi_tok_linear_supported(int K) { return K == 144 || K == 288 || K == 576; }

extern "C" size_t cvmi_tok_linear_packed_bytes(int K, int N) {
  if (!cvmi_tok_linear_supported(K) || N <= 0) return 0;
  if (tl_format(K) == 16) return (size_t)((N + 31) / 32) * (size_t)(K / 16 + 1) * 1024;     // 2 K/32 weight fragments + the bias piece per chunk
  return (size_t)(((N + 31) / 32 + 1) / 2 * 2) * (size_t)(K / 16 + 1) * 1024;        // chunk count padded to even
}
extern "C" int cvmi_tok_linear_stats_bf16(const void* in, int in_ld, int in_f32_layernorm, const float* gamma, const float* beta, float eps,
                                          const void* w_packed, void* out, int out_ld, int out_f32_residual, long long rows, int K, int N, int act,
                                          int dtype, const float* ln_stats_in, int ln_stats_in_parts, float* ln_stats_out, float ln_stats_eps, cvmi_stream_t stream_);
#endif

// cvmi_tok_linear with LayerNorm statistics handed from the launch that WRITES the residual stream to the launch that normalises it:
//   ln_stats_out (out_f32_residual = 1): float2 per row = (mean, 1 / sqrt(var + ln_stats_eps)) over the N updated values of the row
//   ln_stats_in  (in_f32_layernorm = 1): the same pair per row; the prologue then reads every row once instead of twice.
//                ln_stats_in_parts = P > 0: instead P raw (sum, sum of squares) pairs per row, as cvmi_conv_desc.row_stats writes them
// (x = x + proj(attn) followed by mlp.layers[0](norm2(x)) in sam2 hieradet MultiScaleBlock; behind /root/reference/src/sam2_infer.py:226).
extern "C" int CVMI_ENTRY(cvmi_tok_linear_stats)(const void* in, int in_ld, int in_f32_layernorm, const float* gamma, const float* beta, float eps,
                                                 const void* w_packed, void* out, int out_ld, int out_f32_residual, long long rows, int K, int N, int act,
                                                 int dtype, const float* ln_stats_in, int ln_stats_in_parts, float* ln_stats_out, float ln_stats_eps, cvmi_stream_t stream_) {
#ifndef CVMI_OPERAND_BF16
  if (dtype == CVMI_BF16)
    return cvmi_tok_linear_stats_bf16(in, in_ld, in_f32_layernorm, gamma, beta, eps, w_packed, out, out_ld, out_f32_residual, rows, K, N, act, dtype,
                                      ln_stats_in, ln_stats_in_parts, ln_stats_out, ln_stats_eps, stream_);
#endif
  CVMI_CHECK(dtype == CVMI_T16, "tok_linear: dtype must be CVMI_F16 or CVMI_BF16");
  CVMI_CHECK(in && w_packed && out && rows > 0 && rows % 256 == 0 && N > 0, "tok_linear: bad arguments (rows must be a multiple of 256)");
  CVMI_CHECK(K == 144 || K == 288 || K == 576, "tok_linear: K=%d is not built (144, 288, 576)", K);
  CVMI_CHECK(in_f32_layernorm != 1 || (gamma && beta), "tok_linear: LayerNorm input needs gamma / beta");
  CVMI_CHECK(in_f32_layernorm >= 0 && in_f32_layernorm <= 2 && (in_f32_layernorm != 2 || (!out_f32_residual && act == CVMI_ACT_NONE)),
             "tok_linear: in_f32_layernorm must be 0, 1 or 2 (2: 16-bit output, no activation)");
  CVMI_CHECK(act == CVMI_ACT_NONE || (act == CVMI_ACT_GELU && !out_f32_residual), "tok_linear: act must be NONE, or GELU with 16-bit output");
  CVMI_CHECK(in_ld >= K && in_ld % (in_f32_layernorm ? 4 : 8) == 0 && out_ld >= N && out_ld % 4 == 0 && N % 4 == 0 &&
                 (((uintptr_t)in | (uintptr_t)w_packed | (uintptr_t)out) & 15) == 0 &&
                 (in_f32_layernorm != 1 || (((uintptr_t)gamma | (uintptr_t)beta) & 15) == 0),
             "tok_linear: pointers / ld not aligned (in_ld=%d out_ld=%d N=%d)", in_ld, out_ld, N);
  CVMI_CHECK((!ln_stats_in || in_f32_layernorm == 1) && (!ln_stats_out || out_f32_residual) && (((uintptr_t)ln_stats_in | (uintptr_t)ln_stats_out) & 7) == 0,
             "tok_linear: ln_stats_in needs the LayerNorm input form, ln_stats_out the residual output form (8-byte aligned)");
  CVMI_CHECK(ln_stats_in_parts >= 0 && ln_stats_in_parts <= 64, "tok_linear: ln_stats_in_parts out of range");
  hipStream_t s = (hipStream_t)stream_;
  const int ln = in_f32_layernorm;
  const bool res = out_f32_residual != 0;
  CVMI_CHECK(ln != 2 || tl_format(K) != 16, "tok_linear: plain f32 input (in_f32_layernorm = 2) is built for K = 144 and 288");
  if (tl_format(K) == 16)
    return CVMI_ENTRY(cvmi_tok_linear16_launch)(K, ln, res, act == CVMI_ACT_GELU, in, in_ld, gamma, beta, eps, w_packed, out, out_ld, rows, N, 0, 0, ln_stats_in,
                                                ln_stats_in_parts, ln_stats_out, ln_stats_eps, s);
  const TlExtra ex{0, 0, ln_stats_in, ln_stats_out, ln_stats_eps, ln_stats_in_parts};
  switch (K) {
    case 144: return dispatch_tl<144>(ln, res, act, in, in_ld, gamma, beta, eps, w_packed, out, out_ld, rows, N, s, ex);
    case 288: return dispatch_tl<288>(ln, res, act, in, in_ld, gamma, beta, eps, w_packed, out, out_ld, rows, N, s, ex);
    default: return dispatch_tl<576>(ln, res, act, in, in_ld, gamma, beta, eps, w_packed, out, out_ld, rows, N, s, ex);
  }
}

extern "C" int CVMI_ENTRY(cvmi_tok_linear)(const void* in, int in_ld, int in_f32_layernorm, const float* gamma, const float* beta, float eps,
                                           const void* w_packed, void* out, int out_ld, int out_f32_residual, long long rows, int K, int N, int act,
                                           int dtype, cvmi_stream_t stream_) {
  return CVMI_ENTRY(cvmi_tok_linear_stats)(in, in_ld, in_f32_layernorm, gamma, beta, eps, w_packed, out, out_ld, out_f32_residual, rows, K, N, act, dtype,
                                           nullptr, 0, nullptr, 0.f, stream_);
}

#ifndef CVMI_OPERAND_BF16
extern "C" int cvmi_tok_linear_pool_stats_bf16(const void* in, int in_ld, const float* gamma, const float* beta, float eps, const void* w_packed, void* out,
                                               int out_ld, int B, int H, int W, int K, int N, int dtype, const float* ln_stats_in, cvmi_stream_t stream_);
#endif

// out[b, y, x, :] = max over the 2 x 2 token block of ( LayerNorm(in[b, 2y + dy, 2x + dx, :]) W^T + bias ): the shortcut path of a Hiera
// q-pooling block, `do_pool(self.proj(x_norm))` (sam2 hieradet MultiScaleBlock.forward, behind /root/reference/src/sam2_infer.py:226), in one
// launch -- the full-resolution f32 projection (1.2 GB at the stage 1 -> 2 transition, B = 16) is neither written nor read back.
// ln_stats_in (may be NULL): per source row (mean, rstd) as written by cvmi_hiera_mlp_stats / cvmi_tok_linear_stats -- one pass over the rows.
extern "C" int CVMI_ENTRY(cvmi_tok_linear_pool_stats)(const void* in, int in_ld, const float* gamma, const float* beta, float eps, const void* w_packed,
                                                      void* out, int out_ld, int B, int H, int W, int K, int N, int dtype, const float* ln_stats_in,
                                                      cvmi_stream_t stream_) {
#ifndef CVMI_OPERAND_BF16
  if (dtype == CVMI_BF16)
    return cvmi_tok_linear_pool_stats_bf16(in, in_ld, gamma, beta, eps, w_packed, out, out_ld, B, H, W, K, N, dtype, ln_stats_in, stream_);
#endif
  CVMI_CHECK(dtype == CVMI_T16, "tok_linear_pool: dtype must be CVMI_F16 or CVMI_BF16");
  const long long rows = (long long)B * H * W;
  CVMI_CHECK(in && w_packed && out && gamma && beta && B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && rows % 256 == 0 && N > 0,
             "tok_linear_pool: bad arguments (H, W even; B*H*W a multiple of 256)");
  CVMI_CHECK(K == 144 || K == 288 || K == 576, "tok_linear_pool: K=%d is not built (144, 288, 576)", K);
  CVMI_CHECK(in_ld >= K && in_ld % 4 == 0 && out_ld >= N && out_ld % 4 == 0 && N % 4 == 0 &&
                 (((uintptr_t)in | (uintptr_t)w_packed | (uintptr_t)out | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0,
             "tok_linear_pool: pointers / ld not aligned (in_ld=%d out_ld=%d N=%d)", in_ld, out_ld, N);
  hipStream_t s = (hipStream_t)stream_;
  const int hw2 = (H / 2) * (W / 2);
  if (tl_format(K) == 16)
    return CVMI_ENTRY(cvmi_tok_linear16_launch)(K, 1, false, false, in, in_ld, gamma, beta, eps, w_packed, out, out_ld, rows, N, W, hw2, ln_stats_in, 0, nullptr, 0.f, s);
  switch (K) {
    case 144: return launch_tl1<144, 1, false, false, false, false, true, true>(in, in_ld, gamma, beta, eps, w_packed, out, out_ld, rows, N, s, TlExtra{W, hw2, ln_stats_in, nullptr, 0.f, 0});
    case 288: return launch_tl1<288, 1, false, false, false, false, true, true>(in, in_ld, gamma, beta, eps, w_packed, out, out_ld, rows, N, s, TlExtra{W, hw2, ln_stats_in, nullptr, 0.f, 0});
    default: return launch_tl1<576, 1, false, false, false, false, true, true>(in, in_ld, gamma, beta, eps, w_packed, out, out_ld, rows, N, s, TlExtra{W, hw2, ln_stats_in, nullptr, 0.f, 0});
  }
}

extern "C" int CVMI_ENTRY(cvmi_tok_linear_pool)(const void* in, int in_ld, const float* gamma, const float* beta, float eps, const void* w_packed,
                                                void* out, int out_ld, int B, int H, int W, int K, int N, int dtype, cvmi_stream_t stream_) {
  return CVMI_ENTRY(cvmi_tok_linear_pool_stats)(in, in_ld, gamma, beta, eps, w_packed, out, out_ld, B, H, W, K, N, dtype, nullptr, stream_);
}
