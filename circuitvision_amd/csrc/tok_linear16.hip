// Token-stationary linear layer on the 16x16x32 MFMA shape (K = 576: Hiera stage 3 -- qkv, proj, fc1 -- 15 of the 47 ms of a SAM 2.1-L
// B = 16 pass).  Same contract, same schedule as tok_linear_kernel (tok_linear.hip: a wave keeps its 32 rows as MFMA B fragments for the
// whole launch, the weights stream L2 -> LDS by global_load_lds in fragment order through a ring, two waves per SIMD run half a chunk
// interval apart), different matrix instruction: v_mfma_f32_16x16x32 has the cycles per flop of 32x32x16, but under the board's power
// limit the chip holds a ~10 % higher clock on it (MI355X_MICROARCH.md, DVFS give-back item 7; measured here on gemm256x192_kernel:
// 233 -> 213 us).  What changes with the shape:
//   * B fragments: lane (c16 = lane & 15, g = lane >> 4) holds in[token][32 s + 8 g .. + 7] for the wave's TWO 16-token groups
//     (tokens c16 and 16 + c16): K / 4 registers, as before.
//   * A fragments (1 KiB, lane-linear): F(j, s, hh)[lane (r16, g)][e] = W[32 j + 16 hh + r16][32 s + 8 g + e]; one ds_read_b128 feeds two
//     MFMAs (the two token groups): the LDS read rate per flop is unchanged.
//   * the bias no longer rides on an extra k-step as a hi + lo 16-bit pair: the chunk's 32 f32 bias values travel as a last 1-KiB piece of
//     the chunk (first 128 bytes) and INITIALISE the accumulators (lane (., g) holds channels 16 hh + 4 g .. + 3 of both token groups) --
//     exact f32 bias, four MFMAs fewer per chunk.
//   * C / D: lane (c16, g) holds channels 4 g .. 4 g + 3 (registers) of token c16: residual-stream stores are 64 contiguous bytes per row
//     (four lanes) instead of 16-byte pieces of 32 rows.
#include "common.hpp"
#include <stdlib.h>

// CVMI_TL16_DIAG (compile time, timing-only builds -- tools/r3_call32.sh links them into alternative libraries; results are wrong): bit 0 = no
// weight DMA behind the first chunks, bit 1 = no MFMAs (the ring reads stay), bit 2 = no epilogue (no GELU, no stores).  0 in the shipped library.
#ifndef CVMI_TL16_PRIO
#define CVMI_TL16_PRIO 0               /* A/B builds: 1 = static s_setprio 1 for the younger half (waves 4-7); 2 = s_setprio 1 in every epilogue phase */
#endif
#ifndef CVMI_TL16_DIAG
#define CVMI_TL16_DIAG 0
#endif

namespace {

constexpr int TL_NW = 8;

template <int K> struct Tl16Cfg {
  static_assert(K % 32 == 0, "k-steps of 32");
  static constexpr int KS = K / 32;
  static constexpr int NF = 2 * KS;                            // weight fragments per 32-channel chunk, index 2 s + hh
  static constexpr int CHB = (NF + 1) * 1024;                  // + the bias piece
  static constexpr int SLOTS = 3;
  static constexpr int STG = 32 * 80;                          // per-wave transposition stage: 32 rows x (64 + 16) bytes
  static constexpr int GB = SLOTS * CHB + TL_NW * STG;        // LN = 1: gamma, beta (2 K floats) parked for the row prologue
  static constexpr int LDS = GB + 2 * K * 4;
};

struct Tl16Extra {
  int pool_w, pool_hw2;
  const float* stats_in;
  float* stats_out;
  float stats_eps;
  int stats_parts;
};

struct Acc16 { f32x4 v[2][2]; };                               // [16-channel half hh][16-token group tg]

template <int K, int LN, bool RES, bool GELU, bool POOL>
__global__ __launch_bounds__(TL_NW * 64, 2) void tok_linear16_kernel(const void* __restrict__ in, int in_ld, const float* __restrict__ gamma,
                                                                     const float* __restrict__ beta, float eps, const char* __restrict__ wp,
                                                                     void* __restrict__ out, int out_ld, long long rows, int N, const Tl16Extra ex) {
  using Cfg = Tl16Cfg<K>;
  constexpr int KS = Cfg::KS, NF = Cfg::NF, CHB = Cfg::CHB, SLOTS = Cfg::SLOTS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int c16 = lane & 15, g = lane >> 4;
  const long long wrow0 = ((long long)blockIdx.x * TL_NW + wv) * 32;
  long long row[2], prow[2] = {0, 0};
#pragma unroll
  for (int tg = 0; tg < 2; ++tg) {
    row[tg] = wrow0 + 16 * tg + c16;
    if constexpr (POOL) {                                      // lane quad = the four tokens of a 2 x 2 block of the [B, H, W] grid (tok_linear.hip)
      prow[tg] = row[tg] >> 2;
      const long long b = prow[tg] / ex.pool_hw2;
      const int r = (int)(prow[tg] - b * ex.pool_hw2), w2 = ex.pool_w >> 1;
      const int py = r / w2, px = r - py * w2;
      row[tg] = b * 4 * ex.pool_hw2 + (long long)(2 * py + ((c16 >> 1) & 1)) * ex.pool_w + 2 * px + (c16 & 1);
    }
  }
  // Output-channel chunks of this workgroup: gridDim.y workgroups share a 256-row block when the launch has fewer row blocks than the chip
  // has CUs (the per-rank batch of an 8-GPU job: 8 images = 128 row blocks), each walks its own range [j0, j1) of the 32-channel chunks.  The
  // siblings (x, y) have linear ids x + gridDim.x * y: the host only splits when gridDim.x is a multiple of 8, so they share an XCD's L2 for
  // the rows they both read.
  const int nch_all = (N + 31) / 32, cps = (nch_all + (int)gridDim.y - 1) / (int)gridDim.y;
  const int j0 = (int)blockIdx.y * cps, nch = j0 + cps < nch_all ? j0 + cps : nch_all;

  auto issue_chunk = [&](int j) {
    const char* src = wp + (size_t)j * CHB + lane * 16;
    char* dst = smem + (j % SLOTS) * CHB;
#pragma unroll
    for (int f = 0; f < (NF + 1 + TL_NW - 1) / TL_NW; ++f) {
      const int fi = f * TL_NW + wv;
      if (fi < NF + 1)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)fi * 1024),
                                         (__attribute__((address_space(3))) void*)(dst + fi * 1024), 16, 0, 0);
    }
  };
#pragma unroll
  for (int j = j0; j < j0 + SLOTS - 1; ++j)
    if (j < nch) issue_chunk(j);

  // ---- B fragments
  u32x4 xn[2][KS];
  if constexpr (LN == 1) {
    const float* xr[2] = {reinterpret_cast<const float*>(in) + row[0] * (long long)in_ld, reinterpret_cast<const float*>(in) + row[1] * (long long)in_ld};
    float mean[2], rstd[2];
#pragma unroll
    for (int tg = 0; tg < 2; ++tg) {
      if (ex.stats_in && ex.stats_parts == 0) {
        const float2 st = *reinterpret_cast<const float2*>(ex.stats_in + 2 * row[tg]);
        mean[tg] = st.x; rstd[tg] = st.y;
      } else if (ex.stats_in) {                                // per column slice (mean, sum of squared deviations) of a tiled GEMM's epilogue
        float ms = 0.f, m2 = 0.f;                              // combined as Chan et al.: equal slice sizes n = K / P, fixed order
        for (int t = 0; t < ex.stats_parts; ++t) {
          const float2 st = *reinterpret_cast<const float2*>(ex.stats_in + (row[tg] * ex.stats_parts + t) * 2);
          ms += st.x; m2 += st.y;
        }
        const float inv_p = 1.0f / (float)ex.stats_parts;
        mean[tg] = ms * inv_p;
        // sum_t M2_t + n sum_t (mean_t - mean)^2, the second term as n (sum mean_t^2 - P mean^2): slice means differ by O(sigma / sqrt(n)) only
        // when the row's offset is common to all slices, and then sum mean_t^2 - P mean^2 is a difference of nearly equal numbers of size
        // mean^2 -- so it is formed from the deviations instead
        float dev = 0.f;
        for (int t = 0; t < ex.stats_parts; ++t) {
          const float d = ex.stats_in[(row[tg] * ex.stats_parts + t) * 2] - mean[tg];
          dev = fmaf(d, d, dev);
        }
        const float var = (m2 + ((float)K * inv_p) * dev) / (float)K;
        rstd[tg] = 1.0f / sqrtf(fmaxf(var, 0.f) + eps);
      } else {                                                 // own statistics: shifted single pass over this lane's quarter of the row, 4 lanes per row
        const float x0 = xr[tg][0];
        float s = 0.f, q = 0.f;
#pragma unroll 3
        for (int k = 0; k < KS; ++k) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(xr[tg] + 32 * k + 8 * g), b = *reinterpret_cast<const f32x4*>(xr[tg] + 32 * k + 8 * g + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float da = a[e] - x0, db = b[e] - x0;
            s += da + db;
            q = fmaf(da, da, fmaf(db, db, q));
          }
        }
        s += __shfl_xor(s, 16); q += __shfl_xor(q, 16);
        s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);
        const float dm = s / (float)K;
        mean[tg] = x0 + dm;
        rstd[tg] = 1.0f / sqrtf(fmaxf(q / (float)K - dm * dm, 0.f) + eps);
      }
    }
    // The row stream (151 MB of f32 per Hiera-L launch, from HBM) as a software pipeline: PFX k-steps of this lane's two rows in flight, the
    // next one issued as soon as one is consumed.  gamma / beta come from LDS (parked below: their loads would otherwise sit in the same
    // in-order vmcnt queue, younger than the rows they are needed with).  The grouped form this replaces (two k-steps loaded, waited for and
    // converted at a time) exposed the HBM latency nine times per workgroup: ~30 of fc1's 178 us with nothing else on the CU (timing-only
    // builds, profiles/r03_ab_runs.md) -- the launch is ONE round of 256 workgroups, so no other workgroup hides it.
    constexpr int PFX = 4;
    f32x4 xa[PFX][2][2];
#pragma unroll
    for (int k = 0; k < PFX; ++k)
#pragma unroll
      for (int tg = 0; tg < 2; ++tg) {
        xa[k][tg][0] = *reinterpret_cast<const f32x4*>(xr[tg] + 32 * k + 8 * g);
        xa[k][tg][1] = *reinterpret_cast<const f32x4*>(xr[tg] + 32 * k + 8 * g + 4);
      }
    {                                                          // (behind the first row loads: the barrier's fence waits for every load in flight)
      float* const gb = reinterpret_cast<float*>(smem + Cfg::GB);
      for (int i = tid; i < K; i += TL_NW * 64) { gb[i] = gamma[i]; gb[K + i] = beta[i]; }
      __syncthreads();
    }
    const float* const gl = reinterpret_cast<const float*>(smem + Cfg::GB) + 8 * g;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(gl + 32 * k), g1 = *reinterpret_cast<const f32x4*>(gl + 32 * k + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(gl + K + 32 * k), b1 = *reinterpret_cast<const f32x4*>(gl + K + 32 * k + 4);
#pragma unroll
      for (int tg = 0; tg < 2; ++tg) {
        const f32x4 a = xa[k % PFX][tg][0], b = xa[k % PFX][tg][1];
        f16x8 h;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          h[e] = (f16)((a[e] - mean[tg]) * rstd[tg] * g0[e] + b0[e]);
          h[4 + e] = (f16)((b[e] - mean[tg]) * rstd[tg] * g1[e] + b1[e]);
        }
        xn[tg][k] = __builtin_bit_cast(u32x4, h);
      }
      if (k + PFX < KS) {
#pragma unroll
        for (int tg = 0; tg < 2; ++tg) {
          xa[k % PFX][tg][0] = *reinterpret_cast<const f32x4*>(xr[tg] + 32 * (k + PFX) + 8 * g);
          xa[k % PFX][tg][1] = *reinterpret_cast<const f32x4*>(xr[tg] + 32 * (k + PFX) + 8 * g + 4);
        }
      }
      __builtin_amdgcn_sched_barrier(0);                       // (keeps hipcc from hoisting every load to the top: 210 spilled registers in a first r03 build)
    }
  } else {
#pragma unroll
    for (int tg = 0; tg < 2; ++tg) {
      const f16* xr = reinterpret_cast<const f16*>(in) + row[tg] * (long long)in_ld;
#pragma unroll
      for (int k = 0; k < KS; ++k) xn[tg][k] = *reinterpret_cast<const u32x4*>(xr + 32 * k + 8 * g);
    }
  }

  // ---- epilogue of chunk j: lane (token c16 of group tg, g), half hh -> channels 32 j + 16 hh + 4 g .. + 3
  f32x4 r4[2][2];
  auto res_load = [&](int j) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int tg = 0; tg < 2; ++tg) {
        const int ch = 32 * j + 16 * hh + 4 * g;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        r4[hh][tg] = ch < N ? *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(out) + row[tg] * (long long)out_ld + ch) : z;
      }
  };
  float st_shift[2] = {0.f, 0.f}, st_s[2] = {0.f, 0.f}, st_q[2] = {0.f, 0.f};
  char* const stage = smem + SLOTS * CHB + wv * Cfg::STG;
  auto epilogue = [&](const Acc16& acc, int j) {
    if constexpr (POOL) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int tg = 0; tg < 2; ++tg) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float a = acc.v[hh][tg][e];
            a = fmaxf(a, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a), 0xB1, 0xF, 0xF, true)));   // quad_perm [1,0,3,2]
            a = fmaxf(a, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a), 0x4E, 0xF, 0xF, true)));   // quad_perm [2,3,0,1]
            v[e] = a;
          }
          const int ch = 32 * j + 16 * hh + 4 * g;
          if ((c16 & 3) == 0 && ch < N) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + prow[tg] * (long long)out_ld + ch) = v;
        }
    } else if constexpr (RES) {
      const bool stats = ex.stats_out != nullptr;              // (uniform)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int tg = 0; tg < 2; ++tg) {
          f32x4 v = r4[hh][tg];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += acc.v[hh][tg][e];
          if (stats && j == j0 && hh == 0) st_shift[tg] = __shfl(v[0], c16);      // the row's first updated value (lane g = 0): the variance shift
          const int ch = 32 * j + 16 * hh + 4 * g;
          if (ch < N) {
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + row[tg] * (long long)out_ld + ch) = v;
            if (stats) {
#pragma unroll
              for (int e = 0; e < 4; ++e) { const float dv = v[e] - st_shift[tg]; st_s[tg] += dv; st_q[tg] = fmaf(dv, dv, st_q[tg]); }
            }
          }
        }
    } else {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int tg = 0; tg < 2; ++tg) {
          const f32x4 a = acc.v[hh][tg];
          f16x4 h4;
          if constexpr (GELU) {
            const f16x2 lo2 = gelu_fast_pk(a[0], a[1]), hi2 = gelu_fast_pk(a[2], a[3]);
            h4 = (f16x4){lo2[0], lo2[1], hi2[0], hi2[1]};
          } else {
            h4 = (f16x4){(f16)a[0], (f16)a[1], (f16)a[2], (f16)a[3]};
          }
          *reinterpret_cast<f16x4*>(stage + (16 * tg + c16) * 80 + (16 * hh + 4 * g) * 2) = h4;
        }
      // (measured r03, both no better than these 64-byte row pieces: 8-byte stores straight from the accumulator layout, +12 % on fc1; two
      //  chunks staged and stored as 128-byte pieces, +-0)
      const int sr = lane >> 2, pc = lane & 3;
      const int c0 = 32 * j + pc * 8;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(stage + (16 * i + sr) * 80 + pc * 16);
        if (c0 < N) *reinterpret_cast<u32x4*>(reinterpret_cast<f16*>(out) + (wrow0 + 16 * i + sr) * (long long)out_ld + c0) = v;
      }
    }
  };
  auto stats_tail = [&]() {
    if constexpr (RES) {
      if (ex.stats_out) {
#pragma unroll
        for (int tg = 0; tg < 2; ++tg) {
          float ss = st_s[tg], qq = st_q[tg];
          ss += __shfl_xor(ss, 16); qq += __shfl_xor(qq, 16);
          ss += __shfl_xor(ss, 32); qq += __shfl_xor(qq, 32);
          if (gridDim.y == 1) {
            const float dm = ss / (float)N;
            const float var = fmaxf(qq / (float)N - dm * dm, 0.f);
            if (g == 0) *reinterpret_cast<float2*>(ex.stats_out + 2 * row[tg]) = make_float2(st_shift[tg] + dm, 1.0f / sqrtf(var + ex.stats_eps));
          } else {                                              // split rows: this workgroup's slice as (mean, sum of squared deviations) -- the
            const float n = (float)(32 * (nch - j0));           // `ln_stats_in_parts` format (host: whole chunks, equal slices), Chan et al. in the consumer
            const float dm = ss / n;
            if (g == 0) *reinterpret_cast<float2*>(ex.stats_out + 2 * (row[tg] * (long long)gridDim.y + blockIdx.y)) = make_float2(st_shift[tg] + dm, fmaxf(qq - ss * dm, 0.f));
          }
        }
      }
    }
  };

  // ---- the 2 NF MFMAs of chunk j.  Ring of PF ds_read_b128 in flight ahead of their consumers, counted lgkmcnt waits in inline asm (left
  // to hipcc every MFMA waits a full LDS round trip).  LDS returns data in issue order: the two bias reads are issued FIRST, so the wait in
  // front of MFMA 0 covers them.  Nothing else of the wave touches LDS inside the sequence; no run-time branch between a read and its wait.
  constexpr int PF = 6;
  auto mfma_seq = [&](int j) -> Acc16 {
    const char* const buf = smem + (j % SLOTS) * CHB;
    const unsigned lbase = (unsigned)(size_t)((const __attribute__((address_space(3))) char*)(buf + lane * 16));
    const unsigned bbase = (unsigned)(size_t)((const __attribute__((address_space(3))) char*)(buf + NF * 1024 + g * 16));
    f32x4 b4[2];
    u32x4 ring[PF];
    asm volatile("ds_read_b128 %0, %1" : "=v"(b4[0]) : "v"(bbase));
    asm volatile("ds_read_b128 %0, %1 offset:64" : "=v"(b4[1]) : "v"(bbase));
#pragma unroll
    for (int f = 0; f < PF; ++f) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[f]) : "v"(lbase), "i"(f * 1024));
    Acc16 acc;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      constexpr int dummy = 0; (void)dummy;
      const int young = (NF - 1 - f) < (PF - 1) ? (NF - 1 - f) : (PF - 1);
      if (f == 0) {
        asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(ring[0]), "+v"(b4[0]), "+v"(b4[1]));          // PF - 1 younger ring reads may still fly
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
          for (int tg = 0; tg < 2; ++tg) acc.v[hh][tg] = b4[hh];
      } else {
        switch (young) {
          case 0: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ring[f % PF])); break;
          case 1: asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(ring[f % PF])); break;
          case 2: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ring[f % PF])); break;
          case 3: asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(ring[f % PF])); break;
          case 4: asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ring[f % PF])); break;
          default: asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(ring[f % PF])); break;
        }
      }
      const f16x8 a = __builtin_bit_cast(f16x8, ring[f % PF]);
      const int s = f >> 1, hh = f & 1;
      if constexpr (!(CVMI_TL16_DIAG & 2)) {
        acc.v[hh][0] = CVMI_MFMA_16X16X32(a, __builtin_bit_cast(f16x8, xn[0][s]), acc.v[hh][0], 0, 0, 0);
        acc.v[hh][1] = CVMI_MFMA_16X16X32(a, __builtin_bit_cast(f16x8, xn[1][s]), acc.v[hh][1], 0, 0, 0);
      } else {
        asm volatile("" :: "v"(a));                           // (keeps the ring read and its wait)
      }
      if (f + PF < NF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[f % PF]) : "v"(lbase), "i"((f + PF) * 1024));
    }
    return acc;
  };
  static_assert(PF == 6 && NF > PF, "the first wait is written for PF = 6");

  // ---- ping-pong schedule (tok_linear.hip): two barriers per chunk interval, the halves of the workgroup half an interval apart
  //     waves 0-3:  b1 | MFMAs(j)                | b2 | prefetch, epilogue(j)     |
  //     waves 4-7:  b1 | prefetch, epilogue(j-1) | b2 | MFMAs(j)                  |
  // Ring invariants as there: chunk c is written to slot c % SLOTS after b1 of interval c - SLOTS + 1 (the last reads of chunk c - SLOTS ended
  // before that barrier); every wave waits for its own DMA pieces (explicit vmcnt(0)) at the end of its next matrix phase, at least one barrier
  // before b1 of interval c.
  auto bar = [] {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // chunks 0 .. SLOTS - 2 (and the prologue's loads)
  Acc16 acc;
#pragma unroll
  for (int hh = 0; hh < 2; ++hh)
#pragma unroll
    for (int tg = 0; tg < 2; ++tg) acc.v[hh][tg] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (CVMI_TL16_PRIO == 1 && wv >= TL_NW / 2) __builtin_amdgcn_s_setprio(1);
  if (CVMI_TL16_PRIO == 4 && wv < TL_NW / 2) __builtin_amdgcn_s_setprio(1);
  if (wv < TL_NW / 2) {
#pragma unroll 1
    for (int j = j0; j < nch; ++j) {
      bar();
      if (CVMI_TL16_PRIO == 2) __builtin_amdgcn_s_setprio(0); else if (CVMI_TL16_PRIO == 3) __builtin_amdgcn_s_setprio(1);
      if constexpr (RES) res_load(j);
      acc = mfma_seq(j);
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(acc.v[0][0]), "+v"(acc.v[0][1]), "+v"(acc.v[1][0]), "+v"(acc.v[1][1]) :: "memory");
      bar();
      if (CVMI_TL16_PRIO == 2) __builtin_amdgcn_s_setprio(1); else if (CVMI_TL16_PRIO == 3) __builtin_amdgcn_s_setprio(0);
      if (!(CVMI_TL16_DIAG & 1) && j + SLOTS - 1 < nch) issue_chunk(j + SLOTS - 1);
      if (!(CVMI_TL16_DIAG & 4)) epilogue(acc, j);
    }
  } else {
#pragma unroll 1
    for (int j = j0; j < nch; ++j) {
      bar();
      if (CVMI_TL16_PRIO == 2) __builtin_amdgcn_s_setprio(1); else if (CVMI_TL16_PRIO == 3) __builtin_amdgcn_s_setprio(0);
      if (!(CVMI_TL16_DIAG & 1) && j + SLOTS - 1 < nch) issue_chunk(j + SLOTS - 1);
      if (!(CVMI_TL16_DIAG & 4) && j > j0) epilogue(acc, j - 1);
      if constexpr (RES) res_load(j);
      bar();
      if (CVMI_TL16_PRIO == 2) __builtin_amdgcn_s_setprio(0); else if (CVMI_TL16_PRIO == 3) __builtin_amdgcn_s_setprio(1);
      acc = mfma_seq(j);
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(acc.v[0][0]), "+v"(acc.v[0][1]), "+v"(acc.v[1][0]), "+v"(acc.v[1][1]) :: "memory");
    }
    epilogue(acc, nch - 1);
  }
  stats_tail();
}

// Row-block sharing (see the kernel): how many workgroups split the chunks of one 256-row block.  1 unless the launch would leave CUs idle
// (fewer than 256 row blocks, a multiple of 8 of them); then the divisor of the chunk count that fills most of one round of 256 CUs.
// With LayerNorm statistics out (RES form) the slices must be whole, equal chunk ranges (N % 32 == 0).
int tl16_splits(long long rows, int N, bool stats_out) {
  static const int on = getenv("CVMI_TL16_SPLIT") ? atoi(getenv("CVMI_TL16_SPLIT")) : 1;      // A/B runs only
  const long long wg = rows / 256;
  const int nch = (N + 31) / 32;
  if (!on || wg >= 256 || wg % 8 != 0 || (stats_out && N % 32 != 0)) return 1;
  int best = 1;
  for (int ns = 2; ns <= 8 && ns <= nch; ++ns)
    if (nch % ns == 0 && wg * ns <= 256) best = ns;            // one round of at most 256 workgroups, as full as the divisors of nch allow
  return best;
}

template <int K, int LN, bool RES, bool GELU, bool POOL>
int launch16(const void* in, int in_ld, const float* gamma, const float* beta, float eps, const void* wp, void* out, int out_ld, long long rows, int N,
             hipStream_t s, const Tl16Extra& ex) {
  using Cfg = Tl16Cfg<K>;
  static hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&tok_linear16_kernel<K, LN, RES, GELU, POOL>), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
  CVMI_HIP(attr);
  cvmi_note_kernel("tok_linear16_kernel<%d, %d, %s, %s, %s>", K, LN, CVMI_BOOLNAME(RES), CVMI_BOOLNAME(GELU), CVMI_BOOLNAME(POOL));
  const int ns = tl16_splits(rows, N, RES && ex.stats_out != nullptr);
  hipLaunchKernelGGL((tok_linear16_kernel<K, LN, RES, GELU, POOL>), dim3((unsigned)(rows / 256), (unsigned)ns), dim3(TL_NW * 64), Cfg::LDS, s, in, in_ld, gamma, beta, eps,
                     (const char*)wp, out, out_ld, rows, N, ex);
  CVMI_LAUNCH_CHECK();
  return 0;
}

}  // namespace

// cvmi_tok_linear_stats_splits (tok_linear.hip): the slice count of the statistics a residual-form launch of this shape writes
int CVMI_ENTRY(cvmi_tok_linear16_splits)(long long rows, int N) { return tl16_splits(rows, N, true); }

// Called by the cvmi_tok_linear* entry points (tok_linear.hip) for the K served in the 16x16x32 format (cvmi_tok_linear_format).  Arguments are
// already validated there.  pool_w > 0 selects the POOL form.
int CVMI_ENTRY(cvmi_tok_linear16_launch)(int K, int ln, bool res, bool gelu, const void* in, int in_ld, const float* gamma, const float* beta, float eps,
                                         const void* wp, void* out, int out_ld, long long rows, int N, int pool_w, int pool_hw2, const float* stats_in,
                                         int stats_parts, float* stats_out, float stats_eps, hipStream_t s) {
  CVMI_CHECK(K == 576, "tok_linear (16x16x32 format): K=%d is not built", K);
  const Tl16Extra ex{pool_w, pool_hw2, stats_in, stats_out, stats_eps, stats_parts};
  if (pool_w > 0) return launch16<576, 1, false, false, true>(in, in_ld, gamma, beta, eps, wp, out, out_ld, rows, N, s, ex);
  CVMI_CHECK(res || (N % 8 == 0 && out_ld % 8 == 0), "tok_linear (16x16x32 format): 16-bit output needs N and out_ld multiples of 8");
  if (res) {
    if (ln) return launch16<576, 1, true, false, false>(in, in_ld, gamma, beta, eps, wp, out, out_ld, rows, N, s, ex);
    return launch16<576, 0, true, false, false>(in, in_ld, gamma, beta, eps, wp, out, out_ld, rows, N, s, ex);
  }
  if (ln) {
    if (gelu) return launch16<576, 1, false, true, false>(in, in_ld, gamma, beta, eps, wp, out, out_ld, rows, N, s, ex);
    return launch16<576, 1, false, false, false>(in, in_ld, gamma, beta, eps, wp, out, out_ld, rows, N, s, ex);
  }
  if (gelu) return launch16<576, 0, false, true, false>(in, in_ld, gamma, beta, eps, wp, out, out_ld, rows, N, s, ex);
  return launch16<576, 0, false, false, false>(in, in_ld, gamma, beta, eps, wp, out, out_ld, rows, N, s, ex);
}
