// Fused MLP half of a Hiera block:   x <- x + fc2( GELU( fc1( LayerNorm(x) ) ) )   in place on the f32 residual stream
// (sam2 hieradet MultiScaleBlock.forward: `x = x + self.drop_path(self.mlp(self.norm2(x)))`, behind
//  /root/reference/src/sam2_infer.py:226 `self.sam2_model.image_encoder(images)`).
//
// Why one kernel: unfused, the 4C-wide hidden activation is written and re-read (fp16), the normalised copy is written and
// re-read, and the stream is read three times -- 4.8 GB per stage-1 block at B = 16 against 1.2 GB for "read x, write x".
//
// Token-stationary formulation on 32x32x16 MFMA, weights as the A operand, nothing crosses lanes:
//   * a wave owns 32 tokens (token = MFMA column = lane & 31).  LayerNorm runs on the lane's half row in registers and leaves
//     the normalised row as the B fragments of fc1 (K = C, fp16), resident for the whole kernel;
//   * the hidden dimension is walked in chunks of 32 units:  H^T[32 hidden][32 tokens] = W1_chunk . Xn^T  (C/16 MFMAs + one
//     extra k-step that carries the fc1 bias as a (hi + lo) fp16 pair against constant-1 columns of Xn);
//   * GELU is applied to the accumulator, which converted to fp16 IS the B operand of fc2 (the MFMA C/D layout has the token on
//     the lane and the hidden unit in the register: "accumulator tile as the next MFMA's operand"):
//                        Y^T[C][32 tokens] += W2^T_chunk[C][32 hidden] . H^T          (2 * ceil(C/32) MFMAs)
//     with W2's fragment packed in the permuted k order that layout implies (include/cvmi355.h, cvmi_hiera_mlp);
//   * the weight chunks (both matrices, fragment order, 1 KiB per MFMA operand) stream L2 -> LDS by global_load_lds into a
//     two-slot ring, one barrier per chunk; every ds_read_b128 is lane-linear (conflict-free);
//   * epilogue: + fc2 bias + the old x, 16-byte f32 stores.
// C = 144 (stage 1): 8 waves / workgroup, two waves per SIMD (the GELU VALU of one overlaps the MFMAs of the other).
// C = 288 (stage 2): the Y^T accumulator (144 registers) + Xn (76) leave room for one wave per SIMD only: 4 waves.  Its chunk loop is
// software-pipelined inside the wave and written as asm blocks (VAR 2, below): nobody else is there to fill the matrix pipe while GELU issues.
#include "common.hpp"
#include <stdlib.h>
#include <type_traits>

namespace {

// MFMAs of the pipelined loop are inline asm so that the register FILE of every accumulator is fixed by its constraint: the fc2 accumulators
// ("+a": AGPRs, touched by nothing but MFMAs until the epilogue) and the hidden accumulators ("v": VGPRs, which GELU reads directly).  Written with
// the builtin, hipcc moves the accumulators between the files around every fc2 MFMA once two chunks are in flight (r04: hundreds of
// v_accvgpr_* per pair of chunks).  hipcc's hazard recogniser does not look inside inline asm: the waits an MFMA result needs before a
// non-MFMA instruction reads it (12 wait states on gfx950 for this shape) are provided by the program order below and marked where they are.
#ifdef CVMI_OPERAND_BF16
#define CVMI_MFMA_ASM "v_mfma_f32_32x32x16_bf16"
#else
#define CVMI_MFMA_ASM "v_mfma_f32_32x32x16_f16"
#endif

// fragment of a stage (fc2's 2 NT fragments of one chunk, then fc1's N1 of the next) that the pipelined loop consumes i-th: fc1's k-steps, then
// fc2's k-half 0 of every output tile, then k-half 1
template <int N1, int NT> constexpr int stage_frag(int i) { return i < N1 ? 2 * NT + i : (i - N1 < NT ? 2 * (i - N1) : 2 * (i - N1 - NT) + 1); }

template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// VAR 1: 4-wave workgroups, two waves per SIMD coming from two INDEPENDENT workgroups (C = 144: instead of one 8-wave workgroup; C = 288:
// instead of one wave per SIMD, at 256 registers with ~20 spilled).  SLOTS: depth of the weight ring.  Both measured in r04 (profiles/r04_ab_runs.md).
// VAR 2: one wave per SIMD with the chunk loop software-pipelined inside the wave (see `pipelined chunk loop` in the kernel).
template <int C, int VAR = 0, int SLOTS_ = 2> struct MlpCfg {
  static constexpr int KS = C / 16;                 // k-steps of fc1
  static constexpr int KS1 = KS + 1;                // + the bias step
  static constexpr int NT = (C + 31) / 32;          // 32-channel output tiles of fc2
  static constexpr int NCH = 4 * C / 32;            // hidden chunks
  static constexpr int FR = KS1 + 2 * NT;           // 1 KiB fragments per chunk
  static constexpr int CHB = FR * 1024;             // bytes per chunk
  static constexpr int NW = VAR >= 1 ? 4 : (C <= 144 ? 8 : 4);       // waves per workgroup
  static constexpr int WPS = VAR == 2 ? (C <= 144 ? 2 : 1) : VAR == 1 ? 2 : (C <= 144 ? 2 : 1);      // waves per SIMD the register budget is set for
  static constexpr int SLOTS = SLOTS_;
  static constexpr int CNT = (FR + NW - 1) / NW;    // LDS-DMA instructions every wave issues per chunk (the same count in every wave: counted waits)
  static constexpr int LDS = SLOTS * CHB + 1024;    // + a dump piece for the padding instructions of the waves with fewer real pieces
};

// DIAG (timing-only builds, -DCVMI_MLP_DIAGS; results are wrong by design): 1 no GELU, 2 no MFMAs, 3 no weight DMA inside the loop, 4 no LDS reads,
// 5 = 1 + 3 + 4 (MFMAs and barriers only).  A template parameter so that the shipped instantiation's code is untouched.
template <int C, int VAR = 0, int SLOTS = 2, int DIAG = 0>
__global__ __launch_bounds__((MlpCfg<C, VAR, SLOTS>::NW * 64), (MlpCfg<C, VAR, SLOTS>::WPS)) void hiera_mlp_kernel(float* __restrict__ x, int x_ld, const float* __restrict__ gamma,
                                                                                      const float* __restrict__ beta, float eps,
                                                                                      const char* __restrict__ wp, const float* __restrict__ b2,
                                                                                      long long rows, float* __restrict__ stats_out, float stats_eps) {
  using Cfg = MlpCfg<C, VAR, SLOTS>;
  constexpr int KS = Cfg::KS, KS1 = Cfg::KS1, NT = Cfg::NT, NCH = Cfg::NCH, FR = Cfg::FR, CHB = Cfg::CHB, NW = Cfg::NW, CNT = Cfg::CNT;
  static_assert(SLOTS == 2 || SLOTS == 3, "ring depth");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int lr = lane & 31, lh = lane >> 5;
  const long long row_raw = ((long long)blockIdx.x * NW + wv) * 32 + lr;
  const bool row_ok = row_raw < rows;
  float* const xr = x + (row_ok ? row_raw : rows - 1) * (long long)x_ld;

  // ---- weight stream: chunk j -> ring slot j % SLOTS; wave w moves fragments w, w + NW, ...  Every wave issues exactly CNT instructions per
  // chunk (a wave without a last real piece re-loads piece 0 into the dump area behind the ring), so that one immediate vmcnt count fits all.
  auto issue_chunk = [&](int j) {
    const char* src = wp + (size_t)j * CHB + lane * 16;
    char* dst = smem + (j % SLOTS) * CHB;
#pragma unroll
    for (int f = 0; f < CNT; ++f) {
      const int fi = f * NW + wv;
      const bool real = fi < FR;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)(real ? fi : 0) * 1024),
                                       (__attribute__((address_space(3))) void*)(real ? dst + fi * 1024 : smem + SLOTS * CHB), 16, 0, 0);
    }
  };
  // Pipelined form (VAR 2): the unit of the stream is a STAGE = the FR consecutive fragments that one loop iteration consumes: fc2's fragments of
  // chunk s followed by fc1's of chunk s + 1 (contiguous in the packed stream).  Stage -1 is fc1's part of chunk 0 alone, stage NCH - 1 fc2's alone.
  auto issue_piece = [&](int st, int f) {                            // this wave's f-th DMA instruction of stage st (every wave issues CNT per stage)
    const int lo = st < 0 ? 2 * NT : 0, hi = st == NCH - 1 ? 2 * NT : FR;
    const char* src = wp + ((long long)st * FR + KS1) * 1024 + lane * 16;
    char* dst = smem + ((st + SLOTS) % SLOTS) * CHB;
    const int fi = f * NW + wv;
    const bool real = fi >= lo && fi < hi;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (long long)(real ? fi : lo) * 1024),
                                     (__attribute__((address_space(3))) void*)(real ? dst + fi * 1024 : smem + SLOTS * CHB), 16, 0, 0);
  };
  auto issue_stage = [&](int st) {
#pragma unroll
    for (int f = 0; f < CNT; ++f) issue_piece(st, f);
  };
  if constexpr (VAR == 2) {
#pragma unroll
    for (int st = -1; st < SLOTS - 1; ++st) issue_stage(st);
  } else {
#pragma unroll
    for (int j = 0; j < SLOTS - 1; ++j) issue_chunk(j);
  }

  // ---- LayerNorm of this lane's half row -> B fragments of fc1 (lane (token lr, half lh) holds channels 16 s + 8 lh .. + 7)
  u32x4 xn[KS1];
  {
    float v[KS][8];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(xr + 16 * k + 8 * lh), b = *reinterpret_cast<const f32x4*>(xr + 16 * k + 8 * lh + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[k][e] = a[e]; v[k][4 + e] = b[e]; s += a[e] + b[e]; }
    }
    s += __shfl_xor(s, 32);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < KS; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[k][e] - mean; q += d * d; }
    q += __shfl_xor(q, 32);
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const float* gp = gamma + 16 * k + 8 * lh;
      const float* bp = beta + 16 * k + 8 * lh;
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp), g1 = *reinterpret_cast<const f32x4*>(gp + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 4);
      f16x8 h;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        h[e] = (f16)((v[k][e] - mean) * rstd * g0[e] + b0[e]);
        h[4 + e] = (f16)((v[k][4 + e] - mean) * rstd * g1[e] + b1[e]);
      }
      xn[k] = __builtin_bit_cast(u32x4, h);
    }
    // bias step: k = C and C + 1 are constant-1 columns (lanes of half 0 hold k = C .. C + 7)
    const u32x4 one = {lh == 0 ? CVMI_ONE16X2 : 0u, 0u, 0u, 0u};
    xn[KS] = one;
  }

  f32x16 yacc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) yacc[t][r] = 0.f;

  if constexpr (VAR == 2) {
    // ---- pipelined chunk loop.  Iteration j of one wave:   fc1(j + 1) -> hout   ||   GELU(hin = fc1(j)) -> pfv   ||   fc2(j): yacc += W2 . pfv
    // Left in chunk order, the wave runs fc1's MFMAs, then ~200 issue slots of GELU with the matrix pipe idle, then fc2's MFMAs -- and with one wave per
    // SIMD nobody else fills the gap.  Here GELU's steps sit between the MFMAs of the NEXT chunk's fc1 and of this chunk's first fc2 half (fc2 runs the
    // k-half-0 MFMAs of every tile first: they need only the first eight GELU values).  One wave issues one instruction per ~4 - 5 cycles and an MFMA
    // occupies the pipe for 32: the loop is written as asm blocks -- {counted LDS wait, MFMA, refill of the ring register it read} and the GELU steps of
    // GeluPk2Steps -- because what hipcc makes of the same source carries ~150 s_nop and waits per chunk, each an issue slot (r04: 423 slots per chunk
    // for 37 MFMAs; timing-only builds, profiles/r04_ab_runs.md).
    // Register files are fixed by the constraints: yacc in AGPRs ("+a"), the hidden accumulators in VGPRs (GELU reads them directly).
    // hipcc's hazard recogniser does not look inside asm: what the hardware needs around these MFMAs is provided here by construction --
    //   * a VGPR a VALU instruction wrote needs wait states before an MFMA reads it (measured: a LayerNorm conversion that hipcc had sunk to
    //     right in front of the consuming block gave wrong results in whole waves): xn and pfv pass through an `s_nop 1` asm statement first;
    //   * an MFMA's result needs 12 wait states before a VALU instruction reads it: GELU reads hin >= 18 MFMAs after its last MFMA, the epilogue
    //     reads yacc behind s_nop 15;
    //   * dependent MFMAs on one accumulator may follow each other directly.
    static_assert(SLOTS == 2, "two stages in LDS");
    constexpr int PF = 6;
    constexpr bool NO_GELU = DIAG == 1 || DIAG == 5, NO_MFMA = DIAG == 2, NO_DMA = DIAG == 3 || DIAG == 5, NO_LDS = DIAG == 4 || DIAG == 5;
    f32x16 hA, hB;
    // The PF registers the weight fragments pass through belong to the loop from here to its end ("+v" everywhere: a block reads its fragment and
    // refills the same registers).
    u32x4 ring_[PF];
#pragma unroll
    for (int f = 0; f < PF; ++f) asm volatile("" : "=v"(ring_[f]));
#pragma unroll
    for (int k = 0; k < KS1; ++k) asm volatile("s_nop 1" : "+v"(xn[k]));
    GeluPk2Steps gs[4];                                                 // couple c = value pairs 2 c, 2 c + 1 of the lane's 16 hidden values
    u32x4 pfv[2] = {};                                                  // GELU's output = fc2's B operands: dword p & 3 of pfv[p >> 2] is pair p
    uint32_t c1v = GeluPk2Steps::GELU_C1_H;
    asm volatile("" : "+v"(c1v));
    auto gelu_step = [&](f32x16& h, auto kc) {                          // step k = 4 c + s of 16
      constexpr int k = decltype(kc)::value, c = k >> 2;
      if constexpr (NO_GELU) return;
      GeluPk2Steps& g = gs[c];
      if constexpr ((k & 3) == 0) g.s0(h[4 * c], h[4 * c + 1], h[4 * c + 2], h[4 * c + 3]);
      else if constexpr ((k & 3) == 1) g.s1(c1v);
      else if constexpr ((k & 3) == 2) g.s2();
      else {
        g.s3();
        pfv[c >> 1][2 * (c & 1)] = g.ra;
        pfv[c >> 1][2 * (c & 1) + 1] = g.rb;
      }
    };
    // mode 0: fc1(0) alone out of stage -1;  1: steady state;  2: the last iteration (no next chunk)
    auto iter = [&](auto mode_, int slot, f32x16& hin, f32x16& hout) {
      constexpr int MODE = decltype(mode_)::value;
      constexpr int N1 = MODE == 2 ? 0 : KS1, N2 = MODE == 0 ? 0 : 2 * NT, NF = N1 + N2;
      const unsigned lbase_ = (unsigned)(size_t)((const __attribute__((address_space(3))) char*)(smem + slot * CHB + lane * 16));
      static_for<0, PF>([&](auto fc) {
        constexpr int f = decltype(fc)::value;
        u32x4(&ring)[PF] = ring_;                                       // (clang does not capture a variable that only asm operands name)
        const unsigned lbase = lbase_;
        if constexpr (!NO_LDS) asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(ring[f]) : "v"(lbase), "n"(stage_frag<N1, NT>(f) * 1024));
      });
      if constexpr (MODE != 0) asm volatile("" : "+v"(hin));          // (hin's last MFMA is >= 18 MFMAs back: no wait needed before GELU reads it)
      static_for<0, NF>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        u32x4(&ring)[PF] = ring_;
        const unsigned lbase = lbase_;
        f32x16 &hi_ = hin, &ho_ = hout;
        u32x4(&pfw)[2] = pfv;
        u32x4(&xf)[KS1] = xn;
        f32x16(&ya)[NT] = yacc;
        constexpr int young = NO_LDS ? 15 : ((NF - 1 - i) < (PF - 1) ? (NF - 1 - i) : (PF - 1));
        constexpr bool refill = i + PF < NF && !NO_LDS;
        constexpr int off = refill ? stage_frag<N1, NT>(i + PF) * 1024 : 0;
        if constexpr (i == N1 && MODE != 0) asm volatile("s_nop 1" : "+v"(pfw[0]));            // VALU result -> MFMA operand
        if constexpr (i == N1 + NT && MODE != 0) asm volatile("s_nop 1" : "+v"(pfw[1]));
#define CVMI_MLP_BLK(TAIL, ACCC, ACC, B)                                                                                                              \
  if constexpr (refill)                                                                                                                                \
    asm volatile("s_waitcnt lgkmcnt(%[n])\n\t" CVMI_MFMA_ASM " %[acc], %[a], %[b], " TAIL "\n\tds_read_b128 %[a], %[lb] offset:%[off]"               \
                 : [acc] ACCC(ACC), [a] "+v"(ring[i % PF]) : [b] "v"(B), [lb] "v"(lbase), [n] "n"(young), [off] "n"(off));                             \
  else                                                                                                                                                 \
    asm volatile("s_waitcnt lgkmcnt(%[n])\n\t" CVMI_MFMA_ASM " %[acc], %[a], %[b], " TAIL : [acc] ACCC(ACC), [a] "+v"(ring[i % PF]) : [b] "v"(B), [n] "n"(young))
        if constexpr (NO_MFMA) {
          if constexpr (i == 0 && N1 > 0) asm volatile("" : "=v"(ho_));
          if constexpr (refill) asm volatile("s_waitcnt lgkmcnt(%2)\n\tds_read_b128 %0, %1 offset:%3" : "+v"(ring[i % PF]) : "v"(lbase), "n"(young), "n"(off));
        } else if constexpr (i < N1) {
          if constexpr (i == 0) { CVMI_MLP_BLK("0", "=&v", ho_, xf[i]); }
          else { CVMI_MLP_BLK("%[acc]", "+v", ho_, xf[i]); }
        } else {
          constexpr int t = (i - N1) % NT, s2 = (i - N1) / NT;
          CVMI_MLP_BLK("%[acc]", "+a", ya[t], pfw[s2]);
        }
#undef CVMI_MLP_BLK
        // GELU steps behind this MFMA.  Steady state: the 16 steps spread evenly over fc1's MFMAs and fc2's first half (pfv[0] = steps 0 - 7 is
        // complete well inside fc1, pfv[1] before fc2's second half).  Last iteration: steps 0 - 7 came up front, 8 - 15 sit under fc2's first half.
        constexpr int SL = MODE == 1 ? N1 + NT : NT, K0 = MODE == 1 ? 0 : 8;
        if constexpr (MODE != 0 && i < SL)
          static_for<K0, 16>([&](auto kc) {
            if constexpr ((decltype(kc)::value - K0) * SL / (16 - K0) == i) gelu_step(hi_, kc);
          });
      });
    };
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((SLOTS - 1) * CNT) : "memory");   // stage -1 (all but the youngest stages' DMA instructions of this wave)
    __syncthreads();
    iter(std::integral_constant<int, 0>{}, SLOTS - 1, hA, hA);
    auto step = [&](auto mode, int j, f32x16& hin, f32x16& hout) {
      // own pieces of stage j; the barrier publishes them and retires stage j - 1, whose slot the next fetch overwrites.  (Measured r04 and not kept:
      // three slots with the DMA instructions of stage j + 2 spread one by one between this iteration's MFMAs -- 587 us against 580.)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (j + 1 < NCH && !NO_DMA) issue_stage(j + 1);
      iter(mode, j % SLOTS, hin, hout);
    };
    static_assert(NCH % 2 == 0, "pairs of iterations");
#pragma unroll 1
    for (int j = 0; j < NCH - 2; j += 2) {
      step(std::integral_constant<int, 1>{}, j, hA, hB);
      step(std::integral_constant<int, 1>{}, j + 1, hB, hA);
    }
    step(std::integral_constant<int, 1>{}, NCH - 2, hA, hB);
    {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      asm volatile("" : "+v"(hB));
      static_for<0, 8>([&](auto kc) { gelu_step(hB, kc); });
      iter(std::integral_constant<int, 2>{}, (NCH - 1) % SLOTS, hB, hB);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) asm volatile("s_nop 15" : "+a"(yacc[t]));   // the epilogue's v_accvgpr_read after the last MFMAs
#pragma unroll
    for (int f = 0; f < PF; ++f) asm volatile("" ::"v"(ring_[f]));
  } else
#pragma unroll 1
  for (int j = 0; j < NCH; ++j) {
    // Every wave waits for its OWN DMA pieces (hipcc puts no vmcnt wait in front of a barrier for LDS-DMA writes: without this
    // explicit wait the chunk is read before it has landed -- rare, load-dependent wrong results), then the barrier publishes
    // chunk j to the workgroup and retires the reads of slot (j + 1) & 1.
    // Ring of SLOTS chunks: chunk j + SLOTS - 1 is issued here, after the barrier that retires the reads of chunk j - 1 (its slot); with three
    // slots a chunk has TWO chunk times to land (measured r04: with two slots and one wave per SIMD the loop waited ~1600 of ~2800 cycles per
    // chunk on this wait), and the wait counts: all but the youngest chunk's CNT instructions of this wave.
    if (SLOTS == 2 || j == NCH - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT) : "memory");
    __syncthreads();
    if (j + SLOTS - 1 < NCH) issue_chunk(j + SLOTS - 1);
    const char* const buf = smem + (j % SLOTS) * CHB + lane * 16;
    // The chunk's FR operand fragments are consumed in one fixed order (fc1's KS1, then fc2's 2 NT); a ring of PF registers
    // keeps PF ds_read_b128 in flight ahead of the MFMA that consumes them -- across the GELU as well, so that fc2's first
    // operands arrive while the VALU works.  (Left to itself hipcc emits read -> lgkmcnt(0) -> MFMA per step.)
    // (The reads and their COUNTED waits are inline asm: hipcc answers the same source with read -> lgkmcnt(0) -> MFMA groups.  LDS
    // returns data in issue order, so before fragment f is used at most min(PF - 1, FR - 1 - f) younger reads may be outstanding;
    // the loop holds no other LDS / scalar-memory traffic.)
    constexpr int PF = 6;
    u32x4 ring[PF];
    const unsigned lbase = (unsigned)(size_t)((const __attribute__((address_space(3))) char*)buf);
#pragma unroll
    for (int f = 0; f < PF; ++f) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[f]) : "v"(lbase), "i"(f * 1024));
    f32x16 hacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) hacc[r] = 0.f;
    f16x8 pf[2];
#pragma unroll
    for (int f = 0; f < FR; ++f) {
      const int young = (FR - 1 - f) < (PF - 1) ? (FR - 1 - f) : (PF - 1);
      switch (young) {                                         // (compile-time after unrolling)
        case 0: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ring[f % PF])); break;
        case 1: asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(ring[f % PF])); break;
        case 2: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ring[f % PF])); break;
        case 3: asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(ring[f % PF])); break;
        case 4: asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ring[f % PF])); break;
        default: asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(ring[f % PF])); break;
      }
      const f16x8 a = __builtin_bit_cast(f16x8, ring[f % PF]);
      if (f < KS1) {
        hacc = CVMI_MFMA_32X32X16(a, __builtin_bit_cast(f16x8, xn[f]), hacc, 0, 0, 0);
      } else {
        const int t = (f - KS1) >> 1, s2 = (f - KS1) & 1;
        yacc[t] = CVMI_MFMA_32X32X16(a, pf[s2], yacc[t], 0, 0, 0);
      }
      if (f + PF < FR) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[f % PF]) : "v"(lbase), "i"((f + PF) * 1024));
      if (f == KS1 - 1) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f16x2 g2 = gelu_fast_pk(hacc[r], hacc[r + 1]);
          pf[r >> 3][r & 7] = g2[0];
          pf[r >> 3][(r & 7) + 1] = g2[1];
        }
      }
    }
  }

  // ---- epilogue: x += y + b2; lane (token lr, half lh) owns channels 32 t + 8 g + 4 lh .. + 3.
  // stats_out: the updated row's LayerNorm statistics (mean, 1 / sqrt(var + stats_eps)) for the launch that normalises it next (the next
  // block's norm1): shifted sums over this lane's half of the row, the shift being the row's first updated value.
  float st_shift = 0.f, st_s = 0.f, st_q = 0.f;
  if (stats_out) {
    const float o0 = row_ok ? xr[0] + (yacc[0][0] + b2[0]) : 0.f;       // (the value the lane of half 0 stores at channel 0)
    st_shift = __shfl(o0, lr);
  }
  if (row_ok) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = 32 * t + 8 * g + 4 * lh;
        if (c0 < C) {
          const f32x4 bb = *reinterpret_cast<const f32x4*>(b2 + c0);
          f32x4 o = *reinterpret_cast<const f32x4*>(xr + c0);
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] += yacc[t][4 * g + e] + bb[e];
          *reinterpret_cast<f32x4*>(xr + c0) = o;
          if (stats_out) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float dv = o[e] - st_shift; st_s += dv; st_q = fmaf(dv, dv, st_q); }
          }
        }
      }
  }
  if (stats_out) {
    const float ss = st_s + __shfl_xor(st_s, 32), qq = st_q + __shfl_xor(st_q, 32);
    const float dm = ss / (float)C;
    const float var = fmaxf(qq / (float)C - dm * dm, 0.f);
    if (row_ok && lh == 0) *reinterpret_cast<float2*>(stats_out + 2 * row_raw) = make_float2(st_shift + dm, 1.0f / sqrtf(var + stats_eps));
  }
}

template <int C, int VAR = 0, int SLOTS = 2, int DIAG = 0>
int launch_mlp(float* x, int x_ld, const float* gamma, const float* beta, float eps, const void* wp, const float* b2, long long rows, hipStream_t s,
               float* stats_out, float stats_eps) {
  using Cfg = MlpCfg<C, VAR, SLOTS>;
  static hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&hiera_mlp_kernel<C, VAR, SLOTS, DIAG>), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
  CVMI_HIP(attr);
  const long long per = (long long)Cfg::NW * 32;
  cvmi_note_kernel("hiera_mlp_kernel<%d, %d, %d>", C, VAR, SLOTS);
  hipLaunchKernelGGL((hiera_mlp_kernel<C, VAR, SLOTS, DIAG>), dim3((unsigned)((rows + per - 1) / per)), dim3(Cfg::NW * 64), Cfg::LDS, s, x, x_ld, gamma, beta, eps,
                     (const char*)wp, b2, rows, stats_out, stats_eps);
  CVMI_LAUNCH_CHECK();
  return 0;
}

}  // namespace

#ifndef CVMI_OPERAND_BF16
// C = 576 (stage 3) is not built: its Y^T accumulator alone is 288 registers per lane and, with the 148 of the resident Xn fragments,
// leaves no room to keep LDS reads in flight (hipcc spills 1.8 KB per lane); stage 3 / 4 stay on the tiled GEMM kernels.
extern "C" int cvmi_hiera_mlp_supported(int C) { return C == 144 || C == 288; }

extern "C" size_t cvmi_hiera_mlp_packed_bytes(int C) {
  if (!cvmi_hiera_mlp_supported(C)) return 0;
  const size_t fr = (size_t)(C / 16 + 1) + 2 * (size_t)((C + 31) / 32);
  return (size_t)(4 * C / 32) * fr * 1024;
}
extern "C" int cvmi_hiera_mlp_stats_bf16(void* x, int x_ld, const float* gamma, const float* beta, float eps, const void* w_packed, const float* b2,
                                         long long rows, int C, int dtype, float* ln_stats_out, float ln_stats_eps, cvmi_stream_t stream_);
#endif

// cvmi_hiera_mlp that also writes, per updated row, the LayerNorm statistics (mean, 1 / sqrt(var + ln_stats_eps)) the NEXT launch needs
// (the next block's norm1, fused into cvmi_tok_linear_stats / cvmi_tok_linear_pool_stats): ln_stats_out = float[2 * rows] or NULL.
extern "C" int CVMI_ENTRY(cvmi_hiera_mlp_stats)(void* x, int x_ld, const float* gamma, const float* beta, float eps, const void* w_packed, const float* b2,
                                                long long rows, int C, int dtype, float* ln_stats_out, float ln_stats_eps, cvmi_stream_t stream_) {
#ifndef CVMI_OPERAND_BF16
  if (dtype == CVMI_BF16) return cvmi_hiera_mlp_stats_bf16(x, x_ld, gamma, beta, eps, w_packed, b2, rows, C, dtype, ln_stats_out, ln_stats_eps, stream_);
#endif
  CVMI_CHECK(dtype == CVMI_T16, "hiera_mlp: dtype must be CVMI_F16 or CVMI_BF16");
  CVMI_CHECK(x && gamma && beta && w_packed && b2 && rows > 0, "hiera_mlp: bad arguments");
  CVMI_CHECK(C == 144 || C == 288, "hiera_mlp: C=%d is not built (144, 288)", C);
  CVMI_CHECK(x_ld >= C && x_ld % 4 == 0 && (((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)w_packed | (uintptr_t)b2) & 15) == 0 &&
                 ((uintptr_t)ln_stats_out & 7) == 0,
             "hiera_mlp: pointers / ld must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream_;
  float* xf = (float*)x;
  // A/B switches (measured r04, profiles/r04_ab_runs.md): CVMI_MLP_PIPE bit 1 (default on) -> C = 288 on the software-pipelined loop (VAR 2);
  // CVMI_MLP_VAR bit 0 -> C = 144, bit 1 -> C = 288 (with PIPE = 0) in 4-wave workgroups at two waves per SIMD; CVMI_MLP_SLOTS = 2 | 3, the ring
  // depth of the chunk-order loops
  static const int var = getenv("CVMI_MLP_VAR") ? atoi(getenv("CVMI_MLP_VAR")) : 1;
  const char* const slots_env = getenv("CVMI_MLP_SLOTS");
  const int slots = slots_env ? atoi(slots_env) : 2;
  const char* const pipe_env = getenv("CVMI_MLP_PIPE");        // (read per call: tests/test_ops_gpu.py switches it between two launches)
  const int pipe = pipe_env ? atoi(pipe_env) : 2;
#define CVMI_MLP_GO(CC, VV, SS) return launch_mlp<CC, VV, SS>(xf, x_ld, gamma, beta, eps, w_packed, b2, rows, s, ln_stats_out, ln_stats_eps)
  if (C == 144) {
    if (var & 1) { if (slots == 3) CVMI_MLP_GO(144, 1, 3); CVMI_MLP_GO(144, 1, 2); }
    if (slots == 3) CVMI_MLP_GO(144, 0, 3);
    CVMI_MLP_GO(144, 0, 2);
  }
#ifdef CVMI_MLP_DIAGS
  static const int diag = getenv("CVMI_MLP_DIAG") ? atoi(getenv("CVMI_MLP_DIAG")) : 0;
#define CVMI_MLP_DIAG_GO(D) if (C == 288 && diag == D) return launch_mlp<288, 2, 2, D>(xf, x_ld, gamma, beta, eps, w_packed, b2, rows, s, ln_stats_out, ln_stats_eps)
  CVMI_MLP_DIAG_GO(1); CVMI_MLP_DIAG_GO(2); CVMI_MLP_DIAG_GO(3); CVMI_MLP_DIAG_GO(4); CVMI_MLP_DIAG_GO(5);
#endif
  if (pipe & 2) CVMI_MLP_GO(288, 2, 2);
  if (var & 2) CVMI_MLP_GO(288, 1, 2);                  // (two workgroups per CU leave room for two slots of 37 KiB each only)
  if (slots == 3) CVMI_MLP_GO(288, 0, 3);
  CVMI_MLP_GO(288, 0, 2);
#undef CVMI_MLP_GO
}

extern "C" int CVMI_ENTRY(cvmi_hiera_mlp)(void* x, int x_ld, const float* gamma, const float* beta, float eps, const void* w_packed, const float* b2,
                                          long long rows, int C, int dtype, cvmi_stream_t stream_) {
  return CVMI_ENTRY(cvmi_hiera_mlp_stats)(x, x_ld, gamma, beta, eps, w_packed, b2, rows, C, dtype, nullptr, 0.f, stream_);
}
