// Shared device/host helpers for libcvmi355 (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/cvmi355.h"

// ---- the 16-bit operand type of a translation unit -----------------------------------------------
// Every kernel source that computes on 16-bit operands is compiled TWICE: as is (fp16) and with -DCVMI_OPERAND_BF16 (Makefile: *_bf16.o),
// where `f16` names __bf16, the MFMA / dot builtins are the bf16 forms, CVMI_T16 is CVMI_BF16 and every exported entry point carries the
// suffix _bf16 -- the fp16 build's entry point forwards to it when its dtype argument says CVMI_BF16.  fp32 accumulation, LayerNorm
// statistics, softmax and the residual stream are the same code in both builds.
#ifdef CVMI_OPERAND_BF16
typedef __bf16 f16;
#define CVMI_MFMA_32X32X16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define CVMI_MFMA_16X16X32 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define CVMI_FDOT2 __builtin_amdgcn_fdot2_f32_bf16
#define CVMI_T16 CVMI_BF16
#define CVMI_ENTRY(name) name##_bf16
#define CVMI_ONE16X2 0x3F803F80u                 /* two 1.0 in the 16-bit operand type */
#define CVMI_LOWEST16 (-3.38e38f)
#else
typedef _Float16 f16;
#define CVMI_MFMA_32X32X16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#define CVMI_MFMA_16X16X32 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define CVMI_FDOT2 __builtin_amdgcn_fdot2
#define CVMI_T16 CVMI_F16
#define CVMI_ENTRY(name) name
#define CVMI_ONE16X2 0x3C003C00u
#define CVMI_LOWEST16 (-65504.f)
#endif
#define CVMI_IS16(dt) ((dt) == CVMI_F16 || (dt) == CVMI_BF16)
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
#ifdef CVMI_OPERAND_BF16
#define CVMI_F16NAME "__bf16"
#else
#define CVMI_F16NAME "_Float16"
#endif
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// ---- error plumbing ----------------------------------------------------------------------------
void cvmi_set_error(const char* fmt, ...);
// name of the kernel a dispatcher is about to launch (thread-local; read back by cvmi_last_kernel(): bench.py labels its per-launch
// timings with it, so a roofline entry names the kernel rocprofv3 lists, not a guess of the host side)
void cvmi_note_kernel(const char* fmt, ...);
// template-argument spellings as llvm-cxxfilt prints them, so that a tag equals the demangled rocprofv3 kernel name up to blanks
template <typename T> inline const char* cvmi_tyname() { return sizeof(T) == 4 ? "float" : "?"; }
#define CVMI_BOOLNAME(b) ((b) ? "true" : "false")
#define CVMI_FAIL(...)            \
  do {                            \
    cvmi_set_error(__VA_ARGS__);  \
    return 1;                     \
  } while (0)
#define CVMI_CHECK(cond, ...)          \
  do {                                 \
    if (!(cond)) CVMI_FAIL(__VA_ARGS__); \
  } while (0)
#define CVMI_HIP(expr)                                                                  \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess) CVMI_FAIL("%s failed: %s", #expr, hipGetErrorString(e_));     \
  } while (0)
#define CVMI_LAUNCH_CHECK()                                                             \
  do {                                                                                  \
    hipError_t e_ = hipGetLastError();                                                  \
    if (e_ != hipSuccess) CVMI_FAIL("kernel launch failed: %s", hipGetErrorString(e_)); \
  } while (0)

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- exact division of n < 65536 by d <= 65536 via one 64-bit multiply --------------------------
struct FastDiv {
  unsigned long long mul;
  __host__ void init(unsigned d) { mul = ((1ull << 32) + d - 1) / d; }
  __device__ __forceinline__ unsigned div(unsigned n) const { return (unsigned)((n * mul) >> 32); }
};

// ---- element traits ----------------------------------------------------------------------------
template <typename T> struct Elem;
template <> struct Elem<f16> {
  static constexpr int VEC = 8;
  __device__ static __forceinline__ float to_f(f16 v) { return (float)v; }
  __device__ static __forceinline__ f16 from_f(float v) { return (f16)v; }
};
template <> struct Elem<float> {
  static constexpr int VEC = 4;
  __device__ static __forceinline__ float to_f(float v) { return v; }
  __device__ static __forceinline__ float from_f(float v) { return v; }
};

// exact-GELU with erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below fp16 resolution): ~15 instructions
// instead of libm erff's ~45; used only where the result is stored as fp16.
// v_rcp_f32 (1 ulp).  `__frcp_rn` is the correctly rounded reciprocal: hipcc expands it to the 11-instruction IEEE division
// sequence, which made every fp16-mode activation epilogue ~3x longer than its arithmetic.
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

__device__ __forceinline__ float gelu_fast(float v) {
  // GELU(x) = x * Phi(x) with Phi(x) = sigmoid(x * P(x^2)), P a quadratic fitted (minimax) to the exact erf form on
  // [-8, 8]: |error| <= 2.5e-5 absolute -- below half an fp16 ulp of the activations it is stored as -- for 7 VALU + 2
  // transcendental instructions (the Abramowitz-Stegun erf it replaces took 14 + 2; libm erff ~45).  The coefficients
  // carry the -log2(e) of the exp.  Only where the result is stored as fp16 (f32 parity mode uses erff).
  const float s = fminf(v * v, 64.0f);                    // beyond |x| = 8 the sigmoid is 0 / 1 to fp32 precision; the fit is not valid there
  const float pl = fmaf(s, fmaf(s, 1.01426788e-03f, -1.06775760e-01f), -2.30112128e+00f);
  return v * fast_rcp(1.0f + __builtin_amdgcn_exp2f(v * pl));
}

// Two GELUs whose results are stored as a 16-bit pair (the fc1 epilogues of tok_linear / hiera_mlp: 8.8 G evaluations per SAM 2.1-L B = 16 pass,
// ~3 ms of pure VALU time in the f32 form).  fp16 build: the same sigmoid(x P(x^2)) form evaluated in PACKED fp16 -- v_pk_mul / v_pk_min /
// 2 x v_pk_fma / v_pk_mul / v_pk_add / v_pk_mul for the pair plus 2 x (v_exp_f16 + v_rcp_f16): 4.5 VALU + 2 transcendental issues per value
// instead of 7 + 2, and the result is already the packed pair.  The argument of the exponential carries fp16's 2^-11 relative error
// (|x P| <= 16 where the sigmoid is not yet 0 / 1 to fp16 precision: <= 0.5 % of a value that is itself <= 2^-16 of x there; 1e-3 relative
// around |x| ~ 1), below the fp16 rounding of the stored result everywhere it matters.  bf16 build: no packed bf16 arithmetic on gfx950 --
// two f32 evaluations.
#ifndef CVMI_OPERAND_BF16
__device__ __forceinline__ f16x2 gelu_fast_pk(float a, float b) {
  const f16x2 v = {(f16)a, (f16)b};
  const f16x2 s = __builtin_elementwise_min(v * v, (f16x2){(f16)64.f, (f16)64.f});
  const f16x2 pl = s * (s * (f16x2){(f16)1.01426788e-03f, (f16)1.01426788e-03f} + (f16x2){(f16)-1.06775760e-01f, (f16)-1.06775760e-01f}) +
                   (f16x2){(f16)-2.30112128e+00f, (f16)-2.30112128e+00f};
  const f16x2 arg = v * pl;
  const f16x2 d = (f16x2){(f16)__builtin_exp2f16(arg[0]), (f16)__builtin_exp2f16(arg[1])} + (f16x2){(f16)1.f, (f16)1.f};
  return v * (f16x2){(f16)__builtin_amdgcn_rcph(d[0]), (f16)__builtin_amdgcn_rcph(d[1])};
}
#else
__device__ __forceinline__ f16x2 gelu_fast_pk(float a, float b) { return (f16x2){(f16)gelu_fast(a), (f16)gelu_fast(b)}; }
#endif

// gelu_fast_pk of TWO value pairs cut into four short steps, for a loop that issues them between MFMAs:   s0(a0, a1, b0, b1);  s1();  s2();  s3();
// then ra / rb hold the two packed results.  fp16 build: the instruction sequence hipcc emits for gelu_fast_pk (identical bits), written out with the
// two pairs' chains alternating -- a packed-fp16 or transcendental result must not be read by the next instruction on gfx950 (one wait state), and
// alternating provides it without the s_nop hipcc puts there (9 per pair) -- and as asm blocks so that their place in the program order of the
// volatile asm statements around them is fixed (hipcc's instruction selection otherwise moves pure arithmetic across those, e.g. every s0 to the front).
struct GeluPk2Steps {
#ifndef CVMI_OPERAND_BF16
  uint32_t ra, rb, ta, tb;
  __device__ __forceinline__ void s0(float a0, float a1, float b0, float b1) {
    asm volatile("v_cvt_pk_f16_f32 %0, %4, %5\n\tv_cvt_pk_f16_f32 %2, %6, %7\n\t"
                 "v_pk_mul_f16 %1, %0, %0\n\tv_pk_mul_f16 %3, %2, %2\n\t"
                 "v_pk_min_f16 %1, %1, %8 op_sel_hi:[1,0]\n\tv_pk_min_f16 %3, %3, %8 op_sel_hi:[1,0]"
                 : "=&v"(ra), "=&v"(ta), "=&v"(rb), "=&v"(tb) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "s"(0x5400u));                       // 64.0
  }
  __device__ __forceinline__ void s1(uint32_t c1_vgpr) {                      // c1_vgpr: -1.06775760e-01 as fp16 in the low half of a VGPR (one SGPR per instruction)
    uint32_t ua, ub;
    asm volatile("v_pk_fma_f16 %2, %0, %6, %7 op_sel_hi:[1,0,0]\n\tv_pk_fma_f16 %3, %1, %6, %7 op_sel_hi:[1,0,0]\n\t"
                 "v_pk_fma_f16 %0, %0, %2, %8 op_sel_hi:[1,1,0]\n\tv_pk_fma_f16 %1, %1, %3, %8 op_sel_hi:[1,1,0]\n\t"
                 "v_pk_mul_f16 %0, %4, %0\n\tv_pk_mul_f16 %1, %5, %1"
                 : "+v"(ta), "+v"(tb), "=&v"(ua), "=&v"(ub) : "v"(ra), "v"(rb), "s"(GELU_C2_H), "v"(c1_vgpr), "s"(GELU_C0_H));
  }
  __device__ __forceinline__ void s2() {
    uint32_t ea, eb;
    asm volatile("v_exp_f16_e32 %2, %0\n\tv_exp_f16_e32 %3, %1\n\t"
                 "v_exp_f16_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
                 "v_exp_f16_sdwa %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
                 "v_pack_b32_f16 %0, %2, %0\n\tv_pack_b32_f16 %1, %3, %1\n\t"
                 "v_pk_add_f16 %0, %0, 1.0 op_sel_hi:[1,0]\n\tv_pk_add_f16 %1, %1, 1.0 op_sel_hi:[1,0]"
                 : "+v"(ta), "+v"(tb), "=&v"(ea), "=&v"(eb));
  }
  __device__ __forceinline__ void s3() {
    uint32_t ea, eb;
    asm volatile("v_rcp_f16_e32 %4, %2\n\tv_rcp_f16_e32 %5, %3\n\t"
                 "v_rcp_f16_sdwa %2, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
                 "v_rcp_f16_sdwa %3, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
                 "v_pack_b32_f16 %2, %4, %2\n\tv_pack_b32_f16 %3, %5, %3\n\t"
                 "v_pk_mul_f16 %0, %0, %2\n\tv_pk_mul_f16 %1, %1, %3"
                 : "+v"(ra), "+v"(rb), "+v"(ta), "+v"(tb), "=&v"(ea), "=&v"(eb));
  }
  static constexpr uint32_t GELU_C2_H = 0x1428u, GELU_C0_H = 0xC09Au;        // fp16(1.01426788e-03), fp16(-2.30112128e+00): checked against the
  static constexpr uint32_t GELU_C1_H = 0xAED5u;                              // compiler's constants by tests/test_ops_gpu.py (bit-identity with gelu_fast_pk)
#else
  uint32_t ra, rb;
  float v[4], t[4];
  __device__ __forceinline__ void fence() { asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3])); }
  __device__ __forceinline__ void s0(float a0, float a1, float b0, float b1) {
    v[0] = a0; v[1] = a1; v[2] = b0; v[3] = b1;
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] = fminf(v[e] * v[e], 64.0f);
    fence();
  }
  __device__ __forceinline__ void s1(uint32_t) {
    fence();
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] = v[e] * fmaf(t[e], fmaf(t[e], 1.01426788e-03f, -1.06775760e-01f), -2.30112128e+00f);
    fence();
  }
  __device__ __forceinline__ void s2() {
    fence();
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] = 1.0f + __builtin_amdgcn_exp2f(t[e]);
    fence();
  }
  __device__ __forceinline__ void s3() {
    fence();
    ra = __builtin_bit_cast(uint32_t, (f16x2){(f16)(v[0] * fast_rcp(t[0])), (f16)(v[1] * fast_rcp(t[1]))});
    rb = __builtin_bit_cast(uint32_t, (f16x2){(f16)(v[2] * fast_rcp(t[2])), (f16)(v[3] * fast_rcp(t[3]))});
    asm volatile("" : "+v"(ra), "+v"(rb));
  }
  static constexpr uint32_t GELU_C1_H = 0u;
#endif
};

// FAST = true: v_exp/v_rcp based (fp16 storage mode); false: precise libm (f32 parity mode)
template <bool FAST> __device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    case CVMI_ACT_SILU:
      return FAST ? v * fast_rcp(1.0f + __expf(-v)) : v / (1.0f + expf(-v));
    case CVMI_ACT_RELU: return v > 0.f ? v : 0.f;
    case CVMI_ACT_GELU: return FAST ? gelu_fast(v) : 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    case CVMI_ACT_SIGMOID:
      return FAST ? fast_rcp(1.0f + __expf(-v)) : 1.0f / (1.0f + expf(-v));
    default: return v;
  }
}
// run `body(fn)` with fn = the activation selected ONCE (p.act is uniform): no per-element switch in hot loops
template <bool FAST, typename Body> __device__ __forceinline__ void with_act(int act, Body&& body) {
  switch (act) {
    case CVMI_ACT_SILU: body([](float v) { return act_apply<FAST>(v, CVMI_ACT_SILU); }); break;
    case CVMI_ACT_RELU: body([](float v) { return act_apply<FAST>(v, CVMI_ACT_RELU); }); break;
    case CVMI_ACT_GELU: body([](float v) { return act_apply<FAST>(v, CVMI_ACT_GELU); }); break;
    case CVMI_ACT_SIGMOID: body([](float v) { return act_apply<FAST>(v, CVMI_ACT_SIGMOID); }); break;
    default: body([](float v) { return v; }); break;
  }
}
template <typename T> struct FastMath { static constexpr bool value = true; };
template <> struct FastMath<float> { static constexpr bool value = false; };

// 16-byte chunk <-> float[VEC]
template <typename T> __device__ __forceinline__ void unpack16(const u32x4& v, float* out);
template <> __device__ __forceinline__ void unpack16<f16>(const u32x4& v, float* out) {
  const f16x8 h = __builtin_bit_cast(f16x8, v);
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = (float)h[i];
}
template <> __device__ __forceinline__ void unpack16<float>(const u32x4& v, float* out) {
  const f32x4 f = __builtin_bit_cast(f32x4, v);
#pragma unroll
  for (int i = 0; i < 4; ++i) out[i] = f[i];
}
template <typename T> __device__ __forceinline__ u32x4 pack16(const float* in);
template <> __device__ __forceinline__ u32x4 pack16<f16>(const float* in) {
  f16x8 h;
#pragma unroll
  for (int i = 0; i < 8; ++i) h[i] = (f16)in[i];
  return __builtin_bit_cast(u32x4, h);
}
template <> __device__ __forceinline__ u32x4 pack16<float>(const float* in) {
  f32x4 f;
#pragma unroll
  for (int i = 0; i < 4; ++i) f[i] = in[i];
  return __builtin_bit_cast(u32x4, f);
}
