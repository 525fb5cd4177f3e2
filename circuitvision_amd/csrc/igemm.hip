// Implicit-GEMM convolution / linear layer on MFMA (gfx950).
//
//   y[m, n] = act( sum_k A[m, k] * w[n, k] + bias[n] ) (+ res[m, n])
//
// A (pixels x K) is never materialised: each 16-byte chunk of a K-row is gathered from the NHWC
// source(s) while the K-tile is staged into LDS (im2col on the fly; two channel-concatenated
// sources, each optionally nearest-2x upsampled, so YOLO's Upsample+Concat cost no HBM traffic).
// Operand roles are swapped on purpose: the MFMA "A" operand is the WEIGHT tile (rows = n) and the
// "B" operand the PIXEL tile (cols = m).  The 32x32 accumulator then holds, per lane, runs of 4
// consecutive output channels of ONE pixel, which pack into 8/16-byte LDS writes for the
// transposing epilogue and leave the global stores as full 16-byte, channel-contiguous chunks.
//
// LDS image per K-tile: rows of BKB data bytes + 16 pad bytes (row stride = odd multiple of 16 B,
// so the 32 lanes of a ds_read_b128 lane group hit 16 distinct 16-byte slots: conflict-free).
// The same byte geometry serves fp16 (32x32x16 MFMA, 8 k per lane) and exact-f32 (32x32x2 MFMA,
// 4 k per lane per 16-byte read, 4 MFMAs per read) -- only the inner MFMA call differs.
//
// Pipeline: register-staged double buffer (global loads for tile t+1 are issued before the MFMAs
// of tile t and written to the other LDS buffer after them; one barrier per K-tile).
#include "common.hpp"
#include <stdlib.h>
#include <string.h>

namespace {

struct ConvKArgs {
  const char* x0; const char* x1; const char* w; const float* bias; const char* res; char* y;
  int x0_ld, x1_ld, res_ld, y_ld;
  int c0, ctot;
  int up0, up1;
  int H, W, OH, OW, KW, stride, pad;
  int M, N, K, Kpad;
  int act, scalar_gather;
  int res_mod, act_after_res, shuf_c, res_rep;
  int plain;
  int rows2;                             // 1x1 / s1 / p0 over two concatenated sources with channel counts that are K-tile multiples
  int nb_n;
  int bias_off;                          // igemm_kernel: LDS byte offset of the parked bias row
  int rev_m;                             // gemm256x192: 1 = walk the row blocks last-to-first (see launch_g256x192)
  int per_xcd;                           // gemm256x192r: tiles per XCD
  int diag;                              // gemm256x192r, CVMI_G192_DIAG, timing experiments ONLY (results are wrong): bit 0 = no DMA behind the prologue, bit 1 = no MFMAs, bit 2 = no residual loads
  float* stats;                          // gemm256x192 (f32 out): per row and 96-column slice (mean, sum of squared deviations from it) of the values written, or null
  int im2col_shift;                      // gemm256_kernel<.., IM2COL = true>: log2(K-tiles of 64 channels per 3 x 3 tap) = log2(Cin / 64)
  FastDiv div_ctot, div_kw;
};

// 16 zero bytes for every lane of an LDS-DMA instruction whose 3 x 3 tap falls outside the image (gemm256_kernel, IM2COL)
__device__ __attribute__((aligned(256))) unsigned char cvmi_zero_page[256];

template <typename T> struct Mma;
template <> struct Mma<f16> {
  __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c) {
    c = CVMI_MFMA_32X32X16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c) {
    const f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
#pragma unroll
    for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[e], c, 0, 0, 0);
  }
};

// Returns the RAW 16 bytes at a clamped (always valid) address and sets `ok`; the caller zeroes !ok chunks when it
// writes the tile to LDS, i.e. after the MFMAs of the current tile -- a select next to the load would make the
// compiler wait for the load before the compute it is supposed to overlap.
template <typename T>
__device__ __forceinline__ u32x4 gather_chunk(const ConvKArgs& p, int k, int b, int iy0, int ix0, bool row_ok, bool& ok_out) {
  u32x4 v = {0u, 0u, 0u, 0u};
  ok_out = true;
  constexpr int VEC = Elem<T>::VEC;
  if (!p.scalar_gather) {
    // branch-free: out-of-range chunks read a clamped (valid) address and are zeroed by a select, so that the
    // loads of a K-tile issue back to back (an exec-masked branch per load makes hipcc serialise them with waits)
    const unsigned tap = p.div_ctot.div((unsigned)k);
    int ci = k - (int)tap * p.ctot;
    const unsigned ky = p.div_kw.div(tap);
    const int kx = (int)tap - (int)ky * p.KW;
    const int iy = iy0 + (int)ky, ix = ix0 + kx;
    const bool ok = row_ok && k < p.K && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    const bool s0 = ci < p.c0;
    const char* base = s0 ? p.x0 : p.x1;
    const int ld = s0 ? p.x0_ld : p.x1_ld, up = s0 ? p.up0 : p.up1;
    ci = s0 ? ci : ci - p.c0;
    const int iyc = ok ? iy : 0, ixc = ok ? ix : 0, cic = ok ? ci : 0;
    const size_t pix = ((size_t)b * (p.H >> up) + (iyc >> up)) * (size_t)(p.W >> up) + (ixc >> up);
    v = *reinterpret_cast<const u32x4*>(base + (pix * ld + cic) * sizeof(T));
    ok_out = ok;
  } else {
    // per-element gather (3-channel stems): values are inserted into the vector directly (no local array -> no scratch)
    using VT = T __attribute__((ext_vector_type(16 / sizeof(T))));
    VT tv;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const int ke = k + e;
      const unsigned tap = p.div_ctot.div((unsigned)ke);
      int ci = ke - (int)tap * p.ctot;
      const unsigned ky = p.div_kw.div(tap);
      const int kx = (int)tap - (int)ky * p.KW;
      const int iy = iy0 + (int)ky, ix = ix0 + kx;
      const bool ok = row_ok && ke < p.K && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const bool s0 = ci < p.c0;
      const char* base = s0 ? p.x0 : p.x1;
      const int ld = s0 ? p.x0_ld : p.x1_ld, up = s0 ? p.up0 : p.up1;
      ci = s0 ? ci : ci - p.c0;
      const int iyc = ok ? iy : 0, ixc = ok ? ix : 0, cic = ok ? ci : 0;
      const size_t pix = ((size_t)b * (p.H >> up) + (iyc >> up)) * (size_t)(p.W >> up) + (ixc >> up);
      const T val = *reinterpret_cast<const T*>(base + (pix * ld + cic) * sizeof(T));
      tv[e] = ok ? val : (T)0;
    }
    v = __builtin_bit_cast(u32x4, tv);
  }
  return v;
}

// residual row of output row m: plain, a constant broadcast over the batch (res_mod rows, m % res_mod), or one image's rows
// shared by res_rep consecutive batch entries (res_mod rows per image: entry e reads image e / res_rep)
__device__ __forceinline__ size_t res_row(const ConvKArgs& p, int m) {
  if (p.res_mod <= 0) return (size_t)m;
  const int r = m % p.res_mod;
  if (p.res_rep <= 1) return (size_t)r;
  return (size_t)(m / p.res_mod / p.res_rep) * (size_t)p.res_mod + (size_t)r;
}

// ---- shared epilogue: bias + act in registers -> LDS tile [BM][BN] (TO) -> coalesced 16-byte stores (+ residual,
//      batch-broadcast residual, activation-after-residual, ConvTranspose scatter) -----------------------------------
template <typename T, typename TO, int BM, int BN, int WM, int WN, int KS = 1>
__device__ __forceinline__ void gemm_epilogue(const ConvKArgs& p, f32x16 (&acc)[BN / WN / 32][BM / WM / 32], char* smem, int m0, int n0,
                                              const float* bias_lds = nullptr) {
  constexpr int NT = WM * WN * 64 * KS;             // KS > 1 (intra-workgroup split-K): the first WM*WN waves hold the reduced tile, all waves store
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int OES = sizeof(TO);
  constexpr int OVEC = 16 / OES;
  constexpr int CROWB = BN * OES + 16;
  constexpr bool FAST = FastMath<T>::value;
  const int tid = threadIdx.x;
  const int wv = tid >> 6, lane = tid & 63;
  const int wm = wv % WM, wn = wv / WM;
  const int lr = lane & 31, lh = lane >> 5;
  const int ohow = p.OH * p.OW;
  char* const Ct = smem;
  // Residual prefetch (full-width tiles, no scatter): the tile's residual chunks are requested before the accumulators
  // go through LDS, so their latency overlaps the transposition instead of being paid once per store-loop iteration
  // (loads cannot be hoisted over the stores by the compiler: res and y may be the same buffer).
  constexpr int NCH_ = BN / OVEC, NIT = (BM * NCH_) / NT;
  const bool res_pf = p.res != nullptr && p.shuf_c == 0 && n0 + BN <= p.N && (BM * NCH_) % NT == 0 && NIT <= 16;
  u32x4 rv[NIT > 0 && NIT <= 16 ? NIT : 1];
  if (res_pf) {
#pragma unroll
    for (int it = 0; it < (NIT <= 16 ? NIT : 0); ++it) {
      const int idx = tid + it * NT;
      const int row = idx / NCH_, ch = idx - row * NCH_;
      int m = m0 + row;
      m = m < p.M ? m : p.M - 1;
      const size_t rpix = res_row(p, m);
      rv[it] = *reinterpret_cast<const u32x4*>(p.res + (rpix * p.res_ld + n0 + ch * OVEC) * OES);
    }
  }
  if (KS == 1 || tid < WM * WN * 64)
  with_act<FAST>(p.act_after_res ? CVMI_ACT_NONE : p.act, [&](auto actf) {
#pragma unroll
    for (int i = 0; i < TN; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nl = wn * WTN + i * 32 + 8 * q + 4 * lh;      // 4 consecutive channels
        // (bias_lds: the tile's bias row parked in LDS by the caller -- no dependent global load after the K loop)
        const f32x4 bv = bias_lds ? *reinterpret_cast<const f32x4*>(bias_lds + nl) : *reinterpret_cast<const f32x4*>(p.bias + n0 + nl);
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          const int ml = wm * WTM + j * 32 + lr;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = actf(acc[i][j][4 * q + e] + bv[e]);
          char* dst = Ct + ml * CROWB + nl * OES;
          if constexpr (OES == 2) {
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
  constexpr int NCH = BN / OVEC;                  // 16-byte chunks per output row
  if (res_pf) {
#pragma unroll
    for (int it = 0; it < (NIT <= 16 ? NIT : 0); ++it) {
      const int idx = tid + it * NT;
      const int row = idx / NCH, ch = idx - row * NCH;
      const int m = m0 + row;
      if (m < p.M) {
        float a[OVEC], r[OVEC];
        unpack16<TO>(*reinterpret_cast<const u32x4*>(Ct + row * CROWB + ch * 16), a);
        unpack16<TO>(rv[it], r);
#pragma unroll
        for (int e = 0; e < OVEC; ++e) a[e] += r[e];
        if (p.act_after_res) {
#pragma unroll
          for (int e = 0; e < OVEC; ++e) a[e] = act_apply<FAST>(a[e], p.act);
        }
        *reinterpret_cast<u32x4*>(p.y + ((size_t)m * p.y_ld + n0 + ch * OVEC) * OES) = pack16<TO>(a);
      }
    }
    return;
  }
  for (int idx = tid; idx < BM * NCH; idx += NT) {
    const int row = idx / NCH, ch = idx - row * NCH;
    const int m = m0 + row, n = n0 + ch * OVEC;
    if (m >= p.M || n >= p.N) continue;
    u32x4 cv = *reinterpret_cast<const u32x4*>(Ct + row * CROWB + ch * 16);
    size_t ypix = (size_t)m, rpix = res_row(p, m);
    int nn = n;
    if (p.shuf_c > 0) {                            // ConvTranspose 2x2/s2: scatter to the 2x grid
      const int q = n / p.shuf_c;
      nn = n - q * p.shuf_c;
      const int b = m / ohow, r = m - b * ohow;
      const int oy = r / p.OW, ox = r - oy * p.OW;
      ypix = ((size_t)b * (2 * p.OH) + 2 * oy + (q >> 1)) * (size_t)(2 * p.OW) + 2 * ox + (q & 1);
      rpix = p.res_rep > 1 ? ((size_t)(b / p.res_rep) * (2 * p.OH) + 2 * oy + (q >> 1)) * (size_t)(2 * p.OW) + 2 * ox + (q & 1) : ypix;
    }
    char* yp = p.y + (ypix * p.y_ld + nn) * OES;
    if (n + OVEC <= p.N) {
      if (p.res || p.act_after_res) {
        float a[OVEC];
        unpack16<TO>(cv, a);
        if (p.res) {
          float r[OVEC];
          unpack16<TO>(*reinterpret_cast<const u32x4*>(p.res + (rpix * p.res_ld + nn) * OES), r);
#pragma unroll
          for (int e = 0; e < OVEC; ++e) a[e] += r[e];
        }
        if (p.act_after_res) {
#pragma unroll
          for (int e = 0; e < OVEC; ++e) a[e] = act_apply<FAST>(a[e], p.act);
        }
        cv = pack16<TO>(a);
      }
      *reinterpret_cast<u32x4*>(yp) = cv;
    } else {                                       // ragged channel tail: element-wise
      float a[OVEC];
      unpack16<TO>(cv, a);
#pragma unroll
      for (int e = 0; e < OVEC; ++e) {
        if (n + e < p.N) {
          float av = a[e];
          if (p.res) av += (float)reinterpret_cast<const TO*>(p.res + (rpix * p.res_ld + nn) * OES)[e];
          if (p.act_after_res) av = act_apply<FAST>(av, p.act);
          reinterpret_cast<TO*>(yp)[e] = (TO)av;
        }
      }
    }
  }
}

// KS > 1: intra-workgroup split-K for grids that cannot fill the chip (M = B*20*20 or B*40*40 with a deep K): KS groups of
// WM*WN waves walk disjoint K ranges of the same output tile in lock step, each with its own pair of LDS stages, so a CU
// has KS times the loads in flight and the dependent global -> LDS -> MFMA chain is KS times shorter; the partial tiles
// are summed through LDS in a fixed order (deterministic) before the common epilogue.
template <typename T, typename TO, int BM, int BN, int WM, int WN, int BKB, bool PLAIN, int KS = 1>
__global__ __launch_bounds__(WM * WN * 64 * KS, (KS > 1 ? 4 : WM * WN == 8 ? 4 : (BM * BN >= 128 * 128 ? 2 : 3))) void igemm_kernel(const ConvKArgs p) {
  // BKB = data bytes per LDS row per K-tile (64 or 128); PLAIN = 1x1 / stride 1 / one source: A is a plain
  // row-major matrix, so the per-tile gather arithmetic collapses to "row pointer + k"
  constexpr int ES = sizeof(T);
  constexpr int VEC = 16 / ES;
  constexpr int BK = BKB / ES;
  constexpr int CH = BKB / 16;            // 16-byte chunks per row
  constexpr int ROWB = BKB + 16;
  constexpr int NT = WM * WN * 64;          // threads per workgroup (4 or 8 waves)
  constexpr int A_IT = (BM * CH + NT - 1) / NT;
  constexpr int B_IT = (BN * CH + NT - 1) / NT;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  static_assert((WM * WN == 4 || WM * WN == 8) && TM >= 1 && TN >= 1 && NT % CH == 0, "4 or 8 waves, 32x32 tiles");
  constexpr int OES = sizeof(TO);
  constexpr int OVEC = 16 / OES;
  constexpr int CROWB = BN * OES + 16;    // epilogue tile row stride (bytes)

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int kg = KS > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x / NT) : 0;     // K group of this wave
  char* const As = smem + kg * (2 * (BM + BN) * ROWB);   // pixels  [2][BM][ROWB]
  char* const Bs = As + 2 * BM * ROWB;                    // weights [2][BN][ROWB]

  const int tid = KS > 1 ? (int)threadIdx.x % NT : (int)threadIdx.x;      // thread index inside the K group
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (private L2 each), so give every XCD a
  // CONTIGUOUS run of tiles -- the nb_n column tiles that share one pixel/row tile then hit the same L2 instead of
  // each XCD fetching that A tile again from the Infinity Cache (bijective for any grid size; speed only).
  int wg = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = wg & 7, j = wg >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int bn = wg % p.nb_n, bm = wg / p.nb_n;
  const int m0 = bm * BM, n0 = bn * BN;

  // ---- per-thread staging slots ---------------------------------------------------------------
  const int cchunk = tid % CH;                    // same chunk column for every slot of a thread
  int a_b[A_IT], a_iy0[A_IT], a_ix0[A_IT];
  bool a_ok[A_IT];
  const int ohow = p.OH * p.OW;
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int row = (tid + i * NT) / CH;
    const int m = m0 + row;
    a_ok[i] = (row < BM) && (m < p.M);
    const int mm = a_ok[i] ? m : 0;
    const int b = mm / ohow;
    const int r = mm - b * ohow;
    const int oy = r / p.OW, ox = r - oy * p.OW;
    a_b[i] = b;
    a_iy0[i] = oy * p.stride - p.pad;
    a_ix0[i] = ox * p.stride - p.pad;
  }

  const char* a_ptr[A_IT];
  const char* a_ptr1[A_IT];                       // rows2: row pointer into the second source
  const char* b_ptr[B_IT];
  // rows2: a 1x1 / stride-1 conv over the concat of two sources (either may be 2x nearest-upsampled): every output row reads ONE
  // row of each source, so the gather collapses to two row pointers per slot -- as cheap as PLAIN (the YOLO neck's concat convs)
  const bool rows2 = !PLAIN && p.rows2;
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int row = (tid + i * NT) / CH;
    const int m = a_ok[i] ? m0 + row : 0;
    a_ptr[i] = p.x0 + ((size_t)m * p.x0_ld + cchunk * VEC) * ES;
    a_ptr1[i] = a_ptr[i];
    if (rows2) {
      const int oy = a_iy0[i], ox = a_ix0[i];       // stride 1, pad 0: the window origin is the output pixel
      const size_t pix0 = ((size_t)a_b[i] * (p.H >> p.up0) + (oy >> p.up0)) * (size_t)(p.W >> p.up0) + (ox >> p.up0);
      const size_t pix1 = ((size_t)a_b[i] * (p.H >> p.up1) + (oy >> p.up1)) * (size_t)(p.W >> p.up1) + (ox >> p.up1);
      a_ptr[i] = p.x0 + (pix0 * p.x0_ld + cchunk * VEC) * ES;
      a_ptr1[i] = p.x1 + (pix1 * p.x1_ld + cchunk * VEC) * ES - (size_t)p.c0 * ES;      // indexed by the concat channel
    }
  }
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    const int row = (tid + i * NT) / CH;
    b_ptr[i] = p.w + ((size_t)(n0 + (row < BN ? row : 0)) * p.Kpad + cchunk * VEC) * ES;
  }

  struct Stage { u32x4 a[A_IT]; u32x4 b[B_IT]; bool m[A_IT]; };
  // fast gather: every K-tile lies inside one tap of one source (channel counts are multiples of the tile depth)
  const bool fastg = !PLAIN && !p.scalar_gather && p.ctot % BK == 0 && p.c0 % BK == 0;
  int g_ky = 0, g_kx = 0, g_cb = 0;                          // wave-uniform state of the NEXT tile load_tile() will fetch (tiles are fetched in order)
  auto load_tile = [&](Stage& st, int kt) {
    const int k = kt * BK + cchunk * VEC;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      if constexpr (PLAIN) {
        const bool ok = a_ok[i] && k < p.K;                       // branch-free (see gather_chunk)
        st.a[i] = *reinterpret_cast<const u32x4*>(a_ptr[i] + (ok ? (size_t)kt * BKB : (size_t)0) - (ok ? 0 : cchunk * 16));
        st.m[i] = ok;
      } else if (rows2) {
        const int kc = kt * BK;
        const bool ok = a_ok[i] && kc < p.K;
        const char* src = (ok && kc >= p.c0) ? a_ptr1[i] : a_ptr[i];       // masked slots read source 0's row start (a_ptr1 is biased by -c0)
        st.a[i] = *reinterpret_cast<const u32x4*>(src + (ok ? (size_t)kc * ES : (size_t)0));
        st.m[i] = ok;
      } else if (fastg) {
        // K-tile inside one (tap, source): the tap / channel bookkeeping is wave-uniform scalar state advanced per tile,
        // the per-lane part is "window origin + tap -> pixel offset" (a dozen instructions instead of gather_chunk's ~45)
        const bool s0 = g_cb < p.c0;
        const char* base = s0 ? p.x0 : p.x1;
        const int ld = s0 ? p.x0_ld : p.x1_ld, up = s0 ? p.up0 : p.up1;
        const int cofs = (s0 ? g_cb : g_cb - p.c0) + cchunk * VEC;
        const int iy = a_iy0[i] + g_ky, ix = a_ix0[i] + g_kx;
        const bool ok = a_ok[i] && kt * BK < p.K && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const int iyc = ok ? iy : 0, ixc = ok ? ix : 0;
        const size_t pix = ((size_t)a_b[i] * (p.H >> up) + (iyc >> up)) * (size_t)(p.W >> up) + (ixc >> up);
        st.a[i] = *reinterpret_cast<const u32x4*>(base + (pix * ld + (ok ? cofs : 0)) * ES);
        st.m[i] = ok;
      } else {
        st.a[i] = gather_chunk<T>(p, k, a_b[i], a_iy0[i], a_ix0[i], a_ok[i], st.m[i]);
      }
    }
    if (!PLAIN && fastg) {                                  // advance (tap, channel base) to the next K-tile
      g_cb += BK;
      if (g_cb >= p.ctot) { g_cb = 0; if (++g_kx == p.KW) { g_kx = 0; ++g_ky; } }
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) st.b[i] = *reinterpret_cast<const u32x4*>(b_ptr[i] + (size_t)kt * BKB);   // rows clamped: always valid
  };
  auto store_tile = [&](const Stage& st, int buf) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int row = (tid + i * NT) / CH;
      u32x4 v = st.a[i];
      if (!st.m[i]) v = u32x4{0u, 0u, 0u, 0u};
      if (BM * CH >= NT * (i + 1) || row < BM)
        *reinterpret_cast<u32x4*>(As + (buf * BM + row) * ROWB + cchunk * 16) = v;
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int row = (tid + i * NT) / CH;
      if (BN * CH >= NT * (i + 1) || row < BN)
        *reinterpret_cast<u32x4*>(Bs + (buf * BN + row) * ROWB + cchunk * 16) = st.b[i];
    }
  };

  const int wv = tid >> 6, lane = tid & 63;
  const int wm = wv % WM, wn = wv / WM;
  const int lr = lane & 31, lh = lane >> 5;

  f32x16 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto compute = [&](int buf) {
    const char* Ab = As + (buf * BM + wm * WTM + lr) * ROWB + lh * 16;
    const char* Bb = Bs + (buf * BN + wn * WTN + lr) * ROWB + lh * 16;
#pragma unroll
    for (int s2 = 0; s2 < BKB / 32; ++s2) {
      u32x4 wf[TN], xf[TM];
#pragma unroll
      for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const u32x4*>(Bb + i * 32 * ROWB + s2 * 32);
#pragma unroll
      for (int j = 0; j < TM; ++j) xf[j] = *reinterpret_cast<const u32x4*>(Ab + j * 32 * ROWB + s2 * 32);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) Mma<T>::run(wf[i], xf[j], acc[i][j]);
    }
  };

  // register-staged double buffer: tile t+1 is loaded while tile t is computed, then written to the other LDS buffer
  // (a second register stage was measured: no gain at equal occupancy, and its registers cost a wave per SIMD)
  const int nkt = p.Kpad / BK;
  // the tile's bias row rides along with the first K-tile and is parked in LDS behind the stages / epilogue tile
  float* const bias_s = reinterpret_cast<float*>(smem + p.bias_off);
  const float bias_v = (int)threadIdx.x < BN ? p.bias[n0 + (int)threadIdx.x] : 0.f;      // (bias is padded to 128 columns)
  if constexpr (KS == 1) {
    Stage st;
    load_tile(st, 0);
    store_tile(st, 0);
    if ((int)threadIdx.x < BN) bias_s[threadIdx.x] = bias_v;
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < nkt) load_tile(st, kt + 1);
      compute(buf);
      if (kt + 1 < nkt) store_tile(st, buf ^ 1);
      __syncthreads();
    }
  } else {
    const int nkg = (nkt + KS - 1) / KS;                      // K-tiles per group (the last group may have fewer)
    const int kt0 = kg * nkg;
    const int kend = kt0 + nkg < nkt ? kt0 + nkg : nkt;
    if (!PLAIN && fastg) {                                    // (tap, channel base) of this group's first tile
      const int k0 = kt0 * BK;
      const int tap = k0 / p.ctot;
      g_cb = k0 - tap * p.ctot;
      g_ky = tap / p.KW;
      g_kx = tap - g_ky * p.KW;
    }
    Stage st;
    if (kt0 < kend) { load_tile(st, kt0); store_tile(st, 0); }
    if ((int)threadIdx.x < BN) bias_s[threadIdx.x] = bias_v;
    __syncthreads();
    for (int i = 0; i < nkg; ++i) {                           // every group runs nkg rounds: the barriers are workgroup-wide
      const int kt = kt0 + i, buf = i & 1;
      if (kt + 1 < kend) load_tile(st, kt + 1);
      if (kt < kend) compute(buf);
      if (kt + 1 < kend) store_tile(st, buf ^ 1);
      __syncthreads();
    }
    // fixed-order reduction of the partial tiles: groups 1.. park their accumulators (lane-contiguous), group 0 adds them
    float* const red = reinterpret_cast<float*>(smem);
    constexpr int REGS = TN * TM * 16;
    if (kg > 0) {
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[((kg - 1) * REGS + (i * TM + j) * 16 + r) * NT + tid] = acc[i][j][r];
    }
    __syncthreads();
    if (kg == 0) {
#pragma unroll
      for (int g = 1; g < KS; ++g)
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += red[((g - 1) * REGS + (i * TM + j) * 16 + r) * NT + tid];
    }
    __syncthreads();                                          // the epilogue tile overlays the parked partials
  }

  gemm_epilogue<T, TO, BM, BN, WM, WN, KS>(p, acc, smem, m0, n0, bias_s);
}

// ---- plain GEMM with direct-to-LDS staging -------------------------------------------------------------------
// For A = plain row-major matrix and K a multiple of one 128-byte tile, the K-tiles go global -> LDS by
// global_load_lds (16 bytes per lane, no VGPR round trip, no ds_write -- the register-staged kernel above spends
// ~80 % of the LDS write bandwidth a full-rate MFMA stream would need).  The DMA image is lane-linear (1 KiB = 8 rows
// of 128 bytes per wave-instruction), so rows cannot be padded: bank conflicts are avoided by an XOR swizzle of the
// 16-byte slots, applied on the SOURCE address when loading and on the LDS address when reading
// (slot' = slot ^ ((row >> 1) & 7): the 16 lanes of every ds_read_b128 group then hit 16 distinct slots).
// Two LDS stages; the next tile's DMA is issued before the MFMAs of the current one and drained by the barrier.
template <typename T, typename TO, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64, 2) void gemm_glds_kernel(const ConvKArgs p) {
  constexpr int ES = sizeof(T);
  constexpr int VEC = 16 / ES;
  constexpr int BKB = 128, BK = BKB / ES;
  constexpr int NT = WM * WN * 64, NW = WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int STAGE = (BM + BN) * BKB;
  constexpr int PIECES = (BM + BN) / 8;            // 1-KiB pieces (8 rows) per stage
  static_assert(PIECES % NW == 0, "pieces must split evenly over the waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  int wg = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = wg & 7, j = wg >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int bn = wg % p.nb_n, bm = wg / p.nb_n;
  const int m0 = bm * BM, n0 = bn * BN;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wm = wv % WM, wn = wv / WM;
  const int lr = lane & 31, lh = lane >> 5;

  // per-lane source pointers of this wave's pieces (k advances by one tile per iteration)
  const char* src[PIECES / NW];
  const char* src1[PIECES / NW];                                // rows2 (1x1 conv over two concatenated sources): row pointer into the second source
  int adj[PIECES / NW];
  const int nk = (p.K + BK - 1) / BK, krem = p.K % BK;
  const int kt1 = p.rows2 ? p.c0 / BK : nk;                     // first K-tile of the second source
#pragma unroll
  for (int i = 0; i < PIECES / NW; ++i) {
    const int piece = wv * (PIECES / NW) + i;
    const int row = piece * 8 + (lane >> 3);                    // row inside the stage: [0, BM) = A, [BM, BM+BN) = W
    const int slot = (lane & 7) ^ ((row >> 1) & 7);             // logical 16-byte chunk that lands in physical slot lane & 7
    adj[i] = 0;
    if (row < BM) {
      int m = m0 + row;
      m = m < p.M ? m : p.M - 1;                                // tail rows: any valid row (never stored)
      src[i] = p.x0 + ((size_t)m * p.x0_ld + slot * VEC) * ES;
      src1[i] = src[i];
      if (p.rows2) {                                            // either source may be 2x nearest-upsampled: output pixel -> source pixel
        const int ohow = p.OH * p.OW;
        const int b = m / ohow, r = m - b * ohow;
        const int oy = r / p.OW, ox = r - oy * p.OW;
        const size_t pix0 = ((size_t)b * (p.H >> p.up0) + (oy >> p.up0)) * (size_t)(p.W >> p.up0) + (ox >> p.up0);
        const size_t pix1 = ((size_t)b * (p.H >> p.up1) + (oy >> p.up1)) * (size_t)(p.W >> p.up1) + (ox >> p.up1);
        src[i] = p.x0 + (pix0 * p.x0_ld + slot * VEC) * ES;
        src1[i] = p.x1 + (pix1 * p.x1_ld + slot * VEC) * ES - (size_t)p.c0 * ES;       // indexed by the concat channel
      }
      // K not a multiple of the tile depth: the last tile's out-of-row chunks re-read the previous tile's chunk (finite data
      // against the zero-padded weight columns [K, Kpad)), see gemm256_kernel
      if (krem && slot * VEC >= krem) adj[i] = -BKB;
    } else {
      src[i] = p.w + ((size_t)(n0 + row - BM) * p.Kpad + slot * VEC) * ES;
      src1[i] = src[i];
    }
  }
  auto issue = [&](int stage, int kt) {
#pragma unroll
    for (int i = 0; i < PIECES / NW; ++i) {
      const int piece = wv * (PIECES / NW) + i;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((kt < kt1 ? src[i] : src1[i]) + (long long)kt * BKB + (kt == nk - 1 ? (long long)adj[i] : 0ll)),
                                       (__attribute__((address_space(3))) void*)(smem + stage * STAGE + piece * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addresses: row r of a 32-row group, swizzled slot per k-step
  int a_off[TM], w_off[TN], a_sw[TM], w_sw[TN];
#pragma unroll
  for (int j = 0; j < TM; ++j) { const int row = wm * WTM + j * 32 + lr; a_off[j] = row * BKB; a_sw[j] = (row >> 1) & 7; }
#pragma unroll
  for (int i = 0; i < TN; ++i) { const int row = BM + wn * WTN + i * 32 + lr; w_off[i] = row * BKB; w_sw[i] = (row >> 1) & 7; }

  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // explicit: hipcc puts no vmcnt wait in front of a barrier for LDS-DMA writes
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const char* st = smem + (kt & 1) * STAGE;
    if (kt + 1 < nk) issue((kt + 1) & 1, kt + 1);
#pragma unroll
    for (int s2 = 0; s2 < BKB / 32; ++s2) {
      u32x4 wf[TN], xf[TM];
#pragma unroll
      for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const u32x4*>(st + w_off[i] + (((2 * s2 + lh) ^ w_sw[i]) << 4));
#pragma unroll
      for (int j = 0; j < TM; ++j) xf[j] = *reinterpret_cast<const u32x4*>(st + a_off[j] + (((2 * s2 + lh) ^ a_sw[j]) << 4));
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) Mma<T>::run(wf[i], xf[j], acc[i][j]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces of the next stage (explicit, see above)
    __syncthreads();                               // publishes them and frees the stage just read
  }
  gemm_epilogue<T, TO, BM, BN, WM, WN>(p, acc, smem, m0, n0);
}

template <typename T, typename TO, int BM, int BN, int WM, int WN>
int launch_glds(ConvKArgs& a, hipStream_t stream) {
  constexpr size_t stage = (size_t)2 * (BM + BN) * 128;
  constexpr size_t epi = (size_t)BM * (BN * sizeof(TO) + 16);
  constexpr size_t lds = stage > epi ? stage : epi;
  static bool attr_done = false;
  if (!attr_done && lds > 64 * 1024) {
    CVMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_glds_kernel<T, TO, BM, BN, WM, WN>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  a.nb_n = cdiv(a.N, BN);
  const long long blocks = (long long)cdiv(a.M, BM) * a.nb_n;
  CVMI_CHECK(blocks > 0 && blocks < (1ll << 31), "conv2d: bad grid %lld", blocks);
  cvmi_note_kernel("gemm_glds_kernel<%s, %s, %d, %d, %d, %d>", sizeof(T) == 2 ? CVMI_F16NAME : "float", sizeof(TO) == 2 ? CVMI_F16NAME : "float", BM, BN, WM, WN);
  hipLaunchKernelGGL((gemm_glds_kernel<T, TO, BM, BN, WM, WN>), dim3((unsigned)blocks), dim3(WM * WN * 64), lds, stream, a);
  CVMI_LAUNCH_CHECK();
  return 0;
}

// ---- 256 x 256 tile GEMM: 8 waves, direct-to-LDS staging that stays in flight across barriers -----------------------
// One workgroup per CU (128 KiB of LDS: two 64-KiB stages of X[256][64] + W[256][64] f16).  Waves form a 2 (pixel
// halves) x 4 (channel quarters) grid, wave tile 128 px x 64 ch = 4 x 2 MFMA tiles of 32 x 32.  A K-tile (64 deep)
// is computed in 4 phases of 8 MFMAs; each phase reads only the fragments it newly needs:
//     phase 1: W c0, X p0 p1 -> (c0;p0,p1)     phase 2: W c1 -> (c1;p0,p1)
//     phase 3: X p2 p3       -> (c1;p2,p3)     phase 4: --   -> (c0;p2,p3)
// The stage is DMA'd (global_load_lds, 16 B per lane, XOR-swizzled source slots as in gemm_glds_kernel) in four
// 128-row "half-tiles" named after the phase that consumes them: XA (p0 p1 of both pixel halves), WA (c0 of the
// four channel quarters), WB (c1), XB (p2 p3).  One half-tile (2 DMA instructions per lane) is issued per phase,
// into a region whose last reader finished at least one full phase earlier:
//     phase (t,1): XB(t+1)   (t,2): XA(t+2)   (t,3): WA(t+2)   (t,4): WB(t+2)
// and ONE counted wait per K-tile, `s_waitcnt vmcnt(6)` in phase 4 (everything but the three half-tiles just issued
// has landed, i.e. all of tile t+1), followed by that phase's barriers, orders the DMA before the reads of (t+1,1).
// Every phase is  [ds_reads; DMA issue; lgkmcnt(0)] s_barrier [8 MFMAs] s_barrier ; the two pixel halves (one wave of
// each per SIMD) run offset by one barrier, so one wave's LDS reads overlap the other's MFMAs.
// Hazards (both wave groups, group 1 one barrier behind; barrier instances numbered globally):
//   RAW  every wave waits vmcnt before its first barrier of (t,4); readers of (t+1,1) have passed the barrier after it.
//   WAR  reads are retired (lgkmcnt(0)) BEFORE the phase's first barrier, so a region read in phase p is free once
//        every wave is past that barrier: any wave issuing phase p+1's DMA is.
// IM2COL (r04): the A operand is a 3 x 3 convolution's im2col matrix instead of a plain row-major one -- YOLO11-l's large stride-2 / stride-1
// convs (256 - 512 channels; 250 - 375 TFLOP/s on the register-staged 128 x 128 kernel above).  K runs tap-major (k = tap * Cin + c, as the
// weights are packed), Cin is a power-of-two multiple of 64, so a K-tile of 64 channels lies inside ONE tap: the lane's source row pointer
// (the output pixel's top-left input pixel, computed once) moves by a wave-uniform tap offset per K-tile, and a lane whose tap falls outside
// the image (zero padding) points its DMA at 16 zero bytes instead -- one bit test and one select per DMA instruction, nothing else changes:
// W rows, LDS image, swizzle, phases, counted waits and the epilogue are those of the plain GEMM.
template <typename TO, bool STAGGER, bool IM2COL = false>
__global__ __launch_bounds__(512, 1) void gemm256_kernel(const ConvKArgs p) {
  using T = f16;
  constexpr int BM = 256, BN = 256, BKB = 128;
  constexpr int STAGE = (BM + BN) * BKB;                  // 64 KiB
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  int wg = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = wg & 7, j = wg >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int bn = wg % p.nb_n, bm = wg / p.nb_n;
  const int m0 = bm * BM, n0 = bn * BN;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wr = wv >> 2, wc = wv & 3;                    // waves wv and wv + 4 share a SIMD: one of each pixel half
  const int lr = lane & 31, lh = lane >> 5;

  // DMA sources / LDS destinations: half-tile h (0 XA, 1 WA, 2 WB, 3 XB) = 16 pieces of 8 rows, 2 per wave
  const int nk = (p.K + 63) / 64, krem = p.K & 63;
  const char* src[4][2];
  int dst[4][2], adj[4][2], vmask[4][2];
#pragma unroll
  for (int h = 0; h < 4; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = wv * 2 + i;
      const int row0 = (h == 0 || h == 3) ? (q >> 3) * 128 + (h == 3 ? 64 : 0) + (q & 7) * 8
                                          : BM + (q >> 2) * 64 + (h == 2 ? 32 : 0) + (q & 3) * 8;
      const int row = row0 + (lane >> 3);
      const int slot = (lane & 7) ^ ((row >> 1) & 7);
      adj[h][i] = 0;
      vmask[h][i] = 0;
      if (row < BM) {
        int m = m0 + row;
        m = m < p.M ? m : p.M - 1;
        src[h][i] = p.x0 + ((size_t)m * p.x0_ld + slot * 8) * 2;
        if constexpr (IM2COL) {                              // row m = output pixel (b, oy, ox): pointer to its tap (0, 0) input pixel + which taps exist
          const int ohow = p.OH * p.OW;
          const int b = m / ohow, r = m - b * ohow, oy = r / p.OW, ox = r - oy * p.OW;
          const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
          src[h][i] = p.x0 + ((((long long)b * p.H + iy0) * p.W + ix0) * p.x0_ld + slot * 8) * 2;
          int vm = 0;
#pragma unroll
          for (int t = 0; t < 9; ++t)
            if ((unsigned)(iy0 + t / 3) < (unsigned)p.H && (unsigned)(ix0 + t % 3) < (unsigned)p.W) vm |= 1 << t;
          vmask[h][i] = vm;
        }
        // K not a multiple of 64: in the LAST K-tile the chunks at columns >= K would leave the pixel row (and, for the
        // last row, the tensor); they re-read the previous tile's chunk instead -- any finite value will do, the weight
        // columns [K, Kpad) are zero
        if (krem && slot * 8 >= krem) adj[h][i] = -BKB;
      } else {
        int n = n0 + row - BM;
        n = n < p.N ? n : p.N - 1;
        src[h][i] = p.w + ((size_t)n * p.Kpad + slot * 8) * 2;
      }
      dst[h][i] = row0 * BKB;
    }
#define G256_ISSUE(h, stage, kt)                                                                                         \
  do {                                                                                                                   \
    const long long last_ = (kt) == nk - 1 ? 1 : 0;                                                                      \
    const char *s0_ = src[h][0] + (long long)(kt) * BKB + last_ * adj[h][0], *s1_ = src[h][1] + (long long)(kt) * BKB + last_ * adj[h][1]; \
    if (IM2COL && ((h) == 0 || (h) == 3)) {                 /* X half-tiles: tap offset (wave-uniform), zero page outside the image */ \
      const int tap_ = (kt) >> p.im2col_shift, c64_ = (kt) & ((1 << p.im2col_shift) - 1);                               \
      const int ky_ = (tap_ * 11) >> 5, kx_ = tap_ - 3 * ky_;                                                            \
      const long long off_ = ((long long)(ky_ * p.W + kx_) * p.x0_ld + c64_ * 64) * 2;                                   \
      const char* z_ = reinterpret_cast<const char*>(cvmi_zero_page) + (lane & 7) * 16;                                  \
      s0_ = ((vmask[h][0] >> tap_) & 1) ? src[h][0] + off_ : z_;                                                          \
      s1_ = ((vmask[h][1] >> tap_) & 1) ? src[h][1] + off_ : z_;                                                          \
    }                                                                                                                    \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s0_,                                 \
                                     (__attribute__((address_space(3))) void*)(smem + (stage) * STAGE + dst[h][0]), 16, 0, 0); \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s1_,                                 \
                                     (__attribute__((address_space(3))) void*)(smem + (stage) * STAGE + dst[h][1]), 16, 0, 0); \
  } while (0)

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addresses: rows are (32-aligned base + lr), so one swizzle value serves every fragment of the lane
  const int sw = (lr >> 1) & 7;
  int ko[4];
#pragma unroll
  for (int s2 = 0; s2 < 4; ++s2) ko[s2] = ((2 * s2 + lh) ^ sw) << 4;
  const int xbase = (wr * 128 + lr) * BKB, wbase = (BM + wc * 64 + lr) * BKB;

  // prologue: all of tile 0, and XA WA WB of tile 1
  G256_ISSUE(0, 0, 0); G256_ISSUE(1, 0, 0); G256_ISSUE(2, 0, 0); G256_ISSUE(3, 0, 0);
  if (nk > 1) {
    G256_ISSUE(0, 1, 1); G256_ISSUE(1, 1, 1); G256_ISSUE(2, 1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (STAGGER && wr == 1) __builtin_amdgcn_s_barrier();

  u32x4 xf[2][4], wf[2][4];
#define G256_READ_X(slot_, j)                                                                                            \
  _Pragma("unroll") for (int s2 = 0; s2 < 4; ++s2)                                                                       \
      xf[slot_][s2] = *reinterpret_cast<const u32x4*>(st + xbase + (j) * 32 * BKB + ko[s2])
#define G256_READ_W(i)                                                                                                   \
  _Pragma("unroll") for (int s2 = 0; s2 < 4; ++s2)                                                                       \
      wf[i][s2] = *reinterpret_cast<const u32x4*>(st + wbase + (i) * 32 * BKB + ko[s2])
#define G256_SYNC_IN()                                                                                                   \
  do {                                                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
    __builtin_amdgcn_s_barrier();                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
    __builtin_amdgcn_s_setprio(1);                                                                                       \
  } while (0)
#define G256_SYNC_OUT()                                                                                                  \
  do {                                                                                                                   \
    __builtin_amdgcn_s_setprio(0);                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
    __builtin_amdgcn_s_barrier();                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
  } while (0)
  // the empty asm statements pin the MFMAs between the two barriers of their phase: hipcc otherwise sinks the (pure)
  // MFMA calls below the second barrier, in among the next phase's loads
#define G256_MMA(i, j0)                                                                                                  \
  asm volatile("" : "+v"(wf[i][0]), "+v"(wf[i][1]), "+v"(wf[i][2]), "+v"(wf[i][3]));                                     \
  _Pragma("unroll") for (int s2 = 0; s2 < 4; ++s2) {                                                                     \
    Mma<T>::run(wf[i][s2], xf[0][s2], acc[i][j0]);                                                                       \
    Mma<T>::run(wf[i][s2], xf[1][s2], acc[i][(j0) + 1]);                                                                 \
  }                                                                                                                      \
  asm volatile("" : "+v"(acc[i][j0]), "+v"(acc[i][(j0) + 1]))

  for (int kt = 0; kt < nk; ++kt) {
    const int b = kt & 1;
    const char* st = smem + b * STAGE;
    const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;     // wave-uniform
    // ---- phase 1
    G256_READ_W(0);
    G256_READ_X(0, 0);
    G256_READ_X(1, 1);
    if (more1) G256_ISSUE(3, b ^ 1, kt + 1);
    G256_SYNC_IN();
    G256_MMA(0, 0);
    G256_SYNC_OUT();
    // ---- phase 2
    G256_READ_W(1);
    if (more2) G256_ISSUE(0, b, kt + 2);
    G256_SYNC_IN();
    G256_MMA(1, 0);
    G256_SYNC_OUT();
    // ---- phase 3
    G256_READ_X(0, 2);
    G256_READ_X(1, 3);
    if (more2) G256_ISSUE(1, b, kt + 2);
    G256_SYNC_IN();
    G256_MMA(1, 2);
    G256_SYNC_OUT();
    // ---- phase 4
    if (more2) {
      G256_ISSUE(2, b, kt + 2);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else if (more1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    G256_SYNC_IN();
    G256_MMA(0, 2);
    G256_SYNC_OUT();
  }
  if (STAGGER && wr == 0) __builtin_amdgcn_s_barrier();
#undef G256_ISSUE
#undef G256_READ_X
#undef G256_READ_W
#undef G256_SYNC_IN
#undef G256_SYNC_OUT
#undef G256_MMA

  // ---- epilogue: bias + act -> LDS tile -> 16-byte stores (+ residual); f32 output goes in two 128-column passes
  constexpr int OES = sizeof(TO), OVEC = 16 / OES;
  constexpr int NPASS = OES == 4 ? 2 : 1, CW = BN / NPASS;
  constexpr int CROWB = CW * OES + 16;
  constexpr int NCH = CW / OVEC;
  char* const Ct = smem;
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
    if (NPASS == 1 || (wc >> 1) == pass) {
      const int cbase = NPASS == 1 ? wc * 64 : (wc & 1) * 64;
      with_act<true>(p.act_after_res ? CVMI_ACT_NONE : p.act, [&](auto actf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int nl = cbase + i * 32 + 8 * q + 4 * lh;
            const int n = n0 + pass * CW + nl;
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (n < p.N) bv = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int ml = wr * 128 + j * 32 + lr;
              float v[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = actf(acc[i][j][4 * q + e] + bv[e]);
              char* d = Ct + ml * CROWB + nl * OES;
              if constexpr (OES == 2) {
                f16x4 hv = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                *reinterpret_cast<f16x4*>(d) = hv;
              } else {
                f32x4 fv = {v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4*>(d) = fv;
              }
            }
          }
        }
      });
    }
    __syncthreads();
    for (int idx = tid; idx < BM * NCH; idx += 512) {
      const int row = idx / NCH, ch = idx - row * NCH;
      const int m = m0 + row, n = n0 + pass * CW + ch * OVEC;
      if (m >= p.M || n >= p.N) continue;
      u32x4 cv = *reinterpret_cast<const u32x4*>(Ct + row * CROWB + ch * 16);
      const size_t rpix = p.res_mod > 0 ? (size_t)(m % p.res_mod) : (size_t)m;
      if (p.res || p.act_after_res) {
        float a[OVEC];
        unpack16<TO>(cv, a);
        if (p.res) {
          float r[OVEC];
          unpack16<TO>(*reinterpret_cast<const u32x4*>(p.res + (rpix * p.res_ld + n) * OES), r);
#pragma unroll
          for (int e = 0; e < OVEC; ++e) a[e] += r[e];
        }
        if (p.act_after_res) {
#pragma unroll
          for (int e = 0; e < OVEC; ++e) a[e] = act_apply<true>(a[e], p.act);
        }
        cv = pack16<TO>(a);
      }
      *reinterpret_cast<u32x4*>(p.y + ((size_t)m * p.y_ld + n) * OES) = cv;
    }
    if (pass + 1 < NPASS) __syncthreads();
  }
}

// ---- persistent form of gemm256_kernel for fp16 output without residual (Hiera qkv / fc1) ------------------------------
// One workgroup per CU loops over its tiles (XCD-contiguous); the prologue DMA of tile i + 1 (14 instructions per lane into
// the two stage buffers, which are dead once every wave has left the K-loop) is issued BEFORE the epilogue of tile i, so the
// HBM / L2 latency of a tile's first K-tiles hides behind the previous tile's bias / activation / store work.  For that the
// epilogue must not touch the stage buffers: each wave transposes its 128 x 64 sub-tile through a private 4 KiB scratch
// (32 pixels x 64 channels per step, XOR-swizzled 16-byte chunks) -- no workgroup barrier, and no ordinary global load while
// the DMA is in flight (hipcc would drain the queue at its use): the wave's 64 bias values are parked in its scratch at the
// start of the tile.  K-loop, hazards and counted waits are those of gemm256_kernel (staggered groups).
__global__ __launch_bounds__(512, 1) void gemm256p_kernel(const ConvKArgs p, int ntiles) {
  using T = f16;
  constexpr int BM = 256, BN = 256, BKB = 128;
  constexpr int STAGE = (BM + BN) * BKB;                  // 64 KiB
  constexpr int SCR_OFF = 2 * STAGE, SCR_WAVE = 2048, BIAS_OFF = SCR_OFF + 8 * SCR_WAVE;   // 32 px x 32 ch scratch + 64 bias floats per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wr = wv >> 2, wc = wv & 3;
  const int lr = lane & 31, lh = lane >> 5;
  char* const scr = smem + SCR_OFF + wv * SCR_WAVE;
  char* const bsc = smem + BIAS_OFF + wv * 256;

  // tiles of this workgroup: XCD x (= blockIdx & 7) owns a contiguous run; its workgroups stride through it
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, wpx = gridDim.x >> 3;
  const int tq = ntiles >> 3, trm = ntiles & 7;
  const int tbase = xcd * tq + (xcd < trm ? xcd : trm), tcnt = tq + (xcd < trm ? 1 : 0);

  const int nk = (p.K + 63) / 64, krem = p.K & 63;
  const char* src[4][2];
  int dst[4][2], adj[4][2];
#pragma unroll
  for (int h = 0; h < 4; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = wv * 2 + i;
      dst[h][i] = ((h == 0 || h == 3) ? (q >> 3) * 128 + (h == 3 ? 64 : 0) + (q & 7) * 8
                                       : BM + (q >> 2) * 64 + (h == 2 ? 32 : 0) + (q & 3) * 8) * BKB;
    }
  auto setup = [&](int tile, int& m0, int& n0) {
    const int bn = tile % p.nb_n, bm = tile / p.nb_n;
    m0 = bm * BM; n0 = bn * BN;
    int ln = lane;
    asm volatile("" : "+v"(ln));                           // opaque: the per-lane row / slot terms are recomputed per tile, not hoisted (and spilled)
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = dst[h][i] / BKB + (ln >> 3);
        const int slot_ = (ln & 7) ^ ((row >> 1) & 7);
        adj[h][i] = 0;
        if (row < BM) {
          int m = m0 + row;
          m = m < p.M ? m : p.M - 1;
          src[h][i] = p.x0 + ((size_t)m * p.x0_ld + slot_ * 8) * 2;
          if (krem && slot_ * 8 >= krem) adj[h][i] = -BKB;
        } else {
          int n = n0 + row - BM;
          n = n < p.N ? n : p.N - 1;
          src[h][i] = p.w + ((size_t)n * p.Kpad + slot_ * 8) * 2;
        }
      }
  };
#define G256_ISSUE(h, stage, kt)                                                                                         \
  do {                                                                                                                   \
    const long long last_ = (kt) == nk - 1 ? 1 : 0;                                                                      \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[h][0] + (long long)(kt) * BKB + last_ * adj[h][0]), \
                                     (__attribute__((address_space(3))) void*)(smem + (stage) * STAGE + dst[h][0]), 16, 0, 0); \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[h][1] + (long long)(kt) * BKB + last_ * adj[h][1]), \
                                     (__attribute__((address_space(3))) void*)(smem + (stage) * STAGE + dst[h][1]), 16, 0, 0); \
  } while (0)
#define G256_PROLOGUE()                                                                                                  \
  do {                                                                                                                   \
    G256_ISSUE(0, 0, 0); G256_ISSUE(1, 0, 0); G256_ISSUE(2, 0, 0); G256_ISSUE(3, 0, 0);                                  \
    if (nk > 1) { G256_ISSUE(0, 1, 1); G256_ISSUE(1, 1, 1); G256_ISSUE(2, 1, 1); }                                       \
  } while (0)

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int sw = (lr >> 1) & 7;
  int ko[4];
#pragma unroll
  for (int s2 = 0; s2 < 4; ++s2) ko[s2] = ((2 * s2 + lh) ^ sw) << 4;
  const int xbase = (wr * 128 + lr) * BKB, wbase = (BM + wc * 64 + lr) * BKB;

  u32x4 xf[2][4], wf[2][4];
#define G256_READ_X(slot_, j)                                                                                            \
  _Pragma("unroll") for (int s2 = 0; s2 < 4; ++s2)                                                                       \
      xf[slot_][s2] = *reinterpret_cast<const u32x4*>(st + xbase + (j) * 32 * BKB + ko[s2])
#define G256_READ_W(i)                                                                                                   \
  _Pragma("unroll") for (int s2 = 0; s2 < 4; ++s2)                                                                       \
      wf[i][s2] = *reinterpret_cast<const u32x4*>(st + wbase + (i) * 32 * BKB + ko[s2])
#define G256_SYNC_IN()                                                                                                   \
  do {                                                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
    __builtin_amdgcn_s_barrier();                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
    __builtin_amdgcn_s_setprio(1);                                                                                       \
  } while (0)
#define G256_SYNC_OUT()                                                                                                  \
  do {                                                                                                                   \
    __builtin_amdgcn_s_setprio(0);                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
    __builtin_amdgcn_s_barrier();                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
  } while (0)
#define G256_MMA(i, j0)                                                                                                  \
  asm volatile("" : "+v"(wf[i][0]), "+v"(wf[i][1]), "+v"(wf[i][2]), "+v"(wf[i][3]));                                     \
  _Pragma("unroll") for (int s2 = 0; s2 < 4; ++s2) {                                                                     \
    Mma<T>::run(wf[i][s2], xf[0][s2], acc[i][j0]);                                                                       \
    Mma<T>::run(wf[i][s2], xf[1][s2], acc[i][(j0) + 1]);                                                                 \
  }                                                                                                                      \
  asm volatile("" : "+v"(acc[i][j0]), "+v"(acc[i][(j0) + 1]))

  int m0 = 0, n0 = 0;
  if (slot < tcnt) { setup(tbase + slot, m0, n0); G256_PROLOGUE(); }
  for (int ti = slot; ti < tcnt; ti += wpx) {
    const int cm0 = m0, cn0 = n0;
    // ---- this wave's 64 bias values -> its scratch (drains the prologue DMA: it was issued an epilogue ago)
    if (lane < 16) {
      const int n = cn0 + wc * 64 + lane * 4;
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (n < p.N) bv = *reinterpret_cast<const f32x4*>(p.bias + n);
      *reinterpret_cast<f32x4*>(bsc + lane * 16) = bv;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();

    for (int kt = 0; kt < nk; ++kt) {
      const int b = kt & 1;
      const char* st = smem + b * STAGE;
      const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
      G256_READ_W(0);
      G256_READ_X(0, 0);
      G256_READ_X(1, 1);
      if (more1) G256_ISSUE(3, b ^ 1, kt + 1);
      G256_SYNC_IN();
      G256_MMA(0, 0);
      G256_SYNC_OUT();
      G256_READ_W(1);
      if (more2) G256_ISSUE(0, b, kt + 2);
      G256_SYNC_IN();
      G256_MMA(1, 0);
      G256_SYNC_OUT();
      G256_READ_X(0, 2);
      G256_READ_X(1, 3);
      if (more2) G256_ISSUE(1, b, kt + 2);
      G256_SYNC_IN();
      G256_MMA(1, 2);
      G256_SYNC_OUT();
      if (more2) {
        G256_ISSUE(2, b, kt + 2);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      } else if (more1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      G256_SYNC_IN();
      G256_MMA(0, 2);
      G256_SYNC_OUT();
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();            // groups aligned: every wave has left the K-loop, both stages are dead

    // ---- next tile's prologue flies during this tile's epilogue
    if (ti + wpx < tcnt) { setup(tbase + ti + wpx, m0, n0); G256_PROLOGUE(); }

    // ---- wave-private epilogue: 32 pixels x 32 channels per step through the scratch (64-byte rows, swizzled 16-byte chunks)
    // (one base register per kind of access + immediate offsets: anything the compiler would spill here comes back through
    //  a scratch load, whose vmcnt(0) drains the prologue in flight)
    int le = lane;
    asm volatile("" : "+v"(le));                           // opaque copy: the epilogue's address terms are built here, after the K-loop
    const int lre = le & 31, lhe = le >> 5;
    const unsigned ba_base = (unsigned)(BIAS_OFF + wv * 256 + lhe * 16);
    const unsigned sa_base = (unsigned)(SCR_OFF + wv * SCR_WAVE + (le >> 2) * 64 + (((le & 3) ^ ((le >> 2) & 3)) << 4));
    char* const wr_base = scr + lre * 64 + lhe * 8;
    const int wsw = lre & 3;
    with_act<true>(p.act, [&](auto actf) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          // the 4 x 4 bias values of this channel tile: four reads, ONE wait (each inline-asm read + wait is a full LDS round trip)
          f32x4 b0, b1, b2, b3;
          asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\t"
                       "ds_read_b128 %3, %4 offset:%8\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3)
                       : "v"(ba_base), "n"(i * 128), "n"(i * 128 + 32), "n"(i * 128 + 64), "n"(i * 128 + 96) : "memory");
          __builtin_amdgcn_sched_barrier(0);
          const f32x4 bq[4] = {b0, b1, b2, b3};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f16x4 hv;
#pragma unroll
            for (int e = 0; e < 4; ++e) hv[e] = (f16)actf(acc[i][j][4 * q + e] + bq[q][e]);
            *reinterpret_cast<f16x4*>(wr_base + ((q ^ wsw) << 4)) = hv;
          }
          u32x4 v0, v1;
          asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(v0), "=&v"(v1) : "v"(sa_base) : "memory");
          __builtin_amdgcn_sched_barrier(0);
          const u32x4 vv[2] = {v0, v1};
#pragma unroll
          for (int it = 0; it < 2; ++it) {
            const int c = it * 64 + le, row = c >> 2, ch = c & 3;
            const int m = cm0 + wr * 128 + j * 32 + row, n = cn0 + wc * 64 + i * 32 + ch * 8;
            if (m < p.M && n < p.N) *reinterpret_cast<u32x4*>(p.y + ((size_t)m * p.y_ld + n) * 2) = vv[it];
          }
        }
    });
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }
#undef G256_ISSUE
#undef G256_PROLOGUE
#undef G256_READ_X
#undef G256_READ_W
#undef G256_SYNC_IN
#undef G256_SYNC_OUT
#undef G256_MMA
}

int launch_g256p(ConvKArgs& a, hipStream_t stream) {
  constexpr int bytes = 2 * 512 * 128 + 8 * 2048 + 8 * 256;   // two stages + eight 2 KiB wave scratches + eight 64-float bias rows
  static bool attr_done = false;
  if (!attr_done) {
    CVMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256p_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    attr_done = true;
  }
  a.nb_n = cdiv(a.N, 256);
  const long long tiles = (long long)cdiv(a.M, 256) * a.nb_n;
  CVMI_CHECK(tiles >= 8 && tiles < (1ll << 31), "conv2d: bad grid %lld", tiles);
  const unsigned grid = tiles >= 256 ? 256u : (unsigned)(tiles / 8 * 8);
  cvmi_note_kernel("gemm256p_kernel");
  hipLaunchKernelGGL(gemm256p_kernel, dim3(grid), dim3(512), bytes, stream, a, (int)tiles);
  CVMI_LAUNCH_CHECK();
  return 0;
}

// ---- 256 x 192 tile variant of gemm256_kernel (N = 576 = 3 x 192: Hiera-L stage-3 proj / fc2) -------------------------
// 8 waves as 4 (pixel quarters) x 2 (channel halves; the two staggered groups), wave tile 64 px x 96 ch = 2 x 3 MFMA
// tiles.  A K-tile is three phases of 8 MFMAs:  1: W c0, X p0 p1 -> (c0;p0,p1)   2: W c1 -> (c1;..)   3: W c2 -> (c2;..).
// DMA units per K-tile (56 KiB): X0, X1 (128 pixel rows each, 2 instructions per lane), WC0, WC1, WC2 (the 32-row channel
// block k of both halves, 1 instruction per lane), issued one phase after their region's last read:
//     phase (t,1): WC2(t+1)      (t,2): X0(t+2), WC0(t+2)      (t,3): X1(t+2), WC1(t+2)
// and one counted wait per K-tile, vmcnt(6) in phase 3 (all but the six instructions of phases 2 and 3, i.e. all of tile
// t+1, has landed).  Hazard argument as for gemm256_kernel (reads retired before a phase's first barrier).
// M16 = true: the same tile, DMA schedule and barriers on v_mfma_f32_16x16x32 (16 per phase instead of 8 of 32x32x16: equal cycles per
// flop, equal LDS bytes -- every lane still reads 16 bytes per fragment -- but the chip holds a higher clock on this shape under the
// power limit: MI355X_MICROARCH.md, DVFS give-back item 7).  Fragment (16 rows x 32 k): lane (r16 = lane & 15, g = lane >> 4) reads chunk
// 4 ks + g of row r16; with the source swizzle slot = chunk ^ ((row >> 1) & 7) a 16-lane ds_read_b128 group still covers the 16 slots of
// a 256-byte bank period once.  C / D: lane holds channels 4 g .. 4 g + 3 (rows, from the weight operand) of pixel r16.
template <typename TO, bool M16 = false>
__global__ __launch_bounds__(512, 1) void gemm256x192_kernel(const ConvKArgs p) {
  using T = f16;
  constexpr int BM = 256, BN = 192, BKB = 128;
  constexpr int STAGE = (BM + BN) * BKB;                  // 56 KiB
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  int wg = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = wg & 7, j = wg >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int bn = wg % p.nb_n;
  const int bm = p.rev_m ? (int)(gridDim.x / p.nb_n) - 1 - wg / p.nb_n : wg / p.nb_n;
  const int m0 = bm * BM, n0 = bn * BN;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wp = wv & 3, wh = wv >> 2;                    // pixel quarter, channel half (= stagger group; wv and wv + 4 share a SIMD)
  const int lr = lane & 31, lh = lane >> 5;

  // DMA: unit 0 X0, 1 X1 (pieces 2 wv, 2 wv + 1 of 16), 2..4 WC0..WC2 (piece wv of 8: half wv >> 2, 8-row group wv & 3)
  const int nk = (p.K + 63) / 64, krem = p.K & 63;
  const char* src[7];
  int dst[7], adj[7];
#pragma unroll
  for (int u = 0; u < 7; ++u) {
    int row0;
    if (u < 4) row0 = (u >> 1) * 128 + (wv * 2 + (u & 1)) * 8;                    // u = 0,1: X0; 2,3: X1
    else row0 = BM + (wv >> 2) * 96 + (u - 4) * 32 + (wv & 3) * 8;               // u = 4,5,6: WC0..2
    const int row = row0 + (lane >> 3);
    const int slot = (lane & 7) ^ ((row >> 1) & 7);
    adj[u] = 0;
    if (row < BM) {
      int m = m0 + row;
      m = m < p.M ? m : p.M - 1;
      src[u] = p.x0 + ((size_t)m * p.x0_ld + slot * 8) * 2;
      if (krem && slot * 8 >= krem) adj[u] = -BKB;          // see gemm256_kernel
    } else {
      int n = n0 + row - BM;
      n = n < p.N ? n : p.N - 1;
      src[u] = p.w + ((size_t)n * p.Kpad + slot * 8) * 2;
    }
    dst[u] = row0 * BKB;
  }
#define G192_ISSUE1(u, stage, kt)                                                                                        \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[u] + (long long)(kt) * BKB + ((kt) == nk - 1 ? (long long)adj[u] : 0ll)), \
                                   (__attribute__((address_space(3))) void*)(smem + (stage) * STAGE + dst[u]), 16, 0, 0)
#define G192_X0(stage, kt) do { G192_ISSUE1(0, stage, kt); G192_ISSUE1(1, stage, kt); } while (0)
#define G192_X1(stage, kt) do { G192_ISSUE1(2, stage, kt); G192_ISSUE1(3, stage, kt); } while (0)
#define G192_WC(k, stage, kt) G192_ISSUE1(4 + (k), stage, kt)

  f32x16 acc[3][2];                                       // 32x32x16: [channel block][pixel block]
  f32x4 acc16[3][2][4];                                   // 16x16x32: [channel block][16-channel half][16-pixel block]
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if constexpr (M16) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc16[i][j][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      }
    }

  const int r16 = lane & 15, g16 = lane >> 4;
  const int sw = M16 ? (r16 >> 1) & 7 : (lr >> 1) & 7;
  int ko[4];
#pragma unroll
  for (int s2 = 0; s2 < 4; ++s2) ko[s2] = M16 ? (((4 * (s2 & 1) + g16) ^ sw) << 4) + (s2 >> 1) * 16 * BKB      // s2 = ks + 2 * (16-row half)
                                             : ((2 * s2 + lh) ^ sw) << 4;
  const int xbase = (wp * 64 + (M16 ? r16 : lr)) * BKB, wbase = (BM + wh * 96 + (M16 ? r16 : lr)) * BKB;

  G192_X0(0, 0); G192_WC(0, 0, 0); G192_X1(0, 0); G192_WC(1, 0, 0); G192_WC(2, 0, 0);
  if (nk > 1) {
    G192_X0(1, 1); G192_WC(0, 1, 1); G192_X1(1, 1); G192_WC(1, 1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (wh == 1) __builtin_amdgcn_s_barrier();

  u32x4 xf[2][4], wf[4];
#define G192_READ_X(j)                                                                                                   \
  _Pragma("unroll") for (int s2 = 0; s2 < 4; ++s2)                                                                       \
      xf[j][s2] = *reinterpret_cast<const u32x4*>(st + xbase + (j) * 32 * BKB + ko[s2])
#define G192_READ_W(i)                                                                                                   \
  _Pragma("unroll") for (int s2 = 0; s2 < 4; ++s2)                                                                       \
      wf[s2] = *reinterpret_cast<const u32x4*>(st + wbase + (i) * 32 * BKB + ko[s2])
#define G192_SYNC_IN()                                                                                                   \
  do {                                                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
    __builtin_amdgcn_s_barrier();                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
    __builtin_amdgcn_s_setprio(1);                                                                                       \
  } while (0)
#define G192_SYNC_OUT()                                                                                                  \
  do {                                                                                                                   \
    __builtin_amdgcn_s_setprio(0);                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
    __builtin_amdgcn_s_barrier();                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
  } while (0)
  // M16: wf[ks + 2 hh] = channels 16 hh .. of block i, k-step ks; xf[j][ks + 2 ph] = pixels 32 j + 16 ph .., k-step ks
#define G192_MMA(i)                                                                                                      \
  asm volatile("" : "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]));                                                 \
  if constexpr (M16) {                                                                                                   \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                     \
    _Pragma("unroll") for (int hh = 0; hh < 2; ++hh)                                                                     \
    _Pragma("unroll") for (int pq = 0; pq < 4; ++pq)                                                                     \
      acc16[i][hh][pq] = CVMI_MFMA_16X16X32(__builtin_bit_cast(f16x8, wf[ks + 2 * hh]),                                  \
                                            __builtin_bit_cast(f16x8, xf[pq >> 1][ks + 2 * (pq & 1)]), acc16[i][hh][pq], 0, 0, 0); \
    asm volatile("" : "+v"(acc16[i][0][0]), "+v"(acc16[i][0][1]), "+v"(acc16[i][0][2]), "+v"(acc16[i][0][3]),            \
                      "+v"(acc16[i][1][0]), "+v"(acc16[i][1][1]), "+v"(acc16[i][1][2]), "+v"(acc16[i][1][3]));           \
  } else {                                                                                                               \
    _Pragma("unroll") for (int s2 = 0; s2 < 4; ++s2) {                                                                   \
      Mma<T>::run(wf[s2], xf[0][s2], acc[i][0]);                                                                         \
      Mma<T>::run(wf[s2], xf[1][s2], acc[i][1]);                                                                         \
    }                                                                                                                    \
    asm volatile("" : "+v"(acc[i][0]), "+v"(acc[i][1]));                                                                 \
  }

  for (int kt = 0; kt < nk; ++kt) {
    const int b = kt & 1;
    const char* st = smem + b * STAGE;
    const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
    // ---- phase 1
    G192_READ_W(0);
    G192_READ_X(0);
    G192_READ_X(1);
    if (more1) G192_WC(2, b ^ 1, kt + 1);
    G192_SYNC_IN();
    G192_MMA(0)
    G192_SYNC_OUT();
    // ---- phase 2
    G192_READ_W(1);
    if (more2) { G192_X0(b, kt + 2); G192_WC(0, b, kt + 2); }
    G192_SYNC_IN();
    G192_MMA(1)
    G192_SYNC_OUT();
    // ---- phase 3
    G192_READ_W(2);
    if (more2) {
      G192_X1(b, kt + 2); G192_WC(1, b, kt + 2);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else if (more1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    G192_SYNC_IN();
    G192_MMA(2)
    G192_SYNC_OUT();
  }
  if (wh == 0) __builtin_amdgcn_s_barrier();
#undef G192_ISSUE1
#undef G192_X0
#undef G192_X1
#undef G192_WC
#undef G192_READ_X
#undef G192_READ_W
#undef G192_SYNC_IN
#undef G192_SYNC_OUT
#undef G192_MMA

  // ---- M16, f32 output + residual on a full tile (Hiera stage-3 fc2, the dominant launch): NO LDS transposition, no barriers.  In the
  //      16x16x32 C / D layout a lane holds 4 consecutive channels of a pixel and the four lanes g = 0..3 hold 16 consecutive ones:
  //      residual loads and stores are 64-byte row pieces as they stand.  Row statistics for the next LayerNorm (p.stats): per pixel and
  //      96-channel slice (= this wave's channel half) the slice MEAN and the sum of squared deviations from it, two passes over the 24
  //      values in registers -- no E[x^2] - E[x]^2 cancellation however far the row sits from zero; the consumer combines the
  //      slices as Chan et al. do (tok_linear16.hip / tok_linear.hip, stats_parts).
  if constexpr (M16 && sizeof(TO) == 4) {
    if (p.res && p.act == CVMI_ACT_NONE && !p.act_after_res && p.res_mod == 0 && m0 + BM <= p.M && n0 + BN <= p.N) {      // (uniform)
      const int slices = p.N / 96;
      f32x4 bv[3][2];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) bv[i][hh] = *reinterpret_cast<const f32x4*>(p.bias + n0 + wh * 96 + i * 32 + hh * 16 + 4 * g16);
#pragma unroll
      for (int pq = 0; pq < 4; ++pq) {
        const size_t m = (size_t)m0 + wp * 64 + pq * 16 + r16;
        const float* rrow = reinterpret_cast<const float*>(p.res) + m * p.res_ld + n0 + wh * 96 + 4 * g16;
        float* yrow = reinterpret_cast<float*>(p.y) + m * p.y_ld + n0 + wh * 96 + 4 * g16;
        f32x4 v[3][2];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) v[i][hh] = *reinterpret_cast<const f32x4*>(rrow + i * 32 + hh * 16);
        float sm = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[i][hh][e] = (acc16[i][hh][pq][e] + bv[i][hh][e]) + v[i][hh][e];
            *reinterpret_cast<f32x4*>(yrow + i * 32 + hh * 16) = v[i][hh];
            sm += (v[i][hh][0] + v[i][hh][1]) + (v[i][hh][2] + v[i][hh][3]);
          }
        if (p.stats) {                                          // (uniform)
          sm += __shfl_xor(sm, 16);
          sm += __shfl_xor(sm, 32);
          const float mean = sm * (1.0f / 96.0f);
          float m2 = 0.f;
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
              for (int e = 0; e < 4; ++e) { const float dv = v[i][hh][e] - mean; m2 = fmaf(dv, dv, m2); }
          m2 += __shfl_xor(m2, 16);
          m2 += __shfl_xor(m2, 32);
          if (g16 == 0) *reinterpret_cast<float2*>(p.stats + (m * slices + (n0 / 96 + wh)) * 2) = make_float2(mean, m2);
        }
      }
      return;
    }
  }

  // ---- epilogue: bias + act -> LDS tile -> 16-byte stores (+ residual prefetched before the transposition);
  //      f32 output goes in two 96-column passes (one per channel half)
  constexpr int OES = sizeof(TO), OVEC = 16 / OES;
  constexpr int NPASS = OES == 4 ? 2 : 1, CW = BN / NPASS;
  constexpr int CROWB = CW * OES + 16;
  constexpr int NCH = CW / OVEC;
  constexpr int NIT = BM * NCH / 512;                     // 12 (both output types)
  static_assert(BM * NCH % 512 == 0, "store loop must divide evenly");
  char* const Ct = smem;
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
    const bool full = n0 + pass * CW + CW <= p.N;
    u32x4 rv[NIT];
    if (p.res && full) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * 512, row = idx / NCH, ch = idx - row * NCH;
        int m = m0 + row;
        m = m < p.M ? m : p.M - 1;
        const size_t rpix = p.res_mod > 0 ? (size_t)(m % p.res_mod) : (size_t)m;
        rv[it] = *reinterpret_cast<const u32x4*>(p.res + (rpix * p.res_ld + n0 + pass * CW + ch * OVEC) * OES);
      }
    }
    if (NPASS == 1 || wh == pass) {
      const int cbase = NPASS == 1 ? wh * 96 : 0;
      with_act<true>(p.act_after_res ? CVMI_ACT_NONE : p.act, [&](auto actf) {
        if constexpr (M16) {
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
              const int nl = cbase + i * 32 + hh * 16 + 4 * g16;
              const int n = n0 + pass * CW + nl;
              f32x4 bv = {0.f, 0.f, 0.f, 0.f};
              if (n < p.N) bv = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
              for (int pq = 0; pq < 4; ++pq) {
                const int ml = wp * 64 + pq * 16 + r16;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = actf(acc16[i][hh][pq][e] + bv[e]);
                char* d = Ct + ml * CROWB + nl * OES;
                if constexpr (OES == 2) {
                  f16x4 hv = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                  *reinterpret_cast<f16x4*>(d) = hv;
                } else {
                  f32x4 fv = {v[0], v[1], v[2], v[3]};
                  *reinterpret_cast<f32x4*>(d) = fv;
                }
              }
            }
        } else
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int nl = cbase + i * 32 + 8 * q + 4 * lh;
            const int n = n0 + pass * CW + nl;
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (n < p.N) bv = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const int ml = wp * 64 + j * 32 + lr;
              float v[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = actf(acc[i][j][4 * q + e] + bv[e]);
              char* d = Ct + ml * CROWB + nl * OES;
              if constexpr (OES == 2) {
                f16x4 hv = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                *reinterpret_cast<f16x4*>(d) = hv;
              } else {
                f32x4 fv = {v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4*>(d) = fv;
              }
            }
          }
        }
      });
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + it * 512, row = idx / NCH, ch = idx - row * NCH;
      const int m = m0 + row, n = n0 + pass * CW + ch * OVEC;
      if (m >= p.M || n >= p.N) continue;
      u32x4 cv = *reinterpret_cast<const u32x4*>(Ct + row * CROWB + ch * 16);
      if (p.res || p.act_after_res) {
        float a[OVEC];
        unpack16<TO>(cv, a);
        if (p.res) {
          float r[OVEC];
          if (full) {
            unpack16<TO>(rv[it], r);
          } else {
            const size_t rpix = p.res_mod > 0 ? (size_t)(m % p.res_mod) : (size_t)m;
            unpack16<TO>(*reinterpret_cast<const u32x4*>(p.res + (rpix * p.res_ld + n) * OES), r);
          }
#pragma unroll
          for (int e = 0; e < OVEC; ++e) a[e] += r[e];
        }
        if (p.act_after_res) {
#pragma unroll
          for (int e = 0; e < OVEC; ++e) a[e] = act_apply<true>(a[e], p.act);
        }
        cv = pack16<TO>(a);
        if constexpr (OES == 4) {
          // row statistics for the NEXT LayerNorm over these rows: this thread's 4 values -> (sum, sum of squares), parked in the chunk slot it
          // has just consumed (no other thread reads that slot), reduced per row below
          if (p.stats) {                                      // (mean of the 4 values, their squared deviations from it)
            const float m4 = 0.25f * ((a[0] + a[1]) + (a[2] + a[3]));
            const float d0 = a[0] - m4, d1 = a[1] - m4, d2 = a[2] - m4, d3 = a[3] - m4;
            *reinterpret_cast<float2*>(Ct + row * CROWB + ch * 16) = make_float2(m4, fmaf(d0, d0, fmaf(d1, d1, fmaf(d2, d2, d3 * d3))));
          }
        }
      }
      *reinterpret_cast<u32x4*>(p.y + ((size_t)m * p.y_ld + n) * OES) = cv;
    }
    if constexpr (OES == 4) {
      if (p.stats) {                                        // (uniform; the host admits it only with a residual and whole 96-column slices)
        __syncthreads();
        if (tid < BM && m0 + tid < p.M) {
          float mean_ = 0.f, m2_ = 0.f;
#pragma unroll
          for (int ch = 0; ch < NCH; ++ch) {                // fixed order: bit-identical replays; pairwise update of Chan et al. (groups of 4)
            const float2 t = *reinterpret_cast<const float2*>(Ct + tid * CROWB + ch * 16);
            const float dl = t.x - mean_, nn = (float)(4 * ch);
            mean_ += dl * (4.0f / (nn + 4.0f));
            m2_ += t.y + dl * dl * (nn * 4.0f / (nn + 4.0f));
          }
          const int slices = p.N / CW;                      // (slice mean, sum of squared deviations): the format the direct M16 epilogue writes
          *reinterpret_cast<float2*>(p.stats + ((size_t)(m0 + tid) * slices + (n0 / CW + pass)) * 2) = make_float2(mean_, m2_);
        }
      }
    }
    if (pass + 1 < NPASS) __syncthreads();
  }
}

// ---- Row-block-persistent form of gemm256x192_kernel<float, true> for the launch it is dominant in (Hiera-L stage-3 fc2: M = 65536,
//      N = 576 = 3 tiles, K = 2304, f32 output + f32 residual + row statistics).  Measured on the one-tile-per-workgroup kernel (r03): a
//      round of 256 tiles is ~43 us of K-loop at the matrix pipe's limit followed by ~20 us in which every CU reads its 196 KB of residual
//      and writes its 196 KB of output at the same moment -- the launch pays its MFMA bound and its HBM bound in series.  Here a persistent
//      workgroup per CU walks its three tiles as ONE software pipeline of 3 x 36 K-tiles (the DMA of the next tile's first two K-tiles is
//      issued under the current tile's last two; which tiles: see tile0 below -- a first form that gave a workgroup the three column tiles
//      of ONE row block re-read A from HBM twice, 37 MB of A rows per XCD do not live in a 4 MB L2: 290 us), and the residual is folded
//      INTO THE ACCUMULATOR while the K-loop runs: a tile's K-loop is four unrolled segments, segment s loads the 6 x 16 bytes per lane of
//      16-pixel block s (inline-asm loads the compiler's waitcnt pass does not see; the counted vmcnt of the K-tile after next retires
//      them, loads return in order) and adds them to acc16[..][..][s] in the segment's last K-tile.  The K-tile in which a workgroup issues its loads
//      is staggered by blockIdx (the CUs run in lock step; 256 x 49 KB at one instant would queue behind each other).  What is left
//      between two tiles is + bias, the statistics and 24 stores per lane; stores and DMA loads share vmcnt on gfx9, and because loads
//      retire in order among themselves a counted wait after the stores is conservative, never early.
//      Summation order: (residual + sum of products) + bias instead of (sum + bias) + residual -- f32 either way.
__global__ __launch_bounds__(512, 1) void gemm256x192r_kernel(const ConvKArgs p) {
  constexpr int BM = 256, BKB = 128;
  constexpr int STAGE = (BM + 192) * BKB;                 // 56 KiB
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  // Tiles of this workgroup: the order the hardware would dispatch the one-tile kernel's grid in.  Workgroup i runs on XCD i % 8; XCD x owns
  // tiles [x T / 8, (x + 1) T / 8) (row block = tile / nb_n: the column siblings of a row block are neighbours), and in round r its 32
  // workgroups take 32 consecutive ones -- the three siblings stream the same A rows through that XCD's L2 at the same time, as before.
  const int wgx = (int)gridDim.x >> 3, per_xcd = p.per_xcd;                 // workgroups per XCD, tiles per XCD (host: T % gridDim == 0)
  const int tile0 = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
  const int ntile = per_xcd / wgx;
  const long long xrow = (long long)p.x0_ld * 2 * BM, wrow = (long long)p.Kpad * 2 * 192;      // bytes per row block of X / per column tile of W
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wp = wv & 3, wh = wv >> 2;
  const int r16 = lane & 15, g16 = lane >> 4;
  const int nk = p.K >> 6, nseg = nk >> 2;                // host: K % 256 == 0, K / 256 >= 9
  const int ng = nk * ntile;
  const int ji = (int)(blockIdx.x % 7u);                  // K-tile of a segment in which this workgroup issues its residual loads

  const char* src[7];
  int dst[7];
#pragma unroll
  for (int u = 0; u < 7; ++u) {
    int row0;
    if (u < 4) row0 = (u >> 1) * 128 + (wv * 2 + (u & 1)) * 8;
    else row0 = BM + (wv >> 2) * 96 + (u - 4) * 32 + (wv & 3) * 8;
    const int row = row0 + (lane >> 3);
    const int slot = (lane & 7) ^ ((row >> 1) & 7);
    if (row < BM) src[u] = p.x0 + ((size_t)row * p.x0_ld + slot * 8) * 2;
    else src[u] = p.w + ((size_t)(row - BM) * p.Kpad + slot * 8) * 2;
    dst[u] = row0 * BKB;
  }
#define R192_ISSUE1(u, stage, off)                                                                                       \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[u] + (off)),                      \
                                   (__attribute__((address_space(3))) void*)(smem + (stage) * STAGE + dst[u]), 16, 0, 0)
#define R192_X0(stage, xo) do { R192_ISSUE1(0, stage, xo); R192_ISSUE1(1, stage, xo); } while (0)
#define R192_X1(stage, xo) do { R192_ISSUE1(2, stage, xo); R192_ISSUE1(3, stage, xo); } while (0)
#define R192_WC(k, stage, wo) R192_ISSUE1(4 + (k), stage, wo)

  f32x4 acc16[3][2][4];                                   // [32-channel block][16-channel half][16-pixel block]
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc16[i][hh][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 rv[6];                                            // the residual batch in flight
#pragma unroll
  for (int k = 0; k < 6; ++k) rv[k] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int sw = (r16 >> 1) & 7;
  int ko[4];
#pragma unroll
  for (int s2 = 0; s2 < 4; ++s2) ko[s2] = (((4 * (s2 & 1) + g16) ^ sw) << 4) + (s2 >> 1) * 16 * BKB;
  const int xbase = (wp * 64 + r16) * BKB, wbase = (BM + wh * 96 + r16) * BKB;

  // stream offsets (bytes) of K-tiles g + 1 and g + 2 of the ntile x nk stream
  const long long xo0 = (long long)(tile0 / p.nb_n) * xrow, wo0 = (long long)(tile0 % p.nb_n) * wrow;
  long long xo1 = xo0 + BKB, wo1 = wo0 + BKB, xo2 = xo0 + 2 * BKB, wo2 = wo0 + 2 * BKB;
  int kt2 = 2, tl2 = tile0;                               // (g + 2) % nk and the tile K-tile g + 2 belongs to
  R192_X0(0, xo0); R192_WC(0, 0, wo0); R192_X1(0, xo0); R192_WC(1, 0, wo0); R192_WC(2, 0, wo0);
  R192_X0(1, xo1); R192_WC(0, 1, wo1); R192_X1(1, xo1); R192_WC(1, 1, wo1);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wh == 1) __builtin_amdgcn_s_barrier();

  u32x4 xf[2][4], wf[4];
#define R192_READ_X(j)                                                                                                   \
  _Pragma("unroll") for (int s2 = 0; s2 < 4; ++s2)                                                                       \
      xf[j][s2] = *reinterpret_cast<const u32x4*>(st + xbase + (j) * 32 * BKB + ko[s2])
#define R192_READ_W(i)                                                                                                   \
  _Pragma("unroll") for (int s2 = 0; s2 < 4; ++s2)                                                                       \
      wf[s2] = *reinterpret_cast<const u32x4*>(st + wbase + (i) * 32 * BKB + ko[s2])
#define R192_SYNC_IN()                                                                                                   \
  do {                                                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
    __builtin_amdgcn_s_barrier();                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
    __builtin_amdgcn_s_setprio(1);                                                                                       \
  } while (0)
#define R192_SYNC_OUT()                                                                                                  \
  do {                                                                                                                   \
    __builtin_amdgcn_s_setprio(0);                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
    __builtin_amdgcn_s_barrier();                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
  } while (0)
#define R192_MMA(i)                                                                                                      \
  asm volatile("" : "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]));                                                 \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                       \
  _Pragma("unroll") for (int hh = 0; hh < 2; ++hh)                                                                       \
  _Pragma("unroll") for (int pq = 0; pq < 4; ++pq)                                                                       \
    acc16[i][hh][pq] = CVMI_MFMA_16X16X32(__builtin_bit_cast(f16x8, wf[ks + 2 * hh]),                                    \
                                          __builtin_bit_cast(f16x8, xf[pq >> 1][ks + 2 * (pq & 1)]), acc16[i][hh][pq], 0, 0, 0); \
  asm volatile("" : "+v"(acc16[i][0][0]), "+v"(acc16[i][0][1]), "+v"(acc16[i][0][2]), "+v"(acc16[i][0][3]),              \
                    "+v"(acc16[i][1][0]), "+v"(acc16[i][1][1]), "+v"(acc16[i][1][2]), "+v"(acc16[i][1][3]));
#define R192_RLOAD(k, off)                                                                                               \
  asm volatile("global_load_dwordx4 %0, %1, off offset:" #off : "+v"(rv[k]) : "v"(rptr) : "memory")

  int g = 0;
#pragma unroll 1
  for (int tl = 0; tl < ntile; ++tl) {
    const int tile = tile0 + tl * wgx, bm = tile / p.nb_n, bn = tile - bm * p.nb_n;
    const size_t mrow = (size_t)bm * BM + wp * 64 + r16;  // this lane's pixel of 16-pixel block 0
    const int ncol = bn * 192 + wh * 96 + 4 * g16;        // this lane's first channel of the tile
    // the tile's bias: loaded here, used behind the K-loop, and by inline asm like the residual -- hipcc's waitcnt pass would drain vmcnt
    // (the DMA in flight) in front of the first use of a load it can see.  K-tile 0's counted wait retires them.
    f32x4 bv[3][2];
    {
      const float* bptr = p.bias + ncol;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bv[0][0]) : "v"(bptr) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:64" : "=v"(bv[0][1]) : "v"(bptr) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:128" : "=v"(bv[1][0]) : "v"(bptr) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:192" : "=v"(bv[1][1]) : "v"(bptr) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:256" : "=v"(bv[2][0]) : "v"(bptr) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:320" : "=v"(bv[2][1]) : "v"(bptr) : "memory");
    }
#pragma unroll
    for (int seg = 0; seg < 4; ++seg) {
      const float* rptr = reinterpret_cast<const float*>(p.res) + (mrow + seg * 16) * p.res_ld + ncol;
#pragma unroll 1
      for (int j = 0; j < nseg; ++j, ++g) {
        const int b = g & 1;
        const char* st = smem + b * STAGE;
        const bool dma = !(p.diag & 1), more1 = dma && g + 1 < ng, more2 = dma && g + 2 < ng;
        if (j == nseg - 1) {                                // the same K-tile in EVERY workgroup (the rounding sequence of a row must not depend on where its image sits in the batch); the batch was retired by the counted wait of K-tile ji + 1 <= nseg - 2
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) acc16[i][hh][seg] += rv[2 * i + hh];
        }
        // ---- phase 1
        R192_READ_W(0);
        R192_READ_X(0);
        R192_READ_X(1);
        if (more1) R192_WC(2, b ^ 1, wo1);
        R192_SYNC_IN();
        if (!(p.diag & 2)) { R192_MMA(0) }
        R192_SYNC_OUT();
        // ---- phase 2
        R192_READ_W(1);
        // (both X halves of stage b were read in phase 1: X1 goes out here, a phase earlier than in gemm256x192_kernel -- the A rows are the
        //  loads that come from HBM, and CVMI_G192_DIAG shows the loop waiting on its DMA, not on the matrix pipe: r03, 181 us WITHOUT the MFMAs)
        if (more2) { R192_X0(b, xo2); R192_X1(b, xo2); R192_WC(0, b, wo2); }
        R192_SYNC_IN();
        if (!(p.diag & 2)) { R192_MMA(1) }
        R192_SYNC_OUT();
        // ---- phase 3
        R192_READ_W(2);
        if (more2) { R192_WC(1, b, wo2); }
        if (j == ji && !(p.diag & 4)) {                     // (uniform) six more loads in flight behind the DMA: one counted wait covers both
          R192_RLOAD(0, 0); R192_RLOAD(1, 64); R192_RLOAD(2, 128); R192_RLOAD(3, 192); R192_RLOAD(4, 256); R192_RLOAD(5, 320);
          if (more2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");          // (cannot happen for ji + 2 < nseg; kept exact anyway)
        } else {
          if (more2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
          else if (more1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        R192_SYNC_IN();
        if (!(p.diag & 2)) { R192_MMA(2) }
        R192_SYNC_OUT();
        xo1 = xo2; wo1 = wo2;
        if (++kt2 == nk) {
          kt2 = 0; tl2 += wgx;
          const int bm2 = tl2 / p.nb_n;
          xo2 = (long long)bm2 * xrow; wo2 = (long long)(tl2 - bm2 * p.nb_n) * wrow;
        } else { xo2 += BKB; wo2 += BKB; }
      }
    }
    // ---- between two tiles: + bias, row statistics, 24 stores per lane; no LDS, no barrier (the next tile's first K-tiles are landing)
    {
      const int slices = p.N / 96;
#pragma unroll
      for (int pq = 0; pq < 4; ++pq) {
        const size_t m = mrow + pq * 16;
        float* yrow = reinterpret_cast<float*>(p.y) + m * p.y_ld + ncol;
        f32x4 v[3][2];
        float sm = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[i][hh][e] = acc16[i][hh][pq][e] + bv[i][hh][e];
            *reinterpret_cast<f32x4*>(yrow + i * 32 + hh * 16) = v[i][hh];
            sm += (v[i][hh][0] + v[i][hh][1]) + (v[i][hh][2] + v[i][hh][3]);
            acc16[i][hh][pq] = (f32x4){0.f, 0.f, 0.f, 0.f};
          }
        if (p.stats) {                                          // (uniform)
          sm += __shfl_xor(sm, 16);
          sm += __shfl_xor(sm, 32);
          const float mean = sm * (1.0f / 96.0f);
          float m2 = 0.f;
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
              for (int e = 0; e < 4; ++e) { const float dv = v[i][hh][e] - mean; m2 = fmaf(dv, dv, m2); }
          m2 += __shfl_xor(m2, 16);
          m2 += __shfl_xor(m2, 32);
          if (g16 == 0) *reinterpret_cast<float2*>(p.stats + (m * slices + (bn * 2 + wh)) * 2) = make_float2(mean, m2);
        }
      }
    }
  }
  if (wh == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef R192_ISSUE1
#undef R192_X0
#undef R192_X1
#undef R192_WC
#undef R192_READ_X
#undef R192_READ_W
#undef R192_SYNC_IN
#undef R192_SYNC_OUT
#undef R192_MMA
#undef R192_RLOAD
}

template <typename TO>
int launch_g256x192(ConvKArgs& a, hipStream_t stream) {
  constexpr int lds = 2 * (256 + 192) * 128;
  constexpr int epi = 256 * ((sizeof(TO) == 4 ? 96 : 192) * (int)sizeof(TO) + 16);
  constexpr int bytes = lds > epi ? lds : epi;
  static bool attr_done = false;
  if (!attr_done) {
    CVMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256x192_kernel<TO, false>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    CVMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256x192_kernel<TO, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    attr_done = true;
  }
  a.nb_n = cdiv(a.N, 192);
  const long long blocks = (long long)cdiv(a.M, 256) * a.nb_n;
  CVMI_CHECK(blocks > 0 && blocks < (1ll << 31), "conv2d: bad grid %lld", blocks);
  static const int m16 = getenv("CVMI_G192_M16") ? atoi(getenv("CVMI_G192_M16")) : 1;        // MFMA shape: 1 = 16x16x32, 0 = 32x32x16 (A/B runs)
  // Row blocks last-to-first: the A operand (the fp16 hidden activation, 302 MB at B = 16: more than the 256 MiB Infinity Cache) was written
  // first-to-last by the launch in front of this one, so its LAST rows are the ones still on the die when this launch starts.
  static const int rev = getenv("CVMI_G192_REV") ? atoi(getenv("CVMI_G192_REV")) : 0;   // measured r03 (call r3j): no effect (207.6 vs 206.3 us) -- off
  a.rev_m = rev;
  if constexpr (sizeof(TO) == 4) {
    static const int persist = getenv("CVMI_G192_PERSIST") ? atoi(getenv("CVMI_G192_PERSIST")) : 1;     // A/B runs only
    static const int ncu = [] {                                  // one persistent workgroup per CU of THIS device (MI355X: 256 = 8 XCDs x 32)
      int dev = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
      return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }();
    if (persist && m16 && a.res && a.act == CVMI_ACT_NONE && !a.act_after_res && a.res_mod == 0 && a.M % 256 == 0 && a.N % 192 == 0 &&
        a.K % 256 == 0 && a.K / 256 >= 9 && ncu % 8 == 0 && blocks % ncu == 0) {  // whole rounds of tiles: one workgroup per CU
      static bool attr_r = false;
      if (!attr_r) {
        CVMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256x192r_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_r = true;
      }
      cvmi_note_kernel("gemm256x192r_kernel");
      a.per_xcd = (int)(blocks / 8);
      static const int diag = getenv("CVMI_G192_DIAG") ? atoi(getenv("CVMI_G192_DIAG")) : 0;
      a.diag = diag;
      hipLaunchKernelGGL(gemm256x192r_kernel, dim3((unsigned)ncu), dim3(512), lds, stream, a);
      CVMI_LAUNCH_CHECK();
      return 0;
    }
  }
  cvmi_note_kernel("gemm256x192_kernel<%s, %s>", sizeof(TO) == 2 ? CVMI_F16NAME : "float", CVMI_BOOLNAME(m16));
  if (m16) hipLaunchKernelGGL((gemm256x192_kernel<TO, true>), dim3((unsigned)blocks), dim3(512), bytes, stream, a);
  else hipLaunchKernelGGL((gemm256x192_kernel<TO, false>), dim3((unsigned)blocks), dim3(512), bytes, stream, a);
  CVMI_LAUNCH_CHECK();
  return 0;
}

template <typename TO>
int launch_g256(ConvKArgs& a, hipStream_t stream, int stagger, bool im2col = false) {
  constexpr int lds = 2 * 512 * 128;                     // two stages; the epilogue tile (256 x 528 B) is larger: 135168
  constexpr int epi = 256 * ((sizeof(TO) == 4 ? 128 : 256) * (int)sizeof(TO) + 16);
  constexpr int bytes = lds > epi ? lds : epi;
  static bool attr_done = false;
  if (!attr_done) {
    CVMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<TO, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    CVMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<TO, false>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    CVMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<TO, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    attr_done = true;
  }
  a.nb_n = cdiv(a.N, 256);
  const long long blocks = (long long)cdiv(a.M, 256) * a.nb_n;
  CVMI_CHECK(blocks > 0 && blocks < (1ll << 31), "conv2d: bad grid %lld", blocks);
  if (im2col) {
    cvmi_note_kernel("gemm256_kernel<%s, true, true>", sizeof(TO) == 2 ? CVMI_F16NAME : "float");
    hipLaunchKernelGGL((gemm256_kernel<TO, true, true>), dim3((unsigned)blocks), dim3(512), bytes, stream, a);
    CVMI_LAUNCH_CHECK();
    return 0;
  }
  cvmi_note_kernel("gemm256_kernel<%s, %s>", sizeof(TO) == 2 ? CVMI_F16NAME : "float", CVMI_BOOLNAME(stagger));
  if (stagger) hipLaunchKernelGGL((gemm256_kernel<TO, true>), dim3((unsigned)blocks), dim3(512), bytes, stream, a);
  else hipLaunchKernelGGL((gemm256_kernel<TO, false>), dim3((unsigned)blocks), dim3(512), bytes, stream, a);
  CVMI_LAUNCH_CHECK();
  return 0;
}

template <typename T, typename TO, int BM, int BN, int WM, int WN, int BKB, bool PLAIN, int KS = 1>
int launch_cfg2(ConvKArgs& a, hipStream_t stream) {
  constexpr int ROWB = BKB + 16;
  constexpr size_t stage = (size_t)2 * (BM + BN) * ROWB * KS;
  constexpr size_t epi = (size_t)BM * (BN * sizeof(TO) + 16);
  constexpr size_t red = (size_t)(KS - 1) * BM * BN * sizeof(float);
  constexpr size_t lds0 = stage > epi ? stage : epi;
  constexpr size_t lds1 = ((lds0 > red ? lds0 : red) + 15) / 16 * 16;
  constexpr size_t lds = lds1 + BN * sizeof(float);           // + the parked bias row
  static_assert(lds <= 160 * 1024, "workgroup LDS");
  a.bias_off = (int)lds1;
  static bool attr_done = false;
  if (!attr_done && lds > 64 * 1024) {
    CVMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<T, TO, BM, BN, WM, WN, BKB, PLAIN, KS>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  a.nb_n = cdiv(a.N, BN);
  const long long blocks = (long long)cdiv(a.M, BM) * a.nb_n;
  CVMI_CHECK(blocks > 0 && blocks < (1ll << 31), "conv2d: bad grid %lld", blocks);
  cvmi_note_kernel("igemm_kernel<%s, %s, %d, %d, %d, %d, %d, %s, %d>", sizeof(T) == 2 ? CVMI_F16NAME : "float", sizeof(TO) == 2 ? CVMI_F16NAME : "float", BM, BN, WM, WN, BKB, CVMI_BOOLNAME(PLAIN), KS);
  hipLaunchKernelGGL((igemm_kernel<T, TO, BM, BN, WM, WN, BKB, PLAIN, KS>), dim3((unsigned)blocks), dim3(WM * WN * 64 * KS), lds, stream, a);
  CVMI_LAUNCH_CHECK();
  return 0;
}

// Intra-workgroup split-K (igemm_kernel KS > 1) for the f16 64-row tiles: grids of at most two workgroups per CU with a
// deep K are latency chains (one wave per SIMD, every K-tile a dependent global -> LDS -> MFMA round trip).
template <typename T, typename TO, int BM, int BN, int WM, int WN>
int launch_ksplit(ConvKArgs& a, hipStream_t stream, int ks) {
  if constexpr (sizeof(T) == 2 && sizeof(TO) == 2 && BM == 64 && WM * WN == 4) {
    const bool plain = a.plain != 0;
    if (ks == 4) return plain ? launch_cfg2<T, TO, BM, BN, WM, WN, 64, true, 4>(a, stream) : launch_cfg2<T, TO, BM, BN, WM, WN, 64, false, 4>(a, stream);
    if (ks == 2) return plain ? launch_cfg2<T, TO, BM, BN, WM, WN, 64, true, 2>(a, stream) : launch_cfg2<T, TO, BM, BN, WM, WN, 64, false, 2>(a, stream);
  }
  return -1;
}

template <typename T, typename TO, int BM, int BN, int WM, int WN>
int launch_cfg(ConvKArgs& a, hipStream_t stream) {
  const bool plain = a.plain != 0;
  if constexpr (sizeof(T) == 2 && sizeof(TO) == 2 && BM == 64 && WM * WN == 4) {
    static const int ks_env = getenv("CVMI_KSPLIT") ? atoi(getenv("CVMI_KSPLIT")) : -1;        // tuning experiments only: 0 off, 2 / 4 forced
    static const int ks4_mink = getenv("CVMI_KS4_MINK") ? atoi(getenv("CVMI_KS4_MINK")) : 32;
    static const int ks2_mink = getenv("CVMI_KS2_MINK") ? atoi(getenv("CVMI_KS2_MINK")) : 12;
    const long long blocks = (long long)cdiv(a.M, BM) * cdiv(a.N, BN);
    const int nkt = a.Kpad / 32;                                                              // 64-byte K-tiles
    // measured on the YOLO11-n B=32 layers (us, 1 / 2 / 4 groups): Detect cv2.2.0 (200 tiles of 64x64, 72 K-tiles) 51 / 34 / 28;
    // model.20 (200 of 64x128, 36) 31 / 23 / 25; model.7 (400 of 64x128, 36) 33 / 27 / 39 (four groups of a 64x128 tile need
    // 123 KB of LDS: one workgroup per CU); grids above two workgroups per CU lose (model.5: 47 / 52)
    int ks = 1;
    if (BN == 64 && blocks <= 256 && nkt >= ks4_mink) ks = 4;
    else if (blocks <= 512 && nkt >= ks2_mink) ks = 2;
    if (ks_env >= 0) ks = (ks_env == 2 || ks_env == 4) && nkt >= 2 * ks_env ? ks_env : 1;
    if (ks > 1 && a.Kpad % 32 == 0) {
      const int rc = launch_ksplit<T, TO, BM, BN, WM, WN>(a, stream, ks);
      if (rc >= 0) return rc;
    }
  }
  // 128-byte K-tiles halve the barrier count; 64-byte tiles only when K is too short to fill one
  static const int force_bkb = getenv("CVMI_BKB") ? atoi(getenv("CVMI_BKB")) : 0;      // tuning experiments only
  // measured (tools/gemm_bench.py): 128-byte tiles pay for deep K (>= 2 KB per row: fewer barriers) and for rows
  // that are exactly one tile; in between, 64-byte tiles win through higher occupancy (40 KB vs 74 KB of LDS)
  const int kbytes = a.Kpad * (int)sizeof(T);
  bool wide = plain && kbytes % 128 == 0 && (kbytes >= 2048 || kbytes == 128);
  if (force_bkb == 64) wide = false;
  if (force_bkb == 128) wide = a.Kpad * (int)sizeof(T) >= 256 && (a.Kpad * (int)sizeof(T)) % 128 == 0;
  if (wide) return plain ? launch_cfg2<T, TO, BM, BN, WM, WN, 128, true>(a, stream) : launch_cfg2<T, TO, BM, BN, WM, WN, 128, false>(a, stream);
  return plain ? launch_cfg2<T, TO, BM, BN, WM, WN, 64, true>(a, stream) : launch_cfg2<T, TO, BM, BN, WM, WN, 64, false>(a, stream);
}

// Large 3 x 3 convolutions over >= 64 channels (YOLO11-l: model.1 / .3 / .5 / .7 / .17 / .20, the 256-channel bottlenecks) take the counted-DMA
// 256 x 256 pipeline with an im2col source (gemm256_kernel<.., IM2COL>): Cin a power-of-two multiple of 64 (a K-tile inside one tap), at least half
// a round of tiles, at least half of the column tiles used (measured r04, YOLO11-l B = 64: model.3 1353 -> 529 us, model.5 1285 -> 526, .7 / .20
// 360 / 379 -> 129, model.1 -- N = 128, half of every tile's columns idle -- 957 -> see profiles/r04_ab_runs.md).
bool im2col_256_ok(const ConvKArgs& a) {
  static const int use_g256 = getenv("CVMI_G256") ? atoi(getenv("CVMI_G256")) : 1;
  static const int use_i2c = getenv("CVMI_G256_IM2COL") ? atoi(getenv("CVMI_G256_IM2COL")) : 1;      // A/B runs only
  static const double i2c_eff = getenv("CVMI_G256_I2C_EFF") ? atof(getenv("CVMI_G256_I2C_EFF")) : 0.75;     // tuning experiments only
  const int tpt = a.ctot / 64;
  if (!(use_g256 && use_i2c && !a.plain && !a.rows2 && a.x1 == nullptr && a.up0 == 0 && !a.scalar_gather && a.KW == 3 && a.K == 9 * a.ctot &&
        a.ctot % 64 == 0 && tpt >= 1 && (tpt & (tpt - 1)) == 0 && a.Kpad == a.K && a.N % 8 == 0 && a.shuf_c == 0 && a.res_rep <= 1 && !a.stats && a.res_mod == 0))
    return false;
  const long long tiles = (long long)cdiv(a.M, 256) * cdiv(a.N, 256);
  // (N = 128 fills half of every tile's columns: still a gain over conv_tile_kernel for the 64-channel model.1, 957 -> 633 us, a loss against the
  //  128 x 128 kernel for the 128-channel bottlenecks: YOLO11-l step 12.69 -> 12.82 ms with all of them rerouted)
  const double eff = (double)a.N / (cdiv(a.N, 256) * 256);
  static const int i2c_tiles = getenv("CVMI_G256_I2C_TILES") ? atoi(getenv("CVMI_G256_I2C_TILES")) : 128;      // tuning experiments only
  return (eff >= i2c_eff || (tpt == 1 && eff >= 0.5)) && tiles >= i2c_tiles;
}

template <typename T, typename TO>
int launch_typed(ConvKArgs& a, hipStream_t stream) {
  const long long M = a.M;
  const int N = a.N;
  static const int use_glds = getenv("CVMI_GLDS") ? atoi(getenv("CVMI_GLDS")) : 1;           // tuning experiments only
  static const int use_g256 = getenv("CVMI_G256") ? atoi(getenv("CVMI_G256")) : 1;           // 0 off, 1 staggered, 2 lock-step
  if constexpr (sizeof(T) == 2) {
    if (im2col_256_ok(a)) {
      const int tpt = a.ctot / 64;
      a.im2col_shift = 0;
      while ((1 << a.im2col_shift) < tpt) ++a.im2col_shift;
      return launch_g256<TO>(a, stream, 1, true);
    }
    // big plain f16 GEMMs: >= one 256^2 tile per CU and little column-tile waste (N = 576 -> 3 tiles, 75 % used)
    // measured on Hiera-L shapes: wins when >= 80 % of the column tiles and of the last round of 256 tiles is used
    // (N = 576 -> 75 % of 3 column tiles: ties / loses against the 128-row kernels below)
    static const int g256_mink = getenv("CVMI_G256_MINK") ? atoi(getenv("CVMI_G256_MINK")) : 128;     // tuning experiments only
    static const int g256_minn = getenv("CVMI_G256_MINN") ? atoi(getenv("CVMI_G256_MINN")) : 256;      // (r04: 384 -> 256, YOLO11-l model.2.cv2 654 -> 424 us; Hiera unaffected)
    if (use_g256 && a.plain && a.K % 8 == 0 && a.K >= g256_mink && a.Kpad % 64 == 0 && N % 8 == 0 && N >= g256_minn && a.shuf_c == 0 && a.res_rep <= 1) {
      const long long tiles = (long long)cdiv(M, 256) * cdiv(N, 256);
      const double col_eff = (double)N / (cdiv(N, 256) * 256), wave_eff = (double)tiles / (double)(cdiv(tiles, 256) * 256);
      if (!a.stats && tiles >= 256 && ((col_eff >= 0.8 && wave_eff >= 0.8) || use_g256 >= 3)) {
        static const int use_p = getenv("CVMI_G256P") ? atoi(getenv("CVMI_G256P")) : 1;             // tuning experiments only
        if constexpr (sizeof(TO) == 2) {
          if (use_p && !a.res && !a.act_after_res) return launch_g256p(a, stream);
        }
        return launch_g256<TO>(a, stream, use_g256 != 2);
      }
      // N a multiple of 192 that the 256-wide tiling wastes (576 = 3 x 192)
      static const int use_g192 = getenv("CVMI_G192") ? atoi(getenv("CVMI_G192")) : 1;
      const long long tiles192 = (long long)cdiv(M, 256) * cdiv(N, 192);
      // (measured: K = 2304 309 -> 226 us; at K = 576 the 128 x 64 kernel's two workgroups per CU hide the f32 + residual epilogue better)
      static const int g192_mink = getenv("CVMI_G192_MINK") ? atoi(getenv("CVMI_G192_MINK")) : 1024;   // tuning experiments only
      // (the last round of tiles at least 3/4 used: 384 tiles -- SAM 2.1-L stage-3 fc2 at 8 images, one rank's share of an 8-GPU job -- run 119 us
      //  here against 140 us on the 128 x 64 kernel, r04; the 128 x 192 variant of that kernel: 145 us)
      static const double g192_eff = getenv("CVMI_G192_EFF") ? atof(getenv("CVMI_G192_EFF")) : 0.75;     // tuning experiments only
      if (use_g192 && N % 192 == 0 && a.K >= g192_mink && tiles192 >= 256 && (double)tiles192 / (double)(cdiv(tiles192, 256) * 256) >= g192_eff)
        return launch_g256x192<TO>(a, stream);
    }
  }
  CVMI_CHECK(!a.stats, "conv2d: row_stats is produced by the 256 x 192 GEMM only (plain f16 GEMM, N %% 192 == 0, K >= 1024, >= 256 tiles)");
  static const int glds_min_tiles = getenv("CVMI_GLDS_MINTILES") ? atoi(getenv("CVMI_GLDS_MINTILES")) : 512;   // tuning experiments only
  // (K not a multiple of the 128-byte tile: only from K = 256 elements up -- at K = 144 the padded third tile costs more than the DMA saves)
  if (use_glds && (a.plain || a.rows2) && a.K % (16 / (int)sizeof(T)) == 0 && (a.Kpad * (int)sizeof(T)) % 128 == 0 && a.K * (int)sizeof(T) >= 256 &&
      ((a.K * (int)sizeof(T)) % 128 == 0 || a.K >= 256) && N >= (a.rows2 ? 64 : 96) &&
      (long long)cdiv(M, 128) * cdiv(N, 128) >= glds_min_tiles) {                      // large GEMMs only: small grids need the smaller tiles below
    static const int glds_bn = getenv("CVMI_GLDS_BN") ? atoi(getenv("CVMI_GLDS_BN")) : 0;      // tuning experiments only
    if constexpr (sizeof(T) == 2 && sizeof(TO) == 4) {
      if (glds_bn == 192 && N % 192 == 0 && a.plain) return launch_glds<T, TO, 128, 192, 4, 2>(a, stream);
    }
    if (use_glds == 2 || N <= 640) return launch_glds<T, TO, 128, 64, 2, 2>(a, stream);      // measured: wins up to N = 576
    return launch_glds<T, TO, 128, 128, 2, 2>(a, stream);
  }
  static const char* force_tile = getenv("CVMI_TILE");                         // tuning experiments only: "BMxBN"
  if (force_tile && strchr(force_tile, (int)120)) {
    const int bm = atoi(force_tile), bnn = atoi(strchr(force_tile, 'x') + 1);
    if (bm == 256 && bnn == 32) return launch_cfg<T, TO, 256, 32, 4, 1>(a, stream);
    if (bm == 128 && bnn == 32) return launch_cfg<T, TO, 128, 32, 4, 1>(a, stream);
    if (bm == 128 && bnn == 64) return launch_cfg<T, TO, 128, 64, 2, 2>(a, stream);
    if (bm == 64 && bnn == 64) return launch_cfg<T, TO, 64, 64, 2, 2>(a, stream);
    if (bm == 128 && bnn == 128) return launch_cfg<T, TO, 128, 128, 2, 2>(a, stream);
    if (bm == 64 && bnn == 128) return launch_cfg<T, TO, 64, 128, 2, 2>(a, stream);
  }
  // Tile choice: BN covers Cout where it can (each gathered pixel row is then read once); BM
  // shrinks when the grid would not fill 256 CUs x 2.
  if (N <= 32) {
    if (M >= 256 * 512) return launch_cfg<T, TO, 256, 32, 4, 1>(a, stream);
    return launch_cfg<T, TO, 128, 32, 4, 1>(a, stream);
  }
  if (N <= 64) {
    if (M >= 128 * 512) return launch_cfg<T, TO, 128, 64, 2, 2>(a, stream);
    return launch_cfg<T, TO, 64, 64, 2, 2>(a, stream);
  }
  if (sizeof(TO) == 4 && sizeof(T) == 2 && N <= 192) return launch_cfg<T, TO, 128, 64, 2, 2>(a, stream);   // f32 out: 67 KB epilogue tile at BN=128
  const long long blocks128 = (long long)cdiv(M, 128) * cdiv(N, 128);
  if (blocks128 >= 512) return launch_cfg<T, TO, 128, 128, 2, 2>(a, stream);
  return launch_cfg<T, TO, 64, 128, 2, 2>(a, stream);
}

}  // namespace

int cvmi_conv_tile_try(const cvmi_conv_desc* d, hipStream_t stream);   // conv_tile.hip
#ifndef CVMI_OPERAND_BF16
extern "C" int cvmi_conv2d_bf16(const cvmi_conv_desc* d, cvmi_stream_t stream_);
#endif

extern "C" int CVMI_ENTRY(cvmi_conv2d)(const cvmi_conv_desc* d, cvmi_stream_t stream_) {
  CVMI_CHECK(d != nullptr, "conv2d: null descriptor");
#ifndef CVMI_OPERAND_BF16
  if (d->dtype == CVMI_BF16) return cvmi_conv2d_bf16(d, stream_);
#endif
  CVMI_CHECK(d->x0 && d->w && d->bias && d->y, "conv2d: null pointer");
  CVMI_CHECK(d->dtype == CVMI_T16 || d->dtype == CVMI_F32, "conv2d: bad dtype %d", d->dtype);
  const int es = d->dtype == CVMI_T16 ? 2 : 4;
  const int vec = 16 / es;
  const int oes = (d->dtype == CVMI_F32 || d->out_f32) ? 4 : 2;
  const int ovec = 16 / oes;
  CVMI_CHECK(d->B > 0 && d->H > 0 && d->W > 0 && d->OH > 0 && d->OW > 0 && d->N > 0, "conv2d: bad shape");
  CVMI_CHECK(d->KH > 0 && d->KW > 0 && d->stride > 0 && d->pad >= 0, "conv2d: bad kernel geometry");
  CVMI_CHECK(d->c0 > 0 && d->c1 >= 0 && (d->c1 == 0 || d->x1), "conv2d: bad sources");
  CVMI_CHECK((d->OH - 1) * d->stride - d->pad + d->KH - 1 < d->H + d->pad &&
             (d->OW - 1) * d->stride - d->pad + d->KW - 1 < d->W + d->pad, "conv2d: output larger than input allows");
  const int ctot = d->c0 + d->c1;
  const long long K = (long long)d->KH * d->KW * ctot;
  CVMI_CHECK(K < 65536 && d->Kpad >= K && d->Kpad % 32 == 0 && d->Kpad <= 65536, "conv2d: K=%lld Kpad=%d unsupported", K, d->Kpad);
  if (!d->scalar_gather) {
    CVMI_CHECK(d->c0 % vec == 0 && d->c1 % vec == 0, "conv2d: channels (%d,%d) not multiples of %d", d->c0, d->c1, vec);
    CVMI_CHECK(d->x0_ld % vec == 0 && (d->c1 == 0 || d->x1_ld % vec == 0), "conv2d: source ld not 16-byte aligned");
    CVMI_CHECK(((uintptr_t)d->x0 & 15) == 0 && ((uintptr_t)d->x1 & 15) == 0, "conv2d: source not 16-byte aligned");
  }
  CVMI_CHECK(d->x0_ld >= d->c0 && (d->c1 == 0 || d->x1_ld >= d->c1), "conv2d: ld smaller than channels");
  CVMI_CHECK(d->y_ld % ovec == 0 && ((uintptr_t)d->y & 15) == 0, "conv2d: output not 16-byte aligned");
  CVMI_CHECK(d->y_ld >= d->N || d->shuffle_cout > 0, "conv2d: y_ld < N");
  CVMI_CHECK(!d->res || (d->res_ld % ovec == 0 && ((uintptr_t)d->res & 15) == 0 && (d->res_ld >= d->N || d->shuffle_cout > 0)), "conv2d: residual misaligned");
  CVMI_CHECK(((uintptr_t)d->w & 15) == 0 && ((uintptr_t)d->bias & 15) == 0, "conv2d: weights misaligned");
  CVMI_CHECK(d->up0 == 0 || (d->H % 2 == 0 && d->W % 2 == 0), "conv2d: upsampled source needs even H, W");
  CVMI_CHECK(d->up1 == 0 || (d->H % 2 == 0 && d->W % 2 == 0), "conv2d: upsampled source needs even H, W");
  const long long M = (long long)d->B * d->OH * d->OW;
  CVMI_CHECK(M < (1ll << 31), "conv2d: M too large");

  ConvKArgs a;
  a.x0 = (const char*)d->x0; a.x1 = (const char*)d->x1; a.w = (const char*)d->w; a.bias = d->bias;
  a.res = (const char*)d->res; a.y = (char*)d->y;
  a.x0_ld = d->x0_ld; a.x1_ld = d->x1_ld; a.res_ld = d->res_ld; a.y_ld = d->y_ld;
  a.c0 = d->c0; a.ctot = ctot; a.up0 = d->up0; a.up1 = d->up1;
  a.H = d->H; a.W = d->W; a.OH = d->OH; a.OW = d->OW; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
  a.M = (int)M; a.N = d->N; a.K = (int)K; a.Kpad = d->Kpad;
  a.act = d->act; a.scalar_gather = d->scalar_gather; a.nb_n = 1;
  a.res_mod = d->res_mod; a.act_after_res = d->act_after_res; a.shuf_c = d->shuffle_cout; a.res_rep = d->res_rep;
  a.stats = d->row_stats;
  a.im2col_shift = 0;
  CVMI_CHECK(!d->row_stats || (d->out_f32 && d->res && !d->act_after_res && d->N % 192 == 0 && ((uintptr_t)d->row_stats & 7) == 0),
             "conv2d: row_stats needs the f32-output residual form with N a multiple of 192");
  a.plain = (d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && d->c1 == 0 && d->up0 == 0 && !d->scalar_gather &&
             d->OH == d->H && d->OW == d->W) ? 1 : 0;
  a.rows2 = (!a.plain && d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && !d->scalar_gather && d->OH == d->H && d->OW == d->W &&
             d->c1 > 0 && d->c0 % 64 == 0 && ctot % 64 == 0 && d->dtype == CVMI_T16) ? 1 : 0;
  CVMI_CHECK(d->res_mod >= 0 && d->shuffle_cout >= 0, "conv2d: negative res_mod / shuffle_cout");
  CVMI_CHECK(d->res_rep <= 1 || (d->res && d->B % d->res_rep == 0 && (d->shuffle_cout > 0 || d->res_mod == d->OH * d->OW)),
             "conv2d: res_rep needs a residual, B %% res_rep == 0 and either shuffle_cout or res_mod == OH * OW");
  if (d->shuffle_cout > 0) {
    CVMI_CHECK(d->N == 4 * d->shuffle_cout && d->shuffle_cout % ovec == 0 && d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 &&
               d->res_mod == 0, "conv2d: shuffle_cout needs a 1x1 conv with N == 4*shuffle_cout (multiple of %d)", ovec);
    CVMI_CHECK(d->y_ld >= d->shuffle_cout && (!d->res || d->res_ld >= d->shuffle_cout), "conv2d: shuffle output ld too small");
  }
  a.div_ctot.init((unsigned)ctot); a.div_kw.init((unsigned)d->KW);
  hipStream_t stream = (hipStream_t)stream_;
#ifndef CVMI_OPERAND_BF16
  if (d->KH > 1 && d->y_ld >= d->N && !(d->KH == 3 && d->dtype == CVMI_T16 && im2col_256_ok(a))) {      // (64 -> >= 128 channels at a large M: the 256-tile pipeline)
    const int rc = cvmi_conv_tile_try(d, stream);
    if (rc >= 0) return rc;
  }
  if (d->dtype == CVMI_F32) return launch_typed<float, float>(a, stream);
#else
  CVMI_CHECK(d->dtype == CVMI_BF16, "conv2d (bf16 build): dtype %d", d->dtype);
#endif
  if (d->out_f32) return launch_typed<f16, float>(a, stream);
  return launch_typed<f16, f16>(a, stream);
}
