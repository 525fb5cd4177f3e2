// Scaled-dot-product attention (softmax over keys) for every attention site of the path:
// YOLO C2PSA (400 tokens), Hiera windowed / global attention (head_dim 72, optional 2x2 q max-pool),
// SAM mask-decoder self / cross attention.
//
// fp16: flash-style, one wave per 32 queries, 32x32x16 MFMA, "swapped" products so that nothing
// crosses lanes between the two GEMMs:
//     S^T[key][q] = K . Q^T        (A = K rows from LDS, B = Q rows held in registers)
//     O^T[d][q]  += V^T . P^T      (A = V^T from a transposed LDS image, B = P^T = the S^T
//                                   accumulator itself, converted to fp16 in place)
// Each lane owns ONE query column, so the online-softmax state (max, sum) is a per-lane scalar and
// the row reductions are 15 in-register ops + one cross-half shuffle.
// K/V tiles (32 keys) are staged through LDS by a loader group of GS threads: GS = 256 shares one
// tile among the 4 waves of a workgroup (same batch/head, long sequences), GS = 64 gives every wave
// its own item (many small windows).
// f32 (parity mode): plain VALU kernel, one wave per query.
#include "common.hpp"
#include <stdlib.h>

namespace {

struct AttnArgs {
  const char* q; const char* k; const char* v; char* o;
  long long q_sb, q_sh, q_st, k_sb, k_sh, k_st, v_sb, v_sh, v_st, o_sb, o_sh, o_st;
  int B, heads, Nq, Nk, dqk, dv;
  float scale;
  int win, grid_h, grid_w, q_pool;
  int q_bdiv, kv_bdiv;   // batch sharing (no window): q rows of batch entry b come from entry b / q_bdiv, k / v rows from b / kv_bdiv
  int qtiles;      // ceil(Nq / 32)
  int items;       // B * heads * qtiles
  int diag;        // CVMI_ATTN_DIAG, timing experiments ONLY (results are wrong): bit 0 = attn_res256 skips its key-tile loop, bit 1 = skips its K / V DMA
  float defer;     // deferred-rescale threshold in log2 units (DEFER_LOG2; CVMI_ATTN_DEFER=0 restores "rescale on every new maximum" for A/B runs)
  int q_log2;      // cvmi_attn_desc.q_log2: q already carries scale * log2(e) (the dispatcher then passes scale = 1 / log2(e): every kernel's c = scale * log2(e) is 1 to one ulp; attn_dma72_kernel<.., QL = true> uses exactly 1)
  int xcd;         // 1: XCD-aware workgroup order (xcd_order below); 0: natural order (CVMI_ATTN_XCD=0, A/B runs only)
  FastDiv div_win; // window mode: key -> (row, column) inside the window without a hardware division
  // window mode, attn_res256_kernel: every integer division of its prologue / epilogue as a multiply (they were ~180 of the ~1100 vector
  // instructions a wave of that kernel issues; the vector issue port is what bounds it)
  int wpr, wpi;    // windows per grid row, windows per image
  FastDiv div_wpr, div_wpi, div_ow, div_heads;       // ow = output-window side (win, or win / 2 with q_pool)
};

// Row maxima of the score tiles.  fmaxf on values that come out of an MFMA compiles to v_max_f32 plus a canonicalising `v_max_f32 x, x, x`
// per operand (llvm.maxnum wants quieted inputs): 54 VALU instructions for the 32 values of a 64-key tile.  This translation unit is built
// with -fno-honor-nans (Makefile), under which the same source becomes 16 v_max3_f32.  NOT inline asm: hipcc's hazard recogniser cannot
// see into an asm statement, so an asm v_max3 reading an MFMA result gets no XDL-write -> VALU-read wait states and returns stale
// accumulators now and then (measured: replays of the same graph differed in the last bits -- tests/*_replay_properties).
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// Values of the two lane halves (lane, lane ^ 32) combined without LDS: __shfl_xor(x, 32) compiles to ds_bpermute_b32 -- an LDS round trip on
// the softmax's critical path, twice per key tile; v_permlane32_swap exchanges the halves in the VALU (r[0], r[1] = {own, partner} in one
// order or the other, which max and + do not care about).
__device__ __forceinline__ float xhalf_max(float x) {
  const int xi = __builtin_bit_cast(int, x);
  const auto r = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);
  return fmaxf(__builtin_bit_cast(float, (int)r[0]), __builtin_bit_cast(float, (int)r[1]));
}
__device__ __forceinline__ float xhalf_sum(float x) {
  const int xi = __builtin_bit_cast(int, x);
  const auto r = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);
  return __builtin_bit_cast(float, (int)r[0]) + __builtin_bit_cast(float, (int)r[1]);
}
// Deferred rescaling of the online softmax: the running reference m of a row only moves when the tile's maximum exceeds it by more than
// DEFER_LOG2 in the exponent's (log2) units; until then P = exp2((s - m) c) may reach 2^DEFER_LOG2 -- 256: exact powers of two away from
// overflow in fp16 / bf16 P and in the fp32 sums -- and the accumulator rescale (48 multiplies per lane, taken whenever ANY of the wave's rows
// saw a new maximum: almost every tile) becomes rare.  Everything at the old scale is rescaled exactly once when the reference does move:
// O and l here, and no P is pending (it is exponentiated after the decision).  The e4m3 product keeps DEFER = 0: its P is scaled by 2^8 already.
constexpr float DEFER_LOG2 = 8.0f;

// 16 bytes per lane, global -> LDS, buffer addressing: descriptor over `base` (wave-uniform), lane offset in one VGPR, `soff` in an SGPR.
// (A __device__ function, not kernel-body code: the host pass of hipcc drops a kernel's launch stub when its body names the descriptor type.)
__device__ __forceinline__ void dma16_buffer(const char* base, char* lds, int voff, int soff) {
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 0x7fffffff, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}

// exp2(s c - mc) over two 32 x 32 score tiles -> the 16-bit P fragments; returns this lane's part of the row sum (SUM = false: 0, the
// caller takes the row sums from the matrix pipe).  Scalar f32 on purpose: packed v_pk_fma_f32 / v_pk_add_f32 halve the instruction
// count but not the issue cycles beside MFMAs (MI355X_MICROARCH.md, 'price of one filler'; measured here r03: no change).
template <bool SUM>
__device__ __forceinline__ float softmax_tiles(const f32x16 (&sacc)[2], float c, float mc, f16x8 (&pf)[2][2]) {
  float ps = 0.f;
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pv = __builtin_amdgcn_exp2f(fmaf(sacc[u][r], c, -mc));
      if constexpr (SUM) ps += pv;
      pf[u][r >> 3][r & 7] = (f16)pv;
    }
  return ps;
}

// XCD-aware workgroup order (speed only, bijective for any grid size).  Workgroups are dealt round-robin over the 8 XCDs, each with a
// private L2.  With the natural order the 8 heads of one window -- whose 144-byte K / V rows share 128-byte lines of the interleaved
// [token][3 x heads x 72] qkv buffer -- land on 8 different XCDs and every line is fetched twice (PMC, r03: 413 MB read by the 16 x 16
// window launch against 226 MB of q + k + v), and the query groups that stream the SAME K / V of a global-attention head re-fetch it
// through 8 L2s.  Here XCD x owns a contiguous run of logical workgroups, walked in dispatch order: the heads of a window and the query
// groups of a head are neighbours in space (one L2) and in time.
__device__ __forceinline__ int xcd_order(int wg, int nwg, int on = 1) {
  if (!on) return wg;
  const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, j = wg >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

// element offset of token t of batch entry b (window mode: b enumerates windows of an image grid)
__device__ __forceinline__ long long tok_off(int b, int t, long long sb, long long st, int win, int gh, int gw) {
  if (win <= 0) return (long long)b * sb + (long long)t * st;
  const int wpr = gw / win, wpc = gh / win;
  const int img = b / (wpr * wpc);
  const int wi = b - img * (wpr * wpc);
  const int wy = wi / wpr, wx = wi - wy * wpr;
  const int ty = t / win, tx = t - ty * win;
  const long long pix = ((long long)img * gh + (wy * win + ty)) * gw + (wx * win + tx);
  return pix * st;
}

// tok_off() with the host's multipliers (window mode only): dwin divides by `win` (the side of the window t is counted in)
__device__ __forceinline__ long long tok_off_fast(const AttnArgs& p, int b, int t, long long st, int win, int gh, int gw, const FastDiv& dwin) {
  const int img = (int)p.div_wpi.div((unsigned)b), wi = b - img * p.wpi;
  const int wy = (int)p.div_wpr.div((unsigned)wi), wx = wi - wy * p.wpr;
  const int ty = (int)dwin.div((unsigned)t), tx = t - ty * win;
  const long long pix = ((long long)img * gh + (wy * win + ty)) * gw + (wx * win + tx);
  return pix * st;
}

template <int DQKP, int DVP, int GS>
__global__ __launch_bounds__(256) void attn_f16_kernel(const AttnArgs p) {
  constexpr int KROW = DQKP * 2 + 16;          // K tile row stride (bytes): odd multiple of 16
  constexpr int VROW = 32 * 2 + 8;             // V^T tile row stride (bytes): 72
  constexpr int KTILE = 32 * KROW;
  constexpr int VTILE = DVP * VROW;
  constexpr int NGRP = 256 / GS;               // loader groups per workgroup
  constexpr int QS = DQKP / 16;                // k16 steps of the S product
  constexpr int DT = DVP / 32;                 // 32-row tiles of O^T
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int lr = lane & 31, lh = lane >> 5;
  const int grp = tid / GS, gt = tid % GS;
  char* const Ks = smem + grp * (KTILE + VTILE);
  char* const Vs = Ks + KTILE;

  // work item of this wave
  int item, qt;
  if (GS == 256) {                             // workgroup = 4 consecutive q-tiles of one (b, h)
    const int qgroups = (p.qtiles + 3) / 4;
    const int bh = blockIdx.x / qgroups;
    qt = (blockIdx.x - bh * qgroups) * 4 + wv;
    item = bh;
  } else {
    const int it = blockIdx.x * 4 + wv;
    item = it / p.qtiles;
    qt = it - item * p.qtiles;
  }
  const int nbh = p.B * p.heads;
  const bool live = item < nbh && qt < p.qtiles;
  const int itc = item < nbh ? item : nbh - 1;
  const int b = itc / p.heads, h = itc - b * p.heads;
  // loader group's (b, h): for GS == 64 it is the wave's own; for GS == 256 all waves agree
  const int qwin = p.q_pool ? p.win / 2 : p.win;

  // ---- Q fragments (B operand of S^T = K Q^T): lane (q = lr, half lh) holds Q[q][16s + 8lh .. +7]
  const int qi = qt * 32 + lr;
  const bool q_ok = live && qi < p.Nq;
  u32x4 qf[QS];
  {
#pragma unroll
    for (int s = 0; s < QS; ++s) {
      const int d0 = 16 * s + 8 * lh;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (q_ok && d0 < p.dqk) {
        if (!p.q_pool) {
          const long long off = tok_off(b / p.q_bdiv, qi, p.q_sb, p.q_st, p.win, p.grid_h, p.grid_w) + (long long)h * p.q_sh + d0;
          v = *reinterpret_cast<const u32x4*>(p.q + off * 2);
        } else {                               // q = 2x2 max-pool of the window's projected q tokens
          const int py = qi / qwin, px = qi - py * qwin;
          f16x8 m;
#pragma unroll
          for (int e = 0; e < 8; ++e) m[e] = (f16)(CVMI_LOWEST16);
#pragma unroll
          for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
              const int t = (2 * py + dy) * p.win + 2 * px + dx;
              const long long off = tok_off(b, t, p.q_sb, p.q_st, p.win, p.grid_h, p.grid_w) + (long long)h * p.q_sh + d0;
              const f16x8 x = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(p.q + off * 2));
#pragma unroll
              for (int e = 0; e < 8; ++e) m[e] = x[e] > m[e] ? x[e] : m[e];
            }
          v = __builtin_bit_cast(u32x4, m);
        }
      }
      qf[s] = v;
    }
  }

  f32x16 oacc[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float c = p.scale * 1.44269504088896340736f;   // softmax in base 2

  // loader identity: group grp loads for item of wave (grp * GS / 64)
  int lb, lhd;
  {
    int litem;
    if (GS == 256) litem = item; else litem = (blockIdx.x * 4 + grp) / p.qtiles;
    if (litem >= nbh) litem = nbh - 1;
    lb = litem / p.heads; lhd = litem - lb * p.heads;
  }

  const int nkt = (p.Nk + 31) / 32;
  // Staging: the next K / V^T tile is fetched into registers (branch-free, clamped addresses) BEFORE the MFMAs of
  // the current tile and written to LDS after them, so the global latency overlaps the compute.
  constexpr int KN = (32 * (DQKP / 8) + GS - 1) / GS, VN = (32 * (DVP / 8) + GS - 1) / GS;
  u32x4 kreg[KN], vreg[VN];
  bool kok[KN], vok[VN];
  // Loader addressing, hoisted: everything that depends only on the loader's (batch entry, head) is computed once --
  // tok_off() costs five integer divisions, i.e. > 100 VALU instructions per 16-byte chunk if left in the key loop.
  const char *kbase, *vbase;
  {
    long long korg, vorg;
    if (p.win > 0) {
      const long long pix0 = tok_off(lb, 0, 1, 1, p.win, p.grid_h, p.grid_w);      // window origin, in pixels
      korg = pix0 * p.k_st; vorg = pix0 * p.v_st;
    } else {
      korg = (long long)(lb / p.kv_bdiv) * p.k_sb; vorg = (long long)(lb / p.kv_bdiv) * p.v_sb;
    }
    kbase = p.k + (korg + (long long)lhd * p.k_sh) * 2;
    vbase = p.v + (vorg + (long long)lhd * p.v_sh) * 2;
  }
  const int kst = (int)p.k_st, vst = (int)p.v_st;
  auto key_pix = [&](int key) -> int {                     // pixel offset of a key token from the window origin
    if (p.win <= 0) return key;
    const int ty = (int)p.div_win.div((unsigned)key);
    return ty * p.grid_w + (key - ty * p.win);
  };
  // 32 consecutive keys advance the pixel offset by a constant when 32 is a whole number of window rows (or no window):
  // the per-chunk offsets are then "offset at tile 0 + kt * stride" (wave-uniform choice; windows 14 / 7 take the general path)
  // (shared-tile kernels only: the per-wave-tile kernels run 1-2 key tiles and cannot spare the registers)
  const bool linear = GS == 256 && (p.win <= 0 || (32 % p.win) == 0);
  const int tile_pix = p.win <= 0 ? 32 : (32 / (p.win > 0 ? p.win : 1)) * p.grid_w;
  constexpr int KNL = GS == 256 ? KN : 1, VNL = GS == 256 ? VN : 1;
  int koff0[KNL], voff0[VNL];
  if constexpr (GS == 256) {
#pragma unroll
    for (int i = 0; i < KN; ++i) {
      const int idx = gt + i * GS;
      const int row = idx / (DQKP / 8), ch = idx - row * (DQKP / 8);
      koff0[i] = key_pix(row) * kst + ch * 8;
    }
#pragma unroll
    for (int i = 0; i < VN; ++i) {
      const int idx = gt + i * GS;
      const int ch = idx / 32, key_l = idx - ch * 32;
      voff0[i] = key_pix(key_l) * vst + ch * 8;
    }
  }
  auto fetch = [&](int kt) {
#pragma unroll
    for (int i = 0; i < KN; ++i) {
      const int idx = gt + i * GS;
      const int row = idx / (DQKP / 8), ch = idx - row * (DQKP / 8);
      const int key = kt * 32 + row;
      const bool ok = idx < 32 * (DQKP / 8) && key < p.Nk && ch * 8 < p.dqk;
      const int off = linear ? koff0[GS == 256 ? i : 0] + kt * tile_pix * kst : key_pix(key) * kst + ch * 8;
      kreg[i] = *reinterpret_cast<const u32x4*>(kbase + (long long)(ok ? off : 0) * 2);
      kok[i] = ok;
    }
#pragma unroll
    for (int i = 0; i < VN; ++i) {
      const int idx = gt + i * GS;
      const int ch = idx / 32, key_l = idx - ch * 32;       // consecutive threads -> consecutive keys
      const int key = kt * 32 + key_l;
      const bool ok = idx < 32 * (DVP / 8) && key < p.Nk && ch * 8 < p.dv;
      const int off = linear ? voff0[GS == 256 ? i : 0] + kt * tile_pix * vst : key_pix(key) * vst + ch * 8;
      vreg[i] = *reinterpret_cast<const u32x4*>(vbase + (long long)(ok ? off : 0) * 2);
      vok[i] = ok;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < KN; ++i) {
      const int idx = gt + i * GS;
      const int row = idx / (DQKP / 8), ch = idx - row * (DQKP / 8);
      if (idx < 32 * (DQKP / 8)) *reinterpret_cast<u32x4*>(Ks + row * KROW + ch * 16) = kok[i] ? kreg[i] : u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int i = 0; i < VN; ++i) {
      const int idx = gt + i * GS;
      const int ch = idx / 32, key_l = idx - ch * 32;
      if (idx < 32 * (DVP / 8)) {
        const f16x8 v = __builtin_bit_cast(f16x8, vok[i] ? vreg[i] : u32x4{0u, 0u, 0u, 0u});
#pragma unroll
        for (int e = 0; e < 8; ++e) *reinterpret_cast<f16*>(Vs + (ch * 8 + e) * VROW + key_l * 2) = v[e];
      }
    }
  };
  constexpr bool PREFETCH = (GS == 256);        // per-wave tiles (GS = 64) would need 12 staging registers x 4: not worth a wave of occupancy
  if constexpr (PREFETCH) {
    fetch(0);
    commit();
    __syncthreads();
  }
  for (int kt = 0; kt < nkt; ++kt) {
    if constexpr (PREFETCH) {
      if (kt + 1 < nkt) fetch(kt + 1);
    } else {
      __syncthreads();                         // previous tile fully consumed
      fetch(kt);
      commit();
      __syncthreads();
    }

    // ---- S^T tile: 32 keys x 32 queries ------------------------------------------------------------
    f32x16 sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < QS; ++s) {
      const f16x8 kf = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(Ks + lr * KROW + s * 32 + lh * 16));
      sacc = CVMI_MFMA_32X32X16(kf, __builtin_bit_cast(f16x8, qf[s]), sacc, 0, 0, 0);
    }
    // mask keys past Nk, running max
    if (kt * 32 + 32 > p.Nk) {                    // ragged last tile only (wave-uniform)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (key >= p.Nk) sacc[r] = -INFINITY;
      }
    }
    float mxa = max3f(sacc[0], sacc[1], sacc[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) mxa = max3f(mxa, sacc[r], sacc[r + 1]);
    const float mx = max3f(mxa, sacc[15], sacc[15]);
    const float m_new = max3f(m_run, mx, __shfl_xor(mx, 32));
    // raw v_exp_f32: arguments are <= 0, so exp2f's denormal-range rescue (5 extra instructions per value) buys nothing
    const float mc = m_new * c;
    const float alpha = __builtin_amdgcn_exp2f(fmaf(m_run, c, -mc));
    float psum = 0.f;
    f16x8 pf[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pv = __builtin_amdgcn_exp2f(fmaf(sacc[r], c, -mc));
      psum += pv;
      pf[r >> 3][r & 7] = (f16)pv;
    }
    psum += __shfl_xor(psum, 32);
    l_run = l_run * alpha + psum;
    const float m_prev = m_run;
    m_run = m_new;
    if (__any(m_new != m_prev)) {                 // alpha == 1 in every lane otherwise (wave-uniform branch)
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;
    }
    // ---- O^T += V^T P^T ---------------------------------------------------------------------------
#pragma unroll
    for (int t = 0; t < DT; ++t) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const char* vp = Vs + (t * 32 + lr) * VROW + (16 * s + 4 * lh) * 2;
        const u32x2 lo = *reinterpret_cast<const u32x2*>(vp);
        const u32x2 hi = *reinterpret_cast<const u32x2*>(vp + 16);
        const u32x4 vv = {lo[0], lo[1], hi[0], hi[1]};
        oacc[t] = CVMI_MFMA_32X32X16(__builtin_bit_cast(f16x8, vv), pf[s], oacc[t], 0, 0, 0);
      }
    }
    if constexpr (PREFETCH) {
      __syncthreads();                         // every wave is done reading this tile
      if (kt + 1 < nkt) commit();
      __syncthreads();
    }
  }

  // ---- normalise and store O[q][d] (lane holds runs of 4 consecutive d) ---------------------------
  if (q_ok) {
    const float inv = 1.f / l_run;
    long long obase;
    if (p.win > 0) {
      const int ow = p.q_pool ? p.win / 2 : p.win, ogh = p.q_pool ? p.grid_h / 2 : p.grid_h, ogw = p.q_pool ? p.grid_w / 2 : p.grid_w;
      obase = tok_off(b, qi, p.o_sb, p.o_st, ow, ogh, ogw);
    } else {
      obase = (long long)b * p.o_sb + (long long)qi * p.o_st;
    }
    obase += (long long)h * p.o_sh;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = t * 32 + 8 * g + 4 * lh;
        if (d0 < p.dv) {
          f16x4 ov;
#pragma unroll
          for (int e = 0; e < 4; ++e) ov[e] = (f16)(oacc[t][4 * g + e] * inv);
          *reinterpret_cast<f16x4*>(p.o + (obase + d0) * 2) = ov;
        }
      }
  }
}

// ---- long-sequence variant: 64-key tiles, V kept row-major and transposed by the LDS read --------------------------
// Same products and online softmax as attn_f16_kernel<.., 256> (workgroup = 4 q-tiles of one (b, h) sharing the K / V
// tile), but (1) a tile holds 64 keys, so the two barriers, the staging bookkeeping and the accumulator rescale are
// paid once per 24 MFMAs instead of once per 12, and (2) V is written to LDS as it comes from HBM ([key][d], 16-byte
// stores) and the V^T operand is gathered by ds_read_b64_tr_b16 (per 16-lane group: a 4-key x 16-d block delivered
// column-major), which removes the 8 scalar 2-byte LDS stores per staged chunk the transposed image needed.
// V row stride: a 32-lane half reads 4 consecutive key rows x 64 bytes, conflict-free when (stride mod 256) is 64 or 192.
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
typedef int v8i32 __attribute__((ext_vector_type(8)));
// four floats -> four OCP e4m3 bytes (a = byte 0).  v_cvt_pk_fp8_f32 returns NaN, not the largest value, on overflow: callers bound their inputs.
__device__ __forceinline__ unsigned pack4_e4m3(float a, float b, float c, float d) {
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (unsigned)w;
}
template <int DQKP, int DVP, int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void attn64_kernel(const AttnArgs p) {
  constexpr int NTH = NW * 64;                            // NW = 8: one K / V tile serves 256 queries
  constexpr int KROW = DQKP * 2 + 16;
  constexpr int VRS = DVP * 2 + (((DVP * 2) % 256 == 64 || (DVP * 2) % 256 == 192) ? 0 : 64);
  constexpr int KTILE = 64 * KROW;
  constexpr int QS = DQKP / 16, DT = DVP / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Ks = smem;
  char* const Vs = smem + KTILE;

  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int lr = lane & 31, lh = lane >> 5;
  const int qgroups = (p.qtiles + NW - 1) / NW;
  const int wgx = xcd_order((int)blockIdx.x, (int)gridDim.x, p.xcd);
  const int item = wgx / qgroups;                        // (b, h) of the workgroup
  const int qt = (wgx - item * qgroups) * NW + wv;
  const bool live = qt < p.qtiles;
  const int b = item / p.heads, h = item - b * p.heads;
  const int qwin = p.q_pool ? p.win / 2 : p.win;

  const int qi = qt * 32 + lr;
  const bool q_ok = live && qi < p.Nq;
  // element offset of this lane's query token (pooled: of the top-left token of its 2 x 2 block), computed once
  long long qoff0;
  {
    const int qc = q_ok ? qi : 0;
    int t = qc;
    if (p.q_pool) { const int py = qc / qwin, px = qc - py * qwin; t = (2 * py) * p.win + 2 * px; }
    qoff0 = tok_off(b / p.q_bdiv, t, p.q_sb, p.q_st, p.win, p.grid_h, p.grid_w) + (long long)h * p.q_sh;
  }
  u32x4 qf[QS];
#pragma unroll
  for (int s = 0; s < QS; ++s) {
    const int d0 = 16 * s + 8 * lh;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (q_ok && d0 < p.dqk) {
      if (!p.q_pool) {
        v = *reinterpret_cast<const u32x4*>(p.q + (qoff0 + d0) * 2);
      } else {
        f16x8 m;
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = (f16)(CVMI_LOWEST16);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
          for (int dx = 0; dx < 2; ++dx) {
            const long long off = qoff0 + ((long long)dy * p.grid_w + dx) * p.q_st + d0;   // pooled: the 2 x 2 block's tokens are grid neighbours
            const f16x8 x = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(p.q + off * 2));
#pragma unroll
            for (int e = 0; e < 8; ++e) m[e] = x[e] > m[e] ? x[e] : m[e];
          }
        v = __builtin_bit_cast(u32x4, m);
      }
    }
    qf[s] = v;
  }

  f32x16 oacc[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float c = p.scale * 1.44269504088896340736f;

  // loader addressing (see attn_f16_kernel): bases once, per-chunk offsets at tile 0, a constant stride per 64-key tile
  const char *kbase, *vbase;
  {
    long long korg, vorg;
    if (p.win > 0) {
      const long long pix0 = tok_off(b, 0, 1, 1, p.win, p.grid_h, p.grid_w);
      korg = pix0 * p.k_st; vorg = pix0 * p.v_st;
    } else {
      korg = (long long)(b / p.kv_bdiv) * p.k_sb; vorg = (long long)(b / p.kv_bdiv) * p.v_sb;
    }
    kbase = p.k + (korg + (long long)h * p.k_sh) * 2;
    vbase = p.v + (vorg + (long long)h * p.v_sh) * 2;
  }
  const int kst = (int)p.k_st, vst = (int)p.v_st;
  auto key_pix = [&](int key) -> int {
    if (p.win <= 0) return key;
    const int ty = (int)p.div_win.div((unsigned)key);
    return ty * p.grid_w + (key - ty * p.win);
  };
  const bool linear = p.win <= 0 || (64 % p.win) == 0;
  const int tile_pix = p.win <= 0 ? 64 : (64 / (p.win > 0 ? p.win : 1)) * p.grid_w;
  constexpr int KCH = DQKP / 8, VCH = DVP / 8;
  constexpr int KN = (64 * KCH + NTH - 1) / NTH, VN = (64 * VCH + NTH - 1) / NTH;
  int koff0[KN], voff0[VN];
#pragma unroll
  for (int i = 0; i < KN; ++i) {
    const int idx = tid + i * NTH, row = idx / KCH, ch = idx - row * KCH;
    koff0[i] = key_pix(row < 64 ? row : 0) * kst + ch * 8;
  }
#pragma unroll
  for (int i = 0; i < VN; ++i) {
    const int idx = tid + i * NTH, row = idx / VCH, ch = idx - row * VCH;
    voff0[i] = key_pix(row < 64 ? row : 0) * vst + ch * 8;
  }
  u32x4 kreg[KN], vreg[VN];
  unsigned okmask = 0;                                    // bit i: K chunk i valid; bit 8 + i: V chunk i valid
  auto fetch = [&](int kt) {
    okmask = 0;
#pragma unroll
    for (int i = 0; i < KN; ++i) {
      const int idx = tid + i * NTH, row = idx / KCH, ch = idx - row * KCH;
      const int key = kt * 64 + row;
      const bool ok = idx < 64 * KCH && key < p.Nk && ch * 8 < p.dqk;
      const int off = linear ? koff0[i] + kt * tile_pix * kst : key_pix(key) * kst + ch * 8;
      kreg[i] = *reinterpret_cast<const u32x4*>(kbase + (long long)(ok ? off : 0) * 2);
      okmask |= ok ? 1u << i : 0u;
    }
#pragma unroll
    for (int i = 0; i < VN; ++i) {
      const int idx = tid + i * NTH, row = idx / VCH, ch = idx - row * VCH;
      const int key = kt * 64 + row;
      const bool ok = idx < 64 * VCH && key < p.Nk && ch * 8 < p.dv;
      const int off = linear ? voff0[i] + kt * tile_pix * vst : key_pix(key) * vst + ch * 8;
      vreg[i] = *reinterpret_cast<const u32x4*>(vbase + (long long)(ok ? off : 0) * 2);
      okmask |= ok ? 1u << (8 + i) : 0u;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < KN; ++i) {
      const int idx = tid + i * NTH, row = idx / KCH, ch = idx - row * KCH;
      if (idx < 64 * KCH) *reinterpret_cast<u32x4*>(Ks + row * KROW + ch * 16) = (okmask >> i) & 1u ? kreg[i] : u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int i = 0; i < VN; ++i) {
      const int idx = tid + i * NTH, row = idx / VCH, ch = idx - row * VCH;
      if (idx < 64 * VCH) *reinterpret_cast<u32x4*>(Vs + row * VRS + ch * 16) = (okmask >> (8 + i)) & 1u ? vreg[i] : u32x4{0u, 0u, 0u, 0u};
    }
  };
  // V^T fragment gather: group g = lane / 16 holds d columns 16 (g & 1) .., keys 4 lh .. (the S^T accumulator's key order:
  // k-slot 8 lh + j of a 16-key step is key 4 lh + (j & 3) + 8 (j >> 2)); lane 4 q + pp of the group addresses block row q,
  // columns 4 pp .. 4 pp + 3
  const int li = lane & 15;
  const char* const vt = Vs + (4 * lh + (li >> 2)) * VRS + (16 * (lr >> 4) + 4 * (li & 3)) * 2;
  const char* const kq = Ks + lr * KROW + lh * 16;

  const int nkt = (p.Nk + 63) / 64;
  fetch(0);
  commit();
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) fetch(kt + 1);
    // ---- S^T: 64 keys x 32 queries
    f32x16 sacc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[u][r] = 0.f;
#pragma unroll
    for (int s = 0; s < QS; ++s)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const f16x8 kf = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(kq + u * 32 * KROW + s * 32));
        sacc[u] = CVMI_MFMA_32X32X16(kf, __builtin_bit_cast(f16x8, qf[s]), sacc[u], 0, 0, 0);
      }
    if (kt * 64 + 64 > p.Nk) {                             // ragged last tile only (wave-uniform)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * 64 + u * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (key >= p.Nk) sacc[u][r] = -INFINITY;
        }
    }
    float mxa = max3f(sacc[0][0], sacc[1][0], sacc[0][8]), mxb = max3f(sacc[1][8], sacc[0][1], sacc[1][1]);      // two chains of v_max3_f32
#pragma unroll
    for (int r = 2; r < 8; ++r) { mxa = max3f(mxa, sacc[0][r], sacc[1][r]); mxb = max3f(mxb, sacc[0][r + 7], sacc[1][r + 7]); }
    const float mx = max3f(mxa, mxb, max3f(sacc[0][15], sacc[1][15], mxa));
    const float m_new = max3f(m_run, mx, __shfl_xor(mx, 32));
    const float mc = m_new * c;
    const float alpha = __builtin_amdgcn_exp2f(fmaf(m_run, c, -mc));
    float psum = 0.f;
    f16x8 pf[2][2];
    psum = softmax_tiles<true>(sacc, c, mc, pf);
    psum += __shfl_xor(psum, 32);
    l_run = l_run * alpha + psum;
    const float m_prev = m_run;
    m_run = m_new;
    if (__any(m_new != m_prev)) {
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;
    }
    // ---- O^T += V^T P^T
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < DT; ++t) {
          const char* a0 = vt + (u * 32 + s * 16) * VRS + t * 64;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 8 * VRS));
          const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
          const u32x4 vv = {l2[0], l2[1], h2[0], h2[1]};
          oacc[t] = CVMI_MFMA_32X32X16(__builtin_bit_cast(f16x8, vv), pf[u][s], oacc[t], 0, 0, 0);
        }
    __syncthreads();
    if (kt + 1 < nkt) commit();
    __syncthreads();
  }

  if (q_ok) {
    const float inv = 1.f / l_run;
    long long obase;
    if (p.win > 0) {
      const int ow = p.q_pool ? p.win / 2 : p.win, ogh = p.q_pool ? p.grid_h / 2 : p.grid_h, ogw = p.q_pool ? p.grid_w / 2 : p.grid_w;
      obase = tok_off(b, qi, p.o_sb, p.o_st, ow, ogh, ogw);
    } else {
      obase = (long long)b * p.o_sb + (long long)qi * p.o_st;
    }
    obase += (long long)h * p.o_sh;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = t * 32 + 8 * g + 4 * lh;
        if (d0 < p.dv) {
          f16x4 ov;
#pragma unroll
          for (int e = 0; e < 4; ++e) ov[e] = (f16)(oacc[t][4 * g + e] * inv);
          *reinterpret_cast<f16x4*>(p.o + (obase + d0) * 2) = ov;
        }
      }
  }
}

template <int DQKP, int DVP, int NW>
int launch_attn64(const AttnArgs& a, hipStream_t stream) {
  constexpr int KROW = DQKP * 2 + 16;
  constexpr int VRS = DVP * 2 + (((DVP * 2) % 256 == 64 || (DVP * 2) % 256 == 192) ? 0 : 64);
  constexpr size_t lds = (size_t)64 * (KROW + VRS);
  const long long blocks = (long long)a.B * a.heads * ((a.qtiles + NW - 1) / NW);
  CVMI_CHECK(blocks > 0 && blocks < (1ll << 31), "attention: bad grid");
  cvmi_note_kernel("attn64_kernel<%d, %d, %d>", DQKP, DVP, NW);
  hipLaunchKernelGGL((attn64_kernel<DQKP, DVP, NW>), dim3((unsigned)blocks), dim3(NW * 64), lds, stream, a);
  CVMI_LAUNCH_CHECK();
  return 0;
}

// hipcc places NO `s_waitcnt vmcnt(0)` in front of an s_barrier for in-flight LDS-DMA writes (global_load_lds): __syncthreads()'s fence only
// produces one when an ordinary load happens to be outstanding -- which is how these kernels passed until an unrelated change moved their
// Q-fragment loads.  Every barrier that publishes DMA'd data is therefore preceded by this explicit wait.
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }


// ---- head_dim 72 kernels below: which key a score row stands for --------------------------------------------------------
// K and V live in LDS as unpadded 144-byte rows (what the LDS-DMA writes).  The V^T operand is gathered by ds_read_b64_tr_b16: a
// 32-lane half reads FOUR key rows x 64 bytes, and with consecutive rows at a 36-dword stride two of the four fall on the same
// banks (PMC: SQ_LDS_BANK_CONFLICT = 1/3 of SQ_LDS_IDX_ACTIVE in all three kernels).  Rows 4 apart do not collide (4 x 36 = 144 =
// 16 mod 64: the four 16-bank spans tile the 64 banks).  So score row i of a 32-key tile stands for key key_perm72(i): lane lr
// feeds K row key_perm72(lr) to the QK^T MFMA (still one distinct 4-bank span per lane of a ds_read_b128 group), the softmax does not
// care, and the P fragment's k index (16 s + 8 (j >> 2) + 4 lh + (j & 3)) then wants V rows 16 s + lh + 4 (j & 3) + 2 (j >> 2):
// two transposing reads whose four rows are 4 apart, the second 2 rows below the first.
__device__ __forceinline__ int key_perm72(int i) {            // i = 4 a + b  ->  16 (a >> 2) + 4 b + (a & 3)
  const int a = i >> 2;
  return 16 * (a >> 2) + 4 * (i & 3) + (a & 3);
}

// ---- 256-key windows of head_dim 72 (Hiera stage 3, 16 x 16): K and V of the whole window resident in LDS --------------
// A (window, head) has only four 64-key tiles; with tile-by-tile staging the Q / first-tile latency, eight barriers and the
// staging bookkeeping cost more than the 88 MFMAs.  Here every wave DMAs its share of the window's K and V rows
// (global_load_lds, 16 B per lane, per-lane source = window token row, LDS image = unpadded 144-byte rows: a 16-lane
// group of ds_read_b128 at that stride touches all 64 banks once) and after ONE barrier runs the four key tiles back to
// back with no further synchronisation.  Row reads that run past column 72 (5th k-step, third 32-wide d tile) fetch the
// next row's finite data against zero Q columns / unstored output rows.  4 waves = 128 queries per workgroup; 72 KiB of
// LDS, two workgroups per CU.
// AV8 = true (cvmi_attn_desc.av_fp8, BASELINE configs[4] "fp8 MFMA attention"): the O^T += V^T P^T contraction -- whose K axis is the KEY axis,
// a whole multiple of 64 here -- runs on the block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 (2 x the bf16 rate per clock): one instruction
// per (64-key tile, 32-wide d tile) instead of four.  V is quantised ONCE per workgroup: after the window has landed, every thread builds
// three 16-byte pieces of the e4m3 V^T operand image from the fp16 rows (the same transposing reads the 16-bit path feeds its MFMAs with,
// f32 clamp to +-448 because the conversion returns NaN, not the maximum, on overflow), a barrier, and the pieces overwrite the 16-bit V
// image in place, lane-linear (conflict-free ds_read_b128 of whole operands).  P is quantised from the fp32 softmax as e4m3 of p * 2^8
// (p <= 1: the row maximum is exactly 256, no overflow) with the block scale 2^-8 in the MFMA's scale operand; row sums stay fp32 sums of
// the unquantised p.  Operand maps measured on gfx950 (tools/probe/fp8_probe.hip, fp8_scale_probe.hip): lane (r = l & 31, h = l >> 5) holds
// A[row r] / B[col r] bytes j = 0..31 which pair with the SAME (h, j) of the other operand; scale byte 0 of lane (r, h) scales that row's
// bytes 16 h .. 16 h + 15 of both lane halves; C / D as every 32 x 32 MFMA.
template <int NW, bool AV8 = false, bool QL = false>
__global__ __launch_bounds__(NW * 64, (AV8 || NW != 8) ? 2 : 4) void attn_res256_kernel(const AttnArgs p) {
  constexpr int ROW = 144, NK = 256, QS = 5, DT = 3, CH = 9;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Ks = smem;
  char* const Vs = smem + NK * ROW;
  const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int lr = lane & 31, lh = lane >> 5;
  const int qgroups = (p.qtiles + NW - 1) / NW;
  const int wgx = xcd_order((int)blockIdx.x, (int)gridDim.x, p.xcd);
  const int item = qgroups == 1 ? wgx : wgx / qgroups;      // (16 x 16 windows, 8 waves: one group per item)
  const int qt = (wgx - item * qgroups) * NW + wv;
  const bool live = qt < p.qtiles;
  const int b = (int)p.div_heads.div((unsigned)item), h = item - b * p.heads;
  const int qwin = p.q_pool ? p.win / 2 : p.win;

  // ---- DMA the window's K and V: chunk L -> (key row L / 9, 16-byte chunk L % 9); 36 wave-instructions per matrix
  {
    long long korg, vorg;
    if (p.win > 0) {
      const long long pix0 = tok_off_fast(p, b, 0, 1, p.win, p.grid_h, p.grid_w, p.div_win);
      korg = pix0 * p.k_st; vorg = pix0 * p.v_st;
    } else {
      korg = (long long)b * p.k_sb; vorg = (long long)b * p.v_sb;
    }
    const char* kbase = p.k + (korg + (long long)h * p.k_sh) * 2;
    const char* vbase = p.v + (vorg + (long long)h * p.v_sh) * 2;
#pragma unroll
    for (int j = 0; j < (36 + NW - 1) / NW; ++j) {
      if (j * NW + wv >= 36 || (p.diag & 2)) break;           // 36 wave-instructions per matrix (wave-uniform)
      const int L = (j * NW + wv) * 64 + lane;
      const int row = L / CH, ch = L - row * CH;
      int pix = row;
      if (p.win > 0) { const int ty = (int)p.div_win.div((unsigned)row); pix = ty * p.grid_w + (row - ty * p.win); }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kbase + ((long long)pix * p.k_st + ch * 8) * 2),
                                       (__attribute__((address_space(3))) void*)(Ks + (j * NW + wv) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vbase + ((long long)pix * p.v_st + ch * 8) * 2),
                                       (__attribute__((address_space(3))) void*)(Vs + (j * NW + wv) * 1024), 16, 0, 0);
    }
  }
  // ---- Q fragments while the DMA flies
  const int qi = qt * 32 + lr;
  const bool q_ok = live && qi < p.Nq;
  // element offset of this lane's query token (pooled: of the top-left token of its 2 x 2 block), computed once
  long long qoff0;
  {
    const int qc = q_ok ? qi : 0;
    int t = qc;
    if (p.q_pool) { const int py = (int)p.div_ow.div((unsigned)qc), px = qc - py * qwin; t = (2 * py) * p.win + 2 * px; }
    qoff0 = (p.win > 0 ? tok_off_fast(p, b, t, p.q_st, p.win, p.grid_h, p.grid_w, p.div_win) : (long long)b * p.q_sb + (long long)t * p.q_st) + (long long)h * p.q_sh;
  }
  u32x4 qf[QS];
#pragma unroll
  for (int s = 0; s < QS; ++s) {
    const int d0 = 16 * s + 8 * lh;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (q_ok && d0 < p.dqk) {
      if (!p.q_pool) {
        v = *reinterpret_cast<const u32x4*>(p.q + (qoff0 + d0) * 2);
      } else {
        f16x8 m;
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = (f16)(CVMI_LOWEST16);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
          for (int dx = 0; dx < 2; ++dx) {
            const long long off = qoff0 + ((long long)dy * p.grid_w + dx) * p.q_st + d0;   // pooled: the 2 x 2 block's tokens are grid neighbours
            const f16x8 x = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(p.q + off * 2));
#pragma unroll
            for (int e = 0; e < 8; ++e) m[e] = x[e] > m[e] ? x[e] : m[e];
          }
        v = __builtin_bit_cast(u32x4, m);
      }
    }
    qf[s] = v;
  }
  f32x16 oacc[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float c = p.scale * 1.44269504088896340736f;
  const int li = lane & 15;
  const char* const vt = Vs + (lh + 4 * (li >> 2)) * ROW + (16 * (lr >> 4) + 4 * (li & 3)) * 2;      // key rows 4 apart: see key_perm72
  const char* const kq = Ks + key_perm72(lr) * ROW + lh * 16;
  // Row sums and (QL) the running maximum on the matrix pipe: as in attn_dma72_kernel, which documents both.  The constants live behind V.
  constexpr int ONES = 2 * NK * ROW + 256;
  const bool ones_lane = !AV8 && (lr >> 4) == 0 && (li & 3) == 2;
  if constexpr (!AV8) {
    if (tid < 16) {                                           // two copies: see ONES_V in attn_dma72_kernel
      const int t8 = tid & 7, off = ((t8 >> 2) * 32 + ((t8 >> 1) & 1) * 16 + (t8 & 1) * 2) * ROW;
      *reinterpret_cast<u32x2*>(smem + ONES + (tid < 8 ? 144 : 32) + off) = (u32x2){CVMI_ONE16X2 & 0xFFFFu, 0u};
    } else if (QL && tid < 18) {
      *reinterpret_cast<u32x4*>(smem + ONES + 16 + (tid - 16) * 32 * ROW) = (u32x4){CVMI_ONE16X2 & 0xFFFFu, 0u, 0u, 0u};
    }
  }
  float m_ref = 0.f;                                          // QL: the reference maximum held (negated) in the Q operand
  dma_wait();                                               // every wave waits for its OWN LDS-DMA pieces ...
  __syncthreads();                                          // ... and the barrier publishes the window

  if constexpr (AV8) {
    static_assert(NW == 8, "the e4m3 V image is built by 8 waves x 3 pieces");
    // piece pc = wv + 8 i of 24: (key tile kt, d tile t, operand half u) = (pc / 6, (pc % 6) / 2, pc % 2); this lane's 16 bytes are
    // V[key(u, jj)][32 t + lr] for jj = 0..15 in the k order of the P operand (the two k-steps s = jj >> 3 of the 16-bit path)
    u32x4 piece[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int pc = wv + 8 * i, kt = pc / 6, t = (pc - 6 * kt) >> 1, u = pc & 1;
      unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int sgrp = 0; sgrp < 2; ++sgrp) {
        const char* a0 = vt + (kt * 64 + u * 32 + sgrp * 16) * ROW + t * 64;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 2 * ROW));
        const f16x4 l4 = __builtin_bit_cast(f16x4, lo), h4 = __builtin_bit_cast(f16x4, hi);
        float f[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { f[e] = (float)l4[e]; f[4 + e] = (float)h4[e]; }
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = __builtin_amdgcn_fmed3f(f[e], -448.f, 448.f);
        w[2 * sgrp] = pack4_e4m3(f[0], f[1], f[2], f[3]);
        w[2 * sgrp + 1] = pack4_e4m3(f[4], f[5], f[6], f[7]);
      }
      piece[i] = (u32x4){w[0], w[1], w[2], w[3]};
    }
    __syncthreads();                                        // every transposing read of the 16-bit V image is done ...
#pragma unroll
    for (int i = 0; i < 3; ++i) *reinterpret_cast<u32x4*>(Vs + (wv + 8 * i) * 1024 + lane * 16) = piece[i];
    __syncthreads();                                        // ... before the e4m3 image replaces it
  }

#pragma unroll 1
  for (int kc = 0; kc < ((p.diag & 1) ? 0 : NK / 64); ++kc) {
    f32x16 sacc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[u][r] = 0.f;
#pragma unroll
    for (int s = 0; s < QS; ++s)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const char* ka = (QL && s == 4 && lh) ? smem + ONES + 16 + u * 32 * ROW : kq + (kc * 64 + u * 32) * ROW + s * 32;      // QL: k = 72..79 read (1, 0, .., 0)
        const f16x8 kf = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(ka));
        sacc[u] = CVMI_MFMA_32X32X16(kf, __builtin_bit_cast(f16x8, qf[s]), sacc[u], 0, 0, 0);
      }
    float mxa = max3f(sacc[0][0], sacc[1][0], sacc[0][8]), mxb = max3f(sacc[1][8], sacc[0][1], sacc[1][1]);      // two chains of v_max3_f32
#pragma unroll
    for (int r = 2; r < 8; ++r) { mxa = max3f(mxa, sacc[0][r], sacc[1][r]); mxb = max3f(mxb, sacc[0][r + 7], sacc[1][r + 7]); }
    const float mx = max3f(mxa, mxb, max3f(sacc[0][15], sacc[1][15], mxa));
    f16x8 pf[2][2];
    u32x4 p8[2];                                            // AV8: this lane's 32 e4m3 bytes of P^T (tile u -> bytes 16 u .. 16 u + 15)
    if constexpr (QL) {
      static_assert(!AV8, "QL is the 16-bit form");
      const float top = xhalf_max(mx);                          // the tile's maximum RELATIVE to m_ref (the MFMA subtracted it)
      const bool grow = kc == 0 || top > p.defer;
      if (__any(grow)) {                                      // (a real branch: the first tile, then rare)
        const float m_new = grow ? (float)(f16)(m_ref + top) : m_ref;
        const float dlt = m_new - m_ref;
        m_ref = m_new;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int r = 0; r < 16; ++r) sacc[u][r] -= dlt;
        if (kc > 0) {                                           // (uniform; the accumulators are still zero in the first tile)
          const float alpha = __builtin_amdgcn_exp2f(-dlt);
#pragma unroll
          for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;
        }
        if (lh) {
          f16x8 qv = __builtin_bit_cast(f16x8, qf[4]);
          qv[0] = (f16)(-m_new);
          qf[4] = __builtin_bit_cast(u32x4, qv);
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) pf[u][r >> 3][r & 7] = (f16)__builtin_amdgcn_exp2f(sacc[u][r]);
    } else {
    const float m_top = fmaxf(m_run, xhalf_max(mx));
    const bool grow = (m_top - m_run) * c > (AV8 ? 0.f : p.defer);      // first tile: m_run = -inf -> true
    const float m_new = grow ? m_top : m_run;
    const float mc = m_new * c;
    const float alpha = grow ? __builtin_amdgcn_exp2f(fmaf(m_run, c, -mc)) : 1.0f;
    float psum = 0.f;
    if constexpr (AV8) {
      const float mc8 = mc - 8.f;                           // p * 2^8 through the exponent
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int r = 0; r < 16; r += 4) {
          float e[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) { e[i] = __builtin_amdgcn_exp2f(fmaf(sacc[u][r + i], c, -mc8)); psum += e[i]; }
          w[r >> 2] = pack4_e4m3(e[0], e[1], e[2], e[3]);
        }
        p8[u] = (u32x4){w[0], w[1], w[2], w[3]};
      }
      psum *= 0.00390625f;                                  // back to the scale of l_run (exact: a power of two)
    } else {
      softmax_tiles<false>(sacc, c, mc, pf);
    }
    if constexpr (AV8) {
      psum = xhalf_sum(psum);
      l_run = l_run * alpha + psum;
    }
    m_run = m_new;
    if (__any(grow)) {                                      // (a real branch: rare once the first tiles have set the reference)
      asm volatile("" ::: "memory");
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;
    }
    }
    if constexpr (AV8) {
      const v8i32 pb = {(int)p8[0][0], (int)p8[0][1], (int)p8[0][2], (int)p8[0][3], (int)p8[1][0], (int)p8[1][1], (int)p8[1][2], (int)p8[1][3]};
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        const char* va = Vs + ((kc * 3 + t) * 2) * 1024 + lane * 16;
        const u32x4 a_lo = *reinterpret_cast<const u32x4*>(va), a_hi = *reinterpret_cast<const u32x4*>(va + 1024);
        const v8i32 av = {(int)a_lo[0], (int)a_lo[1], (int)a_lo[2], (int)a_lo[3], (int)a_hi[0], (int)a_hi[1], (int)a_hi[2], (int)a_hi[3]};
        oacc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, pb, oacc[t], 0, 0, 0, 127, 0, 119);      // e4m3 x e4m3, scales 2^0 / 2^-8
      }
    } else {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int t = 0; t < DT; ++t) {
            const char* a0 = (t == 2 && ones_lane) ? smem + ONES + (lh ? 32 : 144) + (u * 32 + s * 16) * ROW     // (the lanes of d = 72..75: the row-sum constants)
                                                   : vt + (kc * 64 + u * 32 + s * 16) * ROW + t * 64;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 2 * ROW));
            const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
            const u32x4 vv = {l2[0], l2[1], h2[0], h2[1]};
            oacc[t] = CVMI_MFMA_32X32X16(__builtin_bit_cast(f16x8, vv), pf[u][s], oacc[t], 0, 0, 0);
          }
    }
  }
  if constexpr (!AV8) l_run = __shfl(oacc[2][4], lr);        // row d = 72 (tile 2, row 8): register 4 of the lanes of half 0, column = query
  if (q_ok) {
    const float inv = 1.f / l_run;
    long long obase;
    if (p.win > 0) {
      const int ow = p.q_pool ? p.win / 2 : p.win, ogh = p.q_pool ? p.grid_h / 2 : p.grid_h, ogw = p.q_pool ? p.grid_w / 2 : p.grid_w;
      obase = tok_off_fast(p, b, qi, p.o_st, ow, ogh, ogw, p.div_ow);
    } else {
      obase = (long long)b * p.o_sb + (long long)qi * p.o_st;
    }
    obase += (long long)h * p.o_sh;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = t * 32 + 8 * g + 4 * lh;
        if (d0 < p.dv) {
          f16x4 ov;
#pragma unroll
          for (int e = 0; e < 4; ++e) ov[e] = (f16)(oacc[t][4 * g + e] * inv);
          *reinterpret_cast<f16x4*>(p.o + (obase + d0) * 2) = ov;
        }
      }
  }
}

// ---- 64-key windows of head_dim 72 (Hiera stages 1 and 4, and the q-pooled stage transitions that keep 64 keys):
//      two (window, head) items per workgroup, QT waves (= 32-query tiles) each, the item's K / V resident in LDS
template <int QT>
__global__ __launch_bounds__(QT * 128, 4) void attn_res64_kernel(const AttnArgs p) {
  constexpr int NW = QT;                                    // waves per item
  constexpr int ROW = 144, NK = 64, QS = 5, DT = 3, CH = 9;
  constexpr int ITEM_B = 2 * NK * ROW + 64;                // K + V of one item (+ slack for the last rows' over-reads)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, wvg = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int sub = wvg / QT, wv = wvg - sub * QT;            // item within the workgroup, wave within the item
  char* const Ks = smem + sub * ITEM_B;
  char* const Vs = Ks + NK * ROW;
  const int lr = lane & 31, lh = lane >> 5;
  const int nitems = p.B * p.heads;
  const int item_raw = xcd_order((int)blockIdx.x, (int)gridDim.x, p.xcd) * 2 + sub;
  const int item = item_raw < nitems ? item_raw : nitems - 1;
  const int qt = wv;
  const bool live = item_raw < nitems && qt < p.qtiles;
  const int b = (int)p.div_heads.div((unsigned)item), h = item - b * p.heads;      // (host-computed multipliers: as attn_res256_kernel, r04)
  const int qwin = p.q_pool ? p.win / 2 : p.win;

  // ---- DMA the window's K and V: chunk L -> (key row L / 9, 16-byte chunk L % 9); 36 wave-instructions per matrix
  {
    long long korg, vorg;
    if (p.win > 0) {
      const long long pix0 = tok_off_fast(p, b, 0, 1, p.win, p.grid_h, p.grid_w, p.div_win);
      korg = pix0 * p.k_st; vorg = pix0 * p.v_st;
    } else {
      korg = (long long)b * p.k_sb; vorg = (long long)b * p.v_sb;
    }
    const char* kbase = p.k + (korg + (long long)h * p.k_sh) * 2;
    const char* vbase = p.v + (vorg + (long long)h * p.v_sh) * 2;
#pragma unroll
    for (int j = 0; j < (9 + NW - 1) / NW; ++j) {
      if (j * NW + wv >= 9) break;                            // 9 wave-instructions per matrix (wave-uniform)
      const int L = (j * NW + wv) * 64 + lane;
      const int row = L / CH, ch = L - row * CH;
      int pix = row;
      if (p.win > 0) { const int ty = (int)p.div_win.div((unsigned)row); pix = ty * p.grid_w + (row - ty * p.win); }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kbase + ((long long)pix * p.k_st + ch * 8) * 2),
                                       (__attribute__((address_space(3))) void*)(Ks + (j * NW + wv) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vbase + ((long long)pix * p.v_st + ch * 8) * 2),
                                       (__attribute__((address_space(3))) void*)(Vs + (j * NW + wv) * 1024), 16, 0, 0);
    }
  }
  // ---- Q fragments while the DMA flies
  const int qi = qt * 32 + lr;
  const bool q_ok = live && qi < p.Nq;
  // element offset of this lane's query token (pooled: of the top-left token of its 2 x 2 block), computed once
  long long qoff0;
  {
    const int qc = q_ok ? qi : 0;
    int t = qc;
    if (p.q_pool) { const int py = (int)p.div_ow.div((unsigned)qc), px = qc - py * qwin; t = (2 * py) * p.win + 2 * px; }
    qoff0 = (p.win > 0 ? tok_off_fast(p, b, t, p.q_st, p.win, p.grid_h, p.grid_w, p.div_win) : (long long)b * p.q_sb + (long long)t * p.q_st) + (long long)h * p.q_sh;
  }
  u32x4 qf[QS];
#pragma unroll
  for (int s = 0; s < QS; ++s) {
    const int d0 = 16 * s + 8 * lh;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (q_ok && d0 < p.dqk) {
      if (!p.q_pool) {
        v = *reinterpret_cast<const u32x4*>(p.q + (qoff0 + d0) * 2);
      } else {
        f16x8 m;
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = (f16)(CVMI_LOWEST16);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
          for (int dx = 0; dx < 2; ++dx) {
            const long long off = qoff0 + ((long long)dy * p.grid_w + dx) * p.q_st + d0;   // pooled: the 2 x 2 block's tokens are grid neighbours
            const f16x8 x = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(p.q + off * 2));
#pragma unroll
            for (int e = 0; e < 8; ++e) m[e] = x[e] > m[e] ? x[e] : m[e];
          }
        v = __builtin_bit_cast(u32x4, m);
      }
    }
    qf[s] = v;
  }
  f32x16 oacc[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float c = p.scale * 1.44269504088896340736f;
  const int li = lane & 15;
  const char* const vt = Vs + (lh + 4 * (li >> 2)) * ROW + (16 * (lr >> 4) + 4 * (li & 3)) * 2;      // key rows 4 apart: see key_perm72
  const char* const kq = Ks + key_perm72(lr) * ROW + lh * 16;
  dma_wait();                                               // every wave waits for its OWN LDS-DMA pieces ...
  __syncthreads();                                          // ... and the barrier publishes the window

  {
    constexpr int kc = 0;
    f32x16 sacc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[u][r] = 0.f;
#pragma unroll
    for (int s = 0; s < QS; ++s)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const f16x8 kf = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(kq + (kc * 64 + u * 32) * ROW + s * 32));
        sacc[u] = CVMI_MFMA_32X32X16(kf, __builtin_bit_cast(f16x8, qf[s]), sacc[u], 0, 0, 0);
      }
    float mxa = max3f(sacc[0][0], sacc[1][0], sacc[0][8]), mxb = max3f(sacc[1][8], sacc[0][1], sacc[1][1]);      // two chains of v_max3_f32
#pragma unroll
    for (int r = 2; r < 8; ++r) { mxa = max3f(mxa, sacc[0][r], sacc[1][r]); mxb = max3f(mxb, sacc[0][r + 7], sacc[1][r + 7]); }
    const float mx = max3f(mxa, mxb, max3f(sacc[0][15], sacc[1][15], mxa));
    const float m_top = fmaxf(m_run, xhalf_max(mx));
    const bool grow = (m_top - m_run) * c > p.defer;      // first tile: m_run = -inf -> true
    const float m_new = grow ? m_top : m_run;
    const float mc = m_new * c;
    const float alpha = grow ? __builtin_amdgcn_exp2f(fmaf(m_run, c, -mc)) : 1.0f;
    float psum = 0.f;
    f16x8 pf[2][2];
    psum = softmax_tiles<true>(sacc, c, mc, pf);
    psum = xhalf_sum(psum);
    l_run = l_run * alpha + psum;
    m_run = m_new;
    if (__any(grow)) {                                      // (a real branch: rare once the first tiles have set the reference)
      asm volatile("" ::: "memory");
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < DT; ++t) {
          const char* a0 = vt + (kc * 64 + u * 32 + s * 16) * ROW + t * 64;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 2 * ROW));
          const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
          const u32x4 vv = {l2[0], l2[1], h2[0], h2[1]};
          oacc[t] = CVMI_MFMA_32X32X16(__builtin_bit_cast(f16x8, vv), pf[u][s], oacc[t], 0, 0, 0);
        }
  }
  if (q_ok) {
    const float inv = 1.f / l_run;
    long long obase;
    if (p.win > 0) {
      const int ow = p.q_pool ? p.win / 2 : p.win, ogh = p.q_pool ? p.grid_h / 2 : p.grid_h, ogw = p.q_pool ? p.grid_w / 2 : p.grid_w;
      obase = tok_off_fast(p, b, qi, p.o_st, ow, ogh, ogw, p.div_ow);
    } else {
      obase = (long long)b * p.o_sb + (long long)qi * p.o_st;
    }
    obase += (long long)h * p.o_sh;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = t * 32 + 8 * g + 4 * lh;
        if (d0 < p.dv) {
          f16x4 ov;
#pragma unroll
          for (int e = 0; e < 4; ++e) ov[e] = (f16)(oacc[t][4 * g + e] * inv);
          *reinterpret_cast<f16x4*>(p.o + (obase + d0) * 2) = ov;
        }
      }
  }
}

template <int QT>
int launch_res64(const AttnArgs& a, hipStream_t stream) {
  constexpr int lds = 2 * (2 * 64 * 144 + 64);
  const long long blocks = ((long long)a.B * a.heads + 1) / 2;
  CVMI_CHECK(blocks > 0 && blocks < (1ll << 31), "attention: bad grid");
  cvmi_note_kernel("attn_res64_kernel<%d>", QT);
  hipLaunchKernelGGL(attn_res64_kernel<QT>, dim3((unsigned)blocks), dim3(QT * 128), lds, stream, a);
  CVMI_LAUNCH_CHECK();
  return 0;
}

// ---- long sequences of head_dim 72 (Hiera global attention): the same inner loop, K / V streamed through two 64-key
//      LDS buffers by DMA (no staging registers: 4 waves per SIMD); one barrier per tile publishes tile t + 1 while it
//      drains, issued a full tile of MFMAs earlier.  Rows past Nk re-read the last key (finite; their scores are masked).
// AV8: as in attn_res256_kernel -- the AV product on the block-scaled fp8 MFMA.  Per 64-key tile, waves 0..5 build one 1-KiB piece each of
// the tile's e4m3 V^T operand image (d tile t = wave / 2, operand half u = wave & 1) from the 16-bit tile in front of the QK^T products; a
// second barrier per tile (LDS writes only: the next tile's DMA stays in flight across it) publishes the image before the three MFMAs.
template <int NW, bool AV8 = false, bool QL = false>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 4 : 2) void attn_dma72_kernel(const AttnArgs p) {
  constexpr int ROW = 144, TK = 64, QS = 5, DT = 3, CH = 9;
  constexpr int TILE_B = TK * ROW;                          // 9216 B per matrix per buffer
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // buffer u: K at u * 2 * TILE_B, V right behind it (a K row's over-read lands in V: finite)
  const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int lr = lane & 31, lh = lane >> 5;
  const int qgroups = (p.qtiles + NW - 1) / NW;
  const int wgx = xcd_order((int)blockIdx.x, (int)gridDim.x, p.xcd);
  const int item = wgx / qgroups;
  const int qt = (wgx - item * qgroups) * NW + wv;
  const bool live = qt < p.qtiles;
  const int b = item / p.heads, h = item - b * p.heads;
  const int qwin = p.q_pool ? p.win / 2 : p.win;

  // ---- DMA addressing: chunk L of a tile -> (key row L / 9, 16-byte chunk L % 9); 9 wave-instructions per matrix per tile
  const char *kbase, *vbase;
  {
    long long korg, vorg;
    if (p.win > 0) {
      const long long pix0 = tok_off(b, 0, 1, 1, p.win, p.grid_h, p.grid_w);
      korg = pix0 * p.k_st; vorg = pix0 * p.v_st;
    } else {
      korg = (long long)b * p.k_sb; vorg = (long long)b * p.v_sb;
    }
    kbase = p.k + (korg + (long long)h * p.k_sh) * 2;
    vbase = p.v + (vorg + (long long)h * p.v_sh) * 2;
  }
  // Plain sequences in whole tiles only (the Hiera global blocks; the dispatcher sends windows and ragged lengths elsewhere): a piece's
  // source is (uniform K / V base of this head + the tile's offset) + a 32-bit lane offset that never changes -- one VGPR per piece instead
  // of row / chunk / clamp / window / pointer arithmetic per tile, which held the kernel above 128 registers, i.e. at ONE workgroup per CU.
  // Addressing: buffer_load ... lds with one descriptor per matrix (base = this head's K / V, uniform), the tile's byte offset in the SGPR
  // offset and the piece's lane offset in ONE VGPR -- flat 64-bit lane pointers (what hipcc makes of global_load_lds here) cost ten VGPRs
  // and a 64-bit add per piece and tile, and were what kept the 4-wave form over the 128 registers of four workgroups per CU.
  int loff[(18 + NW - 1) / NW];
#pragma unroll
  for (int j = 0; j < (18 + NW - 1) / NW; ++j) {
    const int ins = j * NW + wv, isv = ins >= 9 ? 1 : 0, pc = ins - 9 * isv;
    const int L = pc * 64 + lane, row = L / CH, ch = L - row * CH;
    loff[j] = (row * (isv ? p.v_st : p.k_st) + ch * 8) * 2;
  }
  const int ktile_b = TK * p.k_st * 2, vtile_b = TK * p.v_st * 2;                  // bytes per 64-key tile (host: Nk * stride * 2 < 2^31)
  auto issue = [&](int kt, int buf) {
#pragma unroll
    for (int j = 0; j < (18 + NW - 1) / NW; ++j) {
      const int ins = j * NW + wv;                           // 0..8: K pieces, 9..17: V pieces (wave-uniform)
      if (ins >= 18) break;
      const int isv = ins >= 9 ? 1 : 0, pc = ins - 9 * isv;
      dma16_buffer(isv ? vbase : kbase, smem + buf * 2 * TILE_B + isv * TILE_B + pc * 1024, loff[j], kt * (isv ? vtile_b : ktile_b));
    }
  };
  const int nkt = (p.Nk + TK - 1) / TK;
  issue(0, 0);
  // ---- Q fragments while the DMA flies
  const int qi = qt * 32 + lr;
  const bool q_ok = live && qi < p.Nq;
  // element offset of this lane's query token (pooled: of the top-left token of its 2 x 2 block), computed once
  long long qoff0;
  {
    const int qc = q_ok ? qi : 0;
    int t = qc;
    if (p.q_pool) { const int py = qc / qwin, px = qc - py * qwin; t = (2 * py) * p.win + 2 * px; }
    qoff0 = tok_off(b, t, p.q_sb, p.q_st, p.win, p.grid_h, p.grid_w) + (long long)h * p.q_sh;
  }
  u32x4 qf[QS];
#pragma unroll
  for (int s = 0; s < QS; ++s) {
    const int d0 = 16 * s + 8 * lh;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (q_ok && d0 < p.dqk) {
      if (!p.q_pool) {
        v = *reinterpret_cast<const u32x4*>(p.q + (qoff0 + d0) * 2);
      } else {
        f16x8 m;
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = (f16)(CVMI_LOWEST16);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
          for (int dx = 0; dx < 2; ++dx) {
            const long long off = qoff0 + ((long long)dy * p.grid_w + dx) * p.q_st + d0;   // pooled: the 2 x 2 block's tokens are grid neighbours
            const f16x8 x = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(p.q + off * 2));
#pragma unroll
            for (int e = 0; e < 8; ++e) m[e] = x[e] > m[e] ? x[e] : m[e];
          }
        v = __builtin_bit_cast(u32x4, m);
      }
    }
    qf[s] = v;
  }
  f32x16 oacc[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float c = p.scale * 1.44269504088896340736f;
  const int li = lane & 15;
  const int vt_off = TILE_B + (lh + 4 * (li >> 2)) * ROW + (16 * (lr >> 4) + 4 * (li & 3)) * 2;      // key rows 4 apart: see key_perm72
  const int kq_off = key_perm72(lr) * ROW + lh * 16;
  // Row sums from the matrix pipe (16-bit form): the third 32-row tile of O^T = V^T P^T has rows d = 72..95 to spare, and the lanes whose
  // transposing reads would fetch V[key][72..75] (past the row: the next key's first elements, rows never stored) read a constant (1, 0, 0, 0)
  // instead -- row d = 72 of the accumulator is then the sum over keys of the ROUNDED probabilities, rescaled with the rest of the tile when
  // the running maximum moves.  That takes 32 v_add_f32 per tile and wave off the vector issue port, which -- not the matrix pipe -- bounds
  // this loop (ISA counts in DESIGN.md).  The constants sit at the eight offsets (32 u + 16 s) * ROW + {0, 2 ROW} the reads of one tile use.
  // QL (q pre-multiplied by scale * log2 e where it was produced, cvmi_attn_desc.q_log2): the running maximum goes through the matrix pipe
  // too.  head_dim 72 leaves k = 72..79 of the fifth QK^T step unused: the lanes that hold those k read K = (1, 0, .., 0) from a constant
  // instead of the next key's first bytes, and hold Q = (-m, 0, .., 0), m = the row's reference maximum ROUNDED to the operand type -- the
  // score tile comes out of the MFMA as s - m and exp2 applies to it as it stands: no v_fma_f32 per score either.  m moves only when a tile's
  // maximum exceeds it by more than the deferred-rescale threshold (and on the first tile): then the tile is corrected by the exact
  // difference of the two rounded references, which is also what the accumulators are rescaled by.
  constexpr int ONES = 4 * TILE_B + 256;                     // behind the buffers and their over-read slack
  const bool ones_lane = !AV8 && (lr >> 4) == 0 && (li & 3) == 2;
  if constexpr (!AV8) {
    // TWO copies of the V constants, one per 32-lane half (= LDS lane group of a transposing read): the 28 ordinary lanes of a group and
    // the V rows they read cover 56 of the 64 banks exactly once, and the 8 banks left over are the ones the redirected lanes WOULD have
    // used -- banks 36 / 52 / 4 / 20 (+ 1) in the half lh = 0, banks 8 / 24 / 40 / 56 in lh = 1 (tile bases are multiples of 256 bytes).  A
    // constant anywhere else costs every t = 2 read a conflict cycle (PMC r03: 0.15 of the LDS cycles with one copy at bank 0).
    if (tid < 16) {
      const int t8 = tid & 7, off = ((t8 >> 2) * 32 + ((t8 >> 1) & 1) * 16 + (t8 & 1) * 2) * ROW;
      *reinterpret_cast<u32x2*>(smem + ONES + (tid < 8 ? 144 : 32) + off) = (u32x2){CVMI_ONE16X2 & 0xFFFFu, 0u};
    } else if (QL && tid < 18) {
      *reinterpret_cast<u32x4*>(smem + ONES + 16 + (tid - 16) * 32 * ROW) = (u32x4){CVMI_ONE16X2 & 0xFFFFu, 0u, 0u, 0u};      // K constants: (32 u) * ROW apart
    }
  }
  float m_ref = 0.f;                                          // QL: the reference maximum held (negated) in the Q operand
  dma_wait();
  __syncthreads();                                          // tile 0 landed

#pragma unroll 1
  for (int kt = 0; kt < nkt; ++kt) {
    const char* const kq = smem + (kt & 1) * 2 * TILE_B + kq_off;
    const char* const vt = smem + (kt & 1) * 2 * TILE_B + vt_off;
    const char* const vt2 = ones_lane ? smem + ONES + (lh ? 32 : 144) - 128 : vt;      // base of the t = 2 reads (their + 128 lands on the constants)
    const char* const kq4 = (QL && lh) ? smem + ONES + 16 - 128 : kq;   // QL: base of the s = 4 reads of the lanes that hold k = 72..79
    if (kt + 1 < nkt) issue(kt + 1, (kt + 1) & 1);          // the other buffer was last read in iteration kt - 1 (barrier below)
    constexpr int kc = 0;
    if constexpr (AV8) {
      static_assert(NW == 8, "six of eight waves build the e4m3 V image");
      if (wv < 6) {                                         // wave-uniform
        const int t = wv >> 1, u = wv & 1;
        unsigned w[4];
#pragma unroll
        for (int sgrp = 0; sgrp < 2; ++sgrp) {
          const char* a0 = vt + (u * 32 + sgrp * 16) * ROW + t * 64;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 2 * ROW));
          const f16x4 l4 = __builtin_bit_cast(f16x4, lo), h4 = __builtin_bit_cast(f16x4, hi);
          float f[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) { f[e] = (float)l4[e]; f[4 + e] = (float)h4[e]; }
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = __builtin_amdgcn_fmed3f(f[e], -448.f, 448.f);
          w[2 * sgrp] = pack4_e4m3(f[0], f[1], f[2], f[3]);
          w[2 * sgrp + 1] = pack4_e4m3(f[4], f[5], f[6], f[7]);
        }
        *reinterpret_cast<u32x4*>(smem + 4 * TILE_B + wv * 1024 + lane * 16) = (u32x4){w[0], w[1], w[2], w[3]};     // last read before the barrier that ended iteration kt - 1
      }
    }
    f32x16 sacc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[u][r] = 0.f;
#pragma unroll
    for (int s = 0; s < QS; ++s)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const f16x8 kf = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>((s == 4 ? kq4 : kq) + (kc * 64 + u * 32) * ROW + s * 32));
        sacc[u] = CVMI_MFMA_32X32X16(kf, __builtin_bit_cast(f16x8, qf[s]), sacc[u], 0, 0, 0);
      }
    float mxa = max3f(sacc[0][0], sacc[1][0], sacc[0][8]), mxb = max3f(sacc[1][8], sacc[0][1], sacc[1][1]);      // two chains of v_max3_f32
#pragma unroll
    for (int r = 2; r < 8; ++r) { mxa = max3f(mxa, sacc[0][r], sacc[1][r]); mxb = max3f(mxb, sacc[0][r + 7], sacc[1][r + 7]); }
    const float mx = max3f(mxa, mxb, max3f(sacc[0][15], sacc[1][15], mxa));
    f16x8 pf[2][2];
    u32x4 p8[2];
    if constexpr (QL) {
      static_assert(!AV8, "QL is the 16-bit form");
      const float top = xhalf_max(mx);                          // the tile's maximum RELATIVE to m_ref (the MFMA subtracted it)
      const bool grow = kt == 0 || top > p.defer;
      if (__any(grow)) {                                      // (a real branch: the first tile, then rare)
        const float m_new = grow ? (float)(f16)(m_ref + top) : m_ref;       // representable in the operand type, so the Q slot holds it exactly
        const float dlt = m_new - m_ref;                        // exact in f32 (both are 16-bit values of similar magnitude or m_ref = 0)
        const float alpha = __builtin_amdgcn_exp2f(-dlt);
        m_ref = m_new;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int r = 0; r < 16; ++r) sacc[u][r] -= dlt;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;
        if (lh) {                                               // the lanes that hold k = 72..79 of the fifth step
          f16x8 qv = __builtin_bit_cast(f16x8, qf[4]);
          qv[0] = (f16)(-m_new);
          qf[4] = __builtin_bit_cast(u32x4, qv);
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) pf[u][r >> 3][r & 7] = (f16)__builtin_amdgcn_exp2f(sacc[u][r]);
    } else {
    const float m_top = fmaxf(m_run, xhalf_max(mx));
    const bool grow = (m_top - m_run) * c > (AV8 ? 0.f : p.defer);      // first tile: m_run = -inf -> true
    const float m_new = grow ? m_top : m_run;
    const float mc = m_new * c;
    const float alpha = grow ? __builtin_amdgcn_exp2f(fmaf(m_run, c, -mc)) : 1.0f;
    float psum = 0.f;
    if constexpr (AV8) {
      const float mc8 = mc - 8.f;                           // p * 2^8 through the exponent
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        unsigned w[4];
#pragma unroll
        for (int r = 0; r < 16; r += 4) {
          float e[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) { e[i] = __builtin_amdgcn_exp2f(fmaf(sacc[u][r + i], c, -mc8)); psum += e[i]; }
          w[r >> 2] = pack4_e4m3(e[0], e[1], e[2], e[3]);
        }
        p8[u] = (u32x4){w[0], w[1], w[2], w[3]};
      }
      psum *= 0.00390625f;
    } else {
      softmax_tiles<false>(sacc, c, mc, pf);
    }
    if constexpr (AV8) {
      psum = xhalf_sum(psum);
      l_run = l_run * alpha + psum;
    }
    m_run = m_new;
    if (__any(grow)) {                                      // (a real branch: rare once the first tiles have set the reference)
      asm volatile("" ::: "memory");
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;
    }
    }
    if constexpr (AV8) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's piece of the e4m3 image is written ...
      __builtin_amdgcn_s_barrier();                         // ... and published (no vmcnt wait: tile kt + 1's DMA keeps flying)
      const v8i32 pb = {(int)p8[0][0], (int)p8[0][1], (int)p8[0][2], (int)p8[0][3], (int)p8[1][0], (int)p8[1][1], (int)p8[1][2], (int)p8[1][3]};
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        const char* va = smem + 4 * TILE_B + (2 * t) * 1024 + lane * 16;
        const u32x4 a_lo = *reinterpret_cast<const u32x4*>(va), a_hi = *reinterpret_cast<const u32x4*>(va + 1024);
        const v8i32 av = {(int)a_lo[0], (int)a_lo[1], (int)a_lo[2], (int)a_lo[3], (int)a_hi[0], (int)a_hi[1], (int)a_hi[2], (int)a_hi[3]};
        oacc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, pb, oacc[t], 0, 0, 0, 127, 0, 119);
      }
    } else {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int t = 0; t < DT; ++t) {
            const char* a0 = (t == 2 ? vt2 : vt) + (kc * 64 + u * 32 + s * 16) * ROW + t * 64;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 2 * ROW));
            const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
            const u32x4 vv = {l2[0], l2[1], h2[0], h2[1]};
            oacc[t] = CVMI_MFMA_32X32X16(__builtin_bit_cast(f16x8, vv), pf[u][s], oacc[t], 0, 0, 0);
          }
    }
    dma_wait();                                             // this wave's pieces of tile kt + 1 (issued a whole tile of MFMAs ago)
    __syncthreads();                                        // tile kt fully read; tile kt + 1 landed in every wave
  }
  if constexpr (!AV8) l_run = __shfl(oacc[2][4], lr);        // row d = 72 (tile 2, row 8): register 4 of the lanes of half 0, column = query
  if (q_ok) {
    const float inv = 1.f / l_run;
    long long obase;
    if (p.win > 0) {
      const int ow = p.q_pool ? p.win / 2 : p.win, ogh = p.q_pool ? p.grid_h / 2 : p.grid_h, ogw = p.q_pool ? p.grid_w / 2 : p.grid_w;
      obase = tok_off(b, qi, p.o_sb, p.o_st, ow, ogh, ogw);
    } else {
      obase = (long long)b * p.o_sb + (long long)qi * p.o_st;
    }
    obase += (long long)h * p.o_sh;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = t * 32 + 8 * g + 4 * lh;
        if (d0 < p.dv) {
          f16x4 ov;
#pragma unroll
          for (int e = 0; e < 4; ++e) ov[e] = (f16)(oacc[t][4 * g + e] * inv);
          *reinterpret_cast<f16x4*>(p.o + (obase + d0) * 2) = ov;
        }
      }
  }
}

template <int NW, bool AV8 = false, bool QL = false>
int launch_dma72(const AttnArgs& a, hipStream_t stream) {
  constexpr int lds = 4 * 64 * 144 + (AV8 ? 6 * 1024 : 7360) + 256;       // AV8: + the tile's e4m3 V^T image; 16-bit: + the row-sum constants
  const long long blocks = (long long)a.B * a.heads * ((a.qtiles + NW - 1) / NW);
  CVMI_CHECK(blocks > 0 && blocks < (1ll << 31), "attention: bad grid");
  cvmi_note_kernel("attn_dma72_kernel<%d, %s, %s>", NW, CVMI_BOOLNAME(AV8), CVMI_BOOLNAME(QL));
  hipLaunchKernelGGL((attn_dma72_kernel<NW, AV8, QL>), dim3((unsigned)blocks), dim3(NW * 64), lds, stream, a);
  CVMI_LAUNCH_CHECK();
  return 0;
}

template <int NW, bool AV8 = false, bool QL = false>
int launch_res256(const AttnArgs& a, hipStream_t stream) {
  constexpr int lds = 2 * 256 * 144 + 256 + (AV8 ? 0 : 7360);      // + slack: the last rows' over-reads stay inside the allocation; 16-bit: + the row-sum / maximum constants (two workgroups: 162,688 of a CU's 163,840 bytes)
  static bool attr_done = false;
  if (!attr_done) {
    CVMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_res256_kernel<NW, AV8, QL>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_done = true;
  }
  const long long blocks = (long long)a.B * a.heads * ((a.qtiles + NW - 1) / NW);
  CVMI_CHECK(blocks > 0 && blocks < (1ll << 31), "attention: bad grid");
  cvmi_note_kernel("attn_res256_kernel<%d, %s, %s>", NW, CVMI_BOOLNAME(AV8), CVMI_BOOLNAME(QL));
  hipLaunchKernelGGL((attn_res256_kernel<NW, AV8, QL>), dim3((unsigned)blocks), dim3(NW * 64), lds, stream, a);
  CVMI_LAUNCH_CHECK();
  return 0;
}

// ---- 16-token windows (Hiera stage 2, and the q-pooled stage 2 -> 3 transition): VALU kernel ----------------------------
// A 4 x 4 window gives the MFMA kernels 16 keys x 16 (or 4 pooled) queries per 32 x 32 tile: >= 75 % padding, and the
// per-item cost (window addressing, staging, barriers) dominates -- 22 TFLOP/s.  The arithmetic is tiny (37 kMAC per
// window-head), so here ONE THREAD owns one (window, head, query): its q row lives in registers, the item's K and V rows
// are staged in LDS (a wave serves 4 items, all lanes load), scores are 36 v_dot2_f32_f16 per key, the softmax is a
// two-pass over 16 registers, and the output row accumulates in 72 fp32 registers.  No cross-lane traffic at all.
template <int NQ, int DCH>                                  // NQ queries per item (16, or 4 when q-pooled); DCH = head_dim / 8
__global__ __launch_bounds__(128) void attn_win16_kernel(const AttnArgs p) {
  constexpr int NK = 16, IPW = 4, D = DCH * 8, ROW = D * 2;        // row bytes (unpadded: the 16 query lanes of an item read one address)
  // Item pitch: a wave's read touches FOUR addresses (its four items, same key / chunk).  At the natural pitch (16 x 144 = 2304 B = 9 x 256) all
  // four fall on the same 4 banks -- a 4-way conflict on every ds_read_b128 of the kernel (PMC r02 / r03: SQ_LDS_BANK_CONFLICT = 0.46 of the LDS
  // cycles, and 288 such reads per thread are more LDS time than the kernel's VALU time).  + 64 B per item puts them on 16 distinct banks.
  constexpr int ITEM_B = NK * ROW + 64;
  __shared__ __attribute__((aligned(16))) char lds[2 * 2 * IPW * ITEM_B];          // 2 waves x (K, V) x 4 items x 2368 B = 37 KiB
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  char* const Kl = lds + wv * 2 * IPW * ITEM_B;
  char* const Vl = Kl + IPW * ITEM_B;
  const int nitems = p.B * p.heads;
  const int item0 = (xcd_order((int)blockIdx.x, (int)gridDim.x, p.xcd) * 2 + wv) * IPW;   // first item of this wave
  // ---- per-lane item for the compute phase: lane = ti * NQ + tq, ti < IPW
  const int ti = lane / NQ, tq = lane - ti * NQ;
  const bool active = ti < IPW && item0 + ti < nitems;
  const int my_item = item0 + (ti < IPW ? ti : 0);
  const int itc = my_item < nitems ? my_item : nitems - 1;
  const int b = itc / p.heads, h = itc - b * p.heads;
  const long long pix0 = tok_off(b, 0, 1, 1, p.win, p.grid_h, p.grid_w);        // window origin (pixels) of my item
  // ---- stage K and V of the wave's 4 items: chunk id -> (item, key, 16-byte chunk); origins come from the lane that owns the item
  constexpr int NCHUNK = IPW * NK * DCH, NLD = (NCHUNK + 63) / 64;
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    u32x4 r[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int id = lane + i * 64;
      const int ci = id / (NK * DCH), rem = id - ci * (NK * DCH);
      const int key = rem / DCH, ch = rem - key * DCH;
      const long long o0 = __shfl(pix0, (ci < IPW ? ci : 0) * NQ);
      const int hh = __shfl(h, (ci < IPW ? ci : 0) * NQ);
      const bool ok = id < NCHUNK && item0 + ci < nitems;
      const int ky = key >> 2, kx = key & 3;                          // 4 x 4 window
      const long long st = m == 0 ? p.k_st : p.v_st, sh = m == 0 ? p.k_sh : p.v_sh;
      const long long off = (o0 + (long long)ky * p.grid_w + kx) * st + (long long)hh * sh + ch * 8;
      const char* base = m == 0 ? p.k : p.v;
      r[i] = *reinterpret_cast<const u32x4*>(base + (ok ? off : 0) * 2);
    }
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int id = lane + i * 64;
      if (id < NCHUNK) *reinterpret_cast<u32x4*>((m == 0 ? Kl : Vl) + id * 16 + (id / (NK * DCH)) * 64) = r[i];       // (+ the item pad)
    }
  }
  // ---- my query row (2 x 2 max-pool of the projected q tokens when q_pool)
  u32x4 qv[DCH];
  {
    const int qw = p.q_pool ? 2 : 4;                                  // query grid inside the window
    const int py = tq / qw, px = tq - py * qw;
#pragma unroll
    for (int c = 0; c < DCH; ++c) {
      if (!p.q_pool) {
        const long long off = (pix0 + (long long)py * p.grid_w + px) * p.q_st + (long long)h * p.q_sh + c * 8;
        qv[c] = *reinterpret_cast<const u32x4*>(p.q + (active ? off : 0) * 2);
      } else {
        f16x8 m;
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = (f16)(CVMI_LOWEST16);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
          for (int dx = 0; dx < 2; ++dx) {
            const long long off = (pix0 + (long long)(2 * py + dy) * p.grid_w + 2 * px + dx) * p.q_st + (long long)h * p.q_sh + c * 8;
            const f16x8 x = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(p.q + (active ? off : 0) * 2));
#pragma unroll
            for (int e = 0; e < 8; ++e) m[e] = x[e] > m[e] ? x[e] : m[e];
          }
        qv[c] = __builtin_bit_cast(u32x4, m);
      }
    }
  }
  __syncthreads();
  const char* const kr = Kl + (ti < IPW ? ti : 0) * ITEM_B;
  const char* const vr = Vl + (ti < IPW ? ti : 0) * ITEM_B;
  // ---- scores
  float sc[NK];
  const float c2 = p.scale * 1.44269504088896340736f;
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < DCH; ++c) {
      const f16x8 kh = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(kr + k * ROW + c * 16));
      const f16x8 qh = __builtin_bit_cast(f16x8, qv[c]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const f16x2 a2 = {qh[2 * e], qh[2 * e + 1]}, b2 = {kh[2 * e], kh[2 * e + 1]};
        acc = CVMI_FDOT2(a2, b2, acc, false);
      }
    }
    sc[k] = acc * c2;
    mx = fmaxf(mx, sc[k]);
  }
  float l = 0.f;
#pragma unroll
  for (int k = 0; k < NK; ++k) { sc[k] = __builtin_amdgcn_exp2f(sc[k] - mx); l += sc[k]; }
  // P is rounded to fp16 before the PV product, as the MFMA kernels (and a torch fp16 pipeline) do
  // ---- output row
  float o[D];
#pragma unroll
  for (int d = 0; d < D; ++d) o[d] = 0.f;
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const float pk = (float)(f16)sc[k];
#pragma unroll
    for (int c = 0; c < DCH; ++c) {
      const f16x8 vv = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(vr + k * ROW + c * 16));
#pragma unroll
      for (int e = 0; e < 8; ++e) o[c * 8 + e] = fmaf(pk, (float)vv[e], o[c * 8 + e]);
    }
  }
  if (active) {
    const float inv = 1.f / l;
    const int ow = p.q_pool ? 2 : 4, ogw = p.q_pool ? p.grid_w / 2 : p.grid_w, ogh = p.q_pool ? p.grid_h / 2 : p.grid_h;
    const long long obase = tok_off(b, tq, p.o_sb, p.o_st, ow, ogh, ogw) + (long long)h * p.o_sh;
#pragma unroll
    for (int c = 0; c < DCH; ++c) {
      f16x8 ov;
#pragma unroll
      for (int e = 0; e < 8; ++e) ov[e] = (f16)(o[c * 8 + e] * inv);
      *reinterpret_cast<u32x4*>(p.o + (obase + c * 8) * 2) = __builtin_bit_cast(u32x4, ov);
    }
  }
}

// ---- f32 parity kernel: one wave per query; lanes = keys for S, lanes = d for the PV sum ----------
__global__ __launch_bounds__(256) void attn_f32_kernel(const AttnArgs p) {
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long total = (long long)p.B * p.heads * p.Nq;
  const long long row_raw = (long long)blockIdx.x * 4 + wv;      // (b, h, q)
  const bool valid = row_raw < total;
  const long long row = valid ? row_raw : total - 1;
  const int qi = (int)(row % p.Nq);
  const int bh = (int)(row / p.Nq);
  const int b = bh / p.heads, h = bh - b * p.heads;
  const float* Q = reinterpret_cast<const float*>(p.q);
  const float* K = reinterpret_cast<const float*>(p.k);
  const float* V = reinterpret_cast<const float*>(p.v);
  float* O = reinterpret_cast<float*>(p.o);
  __shared__ float qs[4][128];
  const int qwin = p.q_pool ? p.win / 2 : p.win;
  for (int d = lane; d < p.dqk; d += 64) {
    float v;
    if (!p.q_pool) {
      v = Q[tok_off(b, qi, p.q_sb, p.q_st, p.win, p.grid_h, p.grid_w) + (long long)h * p.q_sh + d];
    } else {
      const int py = qi / qwin, px = qi - py * qwin;
      v = -INFINITY;
      for (int dy = 0; dy < 2; ++dy)
        for (int dx = 0; dx < 2; ++dx) {
          const int t = (2 * py + dy) * p.win + 2 * px + dx;
          v = fmaxf(v, Q[tok_off(b, t, p.q_sb, p.q_st, p.win, p.grid_h, p.grid_w) + (long long)h * p.q_sh + d]);
        }
    }
    qs[wv][d] = v;
  }
  __syncthreads();
  float m_run = -INFINITY, l_run = 0.f;
  float o0 = 0.f, o1 = 0.f;                                       // d = lane, lane + 64
  for (int k0 = 0; k0 < p.Nk; k0 += 64) {
    const int key = k0 + lane;
    float s = -INFINITY;
    if (key < p.Nk) {
      const float* kp = K + tok_off(b, key, p.k_sb, p.k_st, p.win, p.grid_h, p.grid_w) + (long long)h * p.k_sh;
      float acc = 0.f;
      for (int d = 0; d < p.dqk; ++d) acc = fmaf(qs[wv][d], kp[d], acc);
      s = acc * p.scale;
    }
    float mx = s;
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = expf(m_run - m_new);
    const float pv = key < p.Nk ? expf(s - m_new) : 0.f;
    float ps = pv;
    for (int off = 32; off > 0; off >>= 1) ps += __shfl_xor(ps, off);
    l_run = l_run * alpha + ps;
    m_run = m_new;
    o0 *= alpha; o1 *= alpha;
    const int nk = p.Nk - k0 < 64 ? p.Nk - k0 : 64;
    for (int j = 0; j < nk; ++j) {
      const float pj = __shfl(pv, j);
      const float* vp = V + tok_off(b, k0 + j, p.v_sb, p.v_st, p.win, p.grid_h, p.grid_w) + (long long)h * p.v_sh;
      if (lane < p.dv) o0 = fmaf(pj, vp[lane], o0);
      if (lane + 64 < p.dv) o1 = fmaf(pj, vp[lane + 64], o1);
    }
  }
  long long obase;
  if (p.win > 0) {
    const int ow = p.q_pool ? p.win / 2 : p.win, ogh = p.q_pool ? p.grid_h / 2 : p.grid_h, ogw = p.q_pool ? p.grid_w / 2 : p.grid_w;
    obase = tok_off(b, qi, p.o_sb, p.o_st, ow, ogh, ogw);
  } else {
    obase = (long long)b * p.o_sb + (long long)qi * p.o_st;
  }
  obase += (long long)h * p.o_sh;
  const float inv = 1.f / l_run;
  if (valid && lane < p.dv) O[obase + lane] = o0 * inv;
  if (valid && lane + 64 < p.dv) O[obase + lane + 64] = o1 * inv;
}

template <int DQKP, int DVP, int GS>
int launch_f16(const AttnArgs& a, hipStream_t stream) {
  constexpr int KROW = DQKP * 2 + 16, VROW = 72;
  constexpr size_t lds = (size_t)(256 / GS) * (32 * KROW + DVP * VROW);
  long long blocks;
  if (GS == 256) blocks = (long long)a.B * a.heads * ((a.qtiles + 3) / 4);
  else blocks = ((long long)a.items + 3) / 4;
  CVMI_CHECK(blocks > 0 && blocks < (1ll << 31), "attention: bad grid");
  static bool attr_done = false;
  if (!attr_done && lds > 64 * 1024) {
    CVMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_f16_kernel<DQKP, DVP, GS>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  cvmi_note_kernel("attn_f16_kernel<%d, %d, %d>", DQKP, DVP, GS);
  hipLaunchKernelGGL((attn_f16_kernel<DQKP, DVP, GS>), dim3((unsigned)blocks), dim3(256), lds, stream, a);
  CVMI_LAUNCH_CHECK();
  return 0;
}

template <int DQKP, int DVP>
int launch_f16_gs(const AttnArgs& a, hipStream_t stream) {
  // share K/V tiles across the workgroup when each (batch, head) has >= 4 query tiles
  static const int use64 = getenv("CVMI_ATTN64") ? atoi(getenv("CVMI_ATTN64")) : 2;      // tuning experiments only: 0 old, 1 four waves, 2 eight
  if (a.qtiles >= 8 && use64 == 2) return launch_attn64<DQKP, DVP, 8>(a, stream);
  if (a.qtiles >= 4) return use64 ? launch_attn64<DQKP, DVP, 4>(a, stream) : launch_f16<DQKP, DVP, 256>(a, stream);
  // two or three query tiles against a LONG key axis (mask-decoder token -> image attention: 38 tokens x 4096 positions): alone, each wave
  // walks 128 key tiles through its private LDS slice; in the shared-tile kernel the idle waves of the workgroup help to stage the tiles
  static const int longk = getenv("CVMI_ATTN64_LONGK") ? atoi(getenv("CVMI_ATTN64_LONGK")) : 1;         // tuning experiments only
  if (longk && use64 && a.qtiles >= 2 && a.Nk >= 2048 && a.win == 0) return longk == 2 ? launch_attn64<DQKP, DVP, 8>(a, stream) : launch_attn64<DQKP, DVP, 4>(a, stream);
  return launch_f16<DQKP, DVP, 64>(a, stream);
}

}  // namespace

#ifndef CVMI_OPERAND_BF16
extern "C" int cvmi_attention_bf16(const cvmi_attn_desc* d, cvmi_stream_t stream_);
#endif

extern "C" int CVMI_ENTRY(cvmi_attention)(const cvmi_attn_desc* d, cvmi_stream_t stream_) {
  CVMI_CHECK(d && d->q && d->k && d->v && d->o, "attention: null pointer");
#ifndef CVMI_OPERAND_BF16
  if (d->dtype == CVMI_BF16) return cvmi_attention_bf16(d, stream_);
#endif
  CVMI_CHECK(d->dtype == CVMI_T16 || d->dtype == CVMI_F32, "attention: bad dtype");
  CVMI_CHECK(d->B > 0 && d->heads > 0 && d->Nq > 0 && d->Nk > 0 && d->dqk > 0 && d->dv > 0, "attention: bad shape");
  AttnArgs a;
  a.q = (const char*)d->q; a.k = (const char*)d->k; a.v = (const char*)d->v; a.o = (char*)d->o;
  a.q_sb = d->q_sb; a.q_sh = d->q_sh; a.q_st = d->q_st; a.k_sb = d->k_sb; a.k_sh = d->k_sh; a.k_st = d->k_st;
  a.v_sb = d->v_sb; a.v_sh = d->v_sh; a.v_st = d->v_st; a.o_sb = d->o_sb; a.o_sh = d->o_sh; a.o_st = d->o_st;
  a.B = d->B; a.heads = d->heads; a.Nq = d->Nq; a.Nk = d->Nk; a.dqk = d->dqk; a.dv = d->dv; a.scale = d->q_log2 ? 0.6931471805599453f : d->scale; a.q_log2 = d->q_log2;
  a.win = d->win; a.grid_h = d->grid_h; a.grid_w = d->grid_w; a.q_pool = d->q_pool;
  a.q_bdiv = d->q_bdiv > 1 ? d->q_bdiv : 1; a.kv_bdiv = d->kv_bdiv > 1 ? d->kv_bdiv : 1;
  const bool shared = a.q_bdiv > 1 || a.kv_bdiv > 1;
  CVMI_CHECK(!shared || (d->win == 0 && d->dtype == CVMI_T16 && d->dqk <= 64 && d->dv <= 64 && d->B % a.q_bdiv == 0 && d->B % a.kv_bdiv == 0),
             "attention: batch sharing (q_bdiv / kv_bdiv) needs fp16, no window, head dims <= 64 and B a multiple of the divisor");
  a.qtiles = (d->Nq + 31) / 32;
  a.div_win.init(d->win > 0 ? (unsigned)d->win : 1u);
  a.wpr = d->win > 0 ? d->grid_w / d->win : 1;
  a.wpi = d->win > 0 ? a.wpr * (d->grid_h / d->win) : 1;
  a.div_wpr.init((unsigned)(a.wpr > 0 ? a.wpr : 1)); a.div_wpi.init((unsigned)(a.wpi > 0 ? a.wpi : 1));
  a.div_ow.init(d->win > 0 ? (unsigned)(d->q_pool ? (d->win / 2 > 0 ? d->win / 2 : 1) : d->win) : 1u);
  a.div_heads.init((unsigned)(d->heads > 0 ? d->heads : 1));
  a.items = d->B * d->heads * a.qtiles;
  static const int use_xcd = getenv("CVMI_ATTN_XCD") ? atoi(getenv("CVMI_ATTN_XCD")) : 1;                 // A/B runs only
  a.xcd = use_xcd;
  static const float defer = getenv("CVMI_ATTN_DEFER") ? (float)atof(getenv("CVMI_ATTN_DEFER")) : DEFER_LOG2;   // A/B runs only
  a.defer = defer;
  static const int diag = getenv("CVMI_ATTN_DIAG") ? atoi(getenv("CVMI_ATTN_DIAG")) : 0;
  a.diag = diag;
  if (d->win > 0) {
    CVMI_CHECK(d->grid_h % d->win == 0 && d->grid_w % d->win == 0, "attention: grid %dx%d not divisible by window %d", d->grid_h, d->grid_w, d->win);
    CVMI_CHECK(d->Nk == d->win * d->win, "attention: window mode needs Nk == win^2");
    CVMI_CHECK(d->B % ((d->grid_h / d->win) * (d->grid_w / d->win)) == 0, "attention: B is not a whole number of images");
    if (d->q_pool) CVMI_CHECK(d->win % 2 == 0 && d->Nq == (d->win / 2) * (d->win / 2), "attention: q_pool needs Nq == (win/2)^2");
    else CVMI_CHECK(d->Nq == d->Nk, "attention: window mode needs Nq == Nk");
  } else {
    CVMI_CHECK(!d->q_pool, "attention: q_pool requires window mode");
  }
  hipStream_t stream = (hipStream_t)stream_;
  if (d->dtype == CVMI_F32) {
    CVMI_CHECK(d->dqk <= 128 && d->dv <= 128, "attention(f32): head dims up to 128");
    const long long rows = (long long)d->B * d->heads * d->Nq;
    cvmi_note_kernel("attn_f32_kernel");
    hipLaunchKernelGGL(attn_f32_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, a);
    CVMI_LAUNCH_CHECK();
    return 0;
  }
  CVMI_CHECK(d->dqk % 8 == 0 && d->dv % 8 == 0, "attention(f16): head dims must be multiples of 8");
  CVMI_CHECK(d->k_st > 0 && d->v_st > 0 && d->k_st < (1 << 16) && d->v_st < (1 << 16) && (long long)d->Nk * d->grid_w < (1 << 15) + (long long)d->Nk,
             "attention(f16): token strides must be below 65536 elements");
  const long long strides[] = {d->q_sb, d->q_sh, d->q_st, d->k_sb, d->k_sh, d->k_st, d->v_sb, d->v_sh, d->v_st, d->o_sb, d->o_sh, d->o_st};
  for (long long s : strides) CVMI_CHECK(s % 4 == 0, "attention(f16): strides must be multiples of 4 elements");
  CVMI_CHECK(d->q_st % 8 == 0 && d->k_st % 8 == 0 && d->v_st % 8 == 0 && d->q_sh % 8 == 0 && d->k_sh % 8 == 0 && d->v_sh % 8 == 0 &&
             d->q_sb % 8 == 0 && d->k_sb % 8 == 0 && d->v_sb % 8 == 0, "attention(f16): q/k/v strides must be multiples of 8 elements");
  CVMI_CHECK((((uintptr_t)d->q | (uintptr_t)d->k | (uintptr_t)d->v) & 15) == 0 && ((uintptr_t)d->o & 7) == 0, "attention(f16): misaligned pointer");
  // 4 x 4 windows of Hiera's head_dim 72: the per-thread VALU kernel
  static const int use_win16 = getenv("CVMI_ATTN_WIN16") ? atoi(getenv("CVMI_ATTN_WIN16")) : 1;           // tuning experiments only
  if (use_win16 && d->win == 4 && d->Nk == 16 && d->dqk == 72 && d->dv == 72 && (d->Nq == 16 || (d->q_pool && d->Nq == 4)) &&
      ((uintptr_t)d->o & 15) == 0 && d->o_st % 8 == 0 && d->o_sh % 8 == 0) {
    const long long items = (long long)d->B * d->heads;
    const unsigned blocks = (unsigned)((items + 7) / 8);
    cvmi_note_kernel("attn_win16_kernel<%d, 9>", d->Nq == 16 ? 16 : 4);
    if (d->Nq == 16) hipLaunchKernelGGL((attn_win16_kernel<16, 9>), dim3(blocks), dim3(128), 0, stream, a);
    else hipLaunchKernelGGL((attn_win16_kernel<4, 9>), dim3(blocks), dim3(128), 0, stream, a);
    CVMI_LAUNCH_CHECK();
    return 0;
  }
  static const int use_res256 = getenv("CVMI_ATTN_RES256") ? atoi(getenv("CVMI_ATTN_RES256")) : 2;      // tuning experiments only: 0 off, 1 four waves, 2 eight waves (4 per SIMD at 125 VGPRs)
  if (use_res256 && d->Nk == 256 && d->dqk == 72 && d->dv == 72 && d->k_st % 8 == 0 && d->v_st % 8 == 0) {
    if (d->av_fp8 && a.qtiles >= 8) return launch_res256<8, true>(a, stream);          // block-scaled fp8 AV product (configs[4])
    if (d->q_log2) return (use_res256 == 2 && a.qtiles >= 8) ? launch_res256<8, false, true>(a, stream) : launch_res256<4, false, true>(a, stream);
    return (use_res256 == 2 && a.qtiles >= 8) ? launch_res256<8>(a, stream) : launch_res256<4>(a, stream);
  }
  static const int use_res64 = getenv("CVMI_ATTN_RES64") ? atoi(getenv("CVMI_ATTN_RES64")) : 1;          // tuning experiments only
  if (use_res64 && d->Nk == 64 && d->dqk == 72 && d->dv == 72 && d->k_st % 8 == 0 && d->v_st % 8 == 0 && (a.qtiles == 1 || a.qtiles == 2))
    return a.qtiles == 2 ? launch_res64<2>(a, stream) : launch_res64<1>(a, stream);
  static const int use_dma72 = getenv("CVMI_ATTN_DMA72") ? atoi(getenv("CVMI_ATTN_DMA72")) : 1;          // tuning experiments only
  if (use_dma72 && d->Nk >= 512 && d->Nk % 64 == 0 && d->win == 0 && a.qtiles >= 8 && d->dqk == 72 && d->dv == 72 && d->k_st % 8 == 0 && d->v_st % 8 == 0 && !d->q_pool &&
      (long long)d->Nk * d->k_st * 2 < (1ll << 31) && (long long)d->Nk * d->v_st * 2 < (1ll << 31))
  {
    // 8 waves x 2 workgroups per CU (4 waves per SIMD: the kernel holds 127 registers since its DMA went to buffer addressing; at 139 it ran ONE
    // workgroup per CU, 830 us per Hiera-L global block at B = 16).  4-wave workgroups, four per CU, were the stop-gap that showed it (-12 %);
    // they stream K / V from L2 twice as often and are kept for A/B runs.
    static const int nw = getenv("CVMI_ATTN_DMA72_NW") ? atoi(getenv("CVMI_ATTN_DMA72_NW")) : 8;
    if (d->av_fp8) return launch_dma72<8, true>(a, stream);
    if (d->q_log2) return nw == 4 ? launch_dma72<4, false, true>(a, stream) : launch_dma72<8, false, true>(a, stream);
    return nw == 4 ? launch_dma72<4>(a, stream) : launch_dma72<8>(a, stream);
  }
  if (d->dqk <= 32 && d->dv <= 32) return launch_f16_gs<32, 32>(a, stream);
  if (d->dqk <= 32 && d->dv <= 64) return launch_f16_gs<32, 64>(a, stream);
  if (d->dqk <= 64 && d->dv <= 64) return launch_f16_gs<64, 64>(a, stream);
  if (d->dqk <= 96 && d->dv <= 96) return launch_f16_gs<96, 96>(a, stream);
  if (d->dqk <= 128 && d->dv <= 128) return launch_f16_gs<128, 128>(a, stream);
  CVMI_FAIL("attention(f16): head dims (%d, %d) unsupported", d->dqk, d->dv);
}
