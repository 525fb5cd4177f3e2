// SAM 2.1 path helpers (everything that is not a GEMM or an attention): LayerNorm, 2x2 max-pool, casts,
// the mask-decoder tail (hypernetwork product + stability selection), bilinear resizes, the fused
// upsample + multi-kernel refinement head, and the antialiased input transform.
#include "common.hpp"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// ---- LayerNorm: a row is shared by G lanes (G = 4..64, chosen so that C/VEC chunks fill G * NCH slots tightly),
//      64/G rows per wave; the row stays in registers between the mean and the variance pass.
template <typename TI, typename TO, int NCH, int WD>
__global__ __launch_bounds__(256) void layernorm_kernel(const TI* __restrict__ x, int x_ld, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, TO* __restrict__ y, int y_ld, long long rows, int C,
                                                       float eps, int act, int G, int pw, int pwp, int phw, int phpwp,
                                                       f16* __restrict__ y2, int y2_ld) {
  // y2 (optional): a second, fp16 copy of the output rows (the GEMM-operand copy of an f32 stream -- saves the cast pass)
  // pw > 0: rows are pixels of [*, H, W] images (phw = H*W, pw = W) and are written into a zero-padded
  // [*, Hp, Wp] grid (phpwp = Hp*Wp, pwp = Wp) -- Hiera's pad-to-window-multiple, applied after the norm
  // WD = 16-byte input chunks per slot: 2 for f32 -> f16, so that a slot's 8 outputs leave as ONE 16-byte store
  constexpr int VI = Elem<TI>::VEC, SV = VI * WD;
  const int lane = threadIdx.x & 63;
  const int rpw = 64 / G;
  const long long row_raw = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * rpw + lane / G;
  const bool row_ok = row_raw < rows;
  const long long row = row_ok ? row_raw : rows - 1;
  const int g = lane % G;
  const TI* xr = x + row * x_ld;
  float v[NCH][SV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (i * G + g) * SV;
    if (c < C) {
#pragma unroll
      for (int w = 0; w < WD; ++w) unpack16<TI>(*reinterpret_cast<const u32x4*>(xr + c + w * VI), v[i] + w * VI);
#pragma unroll
      for (int e = 0; e < SV; ++e) s += v[i][e];
    }
  }
  for (int off = G >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off);
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (i * G + g) * SV;
    if (c < C) {
#pragma unroll
      for (int e = 0; e < SV; ++e) { const float d = v[i][e] - mean; q += d * d; }
    }
  }
  for (int off = G >> 1; off > 0; off >>= 1) q += __shfl_xor(q, off);
  const float rstd = 1.0f / sqrtf(q / (float)C + eps);
  if (!row_ok) return;
  long long orow = row;
  if (pw > 0) {
    const long long img = row / phw;
    const int rem = (int)(row - img * phw);
    orow = img * phpwp + (long long)(rem / pw) * pwp + rem % pw;
  }
  TO* yr = y + orow * y_ld;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (i * G + g) * SV;
    if (c < C) {
      struct alignas(sizeof(TO) * SV > 16 ? 16 : sizeof(TO) * SV) Pack { TO v[SV]; } pk;
      float gm[SV], bt[SV];
#pragma unroll
      for (int h = 0; h < SV / 4; ++h) {
        *reinterpret_cast<f32x4*>(gm + 4 * h) = *reinterpret_cast<const f32x4*>(gamma + c + 4 * h);
        *reinterpret_cast<f32x4*>(bt + 4 * h) = *reinterpret_cast<const f32x4*>(beta + c + 4 * h);
      }
      if (act == CVMI_ACT_NONE) {                       // wave-uniform: no per-element activation switch on the common path
#pragma unroll
        for (int e = 0; e < SV; ++e) pk.v[e] = (TO)((v[i][e] - mean) * rstd * gm[e] + bt[e]);
      } else {
#pragma unroll
        for (int e = 0; e < SV; ++e) pk.v[e] = (TO)act_apply<false>((v[i][e] - mean) * rstd * gm[e] + bt[e], act);
      }
      *reinterpret_cast<Pack*>(yr + c) = pk;
      if (y2) {
        struct alignas(2 * SV > 16 ? 16 : 2 * SV) Pack2 { f16 v[SV]; } p2;
#pragma unroll
        for (int e = 0; e < SV; ++e) p2.v[e] = (f16)(float)pk.v[e];
        *reinterpret_cast<Pack2*>(y2 + orow * y2_ld + c) = p2;
      }
    }
  }
}

// ---- 2x2 / s2 max-pool (NHWC) --------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_kernel(const char* __restrict__ x, int x_ld, char* __restrict__ y, int y_ld, int B, int H, int W, int C) {
  constexpr int VEC = Elem<T>::VEC;
  const int nch = C / VEC, OH = H / 2, OW = W / 2;
  const long long total = (long long)B * OH * OW * nch;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int ch = (int)(idx % nch);
    const long long pix = idx / nch;
    const int ox = (int)(pix % OW);
    const long long t = pix / OW;
    const int oy = (int)(t % OH);
    const long long b = t / OH;
    float m[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) m[e] = -INFINITY;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        float v[VEC];
        const size_t ip = ((size_t)b * H + 2 * oy + dy) * W + 2 * ox + dx;
        unpack16<T>(*reinterpret_cast<const u32x4*>(x + (ip * x_ld + ch * VEC) * sizeof(T)), v);
#pragma unroll
        for (int e = 0; e < VEC; ++e) m[e] = fmaxf(m[e], v[e]);
      }
    *reinterpret_cast<u32x4*>(y + ((size_t)pix * y_ld + ch * VEC) * sizeof(T)) = pack16<T>(m);
  }
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void cast_kernel(const TI* __restrict__ x, int x_ld, TO* __restrict__ y, int y_ld, long long rows, int C) {
  const long long total = rows * C;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const long long r = idx / C;
    const int c = (int)(idx - r * C);
    y[r * y_ld + c] = (TO)(float)x[r * x_ld + c];
  }
}

// ---- prompt tokens (upstream PromptEncoder._embed_points + decoder token concat) ------------------------------
// one workgroup of 256 threads per prompt; thread c owns channel c of every token
template <typename TL>
__global__ __launch_bounds__(256) void prompt_tokens_kernel(const float* __restrict__ coords, const int* __restrict__ labels, const float* __restrict__ gauss,
                                                           const float* __restrict__ out_tokens, const float* __restrict__ table, float size,
                                                           float* __restrict__ tok, TL* __restrict__ tok_lp, int K, int T0) {
  const int i = blockIdx.x, c = threadIdx.x, T = T0 + K;
  float* t = tok + (size_t)i * T * 256;
  TL* tl = tok_lp ? tok_lp + (size_t)i * T * 256 : nullptr;
  for (int r = 0; r < T0; ++r) {
    const float v = out_tokens[r * 256 + c];
    t[r * 256 + c] = v;
    if (tl) tl[r * 256 + c] = (TL)v;
  }
  const int g = c & 127;
  const float g0 = gauss[g], g1 = gauss[128 + g];
  for (int j = 0; j < K; ++j) {
    const int lab = labels[(size_t)i * K + j];
    float v;
    if (lab < 0) {
      v = table[c];
    } else {
      const float x = 2.f * ((coords[((size_t)i * K + j) * 2] + 0.5f) / size) - 1.f;
      const float y = 2.f * ((coords[((size_t)i * K + j) * 2 + 1] + 0.5f) / size) - 1.f;
      const float a = 6.283185307179586f * (x * g0 + y * g1);
      v = (c < 128 ? sinf(a) : cosf(a)) + table[(1 + lab) * 256 + c];
    }
    t[(T0 + j) * 256 + c] = v;
    if (tl) tl[(T0 + j) * 256 + c] = (TL)v;
  }
}

__global__ __launch_bounds__(256) void repeat_images_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, long long chunks, int rep) {
  // grid.y = image; every 16-byte chunk is read once and stored `rep` times
  const int b = blockIdx.y;
  const u32x4* s = src + (size_t)b * chunks;
  u32x4* d = dst + (size_t)b * rep * chunks;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < chunks; i += (long long)gridDim.x * blockDim.x) {
    const u32x4 v = s[i];
    for (int r = 0; r < rep; ++r) d[(size_t)r * chunks + i] = v;
  }
}

// ---- space-to-depth(4) of a 3-channel NHWC image: [B,H,W,3] -> [B,H/4,W/4,48], channel = (sy*4 + sx)*3 + c.
//      Turns Hiera's 7x7 / stride-4 / 3-channel patch embedding (a per-element gather) into a 2x2 conv over 48 channels.
template <typename T>
__global__ __launch_bounds__(256) void s2d4_kernel(const T* __restrict__ x, T* __restrict__ y, long long blocks, int H, int W) {
  // one thread per (block, sy): 4 pixels x 3 channels = 12 contiguous input elements -> 12 contiguous outputs
  const long long total = blocks * 4;
  const int bw = W / 4, bh = H / 4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int sy = (int)(i & 3);
    const long long blk = i >> 2;
    const int bx = (int)(blk % bw);
    const long long t = blk / bw;
    const int by = (int)(t % bh);
    const long long b = t / bh;
    const T* src = x + ((b * H + by * 4 + sy) * W + bx * 4) * 3;
    T* dst = y + blk * 48 + sy * 12;
#pragma unroll
    for (int e = 0; e < 12; ++e) dst[e] = src[e];
  }
}

// ---- extent of a binary mask: min / max x, y over the non-zero pixels of each plane ----------------------------------
// (circuit_analyzer.py:364-370: cv2.findContours(EXTERNAL) + boundingRect over all contour points == the bounding
// rectangle of the non-zero pixels)
__global__ __launch_bounds__(256) void mask_extent_kernel(const uint8_t* __restrict__ m, int H, int W, int* __restrict__ ext) {
  const int n = blockIdx.y;
  const uint8_t* pl = m + (size_t)n * H * W;
  int x0 = W, y0 = H, x1 = -1, y1 = -1;
  const long long total = (long long)H * W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    if (pl[i]) {
      const int y = (int)(i / W), x = (int)(i - (long long)y * W);
      x0 = x < x0 ? x : x0; x1 = x > x1 ? x : x1; y0 = y < y0 ? y : y0; y1 = y > y1 ? y : y1;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    x0 = min(x0, __shfl_xor(x0, off)); y0 = min(y0, __shfl_xor(y0, off));
    x1 = max(x1, __shfl_xor(x1, off)); y1 = max(y1, __shfl_xor(y1, off));
  }
  if ((threadIdx.x & 63) == 0 && x1 >= 0) {
    atomicMin(ext + n * 4 + 0, x0); atomicMin(ext + n * 4 + 1, y0);
    atomicMax(ext + n * 4 + 2, x1); atomicMax(ext + n * 4 + 3, y1);
  }
}
__global__ void mask_extent_init_kernel(int* ext, int N, int H, int W) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) { ext[i * 4] = W; ext[i * 4 + 1] = H; ext[i * 4 + 2] = -1; ext[i * 4 + 3] = -1; }
}

// ---- mask decoder tail ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void zero_i32_kernel(int* __restrict__ p, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0;
}

template <typename T>
__global__ __launch_bounds__(256) void hyper_masks_kernel(const float* __restrict__ hyper, int hyper_ld, const T* __restrict__ up, int up_ld, int C,
                                                         float* __restrict__ masks, int* __restrict__ areas, int P, float delta) {
  __shared__ float hs[4 * 64];
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < 4 * C; i += blockDim.x) hs[i] = hyper[((size_t)b * 4 + i / C) * hyper_ld + (i % C)];
  __syncthreads();
  int ai = 0, au = 0;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
    const T* u = up + ((size_t)b * P + p) * up_ld;
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
    for (int c = 0; c < C; ++c) {
      const float v = (float)u[c];
      m0 = fmaf(hs[c], v, m0); m1 = fmaf(hs[C + c], v, m1); m2 = fmaf(hs[2 * C + c], v, m2); m3 = fmaf(hs[3 * C + c], v, m3);
    }
    float* o = masks + (size_t)b * 4 * P + p;
    o[0] = m0; o[(size_t)P] = m1; o[(size_t)2 * P] = m2; o[(size_t)3 * P] = m3;
    ai += m0 > delta;
    au += m0 > -delta;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { ai += __shfl_xor(ai, off); au += __shfl_xor(au, off); }
  if ((threadIdx.x & 63) == 0) { atomicAdd(&areas[b * 2], ai); atomicAdd(&areas[b * 2 + 1], au); }
}

__global__ __launch_bounds__(256) void select_mask_kernel(const float* __restrict__ masks, const int* __restrict__ areas, const float* __restrict__ iou,
                                                         int iou_ld, int dynamic, float thresh, float* __restrict__ low, float* __restrict__ iou_out,
                                                         int* __restrict__ sel, int P) {
  const int b = blockIdx.y;
  int idx = 0;
  if (dynamic) {
    const float ai = (float)areas[b * 2], au = (float)areas[b * 2 + 1];
    const float stab = au > 0.f ? ai / au : 1.0f;
    if (!(stab >= thresh)) {
      const float* io = iou + (size_t)b * iou_ld;
      idx = 1;
      float best = io[1];
      if (io[2] > best) { best = io[2]; idx = 2; }
      if (io[3] > best) { best = io[3]; idx = 3; }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { sel[b] = idx; iou_out[b] = iou[(size_t)b * iou_ld + idx]; }
  const float* src = masks + ((size_t)b * 4 + idx) * P;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) low[(size_t)b * P + p] = src[p];
}

// ---- bilinear, align_corners = False (PyTorch semantics) ----------------------------------------------------
__device__ __forceinline__ void bil_axis(int d, float scale, int in, int& i0, int& i1, float& l1) {
  float s = ((float)d + 0.5f) * scale - 0.5f;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = s - (float)i0;
}

__device__ __forceinline__ float bil_sample(const float* __restrict__ pl, int w, int y0, int y1, float ly, int x0, int x1, float lx) {
  const float hy = 1.f - ly, hx = 1.f - lx;
  return hy * (hx * pl[(size_t)y0 * w + x0] + lx * pl[(size_t)y0 * w + x1]) + ly * (hx * pl[(size_t)y1 * w + x0] + lx * pl[(size_t)y1 * w + x1]);
}

// y (f32 map), mask (u8 {0,255} of v > thresh) and ext (per-plane extent of the mask, pre-initialised to {W, H, -1, -1}) are each
// optional: the reference's post-processing (circuit_analyzer.py:354-370) only needs the u8 mask and its bounding rectangle, so the
// f32 [N,H,W] map need not be written at all.
__global__ __launch_bounds__(256) void bilinear_kernel(const float* __restrict__ x, int h, int w, float* __restrict__ y, int H, int W,
                                                      uint8_t* __restrict__ mask, float thresh, float sy, float sx, int* __restrict__ ext) {
  const int n = blockIdx.y;
  const float* pl = x + (size_t)n * h * w;
  const int total = H * W;
  int ex0 = W, ey0 = H, ex1 = -1, ey1 = -1;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int oy = idx / W, ox = idx - oy * W;
    int y0, y1, x0, x1; float ly, lx;
    bil_axis(oy, sy, h, y0, y1, ly);
    bil_axis(ox, sx, w, x0, x1, lx);
    const float v = bil_sample(pl, w, y0, y1, ly, x0, x1, lx);
    if (y) y[(size_t)n * total + idx] = v;
    const bool on = v > thresh;
    if (mask) mask[(size_t)n * total + idx] = on ? 255 : 0;
    if (on) { ex0 = min(ex0, ox); ex1 = max(ex1, ox); ey0 = min(ey0, oy); ey1 = max(ey1, oy); }
  }
  if (ext) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      ex0 = min(ex0, __shfl_xor(ex0, off)); ey0 = min(ey0, __shfl_xor(ey0, off));
      ex1 = max(ex1, __shfl_xor(ex1, off)); ey1 = max(ey1, __shfl_xor(ey1, off));
    }
    if ((threadIdx.x & 63) == 0 && ex1 >= 0) {
      atomicMin(ext + n * 4 + 0, ex0); atomicMin(ext + n * 4 + 1, ey0);
      atomicMax(ext + n * 4 + 2, ex1); atomicMax(ext + n * 4 + 3, ey1);
    }
  }
}

// ---- fused upsample + MultiKernelRefinement -------------------------------------------------------------------
// 16x16 output tile per workgroup; the bilinear-upsampled tile (+ halo of max(ks)/2, zero outside the
// image = the convs' 'same' zero padding) lives in LDS; every thread walks the largest window once and
// feeds each branch whose kernel covers the tap.
constexpr int RF_TILE = 16;
constexpr int RF_MAXK = 4;
struct RefineArgs { int ks[RF_MAXK]; int woff[RF_MAXK]; int boff[RF_MAXK]; int nk, ic, comb_w, comb_b, halo; };

template <int IC>
__global__ __launch_bounds__(256) void upsample_refine_kernel(const float* __restrict__ low, int h, int w, float* __restrict__ high, int H, int W,
                                                             const float* __restrict__ prm, const RefineArgs a, float sy, float sx) {
  extern __shared__ __attribute__((aligned(16))) float tile[];
  const int halo = a.halo, TS = RF_TILE + 2 * halo, TSP = TS + 1;
  const int n = blockIdx.z;
  const int ty0 = blockIdx.y * RF_TILE - halo, tx0 = blockIdx.x * RF_TILE - halo;
  const float* pl = low + (size_t)n * h * w;
  for (int i = threadIdx.x; i < TS * TS; i += 256) {
    const int ly = i / TS, lx = i - ly * TS;
    const int gy = ty0 + ly, gx = tx0 + lx;
    float v = 0.f;
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
      int y0, y1, x0, x1; float fy, fx;
      bil_axis(gy, sy, h, y0, y1, fy);
      bil_axis(gx, sx, w, x0, x1, fx);
      v = bil_sample(pl, w, y0, y1, fy, x0, x1, fx);
    }
    tile[ly * TSP + lx] = v;
  }
  __syncthreads();
  const int py = threadIdx.x / RF_TILE, px = threadIdx.x % RF_TILE;
  const int oy = blockIdx.y * RF_TILE + py, ox = blockIdx.x * RF_TILE + px;
  float acc[RF_MAXK][IC];
#pragma unroll
  for (int j = 0; j < RF_MAXK; ++j)
#pragma unroll
    for (int c = 0; c < IC; ++c) acc[j][c] = j < a.nk ? prm[a.boff[j] + c] : 0.f;
  for (int dy = -halo; dy <= halo; ++dy) {
    for (int dx = -halo; dx <= halo; ++dx) {
      const float v = tile[(py + halo + dy) * TSP + px + halo + dx];
      const int r = max(abs(dy), abs(dx));
#pragma unroll
      for (int j = 0; j < RF_MAXK; ++j) {
        if (j < a.nk) {
          const int k = a.ks[j], kh = k >> 1;
          if (r <= kh) {
            const float* wp = prm + a.woff[j] + (dy + kh) * k + (dx + kh);
#pragma unroll
            for (int c = 0; c < IC; ++c) acc[j][c] = fmaf(wp[c * k * k], v, acc[j][c]);
          }
        }
      }
    }
  }
  if (oy < H && ox < W) {
    float out = prm[a.comb_b];
#pragma unroll
    for (int j = 0; j < RF_MAXK; ++j)
      if (j < a.nk) {
#pragma unroll
        for (int c = 0; c < IC; ++c) out = fmaf(prm[a.comb_w + j * IC + c], act_apply<false>(acc[j][c], CVMI_ACT_GELU), out);
      }
    high[((size_t)n * H + oy) * W + ox] = out;
  }
}

// Fast path for the reference's head (kernels 3,5,7,11 x 4 channels, sam2_infer.py:211-216): a workgroup owns a
// 16 x 64 output tile, a thread a 1 x 4 strip; weights are re-laid out [tap][channel] in LDS so one broadcast
// ds_read_b128 feeds 16 FMAs; the 11 rows of the window are walked once and shared by the four branches.
constexpr int RF_HALO = 5, RF_TW = 64, RF_TH = 16, RF_TS = RF_TW + 2 * RF_HALO + 2;     // 76: row stride (floats)

template <int K>
__device__ __forceinline__ void refine_branch_row(const float* __restrict__ r, const float* __restrict__ wrow, float (&acc)[4][4]) {
  constexpr int kh = K / 2;
#pragma unroll
  for (int dx = 0; dx < K; ++dx) {
    const f32x4 w = *reinterpret_cast<const f32x4*>(wrow + dx * 4);
#pragma unroll
    for (int px = 0; px < 4; ++px) {
      const float v = r[px + dx + (RF_HALO - kh)];
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c][px] = fmaf(w[c], v, acc[c][px]);
    }
  }
}

__global__ __launch_bounds__(256) void upsample_refine_fast_kernel(const float* __restrict__ low, int h, int w, float* __restrict__ high, int H, int W,
                                                                  const float* __restrict__ prm, float sy, float sx) {
  constexpr int KS[4] = {3, 5, 7, 11};
  constexpr int WOFF[4] = {0, 3 * 3 * 4 + 4, 3 * 3 * 4 + 4 + 5 * 5 * 4 + 4, 3 * 3 * 4 + 4 + 5 * 5 * 4 + 4 + 7 * 7 * 4 + 4};   // offsets in prm (w then b)
  constexpr int NW = 9 * 4 + 25 * 4 + 49 * 4 + 121 * 4;                                                               // 816 weights
  __shared__ __attribute__((aligned(16))) float wl[NW];                       // [branch][tap][channel]
  __shared__ __attribute__((aligned(16))) float tile[(RF_TH + 2 * RF_HALO) * RF_TS];
  const int n = blockIdx.z, tid = threadIdx.x;
  {
    int o = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kk = KS[j] * KS[j];
      for (int i = tid; i < kk * 4; i += 256) wl[o + i] = prm[WOFF[j] + (i & 3) * kk + (i >> 2)];
      o += kk * 4;
    }
  }
  const int ty0 = blockIdx.y * RF_TH - RF_HALO, tx0 = blockIdx.x * RF_TW - RF_HALO;
  const float* pl = low + (size_t)n * h * w;
  constexpr int TSW = RF_TW + 2 * RF_HALO;
  for (int i = tid; i < (RF_TH + 2 * RF_HALO) * TSW; i += 256) {
    const int ly = i / TSW, lx = i - ly * TSW;
    const int gy = ty0 + ly, gx = tx0 + lx;
    float v = 0.f;
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
      int y0, y1, x0, x1; float fy, fx;
      bil_axis(gy, sy, h, y0, y1, fy);
      bil_axis(gx, sx, w, x0, x1, fx);
      v = bil_sample(pl, w, y0, y1, fy, x0, x1, fx);
    }
    tile[ly * RF_TS + lx] = v;
  }
  __syncthreads();
  const int py = tid >> 4, px0 = (tid & 15) * 4;
  float acc[4][4][4];                                   // [branch][channel][pixel]
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[j][c][q] = prm[WOFF[j] + KS[j] * KS[j] * 4 + c];
  for (int dy = -RF_HALO; dy <= RF_HALO; ++dy) {
    float r[4 + 2 * RF_HALO];
    const float* tr = tile + (py + RF_HALO + dy) * RF_TS + px0;
#pragma unroll
    for (int i = 0; i < 4 + 2 * RF_HALO; ++i) r[i] = tr[i];
    const int ady = dy < 0 ? -dy : dy;
    refine_branch_row<11>(r, wl + (9 + 25 + 49) * 4 + (dy + 5) * 11 * 4, acc[3]);
    if (ady <= 3) refine_branch_row<7>(r, wl + (9 + 25) * 4 + (dy + 3) * 7 * 4, acc[2]);
    if (ady <= 2) refine_branch_row<5>(r, wl + 9 * 4 + (dy + 2) * 5 * 4, acc[1]);
    if (ady <= 1) refine_branch_row<3>(r, wl + (dy + 1) * 3 * 4, acc[0]);
  }
  const int oy = blockIdx.y * RF_TH + py, ox = blockIdx.x * RF_TW + px0;
  if (oy >= H) return;
  constexpr int CW = WOFF[3] + 121 * 4 + 4;             // combiner weights, then bias
  float out[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) out[q] = prm[CW + 16];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float cw = prm[CW + j * 4 + c];
#pragma unroll
      for (int q = 0; q < 4; ++q) out[q] = fmaf(cw, act_apply<false>(acc[j][c][q], CVMI_ACT_GELU), out[q]);
    }
  float* op = high + ((size_t)n * H + oy) * W + ox;
  if (ox + 4 <= W && (W & 3) == 0) *reinterpret_cast<f32x4*>(op) = f32x4{out[0], out[1], out[2], out[3]};
  else
    for (int q = 0; q < 4; ++q) if (ox + q < W) op[q] = out[q];
}

// ---- SAM2Transforms.__call__: /255, antialiased bilinear (torch _upsample_bilinear2d_aa), normalise -----------
constexpr int AA_MAXTAPS = 24;
__device__ __forceinline__ void aa_axis(int i, float scale, int in, int& xmin, int& xsize, float* wt) {
  const float support = scale >= 1.f ? scale : 1.f;
  const float invscale = scale >= 1.f ? 1.f / scale : 1.f;
  const float center = scale * ((float)i + 0.5f);
  xmin = max((int)(center - support + 0.5f), 0);
  xsize = min((int)(center + support + 0.5f), in) - xmin;
  if (xsize > AA_MAXTAPS) xsize = AA_MAXTAPS;
  float total = 0.f;
  for (int j = 0; j < xsize; ++j) {
    float t = ((float)(j + xmin) - center + 0.5f) * invscale;
    t = fabsf(t);
    const float wv = t < 1.f ? 1.f - t : 0.f;
    wt[j] = wv;
    total += wv;
  }
  const float ws = total != 0.f ? 1.f / total : 0.f;
  for (int j = 0; j < xsize; ++j) wt[j] *= ws;
}

template <typename T>
__global__ __launch_bounds__(256) void sam2_transform_kernel(const uint8_t* __restrict__ src, int H, int W, T* __restrict__ dst, int R, float sy, float sx,
                                                            int swap_rb) {
  src += (size_t)blockIdx.y * H * W * 3;                  // blockIdx.y = image of a batch of equally sized sources
  dst += (size_t)blockIdx.y * R * R * 3;
  const int c0 = swap_rb ? 2 : 0, c2 = swap_rb ? 0 : 2;   // swap_rb: output channel c reads source channel 2 - c (the caller's cv2.COLOR_BGR2RGB)
  const int total = R * R;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int oy = idx / R, ox = idx - oy * R;
    int ymin, ysize, xmin, xsize;
    float wy[AA_MAXTAPS], wx[AA_MAXTAPS];
    aa_axis(oy, sy, H, ymin, ysize, wy);
    aa_axis(ox, sx, W, xmin, xsize, wx);
    float acc[3] = {0.f, 0.f, 0.f};
    for (int j = 0; j < ysize; ++j) {
      float row[3] = {0.f, 0.f, 0.f};
      const uint8_t* sp = src + ((size_t)(ymin + j) * W + xmin) * 3;
      for (int i = 0; i < xsize; ++i) {
#pragma unroll
        for (int c = 0; c < 3; ++c) row[c] += wx[i] * ((float)sp[i * 3 + (c == 0 ? c0 : c == 2 ? c2 : 1)] / 255.0f);
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[c] += wy[j] * row[c];
    }
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
#pragma unroll
    for (int c = 0; c < 3; ++c) dst[(size_t)idx * 3 + c] = (T)((acc[c] - mean[c]) / stdv[c]);
  }
}

// The same on a WINDOW of each source image (the crop between the two stages, circuit_analyzer.py:937-1284 as called at
// analysis_pipeline.py:177, without a cropped copy): image b of the launch is the rectangle win[b] = {x0, y0, w, h} of source image b.  The
// table travels by value in the kernel arguments (<= 64 images per launch).
constexpr int RECT_MAX = 64;
struct RectTable { int4 r[RECT_MAX]; };

template <typename T>
__global__ __launch_bounds__(256) void sam2_transform_rects_kernel(const uint8_t* __restrict__ src, long long image_stride, int W, const RectTable win,
                                                                  T* __restrict__ dst, int R, int swap_rb) {
  const int4 rc = win.r[blockIdx.y];                      // blockIdx.y = image
  const int H0 = rc.w, W0 = rc.z;                         // the window's height, width: the resize sees nothing outside it
  const float sy = (float)H0 / (float)R, sx = (float)W0 / (float)R;
  src += (size_t)blockIdx.y * image_stride + ((size_t)rc.y * W + rc.x) * 3;
  dst += (size_t)blockIdx.y * R * R * 3;
  const int c0 = swap_rb ? 2 : 0, c2 = swap_rb ? 0 : 2;
  const int total = R * R;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int oy = idx / R, ox = idx - oy * R;
    int ymin, ysize, xmin, xsize;
    float wy[AA_MAXTAPS], wx[AA_MAXTAPS];
    aa_axis(oy, sy, H0, ymin, ysize, wy);
    aa_axis(ox, sx, W0, xmin, xsize, wx);
    float acc[3] = {0.f, 0.f, 0.f};
    for (int j = 0; j < ysize; ++j) {
      float row[3] = {0.f, 0.f, 0.f};
      const uint8_t* sp = src + ((size_t)(ymin + j) * W + xmin) * 3;
      for (int i = 0; i < xsize; ++i) {
#pragma unroll
        for (int c = 0; c < 3; ++c) row[c] += wx[i] * ((float)sp[i * 3 + (c == 0 ? c0 : c == 2 ? c2 : 1)] / 255.0f);
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[c] += wy[j] * row[c];
    }
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
#pragma unroll
    for (int c = 0; c < 3; ++c) dst[(size_t)idx * 3 + c] = (T)((acc[c] - mean[c]) / stdv[c]);
  }
}

// cvmi_mask_postprocess for planes that go back to DIFFERENT sizes (each image's own crop window): plane n is resized to sz.r[n] = {H, W} and
// written at mask + sz.r[n].z (a byte offset the host laid out: planes packed back to back); extents pre-initialised by the kernel's first block.
__global__ __launch_bounds__(256) void bilinear_sizes_kernel(const float* __restrict__ x, int h, int w, const RectTable sz, uint8_t* __restrict__ mask,
                                                            long long mask_base, float thresh, int* __restrict__ ext) {
  const int n = blockIdx.y;
  const int H = sz.r[n].x, W = sz.r[n].y;
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  const float* pl = x + (size_t)n * h * w;
  uint8_t* mp = mask + mask_base + (((long long)(unsigned)sz.r[n].w << 32) | (unsigned)sz.r[n].z);
  const int total = H * W;
  int ex0 = W, ey0 = H, ex1 = -1, ey1 = -1;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int oy = idx / W, ox = idx - oy * W;
    int y0, y1, x0, x1; float ly, lx;
    bil_axis(oy, sy, h, y0, y1, ly);
    bil_axis(ox, sx, w, x0, x1, lx);
    const bool on = bil_sample(pl, w, y0, y1, ly, x0, x1, lx) > thresh;
    mp[idx] = on ? 255 : 0;
    if (on) { ex0 = min(ex0, ox); ex1 = max(ex1, ox); ey0 = min(ey0, oy); ey1 = max(ey1, oy); }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    ex0 = min(ex0, __shfl_xor(ex0, off)); ey0 = min(ey0, __shfl_xor(ey0, off));
    ex1 = max(ex1, __shfl_xor(ex1, off)); ey1 = max(ey1, __shfl_xor(ey1, off));
  }
  if ((threadIdx.x & 63) == 0 && ex1 >= 0) {
    atomicMin(ext + n * 4 + 0, ex0); atomicMin(ext + n * 4 + 1, ey0);
    atomicMax(ext + n * 4 + 2, ex1); atomicMax(ext + n * 4 + 3, ey1);
  }
}
__global__ void mask_extent_init_sizes_kernel(int* ext, int N, const RectTable sz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) { ext[i * 4] = sz.r[i].y; ext[i * 4 + 1] = sz.r[i].x; ext[i * 4 + 2] = -1; ext[i * 4 + 3] = -1; }
}

inline int grid_for(long long total, int block = 256, int cap = 256 * 16) {
  long long g = (total + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

template <typename TI, typename TO>
int launch_ln(const void* x, int x_ld, const float* gamma, const float* beta, void* y, int y_ld, long long rows, int C, float eps, int act,
              int pw, int pwp, int phw, int phpwp, hipStream_t stream, void* y2 = nullptr, int y2_ld = 0) {
  constexpr int VI = Elem<TI>::VEC;
  // f32 -> f16 with C a multiple of 8: slots of 8 channels (two 16-byte loads, one 16-byte store)
  constexpr bool CAN_WIDE = sizeof(TI) == 4 && sizeof(TO) == 2;
  const bool wide = CAN_WIDE && C % 8 == 0 && y_ld % 8 == 0;
  const int chunks = C / (wide ? 2 * VI : VI);
  // lanes per row: the tightest fit of `chunks` into G * NCH slots with NCH in {1,2,3,5,8,9}
  int bestG = 64, bestN = 9, bestWaste = 1 << 30;
  const int ncand[6] = {1, 2, 3, 5, 8, 9};
  for (int G = 4; G <= 64; G <<= 1)
    for (int k = 0; k < 6; ++k) {
      const int n = ncand[k];
      if (G * n < chunks) continue;
      const int waste = G * n - chunks;
      if (waste < bestWaste || (waste == bestWaste && n < bestN)) { bestWaste = waste; bestG = G; bestN = n; }
      break;
    }
  CVMI_CHECK(bestG * bestN >= chunks, "layernorm: C=%d too wide", C);
  const int rpw = 64 / bestG;
  const dim3 g((unsigned)((rows + 4 * rpw - 1) / (4 * rpw))), b(256);
#define CVMI_LN(N, W) hipLaunchKernelGGL((layernorm_kernel<TI, TO, N, W>), g, b, 0, stream, (const TI*)x, x_ld, gamma, beta, (TO*)y, y_ld, rows, C, eps, act, bestG, pw, pwp, phw, phpwp, (f16*)y2, y2_ld)
#define CVMI_LN_SW(W)                                                                                                    \
  switch (bestN) {                                                                                                       \
    case 1: CVMI_LN(1, W); break;                                                                                        \
    case 2: CVMI_LN(2, W); break;                                                                                        \
    case 3: CVMI_LN(3, W); break;                                                                                        \
    case 5: CVMI_LN(5, W); break;                                                                                        \
    case 8: CVMI_LN(8, W); break;                                                                                        \
    default: CVMI_LN(9, W); break;                                                                                       \
  }
  if constexpr (CAN_WIDE) {
    if (wide) { CVMI_LN_SW(2) } else { CVMI_LN_SW(1) }
  } else {
    CVMI_LN_SW(1)
  }
#undef CVMI_LN_SW
#undef CVMI_LN
  CVMI_LAUNCH_CHECK();
  return 0;
}

}  // namespace

#ifndef CVMI_OPERAND_BF16
extern "C" int cvmi_layernorm_bf16(const void* x, int x_ld, int x_dtype, const float* gamma, const float* beta, void* y, int y_ld, int y_dtype,
                              long long rows, int C, float eps, int act, int pad_h, int pad_w, int pad_hp, int pad_wp, cvmi_stream_t stream_);
#endif
extern "C" int CVMI_ENTRY(cvmi_layernorm)(const void* x, int x_ld, int x_dtype, const float* gamma, const float* beta, void* y, int y_ld, int y_dtype,
                              long long rows, int C, float eps, int act, int pad_h, int pad_w, int pad_hp, int pad_wp, cvmi_stream_t stream_) {
#ifndef CVMI_OPERAND_BF16
  if (x_dtype == CVMI_BF16 || y_dtype == CVMI_BF16) return cvmi_layernorm_bf16(x, x_ld, x_dtype, gamma, beta, y, y_ld, y_dtype, rows, C, eps, act, pad_h, pad_w, pad_hp, pad_wp, stream_);
#endif
  CVMI_CHECK(x && gamma && beta && y && rows > 0 && C > 0, "layernorm: bad arguments");
  const int vi = x_dtype == CVMI_T16 ? 8 : 4;
  CVMI_CHECK((x_dtype == CVMI_T16 || x_dtype == CVMI_F32) && (y_dtype == CVMI_T16 || y_dtype == CVMI_F32), "layernorm: bad dtype");
  CVMI_CHECK(C % vi == 0 && x_ld % vi == 0 && y_ld % vi == 0 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0 && x_ld >= C && y_ld >= C, "layernorm: C=%d / ld not 16-byte aligned", C);
  int pw = 0, pwp = 0, phw = 0, phpwp = 0;
  if (pad_w > 0) {
    CVMI_CHECK(pad_h > 0 && pad_hp >= pad_h && pad_wp >= pad_w && rows % ((long long)pad_h * pad_w) == 0, "layernorm: bad padding geometry");
    pw = pad_w; pwp = pad_wp; phw = pad_h * pad_w; phpwp = pad_hp * pad_wp;
  }
  hipStream_t s = (hipStream_t)stream_;
  if (x_dtype == CVMI_F32 && y_dtype == CVMI_F32) return launch_ln<float, float>(x, x_ld, gamma, beta, y, y_ld, rows, C, eps, act, pw, pwp, phw, phpwp, s);
  if (x_dtype == CVMI_F32 && y_dtype == CVMI_T16) return launch_ln<float, f16>(x, x_ld, gamma, beta, y, y_ld, rows, C, eps, act, pw, pwp, phw, phpwp, s);
  if (x_dtype == CVMI_T16 && y_dtype == CVMI_T16) return launch_ln<f16, f16>(x, x_ld, gamma, beta, y, y_ld, rows, C, eps, act, pw, pwp, phw, phpwp, s);
  return launch_ln<f16, float>(x, x_ld, gamma, beta, y, y_ld, rows, C, eps, act, pw, pwp, phw, phpwp, s);
}

#ifndef CVMI_OPERAND_BF16
extern "C" int cvmi_layernorm_dual_bf16(const void* x, int x_ld, const float* gamma, const float* beta, void* y, int y_ld, void* y2, int y2_ld, int y2_dtype,
                                   long long rows, int C, float eps, cvmi_stream_t stream_);
#endif
extern "C" int CVMI_ENTRY(cvmi_layernorm_dual)(const void* x, int x_ld, const float* gamma, const float* beta, void* y, int y_ld, void* y2, int y2_ld, int y2_dtype,
                                   long long rows, int C, float eps, cvmi_stream_t stream_) {
#ifndef CVMI_OPERAND_BF16
  if (y2_dtype == CVMI_BF16) return cvmi_layernorm_dual_bf16(x, x_ld, gamma, beta, y, y_ld, y2, y2_ld, y2_dtype, rows, C, eps, stream_);
#endif
  CVMI_CHECK(x && gamma && beta && y && y2 && rows > 0 && C > 0 && y2_dtype == CVMI_T16, "layernorm_dual: bad arguments (y2_dtype must be a 16-bit type)");
  CVMI_CHECK(C % 8 == 0 && x_ld % 4 == 0 && y_ld % 4 == 0 && y2_ld % 8 == 0 && x_ld >= C && y_ld >= C && y2_ld >= C &&
             (((uintptr_t)x | (uintptr_t)y | (uintptr_t)y2 | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0, "layernorm_dual: C=%d / ld not 16-byte aligned", C);
  return launch_ln<float, float>(x, x_ld, gamma, beta, y, y_ld, rows, C, eps, CVMI_ACT_NONE, 0, 0, 0, 0, (hipStream_t)stream_, y2, y2_ld);
}

#ifndef CVMI_OPERAND_BF16
extern "C" int cvmi_maxpool2x2_bf16(const void* x, int x_ld, void* y, int y_ld, int B, int H, int W, int C, int dtype, cvmi_stream_t stream_);
#endif
extern "C" int CVMI_ENTRY(cvmi_maxpool2x2)(const void* x, int x_ld, void* y, int y_ld, int B, int H, int W, int C, int dtype, cvmi_stream_t stream_) {
#ifndef CVMI_OPERAND_BF16
  if (dtype == CVMI_BF16) return cvmi_maxpool2x2_bf16(x, x_ld, y, y_ld, B, H, W, C, dtype, stream_);
#endif
  CVMI_CHECK(x && y && B > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0, "maxpool2x2: bad shape");
  CVMI_CHECK(dtype == CVMI_T16 || dtype == CVMI_F32, "maxpool2x2: bad dtype");
  const int vec = dtype == CVMI_T16 ? 8 : 4;
  CVMI_CHECK(C % vec == 0 && x_ld % vec == 0 && y_ld % vec == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0, "maxpool2x2: not 16-byte aligned");
  const long long total = (long long)B * (H / 2) * (W / 2) * (C / vec);
  hipStream_t s = (hipStream_t)stream_;
  if (dtype == CVMI_T16) hipLaunchKernelGGL(maxpool2_kernel<f16>, dim3(grid_for(total)), dim3(256), 0, s, (const char*)x, x_ld, (char*)y, y_ld, B, H, W, C);
  else hipLaunchKernelGGL(maxpool2_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (const char*)x, x_ld, (char*)y, y_ld, B, H, W, C);
  CVMI_LAUNCH_CHECK();
  return 0;
}

#ifndef CVMI_OPERAND_BF16
extern "C" int cvmi_cast_bf16(const void* x, int x_ld, int x_dtype, void* y, int y_ld, int y_dtype, long long rows, int C, cvmi_stream_t stream_);
#endif
extern "C" int CVMI_ENTRY(cvmi_cast)(const void* x, int x_ld, int x_dtype, void* y, int y_ld, int y_dtype, long long rows, int C, cvmi_stream_t stream_) {
#ifndef CVMI_OPERAND_BF16
  if (x_dtype == CVMI_BF16 || y_dtype == CVMI_BF16) return cvmi_cast_bf16(x, x_ld, x_dtype, y, y_ld, y_dtype, rows, C, stream_);
#endif
  CVMI_CHECK(x && y && rows > 0 && C > 0 && x_ld >= C && y_ld >= C, "cast: bad arguments");
  hipStream_t s = (hipStream_t)stream_;
  const dim3 g(grid_for(rows * C)), b(256);
  if (x_dtype == CVMI_F32 && y_dtype == CVMI_T16) hipLaunchKernelGGL((cast_kernel<float, f16>), g, b, 0, s, (const float*)x, x_ld, (f16*)y, y_ld, rows, C);
  else if (x_dtype == CVMI_T16 && y_dtype == CVMI_F32) hipLaunchKernelGGL((cast_kernel<f16, float>), g, b, 0, s, (const f16*)x, x_ld, (float*)y, y_ld, rows, C);
  else if (x_dtype == CVMI_F32 && y_dtype == CVMI_F32) hipLaunchKernelGGL((cast_kernel<float, float>), g, b, 0, s, (const float*)x, x_ld, (float*)y, y_ld, rows, C);
  else if (x_dtype == CVMI_T16 && y_dtype == CVMI_T16) hipLaunchKernelGGL((cast_kernel<f16, f16>), g, b, 0, s, (const f16*)x, x_ld, (f16*)y, y_ld, rows, C);
  else CVMI_FAIL("cast: bad dtypes");
  CVMI_LAUNCH_CHECK();
  return 0;
}

#ifndef CVMI_OPERAND_BF16
extern "C" int cvmi_prompt_tokens_bf16(const float* coords, const int* labels, const float* gauss, const float* out_tokens, const float* table,
                                  float image_size, float* tokens_f32, void* tokens_lp, int dtype, int n, int K, int T0, cvmi_stream_t stream_);
#endif
extern "C" int CVMI_ENTRY(cvmi_prompt_tokens)(const float* coords, const int* labels, const float* gauss, const float* out_tokens, const float* table,
                                  float image_size, float* tokens_f32, void* tokens_lp, int dtype, int n, int K, int T0, cvmi_stream_t stream_) {
#ifndef CVMI_OPERAND_BF16
  if (dtype == CVMI_BF16) return cvmi_prompt_tokens_bf16(coords, labels, gauss, out_tokens, table, image_size, tokens_f32, tokens_lp, dtype, n, K, T0, stream_);
#endif
  CVMI_CHECK(coords && labels && gauss && out_tokens && table && tokens_f32 && n > 0 && K > 0 && T0 >= 0 && image_size > 0.f, "prompt_tokens: bad arguments");
  CVMI_CHECK(!tokens_lp || dtype == CVMI_T16 || dtype == CVMI_F32, "prompt_tokens: bad dtype");
  hipStream_t s = (hipStream_t)stream_;
  const dim3 g(n), b(256);
  const float inv = image_size;
  if (tokens_lp && dtype == CVMI_T16)
    hipLaunchKernelGGL(prompt_tokens_kernel<f16>, g, b, 0, s, coords, labels, gauss, out_tokens, table, inv, tokens_f32, (f16*)tokens_lp, K, T0);
  else
    hipLaunchKernelGGL(prompt_tokens_kernel<float>, g, b, 0, s, coords, labels, gauss, out_tokens, table, inv, tokens_f32, (float*)tokens_lp, K, T0);
  CVMI_LAUNCH_CHECK();
  return 0;
}

#ifndef CVMI_OPERAND_BF16
extern "C" int cvmi_repeat_images(const void* src, void* dst, long long bytes_per_image, int B, int rep, cvmi_stream_t stream_) {
  CVMI_CHECK(src && dst && bytes_per_image > 0 && bytes_per_image % 16 == 0 && B > 0 && B <= 65535 && rep > 0, "repeat_images: bad arguments");
  CVMI_CHECK((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "repeat_images: pointers must be 16-byte aligned");
  const long long chunks = bytes_per_image / 16;
  const dim3 g(grid_for(chunks, 256, 64), B), b(256);
  hipLaunchKernelGGL(repeat_images_kernel, g, b, 0, (hipStream_t)stream_, (const u32x4*)src, (u32x4*)dst, chunks, rep);
  CVMI_LAUNCH_CHECK();
  return 0;
}
#endif

#ifndef CVMI_OPERAND_BF16
extern "C" int cvmi_hyper_masks_bf16(const float* hyper, int hyper_ld, const void* up, int up_ld, int up_dtype, int C, float* masks, int* areas, int B,
                                int P, float delta, cvmi_stream_t stream_);
#endif
extern "C" int CVMI_ENTRY(cvmi_hyper_masks)(const float* hyper, int hyper_ld, const void* up, int up_ld, int up_dtype, int C, float* masks, int* areas, int B,
                                int P, float delta, cvmi_stream_t stream_) {
#ifndef CVMI_OPERAND_BF16
  if (up_dtype == CVMI_BF16) return cvmi_hyper_masks_bf16(hyper, hyper_ld, up, up_ld, up_dtype, C, masks, areas, B, P, delta, stream_);
#endif
  CVMI_CHECK(hyper && up && masks && areas && B > 0 && P > 0 && C > 0 && C <= 64 && hyper_ld >= C && up_ld >= C, "hyper_masks: bad arguments");
  CVMI_CHECK(up_dtype == CVMI_T16 || up_dtype == CVMI_F32, "hyper_masks: bad dtype");
  hipStream_t s = (hipStream_t)stream_;
  // a kernel, not hipMemsetAsync: a memset node of this small size did not replay reliably inside a captured graph
  hipLaunchKernelGGL(zero_i32_kernel, dim3((2 * B + 255) / 256), dim3(256), 0, s, areas, 2 * B);
  const dim3 g(grid_for(P, 256, 64), B), b(256);
  if (up_dtype == CVMI_T16) hipLaunchKernelGGL(hyper_masks_kernel<f16>, g, b, 0, s, hyper, hyper_ld, (const f16*)up, up_ld, C, masks, areas, P, delta);
  else hipLaunchKernelGGL(hyper_masks_kernel<float>, g, b, 0, s, hyper, hyper_ld, (const float*)up, up_ld, C, masks, areas, P, delta);
  CVMI_LAUNCH_CHECK();
  return 0;
}

#ifndef CVMI_OPERAND_BF16
extern "C" int cvmi_select_mask(const float* masks, const int* areas, const float* iou, int iou_ld, int dynamic, float thresh, float* low_res,
                                float* iou_out, int* sel, int B, int P, cvmi_stream_t stream_) {
  CVMI_CHECK(masks && areas && iou && low_res && iou_out && sel && B > 0 && P > 0 && iou_ld >= 4, "select_mask: bad arguments");
  hipLaunchKernelGGL(select_mask_kernel, dim3(grid_for(P, 256, 64), B), dim3(256), 0, (hipStream_t)stream_, masks, areas, iou, iou_ld, dynamic, thresh,
                     low_res, iou_out, sel, P);
  CVMI_LAUNCH_CHECK();
  return 0;
}
#endif

#ifndef CVMI_OPERAND_BF16
extern "C" int cvmi_bilinear_f32(const float* x, int N, int h, int w, float* y, int H, int W, uint8_t* mask_u8, float thresh, cvmi_stream_t stream_) {
  CVMI_CHECK(x && (y || mask_u8) && N > 0 && N <= 65535 && h > 0 && w > 0 && H > 0 && W > 0, "bilinear: bad arguments");
  hipLaunchKernelGGL(bilinear_kernel, dim3(grid_for((long long)H * W, 256, 1024), N), dim3(256), 0, (hipStream_t)stream_, x, h, w, y, H, W, mask_u8,
                     thresh, (float)h / (float)H, (float)w / (float)W, (int*)nullptr);
  CVMI_LAUNCH_CHECK();
  return 0;
}
#endif

#ifndef CVMI_OPERAND_BF16
extern "C" int cvmi_mask_postprocess(const float* x, int N, int h, int w, int H, int W, float thresh, uint8_t* mask_u8, int* extent,
                                     cvmi_stream_t stream_) {
  CVMI_CHECK(x && mask_u8 && extent && N > 0 && N <= 65535 && h > 0 && w > 0 && H > 0 && W > 0, "mask_postprocess: bad arguments");
  hipStream_t s = (hipStream_t)stream_;
  hipLaunchKernelGGL(mask_extent_init_kernel, dim3((N + 63) / 64), dim3(64), 0, s, extent, N, H, W);
  hipLaunchKernelGGL(bilinear_kernel, dim3(grid_for((long long)H * W, 256, 1024), N), dim3(256), 0, s, x, h, w, (float*)nullptr, H, W, mask_u8, thresh,
                     (float)h / (float)H, (float)w / (float)W, extent);
  CVMI_LAUNCH_CHECK();
  return 0;
}
#endif

#ifndef CVMI_OPERAND_BF16
extern "C" int cvmi_space_to_depth4_bf16(const void* x, void* y, int B, int H, int W, int dtype, cvmi_stream_t stream_);
#endif
extern "C" int CVMI_ENTRY(cvmi_space_to_depth4)(const void* x, void* y, int B, int H, int W, int dtype, cvmi_stream_t stream_) {
#ifndef CVMI_OPERAND_BF16
  if (dtype == CVMI_BF16) return cvmi_space_to_depth4_bf16(x, y, B, H, W, dtype, stream_);
#endif
  CVMI_CHECK(x && y && B > 0 && H > 0 && W > 0 && H % 4 == 0 && W % 4 == 0, "space_to_depth4: bad shape");
  CVMI_CHECK(dtype == CVMI_T16 || dtype == CVMI_F32, "space_to_depth4: bad dtype");
  const long long blocks = (long long)B * (H / 4) * (W / 4);
  const dim3 g(grid_for(blocks * 4)), b(256);
  if (dtype == CVMI_T16) hipLaunchKernelGGL(s2d4_kernel<f16>, g, b, 0, (hipStream_t)stream_, (const f16*)x, (f16*)y, blocks, H, W);
  else hipLaunchKernelGGL(s2d4_kernel<float>, g, b, 0, (hipStream_t)stream_, (const float*)x, (float*)y, blocks, H, W);
  CVMI_LAUNCH_CHECK();
  return 0;
}

#ifndef CVMI_OPERAND_BF16
extern "C" int cvmi_mask_extent(const uint8_t* mask, int N, int H, int W, int* extent, cvmi_stream_t stream_) {
  CVMI_CHECK(mask && extent && N > 0 && N <= 65535 && H > 0 && W > 0, "mask_extent: bad arguments");
  hipStream_t s = (hipStream_t)stream_;
  hipLaunchKernelGGL(mask_extent_init_kernel, dim3((N + 63) / 64), dim3(64), 0, s, extent, N, H, W);
  hipLaunchKernelGGL(mask_extent_kernel, dim3(grid_for((long long)H * W, 256, 32), N), dim3(256), 0, s, mask, H, W, extent);
  CVMI_LAUNCH_CHECK();
  return 0;
}
#endif

#ifndef CVMI_OPERAND_BF16
extern "C" int cvmi_upsample_refine(const float* low, int N, int h, int w, float* high, int H, int W, const float* params, const int* ks, int nk,
                                    int ic, cvmi_stream_t stream_) {
  CVMI_CHECK(low && high && params && ks && N > 0 && h > 0 && w > 0 && H > 0 && W > 0, "upsample_refine: bad arguments");
  CVMI_CHECK(nk >= 1 && nk <= RF_MAXK && ic == 4, "upsample_refine: supports 1..%d branches of 4 channels", RF_MAXK);
  RefineArgs a;
  int off = 0, halo = 0;
  for (int j = 0; j < RF_MAXK; ++j) {
    if (j < nk) {
      CVMI_CHECK(ks[j] % 2 == 1 && ks[j] >= 1 && ks[j] <= 15, "upsample_refine: kernel sizes must be odd and <= 15");
      a.ks[j] = ks[j]; a.woff[j] = off; off += ic * ks[j] * ks[j]; a.boff[j] = off; off += ic;
      if (ks[j] / 2 > halo) halo = ks[j] / 2;
    } else { a.ks[j] = 1; a.woff[j] = a.boff[j] = 0; }
  }
  a.nk = nk; a.ic = ic; a.comb_w = off; a.comb_b = off + nk * ic; a.halo = halo;
  if (nk == 4 && ks[0] == 3 && ks[1] == 5 && ks[2] == 7 && ks[3] == 11 && ic == 4 && ((uintptr_t)high & 15) == 0) {   // the reference's head
    const dim3 gf((W + RF_TW - 1) / RF_TW, (H + RF_TH - 1) / RF_TH, N);
    hipLaunchKernelGGL(upsample_refine_fast_kernel, gf, dim3(256), 0, (hipStream_t)stream_, low, h, w, high, H, W, params, (float)h / (float)H,
                       (float)w / (float)W);
    CVMI_LAUNCH_CHECK();
    return 0;
  }
  const int TS = RF_TILE + 2 * halo;
  const size_t lds = (size_t)TS * (TS + 1) * sizeof(float);
  const dim3 g((W + RF_TILE - 1) / RF_TILE, (H + RF_TILE - 1) / RF_TILE, N);
  hipLaunchKernelGGL(upsample_refine_kernel<4>, g, dim3(256), lds, (hipStream_t)stream_, low, h, w, high, H, W, params, a, (float)h / (float)H,
                     (float)w / (float)W);
  CVMI_LAUNCH_CHECK();
  return 0;
}
#endif

#ifndef CVMI_OPERAND_BF16
extern "C" int cvmi_sam2_transform_batch_bf16(const uint8_t* src, int B, int H, int W, void* dst, int R, int dst_dtype, int swap_rb, cvmi_stream_t stream_);
#endif
extern "C" int CVMI_ENTRY(cvmi_sam2_transform_batch)(const uint8_t* src, int B, int H, int W, void* dst, int R, int dst_dtype, int swap_rb,
                                                     cvmi_stream_t stream_) {
#ifndef CVMI_OPERAND_BF16
  if (dst_dtype == CVMI_BF16) return cvmi_sam2_transform_batch_bf16(src, B, H, W, dst, R, dst_dtype, swap_rb, stream_);
#endif
  CVMI_CHECK(src && dst && B >= 1 && B <= 65535 && H > 0 && W > 0 && R > 0, "sam2_transform: bad arguments");
  CVMI_CHECK(dst_dtype == CVMI_T16 || dst_dtype == CVMI_F32, "sam2_transform: bad dtype");
  const float sy = (float)H / (float)R, sx = (float)W / (float)R;
  CVMI_CHECK(2.f * (sy > 1.f ? sy : 1.f) + 2.f <= AA_MAXTAPS && 2.f * (sx > 1.f ? sx : 1.f) + 2.f <= AA_MAXTAPS, "sam2_transform: down-scale factor too large");
  hipStream_t s = (hipStream_t)stream_;
  const dim3 g(grid_for((long long)R * R), B), b(256);
  if (dst_dtype == CVMI_T16) hipLaunchKernelGGL(sam2_transform_kernel<f16>, g, b, 0, s, src, H, W, (f16*)dst, R, sy, sx, swap_rb ? 1 : 0);
  else hipLaunchKernelGGL(sam2_transform_kernel<float>, g, b, 0, s, src, H, W, (float*)dst, R, sy, sx, swap_rb ? 1 : 0);
  CVMI_LAUNCH_CHECK();
  return 0;
}
#ifndef CVMI_OPERAND_BF16
extern "C" int cvmi_sam2_transform_rects_bf16(const uint8_t* src, long long src_image_stride, int H, int W, const int* rects, int B, void* dst, int R,
                                              int dst_dtype, int swap_rb, cvmi_stream_t stream_);
#endif
extern "C" int CVMI_ENTRY(cvmi_sam2_transform_rects)(const uint8_t* src, long long src_image_stride, int H, int W, const int* rects, int B, void* dst, int R,
                                                     int dst_dtype, int swap_rb, cvmi_stream_t stream_) {
#ifndef CVMI_OPERAND_BF16
  if (dst_dtype == CVMI_BF16) return cvmi_sam2_transform_rects_bf16(src, src_image_stride, H, W, rects, B, dst, R, dst_dtype, swap_rb, stream_);
#endif
  CVMI_CHECK(src && dst && rects && B >= 1 && H > 0 && W > 0 && R > 0 && src_image_stride >= 0, "sam2_transform_rects: bad arguments");
  CVMI_CHECK(dst_dtype == CVMI_T16 || dst_dtype == CVMI_F32, "sam2_transform_rects: bad dtype");
  hipStream_t s = (hipStream_t)stream_;
  for (int b0 = 0; b0 < B; b0 += RECT_MAX) {
    const int nb = B - b0 < RECT_MAX ? B - b0 : RECT_MAX;
    RectTable t;
    for (int b = 0; b < nb; ++b) {
      const int* r = rects + 4 * (size_t)(b0 + b);
      CVMI_CHECK(r[0] >= 0 && r[1] >= 0 && r[2] > 0 && r[3] > 0 && (long long)r[0] + r[2] <= W && (long long)r[1] + r[3] <= H,
                 "sam2_transform_rects: window %d = (x %d, y %d, w %d, h %d) leaves the %d x %d image", b0 + b, r[0], r[1], r[2], r[3], W, H);
      const float sy = (float)r[3] / (float)R, sx = (float)r[2] / (float)R;
      CVMI_CHECK(2.f * (sy > 1.f ? sy : 1.f) + 2.f <= AA_MAXTAPS && 2.f * (sx > 1.f ? sx : 1.f) + 2.f <= AA_MAXTAPS,
                 "sam2_transform_rects: down-scale factor of window %d too large", b0 + b);
      t.r[b] = make_int4(r[0], r[1], r[2], r[3]);
    }
    for (int b = nb; b < RECT_MAX; ++b) t.r[b] = make_int4(0, 0, 1, 1);
    const dim3 g(grid_for((long long)R * R), nb), blk(256);
    const uint8_t* sp = src + (size_t)b0 * src_image_stride;
    const size_t dofs = (size_t)b0 * R * R * 3;
    if (dst_dtype == CVMI_T16) hipLaunchKernelGGL(sam2_transform_rects_kernel<f16>, g, blk, 0, s, sp, src_image_stride, W, t, (f16*)dst + dofs, R, swap_rb ? 1 : 0);
    else hipLaunchKernelGGL(sam2_transform_rects_kernel<float>, g, blk, 0, s, sp, src_image_stride, W, t, (float*)dst + dofs, R, swap_rb ? 1 : 0);
    CVMI_LAUNCH_CHECK();
  }
  return 0;
}

#ifndef CVMI_OPERAND_BF16
extern "C" int cvmi_mask_postprocess_sizes(const float* x, int N, int h, int w, const int* sizes, float thresh, uint8_t* mask_u8, int* extent,
                                           cvmi_stream_t stream_) {
  CVMI_CHECK(x && mask_u8 && extent && sizes && N > 0 && h > 0 && w > 0, "mask_postprocess_sizes: bad arguments");
  hipStream_t s = (hipStream_t)stream_;
  long long off = 0;
  for (int n0 = 0; n0 < N; n0 += RECT_MAX) {
    const int nb = N - n0 < RECT_MAX ? N - n0 : RECT_MAX;
    RectTable t;
    long long maxhw = 1, rel = 0;
    for (int n = 0; n < nb; ++n) {
      const int H = sizes[2 * (size_t)(n0 + n)], W = sizes[2 * (size_t)(n0 + n) + 1];
      CVMI_CHECK(H > 0 && W > 0 && (long long)H * W < (1ll << 31), "mask_postprocess_sizes: plane %d has size %d x %d", n0 + n, H, W);
      t.r[n] = make_int4(H, W, (int)(unsigned)(rel & 0xffffffffll), (int)(unsigned)(rel >> 32));
      rel += (long long)H * W;
      if ((long long)H * W > maxhw) maxhw = (long long)H * W;
    }
    for (int n = nb; n < RECT_MAX; ++n) t.r[n] = make_int4(1, 1, 0, 0);
    hipLaunchKernelGGL(mask_extent_init_sizes_kernel, dim3(1), dim3(64), 0, s, extent + 4 * (size_t)n0, nb, t);
    hipLaunchKernelGGL(bilinear_sizes_kernel, dim3(grid_for(maxhw, 256, 1024), nb), dim3(256), 0, s, x + (size_t)n0 * h * w, h, w, t, mask_u8, off, thresh,
                       extent + 4 * (size_t)n0);
    CVMI_LAUNCH_CHECK();
    off += rel;
  }
  return 0;
}
#endif

#ifndef CVMI_OPERAND_BF16
extern "C" int cvmi_sam2_transform(const uint8_t* src, int H, int W, void* dst, int R, int dst_dtype, cvmi_stream_t stream_) {
  return cvmi_sam2_transform_batch(src, 1, H, W, dst, R, dst_dtype, 0, stream_);
}
#endif
