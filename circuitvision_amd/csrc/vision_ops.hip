// HBM-bound helper kernels of the detector path: depthwise 3x3 conv, SPPF pooling chain, Detect
// decode, letterbox, layout conversion.  All NHWC, 16-byte (8 x fp16 / 4 x f32) accesses per lane.
#include "common.hpp"

namespace {

// ------------------------------------------------------------------------------------------------
// depthwise 3x3: a thread owns one 16-byte channel chunk and a strip of 4 output pixels along x: the 3 x 6 input
// columns of the strip are loaded once (18 loads for 4 outputs instead of 36) and the 9 tap weights of the chunk stay
// in registers for the whole strip.
template <typename T>
__global__ __launch_bounds__(256) void dwconv3x3_strip_kernel(const char* __restrict__ x, int x_ld, const char* __restrict__ w,
                                                             const float* __restrict__ bias, const char* __restrict__ res, int res_ld,
                                                             char* __restrict__ y, int y_ld, int B, int H, int W, int C, int act) {
  constexpr int VEC = Elem<T>::VEC;
  constexpr bool FAST = FastMath<T>::value;
  const int nch = C / VEC;
  const int strips = (W + 3) / 4;
  const long long total = (long long)B * H * strips * nch;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int ch = (int)(idx % nch);
  long long t = idx / nch;
  const int sx = (int)(t % strips); t /= strips;
  const int py = (int)(t % H);
  const int b = (int)(t / H);
  const int px0 = sx * 4;
  float wt[9][VEC], bv[VEC];
#pragma unroll
  for (int k = 0; k < 9; ++k) unpack16<T>(*reinterpret_cast<const u32x4*>(w + ((size_t)k * C + ch * VEC) * sizeof(T)), wt[k]);
#pragma unroll
  for (int e = 0; e < VEC; ++e) bv[e] = bias[ch * VEC + e];
  float acc[4][VEC];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[q][e] = bv[e];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = py + ky - 1;
    const bool yok = (unsigned)iy < (unsigned)H;
    const int iyc = yok ? iy : 0;
#pragma unroll
    for (int cx = 0; cx < 6; ++cx) {                        // input columns px0-1 .. px0+4
      const int ix = px0 + cx - 1;
      const bool ok = yok && (unsigned)ix < (unsigned)W;
      const int ixc = ok ? ix : 0;
      const u32x4 raw = *reinterpret_cast<const u32x4*>(x + ((((size_t)b * H + iyc) * W + ixc) * x_ld + ch * VEC) * sizeof(T));
      float xf[VEC];
      unpack16<T>(raw, xf);
#pragma unroll
      for (int e = 0; e < VEC; ++e) xf[e] = ok ? xf[e] : 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int kx = cx - q;                              // tap column of output q that sees input column cx
        if (kx >= 0 && kx < 3) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc[q][e] = fmaf(xf[e], wt[ky * 3 + kx][e], acc[q][e]);
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int px = px0 + q;
    if (px < W) {
      const size_t pix = ((size_t)b * H + py) * W + px;
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[q][e] = act_apply<FAST>(acc[q][e], act);
      if (res) {
        float rf[VEC];
        unpack16<T>(*reinterpret_cast<const u32x4*>(res + (pix * res_ld + ch * VEC) * sizeof(T)), rf);
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[q][e] += rf[e];
      }
      *reinterpret_cast<u32x4*>(y + (pix * y_ld + ch * VEC) * sizeof(T)) = pack16<T>(acc[q]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// SPPF: three chained 5x5/s1/p2 max-pools == clipped 5x5, 9x9, 13x13 maxima of y0.
template <typename T>
__global__ __launch_bounds__(256) void sppf_pool_kernel(char* __restrict__ buf, int ld, int B, int H, int W, int C) {
  constexpr int VEC = Elem<T>::VEC;
  const int nch = C / VEC;
  const long long total = (long long)B * H * W * nch;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int ch = (int)(idx % nch);
    const long long pix = idx / nch;
    const int px = (int)(pix % W);
    const long long t = pix / W;
    const int py = (int)(t % H);
    const int b = (int)(t / H);
    float m1[VEC], m2[VEC], m3[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) m1[e] = m2[e] = m3[e] = -INFINITY;
    for (int dy = -6; dy <= 6; ++dy) {
      const int iy = py + dy;
      if ((unsigned)iy >= (unsigned)H) continue;
      const int ady = dy < 0 ? -dy : dy;
      for (int dx = -6; dx <= 6; ++dx) {
        const int ix = px + dx;
        if ((unsigned)ix >= (unsigned)W) continue;
        const int adx = dx < 0 ? -dx : dx;
        const int r = ady > adx ? ady : adx;
        float v[VEC];
        unpack16<T>(*reinterpret_cast<const u32x4*>(buf + ((((size_t)b * H + iy) * W + ix) * ld + ch * VEC) * sizeof(T)), v);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          m3[e] = fmaxf(m3[e], v[e]);
          if (r <= 4) m2[e] = fmaxf(m2[e], v[e]);
          if (r <= 2) m1[e] = fmaxf(m1[e], v[e]);
        }
      }
    }
    char* o = buf + ((size_t)pix * ld + ch * VEC) * sizeof(T);
    *reinterpret_cast<u32x4*>(o + (size_t)C * sizeof(T)) = pack16<T>(m1);
    *reinterpret_cast<u32x4*>(o + (size_t)2 * C * sizeof(T)) = pack16<T>(m2);
    *reinterpret_cast<u32x4*>(o + (size_t)3 * C * sizeof(T)) = pack16<T>(m3);
  }
}


// SPPF v2: one workgroup per (image, 16-byte channel chunk); the HxW plane of that chunk lives in LDS as f32 and
// the three chained 5x5 max-pools run as separable 5-tap passes (row then column): ~30 LDS reads per pixel per
// stage instead of 169 global reads per pixel.
template <typename T>
__global__ __launch_bounds__(256) void sppf_pool_lds_kernel(char* __restrict__ buf, int ld, int H, int W, int C) {
  constexpr int VEC = Elem<T>::VEC;
  extern __shared__ __attribute__((aligned(16))) float pl[];      // planes A, R: [H*W][VEC] each
  const int nch = C / VEC;
  const int b = blockIdx.x / nch, ch = blockIdx.x % nch;
  const int HW = H * W;
  float* A = pl;
  float* R = pl + (size_t)HW * VEC;
  char* base = buf + ((size_t)b * HW * ld + ch * VEC) * sizeof(T);
  for (int p = threadIdx.x; p < HW; p += 256) {
    float v[VEC];
    unpack16<T>(*reinterpret_cast<const u32x4*>(base + (size_t)p * ld * sizeof(T)), v);
#pragma unroll
    for (int e = 0; e < VEC; ++e) A[p * VEC + e] = v[e];
  }
  __syncthreads();
  for (int stage = 1; stage <= 3; ++stage) {
    for (int p = threadIdx.x; p < HW; p += 256) {             // row pass: A -> R
      const int y = p / W, x = p - y * W;
      float m[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) m[e] = -INFINITY;
      for (int dx = -2; dx <= 2; ++dx) {
        const int xx = x + dx;
        if ((unsigned)xx < (unsigned)W) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) m[e] = fmaxf(m[e], A[(y * W + xx) * VEC + e]);
        }
      }
#pragma unroll
      for (int e = 0; e < VEC; ++e) R[p * VEC + e] = m[e];
    }
    __syncthreads();
    for (int p = threadIdx.x; p < HW; p += 256) {             // column pass: R -> A (next stage's input) + store
      const int y = p / W, x = p - y * W;
      float m[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) m[e] = -INFINITY;
      for (int dy = -2; dy <= 2; ++dy) {
        const int yy = y + dy;
        if ((unsigned)yy < (unsigned)H) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) m[e] = fmaxf(m[e], R[(yy * W + x) * VEC + e]);
        }
      }
      *reinterpret_cast<u32x4*>(base + ((size_t)p * ld + (size_t)stage * C) * sizeof(T)) = pack16<T>(m);
#pragma unroll
      for (int e = 0; e < VEC; ++e) A[p * VEC + e] = m[e];    // A was fully consumed by the row pass (barrier above)
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
struct DetectArgs {
  const char* box[3]; const char* cls[3];
  int box_ld[3], cls_ld[3], hs[3], ws[3], a0[3];
  float strides[3];
  int nlevels, B, nc, A;
};

// Decode v2: one workgroup = 64 consecutive anchors of the flat (image, anchor) space.
//  * box: 4 lanes per anchor (one per side) read 32 contiguous bytes each -> coalesced; softmax-expectation per
//    lane, the 4 distances are exchanged with 2 shuffles
//  * cls: 16-byte chunks read contiguously, sigmoid, transposed through LDS so that the [B,4+nc,A] writes are
//    64 consecutive anchors per class row; the per-anchor best class (first maximum) falls out of the same tile
//    and feeds the NMS kernel directly (best_score / best_cls), so NMS never re-reads the class rows.
template <typename T>
__global__ __launch_bounds__(256) void detect_decode_kernel(const DetectArgs d, float* __restrict__ pred, float* __restrict__ best_score,
                                                           int* __restrict__ best_cls, int write_cls) {
  constexpr bool FAST = FastMath<T>::value;
  constexpr int VEC = Elem<T>::VEC;
  extern __shared__ __attribute__((aligned(16))) float tile[];          // [64][ncp + 1]
  const int ncp = (d.nc + VEC - 1) / VEC * VEC;
  const int TS = ncp + 1;
  const long long total = (long long)d.B * d.A;
  const long long g0 = (long long)blockIdx.x * 64;
  const int tid = threadIdx.x;
  // ---- boxes: thread = (anchor tid/4, side tid%4)
  {
    const long long g = g0 + (tid >> 2);
    const int side = tid & 3;
    float dist = 0.f;
    int a = 0, b = 0, l = 0, ax = 0, ay = 0;
    const bool ok = g < total;
    if (ok) {
      a = (int)(g % d.A); b = (int)(g / d.A);
      if (d.nlevels > 1 && a >= d.a0[1]) l = 1;
      if (d.nlevels > 2 && a >= d.a0[2]) l = 2;
      const int al = a - d.a0[l];
      ay = al / d.ws[l]; ax = al - ay * d.ws[l];
      const T* bp = reinterpret_cast<const T*>(d.box[l]) + ((size_t)b * d.hs[l] * d.ws[l] + al) * d.box_ld[l] + side * 16;
      float v[16];
      if constexpr (VEC == 8) {
        unpack16<T>(*reinterpret_cast<const u32x4*>(bp), v);
        unpack16<T>(*reinterpret_cast<const u32x4*>(bp + 8), v + 8);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) unpack16<T>(*reinterpret_cast<const u32x4*>(bp + 4 * i), v + 4 * i);
      }
      float mx = v[0];
#pragma unroll
      for (int i = 1; i < 16; ++i) mx = fmaxf(mx, v[i]);
      float sum = 0.f, wsum = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float e = FAST ? __expf(v[i] - mx) : expf(v[i] - mx);
        sum += e;
        wsum += e * (float)i;
      }
      dist = wsum / sum;
    }
    const int lane4 = (threadIdx.x & 63) & ~3;
    float dd[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) dd[j] = __shfl(dist, lane4 + j);       // the 4 sides of this anchor
    if (ok) {
      const float cxa = (float)ax + 0.5f, cya = (float)ay + 0.5f;
      const float x1 = cxa - dd[0], y1 = cya - dd[1], x2 = cxa + dd[2], y2 = cya + dd[3];
      const float st = d.strides[l];
      float o;
      if (side == 0) o = (x1 + x2) / 2.f * st;
      else if (side == 1) o = (y1 + y2) / 2.f * st;
      else if (side == 2) o = (x2 - x1) * st;
      else o = (y2 - y1) * st;
      pred[(size_t)b * (4 + d.nc) * d.A + (size_t)side * d.A + a] = o;
    }
  }
  // ---- classes: chunk j = tid + 256*i -> (anchor j / nchunk, chunk j % nchunk)
  const int nchunk = ncp / VEC;
  for (int j = tid; j < 64 * nchunk; j += 256) {
    const int al64 = j / nchunk, c0 = (j - al64 * nchunk) * VEC;
    const long long g = g0 + al64;
    float v[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] = 0.f;
    if (g < total) {
      const int a = (int)(g % d.A), b = (int)(g / d.A);
      int l = 0;
      if (d.nlevels > 1 && a >= d.a0[1]) l = 1;
      if (d.nlevels > 2 && a >= d.a0[2]) l = 2;
      const int al = a - d.a0[l];
      const T* cp = reinterpret_cast<const T*>(d.cls[l]) + ((size_t)b * d.hs[l] * d.ws[l] + al) * d.cls_ld[l] + c0;
      unpack16<T>(*reinterpret_cast<const u32x4*>(cp), v);
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) tile[al64 * TS + c0 + e] = act_apply<FAST>(v[e], CVMI_ACT_SIGMOID);
  }
  __syncthreads();
  if (tid < 64 && best_score) {
    const long long g = g0 + tid;
    if (g < total) {
      float best = tile[tid * TS];
      int bc = 0;
      for (int c = 1; c < d.nc; ++c) {
        const float v = tile[tid * TS + c];
        if (v > best) { best = v; bc = c; }
      }
      best_score[g] = best;
      best_cls[g] = bc;
    }
  }
  if (write_cls) {
    for (int j = tid; j < 64 * d.nc; j += 256) {
      const int c = j >> 6, al64 = j & 63;
      const long long g = g0 + al64;
      if (g < total) {
        const int a = (int)(g % d.A), b = (int)(g / d.A);
        pred[(size_t)b * (4 + d.nc) * d.A + (size_t)(4 + c) * d.A + a] = tile[al64 * TS + c];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Letterbox: OpenCV-style 8-bit fixed-point bilinear (11-bit coefficients) + pad 114 + channel
// flip + /255, written NHWC with 3 channels.
__device__ __forceinline__ void lb_axis(int d, int dst, int src, int& s0, int& s1, int& a0, int& a1) {
  const double scale = (double)src / (double)dst;
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) { f = 0.f; s = 0; }
  if (s >= src - 1) { f = 0.f; s = src - 1; }
  a1 = (int)rintf(f * 2048.f);
  a0 = (int)rintf((1.f - f) * 2048.f);
  s0 = s;
  s1 = s + 1 < src ? s + 1 : src - 1;
}

// s2d = 1: the output is written space-to-depth(2) with 16 channels per 2x2 block,
// dst[(y/2, x/2)][((y&1)*2 + (x&1))*3 + c] (channels 12..15 zero), so that the stride-2 3x3 stem conv becomes a
// stride-1 2x2 conv with 16-byte-aligned channel vectors (see Yolo11Weights.stem).
template <typename T>
__global__ __launch_bounds__(256) void letterbox_kernel(const uint8_t* __restrict__ src, int H, int W, T* __restrict__ dst, int out_h,
                                                       int out_w, int new_h, int new_w, int top, int left, int s2d, long long dst_image_stride) {
  src += (size_t)blockIdx.y * H * W * 3;                  // blockIdx.y = image of a batch of equally sized sources
  dst += (size_t)blockIdx.y * dst_image_stride;
  const int total = out_h * out_w;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int ox = idx % out_w, oy = idx / out_w;
    const int rx = ox - left, ry = oy - top;
    int v[3] = {114, 114, 114};
    if (rx >= 0 && rx < new_w && ry >= 0 && ry < new_h) {
      if (new_w == W && new_h == H) {
        const uint8_t* s = src + ((size_t)ry * W + rx) * 3;
        v[0] = s[0]; v[1] = s[1]; v[2] = s[2];
      } else {
        int x0, x1, ax0, ax1, y0, y1, ay0, ay1;
        lb_axis(rx, new_w, W, x0, x1, ax0, ax1);
        lb_axis(ry, new_h, H, y0, y1, ay0, ay1);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int r0 = src[((size_t)y0 * W + x0) * 3 + c] * ax0 + src[((size_t)y0 * W + x1) * 3 + c] * ax1;
          const int r1 = src[((size_t)y1 * W + x0) * 3 + c] * ax0 + src[((size_t)y1 * W + x1) * 3 + c] * ax1;
          int o = (((ay0 * (r0 >> 4)) >> 16) + ((ay1 * (r1 >> 4)) >> 16) + 2) >> 2;
          v[c] = o < 0 ? 0 : (o > 255 ? 255 : o);
        }
      }
    }
    T* o = dst + (size_t)idx * 3;
    if (s2d) {
      o = dst + ((size_t)(oy >> 1) * (out_w >> 1) + (ox >> 1)) * 16 + ((oy & 1) * 2 + (ox & 1)) * 3;
      if ((oy & 1) && (ox & 1)) { o[3] = (T)0.f; o[4] = (T)0.f; o[5] = (T)0.f; o[6] = (T)0.f; }     // channels 12..15
    }
    o[0] = (T)((float)v[2] / 255.0f);   // reversed channel order (ultralytics im[..., ::-1])
    o[1] = (T)((float)v[1] / 255.0f);
    o[2] = (T)((float)v[0] / 255.0f);
  }
}

// ------------------------------------------------------------------------------------------------
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int dst_ld, int B, int C, int H, int W) {
  const long long total = (long long)B * C * H * W;
  const long long hw = (long long)H * W;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    const long long pix = idx / C;            // b*H*W + y*W + x
    const long long b = pix / hw, p = pix - b * hw;
    dst[pix * dst_ld + c] = (TD)(float)src[(b * C + c) * hw + p];
  }
}
template <typename TS>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const TS* __restrict__ src, int src_ld, float* __restrict__ dst, int B, int C, int H, int W) {
  const long long total = (long long)B * C * H * W;
  const long long hw = (long long)H * W;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const long long p = idx % hw;
    const long long bc = idx / hw;
    const int c = (int)(bc % C);
    const long long b = bc / C;
    dst[idx] = (float)src[(b * hw + p) * src_ld + c];
  }
}

inline int grid_for(long long total, int block = 256) {
  long long g = (total + block - 1) / block;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

#ifndef CVMI_OPERAND_BF16
extern "C" int cvmi_dwconv3x3(const void* x, int x_ld, const void* w, const float* bias, const void* res, int res_ld, void* y, int y_ld,
                              int B, int H, int W, int C, int act, int dtype, cvmi_stream_t stream_) {
  CVMI_CHECK(x && w && bias && y, "dwconv3x3: null pointer");
  CVMI_CHECK(dtype == CVMI_F16 || dtype == CVMI_F32, "dwconv3x3: bad dtype");
  const int vec = dtype == CVMI_F16 ? 8 : 4;
  CVMI_CHECK(B > 0 && H > 0 && W > 0 && C > 0 && C % vec == 0, "dwconv3x3: C=%d must be a multiple of %d", C, vec);
  CVMI_CHECK(x_ld % vec == 0 && y_ld % vec == 0 && (!res || res_ld % vec == 0), "dwconv3x3: ld not 16-byte aligned");
  CVMI_CHECK((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)res) & 15) == 0, "dwconv3x3: pointer not 16-byte aligned");
  hipStream_t stream = (hipStream_t)stream_;
  const long long total = (long long)B * H * ((W + 3) / 4) * (C / vec);
  CVMI_CHECK(total < (1ll << 31) * 256, "dwconv3x3: grid too large");
  const unsigned blocks = (unsigned)((total + 255) / 256);
  if (dtype == CVMI_F16)
    hipLaunchKernelGGL(dwconv3x3_strip_kernel<f16>, dim3(blocks), dim3(256), 0, stream, (const char*)x, x_ld, (const char*)w, bias,
                       (const char*)res, res_ld, (char*)y, y_ld, B, H, W, C, act);
  else
    hipLaunchKernelGGL(dwconv3x3_strip_kernel<float>, dim3(blocks), dim3(256), 0, stream, (const char*)x, x_ld, (const char*)w, bias,
                       (const char*)res, res_ld, (char*)y, y_ld, B, H, W, C, act);
  CVMI_LAUNCH_CHECK();
  return 0;
}
#endif

#ifndef CVMI_OPERAND_BF16
extern "C" int cvmi_sppf_pool(void* buf, int ld, int B, int H, int W, int C, int dtype, cvmi_stream_t stream_) {
  CVMI_CHECK(buf, "sppf_pool: null pointer");
  CVMI_CHECK(dtype == CVMI_F16 || dtype == CVMI_F32, "sppf_pool: bad dtype");
  const int vec = dtype == CVMI_F16 ? 8 : 4;
  CVMI_CHECK(B > 0 && H > 0 && W > 0 && C > 0 && C % vec == 0 && ld >= 4 * C && ld % vec == 0, "sppf_pool: bad shape C=%d ld=%d", C, ld);
  CVMI_CHECK(((uintptr_t)buf & 15) == 0, "sppf_pool: pointer not 16-byte aligned");
  hipStream_t stream = (hipStream_t)stream_;
  const long long total = (long long)B * H * W * (C / vec);
  const size_t lds = (size_t)2 * H * W * vec * sizeof(float);
  if (lds <= 64 * 1024) {
    const unsigned blocks = (unsigned)(B * (C / vec));
    if (dtype == CVMI_F16) hipLaunchKernelGGL(sppf_pool_lds_kernel<f16>, dim3(blocks), dim3(256), lds, stream, (char*)buf, ld, H, W, C);
    else hipLaunchKernelGGL(sppf_pool_lds_kernel<float>, dim3(blocks), dim3(256), lds, stream, (char*)buf, ld, H, W, C);
  } else if (dtype == CVMI_F16)
    hipLaunchKernelGGL(sppf_pool_kernel<f16>, dim3(grid_for(total)), dim3(256), 0, stream, (char*)buf, ld, B, H, W, C);
  else
    hipLaunchKernelGGL(sppf_pool_kernel<float>, dim3(grid_for(total)), dim3(256), 0, stream, (char*)buf, ld, B, H, W, C);
  CVMI_LAUNCH_CHECK();
  return 0;
}
#endif

#ifndef CVMI_OPERAND_BF16
extern "C" int cvmi_detect_decode(const void* const* box, const int* box_ld, const void* const* cls, const int* cls_ld, const int* hs,
                                  const int* ws, const float* strides, int nlevels, int B, int nc, int dtype, float* pred,
                                  float* best_score, int* best_cls, int write_cls, cvmi_stream_t stream_) {
  CVMI_CHECK(box && cls && box_ld && cls_ld && hs && ws && strides && pred, "detect_decode: null pointer");
  CVMI_CHECK(nlevels >= 1 && nlevels <= 3 && B > 0 && nc > 0 && nc <= 1024, "detect_decode: bad shape");
  CVMI_CHECK(dtype == CVMI_F16 || dtype == CVMI_F32, "detect_decode: bad dtype");
  CVMI_CHECK((best_score == nullptr) == (best_cls == nullptr), "detect_decode: best_score / best_cls go together");
  const int vec = dtype == CVMI_F16 ? 8 : 4;
  const int ncp = (nc + vec - 1) / vec * vec;
  DetectArgs d;
  int A = 0;
  for (int l = 0; l < 3; ++l) {
    if (l < nlevels) {
      CVMI_CHECK(box[l] && cls[l] && hs[l] > 0 && ws[l] > 0 && box_ld[l] >= 64 && cls_ld[l] >= ncp, "detect_decode: level %d: ld must cover 64 box / %d class channels", l, ncp);
      CVMI_CHECK(box_ld[l] % vec == 0 && cls_ld[l] % vec == 0 && (((uintptr_t)box[l] | (uintptr_t)cls[l]) & 15) == 0, "detect_decode: level %d not 16-byte aligned", l);
      d.box[l] = (const char*)box[l]; d.cls[l] = (const char*)cls[l];
      d.box_ld[l] = box_ld[l]; d.cls_ld[l] = cls_ld[l]; d.hs[l] = hs[l]; d.ws[l] = ws[l]; d.strides[l] = strides[l];
      d.a0[l] = A;
      A += hs[l] * ws[l];
    } else {
      d.box[l] = d.cls[l] = nullptr; d.box_ld[l] = d.cls_ld[l] = d.hs[l] = d.ws[l] = 0; d.a0[l] = 1 << 30; d.strides[l] = 0.f;
    }
  }
  d.nlevels = nlevels; d.B = B; d.nc = nc; d.A = A;
  hipStream_t stream = (hipStream_t)stream_;
  const long long total = (long long)B * A;
  const unsigned blocks = (unsigned)((total + 63) / 64);
  const size_t lds = (size_t)64 * (ncp + 1) * sizeof(float);
  CVMI_CHECK(lds <= 64 * 1024, "detect_decode: nc too large for the LDS tile");
  if (dtype == CVMI_F16)
    hipLaunchKernelGGL(detect_decode_kernel<f16>, dim3(blocks), dim3(256), lds, stream, d, pred, best_score, best_cls, write_cls);
  else
    hipLaunchKernelGGL(detect_decode_kernel<float>, dim3(blocks), dim3(256), lds, stream, d, pred, best_score, best_cls, write_cls);
  CVMI_LAUNCH_CHECK();
  return 0;
}
#endif

#ifndef CVMI_OPERAND_BF16
extern "C" int cvmi_letterbox(const uint8_t* src, int H, int W, void* dst, int out_h, int out_w, int new_h, int new_w, int top, int left,
                              int dtype, int s2d, cvmi_stream_t stream_) {
  return cvmi_letterbox_batch(src, 1, H, W, dst, 0, out_h, out_w, new_h, new_w, top, left, dtype, s2d, stream_);
}

extern "C" int cvmi_letterbox_batch(const uint8_t* src, int B, int H, int W, void* dst, long long dst_image_stride, int out_h, int out_w, int new_h,
                                    int new_w, int top, int left, int dtype, int s2d, cvmi_stream_t stream_) {
  CVMI_CHECK(!s2d || (out_h % 2 == 0 && out_w % 2 == 0), "letterbox: space-to-depth output needs even out_h, out_w");
  CVMI_CHECK(src && dst, "letterbox: null pointer");
  CVMI_CHECK(B >= 1 && B <= 65535 && (B == 1 || dst_image_stride >= (long long)out_h * out_w * (s2d ? 4 : 3)), "letterbox: bad batch / image stride");
  CVMI_CHECK(H > 0 && W > 0 && out_h > 0 && out_w > 0 && new_h > 0 && new_w > 0 && top >= 0 && left >= 0 && top + new_h <= out_h &&
                 left + new_w <= out_w, "letterbox: bad geometry");
  CVMI_CHECK(dtype == CVMI_F16 || dtype == CVMI_F32, "letterbox: bad dtype");
  hipStream_t stream = (hipStream_t)stream_;
  const long long total = (long long)out_h * out_w;
  if (dtype == CVMI_F16)
    hipLaunchKernelGGL(letterbox_kernel<f16>, dim3(grid_for(total), B), dim3(256), 0, stream, src, H, W, (f16*)dst, out_h, out_w, new_h, new_w, top, left, s2d, dst_image_stride);
  else
    hipLaunchKernelGGL(letterbox_kernel<float>, dim3(grid_for(total), B), dim3(256), 0, stream, src, H, W, (float*)dst, out_h, out_w, new_h, new_w, top, left, s2d, dst_image_stride);
  CVMI_LAUNCH_CHECK();
  return 0;
}
#endif

#ifndef CVMI_OPERAND_BF16
#ifndef CVMI_OPERAND_BF16
extern "C" int cvmi_nchw_to_nhwc_bf16(const void* src, int src_dtype, void* dst, int dst_dtype, int dst_ld, int B, int C, int H, int W,
                                      cvmi_stream_t stream_);
#endif
#endif

extern "C" int CVMI_ENTRY(cvmi_nchw_to_nhwc)(const void* src, int src_dtype, void* dst, int dst_dtype, int dst_ld, int B, int C, int H, int W,
                                 cvmi_stream_t stream_) {
#ifndef CVMI_OPERAND_BF16
  if (src_dtype == CVMI_BF16 || dst_dtype == CVMI_BF16) return cvmi_nchw_to_nhwc_bf16(src, src_dtype, dst, dst_dtype, dst_ld, B, C, H, W, stream_);
#endif
  CVMI_CHECK(src && dst && B > 0 && C > 0 && H > 0 && W > 0 && dst_ld >= C, "nchw_to_nhwc: bad arguments");
  hipStream_t stream = (hipStream_t)stream_;
  const long long total = (long long)B * C * H * W;
  const dim3 g(grid_for(total)), b(256);
  if (src_dtype == CVMI_F32 && dst_dtype == CVMI_T16) hipLaunchKernelGGL((nchw_to_nhwc_kernel<float, f16>), g, b, 0, stream, (const float*)src, (f16*)dst, dst_ld, B, C, H, W);
  else if (src_dtype == CVMI_F32 && dst_dtype == CVMI_F32) hipLaunchKernelGGL((nchw_to_nhwc_kernel<float, float>), g, b, 0, stream, (const float*)src, (float*)dst, dst_ld, B, C, H, W);
  else if (src_dtype == CVMI_T16 && dst_dtype == CVMI_T16) hipLaunchKernelGGL((nchw_to_nhwc_kernel<f16, f16>), g, b, 0, stream, (const f16*)src, (f16*)dst, dst_ld, B, C, H, W);
  else if (src_dtype == CVMI_T16 && dst_dtype == CVMI_F32) hipLaunchKernelGGL((nchw_to_nhwc_kernel<f16, float>), g, b, 0, stream, (const f16*)src, (float*)dst, dst_ld, B, C, H, W);
  else CVMI_FAIL("nchw_to_nhwc: bad dtypes %d -> %d", src_dtype, dst_dtype);
  CVMI_LAUNCH_CHECK();
  return 0;
}

#ifndef CVMI_OPERAND_BF16
extern "C" int cvmi_nhwc_to_nchw_f32(const void* src, int src_dtype, int src_ld, float* dst, int B, int C, int H, int W, cvmi_stream_t stream_) {
  CVMI_CHECK(src && dst && B > 0 && C > 0 && H > 0 && W > 0 && src_ld >= C, "nhwc_to_nchw: bad arguments");
  hipStream_t stream = (hipStream_t)stream_;
  const long long total = (long long)B * C * H * W;
  const dim3 g(grid_for(total)), b(256);
  if (src_dtype == CVMI_F16) hipLaunchKernelGGL(nhwc_to_nchw_kernel<f16>, g, b, 0, stream, (const f16*)src, src_ld, dst, B, C, H, W);
  else if (src_dtype == CVMI_F32) hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, g, b, 0, stream, (const float*)src, src_ld, dst, B, C, H, W);
  else CVMI_FAIL("nhwc_to_nchw: bad dtype %d", src_dtype);
  CVMI_LAUNCH_CHECK();
  return 0;
}
#endif

