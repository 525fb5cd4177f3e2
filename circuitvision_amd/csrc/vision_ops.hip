// HBM-bound helper kernels of the detector path: depthwise 3x3 conv, SPPF pooling chain, Detect
// decode, letterbox, layout conversion.  All NHWC, 16-byte (8 x fp16 / 4 x f32) accesses per lane.
#include "common.hpp"

namespace {

// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const char* __restrict__ x, int x_ld, const char* __restrict__ w,
                                                       const float* __restrict__ bias, const char* __restrict__ res, int res_ld,
                                                       char* __restrict__ y, int y_ld, int B, int H, int W, int C, int act) {
  constexpr int VEC = Elem<T>::VEC;
  constexpr bool FAST = FastMath<T>::value;
  const int nch = C / VEC;
  const long long total = (long long)B * H * W * nch;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int ch = (int)(idx % nch);
    const long long pix = idx / nch;
    const int px = (int)(pix % W);
    const long long t = pix / W;
    const int py = (int)(t % H);
    const int b = (int)(t / H);
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = bias[ch * VEC + e];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = py + ky - 1;
      if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = px + kx - 1;
        if ((unsigned)ix >= (unsigned)W) continue;
        const size_t ip = ((size_t)b * H + iy) * W + ix;
        const u32x4 xv = *reinterpret_cast<const u32x4*>(x + (ip * x_ld + ch * VEC) * sizeof(T));
        const u32x4 wv = *reinterpret_cast<const u32x4*>(w + ((size_t)(ky * 3 + kx) * C + ch * VEC) * sizeof(T));
        float xf[VEC], wf[VEC];
        unpack16<T>(xv, xf);
        unpack16<T>(wv, wf);
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = fmaf(xf[e], wf[e], acc[e]);
      }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = act_apply<FAST>(acc[e], act);
    if (res) {
      float rf[VEC];
      unpack16<T>(*reinterpret_cast<const u32x4*>(res + ((size_t)pix * res_ld + ch * VEC) * sizeof(T)), rf);
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] += rf[e];
    }
    *reinterpret_cast<u32x4*>(y + ((size_t)pix * y_ld + ch * VEC) * sizeof(T)) = pack16<T>(acc);
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

// ------------------------------------------------------------------------------------------------
struct DetectArgs {
  const char* box[3]; const char* cls[3];
  int box_ld[3], cls_ld[3], hs[3], ws[3], a0[3];
  float strides[3];
  int nlevels, B, nc, A;
};

template <typename T>
__global__ __launch_bounds__(256) void detect_decode_kernel(const DetectArgs d, float* __restrict__ pred) {
  constexpr bool FAST = FastMath<T>::value;
  const long long total = (long long)d.B * d.A;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int a = (int)(idx % d.A);
    const int b = (int)(idx / d.A);
    int l = 0;
    if (d.nlevels > 1 && a >= d.a0[1]) l = 1;
    if (d.nlevels > 2 && a >= d.a0[2]) l = 2;
    const int al = a - d.a0[l];
    const int w = d.ws[l], h = d.hs[l];
    const int ay = al / w, ax = al - ay * w;
    const size_t pix = (size_t)b * h * w + al;
    const T* bp = reinterpret_cast<const T*>(d.box[l]) + pix * d.box_ld[l];
    float dist[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float v[16];
      float mx = -INFINITY;
#pragma unroll
      for (int i = 0; i < 16; ++i) { v[i] = (float)bp[s * 16 + i]; mx = fmaxf(mx, v[i]); }
      float sum = 0.f, ws = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float e = FAST ? __expf(v[i] - mx) : expf(v[i] - mx);
        sum += e;
        ws += e * (float)i;
      }
      dist[s] = ws / sum;
    }
    const float cxa = (float)ax + 0.5f, cya = (float)ay + 0.5f;
    const float x1 = cxa - dist[0], y1 = cya - dist[1], x2 = cxa + dist[2], y2 = cya + dist[3];
    const float st = d.strides[l];
    float* o = pred + (size_t)b * (4 + d.nc) * d.A + a;
    o[0] = (x1 + x2) / 2.f * st;
    o[(size_t)d.A] = (y1 + y2) / 2.f * st;
    o[(size_t)2 * d.A] = (x2 - x1) * st;
    o[(size_t)3 * d.A] = (y2 - y1) * st;
    const T* cp = reinterpret_cast<const T*>(d.cls[l]) + pix * d.cls_ld[l];
    for (int c = 0; c < d.nc; ++c) o[(size_t)(4 + c) * d.A] = act_apply<FAST>((float)cp[c], CVMI_ACT_SIGMOID);
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

template <typename T>
__global__ __launch_bounds__(256) void letterbox_kernel(const uint8_t* __restrict__ src, int H, int W, T* __restrict__ dst, int out_h,
                                                       int out_w, int new_h, int new_w, int top, int left) {
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

extern "C" int cvmi_dwconv3x3(const void* x, int x_ld, const void* w, const float* bias, const void* res, int res_ld, void* y, int y_ld,
                              int B, int H, int W, int C, int act, int dtype, cvmi_stream_t stream_) {
  CVMI_CHECK(x && w && bias && y, "dwconv3x3: null pointer");
  CVMI_CHECK(dtype == CVMI_F16 || dtype == CVMI_F32, "dwconv3x3: bad dtype");
  const int vec = dtype == CVMI_F16 ? 8 : 4;
  CVMI_CHECK(B > 0 && H > 0 && W > 0 && C > 0 && C % vec == 0, "dwconv3x3: C=%d must be a multiple of %d", C, vec);
  CVMI_CHECK(x_ld % vec == 0 && y_ld % vec == 0 && (!res || res_ld % vec == 0), "dwconv3x3: ld not 16-byte aligned");
  CVMI_CHECK((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)res) & 15) == 0, "dwconv3x3: pointer not 16-byte aligned");
  hipStream_t stream = (hipStream_t)stream_;
  const long long total = (long long)B * H * W * (C / vec);
  if (dtype == CVMI_F16)
    hipLaunchKernelGGL(dwconv3x3_kernel<f16>, dim3(grid_for(total)), dim3(256), 0, stream, (const char*)x, x_ld, (const char*)w, bias,
                       (const char*)res, res_ld, (char*)y, y_ld, B, H, W, C, act);
  else
    hipLaunchKernelGGL(dwconv3x3_kernel<float>, dim3(grid_for(total)), dim3(256), 0, stream, (const char*)x, x_ld, (const char*)w, bias,
                       (const char*)res, res_ld, (char*)y, y_ld, B, H, W, C, act);
  CVMI_LAUNCH_CHECK();
  return 0;
}

extern "C" int cvmi_sppf_pool(void* buf, int ld, int B, int H, int W, int C, int dtype, cvmi_stream_t stream_) {
  CVMI_CHECK(buf, "sppf_pool: null pointer");
  CVMI_CHECK(dtype == CVMI_F16 || dtype == CVMI_F32, "sppf_pool: bad dtype");
  const int vec = dtype == CVMI_F16 ? 8 : 4;
  CVMI_CHECK(B > 0 && H > 0 && W > 0 && C > 0 && C % vec == 0 && ld >= 4 * C && ld % vec == 0, "sppf_pool: bad shape C=%d ld=%d", C, ld);
  CVMI_CHECK(((uintptr_t)buf & 15) == 0, "sppf_pool: pointer not 16-byte aligned");
  hipStream_t stream = (hipStream_t)stream_;
  const long long total = (long long)B * H * W * (C / vec);
  if (dtype == CVMI_F16)
    hipLaunchKernelGGL(sppf_pool_kernel<f16>, dim3(grid_for(total)), dim3(256), 0, stream, (char*)buf, ld, B, H, W, C);
  else
    hipLaunchKernelGGL(sppf_pool_kernel<float>, dim3(grid_for(total)), dim3(256), 0, stream, (char*)buf, ld, B, H, W, C);
  CVMI_LAUNCH_CHECK();
  return 0;
}

extern "C" int cvmi_detect_decode(const void* const* box, const int* box_ld, const void* const* cls, const int* cls_ld, const int* hs,
                                  const int* ws, const float* strides, int nlevels, int B, int nc, int dtype, float* pred,
                                  cvmi_stream_t stream_) {
  CVMI_CHECK(box && cls && box_ld && cls_ld && hs && ws && strides && pred, "detect_decode: null pointer");
  CVMI_CHECK(nlevels >= 1 && nlevels <= 3 && B > 0 && nc > 0, "detect_decode: bad shape");
  CVMI_CHECK(dtype == CVMI_F16 || dtype == CVMI_F32, "detect_decode: bad dtype");
  DetectArgs d;
  int A = 0;
  for (int l = 0; l < 3; ++l) {
    if (l < nlevels) {
      CVMI_CHECK(box[l] && cls[l] && hs[l] > 0 && ws[l] > 0 && box_ld[l] >= 64 && cls_ld[l] >= nc, "detect_decode: bad level %d", l);
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
  if (dtype == CVMI_F16)
    hipLaunchKernelGGL(detect_decode_kernel<f16>, dim3(grid_for(total)), dim3(256), 0, stream, d, pred);
  else
    hipLaunchKernelGGL(detect_decode_kernel<float>, dim3(grid_for(total)), dim3(256), 0, stream, d, pred);
  CVMI_LAUNCH_CHECK();
  return 0;
}

extern "C" int cvmi_letterbox(const uint8_t* src, int H, int W, void* dst, int out_h, int out_w, int new_h, int new_w, int top, int left,
                              int dtype, cvmi_stream_t stream_) {
  CVMI_CHECK(src && dst, "letterbox: null pointer");
  CVMI_CHECK(H > 0 && W > 0 && out_h > 0 && out_w > 0 && new_h > 0 && new_w > 0 && top >= 0 && left >= 0 && top + new_h <= out_h &&
                 left + new_w <= out_w, "letterbox: bad geometry");
  CVMI_CHECK(dtype == CVMI_F16 || dtype == CVMI_F32, "letterbox: bad dtype");
  hipStream_t stream = (hipStream_t)stream_;
  const long long total = (long long)out_h * out_w;
  if (dtype == CVMI_F16)
    hipLaunchKernelGGL(letterbox_kernel<f16>, dim3(grid_for(total)), dim3(256), 0, stream, src, H, W, (f16*)dst, out_h, out_w, new_h, new_w, top, left);
  else
    hipLaunchKernelGGL(letterbox_kernel<float>, dim3(grid_for(total)), dim3(256), 0, stream, src, H, W, (float*)dst, out_h, out_w, new_h, new_w, top, left);
  CVMI_LAUNCH_CHECK();
  return 0;
}

extern "C" int cvmi_nchw_to_nhwc(const void* src, int src_dtype, void* dst, int dst_dtype, int dst_ld, int B, int C, int H, int W,
                                 cvmi_stream_t stream_) {
  CVMI_CHECK(src && dst && B > 0 && C > 0 && H > 0 && W > 0 && dst_ld >= C, "nchw_to_nhwc: bad arguments");
  hipStream_t stream = (hipStream_t)stream_;
  const long long total = (long long)B * C * H * W;
  const dim3 g(grid_for(total)), b(256);
  if (src_dtype == CVMI_F32 && dst_dtype == CVMI_F16) hipLaunchKernelGGL((nchw_to_nhwc_kernel<float, f16>), g, b, 0, stream, (const float*)src, (f16*)dst, dst_ld, B, C, H, W);
  else if (src_dtype == CVMI_F32 && dst_dtype == CVMI_F32) hipLaunchKernelGGL((nchw_to_nhwc_kernel<float, float>), g, b, 0, stream, (const float*)src, (float*)dst, dst_ld, B, C, H, W);
  else if (src_dtype == CVMI_F16 && dst_dtype == CVMI_F16) hipLaunchKernelGGL((nchw_to_nhwc_kernel<f16, f16>), g, b, 0, stream, (const f16*)src, (f16*)dst, dst_ld, B, C, H, W);
  else if (src_dtype == CVMI_F16 && dst_dtype == CVMI_F32) hipLaunchKernelGGL((nchw_to_nhwc_kernel<f16, float>), g, b, 0, stream, (const f16*)src, (float*)dst, dst_ld, B, C, H, W);
  else CVMI_FAIL("nchw_to_nhwc: bad dtypes %d -> %d", src_dtype, dst_dtype);
  CVMI_LAUNCH_CHECK();
  return 0;
}

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
