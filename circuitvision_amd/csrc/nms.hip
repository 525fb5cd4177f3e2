// Detector NMS with ultralytics / torchvision CPU semantics, one 1024-thread workgroup per image.
// (compiled with -ffp-contract=off: IoU arithmetic must round exactly like the fp32 CPU kernel)
//
//   1. per anchor: best class score (first maximum), candidate iff score > conf_thres;
//      key = (score bits << 32) | ~anchor  -> unique, so the order is deterministic
//   2. bitonic sort of the keys in LDS, descending  (= score desc, anchor index asc on ties)
//   3. boxes -> xyxy (xy -/+ wh/2), offset by cls * max_wh, area; staged in LDS behind the sorted keys when they fit
//      (up to ~4 k candidates; the workspace otherwise) -- the greedy pass re-reads them once per kept box
//   4. greedy pass: next unsuppressed candidate is kept, every later candidate with
//      inter / (area_i + area_j - inter) > iou_thres is marked in an LDS bitmask; stops at max_det.
//      The pass only records the kept sorted positions; the detections are written afterwards, one thread each
//      (a kept box costs one LDS round + one barrier, not a chain of dependent global loads).
#include "common.hpp"
#include <mutex>

namespace {

constexpr int NMS_THREADS = 1024;
constexpr int NMS_LDS_A = 16384;                   // sort keys live in LDS up to this many anchors (imgsz <= 864 square) ...
// byte offset of the GK sort keys in the workspace: behind the candidates + best score / class arrays, rounded up to 16 bytes (u64 keys:
// B * A * 28 + 256 alone is only 4-byte aligned when B * A is odd, e.g. imgsz 928 -> A = 17661 at B = 1)
__host__ __device__ inline size_t nms_keys_offset(size_t nb, size_t A) { return (nb * A * 28 + 256 + 15) & ~(size_t)15; }
constexpr int NMS_MAX_A = 65536;                   // ... and in the workspace above it (checkpoints trained at imgsz 960 / 1024 / 1280)
constexpr int NMS_MAX_DET = 4096;                  // LDS list of kept positions

struct Cand { float x1, y1, x2, y2, area; };   // offset boxes (class * max_wh added)

// per-anchor best class (first maximum), coalesced over anchors; used when the decode kernel did not provide it
__global__ __launch_bounds__(256) void best_class_kernel(const float* __restrict__ pred, int nc, int A, long long total, float* __restrict__ best_score,
                                                        int* __restrict__ best_cls) {
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
    const int a = (int)(g % A);
    const long long b = g / A;
    const float* pb = pred + (size_t)b * (4 + nc) * A + a;
    float best = pb[(size_t)4 * A];
    int bc = 0;
    for (int c = 1; c < nc; ++c) {
      const float v = pb[(size_t)(4 + c) * A];
      if (v > best) { best = v; bc = c; }
    }
    best_score[g] = best;
    best_cls[g] = bc;
  }
}

// GK: the P sort keys (and the candidate geometry) live in the workspace instead of LDS.  One workgroup still owns the image, so
// every pass is ordered by __syncthreads() alone (a workgroup's global accesses go through its own CU's L1).
// Invariant on the early `return` below: the waves with tid >= nthr leave BEFORE any later barrier; on gfx950 an s_barrier
// counts only the waves of the workgroup that are still alive, so the remaining waves synchronise among themselves.
template <bool GK>
__global__ __launch_bounds__(NMS_THREADS) void yolo_nms_kernel(const float* __restrict__ pred, const float* __restrict__ best_score,
                                                              const int* __restrict__ best_cls, int nc, int A, float conf_thres,
                                                              float iou_thres, int max_det, float max_wh, float* __restrict__ out_det,
                                                              int* __restrict__ out_idx, int* __restrict__ out_count,
                                                              char* __restrict__ workspace, int P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int nb = gridDim.x;
  unsigned long long* keys;                                                          // [P]
  unsigned long long* supp;                                                          // [P/64] bitmask, always LDS
  if constexpr (GK) {
    static_assert(sizeof(Cand) + sizeof(float) + sizeof(int) == 28, "nms_keys_offset assumes 28 bytes per anchor");
    keys = reinterpret_cast<unsigned long long*>(workspace + nms_keys_offset(nb, A)) + (size_t)b * P;
    supp = reinterpret_cast<unsigned long long*>(smem);
  } else {
    keys = reinterpret_cast<unsigned long long*>(smem);
    supp = keys + P;
  }
  int& s_count = *reinterpret_cast<int*>(supp + (P >> 6));                           // all LDS dynamic (16-B aligned base)
  const float* pb = pred + (size_t)b * (4 + nc) * A;
  const int* cls_ws = best_cls + (size_t)b * A;
  const float* sc_ws = best_score + (size_t)b * A;
  Cand* cand = reinterpret_cast<Cand*>(workspace + (size_t)b * (size_t)A * sizeof(Cand));

  if (tid == 0) s_count = 0;
  for (int i = tid; i < P; i += NMS_THREADS) keys[i] = 0ull;
  __syncthreads();

  // 1. candidates
  for (int a = tid; a < A; a += NMS_THREADS) {
    const float best = sc_ws[a];
    if (best > conf_thres) {
      const int slot = atomicAdd(&s_count, 1);
      keys[slot] = ((unsigned long long)__float_as_uint(best) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)a);
    }
  }
  __syncthreads();
  const int n = s_count;
  // From here on every step is a short data-parallel pass followed by a barrier.  With up to 2048 candidates the passes fit four
  // waves (one per SIMD): the other twelve retire now, and a barrier among four waves is several times cheaper than among sixteen
  // (36 sort passes + one barrier per kept box at 256 candidates: 52 -> 20 us per image).
  const int nthr = n <= 2048 ? 256 : NMS_THREADS;
  if (tid >= nthr) return;

  // 2. bitonic sort, descending, over the smallest power of two >= n
  int Ps = 1;
  while (Ps < n) Ps <<= 1;
  for (int k = 2; k <= Ps; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < Ps / 2; t += nthr) {
        const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int hi = lo | j;
        const bool desc = (lo & k) == 0;
        const unsigned long long x = keys[lo], y = keys[hi];
        if ((x < y) == desc) { keys[lo] = y; keys[hi] = x; }
      }
      __syncthreads();
    }
  }

  // 3. candidate geometry in sorted order: in LDS behind the Ps sorted keys when it fits, in the workspace otherwise
  const bool geo_lds = !GK && (size_t)Ps * 8 + (size_t)n * sizeof(Cand) <= (size_t)P * 8;
  // boxes as 16-byte vectors + a separate area array (one ds_read_b128 + one ds_read_b32 per candidate instead of five scalar reads)
  f32x4* const box_l = reinterpret_cast<f32x4*>(keys + (Ps > 2 ? Ps : 2));
  float* const area_l = reinterpret_cast<float*>(box_l + n);
  f32x4* const box_g = reinterpret_cast<f32x4*>(cand);
  float* const area_g = reinterpret_cast<float*>(box_g + A);
  for (int i = tid; i < n; i += nthr) {
    const int a = (int)(0xFFFFFFFFu - (unsigned)(keys[i] & 0xFFFFFFFFull));
    const float cx = pb[a], cy = pb[(size_t)A + a], w = pb[(size_t)2 * A + a], h = pb[(size_t)3 * A + a];
    const float hw = w / 2.f, hh = h / 2.f;
    const float off = (float)cls_ws[a] * max_wh;
    f32x4 c;
    c[0] = (cx - hw) + off; c[1] = (cy - hh) + off; c[2] = (cx + hw) + off; c[3] = (cy + hh) + off;
    const float area = (c[2] - c[0]) * (c[3] - c[1]);
    if (geo_lds) { box_l[i] = c; area_l[i] = area; } else { box_g[i] = c; area_g[i] = area; }
  }
  for (int i = tid; i < (P >> 6); i += nthr) supp[i] = 0ull;
  __threadfence_block();
  __syncthreads();

  // 4. greedy suppression (kept sorted positions -> LDS list); a kept box costs one LDS round and one barrier
  int* const kept_pos = reinterpret_cast<int*>(&s_count + 4);                        // [max_det], behind the counter
  int kept = 0;
  int i = 0;
  while (kept < max_det) {
    // next unsuppressed candidate at or after i (uniform across the block: all threads read LDS)
    int word = i >> 6;
    const int nwords = (n + 63) >> 6;
    int found = -1;
    while (word < nwords) {
      unsigned long long m = ~supp[word];
      if (word == (i >> 6)) m &= ~0ull << (i & 63);
      if (m) { found = (word << 6) + __builtin_ctzll(m); break; }
      ++word;
    }
    if (found < 0 || found >= n) break;
    i = found;
    const f32x4 ci = geo_lds ? box_l[i] : box_g[i];
    const float ai = geo_lds ? area_l[i] : area_g[i];
    if (tid == 0) kept_pos[kept] = i;
    ++kept;
    for (int j = i + 1 + tid; j < n; j += nthr) {
      if ((supp[j >> 6] >> (j & 63)) & 1ull) continue;
      const f32x4 cj = geo_lds ? box_l[j] : box_g[j];
      const float xx1 = fmaxf(ci[0], cj[0]), yy1 = fmaxf(ci[1], cj[1]);
      const float xx2 = fminf(ci[2], cj[2]), yy2 = fminf(ci[3], cj[3]);
      const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
      const float inter = w * h;
      if (inter > 0.f || iou_thres < 0.f) {            // (disjoint boxes -- other classes, mostly -- have ovr = 0: never above a threshold >= 0)
        const float aj = geo_lds ? area_l[j] : area_g[j];
        const float ovr = inter / (ai + aj - inter);
        if (ovr > iou_thres) atomicOr(&supp[j >> 6], 1ull << (j & 63));
      }
    }
    ++i;
    __syncthreads();
  }
  // 5. detections of the kept candidates, one thread each
  for (int k = tid; k < kept; k += nthr) {
    const unsigned long long key = keys[kept_pos[k]];
    const int a = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
    const float cx = pb[a], cy = pb[(size_t)A + a], w = pb[(size_t)2 * A + a], h = pb[(size_t)3 * A + a];
    const float hw = w / 2.f, hh = h / 2.f;
    float* o = out_det + ((size_t)b * max_det + k) * 6;
    o[0] = cx - hw; o[1] = cy - hh; o[2] = cx + hw; o[3] = cy + hh;
    o[4] = __uint_as_float((unsigned)(key >> 32));
    o[5] = (float)cls_ws[a];
    out_idx[(size_t)b * max_det + k] = a;
  }
  if (tid == 0) out_count[b] = kept;
}

}  // namespace

extern "C" size_t cvmi_yolo_nms_workspace(int B, int A) {
  if (B <= 0 || A <= 0) return 0;
  size_t P = 1024;
  while (P < (size_t)A) P <<= 1;
  return nms_keys_offset(B, A) + (A > NMS_LDS_A ? (size_t)B * P * 8 : 0);
}

static int nms_launch(const float* pred, const float* best_score, const int* best_cls, int B, int nc, int A, float conf_thres, float iou_thres,
                      int max_det, float max_wh, float* out_det, int* out_idx, int* out_count, void* workspace, hipStream_t stream) {
  int P = 1024;
  while (P < A) P <<= 1;
  const bool gk = A > NMS_LDS_A;
  const size_t lds = (gk ? 0 : (size_t)P * 8) + (size_t)(P / 64) * 8 + 16 + (size_t)max_det * sizeof(int);
  // once per process; std::call_once so that concurrent first calls from two host threads both see the attribute set
  static std::once_flag attr_once;
  static hipError_t attr_err = hipSuccess;
  std::call_once(attr_once, [] {
    attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&yolo_nms_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   NMS_LDS_A * 8 + (NMS_LDS_A / 64) * 8 + 16 + NMS_MAX_DET * (int)sizeof(int));
  });
  CVMI_HIP(attr_err);
  if (gk)
    hipLaunchKernelGGL(yolo_nms_kernel<true>, dim3(B), dim3(NMS_THREADS), lds, stream, pred, best_score, best_cls, nc, A, conf_thres, iou_thres,
                       max_det, max_wh, out_det, out_idx, out_count, (char*)workspace, P);
  else
    hipLaunchKernelGGL(yolo_nms_kernel<false>, dim3(B), dim3(NMS_THREADS), lds, stream, pred, best_score, best_cls, nc, A, conf_thres, iou_thres,
                       max_det, max_wh, out_det, out_idx, out_count, (char*)workspace, P);
  CVMI_LAUNCH_CHECK();
  return 0;
}

extern "C" int cvmi_yolo_nms(const float* pred, int B, int nc, int A, float conf_thres, float iou_thres, int max_det, float max_wh,
                             float* out_det, int* out_idx, int* out_count, void* workspace, cvmi_stream_t stream_) {
  CVMI_CHECK(pred && out_det && out_idx && out_count && workspace, "yolo_nms: null pointer");
  CVMI_CHECK(B > 0 && nc > 0 && A > 0 && max_det > 0 && max_det <= NMS_MAX_DET, "yolo_nms: bad shape");
  CVMI_CHECK(A <= NMS_MAX_A, "yolo_nms: A=%d exceeds %d anchors", A, NMS_MAX_A);
  CVMI_CHECK(conf_thres >= 0.f, "yolo_nms: conf_thres must be >= 0 (keys rely on non-negative scores)");
  hipStream_t stream = (hipStream_t)stream_;
  char* ws = (char*)workspace;
  float* bs = reinterpret_cast<float*>(ws + (size_t)B * A * sizeof(Cand));
  int* bc = reinterpret_cast<int*>(bs + (size_t)B * A);
  const long long total = (long long)B * A;
  long long g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(best_class_kernel, dim3((unsigned)g), dim3(256), 0, stream, pred, nc, A, total, bs, bc);
  CVMI_LAUNCH_CHECK();
  return nms_launch(pred, bs, bc, B, nc, A, conf_thres, iou_thres, max_det, max_wh, out_det, out_idx, out_count, workspace, stream);
}

extern "C" int cvmi_yolo_nms_best(const float* pred, const float* best_score, const int* best_cls, int B, int nc, int A, float conf_thres,
                                  float iou_thres, int max_det, float max_wh, float* out_det, int* out_idx, int* out_count, void* workspace,
                                  cvmi_stream_t stream_) {
  CVMI_CHECK(pred && best_score && best_cls && out_det && out_idx && out_count && workspace, "yolo_nms_best: null pointer");
  CVMI_CHECK(B > 0 && nc > 0 && A > 0 && max_det > 0 && max_det <= NMS_MAX_DET, "yolo_nms_best: bad shape");
  CVMI_CHECK(A <= NMS_MAX_A, "yolo_nms_best: A=%d exceeds %d anchors", A, NMS_MAX_A);
  CVMI_CHECK(conf_thres >= 0.f, "yolo_nms_best: conf_thres must be >= 0");
  return nms_launch(pred, best_score, best_cls, B, nc, A, conf_thres, iou_thres, max_det, max_wh, out_det, out_idx, out_count, workspace,
                    (hipStream_t)stream_);
}
