/*
 * cvmi355.h -- C ABI of libcvmi355.so: the MI355X (gfx950) kernels behind CircuitVision's
 * dense-vision hot path (YOLO11 detector + SAM 2.1 forward).
 *
 * The reference (JKc66/CircuitVision) is pure Python: its boundary for this path is Python duck
 * typing (`YOLO(path).predict(img)`, `get_modified_sam2(...)(x)`), and the
 * arithmetic lives in torch / torchvision / ultralytics / sam2 wheels.  The entry points below are
 * what a maintainer binds (ctypes) to replace those wheels' operators.  Each
 * one cites the reference call site whose arithmetic it replaces (file:line under /root/reference).
 *
 * Conventions
 *   - plain C: device pointers + sizes; no torch types.  All device buffers are owned by the
 *     caller; kernels never allocate.  `stream` is a hipStream_t passed as void*.
 *   - every function returns 0 on success, non-zero on error; cvmi_last_error() returns a
 *     thread-local message.  Nothing aborts the process (circuit_analyzer.py:255-263, :381-386
 *     catch ordinary exceptions).
 *   - activations are NHWC ("pixels x channels", channel contiguous); a tensor view is
 *     (ptr, ld) where ld = elements between consecutive pixels, so channel slices of a wider
 *     buffer (zero-copy concat / split) are first-class.
 *   - dtype: CVMI_F16 (fp16 storage, fp32 accumulate), CVMI_F32 (exact-f32 MFMA, parity mode) or -- on the SAM 2 path's entry points
 *     (conv2d, attention, layernorm, cast, maxpool2x2, space_to_depth4, nchw_to_nhwc, prompt_tokens, hyper_masks, sam2_transform,
 *     hiera_mlp, tok_linear) -- CVMI_BF16 (bf16 storage / MFMA operands, fp32 accumulate; BASELINE configs[4]).
 */
#ifndef CVMI355_H
#define CVMI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CVMI_VERSION 121

typedef void* cvmi_stream_t; /* hipStream_t */

enum { CVMI_F16 = 0, CVMI_F32 = 1, CVMI_BF16 = 2 };   /* BF16: bf16 storage / operands, fp32 accumulate (SAM 2 path entry points) */
enum { CVMI_ACT_NONE = 0, CVMI_ACT_SILU = 1, CVMI_ACT_RELU = 2, CVMI_ACT_GELU = 3, CVMI_ACT_SIGMOID = 4 };

/* ---- library -------------------------------------------------------------------------------- */
int cvmi_version(void);
const char* cvmi_last_error(void);
/* Diagnostic: the kernel (template instance) the calling thread's most recent cvmi_conv2d / cvmi_attention / cvmi_tok_linear* /
 * cvmi_hiera_mlp* call dispatched to, as rocprofv3 names it up to the spelling of the 16-bit type (e.g. "gemm256x192_kernel<float>").  Read-and-clear: "" when no such
 * call happened since the last read. */
const char* cvmi_last_kernel(void);
/* fills: [0] CU count, [1] wave size, [2] LDS bytes per CU-workgroup, [3] gfx arch number (950) */
int cvmi_device_info(int device, int* out4);
/* ABI guard for the descriptor structs below: sizeof(struct) as THIS library was compiled, for kind = CVMI_DESC_CONV / _C3K2 / _DWPW /
 * _ATTN (0 for an unknown kind).  A binding compares it with the size of its own mirror of the struct once, at load, and refuses to
 * continue on a mismatch (circuitvision_amd/_lib.py does; INTEGRATION.md shows the check): a struct that is short by a trailing field
 * would otherwise be read past its end. */
enum { CVMI_DESC_CONV = 0, CVMI_DESC_C3K2 = 1, CVMI_DESC_DWPW = 2, CVMI_DESC_ATTN = 3 };
size_t cvmi_desc_size(int kind);

/* ---- HIP graph capture of a launch sequence (replaces per-op Python dispatch) ---------------- */
int cvmi_graph_begin(cvmi_stream_t stream);
int cvmi_graph_end(cvmi_stream_t stream, void** graph_exec_out);
int cvmi_graph_launch(void* graph_exec, cvmi_stream_t stream);
int cvmi_graph_destroy(void* graph_exec);

/* ---- fused convolution / linear layer as implicit GEMM on MFMA -------------------------------
 * y[m, n] = act( sum_k A[m, k] * w[n, k] + bias[n] ) (+ res[m, n])
 * A is gathered on the fly (im2col) from up to two NHWC sources that are concatenated along the
 * channel axis (source 0 first), each optionally nearest-2x upsampled (up = 1).
 * k = (ky*KW + kx)*(c0+c1) + c.  m = (b*OH + oy)*OW + ox.
 * Replaces: ultralytics Conv(+BN folded)+SiLU, Concat, Upsample inside YOLO.predict
 * (circuit_analyzer.py:268); nn.Linear / 1x1 Conv2d of Hiera, FpnNeck, mask decoder inside
 * SAM2ImageWrapper.forward (sam2_infer.py:226-260).
 * w: [Npad][Kpad] (dtype), zero padded, Npad % 128 == 0, Kpad % 32 == 0.  bias: [Npad] f32.
 * c0, c1 must be multiples of 16/sizeof(dtype) unless scalar_gather = 1 (e.g. 3-channel stems).
 */
typedef struct cvmi_conv_desc {
  const void* x0; const void* x1;   /* sources (x1 may be NULL when c1 == 0) */
  const void* w; const float* bias;
  const void* res;                  /* optional residual, added after the activation */
  void* y;
  int x0_ld, x1_ld, res_ld, y_ld;   /* elements between consecutive pixels */
  int c0, c1;                       /* channels taken from each source */
  int up0, up1;                     /* 1: source is at half resolution (nearest 2x upsample) */
  int B, H, W;                      /* logical input size (after upsampling) */
  int OH, OW;
  int KH, KW, stride, pad;
  int N;                            /* output channels */
  int Kpad;                         /* padded K (row length of w) */
  int act;                          /* CVMI_ACT_* */
  int dtype;                        /* CVMI_F16 / CVMI_F32: type of x, w */
  int out_f32;                      /* 1: y and res are f32 even when dtype is F16 */
  int scalar_gather;                /* 1: per-element gather (channel count not vectorizable) */
  int res_mod;                      /* > 0: residual row = m % res_mod (batch-broadcast constants) */
  int act_after_res;                /* 1: y = act(conv + bias + res) instead of act(conv + bias) + res */
  int shuffle_cout;                 /* > 0: ConvTranspose2d(k=2,s=2) as GEMM: N = 4*shuffle_cout, column
                                       n = (dy*2+dx)*shuffle_cout + co is stored at output pixel
                                       (2*oy+dy, 2*ox+dx), channel co of a [B,2*OH,2*OW,*] tensor (y, res) */
  int res_rep;                      /* > 1 (with shuffle_cout, or with res_mod == OH * OW): the residual has B / res_rep images, image b
                                       reads residual image b / res_rep (tensors shared by the prompts of an image) */
  float* row_stats;                 /* optional (f32 output + residual, plain 16-bit GEMM with N % 192 == 0 and K >= 1024, i.e. the Hiera stage-3
                                       fc2 shape): float[rows][N / 96][2] = (mean, sum of squared deviations from that mean) of the values
                                       written to each row's 96-column slices -- the statistics of the NEXT LayerNorm over these rows, free of
                                       the E[x^2] - E[x]^2 cancellation (cvmi_tok_linear_stats with ln_stats_in_parts = N / 96 combines the
                                       slices); NULL = off */
} cvmi_conv_desc;
int cvmi_conv2d(const cvmi_conv_desc* d, cvmi_stream_t stream);

/* ---- fused C3k2 block (ultralytics C3k2.forward with c3k = False, n = 1; YOLO11 model.2 / .4 / .16 at scale n):
 *   [a|b] = SiLU(cv1 x) (1x1, c1 -> 2c; only when fuse_cv1, else x IS the [a|b] tensor of 2c channels),
 *   t = SiLU(m.cv1 (*) b) (3x3, c -> h), m = b + SiLU(m.cv2 (*) t) (3x3, h -> c; `shortcut`), y = SiLU(cv2 [a|b|m]) (1x1, 3c -> c2).
 * One launch, intermediates in LDS.  Weights / biases are cvmi_conv2d-packed ([Npad >= 128][kpad], BN folded).
 * cvmi_c3k2_supported() tells which (c1, c, h, c2) configurations are built (fp16 only). */
typedef struct {
  const void* x; void* y;
  const void *w0, *w1, *w2, *w3;
  const float *b0, *b1, *b2, *b3;
  int x_ld, y_ld, kpad0, kpad1, kpad2, kpad3;
  int B, H, W, c1, c, h, c2;
  int fuse_cv1, shortcut, dtype;
} cvmi_c3k2_desc;
int cvmi_c3k2_supported(int c1, int c, int h, int c2, int fuse_cv1, int dtype);
int cvmi_c3k2(const cvmi_c3k2_desc* d, cvmi_stream_t stream);

/* ---- fused YOLO11 stem: model.0 (Conv 3x3 s2, 3 -> c0) + model.1 (Conv 3x3 s2, c0 -> c1), SiLU each -------
 * Replaces the first two layers of ultralytics' DetectionModel inside YOLO.predict (circuit_analyzer.py:268) in one
 * launch: the c0-channel half-resolution map never reaches HBM.  x: the space-to-depth(2) image cvmi_letterbox(s2d=1)
 * writes, [B, H2, W2, x_ld >= 16] (12 real channels); w0 / b0: model.0 as the equivalent 2x2 / stride-1 conv over the
 * 16 s2d channels, taps at block offsets (-1, 0), cvmi_conv2d-packed (k = (ty*2 + tx)*16 + c); w1 / b1: model.1,
 * cvmi_conv2d-packed; y: [B, ceil(H2/2), ceil(W2/2), y_ld >= c1].  fp16, (c0, c1) = (16, 32) (YOLO11-n) is built. */
int cvmi_stem2_supported(int c0, int c1, int dtype);
int cvmi_stem2(const void* x, int x_ld, const void* w0, const float* b0, int kpad0, const void* w1, const float* b1, int kpad1,
               void* y, int y_ld, int B, int H2, int W2, int c0, int c1, int dtype, cvmi_stream_t stream);

/* ---- fused depthwise 3x3 + pointwise 1x1 (+ chained 1x1): YOLO11 Detect class branch ---------
 * Replaces, inside YOLO.predict (circuit_analyzer.py:268), the ultralytics Detect.cv3[i] sub-chains
 *   N2 == 0:  y = SiLU(pw1 * SiLU(dw (*) x))                       [DWConv(C, C, 3), Conv(C, N1, 1)]
 *   N2 >  0:  y = w2 * SiLU(pw1 * SiLU(dw (*) x)) + b2             [... , Conv2d(N1, N2, 1)] (class logits, no activation)
 * in one launch each (intermediates stay in LDS).  wd / bd: cvmi_dwconv3x3 layout ([9][C] tap-major, bias [C]);
 * w1 / b1, w2 / b2: cvmi_conv2d-packed ([Npad >= 128][kpad], zero padded, BN folded).  Only the first N2 (or N1)
 * channels of a y pixel are written.  cvmi_dwpw_supported() tells which (C, N1, N2) are built (fp16 only). */
typedef struct {
  const void* x; void* y;
  const void* wd; const float* bd;
  const void* w1; const float* b1;
  const void* w2; const float* b2;
  int x_ld, y_ld, kpad1, kpad2;
  int B, H, W, C, N1, N2;
  int dtype;
} cvmi_dwpw_desc;
int cvmi_dwpw_supported(int C, int N1, int N2, int dtype);
int cvmi_dwpw(const cvmi_dwpw_desc* d, cvmi_stream_t stream);

/* ---- depthwise 3x3 stride-1 conv + bias + act (YOLO Detect cls branch, C2PSA pe) -------------
 * w: [9][C] (tap-major), bias [C] f32.  Replaces ultralytics DWConv inside YOLO.predict. */
int cvmi_dwconv3x3(const void* x, int x_ld, const void* w, const float* bias, const void* res,
                   int res_ld, void* y, int y_ld, int B, int H, int W, int C, int act, int dtype,
                   cvmi_stream_t stream);

/* ---- SPPF pooling: y1 = mp5(y0), y2 = mp5(y1), y3 = mp5(y2) (k=5,s=1,p=2) in one pass ---------
 * buf is the [B,H,W,ld] concat buffer; channels [0,C) hold y0; writes [C,2C), [2C,3C), [3C,4C). */
int cvmi_sppf_pool(void* buf, int ld, int B, int H, int W, int C, int dtype, cvmi_stream_t stream);

/* ---- scaled-dot-product attention, softmax over keys, one launch for all (batch, head) --------
 * Element (b, h, t, d) of q lives at q + b*q_sb + h*q_sh + t*q_st + d (elements); same for k, v, o.
 * Replaces: ultralytics Attention in C2PSA (YOLO.predict); Hiera MultiScaleAttention and the
 * mask decoder's Attention (sam2_infer.py:226, :252).
 * Window mode (win > 0): tokens of batch entry b are the win x win pixels of window
 *   (b / (gw*gh*...)) of a [img][gh*win][gw*win] grid laid out NHWC with pixel stride *_st;
 *   q_pool = 1 takes q as the 2x2 max-pool of the window's q tokens (Hiera q-pooling). */
typedef struct cvmi_attn_desc {
  const void* q; const void* k; const void* v; void* o;
  long long q_sb, q_sh, q_st, k_sb, k_sh, k_st, v_sb, v_sh, v_st, o_sb, o_sh, o_st;
  int B, heads, Nq, Nk, dqk, dv;
  float scale;
  int dtype;
  int win, grid_h, grid_w;   /* window mode (0 = off): window side, image grid size in pixels */
  int q_pool;                /* 1: q window is max-pooled 2x2 (Nq = (win/2)^2) */
  int q_bdiv, kv_bdiv;       /* > 1 (fp16, no window, head dims <= 64): q rows of batch entry b are read from entry b / q_bdiv,
                                k / v rows from entry b / kv_bdiv -- the prompts of one image sharing its image-side tensors
                                (SAM 2 MaskDecoder with repeat_image, first two-way layer); 0 / 1 = off */
  int av_fp8;                /* 1 (16-bit dtype, head_dim 72; BASELINE configs[4] "fp8 MFMA attention"): the softmax(QK^T) V contraction runs on the
                                block-scaled fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4): P as e4m3 of p * 2^8 (block scale 2^-8), V as e4m3,
                                fp32 accumulation, fp32 softmax statistics.  Honoured by the 256-key window kernel and the long-sequence
                                kernel (Hiera stage-3 windows / global blocks); shapes served by other kernels ignore it.  0 = off (default) */
  int q_log2;                /* 1: q was produced already multiplied by scale * log2(e) (e.g. folded into the rows of the projection that makes it):
                                softmax(q k^T) is then exp2 of the products as they stand and `scale` is ignored.  Same result as 0 with the
                                unscaled q; the long-sequence head_dim-72 kernel uses it to carry the running maximum through the matrix pipe. */
} cvmi_attn_desc;
int cvmi_attention(const cvmi_attn_desc* d, cvmi_stream_t stream);

/* ---- YOLO Detect decode: DFL expectation + dist2bbox + stride scale + class sigmoid -----------
 * box[l]: [B,Hl,Wl,64] (ld box_ld), cls[l]: [B,Hl,Wl,>=roundup(nc,16B)] (ld cls_ld) for l = 0..2.
 * pred: f32 [B, 4+nc, A], A = sum Hl*Wl (ultralytics layout: xywh rows, then class-score rows).
 * best_score f32 [B*A] / best_cls i32 [B*A] (optional, both or neither): per-anchor best class
 * (first maximum) for cvmi_yolo_nms_best.  write_cls = 0 skips the class-score rows of pred. */
int cvmi_detect_decode(const void* const* box, const int* box_ld, const void* const* cls,
                       const int* cls_ld, const int* hs, const int* ws, const float* strides,
                       int nlevels, int B, int nc, int dtype, float* pred, float* best_score,
                       int* best_cls, int write_cls, cvmi_stream_t stream);

/* ---- ultralytics-semantics NMS on the decoded predictions -------------------------------------
 * pred f32 [B, 4+nc, A].  Candidates: max class score > conf_thres; sorted by score desc (ties:
 * lower anchor index first); per-class via +cls*max_wh; greedy suppress IoU > iou_thres; first
 * max_det kept.  out_det f32 [B, max_det, 6] (x1,y1,x2,y2,conf,cls), out_idx i32 [B, max_det]
 * (anchor index), out_count i32 [B].  workspace: >= cvmi_yolo_nms_workspace(B, A) bytes.
 * Replaces ultralytics ops.non_max_suppression + torchvision.ops.nms (circuit_analyzer.py:268). */
size_t cvmi_yolo_nms_workspace(int B, int A);
int cvmi_yolo_nms(const float* pred, int B, int nc, int A, float conf_thres, float iou_thres,
                  int max_det, float max_wh, float* out_det, int* out_idx, int* out_count,
                  void* workspace, cvmi_stream_t stream);
/* same, with the per-anchor best class supplied by cvmi_detect_decode (only the 4 box rows of pred are read) */
int cvmi_yolo_nms_best(const float* pred, const float* best_score, const int* best_cls, int B, int nc,
                       int A, float conf_thres, float iou_thres, int max_det, float max_wh,
                       float* out_det, int* out_idx, int* out_count, void* workspace,
                       cvmi_stream_t stream);

/* ---- letterbox pre-processing (ultralytics LetterBox + BGR flip + /255) -----------------------
 * src u8 [H,W,3] -> one letterboxed image at dst: the resized region new_w x new_h is placed at
 * (left, top) of an out_h x out_w canvas, pad value 114; 8-bit fixed-point bilinear as OpenCV.
 * s2d = 0: dst is NHWC [out_h, out_w, 3].  s2d = 1: dst is space-to-depth(2) [out_h/2, out_w/2, 16],
 * channel ((y&1)*2 + (x&1))*3 + c, channels 12..15 zero -- the layout the YOLO stem conv reads. */
int cvmi_letterbox(const uint8_t* src, int H, int W, void* dst, int out_h, int out_w, int new_h,
                   int new_w, int top, int left, int dtype, int s2d, cvmi_stream_t stream);
/* The same for B equally sized sources in ONE launch: src u8 [B,H,W,3] contiguous (e.g. one pinned staging buffer copied by one
 * hipMemcpyAsync), image b written at dst + b * dst_image_stride elements (the detector's batched `predict([img, ...])`). */
int cvmi_letterbox_batch(const uint8_t* src, int B, int H, int W, void* dst, long long dst_image_stride, int out_h, int out_w,
                         int new_h, int new_w, int top, int left, int dtype, int s2d, cvmi_stream_t stream);

/* ---- dtype conversion / layout helpers -------------------------------------------------------- */
/* NCHW (f32 or f16) -> NHWC dtype */
int cvmi_nchw_to_nhwc(const void* src, int src_dtype, void* dst, int dst_dtype, int dst_ld, int B,
                      int C, int H, int W, cvmi_stream_t stream);
/* NHWC dtype -> NCHW f32 */
int cvmi_nhwc_to_nchw_f32(const void* src, int src_dtype, int src_ld, float* dst, int B, int C,
                          int H, int W, cvmi_stream_t stream);

/* ==== SAM 2.1 path (sam2_infer.py:220-275 and the un-vendored sam2 package behind it) ========= */

/* LayerNorm over the channel axis of rows x C (nn.LayerNorm in Hiera / TwoWayTransformer, LayerNorm2d
 * in the mask decoder's upscaling); optional activation after the affine.  x / y dtypes independent
 * (fp16 mode keeps the residual stream in f32 and feeds fp16 to the GEMMs). gamma, beta: f32 [C]. */
/* pad_w > 0: the rows are the pixels of [*, pad_h, pad_w] images and are written into a zero-padded
 * [*, pad_hp, pad_wp] grid (Hiera pads the NORMALISED tokens up to a window multiple; padding stays 0). */
int cvmi_layernorm(const void* x, int x_ld, int x_dtype, const float* gamma, const float* beta,
                   void* y, int y_ld, int y_dtype, long long rows, int C, float eps, int act,
                   int pad_h, int pad_w, int pad_hp, int pad_wp, cvmi_stream_t stream);
/* Same over an f32 stream, writing the f32 result to y AND a 16-bit (y2_dtype) copy to y2 (the GEMM-operand copy the next layer
 * reads: saves a cast pass over the stream; SAM 2 two-way transformer norm4, sam2_infer.py:252). */
int cvmi_layernorm_dual(const void* x, int x_ld, const float* gamma, const float* beta, void* y, int y_ld, void* y2, int y2_ld,
                        int y2_dtype /* CVMI_F16 | CVMI_BF16 */, long long rows, int C, float eps, cvmi_stream_t stream);

/* Fused MLP half of a Hiera block, in place on the f32 residual stream (sam2 hieradet MultiScaleBlock: `x = x + mlp(norm2(x))`,
 * behind sam2_infer.py:226):   x[r, :] += fc2( GELU( fc1( LayerNorm(x[r, :]; gamma, beta, eps) ) ) )   for r < rows, hidden = 4 C,
 * fp16 operands / fp32 accumulation; the normalised copy and the hidden activation never reach HBM.  C in {144, 288} are built
 * (cvmi_hiera_mlp_supported); stages with wider rows keep the separate LayerNorm + two cvmi_conv2d launches.
 * w_packed: both weight matrices in MFMA-fragment order, hidden chunk by hidden chunk (32 hidden units each), every fragment 64 lanes
 * x 8 fp16 = 1 KiB (lane l: r = l & 31, h = l >> 5):
 *   chunk j = [ F1(j, s) for s = 0 .. C/16 ]  ++  [ F2(j, t, s2) for t = 0 .. ceil(C/32) - 1, s2 = 0, 1 ]
 *   F1(j, s)[l][e]     = W1x[32 j + r][16 s + 8 h + e],  W1x = [ fc1.weight | fp16(b1) | fp16(b1 - fp16(b1)) | 0 ... ] (C + 16 columns:
 *                        the fc1 bias rides on two constant-1 input columns as a hi + lo fp16 pair)
 *   F2(j, t, s2)[l][e] = fc2.weight[32 t + r][32 j + 16 s2 + 8 (e >> 2) + 4 h + (e & 3)]   (0 for output rows >= C): the k order in
 *                        which a 32x32 MFMA accumulator, converted in place, serves as the next MFMA's B operand.
 * cvmi_hiera_mlp_packed_bytes(C) = (4 C / 32) * (C/16 + 1 + 2 ceil(C/32)) * 1024.  b2: fc2 bias, f32 [C]. */
int cvmi_hiera_mlp_supported(int C);
size_t cvmi_hiera_mlp_packed_bytes(int C);
int cvmi_hiera_mlp(void* x, int x_ld, const float* gamma, const float* beta, float eps, const void* w_packed,
                   const float* b2, long long rows, int C, int dtype /* CVMI_F16 | CVMI_BF16: type of w_packed */, cvmi_stream_t stream);
/* same; additionally writes per updated row (mean, 1 / sqrt(var + ln_stats_eps)) to ln_stats_out (float[2 * rows], or NULL): the statistics of
 * the NEXT LayerNorm over these rows (the next block's norm1), consumed by cvmi_tok_linear_stats / cvmi_tok_linear_pool_stats */
int cvmi_hiera_mlp_stats(void* x, int x_ld, const float* gamma, const float* beta, float eps, const void* w_packed, const float* b2,
                         long long rows, int C, int dtype, float* ln_stats_out, float ln_stats_eps, cvmi_stream_t stream);

/* Token-stationary linear layer for Hiera's short-K GEMMs (sam2 hieradet MultiScaleBlock: qkv(norm1(x)), x = shortcut + proj(attn),
 * mlp.layers[0](norm2(x)) + GELU; behind sam2_infer.py:226):
 *   y[r, n] = act( sum_k in[r, k] * W[n, k] + b[n] ),   r < rows (a multiple of 256), n < N, K in {144, 288, 576}
 *   in_f32_layernorm = 1: in is the f32 residual stream, normalised on the fly with gamma / beta / eps (the separate LayerNorm pass
 *                         and its fp16 copy disappear); 0: in is an fp16 matrix; 2: in is an f32 matrix converted as it is (K = 144, 288;
 *                         16-bit output, no activation: the neck's lateral convs read the f32 stage outputs, FpnNeck behind sam2_infer.py:226)
 *   out_f32_residual = 1: out is the f32 residual stream, updated in place (out[r, n] += y[r, n], act must be NONE);
 *                      0: out is fp16 (act NONE or GELU)
 * w_packed: ceil(N/32) chunks of (K/16 + 1) MFMA fragments of 1 KiB, fragment (j, s), lane l (r = l & 31, h = l >> 5), element e:
 *   Wx[32 j + r][16 s + 8 h + e],  Wx = [ W | fp16(b) | fp16(b - fp16(b)) | 0 ... ]  (K + 16 columns, rows >= N zero).
 * cvmi_tok_linear_packed_bytes(K, N) = ceil(N/32) * (K/16 + 1) * 1024. */
int cvmi_tok_linear_supported(int K);
size_t cvmi_tok_linear_packed_bytes(int K, int N);
/* Packed-weight format the library expects for this K: 32 = the layout above (32x32x16 MFMA fragments, bias on an extra k-step);
 * 16 (K = 576) = fragments of the 16x16x32 MFMA shape: ceil(N/32) chunks of 2 K/32 + 1 pieces of 1 KiB, piece (j, 2 s + hh), lane l
 * (r16 = l & 15, g = l >> 4), element e:  W[32 j + 16 hh + r16][32 s + 8 g + e]  (rows >= N zero);  the chunk's last piece holds its 32 bias
 * values as f32 in its first 128 bytes (rest zero).  cvmi_tok_linear_packed_bytes follows the format. */
int cvmi_tok_linear_format(int K);
int cvmi_tok_linear(const void* in, int in_ld, int in_f32_layernorm, const float* gamma, const float* beta, float eps,
                    const void* w_packed, void* out, int out_ld, int out_f32_residual, long long rows, int K, int N, int act,
                    int dtype /* CVMI_F16 | CVMI_BF16: type of w_packed and of the 16-bit input / output */, cvmi_stream_t stream);

/* cvmi_tok_linear with LayerNorm statistics handed from the launch that writes the f32 residual stream to the launch that normalises it
 * (sam2 hieradet MultiScaleBlock: x = shortcut + proj(attn), then mlp.layers[0](norm2(x)); behind sam2_infer.py:226):
 *   ln_stats_out (out_f32_residual = 1, may be NULL): float[2 * rows] = per row (mean, 1 / sqrt(var + ln_stats_eps)) over the N updated values
 *   ln_stats_in  (in_f32_layernorm = 1, may be NULL): the same pairs; the LayerNorm prologue then reads every row once instead of twice.
 *   ln_stats_in_parts = P > 0: ln_stats_in is float[rows][P][2] of per-slice (mean, sum of squared deviations) instead, P equal column slices
 *                     of the row (cvmi_conv_desc.row_stats). */
int cvmi_tok_linear_stats(const void* in, int in_ld, int in_f32_layernorm, const float* gamma, const float* beta, float eps,
                          const void* w_packed, void* out, int out_ld, int out_f32_residual, long long rows, int K, int N, int act,
                          int dtype, const float* ln_stats_in, int ln_stats_in_parts, float* ln_stats_out, float ln_stats_eps,
                          cvmi_stream_t stream);
/* The layout of ln_stats_out for a residual-form launch of this shape.  0: float[rows][2] of (mean, rstd) as above.  P > 0: the launch has
 * fewer 256-row blocks than the chip has CUs (the per-rank batch of an 8-GPU job: SAM 2.1-L at 8 images), so P workgroups share a row block,
 * each writing its output-channel slice, and ln_stats_out is float[rows][P][2] of per-slice (mean, sum of squared deviations) -- exactly what
 * the consumer takes with ln_stats_in_parts = P.  Depends on (rows, K, N) only; K = 576 (the 16x16x32 format) is the only K that splits.
 * Requirements of that format beyond the ones above: 16-bit output needs N % 8 == 0 and out_ld % 8 == 0; in_f32_layernorm = 2 is not built. */
int cvmi_tok_linear_stats_parts(long long rows, int K, int N);

/* Shortcut path of a Hiera q-pooling block in ONE launch: out[b, y, x, :] = max over the 2 x 2 token block of
 * ( LayerNorm(in[b, 2y + dy, 2x + dx, :]) W^T + bias )  = `do_pool(self.proj(norm1(x)))` of sam2 hieradet MultiScaleBlock.forward (behind
 * /root/reference/src/sam2_infer.py:226).  in: f32 [B, H, W, K] token grid (row stride in_ld); out: f32 [B, H/2, W/2, N] (row stride out_ld);
 * w_packed as for cvmi_tok_linear; H, W even, B*H*W a multiple of 256, K in {144, 288, 576}. */
int cvmi_tok_linear_pool(const void* in, int in_ld, const float* gamma, const float* beta, float eps, const void* w_packed, void* out,
                         int out_ld, int B, int H, int W, int K, int N, int dtype, cvmi_stream_t stream);
/* same, with the rows' LayerNorm statistics supplied (float[2 * B*H*W] of (mean, rstd) pairs, or NULL) */
int cvmi_tok_linear_pool_stats(const void* in, int in_ld, const float* gamma, const float* beta, float eps, const void* w_packed, void* out,
                               int out_ld, int B, int H, int W, int K, int N, int dtype, const float* ln_stats_in, cvmi_stream_t stream);

/* Diagnostic only (CVMI_TOKLIN_STAMP=1 selects a stamped build of the K = 576 LayerNorm form of cvmi_tok_linear, never for timing
 * runs): reads and clears 24 s_memtime sums (shader cycles) of workgroup 0.  Ping-pong schedule: [0..8] = wave 0 {b1 wait, MFMAs,
 * vmcnt wait, b2 wait, epilogue, prefetch issue, whole kernel, chunks, launches}, [12..20] = wave 4 {b1 wait, epilogue, b2 wait,
 * MFMAs, vmcnt wait, prefetch issue, whole kernel, chunks, launches}.  One-barrier schedule (CVMI_TOKLIN_PP=0): [0..6] = {vmcnt wait,
 * barrier, issue (epilogue + prefetch), MFMAs, whole kernel, chunks, launches}. */
int cvmi_debug_stamps(unsigned long long* out24);

/* 2x2 / stride 2 max-pool, NHWC (Hiera shortcut path of the q-pooling blocks: do_pool(proj(x))). */
int cvmi_maxpool2x2(const void* x, int x_ld, void* y, int y_ld, int B, int H, int W, int C, int dtype,
                    cvmi_stream_t stream);

/* Space-to-depth(4) of a 3-channel NHWC image: [B,H,W,3] -> [B,H/4,W/4,48], channel (sy*4 + sx)*3 + c.  With it Hiera's
 * PatchEmbed (Conv2d 3 -> E, 7x7, stride 4, pad 3; sam2 hieradet.py) is a 2x2 / stride-1 conv over 48 channels
 * (block taps -1, 0), i.e. a vectorised cvmi_conv2d instead of a per-element gather. */
int cvmi_space_to_depth4(const void* x, void* y, int B, int H, int W, int dtype, cvmi_stream_t stream);

/* rows x C copy with dtype conversion (f32 <-> f16). */
int cvmi_cast(const void* x, int x_ld, int x_dtype, void* y, int y_ld, int y_dtype, long long rows, int C,
              cvmi_stream_t stream);

/* Prompt tokens of the SAM 2 mask decoder for n prompts of K labelled points each (upstream PromptEncoder
 * `_embed_points` + MaskDecoder token concat; box = corners labelled 2 / 3 + one padding point labelled -1):
 *   tokens[i, 0:T0]   = out_tokens (obj-score, iou, 4 mask tokens)
 *   tokens[i, T0 + j] = label -1 ? table[0] : [sin, cos](2*pi * ((2*(coords[i,j] + 0.5)/image_size - 1) @ gauss)) + table[1 + label]
 * coords f32 [n,K,2] (x, y in input pixels), labels i32 [n,K] in {-1,0,1,2,3}, gauss f32 [2,128],
 * out_tokens f32 [T0,256], table f32 [5,256] (not_a_point, point_embeddings 0..3).
 * Writes tokens_f32 [n, T0+K, 256] and (optional) the same values in `dtype` to tokens_lp. */
int cvmi_prompt_tokens(const float* coords, const int* labels, const float* gauss, const float* out_tokens,
                       const float* table, float image_size, float* tokens_f32, void* tokens_lp, int dtype,
                       int n, int K, int T0, cvmi_stream_t stream);

/* dst[b*rep + r] = src[b] for r < rep: image-major broadcast of per-image tensors to the prompts of each image
 * (upstream MaskDecoder repeat_image=True).  bytes_per_image must be a multiple of 16. */
int cvmi_repeat_images(const void* src, void* dst, long long bytes_per_image, int B, int rep, cvmi_stream_t stream);

/* Mask decoder tail: masks[b,i,p] = sum_c hyper[b,i,c] * up[b,p,c] for the 4 mask tokens (c = 32),
 * plus the stability counters of token 0 (area over +delta / -delta) for
 * dynamic_multimask_via_stability.  hyper f32 [B,4,hyper_ld], up dtype [B,P,up_ld], masks f32 [B,4,P],
 * areas i32 [B,2] (zeroed by this call). */
int cvmi_hyper_masks(const float* hyper, int hyper_ld, const void* up, int up_ld, int up_dtype, int C,
                     float* masks, int* areas, int B, int P, float delta, cvmi_stream_t stream);
/* Select per image: token 0 when dynamic == 0 or stability >= thresh, else 1 + argmax(iou[1:4]).
 * iou f32 [B, iou_ld] (4 values used).  Writes low_res f32 [B,P], iou_out f32 [B], sel i32 [B]. */
int cvmi_select_mask(const float* masks, const int* areas, const float* iou, int iou_ld, int dynamic,
                     float thresh, float* low_res, float* iou_out, int* sel, int B, int P,
                     cvmi_stream_t stream);

/* F.interpolate(bilinear, align_corners=False) of f32 planes [N,h,w] -> [N,H,W] (sam2_infer.py:263-268,
 * postprocess_masks :127).  mask_u8 (optional): also writes (value > thresh) ? 255 : 0. */
int cvmi_bilinear_f32(const float* x, int N, int h, int w, float* y, int H, int W, uint8_t* mask_u8,
                      float thresh, cvmi_stream_t stream);   /* y may be NULL when mask_u8 is given */

/* The reference's whole mask post-processing in one pass (circuit_analyzer.py:354-370): postprocess_masks' bilinear resize
 * (sam2_infer.py:127) -> `> thresh` -> u8 {0, 255} (:356-357) -> bounding rectangle of the non-zero pixels (:364-370,
 * cv2.findContours + boundingRect), WITHOUT materialising the f32 [N,H,W] map: only the u8 mask and four ints per plane
 * are written.  extent[n] = {min x, min y, max x, max y} or {W, H, -1, -1} for an empty mask. */
int cvmi_mask_postprocess(const float* x, int N, int h, int w, int H, int W, float thresh, uint8_t* mask_u8,
                          int* extent, cvmi_stream_t stream);
/* The same for planes that return to DIFFERENT sizes (each image's own crop window): sizes is a HOST array of N x {H_n, W_n}; plane n is
 * written at mask_u8 + sum over m < n of H_m * W_m (planes packed back to back); extent[n] as above in plane n's own coordinates. */
int cvmi_mask_postprocess_sizes(const float* x, int N, int h, int w, const int* sizes, float thresh, uint8_t* mask_u8, int* extent,
                                cvmi_stream_t stream);

/* Extent of N binary u8 planes [N,H,W]: extent[n] = {min x, min y, max x, max y} over the non-zero pixels, or
 * {W, H, -1, -1} for an empty plane.  Replaces cv2.findContours(RETR_EXTERNAL) + cv2.boundingRect on the SAM 2 mask
 * (circuit_analyzer.py:364-370): sam_extent_bbox = (min x, min y, max x + 1, max y + 1). */
int cvmi_mask_extent(const uint8_t* mask, int N, int H, int W, int* extent, cvmi_stream_t stream);

/* Fused 'bilinear upsample to HxW' + MultiKernelRefinement (sam2_infer.py:130-189, :263-272):
 * nk parallel convs (1 -> ic channels, odd kernels ks[], zero 'same' padding) + exact GELU + 1x1
 * combiner.  The ic*nk x H x W intermediate never leaves LDS/registers.
 * params f32: for each branch j: w[ic][ks_j][ks_j] then b[ic]; then combiner w[nk*ic], b[1]. */
int cvmi_upsample_refine(const float* low, int N, int h, int w, float* high, int H, int W,
                         const float* params, const int* ks, int nk, int ic, cvmi_stream_t stream);

/* SAM2Transforms.__call__ (sam2_infer.py:49-51): u8 HWC -> /255 -> antialiased bilinear resize to
 * R x R -> ImageNet normalise; written NHWC with 3 channels in dst_dtype. */
int cvmi_sam2_transform(const uint8_t* src, int H, int W, void* dst, int R, int dst_dtype,
                        cvmi_stream_t stream);
/* The same for B equally sized sources in ONE launch (SAM2Transforms.forward_batch, sam2_infer.py:53-56): src u8 [B,H,W,3] contiguous,
 * dst [B,R,R,3].  swap_rb = 1 reads the source channels reversed: circuit_analyzer.py:343 applies cv2.COLOR_BGR2RGB to the image it
 * is handed, which on the device is an index, not a pass over the image. */
int cvmi_sam2_transform_batch(const uint8_t* src, int B, int H, int W, void* dst, int R, int dst_dtype, int swap_rb,
                              cvmi_stream_t stream);
/* The same on a WINDOW of each source image: the crop the reference takes between the detector and the segmenter
 * (CircuitAnalyzer.crop_image_and_adjust_bboxes, circuit_analyzer.py:937-1284, called with padding = 80 at analysis_pipeline.py:177; the
 * cropped image is what segment_with_sam2 :206 then resizes), WITHOUT a cropped copy: image b is the window rects[b] = {x0, y0, w, h} of the
 * u8 [H, W, 3] image at src + b * src_image_stride BYTES (0: every window comes from one image), and exactly the pixels of the window take
 * part in the antialiased resize -- bit-identical to cvmi_sam2_transform on a contiguous copy of the window.  rects is a HOST array
 * (B x 4 ints, read during the call; it travels in the kernel arguments).  The source may be the buffer the detector's letterbox read. */
int cvmi_sam2_transform_rects(const uint8_t* src, long long src_image_stride, int H, int W, const int* rects, int B, void* dst, int R,
                              int dst_dtype, int swap_rb, cvmi_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CVMI355_H */
