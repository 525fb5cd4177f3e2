"""Drop-in for the names the reference imports from its `src/sam2_infer.py`
(`from .sam2_infer import SAM2Transforms, get_modified_sam2, device`, circuit_analyzer.py:17-21),
backed by the cvmi355 HIP kernels.  See SURVEY.md 8(b) for the exact surface the callers rely on.
"""
import os
import threading
from types import SimpleNamespace

import numpy as np
import torch

from . import _lib
from ._lib import BF16, F16, F32
from .engine import TORCH_DTYPE, require_gpu
from .sam2 import HIERA_L, Sam2Plan, Sam2Weights, SamBaseCheckpointParams, SamStateDictParams, SamSyntheticParams

# the reference picks its device at import time (sam2_infer.py:19-25); on PyTorch-ROCm "cuda" is the MI355X
device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")


def _hiera_from_yaml(path):
    """Trunk hyper-parameters from a sam2 Hydra YAML (models/configs/sam2.1_hiera_l.yaml:9-16, :89)."""
    import yaml
    with open(path) as f:
        cfg = yaml.safe_load(f)
    m = cfg["model"]
    t = m["image_encoder"]["trunk"]
    hiera = dict(embed_dim=t.get("embed_dim", 96), num_heads=t.get("num_heads", 1), stages=tuple(t.get("stages", (2, 3, 16, 3))),
                 global_att_blocks=tuple(t.get("global_att_blocks", (12, 16, 20))), window_spec=tuple(t.get("window_spec", (8, 4, 14, 7))))
    return hiera, int(m.get("image_size", 1024))


class SAM2Model:
    """What `get_modified_sam2(...)` returns: `.load_state_dict`, `.eval`, `.sam2_model.image_size`,
    `__call__(x) -> (high_res, low_res, iou)` (sam2_infer.py:220-275), plus `infer_masks`."""

    def __init__(self, hiera=HIERA_L, image_size=1024, dtype="f16", dev="cuda", use_refinement=True, refinement_kernels=(3, 5, 7, 11),
                 embedding_r=4, lora_rank=4, lora_alpha=16, dynamic_multimask_via_stability=True, attn="16"):
        """attn="fp8": the softmax(QK^T) V products of Hiera's 256-key windows and global blocks on the block-scaled fp8 MFMA (BASELINE configs[4];
        off by default: it is slower than the 16-bit product on this path -- DESIGN.md section 5 -- and costs ~3 mantissa bits on V)."""
        require_gpu()
        self.attn = attn
        self.hiera, self.image_size = hiera, image_size
        self.dtype = {"f16": F16, "fp16": F16, "f32": F32, "fp32": F32, "bf16": BF16}[dtype] if isinstance(dtype, str) else dtype
        self.dev = dev
        self.cfg = dict(use_refinement=use_refinement, refinement_kernels=tuple(refinement_kernels), embedding_r=embedding_r)
        self.lora = (lora_rank, lora_alpha)
        self.dynamic = dynamic_multimask_via_stability
        self.sam2_model = SimpleNamespace(image_size=image_size)          # circuit_analyzer.py:237-240 reads .sam2_model.image_size
        self.weights, self.params, self._pending = None, None, None
        self.stream = torch.cuda.Stream(device=dev)
        self._slot_streams = {0: self.stream}                               # plan slot -> stream (slot 1: a second plan instance on its own stream, so
        self._plans = {}                                                    # that two independent batches run side by side: CircuitPipeline)
        self._lock = threading.Lock()                                       # one analyzer is shared across sessions (app.py:134)
        self.training = False

    # -- nn.Module-like surface the reference touches
    def load_params(self, params):
        self.params, self._pending = params, None
        with torch.cuda.device(self.dev):
            self.weights = Sam2Weights(params, self.hiera, self.image_size, self.dtype, self.dev, **self.cfg)
        self._plans.clear()
        return self

    def load_state_dict(self, sd, strict=True):
        if "state_dict" in sd and not torch.is_tensor(sd["state_dict"]):
            sd = sd["state_dict"]
        return self.load_params(SamStateDictParams(sd, *self.lora))

    def eval(self):
        self.training = False
        return self

    def to(self, *_a, **_k):
        return self

    def slot_stream(self, slot):
        if slot not in self._slot_streams:
            self._slot_streams[slot] = torch.cuda.Stream(device=self.dev)
        return self._slot_streams[slot]

    def plan(self, B, prompts=0, high_res=True, points=3, slot=0):
        if self.weights is None and self._pending is not None:
            self.load_params(self._pending)           # base checkpoint only (no fine-tuned state dict followed): pack it now
        if self.weights is None:
            raise RuntimeError("SAM2 weights not loaded (call load_state_dict first)")
        key = (B, prompts, high_res, points if prompts else 0, slot)
        if key not in self._plans:
            with torch.cuda.device(self.dev):
                self._plans[key] = Sam2Plan(self.weights, B, self.slot_stream(slot), self.dynamic, prompts=prompts, points=points, high_res=high_res, attn=self.attn)
        return self._plans[key]

    def _stage_images(self, p, images):
        """Checked copy of a [B,3,R,R] tensor into the plan's NHWC input buffer.  Runs inside `torch.cuda.stream(self.stream)`; the caller's
        stream has been made a predecessor (`wait_stream`), so device inputs produced there are complete when the copy reads them."""
        lib = _lib.load()
        B = images.shape[0]
        x = images.to(self.dev)
        if x.dtype not in (torch.float32, torch.float16, torch.bfloat16) or (x.dtype == torch.float16) != (self.dtype == F16) and x.dtype != torch.float32:
            x = x.float()                                     # (a 16-bit input of the OTHER 16-bit type goes through f32)
        src_dt = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}[x.dtype]
        sp = self.stream.cuda_stream
        if x.permute(0, 2, 3, 1).is_contiguous():            # already channels-last memory (SAM2Transforms output)
            _lib.check(lib.cvmi_cast(x.data_ptr(), 3, src_dt, p.x_in.t.data_ptr(), 3, self.dtype, B * self.image_size ** 2, 3, sp), "cast")
        else:
            x = x.contiguous()
            _lib.check(lib.cvmi_nchw_to_nhwc(x.data_ptr(), src_dt, p.x_in.t.data_ptr(), self.dtype, 3, B, 3, self.image_size, self.image_size, sp), "nchw_to_nhwc")
        return x                                              # keep alive until the stream has consumed it

    def _run(self, p, images, outs, before=None):
        """Stage -> replay the plan -> copy `outs` = [(plan tensor, fresh tensor)] -- all on the model's stream, which first waits for the
        caller's current stream and is synchronised before returning (the reference's call is synchronous).  The output copies are
        part of the stream's order, so the next caller's replay (another session thread, app.py:134) cannot overwrite the plan's
        buffers under them."""
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            if before is not None:
                before()
            keep = self._stage_images(p, images)
            p.plan.run()
            for src, dst in outs:
                dst.copy_(src, non_blocking=True)
        self.stream.synchronize()
        del keep

    def _check_images(self, images):
        if not (torch.is_tensor(images) and images.dim() == 4 and images.shape[1] == 3 and images.shape[2] == images.shape[3] == self.image_size):
            raise ValueError(f"expected a [B,3,{self.image_size},{self.image_size}] tensor")

    def __call__(self, images, points=None, point_labels=None, masks_prompt=None, multimask_output=False):
        self._check_images(images)
        if multimask_output:
            raise NotImplementedError("the wrapper always runs multimask_output=False (sam2_infer.py:257)")
        with self._lock, torch.cuda.device(self.dev):
            p = self.plan(images.shape[0])
            out = [torch.empty_like(t) for t in (p.high_res, p.low_res, p.iou)]       # the caller's tensors (its allocator pool, not the model stream's)
            self._run(p, images, list(zip((p.high_res, p.low_res, p.iou), out)))
            return tuple(out)

    forward = __call__

    def infer_masks(self, images, boxes=None, return_high_res=True, points=None, point_labels=None):
        """Batched segmentation entry point (north_star).  images [B,3,R,R].
        No prompt: exactly `SAM2ImageWrapper.forward` (learned prompts) -> (high_res [B,1,R,R], low_res [B,1,R/4,R/4], iou [B,1]).
        boxes [B,P,4] (xyxy in the R x R input pixel space, e.g. detector boxes scaled by R / original size) and / or
        points [B,P,K,2] (x, y) with point_labels [B,P,K] (1 foreground click, 0 background click, -1 "not a point" padding token,
        upstream's filler for ragged click lists -- still a token the decoder sees): upstream
        SAM2ImagePredictor prompting on the same weights -- a box enters as two corner points labelled 2 / 3, clicks follow it,
        one padding point closes the list -- one mask per prompt (multimask_output=False with the stability fallback) ->
        (high_res logits [B,P,R,R] or None, low_res logits [B,P,R/4,R/4] (unclamped), iou [B,P])."""
        if boxes is None and points is None:
            return self(images)
        self._check_images(images)
        B = images.shape[0]
        bx = pts = None
        if boxes is not None:
            bx = torch.as_tensor(boxes, dtype=torch.float32)
            if bx.dim() != 3 or bx.shape[0] != B or bx.shape[2] != 4 or bx.shape[1] == 0:
                raise ValueError(f"boxes must be [B={B}, P>0, 4] xyxy, got {tuple(bx.shape)}")
        if points is not None:
            pts = torch.as_tensor(points, dtype=torch.float32)
            if point_labels is None:
                raise ValueError("points need point_labels")
            lab = torch.as_tensor(point_labels).to(torch.int32)
            if pts.dim() != 4 or pts.shape[0] != B or pts.shape[3] != 2 or pts.shape[1] == 0 or pts.shape[2] == 0 or tuple(lab.shape) != tuple(pts.shape[:3]):
                raise ValueError(f"points must be [B={B}, P>0, K>0, 2] with point_labels [B, P, K], got {tuple(pts.shape)} / {tuple(lab.shape)}")
            if bx is not None and bx.shape[1] != pts.shape[1]:
                raise ValueError("boxes and points disagree on the number of prompts per image")
            if int(lab.min()) < -1 or int(lab.max()) > 1:
                raise ValueError("point_labels must be -1 (padding token), 0 (background) or 1 (foreground)")
        P = (bx if bx is not None else pts).shape[1]
        nb, nk = (2 if bx is not None else 0), (pts.shape[2] if pts is not None else 0)
        K = nb + nk + 1                                            # + the closing padding point
        R, f0 = self.image_size, self.image_size // 4
        with self._lock, torch.cuda.device(self.dev):
            p = self.plan(B, prompts=P, high_res=return_high_res, points=K)
            coords = torch.zeros(B * P, K, 2, dtype=torch.float32)
            labels = torch.full((B * P, K), -1, dtype=torch.int32)
            if bx is not None:
                coords[:, :2] = bx.reshape(B * P, 2, 2)
                labels[:, 0], labels[:, 1] = 2, 3
            if pts is not None:
                coords[:, nb:nb + nk] = pts.reshape(B * P, nk, 2)
                labels[:, nb:nb + nk] = lab.reshape(B * P, nk)
            def prompts_in():                                          # on the model's stream, in front of the replay that reads them
                p.coords.copy_(coords, non_blocking=False)
                p.labels.copy_(labels, non_blocking=False)
            lo, iou = torch.empty(B, P, f0, f0, dtype=torch.float32, device=self.dev), torch.empty(B, P, dtype=torch.float32, device=self.dev)
            hi = torch.empty(B, P, R, R, dtype=torch.float32, device=self.dev) if return_high_res else None
            outs = [(p.low_res.view(B, P, f0, f0), lo), (p.iou.view(B, P), iou)] + ([(p.high_res.view(B, P, R, R), hi)] if return_high_res else [])
            self._run(p, images, outs, before=prompts_in)
            return hi, lo, iou


def get_modified_sam2(model_cfg_path, checkpoint_path, device="cuda", use_high_res_features=True, use_peft=True, lora_rank=12,
                      lora_alpha=16, lora_dropout=0.2, lora_target_modules=None, use_wrapper=True, trainable_embedding_r=4,
                      use_refinement_layer=False, refinement_kernels=(3, 5, 7, 11), kernel_channels=4, dtype="f16", attn="16", **_loss_and_optimizer_kwargs):
    """Same signature as sam2_infer.py:277-305 (loss / optimizer kwargs accepted and ignored).  `checkpoint_path`
    may be 'synthetic[:seed]' for seeded random weights; a real base checkpoint ({'model': state_dict}) is accepted
    but every tensor is expected to come from the fine-tuned state dict loaded afterwards (circuit_analyzer.py:227-233)."""
    if not use_wrapper or not use_high_res_features:
        raise NotImplementedError("only the reference configuration (use_wrapper=True, use_high_res_features=True) is built")
    hiera, image_size = HIERA_L, 1024
    if isinstance(model_cfg_path, str) and os.path.exists(model_cfg_path):
        hiera, image_size = _hiera_from_yaml(model_cfg_path)
    elif isinstance(model_cfg_path, str) and os.path.exists(model_cfg_path.lstrip("/")):
        hiera, image_size = _hiera_from_yaml(model_cfg_path.lstrip("/"))          # the reference prepends "/" for Hydra (circuit_analyzer.py:204)
    model = SAM2Model(hiera, image_size, dtype=dtype, dev=str(device), use_refinement=use_refinement_layer, refinement_kernels=refinement_kernels,
                      embedding_r=trainable_embedding_r, lora_rank=lora_rank, lora_alpha=lora_alpha, attn=attn)
    if isinstance(checkpoint_path, str) and checkpoint_path.startswith("synthetic"):
        seed = int(checkpoint_path.split(":")[1]) if ":" in checkpoint_path else 0
        targets = lora_target_modules if (use_peft and lora_target_modules is not None) else ()
        model.load_params(SamSyntheticParams(seed=seed, lora_targets=targets, r=lora_rank, alpha=lora_alpha))
    elif isinstance(checkpoint_path, str) and os.path.exists(checkpoint_path):
        # The reference always loads the fine-tuned state dict next (circuit_analyzer.py:227-233), which replaces every tensor:
        # packing the base weights is deferred until a forward pass actually needs them.
        model._pending = SamBaseCheckpointParams(torch.load(checkpoint_path, map_location="cpu", weights_only=True))
    return model


class SAM2Transforms:
    """sam2_infer.py:29-128.  `__call__` returns an f32 [3,R,R] device tensor (channels-last memory, so the model
    consumes it without a layout pass); `postprocess_masks` with the reference's settings (areas = 0) is the
    bilinear resize to the original size (:127)."""

    def __init__(self, resolution, mask_threshold, max_hole_area=0.0, max_sprinkle_area=0.0):
        self.resolution, self.mask_threshold = resolution, mask_threshold
        self.max_hole_area, self.max_sprinkle_area = max_hole_area, max_sprinkle_area
        if max_hole_area > 0 or max_sprinkle_area > 0:
            raise NotImplementedError("hole / sprinkle filtering (connected components) is disabled in the reference (circuit_analyzer.py:245-250)")
        self.mean, self.std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]

    # Every method below behaves like a torch op: kernels are enqueued on the caller's CURRENT stream and nothing synchronises the device
    # (the first use of a result on another stream, or `.cpu()`, orders itself the usual way).  One object is shared by all session
    # threads (app.py:134): there is no shared state to protect.
    @staticmethod
    def _check_u8(x):
        img = np.ascontiguousarray(np.asarray(x))
        if img.ndim != 3 or img.shape[2] != 3 or img.dtype != np.uint8:
            raise TypeError("SAM2Transforms expects an RGB uint8 image (PIL or HxWx3 array)")
        return img

    def __call__(self, x):
        return self.forward_batch([x])[0]

    def forward_batch(self, img_list, swap_rb=False, out=None, out_dtype=F32):
        """sam2_infer.py:53-56.  -> f32 [B,3,R,R] (channels-last memory).  Equally sized images go through ONE pinned staging buffer,
        ONE H2D copy and ONE launch; ragged sizes take one launch each.  swap_rb: read the channels reversed (the caller's BGR2RGB,
        circuit_analyzer.py:343) instead of a host pass over every image.  out / out_dtype: write into an existing [B,R,R,3] device
        tensor of that CVMI dtype (the segmenter's input buffer) instead of a fresh f32 one."""
        require_gpu()
        lib = _lib.load()
        R = self.resolution
        imgs = [self._check_u8(im) for im in img_list]
        B = len(imgs)
        res = torch.empty(B, R, R, 3, dtype=torch.float32, device="cuda") if out is None else out
        sp = torch.cuda.current_stream().cuda_stream
        if B > 1 and all(im.shape == imgs[0].shape for im in imgs):
            h, w = imgs[0].shape[:2]
            host = torch.empty(B, h, w, 3, dtype=torch.uint8, pin_memory=True)
            hv = host.numpy()
            for b, im in enumerate(imgs):
                hv[b] = im
            src = host.to("cuda", non_blocking=True)
            _lib.check(lib.cvmi_sam2_transform_batch(src.data_ptr(), B, h, w, res.data_ptr(), R, out_dtype, 1 if swap_rb else 0, sp), "sam2_transform")
        else:
            for b, im in enumerate(imgs):
                src = torch.from_numpy(im).cuda()
                _lib.check(lib.cvmi_sam2_transform_batch(src.data_ptr(), 1, im.shape[0], im.shape[1], res[b].data_ptr(), R, out_dtype, 1 if swap_rb else 0, sp),
                           "sam2_transform")
        return res.permute(0, 3, 1, 2) if out is None else res

    def forward_windows(self, src, windows, swap_rb=False, out=None, out_dtype=F32):
        """`forward_batch` on a WINDOW of each image of a u8 [B, H, W, 3] DEVICE tensor (e.g. the block the detector's letterbox read:
        `PendingDetections.src`): windows = [(x0, y0, x1, y1) | None] per image, None = the whole image.  What the reference does with a host
        crop + a new transform (circuit_analyzer.py:1254 `image[y0:y1, x0:x1]`, then :347) is here a source rectangle of the resize kernel: no
        cropped copy, no second H2D, bit-identical to transforming a contiguous copy of the window."""
        require_gpu()
        lib = _lib.load()
        if not (torch.is_tensor(src) and src.is_cuda and src.dtype == torch.uint8 and src.dim() == 4 and src.shape[3] == 3 and src.is_contiguous()):
            raise TypeError("forward_windows expects a contiguous uint8 [B,H,W,3] device tensor")
        B, H, W = src.shape[:3]
        if len(windows) != B:
            raise ValueError("one window (or None) per image")
        R = self.resolution
        rects = np.empty((B, 4), dtype=np.int32)
        for b, wnd in enumerate(windows):
            x0, y0, x1, y1 = (0, 0, W, H) if wnd is None else wnd
            rects[b] = (x0, y0, x1 - x0, y1 - y0)
        res = torch.empty(B, R, R, 3, dtype=torch.float32, device=src.device) if out is None else out
        _lib.check(lib.cvmi_sam2_transform_rects(src.data_ptr(), H * W * 3, H, W, rects.ctypes.data, B, res.data_ptr(), R, out_dtype, 1 if swap_rb else 0,
                                                 torch.cuda.current_stream().cuda_stream), "sam2_transform_rects")
        return res.permute(0, 3, 1, 2) if out is None else res

    def postprocess_to_masks_sized(self, masks, sizes):
        """postprocess_to_mask_async for planes that return to DIFFERENT sizes (every image its own crop window): masks f32 [N,1,h,w] on the
        device, sizes = [(H, W)] per plane -> ([u8 [H_n, W_n] views of ONE packed buffer], extent int32 [N,4]); one launch, nothing copied to
        the host."""
        require_gpu()
        lib = _lib.load()
        m = masks.float().contiguous()
        N, h, w = m.shape[0] * m.shape[1], m.shape[-2], m.shape[-1]
        if len(sizes) != N:
            raise ValueError("one (H, W) per mask plane")
        sz = np.asarray(sizes, dtype=np.int32).reshape(N, 2)
        offs = np.concatenate(([0], np.cumsum(sz[:, 0].astype(np.int64) * sz[:, 1])))
        u8 = torch.empty(int(offs[-1]), dtype=torch.uint8, device=m.device)
        ext = torch.empty(N, 4, dtype=torch.int32, device=m.device)
        _lib.check(lib.cvmi_mask_postprocess_sizes(m.data_ptr(), N, h, w, sz.ctypes.data, float(self.mask_threshold), u8.data_ptr(), ext.data_ptr(),
                                                   torch.cuda.current_stream().cuda_stream), "mask_postprocess_sizes")
        return [u8[int(offs[n]):int(offs[n + 1])].view(int(sz[n, 0]), int(sz[n, 1])) for n in range(N)], ext

    def mask_extent(self, mask_u8):
        """Bounding boxes of binary masks [N,H,W] / [B,C,H,W] (u8 on the device): list of (x0, y0, x1, y1) or None per
        plane, with the reference's convention (circuit_analyzer.py:364-370: boundingRect of the external contours)."""
        require_gpu()
        lib = _lib.load()
        m = mask_u8.contiguous()
        H, W = m.shape[-2:]
        N = m.numel() // (H * W)
        ext = torch.empty(N, 4, dtype=torch.int32, device=m.device)
        _lib.check(lib.cvmi_mask_extent(m.data_ptr(), N, H, W, ext.data_ptr(), torch.cuda.current_stream().cuda_stream), "mask_extent")
        return self.extents_to_boxes(ext)

    @staticmethod
    def extents_to_boxes(ext):
        """int32 [N,4] {min x, min y, max x, max y} (or {W, H, -1, -1}) -> the reference's sam_extent_bbox tuples / None."""
        return [None if x1 < 0 else (x0, y0, x1 + 1, y1 + 1) for x0, y0, x1, y1 in ext.cpu().tolist()]

    # sam2_infer.py:58-86 -- what a caller needs to feed detector boxes (original pixels) to `infer_masks`
    def transform_coords(self, coords, normalize=False, orig_hw=None):
        """[..., 2] (x, y).  normalize=True: absolute pixels of an orig_hw = (h, w) image -> [0, 1] first; then x resolution."""
        coords = torch.as_tensor(coords, dtype=torch.float32)
        if normalize:
            assert orig_hw is not None
            h, w = orig_hw
            coords = coords.clone()
            coords[..., 0] = coords[..., 0] / w
            coords[..., 1] = coords[..., 1] / h
        return coords * self.resolution

    def transform_boxes(self, boxes, normalize=False, orig_hw=None):
        """[N, 4] xyxy -> [N, 2, 2] corner points in the resolution x resolution input space (as the reference returns them;
        `.reshape(-1, 4)` gives the xyxy rows `infer_masks(images, boxes)` takes)."""
        return self.transform_coords(torch.as_tensor(boxes, dtype=torch.float32).reshape(-1, 2, 2), normalize, orig_hw)

    def postprocess_masks(self, masks, orig_hw, return_u8=False):
        require_gpu()
        lib = _lib.load()
        m = masks.float().contiguous()
        if not m.is_cuda:
            m = m.cuda()
        B, C, h, w = m.shape
        H, W = int(orig_hw[0]), int(orig_hw[1])
        out = torch.empty(B, C, H, W, dtype=torch.float32, device=m.device)
        u8 = torch.empty(B, C, H, W, dtype=torch.uint8, device=m.device) if return_u8 else None
        _lib.check(lib.cvmi_bilinear_f32(m.data_ptr(), B * C, h, w, out.data_ptr(), H, W, u8.data_ptr() if return_u8 else None,
                                         float(self.mask_threshold), torch.cuda.current_stream().cuda_stream), "bilinear")
        return (out, u8) if return_u8 else out

    def postprocess_to_mask_async(self, masks, orig_hw):
        """The device half of postprocess_to_mask: -> (mask_u8 [B,C,H,W], extent int32 [B*C,4]) enqueued on the current stream, nothing
        copied to the host (extents_to_boxes(extent) does that, later, for a whole batch at once)."""
        require_gpu()
        lib = _lib.load()
        m = masks.float().contiguous()
        if not m.is_cuda:
            m = m.cuda()
        B, C, h, w = m.shape
        H, W = int(orig_hw[0]), int(orig_hw[1])
        u8 = torch.empty(B, C, H, W, dtype=torch.uint8, device=m.device)
        ext = torch.empty(B * C, 4, dtype=torch.int32, device=m.device)
        _lib.check(lib.cvmi_mask_postprocess(m.data_ptr(), B * C, h, w, H, W, float(self.mask_threshold), u8.data_ptr(), ext.data_ptr(),
                                             torch.cuda.current_stream().cuda_stream), "mask_postprocess")
        return u8, ext

    def postprocess_to_mask(self, masks, orig_hw):
        """circuit_analyzer.py:354-370 in one pass on the device: bilinear resize to orig_hw -> `> mask_threshold` -> u8 {0, 255}
        -> bounding rectangle.  The f32 [B,C,H,W] map is never written; only the u8 masks and four ints per mask exist afterwards.
        Returns (mask_u8 [B,C,H,W] on the device, [(x0, y0, x1, y1) | None] per mask, the reference's `sam_extent_bbox`)."""
        u8, ext = self.postprocess_to_mask_async(masks, orig_hw)
        return u8, self.extents_to_boxes(ext)
