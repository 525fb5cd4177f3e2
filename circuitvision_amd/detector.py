"""Drop-in for `ultralytics.YOLO` as the reference uses it:

    self.yolo = YOLO(yolo_path)                      circuit_analyzer.py:45
    self.yolo.model.names -> {int: str}              circuit_analyzer.py:111, 121, 2160, 2264
    r = self.yolo.predict(image, verbose=True)[0]    circuit_analyzer.py:268
    r.boxes.cls / .conf / .xyxy (.cpu().numpy().tolist()),  r.names[int]     :270-273

Everything between the uint8 HxWx3 image and the [n,6] detections runs in HIP kernels: letterbox,
the YOLO11 network, Detect decode and NMS (one captured HIP graph per letterboxed shape).
"""
import math
import threading
from types import SimpleNamespace

import numpy as np
import torch

from . import _lib
from ._lib import F16, F32
from .engine import require_gpu
from .yolo11 import StateDictParams, SyntheticParams, Yolo11Plan, Yolo11Weights


def letterbox_geometry(h, w, new_shape=640, stride=32, auto=True):
    """ultralytics LetterBox(new_shape, auto=True, stride=32) geometry."""
    r = min(new_shape / h, new_shape / w)
    nw, nh = int(round(w * r)), int(round(h * r))
    dw, dh = new_shape - nw, new_shape - nh
    if auto:
        dw, dh = dw % stride, dh % stride
    dw /= 2
    dh /= 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return nw, nh, top, bottom, left, right


class Boxes:
    def __init__(self, det):
        self.data = det                      # [n, 6] x1,y1,x2,y2,conf,cls (host tensor: one D2H copy per batch brought it over)

    xyxy = property(lambda s: s.data[:, :4])
    conf = property(lambda s: s.data[:, 4])
    cls = property(lambda s: s.data[:, 5])

    def __len__(self):
        return self.data.shape[0]


class Results:
    def __init__(self, det, names, orig_shape, anchor_idx=None):
        self.boxes = Boxes(det)
        self.names = names
        self.orig_shape = orig_shape
        self.anchor_idx = anchor_idx

    def __len__(self):
        return len(self.boxes)


def parse_spec(path):
    """'synthetic:<scale>:<nc>[:seed]' -> (scale, nc, seed) or None."""
    if isinstance(path, str) and path.startswith("synthetic:"):
        parts = path.split(":")
        return parts[1], int(parts[2]), int(parts[3]) if len(parts) > 3 else 0
    return None


class YOLO:
    """`YOLO(path)`; path is an ultralytics `.pt` (read without ultralytics, see checkpoints.py), a converted
    {'state_dict', 'names', 'scale'} file, or 'synthetic:<scale>:<nc>[:seed]' for seeded random weights."""

    MAX_ANCHORS = 65536          # cvmi_yolo_nms: one workgroup per image, keys in LDS (<= 16384 anchors) or in its workspace

    def __init__(self, path, dtype="f16", device="cuda", imgsz=None, keep_scores=False, graph_lanes=None):
        """keep_scores=True also materialises ultralytics' full [B, 4+nc, A] prediction tensor (tests / debugging).
        imgsz=None: the checkpoint's train_args['imgsz'] (what ultralytics' predict() inherits), else 640.
        graph_lanes: side lanes of the captured graph (None: the default, the Detect heads beside the neck -- lowest latency of ONE call);
        0 = one linear chain: what a batch pipeline wants, whose detector graph runs BESIDE the segmenter's on another stream (a graph with
        internal branches serialises the other streams it shares the device with: DESIGN.md section 5.00)."""
        require_gpu()
        self.keep_scores = keep_scores
        self.graph_lanes = graph_lanes
        self.dtype = {"f16": F16, "fp16": F16, "f32": F32, "fp32": F32}[dtype] if isinstance(dtype, str) else dtype
        self.device = device
        ck_imgsz = None
        spec = parse_spec(path)
        if spec is not None:
            scale, nc, seed = spec
            params = SyntheticParams(seed=seed, nc=nc)
            names = {i: f"class{i}" for i in range(nc)}
        else:
            from .checkpoints import load_ultralytics_pt
            ck = load_ultralytics_pt(path)      # ultralytics .pt (restricted unpickler) or a converted {'state_dict','names','scale'} file
            scale, names = ck["scale"], {int(k): v for k, v in ck["names"].items()}
            nc = len(names)
            params = StateDictParams(ck["state_dict"])
            ck_imgsz = ck.get("imgsz")
        self.imgsz = self._check_imgsz(imgsz if imgsz is not None else (ck_imgsz or 640))
        self.params = params
        self.weights = Yolo11Weights(scale, nc, params, self.dtype, device)
        self.model = SimpleNamespace(names=names, nc=nc, scale=scale, stride=32)
        self.names = names
        self.stream = torch.cuda.Stream(device=device)
        self._plans, self._staging = {}, {}
        self._lock = threading.Lock()      # one analyzer is shared by all Streamlit sessions (app.py:134)

    @classmethod
    def from_weights(cls, weights, names, dtype="f16", device="cuda", imgsz=640, keep_scores=False, graph_lanes=None):
        """A detector over already packed `Yolo11Weights` (e.g. the replica a rank received by broadcast: no checkpoint read, no packing)."""
        require_gpu()
        self = cls.__new__(cls)
        self.keep_scores = keep_scores
        self.graph_lanes = graph_lanes
        self.dtype = {"f16": F16, "fp16": F16, "f32": F32, "fp32": F32}[dtype] if isinstance(dtype, str) else dtype
        self.device = device
        self.imgsz = cls._check_imgsz(imgsz)
        self.params, self.weights = getattr(weights, "params", None), weights
        self.names = {int(k): v for k, v in names.items()}
        self.model = SimpleNamespace(names=self.names, nc=len(self.names), scale=weights.scale, stride=32)
        self.stream = torch.cuda.Stream(device=device)
        self._plans, self._staging = {}, {}
        self._lock = threading.Lock()
        return self

    @classmethod
    def _check_imgsz(cls, imgsz):
        imgsz = int(max(imgsz)) if isinstance(imgsz, (list, tuple)) else int(imgsz)
        if imgsz < 32:
            raise ValueError(f"imgsz={imgsz}: must be >= 32")
        imgsz = -(-imgsz // 32) * 32                       # ultralytics check_imgsz: round up to a stride multiple
        if imgsz * imgsz * 21 // 1024 > cls.MAX_ANCHORS:   # anchors of a square input: (1/64 + 1/256 + 1/1024) per pixel
            raise ValueError(f"imgsz={imgsz} gives more than {cls.MAX_ANCHORS} anchors, beyond what cvmi_yolo_nms handles")
        return imgsz

    # ---- plans ---------------------------------------------------------------------------------
    def plan(self, B, H, W, conf=0.25, iou=0.7, max_det=300):
        key = (B, H, W, conf, iou, max_det)
        if key not in self._plans:
            with torch.cuda.device(self.device):
                self._plans[key] = Yolo11Plan(self.weights, B, H, W, self.stream, conf, iou, max_det, keep_scores=self.keep_scores, lanes=self.graph_lanes)
        return self._plans[key]

    # ---- reference entry point -----------------------------------------------------------------
    def predict(self, image, verbose=True, conf=0.25, iou=0.7, max_det=300, imgsz=None, **_):
        """image: uint8 HxWx3 numpy array (or a list of same-shaped ones).  Returns [Results] (host tensors: the reference reads them
        through `.cpu().numpy().tolist()`, circuit_analyzer.py:270-273)."""
        out = self.predict_async(image, conf=conf, iou=iou, max_det=max_det, imgsz=imgsz).result()
        if verbose:
            H, W = out[0].letterboxed_shape if out else (0, 0)
            print(f"cvmi355 YOLO11{self.model.scale}: {H}x{W} {', '.join(str(len(r)) + ' boxes' for r in out)}")
        return out

    __call__ = predict

    def predict_async(self, image, conf=0.25, iou=0.7, max_det=300, imgsz=None):
        """Batch-shaped, stream-ordered form of `predict`: ONE pinned staging buffer and ONE H2D copy for the batch, ONE letterbox launch,
        the captured graph, ONE D2H copy of (detections, anchor indices, counts) -- all enqueued on the detector's stream -- and nothing
        waits.  Returns a handle; `.result()` waits for the copy and builds the `Results` on the host (scale_boxes + clip there: ultralytics'
        own CPU arithmetic on <= 300 boxes).  The caller may do other work -- or enqueue the segmenter -- in between."""
        imgsz = self.imgsz if imgsz is None else self._check_imgsz(imgsz)
        images = image if isinstance(image, (list, tuple)) else [image]
        if not images:
            raise ValueError("predict needs at least one image")
        for im in images:
            if not (isinstance(im, np.ndarray) and im.ndim == 3 and im.shape[2] == 3 and im.dtype == np.uint8):
                raise TypeError("predict expects uint8 HxWx3 numpy images")
        h0, w0 = images[0].shape[:2]
        if any(im.shape[:2] != (h0, w0) for im in images):
            raise ValueError("a batch must share one image size")
        nw, nh, top, bottom, left, right = letterbox_geometry(h0, w0, imgsz)
        H, W = nh + top + bottom, nw + left + right
        lib = _lib.load()
        B = len(images)
        with self._lock, torch.cuda.device(self.device):
            p = self.plan(B, H, W, conf, iou, max_det)
            st = self._staging.get((B, h0, w0))
            if st is None:                                              # pinned once per (batch, source size); reused by later calls
                st = self._staging[(B, h0, w0)] = [torch.empty(B, h0, w0, 3, dtype=torch.uint8, pin_memory=True), None]
            if st[1] is not None:
                st[1].synchronize()                                     # the previous call's H2D copy has left this buffer
            host = st[0].numpy()
            for b, im in enumerate(images):
                host[b] = im                                            # the one host pass over the pixels
            det_h = torch.empty(B, p.max_det, 6, dtype=torch.float32, pin_memory=True)
            idx_h = torch.empty(B, p.max_det, dtype=torch.int32, pin_memory=True)
            cnt_h = torch.empty(B, dtype=torch.int32, pin_memory=True)
            sp = self.stream.cuda_stream
            with torch.cuda.stream(self.stream):
                src = st[0].to(self.device, non_blocking=True)
                st[1] = torch.cuda.Event()
                st[1].record(self.stream)
                _lib.check(lib.cvmi_letterbox_batch(src.data_ptr(), B, h0, w0, p.x_in.t.data_ptr(), p.x_in.t[0].numel(), H, W, nh, nw, top, left,
                                                    self.dtype, 1, sp), "letterbox")
                p.plan.run()
                det_h.copy_(p.det, non_blocking=True)
                idx_h.copy_(p.det_idx, non_blocking=True)
                cnt_h.copy_(p.det_count, non_blocking=True)
                done = torch.cuda.Event()
                done.record(self.stream)
        return PendingDetections(done, det_h, idx_h, cnt_h, (H, W), (h0, w0), self.names, src)

    def predict_chunks_async(self, images, chunk, conf=0.25, iou=0.7, max_det=300, imgsz=None):
        """A batch pipeline's form of `predict_async`: ONE pinned staging pass and ONE H2D copy for all equally sized `images`, then one
        letterbox launch + graph replay + D2H of the detections per `chunk` images -- so the host can take chunk 0's boxes (and start what
        depends on them: the crop window, the segmenter) while the GPU still runs the later chunks.  Returns one handle per chunk; handle.src
        is that chunk's u8 [b, h, w, 3] DEVICE image block (valid once `.result()` has returned: the segmenter's transform reads its crop
        windows straight out of it, no second H2D)."""
        imgsz = self.imgsz if imgsz is None else self._check_imgsz(imgsz)
        if not images:
            raise ValueError("predict needs at least one image")
        for im in images:
            if not (isinstance(im, np.ndarray) and im.ndim == 3 and im.shape[2] == 3 and im.dtype == np.uint8):
                raise TypeError("predict expects uint8 HxWx3 numpy images")
        h0, w0 = images[0].shape[:2]
        if any(im.shape[:2] != (h0, w0) for im in images):
            raise ValueError("a batch must share one image size")
        nw, nh, top, bottom, left, right = letterbox_geometry(h0, w0, imgsz)
        H, W = nh + top + bottom, nw + left + right
        lib = _lib.load()
        n, chunk = len(images), max(1, int(chunk))
        out = []
        with self._lock, torch.cuda.device(self.device):
            st = self._staging.get((n, h0, w0))
            if st is None:
                st = self._staging[(n, h0, w0)] = [torch.empty(n, h0, w0, 3, dtype=torch.uint8, pin_memory=True), None]
            if st[1] is not None:
                st[1].synchronize()
            host = st[0].numpy()
            for b, im in enumerate(images):
                host[b] = im
            sp = self.stream.cuda_stream
            with torch.cuda.stream(self.stream):
                src = st[0].to(self.device, non_blocking=True)
                st[1] = torch.cuda.Event()
                st[1].record(self.stream)
                for c0 in range(0, n, chunk):
                    B = min(chunk, n - c0)
                    p = self.plan(B, H, W, conf, iou, max_det)
                    det_h = torch.empty(B, p.max_det, 6, dtype=torch.float32, pin_memory=True)
                    idx_h = torch.empty(B, p.max_det, dtype=torch.int32, pin_memory=True)
                    cnt_h = torch.empty(B, dtype=torch.int32, pin_memory=True)
                    _lib.check(lib.cvmi_letterbox_batch(src[c0].data_ptr(), B, h0, w0, p.x_in.t.data_ptr(), p.x_in.t[0].numel(), H, W, nh, nw, top, left,
                                                        self.dtype, 1, sp), "letterbox")
                    p.plan.run()
                    det_h.copy_(p.det, non_blocking=True)
                    idx_h.copy_(p.det_idx, non_blocking=True)
                    cnt_h.copy_(p.det_count, non_blocking=True)
                    done = torch.cuda.Event()
                    done.record(self.stream)
                    out.append(PendingDetections(done, det_h, idx_h, cnt_h, (H, W), (h0, w0), self.names, src, src=src[c0:c0 + B]))
        return out


class PendingDetections:
    """Detections in flight (YOLO.predict_async).  `.result()` -> [Results], one per image."""

    def __init__(self, done, det, idx, cnt, lb_shape, orig_shape, names, keep, src=None):
        self.done, self.det, self.idx, self.cnt, self.lb_shape, self.orig_shape, self.names, self._keep = done, det, idx, cnt, lb_shape, orig_shape, names, keep
        self.src = src                       # (predict_chunks_async) the chunk's u8 device images, for whoever crops from them
        self._out = None

    def result(self):
        if self._out is None:
            self.done.synchronize()
            self._keep = None
            out = []
            for b, n in enumerate(self.cnt.tolist()):
                det = self.det[b, :n].clone()
                det[:, :4] = scale_boxes(self.lb_shape, det[:, :4], self.orig_shape)
                r = Results(det, self.names, self.orig_shape, self.idx[b, :n].clone())
                r.letterboxed_shape = self.lb_shape
                out.append(r)
            self._out = out
        return self._out


def scale_boxes(img1_shape, boxes, img0_shape):
    """ultralytics ops.scale_boxes + clip_boxes (fp32, same op order as the CPU path)."""
    gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
    pad_x = round((img1_shape[1] - img0_shape[1] * gain) / 2 - 0.1)
    pad_y = round((img1_shape[0] - img0_shape[0] * gain) / 2 - 0.1)
    boxes = boxes.clone()
    boxes[..., 0] -= pad_x
    boxes[..., 1] -= pad_y
    boxes[..., 2] -= pad_x
    boxes[..., 3] -= pad_y
    boxes[..., :4] /= gain
    boxes[..., 0].clamp_(0, img0_shape[1])
    boxes[..., 1].clamp_(0, img0_shape[0])
    boxes[..., 2].clamp_(0, img0_shape[1])
    boxes[..., 3].clamp_(0, img0_shape[0])
    return boxes


# ---- second-stage NMS (reference-owned, analysis_pipeline.py:106 / utils.py:346-361) ---------------
def calculate_iou(b1, b2):
    iw = max(min(b1["xmax"], b2["xmax"]) - max(b1["xmin"], b2["xmin"]), 0)
    ih = max(min(b1["ymax"], b2["ymax"]) - max(b1["ymin"], b2["ymin"]), 0)
    inter = iw * ih
    union = ((b1["xmax"] - b1["xmin"]) * (b1["ymax"] - b1["ymin"]) +
             (b2["xmax"] - b2["xmin"]) * (b2["ymax"] - b2["ymin"]) - inter)
    return inter / union if union > 0 else 0.0


def non_max_suppression_by_confidence(bboxes, iou_threshold=0.5):
    """Host-side mirror of utils.py:346-361 (<= 300 boxes; stays on the CPU in the reference too -- the caller, analysis_pipeline.py:106, is
    untouched and keeps using its own).  Same greedy pass -- most confident first (a STABLE descending sort: equal confidences keep their list
    order), a box survives iff its IoU with every kept box is < iou_threshold -- with the inner "filter the rest" step over numpy float64 arrays
    (the reference's Python loop is O(n^2) dict look-ups: 20 ms for the 200 boxes of a busy schematic, 1.7 ms for 60).  Integer pixel boxes are
    exact in float64 and the operations are the reference's in the reference's order (`calculate_iou` above), so the kept list is identical --
    pinned by tests/golden/nms_stage2.json, incl. ties, IoU == threshold and zero-area boxes."""
    n = len(bboxes)
    if n == 0:
        return []
    conf = np.array([b["confidence"] for b in bboxes], dtype=np.float64)
    order = np.argsort(-conf, kind="stable")
    xy = np.array([[b["xmin"], b["ymin"], b["xmax"], b["ymax"]] for b in bboxes], dtype=np.float64)[order]
    area = (xy[:, 2] - xy[:, 0]) * (xy[:, 3] - xy[:, 1])
    alive = np.ones(n, dtype=bool)
    kept = []
    for i in range(n):
        if not alive[i]:
            continue
        kept.append(bboxes[int(order[i])])
        r = np.flatnonzero(alive[i + 1:]) + i + 1
        if r.size == 0:
            break
        iw = np.maximum(np.minimum(xy[i, 2], xy[r, 2]) - np.maximum(xy[i, 0], xy[r, 0]), 0)
        ih = np.maximum(np.minimum(xy[i, 3], xy[r, 3]) - np.maximum(xy[i, 1], xy[r, 1]), 0)
        inter = iw * ih
        union = (area[i] + area[r]) - inter
        iou = np.divide(inter, union, out=np.zeros_like(inter), where=union > 0)
        alive[r[~(iou < iou_threshold)]] = False
    return kept
