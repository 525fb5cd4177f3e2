"""Batched detector -> segmenter pipeline over a shard of circuit images (BASELINE configs[3]).

What it restates, batched and on the device, is the chain the reference runs per uploaded image:

    run_initial_detection              /root/reference/src/analysis_pipeline.py:97-115
        CircuitAnalyzer.bboxes         src/circuit_analyzer.py:267-287   (YOLO.predict -> dicts, python round(), persistent uid)
        non_max_suppression_by_confidence(iou 0.6)                       src/utils.py:346-361
    run_segmentation_and_cropping      src/analysis_pipeline.py:168-225
        crop_image_and_adjust_bboxes   src/circuit_analyzer.py:937-1284  (`crop=True`: this package's mirror, crop.py, padding 80 as at
                                                                          analysis_pipeline.py:177 -- the window is decided on the host
                                                                          from <= 300 boxes and handed to the segmenter's transform as a
                                                                          source rectangle of the u8 image already in HBM; `crop_fn`: any
                                                                          other callable with the reference's signature)
        CircuitAnalyzer.segment_with_sam2                                src/circuit_analyzer.py:321-386

`prompts="learned"` is the reference's own segmentation (one mask per image from the wrapper's learned prompts);
`prompts="boxes"` feeds the detector's boxes to `infer_masks(images, boxes)` (upstream box prompting, BASELINE configs[4]):
one mask per detected component.

Images are independent through both models, so a batch shards contiguously over ranks with no collective on the data path
(`shard_range`); every rank returns the results of its own images, `gather_results` collects them on one rank when a single
caller needs them.  Results do not depend on how the batch is split (tests: sharded == unsharded, bit for bit).
"""
import time
from collections import defaultdict

import numpy as np
import torch

from .crop import adjust_bboxes, crop_image_and_adjust_bboxes, crop_window
from .detector import non_max_suppression_by_confidence
from .distributed import gather_rows, shard_range


def results_to_bboxes(r):
    """circuit_analyzer.py:267-287 on one `Results`: lists via .cpu().numpy().tolist(), python round() (half-to-even), uid string.
    (The four `round()` per box run as ONE np.rint over the float64 copy of the coordinates: the same half-to-even rounding of the same doubles.)"""
    ids = r.boxes.cls.cpu().numpy().tolist()
    conf = r.boxes.conf.cpu().numpy().tolist()
    xy = np.rint(r.boxes.xyxy.cpu().numpy().astype(np.float64)).astype(np.int64).tolist()
    names = r.names
    out = []
    for i, (x0, y0, x1, y1) in enumerate(xy):
        nm = names[int(ids[i])]
        out.append({"class": nm, "_yolo_class_id_temp": int(ids[i]), "confidence": conf[i], "xmin": x0, "ymin": y0, "xmax": x1, "ymax": y1,
                    "persistent_uid": f"{nm}_{x0}_{y0}_{x1}_{y1}"})
    return out


class CircuitPipeline:
    """detector: `circuitvision_amd.detector.YOLO`-like (`predict(list of uint8 HxWx3) -> [Results]`);
    segmenter: `SAM2Model`-like (`infer_masks(x, boxes=None)`, `.image_size`); transforms: `SAM2Transforms`-like."""

    def __init__(self, detector, segmenter, transforms, stage2_iou=0.6, max_prompts=32, crop_fn=None, swap_channels=True, seg_batch=16, crop=False,
                 crop_padding=80):
        """crop=True: the reference's chain -- detector -> stage-2 NMS -> crop window from the boxes (crop.py) -> segmenter on the window
        (analysis_pipeline.py:177 -> :206); False: the segmenter sees the whole image.  crop_fn overrides the built-in crop.
        swap_channels: segment_with_sam2 applies cv2.COLOR_BGR2RGB to whatever it is given (circuit_analyzer.py:343), and the
        pipeline hands it RGB (analysis_pipeline.py:199-203) -- i.e. the reference's segmenter sees the channels reversed.
        seg_batch: images per segmenter launch (BASELINE configs[2] runs SAM 2.1-L at 16)."""
        self.det, self.seg, self.tr = detector, segmenter, transforms
        self.stage2_iou, self.max_prompts, self.swap = stage2_iou, max_prompts, swap_channels
        self.crop_padding = int(crop_padding)
        self.builtin_crop = bool(crop) and crop_fn is None
        self.crop_fn = crop_fn if crop_fn is not None else ((lambda im, bb: crop_image_and_adjust_bboxes(im, bb, padding=self.crop_padding)) if crop else None)
        self.seg_batch = max(1, int(seg_batch))
        self.seg_slots = 2                         # segmenter plan instances (each with its own stream) the overlapped path alternates between
        self.timings = defaultdict(float)          # wall seconds per phase, accumulated over calls (bench.py prints them per step)

    def _tick(self, name, t0):
        t1 = time.perf_counter()
        self.timings[name] += t1 - t0
        return t1

    # ---- stage A: analysis_pipeline.py:97-115
    def detect(self, images):
        """-> per image: the bboxes that survive the second-stage NMS (list of dicts, original pixel coordinates)."""
        out = [None] * len(images)
        groups = {}
        for i, im in enumerate(images):
            groups.setdefault(im.shape[:2], []).append(i)
        for idxs in groups.values():                                   # one detector batch per image size
            t = time.perf_counter()
            res = self.det.predict([images[i] for i in idxs], verbose=False)
            t = self._tick("detect.predict (H2D + letterbox + YOLO11 + NMS + D2H)", t)
            for i, r in zip(idxs, res):
                out[i] = non_max_suppression_by_confidence(results_to_bboxes(r), iou_threshold=self.stage2_iou)
            self._tick("detect.glue (dicts + round + uid + stage-2 NMS)", t)
        return out

    # ---- stage B: analysis_pipeline.py:168-225
    def segment(self, images, bboxes, prompts="learned"):
        if prompts not in ("learned", "boxes"):
            raise ValueError("prompts must be 'learned' or 'boxes'")
        crops, boxes_adj, infos = [], [], []
        for im, bb in zip(images, bboxes):
            info = None
            if self.crop_fn is not None:
                im, bb, info = self.crop_fn(im, [dict(b) for b in bb])
            crops.append(im)
            boxes_adj.append(bb)
            infos.append(info)
        out = []
        for c0 in range(0, len(crops), self.seg_batch):                   # one segmenter launch per seg_batch images
            sl = slice(c0, c0 + self.seg_batch)
            out += self._segment_chunk(crops[sl], boxes_adj[sl], prompts)
        for r, info in zip(out, infos):
            r["crop_debug_info"] = info
        return out

    def _product_objects(self):
        """True when detector / segmenter / transforms are this package's own classes (the stream-ordered fast paths use their internals);
        duck-typed stand-ins (tests, other back ends) take the generic path."""
        from .detector import YOLO
        from .sam2_infer import SAM2Model, SAM2Transforms
        return isinstance(self.det, YOLO) and isinstance(self.seg, SAM2Model) and isinstance(self.tr, SAM2Transforms)

    def _segment_chunk(self, crops, boxes_adj, prompts):
        R = self.seg.image_size
        t = time.perf_counter()
        if self._product_objects():
            x = self.tr.forward_batch(crops, swap_rb=self.swap)           # the BGR2RGB of circuit_analyzer.py:343 as a channel index on the device
        else:
            x = self.tr.forward_batch([np.ascontiguousarray(im[..., ::-1]) if self.swap else im for im in crops])
        t = self._tick("segment.transform (channel swap + H2D + resize / normalise)", t)
        out = []
        if prompts == "learned":
            hi, lo, iou = self.seg.infer_masks(x)
            t = self._tick("segment.infer_masks (SAM 2.1 forward)", t)
            for b, im in enumerate(crops):
                u8, ext = self.tr.postprocess_to_mask(hi[b:b + 1], im.shape[:2])
                out.append({"image": im, "bboxes": boxes_adj[b], "mask": u8[0, 0], "extent": ext[0], "iou": iou[b]})
            self._tick("segment.postprocess (resize + threshold + u8 + extent + D2H of extents)", t)
            return out
        P = self.max_prompts
        bx = torch.zeros(len(crops), P, 4)
        counts = []
        for b, (im, bb) in enumerate(zip(crops, boxes_adj)):
            bb = bb[:P]                                                  # already sorted by confidence (stage-2 NMS order)
            counts.append(len(bb))
            if bb:
                t_ = torch.tensor([[d["xmin"], d["ymin"], d["xmax"], d["ymax"]] for d in bb], dtype=torch.float32)
                t_ = self.tr.transform_boxes(t_, normalize=True, orig_hw=im.shape[:2]).reshape(-1, 4)
                bx[b, :len(bb)] = t_
                bx[b, len(bb):] = t_[0]                                  # unused prompt slots repeat the first box (results dropped)
            else:
                bx[b] = torch.tensor([0.0, 0.0, R, R])
        _, lo, iou = self.seg.infer_masks(x, bx, return_high_res=False)
        t = self._tick("segment.infer_masks (SAM 2.1 forward)", t)
        for b, im in enumerate(crops):
            k = counts[b]
            if k:
                u8, ext = self.tr.postprocess_to_mask(lo[b, :k].unsqueeze(0), im.shape[:2])
                masks, ext = u8[0], ext
            else:
                masks, ext = torch.zeros(0, *im.shape[:2], dtype=torch.uint8, device=lo.device), []
            out.append({"image": im, "bboxes": boxes_adj[b][:P], "masks": masks, "extents": ext, "iou": iou[b, :k]})
        self._tick("segment.postprocess (resize + threshold + u8 + extent + D2H of extents)", t)
        return out

    def run_batch(self, images, prompts="learned", rank=0, world=1):
        """The share [lo, hi) of `images` that belongs to `rank`: detection, stage-2 NMS, (crop), segmentation.
        Returns [(global image index, result dict)]."""
        lo, hi = shard_range(len(images), rank, world)
        mine = list(images[lo:hi])
        if not mine:
            return []
        if prompts == "learned" and self.builtin_crop and self._product_objects():
            res = self._run_cropped(mine)
        elif prompts == "learned" and self.crop_fn is None and self._product_objects():
            res = self._run_overlapped(mine)
        else:
            bboxes = self.detect(mine)
            res = self.segment(mine, bboxes, prompts)
        return [(lo + i, r) for i, r in enumerate(res)]

    # ---- learned prompts, no crop: the segmenter does not depend on the detector's boxes (analysis_pipeline.py:168-225 passes the
    #      image, not the boxes, to segment_with_sam2), so the whole batch is ENQUEUED -- segmenter chunks on the segmenter's stream, the
    #      detector on its own -- and the host-side glue (D2H lists, round(), uid strings, stage-2 NMS) runs while the GPU works.
    def _run_overlapped(self, images):
        t = time.perf_counter()
        chunks = [images[c0:c0 + self.seg_batch] for c0 in range(0, len(images), self.seg_batch)]
        pend = [self._enqueue_learned(chunks[0], 0)]
        t = self._tick("enqueue: segmenter chunk 0 (stage u8 + H2D + transform + SAM 2.1 graph + post-process launches)", t)
        groups = {}
        for i, im in enumerate(images):
            groups.setdefault(im.shape[:2], []).append(i)
        handles = [(idxs, self.det.predict_async([images[i] for i in idxs])) for idxs in groups.values()]
        t = self._tick("enqueue: detector (stage u8 + H2D + letterbox + YOLO11 graph + NMS + D2H launches)", t)
        pend += [self._enqueue_learned(ch, (k + 1) % self.seg_slots) for k, ch in enumerate(chunks[1:])]
        t = self._tick("enqueue: segmenter chunks 1.. (same, while the GPU runs chunk 0)", t)
        bboxes = [None] * len(images)
        for idxs, h in handles:
            res = h.result()
            t = self._tick("wait: detector results on the host", t)
            for i, r in zip(idxs, res):
                bboxes[i] = non_max_suppression_by_confidence(results_to_bboxes(r), iou_threshold=self.stage2_iou)
            t = self._tick("detect.glue (dicts + round + uid + stage-2 NMS), overlapped with the segmenter", t)
        out, k = [], 0
        for ch, pd in zip(chunks, pend):
            u8s, exts, iou = pd()
            for b, im in enumerate(ch):
                out.append({"image": im, "bboxes": bboxes[k], "mask": u8s[b], "extent": exts[b], "iou": iou[b]})
                k += 1
        self._tick("wait: segmenter (GPU time not hidden behind host work) + extents to the host", t)
        return out

    # ---- learned prompts WITH the reference's crop: the segmenter's input depends on the detector's boxes (analysis_pipeline.py:177 -> :206).
    #      What keeps the GPU busy across that dependency: the detector runs in chunks (one H2D for all images, a graph replay per chunk); as soon
    #      as chunk k's boxes are on the host, the glue + stage-2 NMS + crop window of its images run there and segmenter chunk k is enqueued --
    #      its transform reads each window out of the u8 block the detector's letterbox read -- while the GPU is still in detector chunks k + 1 ..
    #      and segmenter chunks < k.  Only chunk 0's detector pass + glue is not hidden.
    def _run_cropped(self, images):
        t = time.perf_counter()
        groups = {}
        for i, im in enumerate(images):
            groups.setdefault(im.shape[:2], []).append(i)
        work = []                                                          # (image indices of the chunk, detector handle)
        for idxs in groups.values():
            hs = self.det.predict_chunks_async([images[i] for i in idxs], self.seg_batch)
            work += [(idxs[k * self.seg_batch:(k + 1) * self.seg_batch], h) for k, h in enumerate(hs)]
        t = self._tick("enqueue: detector chunks (stage u8 + ONE H2D + per chunk: letterbox + YOLO11 graph + NMS + D2H launches)", t)
        out, pend = [None] * len(images), []
        for k, (idxs, h) in enumerate(work):
            res = h.result()
            t = self._tick("wait: detector chunk on the host" if k else "wait: detector chunk 0 on the host (not hidden: the segmenter needs its boxes)", t)
            wins, metas = [], []
            for i, r in zip(idxs, res):
                bb = non_max_suppression_by_confidence(results_to_bboxes(r), iou_threshold=self.stage2_iou)
                win, info = crop_window(bb, images[i].shape[:2], self.crop_padding)
                wins.append(win)
                metas.append((adjust_bboxes(bb, win), info))
            t = self._tick("glue (dicts + round + uid + stage-2 NMS + crop window + box shift)" + (", overlapped with the GPU" if k else ", chunk 0: not hidden"), t)
            pend.append((idxs, wins, metas, self._enqueue_learned(None, k % self.seg_slots, src=h.src, windows=wins, det_stream=self.det.stream)))
            t = self._tick("enqueue: segmenter chunk (transform from the windows of the detector's u8 block + SAM 2.1 graph + post-process)", t)
        for idxs, wins, metas, fin in pend:
            u8s, exts, iou = fin()
            for b, i in enumerate(idxs):
                im, wnd = images[i], wins[b]
                view = im if wnd is None else im[wnd[1]:wnd[3], wnd[0]:wnd[2]]              # (a numpy view, as the reference's slice)
                out[i] = {"image": view, "bboxes": metas[b][0], "mask": u8s[b], "extent": exts[b], "iou": iou[b], "window": wnd,
                          "crop_debug_info": metas[b][1]}
        self._tick("wait: segmenter (GPU time not hidden behind host work) + extents to the host", t)
        return out

    def _enqueue_learned(self, imgs, slot=0, src=None, windows=None, det_stream=None):
        """transform -> SAM 2.1 (learned prompts) -> resize / threshold / u8 / extent for one chunk, enqueued on the stream of segmenter plan
        `slot` (consecutive chunks alternate between two plan instances on two streams: independent images, and two segmenter graphs side by
        side fill each other's kernel tails).  Returns a closure that waits for it: -> (u8 masks [H,W] per image, extent tuples, iou [B,1])."""
        from . import _lib
        seg, tr = self.seg, self.tr
        lib = _lib.load()
        if src is not None:                                             # windows of a u8 device block (the cropped chain)
            B, R = src.shape[0], seg.image_size
            sizes = [tuple(src.shape[1:3]) if w is None else (w[3] - w[1], w[2] - w[0]) for w in windows]
        else:
            B, R = len(imgs), seg.image_size
            sizes = [tuple(im.shape[:2]) for im in imgs]
        same = all(sz == sizes[0] for sz in sizes)
        with seg._lock, torch.cuda.device(seg.dev):
            p = seg.plan(B, slot=slot)
            sst = seg.slot_stream(slot)
            # outputs come from the CALLER's allocator pool; the segmenter's stream is ordered behind the caller's before it writes them
            iou = torch.empty_like(p.iou)
            ext = torch.empty(B, 4, dtype=torch.int32, device=seg.dev)
            ext_h = torch.empty(B, 4, dtype=torch.int32, pin_memory=True)
            if same:
                u8 = torch.empty(B, sizes[0][0], sizes[0][1], dtype=torch.uint8, device=seg.dev)
                u8s = [u8[b] for b in range(B)]
            else:                                                           # planes of different sizes, packed back to back: one launch
                offs = np.concatenate(([0], np.cumsum([h * w for h, w in sizes], dtype=np.int64)))
                u8 = torch.empty(int(offs[-1]), dtype=torch.uint8, device=seg.dev)
                u8s = [u8[int(offs[b]):int(offs[b + 1])].view(h, w) for b, (h, w) in enumerate(sizes)]
                sz_np = np.asarray(sizes, dtype=np.int32)
            sst.wait_stream(torch.cuda.current_stream())
            if det_stream is not None:
                sst.wait_stream(det_stream)                                 # (the u8 block's H2D copy: already complete -- result() waited -- but stated)
            for t_ in (iou, ext, u8) + ((src,) if src is not None else ()):
                t_.record_stream(sst)                                       # written / read on sst: the allocator must not hand them out before sst is past them
            with torch.cuda.stream(sst):
                sp = sst.cuda_stream
                if src is not None:
                    tr.forward_windows(src, windows, swap_rb=self.swap, out=p.x_in.t, out_dtype=seg.dtype)
                else:
                    tr.forward_batch(imgs, swap_rb=self.swap, out=p.x_in.t, out_dtype=seg.dtype)
                p.plan.run()
                iou.copy_(p.iou, non_blocking=True)
                hi = p.high_res                                             # f32 [B,1,R,R]
                if same:
                    _lib.check(lib.cvmi_mask_postprocess(hi.data_ptr(), B, R, R, sizes[0][0], sizes[0][1], float(tr.mask_threshold), u8.data_ptr(),
                                                         ext.data_ptr(), sp), "mask_postprocess")
                else:
                    _lib.check(lib.cvmi_mask_postprocess_sizes(hi.data_ptr(), B, R, R, sz_np.ctypes.data, float(tr.mask_threshold), u8.data_ptr(),
                                                               ext.data_ptr(), sp), "mask_postprocess_sizes")
                ext_h.copy_(ext, non_blocking=True)
                done = torch.cuda.Event()
                done.record(sst)

        def finish():
            done.synchronize()
            return u8s, [None if x1 < 0 else (x0, y0, x1 + 1, y1 + 1) for x0, y0, x1, y1 in ext_h.tolist()], iou
        return finish


def gather_results(results, key="mask", dst=0):
    """Collect one same-shaped tensor per image (e.g. the u8 mask of equally sized images) from every rank on `dst`:
    returns {global index: tensor} there, None elsewhere.  Shards may be uneven, and a rank may hold NO image (more ranks than
    images): it still enters the collectives -- with zero rows of the dtype / trailing shape the other ranks report -- instead
    of leaving them waiting."""
    import torch.distributed as dist
    local = [r[key] for _, r in results]
    multi = dist.is_initialized() and dist.get_world_size() > 1
    meta = (local[0].dtype, tuple(local[0].shape), str(local[0].device)) if local else None
    if multi:
        metas = [None] * dist.get_world_size()
        dist.all_gather_object(metas, meta)                                # every rank, before anything that could raise
        known = [m for m in metas if m is not None]
        if not known:
            return {} if dist.get_rank() == dst else None
        if any(m[:2] != known[0][:2] for m in known):
            raise ValueError(f"gather_results: ranks disagree on the dtype / shape of '{key}': {sorted(set(m[:2] for m in known), key=str)}")
        if meta is None:                                                   # empty shard: zero rows, on this rank's device
            dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
            meta = (known[0][0], known[0][1], str(dev))
    elif meta is None:
        return {}
    dev = torch.device(meta[2])
    idx = torch.tensor([i for i, _ in results], dtype=torch.int64, device=dev)
    vals = torch.stack(local) if local else torch.zeros((0,) + meta[1], dtype=meta[0], device=dev)
    gi, gv = gather_rows(idx, dst), gather_rows(vals, dst)
    if gi is None:
        return None
    return {int(i): v for ids, vs in zip(gi, gv) for i, v in zip(ids.tolist(), vs)}
