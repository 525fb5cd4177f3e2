"""Oracle: detector post-processing, CPU fp32.  TEST INFRASTRUCTURE ONLY.

Two stages, as the reference runs them:

1. `yolo_nms` -- what `self.yolo.predict(image)` does after the network
   (/root/reference/src/circuit_analyzer.py:268; algorithm lives in un-vendored ultralytics
   `ops.non_max_suppression` + `torchvision.ops.nms`, requirements.txt:7,9 -- parity unpinned):
   conf 0.25, IoU 0.7, per-class via +cls*7680 offset, max_det 300.  SURVEY.md 8(a) row A5.
2. `nms_by_confidence` / `calculate_iou` -- restatement of /root/reference/src/utils.py:297-361
   (second-stage class-agnostic NMS on rounded integer boxes, caller
   analysis_pipeline.py:106 with iou_threshold=0.6).  PINNED by tests/golden/nms_stage2.json.
"""
import numpy as np
import torch


# --------------------------------------------------------------------------- stage 1 (ultralytics)
def torchvision_nms(boxes, scores, iou_thr):
    """Greedy NMS with torchvision's CPU kernel arithmetic (fp32):
    order = argsort(scores, descending, stable); suppress j if inter/(a_i + a_j - inter) > thr."""
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros(0, dtype=torch.long)
    b = boxes.detach().to(torch.float32).numpy()
    order = torch.sort(scores.to(torch.float32), descending=True, stable=True).indices.numpy()
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    areas = ((x2 - x1) * (y2 - y1)).astype(np.float32)
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    thr = np.float32(iou_thr)
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        xx1 = np.maximum(x1[i], x1[rest])
        yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest])
        yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(np.float32(0), xx2 - xx1)
        h = np.maximum(np.float32(0), yy2 - yy1)
        inter = (w * h).astype(np.float32)
        ovr = inter / ((areas[i] + areas[rest]).astype(np.float32) - inter)
        suppressed[rest[ovr > thr]] = True
    return torch.as_tensor(np.asarray(keep, dtype=np.int64))


def xywh2xyxy(x):
    y = torch.empty_like(x)
    xy, wh = x[..., :2], x[..., 2:] / 2
    y[..., :2] = xy - wh
    y[..., 2:] = xy + wh
    return y


def yolo_nms(prediction, conf_thres=0.25, iou_thres=0.7, max_det=300, max_nms=30000, max_wh=7680,
             return_indices=False):
    """prediction [B, 4+nc, A] (xywh + class scores) -> list of [n, 6] (x1,y1,x2,y2,conf,cls).

    multi_label=False, agnostic=False, classes=None.  Ties in confidence are broken by anchor
    index (stable sort) -- upstream's argsort is not stable, so equal scores are unpinned there.
    With return_indices also returns, per image, the anchor index of every kept row."""
    prediction = prediction.to(torch.float32)
    B, no, A = prediction.shape
    nc = no - 4
    xc = prediction[:, 4:4 + nc].amax(1) > conf_thres
    prediction = prediction.transpose(-1, -2).clone()
    prediction[..., :4] = xywh2xyxy(prediction[..., :4])
    out, idxs = [], []
    for xi in range(B):
        anchor_idx = torch.nonzero(xc[xi]).flatten()
        x = prediction[xi][xc[xi]]
        if x.shape[0] == 0:
            out.append(torch.zeros(0, 6)); idxs.append(torch.zeros(0, dtype=torch.long))
            continue
        box, cls = x[:, :4], x[:, 4:]
        conf, j = cls.max(1, keepdim=True)
        x = torch.cat((box, conf, j.float()), 1)
        m = conf.view(-1) > conf_thres
        x, anchor_idx = x[m], anchor_idx[m]
        order = torch.sort(x[:, 4], descending=True, stable=True).indices[:max_nms]
        x, anchor_idx = x[order], anchor_idx[order]
        c = x[:, 5:6] * max_wh
        keep = torchvision_nms(x[:, :4] + c, x[:, 4], iou_thres)[:max_det]
        out.append(x[keep]); idxs.append(anchor_idx[keep])
    return (out, idxs) if return_indices else out


def clip_boxes(boxes, shape):
    boxes[..., 0].clamp_(0, shape[1])
    boxes[..., 1].clamp_(0, shape[0])
    boxes[..., 2].clamp_(0, shape[1])
    boxes[..., 3].clamp_(0, shape[0])
    return boxes


def scale_boxes(img1_shape, boxes, img0_shape):
    """Map xyxy boxes from the letterboxed shape back to the original image (ultralytics ops.scale_boxes)."""
    gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
    pad_x = round((img1_shape[1] - img0_shape[1] * gain) / 2 - 0.1)
    pad_y = round((img1_shape[0] - img0_shape[0] * gain) / 2 - 0.1)
    boxes = boxes.clone()
    boxes[..., 0] -= pad_x
    boxes[..., 1] -= pad_y
    boxes[..., 2] -= pad_x
    boxes[..., 3] -= pad_y
    boxes[..., :4] /= gain
    return clip_boxes(boxes, img0_shape)


# --------------------------------------------------------------------------- stage 2 (reference-owned)
def calculate_iou(b1, b2):
    """src/utils.py:297-328 -- no +1, max(.,0) on each side, union>0 guard."""
    iw = max(min(b1["xmax"], b2["xmax"]) - max(b1["xmin"], b2["xmin"]), 0)
    ih = max(min(b1["ymax"], b2["ymax"]) - max(b1["ymin"], b2["ymin"]), 0)
    inter = iw * ih
    a1 = (b1["xmax"] - b1["xmin"]) * (b1["ymax"] - b1["ymin"])
    a2 = (b2["xmax"] - b2["xmin"]) * (b2["ymax"] - b2["ymin"])
    union = a1 + a2 - inter
    return inter / union if union > 0 else 0.0


def _greedy(bboxes, iou_threshold):
    kept = []
    rest = list(bboxes)
    while rest:
        top = rest.pop(0)
        kept.append(top)
        rest = [b for b in rest if calculate_iou(top, b) < iou_threshold]
    return kept


def nms_by_confidence(bboxes, iou_threshold=0.5):
    """src/utils.py:346-361 -- stable sort by confidence desc; keep iff IoU < thr vs every kept box."""
    return _greedy(sorted(bboxes, key=lambda b: b["confidence"], reverse=True), iou_threshold)


def nms_by_area(bboxes, iou_threshold=0.5):
    """src/utils.py:330-344 -- same greedy rule, ordered by box area desc."""
    return _greedy(sorted(bboxes, key=lambda b: (b["xmax"] - b["xmin"]) * (b["ymax"] - b["ymin"]),
                          reverse=True), iou_threshold)


def boxes_to_dicts(xyxy, conf, cls, names):
    """circuit_analyzer.py:270-287 -- python round() (half-to-even) of each coordinate + uid string."""
    out = []
    for (x0, y0, x1, y1), c, k in zip(xyxy, conf, cls):
        n = names[int(k)]
        out.append({"class": n, "_yolo_class_id_temp": int(k), "confidence": c,
                    "xmin": round(x0), "ymin": round(y0), "xmax": round(x1), "ymax": round(y1),
                    "persistent_uid": f"{n}_{round(x0)}_{round(y0)}_{round(x1)}_{round(y1)}"})
    return out
