"""Shared helpers for the GPU parity tests (HIP path vs the CPU oracle)."""
import torch

from circuitvision_amd import _lib
from circuitvision_amd._lib import BF16, F16, F32
from circuitvision_amd.engine import Buf, Plan, TORCH_DTYPE

TOL = {F16: dict(rtol=2e-2, atol=2e-2), F32: dict(rtol=1e-4, atol=1e-4), BF16: dict(rtol=1.2e-1, atol=1.2e-1)}


def stream():
    return torch.cuda.Stream()


def to_buf(x_nchw, dtype, c_total=None, c0=0):
    """NCHW float32 CPU tensor -> Buf (NHWC on device), optionally inside a wider buffer."""
    B, C, H, W = x_nchw.shape
    buf = Buf(B, H, W, c_total or C, dtype, zero=True)
    buf.t[..., c0:c0 + C] = x_nchw.permute(0, 2, 3, 1).to(TORCH_DTYPE[dtype]).cuda()
    return buf


def from_view(view):
    """View -> NCHW float32 CPU tensor."""
    return view.tensor().float().permute(0, 3, 1, 2).contiguous().cpu()


def quant(x, dtype):
    """Round a CPU f32 tensor to the storage dtype (so the oracle sees the same inputs)."""
    return x.to(TORCH_DTYPE[dtype]).float()


def run(plan):
    torch.cuda.synchronize()          # buffer fills ran on the default stream
    plan.run_eager()
    plan.stream.synchronize()


def err_stats(got, ref):
    """(max |err|, rms err, std of the reference) as floats -- tolerances in the whole-model tests are stated RELATIVE to the
    reference's standard deviation, so they keep biting whatever the scale of the synthetic activations is."""
    d = (got.float() - ref.float())
    return float(d.abs().max()), float(d.pow(2).mean().sqrt()), float(ref.float().std())


def assert_rel(name, got, ref, max_rel, rms_rel, failures=None, absolute=False):
    """max |err| <= max_rel * std(ref) and rms err <= rms_rel * std(ref); absolute=True: the bounds are absolute.
    With `failures` (a list) a violation is recorded there instead of raised, so one run reports every tensor."""
    mx, rms, sd = err_stats(got, ref)
    if absolute:
        max_rel, rms_rel = max_rel / sd, rms_rel / sd
    line = f"{name}: max|err| {mx:.3e} ({mx / sd:.2e} std), rms {rms:.3e} ({rms / sd:.2e} std), std(ref) {sd:.3f}"
    print(line)
    ok = sd > 0 and mx <= max_rel * sd and rms <= rms_rel * sd
    if failures is not None:
        if not ok:
            failures.append(line + f"  [bounds: max {max_rel:.2e} std, rms {rms_rel:.2e} std]")
        return
    assert ok, line


def save_converted_yolo(path, params, scale, nc, imgsz=None):
    """The converted-checkpoint format `YOLO(path)` reads ({'state_dict', 'names', 'scale'[, 'imgsz']})."""
    ck = {"state_dict": params.state_dict(), "names": {i: f"class{i}" for i in range(nc)}, "scale": scale}
    if imgsz is not None:
        ck["imgsz"] = imgsz
    torch.save(ck, path)
    return path


def _xyxy_cls(pred, idx, max_wh=7680.0):
    """Class-offset xyxy boxes, best score and class of anchors `idx` of ONE image's oracle prediction [4 + nc, A]."""
    p = pred[:, idx].double()
    sc, cl = p[4:].max(0)
    cx, cy, w, h = p[0], p[1], p[2], p[3]
    off = cl.double() * max_wh
    return torch.stack((cx - w / 2 + off, cy - h / 2 + off, cx + w / 2 + off, cy + h / 2 + off), 1), sc, cl


def explain_flips(pred, got, ref, conf=0.25, iou_thr=0.7, tol_score=1e-3, tol_iou=2e-3):
    """Why do two kept-anchor lists of the same image differ?  `pred`: the ORACLE's prediction [4 + nc, A] of that image.  An anchor in
    exactly one list is explained iff (checked on the oracle's own numbers, in float64)
      (i)  its best class score is within tol_score of the confidence threshold, or
      (ii) its IoU with some other candidate of the same class is within tol_iou of the NMS threshold (one suppression decision sits
           on the threshold: either implementation may round it the other way), or
      (iii) it overlaps (IoU > iou_thr - tol_iou) an anchor that is itself in exactly one list (the flip of (i) / (ii) cascades: what that
           box suppressed is now kept, or the reverse).
    tol_score / tol_iou are the f32-mode error bounds of the network outputs the two NMS runs start from (scores 1e-3 = the north_star
    bound; boxes 2e-2 px on boxes >= 10 px wide move an IoU by <= 2e-3), not free parameters.  Returns [(anchor, reason | None)]."""
    diff = sorted(set(got) ^ set(ref))
    if not diff:
        return []
    sc_all = pred[4:].amax(0)
    cand = torch.nonzero(sc_all > conf - tol_score).flatten()
    cb, _, _ = _xyxy_cls(pred, cand)
    db, dsc, _ = _xyxy_cls(pred, torch.tensor(diff))

    def iou(a, b):
        lt, rb = torch.max(a[:, None, :2], b[None, :, :2]), torch.min(a[:, None, 2:], b[None, :, 2:])
        inter = (rb - lt).clamp(min=0).prod(-1)
        ar = lambda t: (t[:, 2] - t[:, 0]) * (t[:, 3] - t[:, 1])
        return inter / (ar(a)[:, None] + ar(b)[None] - inter)

    i_c, i_d = iou(db, cb), iou(db, db)
    why = {}
    for k, a in enumerate(diff):                                  # rules (i) and (ii): the flip sits on a threshold
        others = cand != a
        if abs(float(dsc[k]) - conf) <= tol_score:
            why[a] = f"score {float(dsc[k]):.6f} within {tol_score} of conf {conf}"
        elif bool(((i_c[k] - iou_thr).abs() <= tol_iou)[others].any()):
            j = int(torch.argmin((i_c[k] - iou_thr).abs() + (~others) * 9.0))
            why[a] = f"IoU {float(i_c[k, j]):.6f} with anchor {int(cand[j])} within {tol_iou} of {iou_thr}"
    changed = True
    while changed:                                                # rule (iii), to the fixed point: a cascade must START at a threshold tie --
        changed = False                                           # two unexplained flips that merely overlap each other explain nothing
        for k, a in enumerate(diff):
            if a in why:
                continue
            for m, b in enumerate(diff):
                if m != k and b in why and float(i_d[k, m]) > iou_thr - tol_iou:
                    why[a] = f"cascade of anchor {b} (IoU {float(i_d[k, m]):.4f}), itself explained: {why[b].split(' within')[0]}"
                    changed = True
                    break
    return [(a, why.get(a)) for a in diff]


def assert_same_detections(name, got, ref, top=20, min_overlap=0.97, pred=None, **thr):
    """Two detection lists (anchor indices or uid strings, confidence-descending) that should be IDENTICAL -- the north_star's
    "identical integer box indices".  The overlap and the "identical order" flag are printed AND carried in every assertion message.
    With `pred` (the oracle's prediction [4 + nc, A] of this image; lists = kept anchor indices) a difference is only accepted when
    explain_flips() traces EVERY differing anchor to a decision that sits on a threshold within the f32 error bound of the network
    outputs -- checked, not assumed; anything else fails.  Without `pred` (uid strings built from rounded coordinates) exact equality
    of the `top` most confident entries, a Jaccard overlap >= min_overlap and lengths within 3 % are required.  Returns the overlap."""
    got, ref = list(got), list(ref)
    union = len(set(got) | set(ref))
    ov = len(set(got) & set(ref)) / max(1, union)
    msg = f"{name}: {len(got)} vs {len(ref)} detections, overlap {ov:.4f}, identical order: {got == ref}"
    print(msg)
    if got == ref:
        return ov
    if pred is not None:
        why = explain_flips(pred, got, ref, **thr)
        bad = [a for a, r in why if r is None]
        print("\n".join(f"  anchor {a}: {r}" for a, r in why))
        assert not bad, f"{msg}; anchors kept by only one side with NO threshold-tie explanation: {bad}"
        # order of the anchors both sides kept: confidence-descending on the ORACLE's scores, up to swaps between scores closer than the
        # f32-mode bound (two implementations may order near-equal scores differently; anything else is a sorting / NMS bug)
        ts = thr.get("tol_score", 1e-3)
        sc = pred[4:].amax(0)
        common = [a for a in got if a in set(ref)]
        for a, b in zip(common, common[1:]):
            assert float(sc[a]) >= float(sc[b]) - ts, f"{msg}; common anchors {a} ({float(sc[a]):.6f}) before {b} ({float(sc[b]):.6f}): not confidence-descending"
    assert got[:top] == ref[:top] or pred is not None, (msg, got[:top], ref[:top])
    assert ov >= min_overlap and abs(len(got) - len(ref)) <= max(1, 0.03 * len(ref)), msg
    return ov


def box_match_rate(got, ref, iou_thr=0.85):
    """Fraction of reference boxes (dicts with xmin/ymin/xmax/ymax/class) that some box of the same class in `got` overlaps with
    IoU >= iou_thr -- the fp16-mode counterpart of list equality (rounded coordinates move by a pixel under fp16 noise)."""
    def iou(a, b):
        iw = max(min(a["xmax"], b["xmax"]) - max(a["xmin"], b["xmin"]), 0)
        ih = max(min(a["ymax"], b["ymax"]) - max(a["ymin"], b["ymin"]), 0)
        u = (a["xmax"] - a["xmin"]) * (a["ymax"] - a["ymin"]) + (b["xmax"] - b["xmin"]) * (b["ymax"] - b["ymin"]) - iw * ih
        return iw * ih / u if u > 0 else 0.0
    hit = sum(1 for r in ref if any(g["class"] == r["class"] and iou(g, r) >= iou_thr for g in got))
    return hit / max(1, len(ref))
