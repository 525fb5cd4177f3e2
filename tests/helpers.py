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


def assert_same_detections(name, got, ref, top=20, min_overlap=0.97):
    """Two detection lists (anchor indices or uid strings, confidence-descending) that should be IDENTICAL.  Exact equality is
    required of the `top` most confident entries and of the lengths within 3 %; over the whole list a Jaccard overlap >= min_overlap
    is accepted instead of equality, because with hundreds of boxes some pair's IoU (or two confidences) lies within fp32 rounding
    noise of a threshold, and ONE such flip in either implementation legitimately changes one kept box -- and shifts every later
    position.  Returns the overlap."""
    got, ref = list(got), list(ref)
    assert got[:top] == ref[:top], (name, got[:top], ref[:top])
    union = len(set(got) | set(ref))
    ov = len(set(got) & set(ref)) / max(1, union)
    print(f"{name}: {len(got)} vs {len(ref)} detections, overlap {ov:.4f}, identical order: {got == ref}")
    assert ov >= min_overlap and abs(len(got) - len(ref)) <= max(1, 0.03 * len(ref)), (name, len(got), len(ref), ov)
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
