"""Seeded synthetic inputs shared by tests, smoke() and bench.py."""
import numpy as np
import torch


def circuit_image(h, w, seed=20250704):
    """uint8 HxWx3: white-ish background, black axis-aligned segments and hollow rectangles."""
    rng = np.random.default_rng(seed)
    img = rng.integers(235, 256, size=(h, w, 3), dtype=np.uint8)
    for _ in range(int(rng.integers(20, 61))):
        t = int(rng.integers(1, 4))
        if rng.random() < 0.5:
            y, x0, x1 = int(rng.integers(0, h - t)), *sorted(rng.integers(0, w, 2).tolist())
            img[y:y + t, x0:x1] = 0
        else:
            x, y0, y1 = int(rng.integers(0, w - t)), *sorted(rng.integers(0, h, 2).tolist())
            img[y0:y1, x:x + t] = 0
    for _ in range(int(rng.integers(5, 33))):
        x0, y0 = int(rng.integers(0, w - 12)), int(rng.integers(0, h - 12))
        bw, bh = int(rng.integers(8, max(9, w // 6))), int(rng.integers(8, max(9, h // 6)))
        x1, y1 = min(x0 + bw, w - 1), min(y0 + bh, h - 1)
        img[y0:y0 + 2, x0:x1] = 0; img[y1:y1 + 2, x0:x1] = 0
        img[y0:y1, x0:x0 + 2] = 0; img[y0:y1, x1:x1 + 2] = 0
    return img


def nms_stress_pred(B, nc=62, hw=(640, 640), seed=0, frac_logit=-4.0):
    """[B, 4+nc, A] xywh + class scores: jittered 8-200 px boxes on the anchor grid, class scores
    sigmoid(N(frac_logit, 1.5)) so that ~1-3 % of anchors clear 0.25; duplicates of strong boxes
    are planted so that suppression actually happens."""
    H, W = hw
    out = []
    for b in range(B):
        g = torch.Generator().manual_seed(seed * 1000 + b)
        cxs, cys = [], []
        for s in (8, 16, 32):
            ys, xs = torch.meshgrid(torch.arange(H // s) + 0.5, torch.arange(W // s) + 0.5, indexing="ij")
            cxs.append(xs.flatten() * s); cys.append(ys.flatten() * s)
        cx, cy = torch.cat(cxs), torch.cat(cys)
        A = cx.numel()
        cx = cx + torch.randn(A, generator=g) * 4
        cy = cy + torch.randn(A, generator=g) * 4
        w = torch.empty(A).uniform_(8, 200, generator=g)
        h = torch.empty(A).uniform_(8, 200, generator=g)
        cls = torch.sigmoid(torch.randn(nc, A, generator=g) * 1.5 + frac_logit)
        # plant clusters: copy a strong anchor's box/class onto ~8 neighbours with small jitter
        strong = torch.nonzero(cls.amax(0) > 0.5).flatten()[:40]
        for a in strong.tolist():
            nb = torch.randint(0, A, (8,), generator=g)
            cx[nb] = cx[a] + torch.randn(8, generator=g) * 2
            cy[nb] = cy[a] + torch.randn(8, generator=g) * 2
            w[nb] = w[a] * (1 + torch.randn(8, generator=g) * 0.05)
            h[nb] = h[a] * (1 + torch.randn(8, generator=g) * 0.05)
            cls[:, nb] = cls[:, a:a + 1] * torch.empty(1, 8).uniform_(0.6, 1.0, generator=g)
        out.append(torch.cat((torch.stack((cx, cy, w, h)), cls), 0))
    return torch.stack(out).float().contiguous()


def calibrated_yolo_params(scale, nc, seed, x, cand_per_image=45, spread=2.0, conf=0.25, box_decay=0.45, bn_beta_shift=1.0):
    """Seeded synthetic YOLO11 weights that DETECT something on the images `x` ([B,3,H,W] f32, letterboxed).

    Seeded random weights leave every class score below ~0.02 and contract the activations to near-constants (neck features of
    std 0.09), so NMS, scale_boxes and the Results plumbing would be compared on empty lists and loose absolute tolerances would
    not bite.  Two calibration steps, both on the CPU oracle:
      1. BatchNorm statistics: one pass over `x` with every BN in cumulative-average training mode sets running_mean / running_var
         to the statistics the layer actually sees, so activations stay O(1) through all 24 layers (as in a trained network);
      2. class branch: the final class conv of each level (model.23.cv3.<i>.2) is rescaled so its logits have standard deviation
         `spread`, and its bias shifted so that every image has about cand_per_image / 3 (or more) anchors per level above `conf` -- the threshold is
         placed in the middle of the widest gap between neighbouring best-class logits near that rank, so rounding noise cannot
         move an anchor across it.
    Returns SyntheticParams (its state_dict() loads strictly into oracle.yolo11.YOLO11).  TEST INFRASTRUCTURE: uses the oracle."""
    import math
    from circuitvision_amd._lib import F32
    from circuitvision_amd.yolo11 import SyntheticParams, Yolo11Weights
    from oracle.yolo11 import YOLO11
    params = SyntheticParams(seed=seed, nc=nc)
    Yolo11Weights(scale, nc, params, F32, device="cpu")           # walks the graph: fills params.sd with every tensor
    m = YOLO11(scale, nc).eval()
    m.load_state_dict(params.state_dict(), strict=True)
    bns = [mod for mod in m.modules() if isinstance(mod, torch.nn.BatchNorm2d)]
    if bn_beta_shift:
        for k in list(params.sd):
            if k.endswith("bn.bias"):
                params.sd[k].add_(bn_beta_shift)
        m.load_state_dict(params.state_dict(), strict=True)
    for bn in bns:
        bn.reset_running_stats()
        bn.momentum = None                                        # cumulative average: after one pass, running stats = batch stats
        bn.train()
    with torch.no_grad():
        m(x)
    m.eval()
    for k, v in m.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            params.sd[k].copy_(v)
    # 3. box branch: a per-bin bias ramp makes the DFL expectation small (boxes of a few strides, like components on a schematic)
    #    instead of ~7.5 bins everywhere (every box 120-480 px wide: the second-stage NMS would keep one or two of them)
    for i in range(3):
        b = params.sd[f"model.23.cv2.{i}.2.bias"]
        b.copy_(b - 1.0 - box_decay * torch.arange(16, dtype=torch.float32).repeat(4))
    m.load_state_dict(params.state_dict(), strict=True)
    with torch.no_grad():
        _, raw, _ = m(x, return_feats=True)
    B = x.shape[0]
    thr = math.log(conf / (1 - conf))
    for i, r in enumerate(raw):
        w, b = params.sd[f"model.23.cv3.{i}.2.weight"], params.sd[f"model.23.cv3.{i}.2.bias"]
        z = r[:, 64:] - b.view(1, -1, 1, 1)                      # W u  (no bias)
        g = spread / float(z.std())
        w.mul_(g)
        b.copy_(b - b.mean())                                     # keep the per-class offsets (class variety), drop the -6 mean
        bl = (z * g + b.view(1, -1, 1, 1)).amax(1).flatten(1)     # [B, anchors of this level]: best-class logit
        kb = max(2, min(int(round(cand_per_image / 3)), bl.shape[1] // 3))
        target = float(bl.sort(1, descending=True).values[:, kb - 1].min())    # EVERY image gets >= kb candidates on this level
        best = bl.flatten().sort(descending=True).values
        k = max(2, int((best >= target).sum()))
        lo, hi = max(1, k - max(1, k // 5)), min(best.numel() - 1, k + max(2, k // 5) + 1)
        gaps = best[lo - 1:hi - 1] - best[lo:hi]                 # gap between rank j-1 and j for j in [lo, hi)
        j = lo + int(gaps.argmax())
        assert float(gaps.max()) > 2e-3, "degenerate logits: no usable gap for the confidence threshold"
        b.add_(thr - float(best[j - 1] + best[j]) / 2)
    return params
