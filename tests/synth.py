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
