"""Oracle: YOLO11 detection model, CPU fp32, plain PyTorch.  TEST INFRASTRUCTURE ONLY.

Restates the network the reference runs through `self.yolo.predict(...)`
(/root/reference/src/circuit_analyzer.py:45, :268).  The architecture itself lives in the
un-vendored `ultralytics` package (requirements.txt:7, unpinned), so this file follows the
published YOLO11 definition (yolo11.yaml + nn/modules) as summarised in SURVEY.md section 8 Table Y.
Module / parameter names mirror ultralytics' (`model.<i>.cv1.conv.weight`, `...bn.running_mean`)
so that a real checkpoint's state_dict keys map one-to-one.

Known-answer anchors (tests/test_oracle_yolo.py): 2.624 M params (n) / 25.37 M (l) at nc=80.
parity unpinned: no reference-owned test or fixture exists for this network.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

SCALES = {  # depth, width, max_channels
    "n": (0.50, 0.25, 1024),
    "s": (0.50, 0.50, 1024),
    "m": (0.50, 1.00, 512),
    "l": (1.00, 1.00, 512),
    "x": (1.00, 1.50, 512),
}


def make_divisible(x, d=8):
    return int(math.ceil(x / d) * d)


class Conv(nn.Module):
    """Conv2d(no bias) + BatchNorm2d(eps=1e-3) + SiLU (or identity)."""

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, act=True):
        super().__init__()
        p = k // 2 if p is None else p
        self.conv = nn.Conv2d(c1, c2, k, s, p, groups=g, bias=False)
        self.bn = nn.BatchNorm2d(c2, eps=1e-3, momentum=0.03)
        self.act = nn.SiLU() if act else nn.Identity()

    def forward(self, x):
        return self.act(self.bn(self.conv(x)))


class Bottleneck(nn.Module):
    def __init__(self, c1, c2, shortcut=True, g=1, k=(3, 3), e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, k[0], 1)
        self.cv2 = Conv(c_, c2, k[1], 1, g=g)
        self.add = shortcut and c1 == c2

    def forward(self, x):
        y = self.cv2(self.cv1(x))
        return x + y if self.add else y


class C3k(nn.Module):
    def __init__(self, c1, c2, n=1, shortcut=True, g=1, e=0.5, k=3):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c1, c_, 1, 1)
        self.cv3 = Conv(2 * c_, c2, 1)
        self.m = nn.Sequential(*(Bottleneck(c_, c_, shortcut, g, k=(k, k), e=1.0) for _ in range(n)))

    def forward(self, x):
        return self.cv3(torch.cat((self.m(self.cv1(x)), self.cv2(x)), 1))


class C3k2(nn.Module):
    """C2f whose inner blocks are Bottleneck (c3k=False) or C3k(n=2) (c3k=True)."""

    def __init__(self, c1, c2, n=1, c3k=False, e=0.5, g=1, shortcut=True):
        super().__init__()
        self.c = int(c2 * e)
        self.cv1 = Conv(c1, 2 * self.c, 1, 1)
        self.cv2 = Conv((2 + n) * self.c, c2, 1)
        self.m = nn.ModuleList(
            C3k(self.c, self.c, 2, shortcut, g) if c3k else Bottleneck(self.c, self.c, shortcut, g)
            for _ in range(n))

    def forward(self, x):
        y = list(self.cv1(x).chunk(2, 1))
        y.extend(m(y[-1]) for m in self.m)
        return self.cv2(torch.cat(y, 1))


class SPPF(nn.Module):
    def __init__(self, c1, c2, k=5):
        super().__init__()
        c_ = c1 // 2
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_ * 4, c2, 1, 1)
        self.m = nn.MaxPool2d(kernel_size=k, stride=1, padding=k // 2)

    def forward(self, x):
        y = [self.cv1(x)]
        y.extend(self.m(y[-1]) for _ in range(3))
        return self.cv2(torch.cat(y, 1))


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, attn_ratio=0.5):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.key_dim = int(self.head_dim * attn_ratio)
        self.scale = self.key_dim ** -0.5
        nh_kd = self.key_dim * num_heads
        h = dim + nh_kd * 2
        self.qkv = Conv(dim, h, 1, act=False)
        self.proj = Conv(dim, dim, 1, act=False)
        self.pe = Conv(dim, dim, 3, 1, g=dim, act=False)

    def forward(self, x):
        B, C, H, W = x.shape
        N = H * W
        qkv = self.qkv(x)
        q, k, v = qkv.view(B, self.num_heads, self.key_dim * 2 + self.head_dim, N).split(
            [self.key_dim, self.key_dim, self.head_dim], dim=2)
        attn = (q.transpose(-2, -1) @ k) * self.scale
        attn = attn.softmax(dim=-1)
        x = (v @ attn.transpose(-2, -1)).view(B, C, H, W) + self.pe(v.reshape(B, C, H, W))
        return self.proj(x)


class PSABlock(nn.Module):
    def __init__(self, c, attn_ratio=0.5, num_heads=4, shortcut=True):
        super().__init__()
        self.attn = Attention(c, attn_ratio=attn_ratio, num_heads=num_heads)
        self.ffn = nn.Sequential(Conv(c, c * 2, 1), Conv(c * 2, c, 1, act=False))
        self.add = shortcut

    def forward(self, x):
        x = x + self.attn(x) if self.add else self.attn(x)
        x = x + self.ffn(x) if self.add else self.ffn(x)
        return x


class C2PSA(nn.Module):
    def __init__(self, c1, c2, n=1, e=0.5):
        super().__init__()
        assert c1 == c2
        self.c = int(c1 * e)
        self.cv1 = Conv(c1, 2 * self.c, 1, 1)
        self.cv2 = Conv(2 * self.c, c1, 1)
        self.m = nn.Sequential(*(PSABlock(self.c, attn_ratio=0.5, num_heads=self.c // 64) for _ in range(n)))

    def forward(self, x):
        a, b = self.cv1(x).split((self.c, self.c), dim=1)
        b = self.m(b)
        return self.cv2(torch.cat((a, b), 1))


class DFL(nn.Module):
    def __init__(self, c1=16):
        super().__init__()
        self.conv = nn.Conv2d(c1, 1, 1, bias=False).requires_grad_(False)
        self.conv.weight.data[:] = torch.arange(c1, dtype=torch.float).view(1, c1, 1, 1)
        self.c1 = c1

    def forward(self, x):
        b, _, a = x.shape
        return self.conv(x.view(b, 4, self.c1, a).transpose(2, 1).softmax(1)).view(b, 4, a)


def make_anchors(feats, strides, offset=0.5):
    pts, st = [], []
    for f, s in zip(feats, strides):
        h, w = f.shape[2:]
        sx = torch.arange(w, dtype=torch.float32) + offset
        sy = torch.arange(h, dtype=torch.float32) + offset
        sy, sx = torch.meshgrid(sy, sx, indexing="ij")
        pts.append(torch.stack((sx, sy), -1).view(-1, 2))
        st.append(torch.full((h * w, 1), float(s), dtype=torch.float32))
    return torch.cat(pts), torch.cat(st)


def dist2bbox_xywh(distance, anchor_points):
    lt, rb = distance.chunk(2, 1)
    x1y1 = anchor_points - lt
    x2y2 = anchor_points + rb
    return torch.cat(((x1y1 + x2y2) / 2, x2y2 - x1y1), 1)


class Detect(nn.Module):
    """Non-legacy (YOLO11) Detect head; inference path returns [B, 4+nc, A]."""

    def __init__(self, nc=80, ch=()):
        super().__init__()
        self.nc, self.nl, self.reg_max = nc, len(ch), 16
        self.no = nc + self.reg_max * 4
        self.stride = torch.tensor([8.0, 16.0, 32.0])
        c2, c3 = max(16, ch[0] // 4, self.reg_max * 4), max(ch[0], min(nc, 100))
        self.cv2 = nn.ModuleList(
            nn.Sequential(Conv(x, c2, 3), Conv(c2, c2, 3), nn.Conv2d(c2, 4 * self.reg_max, 1)) for x in ch)
        self.cv3 = nn.ModuleList(
            nn.Sequential(
                nn.Sequential(Conv(x, x, 3, g=x), Conv(x, c3, 1)),
                nn.Sequential(Conv(c3, c3, 3, g=c3), Conv(c3, c3, 1)),
                nn.Conv2d(c3, nc, 1)) for x in ch)
        self.dfl = DFL(self.reg_max)

    def forward(self, x):
        x = [torch.cat((self.cv2[i](x[i]), self.cv3[i](x[i])), 1) for i in range(self.nl)]
        return self.decode(x), x

    def decode(self, x):
        B = x[0].shape[0]
        x_cat = torch.cat([xi.view(B, self.no, -1) for xi in x], 2)
        anchors, strides = (t.transpose(0, 1) for t in make_anchors(x, self.stride, 0.5))
        box, cls = x_cat.split((self.reg_max * 4, self.nc), 1)
        dbox = dist2bbox_xywh(self.dfl(box), anchors.unsqueeze(0)) * strides
        return torch.cat((dbox, cls.sigmoid()), 1)


class YOLO11(nn.Module):
    """24-layer YOLO11 DetectionModel (yolo11.yaml) for a given scale letter and class count."""

    def __init__(self, scale="n", nc=80):
        super().__init__()
        d, w, mc = SCALES[scale]
        c3k_all = scale in "mlx"

        def ch(c):
            return make_divisible(min(c, mc) * w, 8)

        def rep(n):
            return max(round(n * d), 1) if n > 1 else n

        L = []
        L.append(Conv(3, ch(64), 3, 2))                                            # 0
        L.append(Conv(ch(64), ch(128), 3, 2))                                      # 1
        L.append(C3k2(ch(128), ch(256), rep(2), c3k_all, 0.25))                    # 2
        L.append(Conv(ch(256), ch(256), 3, 2))                                     # 3
        L.append(C3k2(ch(256), ch(512), rep(2), c3k_all, 0.25))                    # 4
        L.append(Conv(ch(512), ch(512), 3, 2))                                     # 5
        L.append(C3k2(ch(512), ch(512), rep(2), True))                             # 6
        L.append(Conv(ch(512), ch(1024), 3, 2))                                    # 7
        L.append(C3k2(ch(1024), ch(1024), rep(2), True))                           # 8
        L.append(SPPF(ch(1024), ch(1024), 5))                                      # 9
        L.append(C2PSA(ch(1024), ch(1024), rep(2)))                                # 10
        L.append(nn.Upsample(None, 2, "nearest"))                                  # 11
        L.append(nn.Identity())                                                    # 12 concat [-1, 6]
        L.append(C3k2(ch(1024) + ch(512), ch(512), rep(2), c3k_all))               # 13
        L.append(nn.Upsample(None, 2, "nearest"))                                  # 14
        L.append(nn.Identity())                                                    # 15 concat [-1, 4]
        L.append(C3k2(ch(512) + ch(512), ch(256), rep(2), c3k_all))                # 16
        L.append(Conv(ch(256), ch(256), 3, 2))                                     # 17
        L.append(nn.Identity())                                                    # 18 concat [-1, 13]
        L.append(C3k2(ch(256) + ch(512), ch(512), rep(2), c3k_all))                # 19
        L.append(Conv(ch(512), ch(512), 3, 2))                                     # 20
        L.append(nn.Identity())                                                    # 21 concat [-1, 10]
        L.append(C3k2(ch(512) + ch(1024), ch(1024), rep(2), True))                 # 22
        L.append(Detect(nc, (ch(256), ch(512), ch(1024))))                         # 23
        self.model = nn.ModuleList(L)
        self.nc = nc
        self.scale = scale

    def forward(self, x, return_feats=False):
        m = self.model
        x = m[0](x); x = m[1](x); x = m[2](x); x = m[3](x)
        p3 = m[4](x)
        x = m[5](p3)
        p4 = m[6](x)
        x = m[7](p4); x = m[8](x); x = m[9](x)
        p5 = m[10](x)
        x = m[13](torch.cat((m[11](p5), p4), 1))
        h13 = x
        x = m[16](torch.cat((m[14](x), p3), 1))
        h16 = x
        x = m[19](torch.cat((m[17](x), h13), 1))
        h19 = x
        h22 = m[22](torch.cat((m[20](x), p5), 1))
        y, raw = m[23]([h16, h19, h22])
        if return_feats:
            return y, raw, (h16, h19, h22)
        return y


def count_params(model):
    """Parameter count the way ultralytics reports it (DFL's frozen conv included)."""
    return sum(p.numel() for p in model.parameters())


def randomize_(model, seed=0):
    """Seeded synthetic weights: Kaiming-uniform convs, non-trivial BN statistics."""
    g = torch.Generator().manual_seed(seed)
    for name, mod in model.named_modules():
        if isinstance(mod, nn.Conv2d) and not name.endswith("dfl.conv"):
            fan_in = mod.weight.shape[1] * mod.weight.shape[2] * mod.weight.shape[3]
            bound = math.sqrt(6.0 / fan_in) / math.sqrt(1 + 5.0)  # kaiming_uniform(a=sqrt(5))
            mod.weight.data.uniform_(-bound, bound, generator=g)
            if mod.bias is not None:
                mod.bias.data.uniform_(-1, 1, generator=g).mul_(1.0 / math.sqrt(fan_in))
        elif isinstance(mod, nn.BatchNorm2d):
            mod.weight.data.uniform_(0.5, 1.5, generator=g)
            mod.bias.data.normal_(0, 0.1, generator=g)
            mod.running_mean.data.normal_(0, 0.1, generator=g)
            mod.running_var.data.uniform_(0.5, 1.5, generator=g)
    return model
