"""YOLO11 detector on the cvmi355 kernels: weight packing + launch plan.

What it replaces: the network behind `self.yolo.predict(image)` in the reference
(/root/reference/src/circuit_analyzer.py:45, :268; architecture from un-vendored ultralytics,
SURVEY.md section 8 Table Y).  Parameters are addressed by ultralytics' own state_dict keys
(`model.2.m.0.cv1.conv.weight`, `model.2.m.0.cv1.bn.running_var`, ...), so a real checkpoint's
tensors drop in; BatchNorm is folded into the conv at pack time.

Design notes (MI355X):
  * activations NHWC fp16 (or f32 in parity mode); every Conv+BN+SiLU is ONE implicit-GEMM launch
  * Upsample / Concat / chunk are never materialised: convs read two channel-concatenated,
    optionally 2x-upsampled sources and write into channel slices of the consumer's buffer
  * C3k's parallel cv1/cv2 1x1 convs are one launch (weights stacked along Cout)
  * the whole forward + decode + NMS is captured once per input shape into a HIP graph
"""
import hashlib
import math
import os

import torch

from . import _lib
from ._lib import ACT_NONE, ACT_SILU, F16, F32
from .engine import (ESIZE, Buf, PackedConv, PackedDW, Plan, c3k2_supported, dwpw_supported, make_attn_desc, op_attention, op_c3k2, op_call, op_conv,
                     op_dwpw, op_stem2, stem2_supported,
                     op_dwconv, op_sppf_pool)

SCALES = {"n": (0.50, 0.25, 1024), "s": (0.50, 0.50, 1024), "m": (0.50, 1.00, 512),
          "l": (1.00, 1.00, 512), "x": (1.00, 1.50, 512)}
BN_EPS = 1e-3


# ---- parameter sources ---------------------------------------------------------------------------
class SyntheticParams:
    """Seeded synthetic weights; records everything it hands out so that
    `state_dict()` is a complete ultralytics-keyed checkpoint for the oracle to load."""

    def __init__(self, seed=0, nc=62):
        self.seed, self.nc, self.sd = seed, nc, {}

    def _gen(self, name):
        h = int.from_bytes(hashlib.sha256(f"{self.seed}:{name}".encode()).digest()[:8], "little") & 0x7FFFFFFFFFFFFFFF
        return torch.Generator().manual_seed(h)

    def get(self, name, shape):
        if name in self.sd:
            assert tuple(self.sd[name].shape) == tuple(shape), name
            return self.sd[name]
        g = self._gen(name)
        leaf = name.rsplit(".", 1)[-1]
        if name.endswith("bn.weight"):
            t = torch.empty(shape).uniform_(0.5, 1.5, generator=g)
        elif name.endswith("bn.bias") or name.endswith("bn.running_mean"):
            t = torch.empty(shape).normal_(0, 0.1, generator=g)
        elif name.endswith("bn.running_var"):
            t = torch.empty(shape).uniform_(0.5, 1.5, generator=g)
        elif name.endswith("num_batches_tracked"):
            t = torch.zeros(shape, dtype=torch.long)
        elif leaf == "weight":
            fan_in = shape[1] * shape[2] * shape[3]
            t = torch.empty(shape).uniform_(-1, 1, generator=g) * (1.0 / math.sqrt(fan_in))
            if ".cv3." in name and name.endswith(".2.weight"):       # class logits: spread them out
                t = t * 6.0
        elif leaf == "bias":
            if ".cv3." in name:                                          # sparse detections (~1-3 % of anchors)
                t = torch.empty(shape).normal_(-6.0, 0.5, generator=g)
            else:
                t = torch.empty(shape).normal_(1.0, 0.5, generator=g)    # DFL logits bias
        else:
            raise KeyError(name)
        self.sd[name] = t
        return t

    def state_dict(self):
        return dict(self.sd)


class BlankParams:
    """A rank that holds NO checkpoint: every tensor reads as zeros (BatchNorm variance one), so the packed buffers get their final
    shapes and the real values arrive by `distributed.broadcast_packed` from the rank that read the checkpoint."""

    def get(self, name, shape):
        if name.endswith("num_batches_tracked"):
            return torch.zeros(shape, dtype=torch.long)
        return torch.ones(shape) if name.endswith("running_var") else torch.zeros(shape)


class StateDictParams:
    def __init__(self, sd):
        self.sd = sd

    def get(self, name, shape):
        t = self.sd[name]
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{name}: checkpoint shape {tuple(t.shape)} != expected {tuple(shape)}")
        return t.detach().float().cpu()


def make_divisible(x, d=8):
    return int(math.ceil(x / d) * d)


# ---- packed model ----------------------------------------------------------------------------------
class Yolo11Weights:
    """Walks the YOLO11 graph once, pulling every parameter from `params` and packing it."""

    def __init__(self, scale, nc, params, dtype=F16, device="cuda"):
        self.scale, self.nc, self.dtype, self.device = scale, nc, dtype, device
        self.p = params
        d, w, mc = SCALES[scale]
        self.ch = lambda c: make_divisible(min(c, mc) * w, 8)
        self.rep = lambda n: max(round(n * d), 1) if n > 1 else n
        self.c3k_all = scale in "mlx"
        self.packed = {}
        self.param_bytes = 0
        self._walk()

    # -- folding helpers
    def _fold(self, name, c1, c2, k, g=1):
        w = self.p.get(f"{name}.conv.weight", (c2, c1 // g, k, k))
        gamma = self.p.get(f"{name}.bn.weight", (c2,))
        beta = self.p.get(f"{name}.bn.bias", (c2,))
        mean = self.p.get(f"{name}.bn.running_mean", (c2,))
        var = self.p.get(f"{name}.bn.running_var", (c2,))
        self.p.get(f"{name}.bn.num_batches_tracked", ()) if isinstance(self.p, SyntheticParams) else None
        s = gamma / torch.sqrt(var + BN_EPS)
        return w * s.view(-1, 1, 1, 1), beta - mean * s

    def conv(self, name, c1, c2, k=1):
        w, b = self._fold(name, c1, c2, k)
        self._add(name, PackedConv(w, b, self.dtype, self.device))

    def conv_pair(self, key, name_a, name_b, c1, c2, k=1):
        wa, ba = self._fold(name_a, c1, c2, k)
        wb, bb = self._fold(name_b, c1, c2, k)
        self._add(key, PackedConv(torch.cat((wa, wb), 0), torch.cat((ba, bb), 0), self.dtype, self.device))

    def dwconv(self, name, c):
        w, b = self._fold(name, c, c, 3, g=c)
        self._add(name, PackedDW(w, b, self.dtype, self.device))

    def plain_conv(self, name, c1, c2):
        w = self.p.get(f"{name}.weight", (c2, c1, 1, 1))
        b = self.p.get(f"{name}.bias", (c2,))
        self._add(name, PackedConv(w, b, self.dtype, self.device))

    def _add(self, key, packed):
        self.packed[key] = packed
        self.param_bytes += packed.param_bytes

    # -- block walkers (mirror the plan builder below)
    def bottleneck(self, name, c, e):
        c_ = int(c * e)
        self.conv(f"{name}.cv1", c, c_, 3)
        self.conv(f"{name}.cv2", c_, c, 3)

    def c3k(self, name, c):
        c_ = int(c * 0.5)
        self.conv_pair(f"{name}.cv12", f"{name}.cv1", f"{name}.cv2", c, c_, 1)
        for i in range(2):
            self.bottleneck(f"{name}.m.{i}", c_, 1.0)
        self.conv(f"{name}.cv3", 2 * c_, c, 1)

    def c3k2(self, name, c1, c2, n, c3k, e=0.5):
        c = int(c2 * e)
        self.conv(f"{name}.cv1", c1, 2 * c, 1)
        for i in range(n):
            if c3k:
                self.c3k(f"{name}.m.{i}", c)
            else:
                self.bottleneck(f"{name}.m.{i}", c, 0.5)
        self.conv(f"{name}.cv2", (2 + n) * c, c2, 1)

    def stem(self, name, c2):
        """Layer 0 (3x3, stride 2, 3 channels) re-expressed on the space-to-depth(2) input the letterbox kernel writes:
        a 2x2 / stride-1 conv over 16 channels (12 real), taps (ty, tx) = block offsets (-1, 0).  Window row ky sits in
        block row ty with sub-row sy: ky=0 -> (0,1), ky=1 -> (1,0), ky=2 -> (1,1); same for columns."""
        w, b = self._fold(name, 3, c2, 3)
        w2 = torch.zeros(c2, 16, 2, 2)
        m = {0: (0, 1), 1: (1, 0), 2: (1, 1)}
        for ky in range(3):
            for kx in range(3):
                (ty, sy), (tx, sx) = m[ky], m[kx]
                w2[:, (sy * 2 + sx) * 3:(sy * 2 + sx) * 3 + 3, ty, tx] = w[:, :, ky, kx]
        self._add(name, PackedConv(w2, b, self.dtype, self.device))

    def _walk(self):
        ch, rep, ca = self.ch, self.rep, self.c3k_all
        self.stem("model.0", ch(64))
        self.conv("model.1", ch(64), ch(128), 3)
        self.c3k2("model.2", ch(128), ch(256), rep(2), ca, 0.25)
        self.conv("model.3", ch(256), ch(256), 3)
        self.c3k2("model.4", ch(256), ch(512), rep(2), ca, 0.25)
        self.conv("model.5", ch(512), ch(512), 3)
        self.c3k2("model.6", ch(512), ch(512), rep(2), True)
        self.conv("model.7", ch(512), ch(1024), 3)
        self.c3k2("model.8", ch(1024), ch(1024), rep(2), True)
        c = ch(1024)
        self.conv("model.9.cv1", c, c // 2, 1)
        self.conv("model.9.cv2", 2 * c, c, 1)
        # C2PSA
        cp = c // 2
        self.conv("model.10.cv1", c, 2 * cp, 1)
        self.conv("model.10.cv2", 2 * cp, c, 1)
        nh = cp // 64
        for i in range(rep(2)):
            a = f"model.10.m.{i}.attn"
            self.conv(f"{a}.qkv", cp, cp + nh * 32 * 2, 1)
            self.conv(f"{a}.proj", cp, cp, 1)
            # positional depthwise conv acts on v, one 64-channel group per head
            w, b = self._fold(f"{a}.pe", cp, cp, 3, g=cp)
            for h in range(nh):
                self._add(f"{a}.pe.h{h}", PackedDW(w[h * 64:(h + 1) * 64], b[h * 64:(h + 1) * 64], self.dtype, self.device))
            self.conv(f"model.10.m.{i}.ffn.0", cp, 2 * cp, 1)
            self.conv(f"model.10.m.{i}.ffn.1", 2 * cp, cp, 1)
        self.c3k2("model.13", ch(1024) + ch(512), ch(512), rep(2), ca)
        self.c3k2("model.16", ch(512) + ch(512), ch(256), rep(2), ca)
        self.conv("model.17", ch(256), ch(256), 3)
        self.c3k2("model.19", ch(256) + ch(512), ch(512), rep(2), ca)
        self.conv("model.20", ch(512), ch(512), 3)
        self.c3k2("model.22", ch(512) + ch(1024), ch(1024), rep(2), True)
        # Detect
        chs = (ch(256), ch(512), ch(1024))
        self.det_ch = chs
        self.det_c2 = max(16, chs[0] // 4, 64)
        self.det_c3 = max(chs[0], min(self.nc, 100))
        for i, x in enumerate(chs):
            self.conv(f"model.23.cv2.{i}.0", x, self.det_c2, 3)
            self.conv(f"model.23.cv2.{i}.1", self.det_c2, self.det_c2, 3)
            self.plain_conv(f"model.23.cv2.{i}.2", self.det_c2, 64)
            self.dwconv(f"model.23.cv3.{i}.0.0", x)
            self.conv(f"model.23.cv3.{i}.0.1", x, self.det_c3, 1)
            self.dwconv(f"model.23.cv3.{i}.1.0", self.det_c3)
            self.conv(f"model.23.cv3.{i}.1.1", self.det_c3, self.det_c3, 1)
            self.plain_conv(f"model.23.cv3.{i}.2", self.det_c3, self.nc)
        if isinstance(self.p, SyntheticParams):
            self.p.sd["model.23.dfl.conv.weight"] = torch.arange(16, dtype=torch.float32).view(1, 16, 1, 1)


class Yolo11Plan:
    """Launch plan of the full forward for one (B, H, W): input NHWC [B,H,W,3] -> pred f32 [B,4+nc,A]
    -> NMS outputs."""

    def __init__(self, weights, B, H, W, stream, conf=0.25, iou=0.7, max_det=300, with_nms=True, keep_scores=True, fuse_c3k2=True, lanes=None):
        """keep_scores=False: the decode kernel skips the class-score rows of `pred` (NMS takes the per-anchor
        best class straight from the decode kernel); the detections are identical."""
        self.keep_scores = keep_scores
        self.fuse_c3k2 = fuse_c3k2 and os.environ.get("CVMI_FUSE_C3K2", "1") != "0"
        self.fuse_dwpw = fuse_c3k2 and os.environ.get("CVMI_FUSE_DWPW", "1") != "0"
        self.fuse_stem = fuse_c3k2 and os.environ.get("CVMI_FUSE_STEM", "1") != "0"
        assert H % 32 == 0 and W % 32 == 0, "network input must be a multiple of stride 32"
        self.wt, self.B, self.H, self.W = weights, B, H, W
        self.dt, self.dev = weights.dtype, weights.device
        self.plan = Plan(stream)
        self.act_bytes = 0
        self.conf, self.iou, self.max_det = conf, iou, max_det
        self.lanes = int(os.environ.get("CVMI_YOLO_LANES", "3")) if lanes is None else int(lanes)      # side lanes of the captured graph (head())
        self._build(with_nms)
        torch.cuda.synchronize()      # buffer fills / weight uploads ran on the default stream

    # -- helpers
    def buf(self, H, W, C):
        b = Buf(self.B, H, W, C, self.dt, self.dev)
        self.act_bytes += b.nbytes
        return b

    def cv(self, name, srcs, dst, k=1, s=1, act=ACT_SILU, res=None, scalar=False, kind="conv"):
        if not isinstance(srcs, list):
            srcs = [(srcs, 0)]
        op_conv(self.plan, name, self.wt.packed[name], srcs, dst, stride=s, act=act, res=res, scalar_gather=scalar, kind=kind)

    def bottleneck(self, name, src, dst, c, e):
        t = self.buf(src.H, src.W, int(c * e)).view()
        self.cv(f"{name}.cv1", src, t, 3)
        self.cv(f"{name}.cv2", t, dst, 3, res=src)

    def c3k(self, name, src, dst, c):
        c_ = c // 2
        Z = self.buf(src.H, src.W, 2 * c_)
        self.cv(f"{name}.cv12", src, Z.view())
        t = self.buf(src.H, src.W, c_).view()
        self.bottleneck(f"{name}.m.0", Z.view(0, c_), t, c_, 1.0)
        self.bottleneck(f"{name}.m.1", t, Z.view(0, c_), c_, 1.0)
        self.cv(f"{name}.cv3", Z.view(), dst)

    def c3k2(self, name, srcs, c2, n, c3k, e=0.5, dst=None):
        v0, up0 = srcs[0]
        H, W = v0.H << up0, v0.W << up0
        c = int(c2 * e)
        if dst is None:
            dst = self.buf(H, W, c2).view()
        # small-channel blocks: one fused launch (c3k2_fused.hip), with cv1 folded in when the block has a single source
        if self.fuse_c3k2 and not c3k and n == 1:
            h = int(c * 0.5)
            pk = self.wt.packed
            one_src = len(srcs) == 1 and up0 == 0
            args = (pk[f"{name}.cv1"], pk[f"{name}.m.0.cv1"], pk[f"{name}.m.0.cv2"], pk[f"{name}.cv2"], c, h)
            if one_src and c3k2_supported(v0.c, c, h, c2, True, self.dt):
                op_c3k2(self.plan, name, v0, dst, *args, fuse_cv1=True)
                return dst
            if c3k2_supported(0, c, h, c2, False, self.dt):
                Y = self.buf(H, W, 2 * c)
                self.cv(f"{name}.cv1", srcs, Y.view())
                op_c3k2(self.plan, f"{name}.fused", Y.view(), dst, *args, fuse_cv1=False)
                return dst
        Y = self.buf(H, W, (2 + n) * c)
        self.cv(f"{name}.cv1", srcs, Y.view(0, 2 * c))
        for i in range(n):
            src, out = Y.view((1 + i) * c, c), Y.view((2 + i) * c, c)
            if c3k:
                self.c3k(f"{name}.m.{i}", src, out, c)
            else:
                self.bottleneck(f"{name}.m.{i}", src, out, c, 0.5)
        self.cv(f"{name}.cv2", Y.view(), dst)
        return dst

    def _build(self, with_nms):
        wt, ch, rep, ca = self.wt, self.wt.ch, self.wt.rep, self.wt.c3k_all
        B, H, W = self.B, self.H, self.W
        self.x_in = Buf(B, H // 2, W // 2, 16, self.dt, self.dev, zero=True)       # space-to-depth(2) image, see Yolo11Weights.stem
        y = self.buf(H // 4, W // 4, ch(128)).view()
        if self.fuse_stem and stem2_supported(ch(64), ch(128), self.dt):
            op_stem2(self.plan, "model.0-1", wt.packed["model.0"], wt.packed["model.1"], self.x_in.view(), y)
        else:
            x = self.buf(H // 2, W // 2, ch(64)).view()
            op_conv(self.plan, "model.0", wt.packed["model.0"], [(self.x_in.view(), 0)], x, stride=1, pad=1, act=ACT_SILU, out_hw=(H // 2, W // 2), kind="stem")
            self.cv("model.1", x, y, 3, 2)
        x = self.c3k2("model.2", [(y, 0)], ch(256), rep(2), ca, 0.25)
        y = self.buf(H // 8, W // 8, ch(256)).view()
        self.cv("model.3", x, y, 3, 2)
        p3 = self.c3k2("model.4", [(y, 0)], ch(512), rep(2), ca, 0.25)
        y = self.buf(H // 16, W // 16, ch(512)).view()
        self.cv("model.5", p3, y, 3, 2)
        p4 = self.c3k2("model.6", [(y, 0)], ch(512), rep(2), True)
        y = self.buf(H // 32, W // 32, ch(1024)).view()
        self.cv("model.7", p4, y, 3, 2)
        x = self.c3k2("model.8", [(y, 0)], ch(1024), rep(2), True)
        # SPPF
        c = ch(1024)
        S = self.buf(H // 32, W // 32, 2 * c)
        self.cv("model.9.cv1", x, S.view(0, c // 2))
        op_sppf_pool(self.plan, "model.9.pool", S, c // 2)
        x = self.buf(H // 32, W // 32, c).view()
        self.cv("model.9.cv2", S.view(), x)
        p5 = self._c2psa("model.10", x, c, rep(2))
        # head
        h13 = self.c3k2("model.13", [(p5, 1), (p4, 0)], ch(512), rep(2), ca)
        # Each Detect level is forked onto side lanes as soon as its feature map exists, so the long 80 x 80 chains run
        # beside model.17 .. model.22 (small grids that leave most CUs idle) instead of after them.
        self._det_boxes, self._det_cls = [], []
        h16 = self.c3k2("model.16", [(h13, 1), (p3, 0)], ch(256), rep(2), ca)
        self._detect_level(0, h16)
        y = self.buf(H // 16, W // 16, ch(256)).view()
        self.cv("model.17", h16, y, 3, 2)
        h19 = self.c3k2("model.19", [(y, 0), (h13, 0)], ch(512), rep(2), ca)
        self._detect_level(1, h19)
        y = self.buf(H // 32, W // 32, ch(512)).view()
        self.cv("model.20", h19, y, 3, 2)
        h22 = self.c3k2("model.22", [(y, 0), (p5, 0)], ch(1024), rep(2), True)
        self._detect_level(2, h22)
        self.feats = (h16, h19, h22)
        self._detect([h16, h19, h22], with_nms)

    def _c2psa(self, name, x, c, n):
        cp = c // 2
        nh = cp // 64
        kd, hd = 32, 64
        Hh, Ww = x.H, x.W
        N = Hh * Ww
        Y = self.buf(Hh, Ww, 2 * cp)
        self.cv(f"{name}.cv1", x, Y.view())
        b = Y.view(cp, cp)
        per = 2 * kd + hd
        for i in range(n):
            a = f"{name}.m.{i}.attn"
            QKV = self.buf(Hh, Ww, nh * per)
            self.cv(f"{a}.qkv", b, QKV.view(), act=ACT_NONE)
            AO = self.buf(Hh, Ww, cp)
            es = ESIZE[self.dt]
            base = QKV.t.data_ptr()
            desc = make_attn_desc(
                q=base, k=base + kd * es, v=base + 2 * kd * es, o=AO.t.data_ptr(),
                q_sb=N * QKV.C, q_sh=per, q_st=QKV.C, k_sb=N * QKV.C, k_sh=per, k_st=QKV.C,
                v_sb=N * QKV.C, v_sh=per, v_st=QKV.C, o_sb=N * cp, o_sh=hd, o_st=cp,
                B=self.B, heads=nh, Nq=N, Nk=N, dqk=kd, dv=hd, scale=kd ** -0.5, dtype=self.dt,
                win=0, grid_h=0, grid_w=0, q_pool=0)
            op_attention(self.plan, f"{a}.sdpa", desc, (QKV, AO), bytes_=(QKV.nbytes + AO.nbytes),
                         flops=2 * self.B * nh * N * N * (kd + hd))
            for h in range(nh):
                op_dwconv(self.plan, f"{a}.pe.h{h}", self.wt.packed[f"{a}.pe.h{h}"], QKV.view(h * per + 2 * kd, hd),
                          AO.view(h * hd, hd), act=ACT_NONE, res=AO.view(h * hd, hd))
            self.cv(f"{a}.proj", AO.view(), b, act=ACT_NONE, res=b)
            F = self.buf(Hh, Ww, 2 * cp)
            self.cv(f"{name}.m.{i}.ffn.0", b, F.view())
            self.cv(f"{name}.m.{i}.ffn.1", F.view(), b, act=ACT_NONE, res=b)
        out = self.buf(Hh, Ww, c).view()
        self.cv(f"{name}.cv2", Y.view(), out)
        return out

    def _detect_level(self, i, f):
        """Detect branches of level i (box: 3x3, 3x3, 1x1; cls: dw3x3 + 1x1 twice, 1x1) on two side lanes."""
        wt = self.wt
        nc, c2, c3 = wt.nc, wt.det_c2, wt.det_c3
        ncp = (nc + 7) // 8 * 8
        # Side lanes of the captured graph.  Measured (B = 32, ms per step): every branch on its own lane (6 lanes) 1.16, one lane
        # per level 1.10, ONE side lane for all heads 1.09, no side lane 1.16 -- a replayed HIP graph pays ~1.8 us per node on a
        # chain but ~3.3 us per node when nodes alternate between lanes (tools/graph_gap.py), so the heads share one lane that
        # runs beside the rest of the neck.  CVMI_YOLO_LANES: tuning experiments only (0 none, 1 per level, 2 per branch, 3 one).
        lanes = self.lanes              # (Yolo11Plan(lanes=0): one linear chain -- for a step whose graphs run side by side on their own streams)
        if lanes:
            self.plan.fork()
            self.plan.lane(1 if lanes in (3, 5) else (1 + min(i, 1)) if lanes == 4 else 2 * i + 1)
        t1 = self.buf(f.H, f.W, c2).view()
        t2 = self.buf(f.H, f.W, c2).view()
        bx = self.buf(f.H, f.W, 64).view()
        self.cv(f"model.23.cv2.{i}.0", f, t1, 3, kind="head")
        self.cv(f"model.23.cv2.{i}.1", t1, t2, 3, kind="head")
        self.cv(f"model.23.cv2.{i}.2", t2, bx, act=ACT_NONE, kind="head")
        if lanes == 2:
            self.plan.lane(2 * i + 2)
        elif lanes == 5 and i == 2:
            self.plan.lane(2)                # the last level is the serial tail of the step: its two branches side by side
        cl = Buf(self.B, f.H, f.W, ncp, self.dt, self.dev, zero=True)
        self.act_bytes += cl.nbytes
        u1 = self.buf(f.H, f.W, c3).view()
        pk = wt.packed
        if self.fuse_dwpw and dwpw_supported(f.c, c3, 0, self.dt) and dwpw_supported(c3, c3, nc, self.dt):
            # class branch in two launches: [dw3x3 + 1x1], [dw3x3 + 1x1 + class 1x1]; the intermediates stay in LDS
            op_dwpw(self.plan, f"model.23.cv3.{i}.0", pk[f"model.23.cv3.{i}.0.0"], pk[f"model.23.cv3.{i}.0.1"], f, u1)
            op_dwpw(self.plan, f"model.23.cv3.{i}.1-2", pk[f"model.23.cv3.{i}.1.0"], pk[f"model.23.cv3.{i}.1.1"], u1, cl.view(0, nc),
                    pc2=pk[f"model.23.cv3.{i}.2"])
        else:
            d1 = self.buf(f.H, f.W, f.c).view()
            op_dwconv(self.plan, f"model.23.cv3.{i}.0.0", pk[f"model.23.cv3.{i}.0.0"], f, d1, act=ACT_SILU)
            self.cv(f"model.23.cv3.{i}.0.1", d1, u1, kind="head")
            d2 = self.buf(f.H, f.W, c3).view()
            op_dwconv(self.plan, f"model.23.cv3.{i}.1.0", pk[f"model.23.cv3.{i}.1.0"], u1, d2, act=ACT_SILU)
            u2 = self.buf(f.H, f.W, c3).view()
            self.cv(f"model.23.cv3.{i}.1.1", d2, u2, kind="head")
            self.cv(f"model.23.cv3.{i}.2", u2, cl.view(0, nc), act=ACT_NONE, kind="head")
        self._det_boxes.append(bx)
        self._det_cls.append(cl)
        self.plan.lane(0)

    def _detect(self, feats, with_nms):
        import ctypes as C
        wt, lib = self.wt, _lib.load()
        nc = wt.nc
        boxes, clss = self._det_boxes, self._det_cls
        self.plan.join()
        self.box_bufs, self.cls_bufs = boxes, clss
        A = sum(f.H * f.W for f in feats)
        self.A = A
        self.pred = torch.empty(self.B, 4 + nc, A, dtype=torch.float32, device=self.dev)
        nl = len(feats)
        box_p = (C.c_void_p * nl)(*[b.ptr for b in boxes])
        cls_p = (C.c_void_p * nl)(*[c.t.data_ptr() for c in clss])
        box_ld = (C.c_int * nl)(*[b.ld for b in boxes])
        cls_ld = (C.c_int * nl)(*[c.C for c in clss])
        hs = (C.c_int * nl)(*[f.H for f in feats])
        ws = (C.c_int * nl)(*[f.W for f in feats])
        strides = (C.c_float * nl)(*[float(self.H // f.H) for f in feats])
        self.plan.keep.append((box_p, cls_p, box_ld, cls_ld, hs, ws, strides))
        sp, dt, pred_ptr, Bn = self.plan.sptr, self.dt, self.pred.data_ptr(), self.B
        self.best_score = torch.zeros(Bn * A, dtype=torch.float32, device=self.dev)
        self.best_cls = torch.zeros(Bn * A, dtype=torch.int32, device=self.dev)
        bs_ptr, bc_ptr, wcls = self.best_score.data_ptr(), self.best_cls.data_ptr(), 1 if self.keep_scores else 0
        if not self.keep_scores:
            self.pred.zero_()

        def decode(sp=sp):
            _lib.check(lib.cvmi_detect_decode(box_p, box_ld, cls_p, cls_ld, hs, ws, strides, nl, Bn, nc, dt, pred_ptr, bs_ptr, bc_ptr, wcls, sp), "detect_decode")

        es = ESIZE[self.dt]
        self.plan.add("model.23.decode", "decode", decode, Bn * A * ((64 + nc) * es + (4 + (nc if wcls else 2)) * 4), 0)
        if with_nms:
            self.det = torch.zeros(Bn, self.max_det, 6, dtype=torch.float32, device=self.dev)
            self.det_idx = torch.zeros(Bn, self.max_det, dtype=torch.int32, device=self.dev)
            self.det_count = torch.zeros(Bn, dtype=torch.int32, device=self.dev)
            self.nms_ws = torch.empty(lib.cvmi_yolo_nms_workspace(Bn, A), dtype=torch.uint8, device=self.dev)
            a = (pred_ptr, bs_ptr, bc_ptr, Bn, nc, A, float(self.conf), float(self.iou), int(self.max_det), 7680.0, self.det.data_ptr(),
                 self.det_idx.data_ptr(), self.det_count.data_ptr(), self.nms_ws.data_ptr())

            def nms(sp=sp):
                _lib.check(lib.cvmi_yolo_nms_best(*a, sp), "yolo_nms")

            self.plan.add("nms", "nms", nms, Bn * A * 6 * 4, 0)

    def set_input_nchw(self, x):
        """x: [B,3,H,W] float tensor (already letterboxed, RGB-flipped, /255) -> the plan's space-to-depth input."""
        B, _, H, W = x.shape
        t = x.to(self.dev).reshape(B, 3, H // 2, 2, W // 2, 2).permute(0, 2, 4, 3, 5, 1).reshape(B, H // 2, W // 2, 12)
        self.x_in.t.zero_()
        self.x_in.t[..., :12] = t.to(self.x_in.t.dtype)

    def input_nchw(self):
        B, H2, W2, _ = self.x_in.t.shape
        t = self.x_in.t[..., :12].float().reshape(B, H2, W2, 2, 2, 3).permute(0, 5, 1, 3, 2, 4).reshape(B, 3, H2 * 2, W2 * 2)
        return t

    # -- accounting for the roofline line (layer-granular minimum traffic)
    def conv_stack_bytes(self):
        return sum(b for _, kind, _, b, _ in self.plan.ops if kind in ("conv", "stem", "head", "dwconv", "pool", "attention"))

    def flops(self):
        return sum(f for *_, f in self.plan.ops)
