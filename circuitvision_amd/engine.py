"""Launch-plan engine: NHWC buffers, op wrappers over the C ABI, HIP-graph capture.

PyTorch is used only as plumbing (device memory, streams, events).  A `Plan` is a static list of
kernel launches over pre-allocated buffers for one input shape; it runs either launch by launch
(parity tests, per-kernel timing) or as one captured HIP graph (throughput path).
"""
import ctypes as C
import os
import threading

import torch

from . import _lib
from ._lib import BF16, F16, F32, AttnDesc, C3k2Desc, ConvDesc

TORCH_DTYPE = {F16: torch.float16, F32: torch.float32, BF16: torch.bfloat16}
ESIZE = {F16: 2, F32: 4, BF16: 2}
DTYPE_OF = {torch.float16: F16, torch.float32: F32, torch.bfloat16: BF16}


def is16(dtype):
    """16-bit operand storage (fp16 or bf16): fp32 accumulation, f32 residual streams beside it."""
    return dtype in (F16, BF16)


def require_gpu():
    if not torch.cuda.is_available():
        raise _lib.CvmiError("circuitvision_amd needs an MI355X (gfx950) GPU: torch.cuda.is_available() is False; "
                             "there is no CPU fallback")
    _lib.load()


class Buf:
    """NHWC activation buffer [B, H, W, C] of one dtype."""

    def __init__(self, B, H, W, C, dtype, device="cuda", zero=False):
        self.B, self.H, self.W, self.C, self.dtype = B, H, W, C, dtype
        alloc = torch.zeros if zero else torch.empty
        self.t = alloc((B, H, W, C), dtype=TORCH_DTYPE[dtype], device=device)

    def view(self, c0=0, c=None):
        return View(self, c0, self.C - c0 if c is None else c)

    def images(self, b0, nb):
        """Images [b0, b0 + nb) of this buffer as a Buf that shares its storage."""
        assert 0 <= b0 and b0 + nb <= self.B
        o = Buf.__new__(Buf)
        o.B, o.H, o.W, o.C, o.dtype = nb, self.H, self.W, self.C, self.dtype
        o.t = self.t[b0:b0 + nb]
        return o

    @property
    def nbytes(self):
        return self.t.numel() * self.t.element_size()


class View:
    """Channel slice [c0, c0+c) of a Buf: (ptr, ld) as the C ABI wants it."""

    def __init__(self, buf, c0, c):
        assert 0 <= c0 and c0 + c <= buf.C, (c0, c, buf.C)
        self.buf, self.c0, self.c = buf, c0, c

    @property
    def ptr(self):
        return self.buf.t.data_ptr() + self.c0 * ESIZE[self.buf.dtype]

    @property
    def ld(self):
        return self.buf.C

    B = property(lambda s: s.buf.B)
    H = property(lambda s: s.buf.H)
    W = property(lambda s: s.buf.W)
    dtype = property(lambda s: s.buf.dtype)

    def tensor(self):
        return self.buf.t[..., self.c0:self.c0 + self.c]


class PackedConv:
    """Weights of one fused conv / linear layer in the layout cvmi_conv2d reads:
    w [Npad][Kpad] (k = (ky*KW + kx)*Cin + c), bias f32 [Npad]."""

    def __init__(self, weight, bias, dtype, device="cuda"):
        # weight: [N, Cin, KH, KW] float32 (BN already folded), bias: [N] or None
        N, Cin, KH, KW = weight.shape
        K = Cin * KH * KW
        Npad = (N + 127) // 128 * 128
        Kpad = (K + 63) // 64 * 64
        w = torch.zeros(Npad, Kpad, dtype=torch.float32)
        w[:N, :K] = weight.permute(0, 2, 3, 1).reshape(N, K)
        b = torch.zeros(Npad, dtype=torch.float32)
        if bias is not None:
            b[:N] = bias
        self.w = w.to(TORCH_DTYPE[dtype]).to(device)
        self.bias = b.to(device)
        self.N, self.Cin, self.KH, self.KW, self.K, self.Kpad, self.dtype = N, Cin, KH, KW, K, Kpad, dtype
        self.param_bytes = N * K * ESIZE[dtype]


class PackedDW:
    """Depthwise 3x3 weights: w [9][C] tap-major, bias f32 [C]."""

    def __init__(self, weight, bias, dtype, device="cuda"):
        C_ = weight.shape[0]
        assert weight.shape[1:] == (1, 3, 3)
        self.w = weight.reshape(C_, 9).t().contiguous().to(TORCH_DTYPE[dtype]).to(device)
        self.bias = (bias if bias is not None else torch.zeros(C_)).float().to(device)
        self.C, self.dtype = C_, dtype
        self.param_bytes = C_ * 9 * ESIZE[dtype]


class Plan:
    """Static launch list.  Each entry: (label, kind, thunk, algorithmic_bytes, flops)."""

    def __init__(self, stream):
        self.stream = stream
        self.ops = []
        self.keep = []          # objects that must outlive the plan (descriptors, buffers)
        self.graph = None
        self.lanes, self.cur_lane = [], 0
        self._lane_streams = {}
        self._lock = threading.Lock()

    @property
    def sptr(self):
        return self.stream.cuda_stream

    def add(self, label, kind, thunk, bytes_=0, flops=0):
        self.ops.append((label, kind, thunk, bytes_, flops))
        self.lanes.append(self.cur_lane)

    # ---- independent chains: ops added between fork(n) and join() carry a lane id; under graph capture each lane
    #      is launched on its own stream (fork / join through events), eager runs keep everything on one stream
    def fork(self):
        self.add("fork", "sync", lambda sp=None: None)
        self.lanes[-1] = -1

    def lane(self, i):
        self.cur_lane = i

    def join(self):
        self.cur_lane = 0
        self.add("join", "sync", lambda sp=None: None)
        self.lanes[-1] = -2

    # ---- execution ---------------------------------------------------------------------------
    def run_eager(self):
        for _, _, thunk, _, _ in self.ops:
            thunk()

    def capture(self):
        lib = _lib.load()
        self.run_eager()                       # warm: first-use hipFuncSetAttribute etc. outside capture
        self.stream.synchronize()
        _lib.check(lib.cvmi_graph_begin(self.sptr), "graph_begin")
        try:
            self._run_lanes()
        finally:
            g = C.c_void_p()
            rc = lib.cvmi_graph_end(self.sptr, C.byref(g))
        _lib.check(rc, "graph_end")
        self.graph = g

    def _run_lanes(self):
        """Issue the ops for capture: lane 0 on the plan's stream, other lanes on side streams between fork/join."""
        active, lane_fork = {}, {}
        for (label, kind, thunk, _, _), lane in zip(self.ops, self.lanes):
            if lane == -1:                                   # fork point: lanes that START after it wait for it;
                self._fork_ev = torch.cuda.Event()           # lanes forked earlier keep running until the join
                self._fork_ev.record(self.stream)
            elif lane == -2:                                 # join
                for st in active.values():
                    ev = torch.cuda.Event()
                    ev.record(st)
                    self.stream.wait_event(ev)
                active, lane_fork = {}, {}
            elif lane == 0:
                thunk()
            else:
                if lane not in self._lane_streams:
                    self._lane_streams[lane] = torch.cuda.Stream(device=self.stream.device)
                st = self._lane_streams[lane]
                if lane_fork.get(lane) is not self._fork_ev:     # first op of this lane after the latest fork point (a lane may be
                    st.wait_event(self._fork_ev)                 # reused after a later fork: it then also waits for that one)
                    lane_fork[lane] = self._fork_ev
                    active[lane] = st
                thunk(st.cuda_stream)

    def run(self):
        if self.graph is None:
            self.capture()
        _lib.check(_lib.load().cvmi_graph_launch(self.graph, self.sptr), "graph_launch")

    def timed_eager(self, with_kernels=False):
        """One eager pass with an event pair around every launch (on the plan's stream).
        Returns [(label, kind, ms, bytes, flops)]; with_kernels=True appends the kernel the library dispatched to
        (cvmi_last_kernel: "" for launches whose dispatcher does not tag itself)."""
        lib = _lib.load()
        lib.cvmi_last_kernel()
        evs = []
        for label, kind, thunk, b, f in self.ops:
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record(self.stream)
            thunk()
            e1.record(self.stream)
            evs.append((label, kind, e0, e1, b, f, lib.cvmi_last_kernel().decode()))
        self.stream.synchronize()
        if with_kernels:
            return [(l, k, e0.elapsed_time(e1), b, f, kn) for l, k, e0, e1, b, f, kn in evs]
        return [(l, k, e0.elapsed_time(e1), b, f) for l, k, e0, e1, b, f, _ in evs]

    def __del__(self):
        # A captured graph WITH side lanes is never destroyed.  Measured on this runtime (ROCm 7.2, r04): hipGraphExecDestroy of one
        # multi-branch graph leaves ANOTHER multi-branch graph exec with dangling parallel-stream pointers -- its next hipGraphLaunch
        # dies in hip::Graph::UpdateStreams (rocgdb backtrace; order-dependent: a detector plan collected between two tests killed a
        # later detector's replay).  Linear graphs (no internal streams) are destroyed as before; the others are parked until the
        # process exits -- a few KB each, one per (model, input shape).
        try:
            if self.graph is not None:
                if self._lane_streams or os.environ.get("CVMI_KEEP_GRAPHS") == "1":
                    _PARKED_GRAPHS.append((self.graph, self._lane_streams))
                else:
                    _lib.load().cvmi_graph_destroy(self.graph)
        except Exception:
            pass


_PARKED_GRAPHS = []


# ---- op wrappers (each appends one launch to a plan) --------------------------------------------
def row_stats_supported(M, N, K):
    """True when cvmi_conv2d sends a plain 16-bit GEMM [M, K] x [K, N] with f32 output + residual to the 256 x 192 kernel, the one that can write
    cvmi_conv_desc.row_stats (mirrors launch_typed in igemm.hip: N a multiple of 192 that 256-wide tiles would waste, K >= 1024, at least one
    tile per CU and >= 75 % of the last round of tiles used)."""
    if N % 192 or N < 384 or K < 1024 or K % 64:
        return False
    t256 = -(-M // 256) * -(-N // 256)
    if t256 >= 256 and N / (-(-N // 256) * 256) >= 0.8 and t256 / (-(-t256 // 256) * 256) >= 0.8:
        return False                                  # the 256 x 256 kernel takes it
    t192 = -(-M // 256) * (N // 192)
    return t192 >= 256 and t192 / (-(-t192 // 256) * 256) >= 0.75


def op_conv(plan, label, pc, srcs, dst, stride=1, pad=None, act=_lib.ACT_NONE, res=None, out_hw=None,
            scalar_gather=False, kind="conv", res_mod=0, act_after_res=False, shuffle_cout=0, res_rep=0, row_stats=None):
    """srcs: [(View, up)] (1 or 2 channel-concatenated sources).  dst / res: View."""
    lib = _lib.load()
    (v0, up0) = srcs[0]
    v1, up1 = (srcs[1] if len(srcs) > 1 else (None, 0))
    H, W = v0.H << up0, v0.W << up0
    if v1 is not None:
        assert (v1.H << up1, v1.W << up1) == (H, W), "concatenated sources disagree on size"
    ctot = v0.c + (v1.c if v1 is not None else 0)
    assert ctot == pc.Cin, (label, ctot, pc.Cin)
    pad = pc.KH // 2 if pad is None else pad
    OH = (H + 2 * pad - pc.KH) // stride + 1
    OW = (W + 2 * pad - pc.KW) // stride + 1
    if out_hw is not None:                     # explicit output size (asymmetric padding: fewer rows/cols than the formula)
        assert out_hw[0] <= OH and out_hw[1] <= OW
        OH, OW = out_hw
    if shuffle_cout:
        assert (dst.B, dst.H, dst.W) == (v0.B, 2 * OH, 2 * OW) and dst.c == shuffle_cout and pc.N == 4 * shuffle_cout, label
    else:
        assert (dst.B, dst.H, dst.W) == (v0.B, OH, OW) and dst.c == pc.N, (label, (dst.B, dst.H, dst.W, dst.c), (v0.B, OH, OW, pc.N))
    if res_rep > 1:
        assert res is not None and res.B * res_rep == v0.B, (label, "res_rep needs a per-image residual")
        assert shuffle_cout or res_mod == dst.H * dst.W, (label, "res_rep: ConvTranspose scatter or res_mod = rows per image")
    out_f32 = 1 if (dst.dtype == F32 and is16(pc.dtype)) else 0
    d = ConvDesc(
        x0=v0.ptr, x1=(v1.ptr if v1 is not None else None), w=pc.w.data_ptr(), bias=pc.bias.data_ptr(),
        res=(res.ptr if res is not None else None), y=dst.ptr,
        x0_ld=v0.ld, x1_ld=(v1.ld if v1 is not None else 0), res_ld=(res.ld if res is not None else 0), y_ld=dst.ld,
        c0=v0.c, c1=(v1.c if v1 is not None else 0), up0=up0, up1=up1,
        B=v0.B, H=H, W=W, OH=OH, OW=OW, KH=pc.KH, KW=pc.KW, stride=stride, pad=pad,
        N=pc.N, Kpad=pc.Kpad, act=act, dtype=pc.dtype, out_f32=out_f32, scalar_gather=1 if scalar_gather else 0,
        res_mod=res_mod, act_after_res=1 if act_after_res else 0, shuffle_cout=shuffle_cout, res_rep=res_rep,
        row_stats=(row_stats.data_ptr() if row_stats is not None else None))
    if row_stats is not None:       # f32 [rows, N / 96, 2]: (mean, sum of squared deviations) per 96-column slice of every written row (Hiera stage-3 fc2 shape only)
        assert row_stats.dtype == torch.float32 and row_stats.is_contiguous() and row_stats.numel() == v0.B * OH * OW * (pc.N // 96) * 2, label
    plan.keep.append((d, pc, srcs, dst, res, row_stats))
    sp0 = plan.sptr
    fn = lib.cvmi_conv2d

    def thunk(sp=None):
        sp = sp0 if sp is None else sp
        _lib.check(fn(C.byref(d), sp), label)

    M = v0.B * OH * OW
    es, oes = ESIZE[pc.dtype], ESIZE[dst.dtype]
    in_elems = v0.B * (v0.H * v0.W * v0.c + (v1.H * v1.W * v1.c if v1 is not None else 0))
    bytes_ = in_elems * es + M * pc.N * oes + (M * pc.N * oes if res is not None else 0)
    plan.add(label, kind, thunk, bytes_, 2 * M * pc.N * pc.K)
    return d


def c3k2_supported(c1, c, h, c2, fuse_cv1, dtype):
    return bool(_lib.load().cvmi_c3k2_supported(c1, c, h, c2, 1 if fuse_cv1 else 0, dtype))


def op_c3k2(plan, label, src, dst, pc_cv1, pc_m1, pc_m2, pc_cv2, c, h, fuse_cv1, shortcut=True):
    """Fused C3k2 block (one launch).  src: block input (fuse_cv1) or the [a|b] view; dst: block output view."""
    lib = _lib.load()
    assert (src.B, src.H, src.W) == (dst.B, dst.H, dst.W) and dst.c == pc_cv2.N
    assert src.c == (pc_cv1.Cin if fuse_cv1 else 2 * c), (label, src.c)
    d = C3k2Desc(x=src.ptr, y=dst.ptr, w0=pc_cv1.w.data_ptr() if fuse_cv1 else None, w1=pc_m1.w.data_ptr(), w2=pc_m2.w.data_ptr(),
                 w3=pc_cv2.w.data_ptr(), b0=pc_cv1.bias.data_ptr() if fuse_cv1 else None, b1=pc_m1.bias.data_ptr(), b2=pc_m2.bias.data_ptr(),
                 b3=pc_cv2.bias.data_ptr(), x_ld=src.ld, y_ld=dst.ld, kpad0=pc_cv1.Kpad if fuse_cv1 else 0, kpad1=pc_m1.Kpad, kpad2=pc_m2.Kpad,
                 kpad3=pc_cv2.Kpad, B=src.B, H=src.H, W=src.W, c1=pc_cv1.Cin if fuse_cv1 else 0, c=c, h=h, c2=pc_cv2.N,
                 fuse_cv1=1 if fuse_cv1 else 0, shortcut=1 if shortcut else 0, dtype=pc_cv2.dtype)
    plan.keep.append((d, src, dst, pc_cv1, pc_m1, pc_m2, pc_cv2))
    sp0, fn = plan.sptr, lib.cvmi_c3k2

    def thunk(sp=None):
        sp = sp0 if sp is None else sp
        _lib.check(fn(C.byref(d), sp), label)

    px = src.B * src.H * src.W
    flops = 2 * px * ((pc_cv1.Cin * 2 * c if fuse_cv1 else 0) + 9 * c * h * 2 + 3 * c * pc_cv2.N)
    plan.add(label, "conv", thunk, px * (src.c + dst.c) * ESIZE[pc_cv2.dtype], flops)


def stem2_supported(c0, c1, dtype):
    return bool(_lib.load().cvmi_stem2_supported(c0, c1, dtype))


def op_stem2(plan, label, pc0, pc1, src, dst):
    """Fused model.0 + model.1 (one launch).  src: space-to-depth(2) image view (16 channels), dst: model.1 output view."""
    lib = _lib.load()
    assert src.c == 16 and pc0.Cin == 16 and pc1.Cin == pc0.N and dst.c == pc1.N
    assert (dst.H, dst.W) == ((src.H - 1) // 2 + 1, (src.W - 1) // 2 + 1) and dst.B == src.B
    args = (src.ptr, src.ld, pc0.w.data_ptr(), pc0.bias.data_ptr(), pc0.Kpad, pc1.w.data_ptr(), pc1.bias.data_ptr(), pc1.Kpad,
            dst.ptr, dst.ld, src.B, src.H, src.W, pc0.N, pc1.N, pc1.dtype)
    plan.keep.append((pc0, pc1, src, dst))
    sp0, fn = plan.sptr, lib.cvmi_stem2

    def thunk(sp=None):
        sp = sp0 if sp is None else sp
        _lib.check(fn(*args, sp), label)

    px0, px1 = src.B * src.H * src.W, dst.B * dst.H * dst.W
    plan.add(label, "stem", thunk, (px0 * src.c + px1 * dst.c) * ESIZE[pc1.dtype], 2 * px0 * 27 * pc0.N + 2 * px1 * 9 * pc0.N * pc1.N)


def dwpw_supported(C_, n1, n2, dtype):
    return bool(_lib.load().cvmi_dwpw_supported(C_, n1, n2, dtype))


def op_dwpw(plan, label, pd, pc1, src, dst, pc2=None):
    """Fused DWConv3x3+SiLU -> Conv1x1+SiLU (-> Conv2d 1x1, no activation, when pc2 is given): one launch."""
    lib = _lib.load()
    assert src.c == pd.C == pc1.Cin and (src.B, src.H, src.W) == (dst.B, dst.H, dst.W)
    assert dst.c == (pc2.N if pc2 is not None else pc1.N) and (pc2 is None or pc2.Cin == pc1.N)
    d = _lib.DwPwDesc(x=src.ptr, y=dst.ptr, wd=pd.w.data_ptr(), bd=pd.bias.data_ptr(), w1=pc1.w.data_ptr(), b1=pc1.bias.data_ptr(),
                      w2=pc2.w.data_ptr() if pc2 is not None else None, b2=pc2.bias.data_ptr() if pc2 is not None else None,
                      x_ld=src.ld, y_ld=dst.ld, kpad1=pc1.Kpad, kpad2=pc2.Kpad if pc2 is not None else 0,
                      B=src.B, H=src.H, W=src.W, C=pd.C, N1=pc1.N, N2=pc2.N if pc2 is not None else 0, dtype=pc1.dtype)
    plan.keep.append((d, pd, pc1, pc2, src, dst))
    sp0, fn = plan.sptr, lib.cvmi_dwpw

    def thunk(sp=None):
        sp = sp0 if sp is None else sp
        _lib.check(fn(C.byref(d), sp), label)

    px = src.B * src.H * src.W
    flops = px * (18 * pd.C + 2 * pd.C * pc1.N + (2 * pc1.N * pc2.N if pc2 is not None else 0))
    plan.add(label, "head", thunk, px * (src.c + dst.c) * ESIZE[pc1.dtype], flops)


def op_dwconv(plan, label, pd, src, dst, act=_lib.ACT_NONE, res=None):
    lib = _lib.load()
    assert src.c == pd.C == dst.c and (src.B, src.H, src.W) == (dst.B, dst.H, dst.W)
    args = (src.ptr, src.ld, pd.w.data_ptr(), pd.bias.data_ptr(), res.ptr if res is not None else None,
            res.ld if res is not None else 0, dst.ptr, dst.ld, src.B, src.H, src.W, pd.C, act, pd.dtype)
    plan.keep.append((pd, src, dst, res))
    sp0 = plan.sptr
    fn = lib.cvmi_dwconv3x3

    def thunk(sp=None):
        sp = sp0 if sp is None else sp
        _lib.check(fn(*args, sp), label)

    n = src.B * src.H * src.W * pd.C
    plan.add(label, "dwconv", thunk, n * ESIZE[pd.dtype] * (3 if res is not None else 2), 18 * n)


def op_sppf_pool(plan, label, buf, c):
    lib = _lib.load()
    args = (buf.t.data_ptr(), buf.C, buf.B, buf.H, buf.W, c, buf.dtype)
    plan.keep.append(buf)
    sp0 = plan.sptr
    fn = lib.cvmi_sppf_pool

    def thunk(sp=None):
        sp = sp0 if sp is None else sp
        _lib.check(fn(*args, sp), label)

    n = buf.B * buf.H * buf.W * c
    plan.add(label, "pool", thunk, 4 * n * ESIZE[buf.dtype], 0)


def op_attention(plan, label, desc, keep, bytes_=0, flops=0):
    lib = _lib.load()
    plan.keep.append((desc, keep))
    sp0 = plan.sptr
    fn = lib.cvmi_attention

    def thunk(sp=None):
        sp = sp0 if sp is None else sp
        _lib.check(fn(C.byref(desc), sp), label)

    plan.add(label, "attention", thunk, bytes_, flops)


def make_attn_desc(**kw):
    d = AttnDesc()
    for k, v in kw.items():
        setattr(d, k, v)
    return d


# ---- SAM 2 path op wrappers ----------------------------------------------------------------------------
class Rows:
    """rows x C matrix view (ptr, ld, dtype) over any device tensor; used for token / pixel matrices."""

    def __init__(self, t, rows, C, ld=None, offset=0, dtype=None):
        self.t, self.rows, self.C = t, rows, C
        self.ld = ld if ld is not None else C
        self.offset = offset
        self.dtype = dtype if dtype is not None else DTYPE_OF[t.dtype]

    @property
    def ptr(self):
        return self.t.data_ptr() + self.offset * self.t.element_size()


def op_layernorm(plan, label, src, gamma, beta, dst, eps=1e-6, act=_lib.ACT_NONE, pad=None, dst2=None):
    """src / dst: Rows (or View, treated as B*H*W rows).  pad = (H, W, Hp, Wp): dst is the zero-padded grid.
    dst2 (f32 -> f32 only): an additional fp16 copy of the result."""
    lib = _lib.load()
    src, dst = _as_rows(src), _as_rows(dst)
    assert src.C == dst.C == gamma.numel()
    if dst2 is not None:
        dst2 = _as_rows(dst2)
        assert pad is None and act == _lib.ACT_NONE and src.dtype == dst.dtype == _lib.F32 and is16(dst2.dtype)
        assert dst2.rows == src.rows and dst2.C == src.C
        args2 = (src.ptr, src.ld, gamma.data_ptr(), beta.data_ptr(), dst.ptr, dst.ld, dst2.ptr, dst2.ld, dst2.dtype, src.rows, src.C, float(eps))
        plan.keep.append((src, dst, dst2, gamma, beta))
        sp1, fn2 = plan.sptr, lib.cvmi_layernorm_dual

        def thunk2(sp=None):
            sp = sp1 if sp is None else sp
            _lib.check(fn2(*args2, sp), label)

        plan.add(label, "layernorm", thunk2, src.rows * src.C * (4 + 4 + 2), 8 * src.rows * src.C)
        return
    if pad is None:
        assert src.rows == dst.rows
        pad = (0, 0, 0, 0)
    else:
        assert src.rows % (pad[0] * pad[1]) == 0 and dst.rows == src.rows // (pad[0] * pad[1]) * pad[2] * pad[3]
    args = (src.ptr, src.ld, src.dtype, gamma.data_ptr(), beta.data_ptr(), dst.ptr, dst.ld, dst.dtype, src.rows, src.C, float(eps), act) + tuple(pad)
    plan.keep.append((src, dst, gamma, beta))
    sp0, fn = plan.sptr, lib.cvmi_layernorm

    def thunk(sp=None):
        sp = sp0 if sp is None else sp
        _lib.check(fn(*args, sp), label)

    plan.add(label, "layernorm", thunk, src.rows * src.C * (ESIZE[src.dtype] + ESIZE[dst.dtype]), 8 * src.rows * src.C)


def _as_rows(v):
    if isinstance(v, Rows):
        return v
    if isinstance(v, View):
        return Rows(v.buf.t, v.B * v.H * v.W, v.c, ld=v.ld, offset=v.c0, dtype=v.dtype)
    raise TypeError(type(v))


def op_maxpool2(plan, label, src, dst):
    lib = _lib.load()
    assert src.c == dst.c and (dst.H, dst.W) == (src.H // 2, src.W // 2) and src.dtype == dst.dtype
    args = (src.ptr, src.ld, dst.ptr, dst.ld, src.B, src.H, src.W, src.c, src.dtype)
    plan.keep.append((src, dst))
    sp0, fn = plan.sptr, lib.cvmi_maxpool2x2

    def thunk(sp=None):
        sp = sp0 if sp is None else sp
        _lib.check(fn(*args, sp), label)

    n = src.B * src.H * src.W * src.c
    plan.add(label, "pool", thunk, n * ESIZE[src.dtype] * 5 // 4, 0)


def op_cast(plan, label, src, dst):
    lib = _lib.load()
    src, dst = _as_rows(src), _as_rows(dst)
    args = (src.ptr, src.ld, src.dtype, dst.ptr, dst.ld, dst.dtype, src.rows, src.C)
    plan.keep.append((src, dst))
    sp0, fn = plan.sptr, lib.cvmi_cast

    def thunk(sp=None):
        sp = sp0 if sp is None else sp
        _lib.check(fn(*args, sp), label)

    plan.add(label, "cast", thunk, src.rows * src.C * (ESIZE[src.dtype] + ESIZE[dst.dtype]), 0)


def op_call(plan, label, kind, fn, args, keep=(), bytes_=0, flops=0):
    """Generic: fn(*args, stream)."""
    plan.keep.append(keep)
    sp0 = plan.sptr

    def thunk(sp=None):
        sp = sp0 if sp is None else sp
        _lib.check(fn(*args, sp), label)

    plan.add(label, kind, thunk, bytes_, flops)


# ---- fused Hiera MLP (hiera_mlp.hip) ---------------------------------------------------------------------------------------
def hiera_mlp_supported(C_, dtype):
    return is16(dtype) and bool(_lib.load().cvmi_hiera_mlp_supported(C_))


class PackedHieraMlp:
    """fc1 / fc2 of one Hiera block in the MFMA-fragment order cvmi_hiera_mlp streams (include/cvmi355.h)."""

    def __init__(self, w1, b1, w2, b2, device="cuda", dtype=F16):
        # w1 [4C, C], b1 [4C], w2 [C, 4C], b2 [C]  (float32, LoRA already merged); dtype: F16 or BF16 operands
        td = TORCH_DTYPE[dtype]
        Hd, C_ = w1.shape
        assert Hd == 4 * C_ and tuple(w2.shape) == (C_, Hd) and C_ % 16 == 0
        ks1, nt, nch = C_ // 16 + 1, (C_ + 31) // 32, Hd // 32
        b_hi = b1.to(td).float()
        b_lo = (b1 - b_hi).to(td).float()
        w1x = torch.cat((w1, b_hi[:, None], b_lo[:, None], torch.zeros(Hd, 14)), 1)              # [4C, C + 16]
        f1 = w1x.view(nch, 32, ks1, 2, 8).permute(0, 2, 3, 1, 4)                                   # (j, s, h, r, e)
        w2p = torch.zeros(nt * 32, Hd)
        w2p[:C_] = w2
        f2 = w2p.view(nt, 32, nch, 2, 2, 2, 4).permute(2, 0, 3, 5, 1, 4, 6)                        # (j, t, s2, h, r, e_hi, e_lo)
        packed = torch.cat((f1.reshape(nch, -1), f2.reshape(nch, -1)), 1).contiguous()
        assert packed.numel() * 2 == _lib.load().cvmi_hiera_mlp_packed_bytes(C_)
        self.w = packed.to(td).to(device)
        self.bias = b2.float().contiguous().to(device)          # (.w / .bias: what distributed.packed_tensors broadcasts)
        self.C, self.dtype = C_, dtype
        self.param_bytes = 2 * Hd * C_ * 2


def op_hiera_mlp(plan, label, pm, x, gamma, beta, eps=1e-6, stats_out=None, stats_eps=1e-6):
    """x <- x + fc2(GELU(fc1(LayerNorm(x)))) in place; x: f32 View (full rows).  stats_out: f32 tensor [rows, 2] that receives each updated
    row's (mean, rstd) for the NEXT LayerNorm over x (eps = stats_eps)."""
    lib = _lib.load()
    assert x.dtype == F32 and x.c == pm.C and x.c0 == 0
    rows = x.B * x.H * x.W
    assert stats_out is None or (stats_out.dtype == torch.float32 and stats_out.numel() == 2 * rows and stats_out.is_contiguous()), label
    args = (x.ptr, x.ld, gamma.data_ptr(), beta.data_ptr(), float(eps), pm.w.data_ptr(), pm.bias.data_ptr(), rows, pm.C, pm.dtype,
            stats_out.data_ptr() if stats_out is not None else None, float(stats_eps))
    plan.keep.append((pm, x, gamma, beta, stats_out))
    sp0, fn = plan.sptr, lib.cvmi_hiera_mlp_stats

    def thunk(sp=None):
        sp = sp0 if sp is None else sp
        _lib.check(fn(*args, sp), label)

    plan.add(label, "mlp_fused", thunk, rows * pm.C * 8, 2 * rows * 2 * 4 * pm.C * pm.C)


# ---- token-stationary linear layer (tok_linear.hip) ------------------------------------------------------------------------
def tok_linear_supported(K, dtype, rows):
    return is16(dtype) and rows % 256 == 0 and bool(_lib.load().cvmi_tok_linear_supported(K))


class PackedTokLinear:
    """One linear layer in the MFMA-fragment order cvmi_tok_linear streams (include/cvmi355.h); the bias rides in the weights."""

    def __init__(self, w, b, device="cuda", dtype=F16):
        N, K = w.shape                                       # float32, LoRA already merged; dtype: F16 or BF16 operands
        td = TORCH_DTYPE[dtype]
        assert K % 16 == 0
        if _lib.load().cvmi_tok_linear_format(K) == 16:      # 16x16x32 MFMA fragments + an f32 bias piece per chunk (include/cvmi355.h)
            nch, ks = (N + 31) // 32, K // 32
            b = b if b is not None else torch.zeros(N)
            wx = torch.zeros(nch * 32, K)
            wx[:N] = w
            frag = wx.view(nch, 2, 16, ks, 4, 8).permute(0, 3, 1, 4, 2, 5).reshape(nch, ks * 2 * 512)      # (j, s, hh, g, r16, e)
            bias = torch.zeros(nch * 32)
            bias[:N] = b
            piece = torch.zeros(nch, 512, dtype=torch.int16)
            piece[:, :64] = bias.view(nch, 32).contiguous().view(torch.int16)
            packed = torch.cat((frag.to(td).contiguous().view(torch.int16), piece), 1).contiguous()
            assert packed.numel() * 2 == _lib.load().cvmi_tok_linear_packed_bytes(K, N)
            self.w = packed.to(device)
            self.bias = torch.zeros(4, device=device)
            self.N, self.K, self.dtype = N, K, dtype
            self.param_bytes = N * K * 2
            return
        ks1, nch = K // 16 + 1, ((N + 31) // 32 + 1) // 2 * 2       # chunk count padded to even (the kernel may take two per barrier)
        b = b if b is not None else torch.zeros(N)
        b_hi = b.to(td).float()
        b_lo = (b - b_hi).to(td).float()
        wx = torch.zeros(nch * 32, K + 16)
        wx[:N, :K], wx[:N, K], wx[:N, K + 1] = w, b_hi, b_lo
        packed = wx.view(nch, 32, ks1, 2, 8).permute(0, 2, 3, 1, 4).contiguous()                  # (j, s, h, r, e)
        assert packed.numel() * 2 == _lib.load().cvmi_tok_linear_packed_bytes(K, N)
        self.w = packed.to(td).to(device)
        self.bias = torch.zeros(4, device=device)             # (placeholder: distributed.packed_tensors expects .w / .bias)
        self.N, self.K, self.dtype = N, K, dtype
        self.param_bytes = N * K * 2


def tok_linear_stats_parts(rows, K, N):
    """Slices P of the LayerNorm statistics a residual-form op_tok_linear of this shape writes: 0 = [rows, 2] of (mean, rstd); P > 0 = [rows, P, 2]
    of per-slice (mean, sum of squared deviations), to be consumed with stats_parts = P (small launches share row blocks: cvmi355.h)."""
    return int(_lib.load().cvmi_tok_linear_stats_parts(rows, K, N))


def op_tok_linear(plan, label, pt, src, dst, ln=None, act=_lib.ACT_NONE, residual=False, kind="gemm", stats_in=None, stats_out=None, stats_eps=1e-6,
                  stats_parts=0):
    """src: f32 View with ln = (gamma, beta, eps) or ln = "cast" (f32 rows converted as they are, K <= 288), or an fp16 View.
    dst: fp16 View, or (residual=True) the f32 View updated in place.
    stats_out (residual=True): f32 tensor [rows, 2] that receives each updated row's (mean, rstd) for the NEXT LayerNorm (eps = stats_eps) --
    or [rows, P, 2] per-slice pairs when tok_linear_stats_parts(rows, K, N) = P > 0;
    stats_in (ln = (gamma, beta, eps)): such a tensor written by the launch that produced src -- the prologue then reads src once.
    stats_parts = P > 0: stats_in is instead f32 [rows, P, 2] of per-slice (mean, sum of squared deviations) (op_conv(row_stats=...))."""
    lib = _lib.load()
    src, dst = _as_rows(src), _as_rows(dst)
    assert src.C == pt.K and dst.C == pt.N and src.rows == dst.rows and src.rows % 256 == 0, label
    assert (src.dtype == F32) == (ln is not None) and (dst.dtype == F32) == bool(residual), label
    assert (ln is not None or src.dtype == pt.dtype) and (residual or dst.dtype == pt.dtype), label
    cast = isinstance(ln, str)
    assert not cast or ln == "cast", label
    assert stats_in is None or (ln is not None and not cast), label
    assert stats_out is None or residual, label
    out_parts = tok_linear_stats_parts(src.rows, pt.K, pt.N) if stats_out is not None else 0        # (the library decides: include/cvmi355.h)
    assert stats_out is None or (stats_out.dtype == torch.float32 and stats_out.numel() == 2 * src.rows * max(out_parts, 1) and stats_out.is_contiguous()), label
    assert stats_in is None or (stats_in.dtype == torch.float32 and stats_in.numel() == 2 * src.rows * max(stats_parts, 1) and stats_in.is_contiguous()), label
    gam, bet, eps = ln if (ln is not None and not cast) else (None, None, 0.0)
    args = (src.ptr, src.ld, 2 if cast else 1 if ln is not None else 0, gam.data_ptr() if gam is not None else None, bet.data_ptr() if bet is not None else None,
            float(eps), pt.w.data_ptr(), dst.ptr, dst.ld, 1 if residual else 0, src.rows, pt.K, pt.N, act, pt.dtype,
            stats_in.data_ptr() if stats_in is not None else None, int(stats_parts), stats_out.data_ptr() if stats_out is not None else None, float(stats_eps))
    plan.keep.append((pt, src, dst, gam, bet, stats_in, stats_out))
    sp0, fn = plan.sptr, lib.cvmi_tok_linear_stats

    def thunk(sp=None):
        sp = sp0 if sp is None else sp
        _lib.check(fn(*args, sp), label)

    bytes_ = src.rows * (pt.K * ESIZE[src.dtype] + pt.N * ESIZE[dst.dtype] * (2 if residual else 1))
    plan.add(label, kind, thunk, bytes_, 2 * src.rows * pt.N * pt.K)


def op_tok_linear_pool(plan, label, pt, src, dst, ln, kind="gemm", stats_in=None):
    """dst[b, y, x, :] = max over the 2 x 2 token block of (LayerNorm(src) W^T + b): src an f32 [B, H, W, K] View, dst f32 [B, H/2, W/2, N]
    (`do_pool(self.proj(norm1(x)))` of a Hiera q-pooling block in one launch)."""
    lib = _lib.load()
    B, H, W = src.B, src.H, src.W
    assert src.dtype == F32 and dst.dtype == F32 and src.c == pt.K and dst.c == pt.N, label
    assert (dst.B, dst.H, dst.W) == (B, H // 2, W // 2) and H % 2 == 0 and W % 2 == 0 and (B * H * W) % 256 == 0, label
    gam, bet, eps = ln
    assert stats_in is None or (stats_in.dtype == torch.float32 and stats_in.numel() == 2 * B * H * W and stats_in.is_contiguous()), label
    args = (src.ptr, src.ld, gam.data_ptr(), bet.data_ptr(), float(eps), pt.w.data_ptr(), dst.ptr, dst.ld, B, H, W, pt.K, pt.N, pt.dtype,
            stats_in.data_ptr() if stats_in is not None else None)
    plan.keep.append((pt, src, dst, gam, bet, stats_in))
    sp0, fn = plan.sptr, lib.cvmi_tok_linear_pool_stats

    def thunk(sp=None):
        sp = sp0 if sp is None else sp
        _lib.check(fn(*args, sp), label)

    rows = B * H * W
    plan.add(label, kind, thunk, rows * pt.K * 4 + rows // 4 * pt.N * 4, 2 * rows * pt.N * pt.K)
