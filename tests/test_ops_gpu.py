"""GPU parity: every HIP op through the C ABI vs a CPU fp32 statement of the same op.

Tolerances: f32 mode 1e-4 (exact-f32 MFMA, only the summation order differs from the CPU);
f16 mode 2e-2 (fp16 storage of inputs/outputs, fp32 accumulate; inputs are pre-rounded to fp16
so the oracle sees the same numbers).  NMS / integer outputs: bit-exact.
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from circuitvision_amd import _lib
from circuitvision_amd._lib import ACT_GELU, ACT_NONE, ACT_RELU, ACT_SILU, BF16, F16, F32
from circuitvision_amd.engine import (Buf, PackedConv, PackedDW, Plan, make_attn_desc, op_attention, op_conv, op_dwconv,
                                      op_sppf_pool)
from helpers import TOL, from_view, quant, run, stream, to_buf

pytestmark = pytest.mark.gpu
DTYPES = [F16, F32]
ACT_FN = {ACT_NONE: lambda x: x, ACT_SILU: F.silu, ACT_RELU: F.relu, ACT_GELU: F.gelu}


def _conv_case(dtype, B, Cin, H, W, Cout, k, s, act, res=False, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = quant(torch.randn(B, Cin, H, W, generator=g), dtype)
    w = quant(torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5, dtype)
    b = torch.randn(Cout, generator=g)
    ref = ACT_FN[act](F.conv2d(x, w, b, stride=s, padding=k // 2))
    OH, OW = ref.shape[2:]
    r = quant(torch.randn(B, Cout, OH, OW, generator=g), dtype) if res else None
    if res:
        ref = ref + r
    plan = Plan(stream())
    xb = to_buf(x, dtype)
    yb = Buf(B, OH, OW, (Cout + 7) // 8 * 8, dtype, zero=True)
    rb = to_buf(r, dtype, c_total=yb.C) if res else None
    pc = PackedConv(w, b, dtype)
    op_conv(plan, "t", pc, [(xb.view(), 0)], yb.view(0, Cout), stride=s, act=act,
            res=rb.view(0, Cout) if res else None, scalar_gather=(Cin % 8 != 0))
    run(plan)
    torch.testing.assert_close(from_view(yb.view(0, Cout)), ref, **TOL[dtype])
    if yb.C > Cout:                                   # padding lanes of the output buffer untouched
        assert float(yb.t[..., Cout:].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [
    # B, Cin, H, W, Cout, k, s, act, res
    (2, 16, 20, 24, 32, 1, 1, ACT_SILU, False),      # 1x1, N<=32 tile
    (2, 32, 17, 13, 64, 3, 1, ACT_SILU, True),       # 3x3 + residual, odd spatial, N<=64 tile
    (1, 64, 32, 32, 128, 3, 2, ACT_SILU, False),     # stride 2, N=128 tile
    (3, 8, 9, 7, 16, 3, 1, ACT_NONE, False),         # tiny channels (K=72 -> Kpad 96)
    (1, 128, 40, 40, 62, 1, 1, ACT_NONE, False),     # ragged Cout (class logits)
    (2, 3, 32, 48, 16, 3, 2, ACT_SILU, False),       # 3-channel stem, scalar gather
    (1, 256, 20, 20, 256, 3, 1, ACT_RELU, True),     # K=2304
    (1, 144, 16, 16, 432, 1, 1, ACT_GELU, False),    # Hiera-like linear: several N tiles
    (4, 32, 80, 80, 24, 3, 1, ACT_SILU, False),      # big-M path (BM=256 tile when M large)
])
def test_conv2d(dtype, cfg):
    _conv_case(dtype, *cfg)


@pytest.mark.parametrize("cfg", [
    # B, Cin, H, W, Cout, k, s, act, res
    (2, 64, 128, 128, 256, 3, 1, ACT_SILU, False),    # 128 tiles, ONE K-tile per tap (Cin = 64), stride 1: every border tap of every edge pixel
    (4, 128, 131, 129, 512, 3, 2, ACT_SILU, True),    # stride 2 on odd sizes (the last row / column sees 2 of 3 taps), two column tiles, residual
    (1, 256, 182, 180, 200, 3, 1, ACT_RELU, False),   # ragged Cout (200 of 256 columns: clamped weight rows), ragged M, 4 K-tiles per tap
    (4, 512, 129, 129, 512, 3, 2, ACT_SILU, False),   # YOLO11-l model.5 / .7 / .20 geometry: 512 -> 512 channels, stride 2, 72 K-tiles
])
def test_conv2d_3x3_im2col_on_the_256_tile_dma_pipeline(cfg):
    """Large 3 x 3 convolutions (>= 64 channels, a power-of-two multiple of 64; >= 128 tiles of 256 x 256) run on the counted-DMA GEMM with an im2col
    source: per-tap source offsets, DMA from a zero page for taps outside the image.  vs torch conv2d on fp16-rounded operands; the kernel that
    ran is checked, and a rerun is bit-identical."""
    lib = _lib.load()
    _conv_case(F16, *cfg)
    assert lib.cvmi_last_kernel().decode() == "gemm256_kernel<_Float16, true, true>", lib.cvmi_last_kernel()


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv2d_direct_to_lds_gemm(dtype):
    """Plain GEMMs with K a multiple of one 128-byte tile take the global_load_lds kernel (swizzled LDS image)."""
    _conv_case(dtype, 2, 256, 64, 64, 192, 1, 1, ACT_SILU)                 # 128x64 tiles, 4 (f16) / 8 (f32) K-tiles
    _conv_case(dtype, 1, 576, 70, 70, 432, 1, 1, ACT_GELU, res=True)        # 128x128 tiles, ragged M (4900) and N, residual
    _conv_case(dtype, 3, 128, 40, 40, 96, 1, 1, ACT_NONE)                   # exactly two tiles


def test_conv2d_intra_workgroup_split_k():
    """Small grids with a deep K (the 20x20 / 40x40 YOLO levels) split K over 2 or 4 wave groups of one workgroup."""
    _conv_case(F16, 2, 256, 20, 20, 64, 3, 1, ACT_SILU)                   # Detect cv2 level 2: 72 K-tiles, 4 groups, N = 64 tile
    _conv_case(F16, 2, 128, 40, 40, 256, 3, 2, ACT_SILU)                  # model.7: stride 2, 36 K-tiles, 64 x 128 tiles
    _conv_case(F16, 1, 136, 19, 21, 72, 3, 1, ACT_RELU, res=True)         # K = 1224 -> 39 K-tiles (ragged last group), ragged M / N, slow gather, residual
    _conv_case(F16, 3, 384, 20, 20, 128, 1, 1, ACT_SILU)                  # plain 1x1, 12 K-tiles: 2 groups
    _conv_case(F16, 1, 1096, 10, 10, 40, 1, 1, ACT_NONE)                  # plain, 35 K-tiles over 4 groups (9,9,9,8), ragged N
    # two sources (upsampled + skip) with a deep K: the fast gather restarts at each group's first tap
    g = torch.Generator().manual_seed(11)
    B, H, W = 2, 20, 20
    a = quant(torch.randn(B, 256, H // 2, W // 2, generator=g), F16)
    b = quant(torch.randn(B, 128, H, W, generator=g), F16)
    w = quant(torch.randn(128, 384, 3, 3, generator=g) / 58, F16)
    bias = torch.randn(128, generator=g)
    ref = F.silu(F.conv2d(torch.cat((F.interpolate(a, scale_factor=2, mode="nearest"), b), 1), w, bias, padding=1))
    ab, bb = to_buf(a, F16), to_buf(b, F16)
    yb = Buf(B, H, W, 128, F16, zero=True)
    outs = []
    for _ in range(3):
        plan = Plan(stream())
        op_conv(plan, "t", PackedConv(w, bias, F16), [(ab.view(), 1), (bb.view(), 0)], yb.view(), act=ACT_SILU)
        run(plan)
        outs.append(yb.t.clone())
    torch.testing.assert_close(from_view(yb.view()), ref, **TOL[F16])
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])   # fixed-order reduction: bit-identical reruns


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv2d_large_m_tiles(dtype):
    _conv_case(dtype, 2, 16, 264, 256, 16, 3, 2, ACT_SILU)      # M = 2*132*128 -> BM=256 config for N<=32? (M>=131072 not reached) still covers tails
    _conv_case(dtype, 8, 16, 128, 130, 32, 1, 1, ACT_SILU)      # M = 133120 >= 256*512 -> 256x32 tile
    _conv_case(dtype, 8, 64, 100, 90, 64, 1, 1, ACT_SILU)       # M = 72000 >= 65536 -> 128x64 tile
    _conv_case(dtype, 8, 64, 100, 90, 160, 1, 1, ACT_SILU)      # 128x128 tiles, ragged N tile


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv2d_two_sources_upsample(dtype):
    """Upsample(2x nearest) + Concat + 1x1 conv in one launch, reading channel slices of wider buffers."""
    g = torch.Generator().manual_seed(3)
    B, H, W = 2, 12, 10
    a = quant(torch.randn(B, 32, H // 2, W // 2, generator=g), dtype)     # half-res source
    b = quant(torch.randn(B, 16, H, W, generator=g), dtype)
    w = quant(torch.randn(40, 48, 1, 1, generator=g) / 7, dtype)
    bias = torch.randn(40, generator=g)
    ref = F.silu(F.conv2d(torch.cat((F.interpolate(a, scale_factor=2, mode="nearest"), b), 1), w, bias))
    plan = Plan(stream())
    ab = to_buf(a, dtype, c_total=48, c0=8)           # embedded at channel offset 8
    bb = to_buf(b, dtype, c_total=32, c0=16)
    yb = Buf(B, H, W, 64, dtype, zero=True)
    op_conv(plan, "t", PackedConv(w, bias, dtype), [(ab.view(8, 32), 1), (bb.view(16, 16), 0)], yb.view(16, 40), act=ACT_SILU)
    run(plan)
    torch.testing.assert_close(from_view(yb.view(16, 40)), ref, **TOL[dtype])
    assert float(yb.t[..., :16].abs().max()) == 0.0 and float(yb.t[..., 56:].abs().max()) == 0.0
    # channel counts that are K-tile multiples take the two-row-pointer gather (the YOLO neck's concat convs); both orders of
    # (upsampled, plain) sources, ragged M
    for up_first in (True, False):
        B, H, W = 3, 18, 22
        a = quant(torch.randn(B, 128, H // 2, W // 2, generator=g), dtype)
        b = quant(torch.randn(B, 64, H, W, generator=g), dtype)
        w = quant(torch.randn(96, 192, 1, 1, generator=g) / 14, dtype)
        bias = torch.randn(96, generator=g)
        ua = F.interpolate(a, scale_factor=2, mode="nearest")
        ref = F.silu(F.conv2d(torch.cat((ua, b) if up_first else (b, ua), 1), w, bias))
        plan = Plan(stream())
        ab, bb = to_buf(a, dtype), to_buf(b, dtype)
        yb = Buf(B, H, W, 96, dtype, zero=True)
        srcs = [(ab.view(), 1), (bb.view(), 0)] if up_first else [(bb.view(), 0), (ab.view(), 1)]
        op_conv(plan, "t", PackedConv(w, bias, dtype), srcs, yb.view(), act=ACT_SILU)
        run(plan)
        torch.testing.assert_close(from_view(yb.view()), ref, **TOL[dtype])


def test_conv2d_f16_in_f32_out():
    g = torch.Generator().manual_seed(5)
    x = quant(torch.randn(2, 64, 8, 8, generator=g), F16)
    w = quant(torch.randn(96, 64, 1, 1, generator=g) / 8, F16)
    b = torch.randn(96, generator=g)
    r = torch.randn(2, 96, 8, 8, generator=g)
    ref = F.conv2d(x, w, b) + r
    plan = Plan(stream())
    xb, rb = to_buf(x, F16), to_buf(r, F32)
    yb = Buf(2, 8, 8, 96, F32, zero=True)
    op_conv(plan, "t", PackedConv(w, b, F16), [(xb.view(), 0)], yb.view(), res=rb.view())
    run(plan)
    torch.testing.assert_close(from_view(yb.view()), ref, rtol=1e-3, atol=1e-3)


def test_conv2d_rejects_bad_arguments():
    lib = _lib.load()
    d = _lib.ConvDesc()
    assert lib.cvmi_conv2d(C.byref(d), None) != 0
    assert b"null" in lib.cvmi_last_error()
    x = Buf(1, 4, 4, 12, F16, zero=True)              # 12 channels: not a multiple of 8
    y = Buf(1, 4, 4, 16, F16, zero=True)
    pc = PackedConv(torch.zeros(16, 12, 1, 1), None, F16)
    plan = Plan(stream())
    op_conv(plan, "bad", pc, [(x.view(), 0)], y.view())
    with pytest.raises(_lib.CvmiError, match="multiples"):
        plan.run_eager()


@pytest.mark.parametrize("dtype", DTYPES)
def test_dwconv3x3(dtype):
    g = torch.Generator().manual_seed(1)
    x = quant(torch.randn(2, 64, 13, 11, generator=g), dtype)
    w = quant(torch.randn(64, 1, 3, 3, generator=g) / 3, dtype)
    b = torch.randn(64, generator=g)
    r = quant(torch.randn(2, 64, 13, 11, generator=g), dtype)
    ref = F.silu(F.conv2d(x, w, b, padding=1, groups=64)) + r
    plan = Plan(stream())
    xb, rb = to_buf(x, dtype, c_total=80, c0=16), to_buf(r, dtype)
    yb = Buf(2, 13, 11, 64, dtype, zero=True)
    op_dwconv(plan, "t", PackedDW(w, b, dtype), xb.view(16, 64), yb.view(), act=ACT_SILU, res=rb.view())
    run(plan)
    torch.testing.assert_close(from_view(yb.view()), ref, **TOL[dtype])


@pytest.mark.parametrize("hw", [(64, 96), (96, 72), (40, 56)])
def test_stem2_fused_first_two_layers(hw):
    """model.0 + model.1 (Conv 3x3 s2 + SiLU twice) in one launch on the space-to-depth image vs the fp32 statement;
    (96, 72) and (40, 56) give ragged output tiles (model.1 grid 24 x 18 / 10 x 14 against 8 x 16 tiles)."""
    from circuitvision_amd.engine import op_stem2, stem2_supported
    from circuitvision_amd.yolo11 import SyntheticParams, Yolo11Weights
    assert stem2_supported(16, 32, F16) and not stem2_supported(64, 128, F16)
    H, W = hw
    B = 2
    wt = Yolo11Weights("n", 62, SyntheticParams(seed=5, nc=62), F16)
    g = torch.Generator().manual_seed(9)
    x = quant(torch.rand(B, 3, H, W, generator=g), F16)
    w0, b0 = wt._fold("model.0", 3, 16, 3)
    w1, b1 = wt._fold("model.1", 16, 32, 3)
    t = quant(F.silu(F.conv2d(x, quant(w0, F16), b0, stride=2, padding=1)), F16)     # the fused kernel keeps model.0's output as fp16 too
    ref = F.silu(F.conv2d(t, quant(w1, F16), b1, stride=2, padding=1))
    s2d = x.reshape(B, 3, H // 2, 2, W // 2, 2).permute(0, 2, 4, 3, 5, 1).reshape(B, H // 2, W // 2, 12)
    xb = Buf(B, H // 2, W // 2, 16, F16, zero=True)
    xb.t[..., :12] = s2d.to(xb.t.dtype)
    yb = Buf(B, H // 4, W // 4, 40, F16, zero=True)
    plan = Plan(stream())
    op_stem2(plan, "t", wt.packed["model.0"], wt.packed["model.1"], xb.view(), yb.view(0, 32))
    run(plan)
    torch.testing.assert_close(from_view(yb.view(0, 32)), ref, **TOL[F16])
    assert float(yb.t[..., 32:].abs().max()) == 0.0


@pytest.mark.parametrize("cfg", [
    # B, C, H, W, N1, N2
    (2, 64, 24, 32, 80, 0),        # one 64-channel chunk, whole tiles
    (3, 128, 20, 20, 80, 0),       # two chunks, ragged tiles (20 = 2.5 x 8 rows, 1.25 x 16 columns)
    (1, 256, 13, 21, 80, 0),       # four chunks, odd sizes
    (2, 80, 20, 20, 80, 62),       # chained class conv: ragged channel tail (62 of 64), padding lanes untouched
    (1, 80, 40, 40, 80, 62),
    (2, 64, 9, 17, 72, 0),         # N1 < 80
    (2, 128, 20, 20, 64, 0),       # 64-column pointwise tile (YOLO11-n with nc <= 64: c3 = 64)
    (2, 64, 40, 40, 64, 62),       # chained class conv on the 64-column tile
])
def test_dwpw_fused_detect_class_branch(cfg):
    """DWConv3x3+SiLU -> Conv1x1+SiLU (-> Conv2d 1x1) in one launch vs the fp32 statement (ultralytics Detect.cv3)."""
    from circuitvision_amd.engine import dwpw_supported, op_dwpw
    B, C_, H, W, N1, N2 = cfg
    assert dwpw_supported(C_, N1, N2, F16)
    g = torch.Generator().manual_seed(7)
    x = quant(torch.randn(B, C_, H, W, generator=g), F16)
    wd = quant(torch.randn(C_, 1, 3, 3, generator=g) / 3, F16)
    bd = torch.randn(C_, generator=g) * 0.3
    w1 = quant(torch.randn(N1, C_, 1, 1, generator=g) / C_ ** 0.5, F16)
    b1 = torch.randn(N1, generator=g) * 0.3
    ref = F.silu(F.conv2d(F.silu(F.conv2d(x, wd, bd, padding=1, groups=C_)), w1, b1))
    pc2 = None
    if N2:
        w2 = quant(torch.randn(N2, N1, 1, 1, generator=g) / N1 ** 0.5, F16)
        b2 = torch.randn(N2, generator=g)
        ref = F.conv2d(ref, w2, b2)
        pc2 = PackedConv(w2, b2, F16)
    nout = N2 or N1
    xb = to_buf(x, F16)
    yb = Buf(B, H, W, (nout + 7) // 8 * 8 + 8, F16, zero=True)
    plan = Plan(stream())
    op_dwpw(plan, "t", PackedDW(wd, bd, F16), PackedConv(w1, b1, F16), xb.view(), yb.view(0, nout), pc2=pc2)
    run(plan)
    torch.testing.assert_close(from_view(yb.view(0, nout)), ref, **TOL[F16])
    assert float(yb.t[..., nout:].abs().max()) == 0.0
    assert not dwpw_supported(48, 80, 0, F16) and not dwpw_supported(64, 80, 0, F32)


@pytest.mark.parametrize("dtype", DTYPES)
def test_sppf_pool(dtype):
    g = torch.Generator().manual_seed(2)
    x = quant(torch.randn(2, 32, 20, 12, generator=g), dtype)
    y1 = F.max_pool2d(x, 5, 1, 2); y2 = F.max_pool2d(y1, 5, 1, 2); y3 = F.max_pool2d(y2, 5, 1, 2)
    ref = torch.cat((x, y1, y2, y3), 1)
    plan = Plan(stream())
    buf = to_buf(x, dtype, c_total=128)
    op_sppf_pool(plan, "t", buf, 32)
    run(plan)
    torch.testing.assert_close(from_view(buf.view()), ref, rtol=0, atol=0)


def _attn_ref(q, k, v, scale):
    return torch.softmax((q @ k.transpose(-1, -2)) * scale, -1) @ v


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [
    # B, heads, Nq, Nk, dqk, dv
    (2, 2, 400, 400, 32, 64),      # YOLO C2PSA
    (3, 4, 64, 64, 72, 72),        # Hiera window, head_dim 72
    (1, 8, 38, 300, 16, 16),       # decoder token->image (internal dim 128 / 8 heads)
    (1, 8, 300, 38, 32, 32),       # decoder image->token
    (1, 2, 130, 257, 72, 72),      # ragged q and key tiles
    (5, 3, 16, 16, 72, 72),        # tiny windows, private K/V tiles
    (2, 3, 600, 600, 72, 72),      # long sequence, head_dim 72, ragged last key tile: the general kernel
    (2, 3, 600, 640, 72, 72),      # long sequence in whole 64-key tiles: DMA-streamed tiles (attn_dma72_kernel), ragged q tiles, idle waves in the last workgroup
    (1, 2, 1024, 1024, 72, 72),    # same, whole q tiles
])
def test_attention(dtype, shape):
    B, Hh, Nq, Nk, dqk, dv = shape
    g = torch.Generator().manual_seed(11)
    q = quant(torch.randn(B, Hh, Nq, dqk, generator=g), dtype)
    k = quant(torch.randn(B, Hh, Nk, dqk, generator=g), dtype)
    v = quant(torch.randn(B, Hh, Nk, dv, generator=g), dtype)
    scale = dqk ** -0.5
    ref = _attn_ref(q, k, v, scale)
    # device layout: tokens x (heads * d), like a fused projection output
    from circuitvision_amd.engine import TORCH_DTYPE
    td = TORCH_DTYPE[dtype]
    qd = q.permute(0, 2, 1, 3).reshape(B, Nq, Hh * dqk).to(td).cuda().contiguous()
    kd = k.permute(0, 2, 1, 3).reshape(B, Nk, Hh * dqk).to(td).cuda().contiguous()
    vd = v.permute(0, 2, 1, 3).reshape(B, Nk, Hh * dv).to(td).cuda().contiguous()
    od = torch.zeros(B, Nq, Hh * dv, dtype=td, device="cuda")
    desc = make_attn_desc(q=qd.data_ptr(), k=kd.data_ptr(), v=vd.data_ptr(), o=od.data_ptr(),
                          q_sb=Nq * Hh * dqk, q_sh=dqk, q_st=Hh * dqk, k_sb=Nk * Hh * dqk, k_sh=dqk, k_st=Hh * dqk,
                          v_sb=Nk * Hh * dv, v_sh=dv, v_st=Hh * dv, o_sb=Nq * Hh * dv, o_sh=dv, o_st=Hh * dv,
                          B=B, heads=Hh, Nq=Nq, Nk=Nk, dqk=dqk, dv=dv, scale=scale, dtype=dtype,
                          win=0, grid_h=0, grid_w=0, q_pool=0)
    plan = Plan(stream())
    op_attention(plan, "t", desc, (qd, kd, vd, od))
    run(plan)
    out = od.float().cpu().view(B, Nq, Hh, dv).permute(0, 2, 1, 3)
    tol = dict(rtol=1e-2, atol=5e-3) if dtype == F16 else dict(rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(out, ref, **tol)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("q_pool", [0, 1])
@pytest.mark.parametrize("win", [8, 4, 16])
def test_attention_window_mode(dtype, q_pool, win):
    """Hiera windowed attention straight off the NHWC token grid (+ 2x2 q max-pool).  win = 4 takes the per-thread
    16-token kernel (fp16), win = 16 the 64-key-tile kernel, win = 8 the per-wave-tile kernel."""
    from circuitvision_amd.engine import TORCH_DTYPE
    td = TORCH_DTYPE[dtype]
    g = torch.Generator().manual_seed(13)
    imgs, gh, gw, heads, hd = 2, 16, 48 if win == 16 else 24, 3 if win == 4 else 2, 72
    C_ = heads * hd
    qkv = quant(torch.randn(imgs, gh, gw, 3 * C_, generator=g), dtype)
    # reference: partition windows
    def part(t):                                                 # [imgs,gh,gw,C] -> [imgs*nwin, win*win, heads, hd]
        t = t.view(imgs, gh // win, win, gw // win, win, heads, hd).permute(0, 1, 3, 2, 4, 5, 6)
        return t.reshape(-1, win * win, heads, hd)
    q, k, v = (part(qkv[..., i * C_:(i + 1) * C_]) for i in range(3))
    if q_pool:
        qq = q.view(-1, win, win, heads * hd).permute(0, 3, 1, 2)
        qq = F.max_pool2d(qq, 2, 2).permute(0, 2, 3, 1)
        q = qq.reshape(-1, (win // 2) ** 2, heads, hd)
    scale = hd ** -0.5
    ref = _attn_ref(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), scale).transpose(1, 2)   # [nw, Nq, heads, hd]
    ow = win // 2 if q_pool else win
    ogh, ogw = (gh // 2, gw // 2) if q_pool else (gh, gw)
    ref = ref.reshape(imgs, gh // win, gw // win, ow, ow, C_).permute(0, 1, 3, 2, 4, 5).reshape(imgs, ogh, ogw, C_)
    qkv_d = qkv.to(td).cuda().contiguous()
    od = torch.zeros(imgs, ogh, ogw, C_, dtype=td, device="cuda")
    es = qkv_d.element_size()
    nwin = imgs * (gh // win) * (gw // win)
    desc = make_attn_desc(q=qkv_d.data_ptr(), k=qkv_d.data_ptr() + C_ * es, v=qkv_d.data_ptr() + 2 * C_ * es, o=od.data_ptr(),
                          q_sb=0, q_sh=hd, q_st=3 * C_, k_sb=0, k_sh=hd, k_st=3 * C_, v_sb=0, v_sh=hd, v_st=3 * C_,
                          o_sb=0, o_sh=hd, o_st=C_, B=nwin, heads=heads, Nq=(ow * ow), Nk=win * win, dqk=hd, dv=hd,
                          scale=scale, dtype=dtype, win=win, grid_h=gh, grid_w=gw, q_pool=q_pool)
    plan = Plan(stream())
    op_attention(plan, "t", desc, (qkv_d, od))
    run(plan)
    tol = dict(rtol=1e-2, atol=5e-3) if dtype == F16 else dict(rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(od.float().cpu(), ref, **tol)


def _fp8_av_emulation(q, k, v, scale, tile=64):
    """The arithmetic cvmi_attn_desc.av_fp8 states, tile by tile as the kernels run it: per 64-key tile the running row maximum m, P = e4m3(exp(s - m)
    * 2^8) / 2^8, V = e4m3(clamp(v, +-448)), O <- O * exp(m_old - m) + P V in fp32, row sums of the UNquantised p.  [.., Nq, d] x [.., Nk, d]."""
    f8 = lambda t: t.to(torch.float8_e4m3fn).float()
    s = (q @ k.transpose(-1, -2)) * scale
    Nk = k.shape[-2]
    m = torch.full(s.shape[:-1] + (1,), float("-inf"))
    o = torch.zeros(q.shape[:-1] + (v.shape[-1],))
    l = torch.zeros_like(m)
    v8 = f8(v.clamp(-448, 448))
    for k0 in range(0, Nk, tile):
        st = s[..., k0:k0 + tile]
        m_new = torch.maximum(m, st.amax(-1, keepdim=True))
        alpha = torch.exp(m - m_new)
        pt = torch.exp(st - m_new)
        o = o * alpha + (f8(pt * 256.0) / 256.0) @ v8[..., k0:k0 + tile, :]
        l = l * alpha + pt.sum(-1, keepdim=True)
        m = m_new
    return o / l


def _check_fp8_av(name, got8, got16, ref, emu, dtype, absref):
    """(1) vs the emulation of the stated arithmetic: 1e-2 (+ the output's 16-bit rounding) -- a P element on a rounding boundary may flip by one
    e4m3 step (measured r03: 5.5e-3 windows, 1.2e-3 global); (2) vs the fp32 reference ON THE CLAMPED V: the price of 3 mantissa bits on P and
    V, bounded at 0.03 rms / 0.15 max for V of std 1.7 (measured: 2.0e-2 rms windows, 3e-3 rms / 3.7e-2 max global; the 16-bit product: 6e-3 max)."""
    assert torch.isfinite(got8).all()
    d_emu, d_ref = (got8 - emu).abs(), got8 - ref
    print(f"{name}: fp8 AV vs emulation max {float(d_emu.max()):.2e}; vs fp32 reference max {float(d_ref.abs().max()):.2e} rms {float(d_ref.pow(2).mean().sqrt()):.2e}; "
          f"16-bit AV vs fp32 reference max {float((got16 - ref).abs().max()):.2e}")
    rt = 2.0 ** -10 if dtype == F16 else 2.0 ** -7                # the stored output's own rounding (11 / 8 significant bits), with head-room
    assert bool((d_emu <= 1e-2 + rt * emu.abs()).all()), float((d_emu - rt * emu.abs()).max())
    # e4m3 rounds P and V to 3 mantissa bits each (relative step 2^-3, error <= 2^-4 each): |err| <= (2^-4 + 2^-4 + 2^-8) sum_k p |v| in the worst
    # case, far less on average (rms bound below)
    bound = 0.13 * absref + 1e-2 + rt * ref.abs()
    assert bool((d_ref.abs() <= bound).all()), float((d_ref.abs() - bound).max())
    assert float(d_ref.pow(2).mean().sqrt()) <= 0.03


@pytest.mark.parametrize("dtype", [F16, BF16])
def test_attention_window_fp8_av_product(dtype):
    """cvmi_attn_desc.av_fp8 (BASELINE configs[4] "fp8 MFMA attention") on Hiera's 16 x 16 windows (attn_res256_kernel<8, true>): softmax(QK^T) V
    with P as e4m3 of p * 2^8 and V as e4m3 on the block-scaled fp8 MFMA, vs the emulation of that arithmetic and vs fp32."""
    from circuitvision_amd.engine import TORCH_DTYPE
    td = TORCH_DTYPE[dtype]
    g = torch.Generator().manual_seed(17)
    imgs, gh, gw, heads, hd, win = 2, 32, 48, 4, 72, 16
    C_ = heads * hd
    qkv = torch.randn(imgs, gh, gw, 3 * C_, generator=g)
    qkv[..., 2 * C_:] *= 1.7                                      # V off the unit scale
    qkv[0, 3, 5, 2 * C_ + 7] = 600.0                              # one V element beyond e4m3's 448: clamps (the conversion itself would give NaN)
    qkv = quant(qkv, dtype)

    def part(t):
        t = t.view(imgs, gh // win, win, gw // win, win, heads, hd).permute(0, 1, 3, 2, 4, 5, 6)
        return t.reshape(-1, win * win, heads, hd).transpose(1, 2)            # [nwin, heads, 256, hd]
    q, k, v = (part(qkv[..., i * C_:(i + 1) * C_]) for i in range(3))
    scale = hd ** -0.5
    ref, emu = _attn_ref(q, k, v.clamp(-448, 448), scale), _fp8_av_emulation(q, k, v, scale)     # |v| <= 448 is part of the fp8 product's contract
    unwin = lambda t: t.transpose(1, 2).reshape(imgs, gh // win, gw // win, win, win, C_).permute(0, 1, 3, 2, 4, 5).reshape(imgs, gh, gw, C_)
    qkv_d = qkv.to(td).cuda().contiguous()
    outs = {}
    lib = _lib.load()
    for fp8 in (1, 0):
        od = torch.zeros(imgs, gh, gw, C_, dtype=td, device="cuda")
        es = qkv_d.element_size()
        desc = make_attn_desc(q=qkv_d.data_ptr(), k=qkv_d.data_ptr() + C_ * es, v=qkv_d.data_ptr() + 2 * C_ * es, o=od.data_ptr(),
                              q_sb=0, q_sh=hd, q_st=3 * C_, k_sb=0, k_sh=hd, k_st=3 * C_, v_sb=0, v_sh=hd, v_st=3 * C_,
                              o_sb=0, o_sh=hd, o_st=C_, B=imgs * (gh // win) * (gw // win), heads=heads, Nq=win * win, Nk=win * win, dqk=hd, dv=hd,
                              scale=scale, dtype=dtype, win=win, grid_h=gh, grid_w=gw, q_pool=0, av_fp8=fp8)
        plan = Plan(stream())
        op_attention(plan, "t", desc, (qkv_d, od))
        lib.cvmi_last_kernel()
        run(plan)
        assert lib.cvmi_last_kernel().decode() == ("attn_res256_kernel<8, true, false>" if fp8 else "attn_res256_kernel<8, false, false>")
        outs[fp8] = od.float().cpu()
    _check_fp8_av("16 x 16 windows", outs[1], outs[0], unwin(ref), unwin(emu), dtype, unwin(_attn_ref(q, k, v.clamp(-448, 448).abs(), scale)))


@pytest.mark.parametrize("dtype", [F16, BF16])
@pytest.mark.parametrize("N", [1024, 1536])
def test_attention_global_fp8_av_product(dtype, N):
    """The same on the long-sequence kernel (attn_dma72_kernel<8, true, false>: Hiera's global blocks; whole 64-key tiles -- the dispatcher sends
    ragged lengths and windows to the general kernel since r03)."""
    from circuitvision_amd.engine import TORCH_DTYPE
    td = TORCH_DTYPE[dtype]
    B, Hh, hd = 2, 3, 72
    g = torch.Generator().manual_seed(19)
    q = quant(torch.randn(B, Hh, N, hd, generator=g), dtype)
    k = quant(torch.randn(B, Hh, N, hd, generator=g), dtype)
    v = quant(torch.randn(B, Hh, N, hd, generator=g) * 1.7, dtype)
    scale = hd ** -0.5
    ref, emu = _attn_ref(q, k, v, scale), _fp8_av_emulation(q, k, v, scale)
    qd, kd, vd = (t.permute(0, 2, 1, 3).reshape(B, N, Hh * hd).to(td).cuda().contiguous() for t in (q, k, v))
    outs = {}
    lib = _lib.load()
    for fp8 in (1, 0):
        od = torch.zeros(B, N, Hh * hd, dtype=td, device="cuda")
        desc = make_attn_desc(q=qd.data_ptr(), k=kd.data_ptr(), v=vd.data_ptr(), o=od.data_ptr(),
                              q_sb=N * Hh * hd, q_sh=hd, q_st=Hh * hd, k_sb=N * Hh * hd, k_sh=hd, k_st=Hh * hd,
                              v_sb=N * Hh * hd, v_sh=hd, v_st=Hh * hd, o_sb=N * Hh * hd, o_sh=hd, o_st=Hh * hd,
                              B=B, heads=Hh, Nq=N, Nk=N, dqk=hd, dv=hd, scale=scale, dtype=dtype, win=0, grid_h=0, grid_w=0, q_pool=0, av_fp8=fp8)
        plan = Plan(stream())
        op_attention(plan, "t", desc, (qd, kd, vd, od))
        lib.cvmi_last_kernel()
        run(plan)
        assert lib.cvmi_last_kernel().decode() in (("attn_dma72_kernel<8, true, false>",) if fp8 else ("attn_dma72_kernel<4, false, false>", "attn_dma72_kernel<8, false, false>"))
        outs[fp8] = od.float().cpu().view(B, N, Hh, hd).permute(0, 2, 1, 3)
    _check_fp8_av(f"global N = {N}", outs[1], outs[0], ref, emu, dtype, _attn_ref(q, k, v.abs(), scale))


@pytest.mark.parametrize("dtype", [F16, BF16])
@pytest.mark.parametrize("case", ["plain", "far_maxima", "late_maximum"])
def test_attention_q_log2_prescaled_query(dtype, case):
    """cvmi_attn_desc.q_log2: q handed over already multiplied by scale * log2(e) gives the attention of the unscaled q.  On the long-sequence
    kernel (attn_dma72_kernel<8, false, true>) the running maximum then travels through the matrix pipe as a ROUNDED reference in a spare k slot
    of Q: "far_maxima" puts the row maxima at 90-110 in log2 units (a reference that is many ulps of the 16-bit type wide), "late_maximum"
    lets the maximum jump by >> the deferred-rescale threshold in the LAST key tile (the correction path after many plain tiles).  Reference:
    fp32 softmax of the same 16-bit operands; also the 256-key window kernel and a general-kernel shape, which only see scale = 1 / log2(e)."""
    from circuitvision_amd.engine import TORCH_DTYPE
    td = TORCH_DTYPE[dtype]
    hd = 72
    c = hd ** -0.5 * 1.4426950408889634
    lib = _lib.load()
    for (B, Hh, Nq, N, expect) in ((2, 3, 1024, 1024, "attn_dma72_kernel<8, false, true>"), (2, 2, 600, 640, "attn_dma72_kernel<8, false, true>"),
                                   (3, 2, 256, 256, "attn_res256_kernel<8, false, true>"), (1, 2, 600, 600, None)):
        g = torch.Generator().manual_seed(23 + N)
        q = torch.randn(B, Hh, Nq, hd, generator=g)
        k = torch.randn(B, Hh, N, hd, generator=g)
        v = quant(torch.randn(B, Hh, N, hd, generator=g), dtype)
        if case == "far_maxima":
            q = q * 6.0; k = k * 4.0
        if case == "late_maximum":
            k[:, :, -7] = q[:, :, 5] * 3.0                       # key N - 7 scores far above the rest for rows like row 5
        k = quant(k, dtype)
        qs = quant(q * c, dtype)                                 # what a projection with c folded into its rows writes
        ref = torch.softmax((qs @ k.transpose(-1, -2)) * 0.6931471805599453, -1) @ v
        qd = qs.permute(0, 2, 1, 3).reshape(B, Nq, Hh * hd).to(td).cuda().contiguous()
        kd, vd = (t.permute(0, 2, 1, 3).reshape(B, N, Hh * hd).to(td).cuda().contiguous() for t in (k, v))
        od = torch.zeros(B, Nq, Hh * hd, dtype=td, device="cuda")
        desc = make_attn_desc(q=qd.data_ptr(), k=kd.data_ptr(), v=vd.data_ptr(), o=od.data_ptr(),
                              q_sb=Nq * Hh * hd, q_sh=hd, q_st=Hh * hd, k_sb=N * Hh * hd, k_sh=hd, k_st=Hh * hd,
                              v_sb=N * Hh * hd, v_sh=hd, v_st=Hh * hd, o_sb=Nq * Hh * hd, o_sh=hd, o_st=Hh * hd,
                              B=B, heads=Hh, Nq=Nq, Nk=N, dqk=hd, dv=hd, scale=123.0, dtype=dtype, win=0, grid_h=0, grid_w=0, q_pool=0, av_fp8=0, q_log2=1)
        plan = Plan(stream())
        op_attention(plan, "t", desc, (qd, kd, vd, od))
        lib.cvmi_last_kernel()
        run(plan)
        kn = lib.cvmi_last_kernel().decode()
        assert expect is None or kn == expect, kn
        got = od.float().cpu().view(B, Nq, Hh, hd).permute(0, 2, 1, 3)
        assert torch.isfinite(got).all()
        tol = 4e-3 if dtype == F16 else 2.5e-2                    # P and O rounded to the 16-bit type, fp32 accumulation
        torch.testing.assert_close(got, ref, rtol=tol, atol=tol, msg=lambda m: f"{case} N={N} {kn}: {m}")
        first = od.clone()
        run(plan)
        assert torch.equal(od, first)


def test_graph_capture_replays_identically():
    g = torch.Generator().manual_seed(9)
    x = quant(torch.randn(2, 32, 16, 16, generator=g), F16)
    w = quant(torch.randn(32, 32, 3, 3, generator=g) / 17, F16)
    plan = Plan(stream())
    xb = to_buf(x, F16)
    y1, y2 = Buf(2, 16, 16, 32, F16, zero=True), Buf(2, 16, 16, 32, F16, zero=True)
    pc = PackedConv(w, None, F16)
    op_conv(plan, "a", pc, [(xb.view(), 0)], y1.view(), act=ACT_SILU)
    op_conv(plan, "b", pc, [(y1.view(), 0)], y2.view(), act=ACT_SILU, res=xb.view())
    run(plan)
    eager = y2.t.clone()
    y1.t.zero_(); y2.t.zero_()
    torch.cuda.synchronize()
    plan.run(); plan.run()
    plan.stream.synchronize()
    assert torch.equal(eager, y2.t)


@pytest.mark.parametrize("M,N,K,out_f32,act", [
    (8192, 2048, 576, False, _lib.ACT_NONE),      # odd number of K-tiles (9)
    (8192, 2048, 256, False, _lib.ACT_GELU),      # 4 K-tiles
    (8192, 2048, 128 * 5, True, _lib.ACT_NONE),   # f32 output + residual: two-pass epilogue
    (25600, 1160, 320, True, _lib.ACT_NONE),      # ragged N (last column tile 136 / 256 valid), f32 + residual, 5 K-tiles
    (25600, 1160, 256, False, _lib.ACT_RELU),     # ragged N, f16 output
    (8192, 2048, 288, False, _lib.ACT_NONE),      # K = 4.5 K-tiles: the last tile's out-of-row chunks re-read valid data (zero weights)
    (8192, 2048, 264, True, _lib.ACT_NONE),       # K % 64 = 8
    (65536, 288, 288, True, _lib.ACT_NONE),       # 128 x 64 DMA kernel with a ragged K (Hiera stage-2 proj)
    (65536, 144, 264, False, _lib.ACT_NONE),      # same kernel, f16 output, K % 64 = 8
    (65536, 576, 1160, True, _lib.ACT_NONE),      # 256 x 192 tiles with a ragged K
    (65536, 576, 1152, True, _lib.ACT_NONE),      # 256 x 192 tiles (N = 3 x 192, deep K): f32 + residual, 18 K-tiles
    (65536, 384, 1088, False, _lib.ACT_GELU),     # 256 x 192 tiles, f16 output, odd number of K-tiles (17)
])
def test_gemm_256_tile_counted_dma_pipeline(M, N, K, out_f32, act):
    """gemm256_kernel (8 waves, DMA in flight across barriers, staggered wave groups): values vs an fp32 matmul of the
    same f16 operands, and bit-identical results over repeated launches (a DMA / LDS race shows as run-to-run noise)."""
    g = torch.Generator().manual_seed(M + N + K)
    x = quant(torch.randn(M, K, generator=g), F16)
    w = quant(torch.randn(N, K, generator=g) / K ** 0.5, F16)
    b = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g) if out_f32 else None
    ref = x @ w.t() + b
    if act == _lib.ACT_GELU:
        ref = torch.nn.functional.gelu(ref)
    elif act == _lib.ACT_RELU:
        ref = torch.relu(ref)
    if res is not None:
        ref = ref + res
    xb = Buf(1, 1, M, K, F16); xb.t.copy_(x.view(1, 1, M, K).half())
    yb = Buf(1, 1, M, N, F32 if out_f32 else F16, zero=True)
    rb = None
    if res is not None:
        rb = Buf(1, 1, M, N, F32); rb.t.copy_(res.view(1, 1, M, N))
    pc = PackedConv(w.view(N, K, 1, 1), b, F16)
    plan = Plan(stream())
    op_conv(plan, "g256", pc, [(xb.view(), 0)], yb.view(), act=act, res=rb.view() if rb is not None else None)
    run(plan)
    first = yb.t.clone()
    got = first.float().view(M, N).cpu()
    tol = dict(rtol=2e-3, atol=2e-3) if out_f32 else dict(rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(got, ref, **tol)
    for _ in range(5):
        yb.t.zero_()
        run(plan)
        assert torch.equal(yb.t, first)


@pytest.mark.parametrize("dt", [F16, BF16])
@pytest.mark.parametrize("C_,rows", [(144, 1000), (288, 777), (144, 256 * 37 + 5)])
def test_hiera_mlp_fused_vs_torch(C_, rows, dt):
    """x + fc2(GELU(fc1(LayerNorm(x)))) in one launch (hiera_mlp.hip) vs fp32 torch on the same fp16-rounded weights:
    the pre-activations are computed from fp16 operands and the hidden activation is rounded to fp16, as in the unfused chain.
    Ragged row counts exercise the clamped last tile; rows beyond `rows` must stay untouched."""
    import torch.nn.functional as TF
    from circuitvision_amd.engine import PackedHieraMlp, op_hiera_mlp
    g = torch.Generator().manual_seed(C_ + rows)
    x = torch.randn(rows + 3, C_, generator=g) * 1.5 + 0.3
    gam, bet = torch.rand(C_, generator=g) + 0.5, torch.randn(C_, generator=g) * 0.2
    w1 = quant(torch.randn(4 * C_, C_, generator=g) / C_ ** 0.5, dt)
    b1 = torch.randn(4 * C_, generator=g) * 0.3
    w2 = quant(torch.randn(C_, 4 * C_, generator=g) / (4 * C_) ** 0.5, dt)
    b2 = torch.randn(C_, generator=g) * 0.3
    xn = quant(TF.layer_norm(x[:rows], (C_,), gam, bet, 1e-6), dt)
    hid = quant(TF.gelu(xn @ w1.t() + b1), dt)
    ref = x[:rows] + hid @ w2.t() + b2
    pm = PackedHieraMlp(w1, b1, w2, b2, dtype=dt)
    xb = Buf(1, 1, rows + 3, C_, F32)
    xb.t.copy_(x.view(1, 1, rows + 3, C_))
    view = xb.images(0, 1).view()
    view.buf.W = rows                                             # the op covers the first `rows` rows only
    plan = Plan(stream())
    stats = torch.full((rows, 2), -7.0, device="cuda")                 # LayerNorm statistics of the UPDATED rows, for the next block's norm1
    op_hiera_mlp(plan, "mlp", pm, view, gam.cuda(), bet.cuda(), 1e-6, stats_out=stats, stats_eps=1e-6)
    run(plan)
    got = xb.t.view(rows + 3, C_).cpu()
    assert torch.equal(got[rows:], x[rows:])
    exp_stats = torch.stack((got[:rows].mean(1), 1.0 / torch.sqrt(got[:rows].var(1, unbiased=False) + 1e-6)), 1)
    torch.testing.assert_close(stats.cpu(), exp_stats, rtol=2e-5, atol=2e-5)
    err = (got[:rows] - ref).abs().max().item()
    tol = 3e-3 if dt == F16 else 2.4e-2                       # bf16: 8 mantissa bits against 11
    torch.testing.assert_close(got[:rows], ref, rtol=tol, atol=tol), err
    # bit-identical reruns (a second pass on the SAME input)
    xb.t.copy_(x.view(1, 1, rows + 3, C_))
    run(plan)
    assert torch.equal(xb.t.view(rows + 3, C_).cpu(), got)


@pytest.mark.parametrize("dt", [F16, BF16])
@pytest.mark.parametrize("rows", [5, 777, 128 * 40 + 5])
def test_hiera_mlp_pipelined_loop_equals_chunk_order_loop(rows, dt, monkeypatch):
    """C = 288: the software-pipelined chunk loop (asm blocks; GELU of chunk j between the MFMAs of fc1(j + 1) and fc2(j)) computes every value with
    the same operations in the same order as the chunk-order loop it replaces: outputs and forwarded LayerNorm statistics must be bit-identical.
    (A scheduling hazard inside the asm blocks -- nothing hipcc checks -- shows up here as a difference.)"""
    from circuitvision_amd.engine import PackedHieraMlp, op_hiera_mlp
    C_ = 288
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, C_, generator=g) * 1.5 + 0.3
    gam, bet = torch.rand(C_, generator=g) + 0.5, torch.randn(C_, generator=g) * 0.2
    pm = PackedHieraMlp(torch.randn(4 * C_, C_, generator=g) / C_ ** 0.5, torch.randn(4 * C_, generator=g) * 0.3,
                        torch.randn(C_, 4 * C_, generator=g) / (4 * C_) ** 0.5, torch.randn(C_, generator=g) * 0.3, dtype=dt)
    outs = []
    for pipe in ("0", "2"):
        monkeypatch.setenv("CVMI_MLP_PIPE", pipe)
        xb = Buf(1, 1, rows, C_, F32)
        xb.t.copy_(x.view(1, 1, rows, C_))
        stats = torch.zeros((rows, 2), device="cuda")
        plan = Plan(stream())
        op_hiera_mlp(plan, "mlp", pm, xb.images(0, 1).view(), gam.cuda(), bet.cuda(), 1e-6, stats_out=stats, stats_eps=1e-6)
        torch.cuda.synchronize()
        kern = plan.timed_eager(with_kernels=True)[0][5]
        outs.append((xb.t.cpu().clone(), stats.cpu().clone(), kern))
    assert outs[0][2] != outs[1][2], outs[0][2]                     # (two different kernels ran)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("K,N,ln,res,act", [(144, 432, True, False, ACT_NONE), (288, 864, True, False, ACT_NONE), (576, 1728, True, False, ACT_NONE),
                                            (576, 2304, True, False, ACT_GELU), (576, 576, False, True, ACT_NONE), (144, 144, False, True, ACT_NONE),
                                            (288, 104, False, False, ACT_GELU), (576, 40, True, True, ACT_NONE), (144, 432, True, False, ACT_GELU), (288, 288, True, True, ACT_NONE),
                                            (144, 32, "cast", False, ACT_NONE), (288, 64, "cast", False, ACT_NONE)])
@pytest.mark.parametrize("dt", [F16, BF16])
def test_tok_linear_vs_torch(K, N, ln, res, act, dt):
    """Token-stationary linear layer (tok_linear.hip): optional fused LayerNorm of the f32 stream on the way in, 16-bit output with
    optional GELU or in-place f32 residual update, N not a multiple of 32 (masked last chunk), vs fp32 torch on weights rounded to the
    operand type (fp16, or bf16: the -DCVMI_OPERAND_BF16 build of the same kernel)."""
    import torch.nn.functional as TF
    from circuitvision_amd.engine import TORCH_DTYPE, PackedTokLinear, Rows, op_tok_linear
    td = TORCH_DTYPE[dt]
    tol = 1.0 if dt == F16 else 8.0                            # bf16 carries 8 mantissa bits against fp16's 11
    rows = 512
    g = torch.Generator().manual_seed(K + N)
    w = quant(torch.randn(N, K, generator=g) / K ** 0.5, dt)
    b = torch.randn(N, generator=g) * 0.3
    gam, bet = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.2
    if ln == "cast":                                        # f32 rows converted as they are (the neck's lateral convs on the f32 stream)
        x = torch.randn(rows, K, generator=g) * 1.5 + 0.7
        xin = quant(x, dt)
        src_t = x.cuda()
    elif ln:
        x = torch.randn(rows, K, generator=g) * 1.5 + 0.7
        x[:, 3] += 40.0                                     # an outlier channel: the variance pass must cope
        xin = quant(TF.layer_norm(x, (K,), gam, bet, 1e-6), dt)
        src_t = x.cuda()
    else:
        x = quant(torch.randn(rows, K, generator=g), dt)
        xin = x
        src_t = x.to(td).cuda()
    y = xin @ w.t() + b
    if act == ACT_GELU:
        y = TF.gelu(y)
    Np = (N + 7) // 8 * 8
    if res:
        r0 = torch.randn(rows, Np, generator=g)
        ref = r0.clone(); ref[:, :N] += y
        dst_t = r0.cuda()
    else:
        ref = None
        dst_t = torch.full((rows, Np), 7.0, dtype=td, device="cuda")
    pt = PackedTokLinear(w, b, dtype=dt)
    plan = Plan(stream())
    op_tok_linear(plan, "tl", pt, Rows(src_t, rows, K), Rows(dst_t, rows, N, ld=Np), ln="cast" if ln == "cast" else (gam.cuda(), bet.cuda(), 1e-6) if ln else None,
                  act=act, residual=res)
    run(plan)
    got = dst_t.float().cpu()
    if res:
        torch.testing.assert_close(got, ref, rtol=3e-3 * tol, atol=3e-3 * tol)
    else:
        torch.testing.assert_close(got[:, :N], y, rtol=4e-3 * tol, atol=4e-3 * tol)
        assert bool((got[:, N:] == 7.0).all())               # columns beyond N untouched


@pytest.mark.parametrize("K,N,B,H,W", [(144, 288, 2, 16, 24), (288, 576, 1, 16, 16), (576, 1152, 3, 8, 32), (144, 40, 1, 32, 8)])
@pytest.mark.parametrize("dt", [F16, BF16])
def test_tok_linear_pool_vs_torch(K, N, B, H, W, dt):
    """Shortcut path of a Hiera q-pooling block in one launch (tok_linear.hip, POOL form): LayerNorm of the f32 token grid, linear layer,
    2 x 2 max-pool over the token grid (the four tokens of a block sit in one lane quad), f32 output -- vs fp32 torch on weights rounded
    to the operand type.  Also: N not a multiple of 32 and columns beyond N untouched."""
    import torch.nn.functional as TF
    from circuitvision_amd.engine import Buf, PackedTokLinear, op_tok_linear_pool
    tol = 1.0 if dt == F16 else 8.0
    g = torch.Generator().manual_seed(K * 7 + N)
    w = quant(torch.randn(N, K, generator=g) / K ** 0.5, dt)
    b = torch.randn(N, generator=g) * 0.3
    gam, bet = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.2
    x = torch.randn(B, H, W, K, generator=g) * 1.5 + 0.7
    x[..., 5] -= 30.0
    xin = quant(TF.layer_norm(x, (K,), gam, bet, 1e-6), dt)
    y = xin @ w.t() + b                                                    # [B, H, W, N]
    ref = TF.max_pool2d(y.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)   # [B, H/2, W/2, N]
    Np = (N + 7) // 8 * 8
    src = Buf(B, H, W, K, F32); src.t.copy_(x)
    dst = Buf(B, H // 2, W // 2, Np, F32); dst.t.fill_(7.0)
    pt = PackedTokLinear(w, b, dtype=dt)
    plan = Plan(stream())
    op_tok_linear_pool(plan, "tlp", pt, src.view(), dst.view(0, N), (gam.cuda(), bet.cuda(), 1e-6))
    run(plan)
    got = dst.t.float().cpu()
    torch.testing.assert_close(got[..., :N], ref, rtol=4e-3 * tol, atol=4e-3 * tol)
    assert bool((got[..., N:] == 7.0).all())
    # the same launch with the rows' LayerNorm statistics supplied (as the previous block's fused MLP writes them): one pass over the rows
    xr = x.reshape(-1, K)
    stats = torch.stack((xr.mean(1), 1.0 / torch.sqrt(xr.var(1, unbiased=False) + 1e-6)), 1).contiguous().cuda()
    dst2 = Buf(B, H // 2, W // 2, Np, F32); dst2.t.fill_(7.0)
    plan = Plan(stream())
    op_tok_linear_pool(plan, "tlp_stats", pt, src.view(), dst2.view(0, N), (gam.cuda(), bet.cuda(), 1e-6), stats_in=stats)
    run(plan)
    torch.testing.assert_close(dst2.t.float().cpu()[..., :N], got[..., :N], rtol=2e-3 * tol, atol=2e-3 * tol)


@pytest.mark.parametrize("K,N2", [(576, 2304), (288, 1152), (144, 576)])
@pytest.mark.parametrize("dt", [F16, BF16])
def test_tok_linear_forwarded_layernorm_statistics(K, N2, dt):
    """x = x + proj(a) writes each updated row's LayerNorm statistics (mean, rstd); the next launch, GELU(fc1(LayerNorm(x))), takes them
    instead of computing them in a first pass over x (cvmi_tok_linear_stats).  Statistics vs fp32 torch on the updated rows (an outlier
    channel and a common offset included); the fc1 output with forwarded statistics vs the one with the two-pass prologue and vs torch."""
    import torch.nn.functional as TF
    from circuitvision_amd.engine import TORCH_DTYPE, PackedTokLinear, Rows, op_tok_linear
    td = TORCH_DTYPE[dt]
    tol = 1.0 if dt == F16 else 8.0
    rows = 512
    g = torch.Generator().manual_seed(K)
    wp = quant(torch.randn(K, K, generator=g) / K ** 0.5, dt); bp = torch.randn(K, generator=g) * 0.2
    w1 = quant(torch.randn(N2, K, generator=g) / K ** 0.5, dt); b1 = torch.randn(N2, generator=g) * 0.2
    gam, bet = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.2
    a = quant(torch.randn(rows, K, generator=g), dt)
    x0 = torch.randn(rows, K, generator=g) * 1.5 + 3.0
    x0[:, 7] += 50.0
    x1 = x0 + a @ wp.t() + bp                                              # the updated stream
    mean, var = x1.mean(1), x1.var(1, unbiased=False)
    ref_stats = torch.stack((mean, 1.0 / torch.sqrt(var + 1e-6)), 1)
    y_ref = TF.gelu(quant(TF.layer_norm(x1, (K,), gam, bet, 1e-6), dt) @ w1.t() + b1)
    xd = x0.cuda()
    stats = torch.zeros(rows, 2, device="cuda")
    out_a = torch.empty(rows, N2, dtype=td, device="cuda"); out_b = torch.empty_like(out_a)
    pp_, p1 = PackedTokLinear(wp, bp, dtype=dt), PackedTokLinear(w1, b1, dtype=dt)
    plan = Plan(stream())
    op_tok_linear(plan, "proj", pp_, Rows(a.to(td).cuda(), rows, K), Rows(xd, rows, K), residual=True, stats_out=stats, stats_eps=1e-6)
    op_tok_linear(plan, "fc1_fwd", p1, Rows(xd, rows, K), Rows(out_a, rows, N2), ln=(gam.cuda(), bet.cuda(), 1e-6), act=ACT_GELU, stats_in=stats)
    op_tok_linear(plan, "fc1_two_pass", p1, Rows(xd, rows, K), Rows(out_b, rows, N2), ln=(gam.cuda(), bet.cuda(), 1e-6), act=ACT_GELU)
    run(plan)
    torch.testing.assert_close(xd.cpu(), x1, rtol=3e-3 * tol, atol=3e-3 * tol)
    got = stats.cpu()
    x1d = xd.cpu()                                                          # statistics are those of the rows as WRITTEN
    exp = torch.stack((x1d.mean(1), 1.0 / torch.sqrt(x1d.var(1, unbiased=False) + 1e-6)), 1)
    torch.testing.assert_close(got, exp, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(got, ref_stats, rtol=5e-3 * tol, atol=5e-3 * tol)
    ya, yb = out_a.float().cpu(), out_b.float().cpu()
    torch.testing.assert_close(ya, yb, rtol=2e-3 * tol, atol=2e-3 * tol)     # same rows, same statistics up to summation order
    torch.testing.assert_close(ya, y_ref, rtol=6e-3 * tol, atol=6e-3 * tol)


@pytest.mark.parametrize("dt", [F16, BF16])
def test_gemm_row_statistics_feed_the_next_layernorm(dt):
    """Hiera stage-3 fc2 shape (65536 x 2304 -> 576, f32 residual stream updated in place): the 256 x 192 GEMM also writes, per row and per
    96-column slice, (mean, sum of squared deviations) of the values it stores (cvmi_conv_desc.row_stats); the next block's qkv
    (cvmi_tok_linear_stats with ln_stats_in_parts = 6) combines them instead of reading the rows twice.  Statistics vs float64 moments of the rows
    AS WRITTEN; the qkv output with forwarded statistics vs the two-pass prologue on the same rows; replays bit-identical.  Rows 0..255 sit 500
    (>= 300 standard deviations) away from zero: raw (sum, sum of squares) partials would lose the variance there."""
    from circuitvision_amd.engine import TORCH_DTYPE, PackedTokLinear, Rows, op_tok_linear, row_stats_supported
    td = TORCH_DTYPE[dt]
    tol = 1.0 if dt == F16 else 8.0
    M, K, N = 65536, 2304, 576
    assert row_stats_supported(M, N, K)
    g = torch.Generator(device="cuda").manual_seed(3)
    hid = (torch.randn(M, K, generator=g, device="cuda") * 0.5).to(td)
    w = quant(torch.randn(N, K) / K ** 0.5, dt); b = torch.randn(N) * 0.2
    x0 = torch.randn(M, N, generator=g, device="cuda") * 1.5 + 0.4
    x0[:, 11] += 30.0
    x0[:256] += 500.0                                            # a common offset of >= 300 sigma
    xb = Buf(1, 1, M, N, F32); xb.t.copy_(x0.view(1, 1, M, N))
    hb = Buf(1, 1, M, K, dt); hb.t.copy_(hid.view(1, 1, M, K))
    parts = torch.zeros(M, N // 96, 2, device="cuda")
    pc = PackedConv(w.view(N, K, 1, 1), b, dt)
    wq = quant(torch.randn(64, N) / N ** 0.5, dt)
    pq = PackedTokLinear(wq, torch.zeros(64), dtype=dt)
    gam, bet = (torch.rand(N) + 0.5).cuda(), (torch.randn(N) * 0.2).cuda()
    oa = torch.empty(M, 64, dtype=td, device="cuda"); ob = torch.empty_like(oa)
    plan = Plan(stream())
    op_conv(plan, "fc2", pc, [(hb.view(), 0)], xb.view(), res=xb.view(), row_stats=parts)
    op_tok_linear(plan, "qkv_fwd", pq, Rows(xb.t.view(M, N), M, N), Rows(oa, M, 64), ln=(gam, bet, 1e-6), stats_in=parts, stats_parts=N // 96)
    op_tok_linear(plan, "qkv_two_pass", pq, Rows(xb.t.view(M, N), M, N), Rows(ob, M, 64), ln=(gam, bet, 1e-6))
    run(plan)
    x1 = xb.t.view(M, N).double()
    sl = x1.view(M, N // 96, 96)
    mu = sl.mean(2, keepdim=True)
    exp = torch.stack((mu[..., 0], ((sl - mu) ** 2).sum(2)), 2).float()
    torch.testing.assert_close(parts, exp, rtol=2e-5, atol=2e-4)
    # the offset rows: the forwarded statistics give the same normalised rows as the two-pass prologue (variance not lost to cancellation)
    torch.testing.assert_close(oa[:256].float(), ob[:256].float(), rtol=2e-3 * tol, atol=2e-3 * tol)
    wd, bd = w.double().cuda().t().contiguous(), b.double().cuda()
    for r0 in range(0, M, 8192):                                 # EVERY row block and column tile (the row-block-persistent kernel walks 3 tiles per workgroup)
        ref_rows = x0[r0:r0 + 8192].double() + hid[r0:r0 + 8192].double() @ wd + bd
        torch.testing.assert_close(x1[r0:r0 + 8192], ref_rows, rtol=2e-3 * tol, atol=2e-3 * tol)
    torch.testing.assert_close(oa.float(), ob.float(), rtol=2e-3 * tol, atol=2e-3 * tol)
    first = (parts.clone(), oa.clone())
    xb.t.copy_(x0.view(1, 1, M, N)); parts.zero_()
    run(plan)
    assert torch.equal(parts, first[0]) and torch.equal(oa, first[1])


@pytest.mark.parametrize("dt", [F16, BF16])
def test_tok_linear16_row_blocks_shared_between_workgroups(dt):
    """A K = 576 launch with fewer 256-row blocks than the chip has CUs (SAM 2.1-L at 8 images per rank = 128 blocks; here 8) gives each row
    block to several workgroups, each walking its own range of output-channel chunks (tok_linear16.hip, gridDim.y).  Every form of the kernel at
    such a shape vs fp32 torch: LayerNorm + GELU (fc1, 8 sharers), LayerNorm plain (qkv, 6), residual with statistics out (proj, 6: the
    statistics then arrive as per-slice (mean, sum of squared deviations) pairs, `cvmi_tok_linear_stats_parts`), their consumption by the next
    LayerNorm launch, a ragged last chunk (N = 40, 2 sharers), the pooled projection; reruns are bit-identical."""
    import torch.nn.functional as TF
    from circuitvision_amd.engine import TORCH_DTYPE, Buf, PackedTokLinear, Rows, op_tok_linear, op_tok_linear_pool, tok_linear_stats_parts
    td = TORCH_DTYPE[dt]
    tol = 1.0 if dt == F16 else 8.0
    K, rows = 576, 2048
    P = tok_linear_stats_parts(rows, K, K)
    assert P == 6 and tok_linear_stats_parts(65536, K, K) == 0 and tok_linear_stats_parts(32768, K, K) == 2 and tok_linear_stats_parts(rows, 288, 288) == 0
    g = torch.Generator().manual_seed(11)
    mk = lambda n: (quant(torch.randn(n, K, generator=g) / K ** 0.5, dt), torch.randn(n, generator=g) * 0.3)
    (wp, bp), (w1, b1), (wq, bq), (ws, bs) = mk(K), mk(2304), mk(1728), mk(40)
    gam, bet = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.2
    a = quant(torch.randn(rows, K, generator=g), dt)
    x0 = torch.randn(rows, K, generator=g) * 1.5 + 3.0
    x0[:, 7] += 50.0
    xd = x0.cuda()
    stats = torch.zeros(rows, P, 2, device="cuda")
    hid = torch.empty(rows, 2304, dtype=td, device="cuda"); hid2 = torch.empty_like(hid)
    qkv = torch.empty(rows, 1728, dtype=td, device="cuda")
    small = torch.randn(rows, 40, generator=g).cuda(); small0 = small.clone()
    pk = lambda w, b: PackedTokLinear(w, b, dtype=dt)
    plan = Plan(stream())
    op_tok_linear(plan, "proj", pk(wp, bp), Rows(a.to(td).cuda(), rows, K), Rows(xd, rows, K), residual=True, stats_out=stats, stats_eps=1e-6)
    op_tok_linear(plan, "fc1_fwd", pk(w1, b1), Rows(xd, rows, K), Rows(hid, rows, 2304), ln=(gam.cuda(), bet.cuda(), 1e-6), act=ACT_GELU, stats_in=stats, stats_parts=P)
    op_tok_linear(plan, "fc1_own", pk(w1, b1), Rows(xd, rows, K), Rows(hid2, rows, 2304), ln=(gam.cuda(), bet.cuda(), 1e-6), act=ACT_GELU)
    op_tok_linear(plan, "qkv", pk(wq, bq), Rows(xd, rows, K), Rows(qkv, rows, 1728), ln=(gam.cuda(), bet.cuda(), 1e-6))
    op_tok_linear(plan, "ragged", pk(ws, bs), Rows(a.to(td).cuda(), rows, K), Rows(small, rows, 40), residual=True)
    run(plan)
    x1 = x0 + a @ wp.t() + bp
    torch.testing.assert_close(xd.cpu(), x1, rtol=3e-3 * tol, atol=3e-3 * tol)
    x1d = xd.cpu().double()                                                 # statistics are those of the rows as WRITTEN, per 96-column slice
    sl = x1d.view(rows, P, K // P)
    exp = torch.stack((sl.mean(2), sl.var(2, unbiased=False) * (K // P)), 2).float()
    torch.testing.assert_close(stats.cpu(), exp, rtol=1e-4, atol=1e-3)
    xn = quant(TF.layer_norm(x1d.float(), (K,), gam, bet, 1e-6), dt)
    ya, yb = hid.float().cpu(), hid2.float().cpu()
    torch.testing.assert_close(ya, yb, rtol=2e-3 * tol, atol=2e-3 * tol)
    torch.testing.assert_close(ya, TF.gelu(xn @ w1.t() + b1), rtol=6e-3 * tol, atol=6e-3 * tol)
    torch.testing.assert_close(qkv.float().cpu(), xn @ wq.t() + bq, rtol=6e-3 * tol, atol=6e-3 * tol)
    torch.testing.assert_close(small.cpu(), small0.cpu() + a @ ws.t() + bs, rtol=3e-3 * tol, atol=3e-3 * tol)
    first = (xd.clone(), hid.clone(), qkv.clone(), stats.clone())
    xd.copy_(x0); small.copy_(small0)
    run(plan)
    assert all(torch.equal(u, v) for u, v in zip(first, (xd, hid, qkv, stats)))
    # pooled projection (the 3 -> 4 stage transition's shortcut), 2048 rows of the token grid
    B, H, W, N = 2, 32, 32, 1152
    wpo, bpo = quant(torch.randn(N, K, generator=g) / K ** 0.5, dt), torch.randn(N, generator=g) * 0.3
    xg = torch.randn(B, H, W, K, generator=g) * 1.5 + 0.7
    ref = TF.max_pool2d((quant(TF.layer_norm(xg, (K,), gam, bet, 1e-6), dt) @ wpo.t() + bpo).permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
    src = Buf(B, H, W, K, F32); src.t.copy_(xg)
    dst = Buf(B, H // 2, W // 2, N, F32)
    plan = Plan(stream())
    op_tok_linear_pool(plan, "pool", pk(wpo, bpo), src.view(), dst.view(), (gam.cuda(), bet.cuda(), 1e-6))
    run(plan)
    torch.testing.assert_close(dst.t.float().cpu(), ref, rtol=4e-3 * tol, atol=4e-3 * tol)


def test_tok_linear_plain_f32_input_is_rejected_at_k576():
    """in_f32_layernorm = 2 (f32 rows converted as they are) is built for K = 144 / 288 only; K = 576 must fail cleanly at the C ABI instead of
    running the LayerNorm form with NULL gamma / beta."""
    lib = _lib.load()
    x = torch.zeros(256, 576, device="cuda")
    y = torch.zeros(256, 64, dtype=torch.float16, device="cuda")
    w = torch.zeros(lib.cvmi_tok_linear_packed_bytes(576, 64) // 2, dtype=torch.float16, device="cuda")
    rc = lib.cvmi_tok_linear(x.data_ptr(), 576, 2, None, None, 0.0, w.data_ptr(), y.data_ptr(), 64, 0, 256, 576, 64, ACT_NONE, F16, None)
    assert rc != 0 and b"K = 144 and 288" in lib.cvmi_last_error()
