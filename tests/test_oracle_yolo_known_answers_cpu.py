"""Known-answer checks of the YOLO11 half of the oracle that need no ultralytics ("what green covers": the restated network
had only its TOTAL parameter counts behind it).  Per layer, for scale n AND l, at 640 x 640 (parameters at nc = 80, FLOPs at the build's nc = 62):

  * output shape                    vs SURVEY.md Table Y (the survey's own derivation from the upstream yaml, validated there against the
  * conv multiply-accumulates         published 2.6 M / 6.5 GFLOP (n) and 25.3 M / 86.9 GFLOP (l)),
  * parameter count                 vs closed forms written here from the block DEFINITIONS (below Table Y) -- a second derivation
                                      of every block's structure (C3k2's c3k rule, C3k's inner width, C2PSA's key_dim, Detect's c2 / c3),
  * DFL                             vs an expectation worked by hand,
  * LetterBox geometry              vs hand-worked cases on both sides of the round(d -+ 0.1) split.
A transcription slip in one block changes that layer's row; the totals alone could hide two compensating slips."""
import math

import pytest
import torch
import torch.nn as nn

from oracle.yolo11 import DFL, YOLO11

# layer -> (out channels n, out channels l, output side at 640, GFLOP n, GFLOP l)   (None: no arithmetic)
TABLE_Y = {0: (16, 64, 320, 0.09, 0.35), 1: (32, 128, 160, 0.24, 3.78), 2: (64, 256, 160, 0.33, 8.81), 3: (64, 256, 80, 0.47, 7.55),
           4: (128, 512, 80, 0.33, 8.81), 5: (128, 512, 40, 0.47, 7.55), 6: (128, 512, 40, 0.28, 7.13), 7: (256, 512, 20, 0.24, 1.89),
           8: (256, 512, 20, 0.28, 1.78), 9: (256, 512, 20, 0.13, 0.52), 10: (256, 512, 20, 0.26, 1.40), 11: (256, 512, 40, None, None),
           13: (128, 512, 40, 0.35, 7.97), 14: (128, 512, 80, None, None), 16: (64, 256, 80, 0.41, 9.65), 17: (64, 256, 40, 0.12, 1.89),
           19: (128, 512, 40, 0.28, 7.55), 20: (128, 512, 20, 0.12, 1.89), 22: (256, 512, 20, 0.30, 1.99), 23: (None, None, None, 1.76, 6.57)}


# ---- closed-form parameter counts from the block definitions (Conv = conv without bias + BatchNorm weight and bias)
def p_conv(c1, c2, k, g=1):
    return c1 // g * c2 * k * k + 2 * c2


def p_bottleneck(c1, c2, e=0.5, k=3):
    c_ = int(c2 * e)
    return p_conv(c1, c_, k) + p_conv(c_, c2, k)


def p_c3k(c1, c2, n=2):
    c_ = int(c2 * 0.5)
    return 2 * p_conv(c1, c_, 1) + p_conv(2 * c_, c2, 1) + n * p_bottleneck(c_, c_, e=1.0)


def p_c3k2(c1, c2, n, c3k, e=0.5):
    c = int(c2 * e)
    inner = p_c3k(c, c) if c3k else p_bottleneck(c, c)
    return p_conv(c1, 2 * c, 1) + p_conv((2 + n) * c, c2, 1) + n * inner


def p_sppf(c1, c2):
    return p_conv(c1, c1 // 2, 1) + p_conv(4 * (c1 // 2), c2, 1)


def p_c2psa(c1, n):
    c = c1 // 2
    heads = c // 64
    attn = p_conv(c, c + 2 * heads * 32, 1) + p_conv(c, c, 1) + p_conv(c, c, 3, g=c)       # qkv (key_dim 32 = head_dim / 2), proj, depthwise pe
    ffn = p_conv(c, 2 * c, 1) + p_conv(2 * c, c, 1)
    return 2 * p_conv(c1, 2 * c, 1) // 2 + p_conv(2 * c, c1, 1) + n * (attn + ffn)           # cv1 (c1 -> 2c), cv2 (2c -> c1), n PSABlocks


def p_detect(nc, ch):
    c2, c3 = max(16, ch[0] // 4, 64), max(ch[0], min(nc, 100))
    tot = 16                                                                                 # DFL's frozen arange conv
    for x in ch:
        tot += p_conv(x, c2, 3) + p_conv(c2, c2, 3) + (c2 * 64 + 64)
        tot += p_conv(x, x, 3, g=x) + p_conv(x, c3, 1) + p_conv(c3, c3, 3, g=c3) + p_conv(c3, c3, 1) + (c3 * nc + nc)
    return tot


def expected_params(scale, nc=80):
    d, w, mc = {"n": (0.5, 0.25, 1024), "l": (1.0, 1.0, 512)}[scale]
    ch = lambda c: int(math.ceil(min(c, mc) * w / 8) * 8)
    rep = lambda n: max(round(n * d), 1) if n > 1 else n
    big = scale == "l"
    c64, c128, c256, c512, c1024 = ch(64), ch(128), ch(256), ch(512), ch(1024)
    return {0: p_conv(3, c64, 3), 1: p_conv(c64, c128, 3), 2: p_c3k2(c128, c256, rep(2), big, 0.25), 3: p_conv(c256, c256, 3),
            4: p_c3k2(c256, c512, rep(2), big, 0.25), 5: p_conv(c512, c512, 3), 6: p_c3k2(c512, c512, rep(2), True), 7: p_conv(c512, c1024, 3),
            8: p_c3k2(c1024, c1024, rep(2), True), 9: p_sppf(c1024, c1024), 10: p_c2psa(c1024, rep(2)),
            13: p_c3k2(c1024 + c512, c512, rep(2), big), 16: p_c3k2(c512 + c512, c256, rep(2), big), 17: p_conv(c256, c256, 3),
            19: p_c3k2(c256 + c512, c512, rep(2), big), 20: p_conv(c512, c512, 3), 22: p_c3k2(c512 + c1024, c1024, rep(2), True),
            23: p_detect(nc, (c256, c512, c1024))}


@pytest.mark.parametrize("scale,total", [("n", 2_624_080), ("l", 25_372_160)])
def test_per_layer_parameter_counts(scale, total):
    m = YOLO11(scale, 80)
    exp = expected_params(scale)
    got = {i: sum(p.numel() for p in layer.parameters()) for i, layer in enumerate(m.model)}
    for i, e in exp.items():
        assert got[i] == e, (scale, i, got[i], e)
    assert all(got[i] == 0 for i in (11, 12, 14, 15, 18, 21))
    assert sum(exp.values()) == total == sum(got.values())


@pytest.mark.parametrize("scale", ["n", "l"])
def test_per_layer_output_shapes_and_flops_match_table_y(scale):
    """One forward pass at 640 x 640 with hooks: each layer's output shape, and 2 x its multiply-accumulates (convolutions + the C2PSA
    attention products), in GFLOP rounded as Table Y prints them."""
    m = YOLO11(scale, 62).eval()                                  # (Table Y's Detect row and SURVEY 8(d)'s totals are quoted at the build's nc = 62)
    macs, shapes = {}, {}
    cur = [None]

    def conv_hook(mod, inp, out):
        macs[cur[0]] = macs.get(cur[0], 0) + out.numel() // out.shape[0] * (mod.in_channels // mod.groups) * mod.kernel_size[0] * mod.kernel_size[1]

    hooks = []
    for i, layer in enumerate(m.model):
        layer.register_forward_pre_hook(lambda mod, inp, i=i: cur.__setitem__(0, i))
        layer.register_forward_hook(lambda mod, inp, out, i=i: shapes.__setitem__(i, out))
        for sub in layer.modules():
            if isinstance(sub, nn.Conv2d):
                hooks.append(sub.register_forward_hook(conv_hook))
    with torch.no_grad():
        y = m(torch.zeros(1, 3, 640, 640))
    assert y.shape == (1, 66, 8400)
    col = 0 if scale == "n" else 1
    for i, row in TABLE_Y.items():
        if row[2] is not None:
            assert tuple(shapes[i].shape) == (1, row[col], row[2], row[2]), (scale, i, tuple(shapes[i].shape))
        if row[3 + col] is not None:
            fl = 2 * macs[i]
            if i == 10:                                           # (added to the layer's count so that the total below has it too)                                           # + q k^T and v attn^T of the PSA blocks: 2 x heads x 400 x 400 x (32 + 64) each
                c = row[col] // 2
                extra = (1 if scale == "n" else 2) * 2 * (c // 64) * 400 * 400 * (32 + 64)
                fl += extra
                macs[i] += extra // 2
            assert abs(fl / 1e9 - row[3 + col]) <= 0.0055, (scale, i, fl / 1e9, row[3 + col])
    tot = 2 * sum(macs.values()) / 1e9
    assert abs(tot - (6.44 if scale == "n" else 87.1)) < 0.1, tot              # 6.44 (n) / 87.1 (l) GFLOP at nc = 62 (rows are rounded to 0.01)


def test_dfl_expectation_by_hand():
    """DFL = softmax over the 16 bins of each side, then the expectation against 0..15, bins of a side contiguous (channel = side * 16 + bin)."""
    d = DFL(16)
    x = torch.full((1, 64, 2), -30.0)
    x[0, 0 * 16 + 3, 0] = 30.0                                   # left: all mass on bin 3
    x[0, 1 * 16 + 0, 0] = 0.0; x[0, 1 * 16 + 15, 0] = 0.0; x[0, 16 + 1:16 + 15, 0] = -1e4     # top: bins 0 and 15 half each -> 7.5
    x[0, 2 * 16:3 * 16, 0] = 0.0                                 # right: uniform -> 7.5
    x[0, 3 * 16 + 4, 0] = math.log(3.0); x[0, 3 * 16 + 8, 0] = 0.0; x[0, 3 * 16:3 * 16 + 4, 0] = -1e4     # bottom: 3/4 on bin 4, 1/4 on bin 8 -> 5
    x[0, 3 * 16 + 5:3 * 16 + 8, 0] = -1e4; x[0, 3 * 16 + 9:, 0] = -1e4
    x[0, :, 1] = torch.arange(64, dtype=torch.float32) % 16 * 0.0   # anchor 1: all zero logits -> 7.5 on every side
    out = d(x)
    torch.testing.assert_close(out[0, :, 0], torch.tensor([3.0, 7.5, 7.5, 5.0]), rtol=0, atol=1e-5)
    torch.testing.assert_close(out[0, :, 1], torch.full((4,), 7.5), rtol=0, atol=1e-6)


def test_letterbox_geometry_by_hand():
    """ultralytics LetterBox(640, auto=True, stride=32): r = min(640 / h, 640 / w); new = round(w r), round(h r); pad = (640 - new) mod 32, halved;
    top = round(p - 0.1), bottom = round(p + 0.1): an odd total pad puts the extra pixel at the bottom / right."""
    from circuitvision_amd.detector import letterbox_geometry
    from oracle.preprocess import letterbox_geometry as oracle_geometry
    cases = {                       # (h, w) -> (new_w, new_h, top, bottom, left, right), worked by hand
        (720, 1280): (640, 360, 12, 12, 0, 0),       # r = 0.5; pad_h = 280 mod 32 = 24 -> 12 / 12
        (531, 640): (640, 531, 6, 7, 0, 0),          # r = 1; pad_h = 109 mod 32 = 13 -> 6.5 -> round(6.4) = 6, round(6.6) = 7
        (640, 531): (531, 640, 0, 0, 6, 7),          # the same split on the other axis
        (530, 640): (640, 530, 7, 7, 0, 0),          # pad 110 mod 32 = 14 -> 7 / 7
        (640, 640): (640, 640, 0, 0, 0, 0),
        (1000, 1333): (640, 480, 0, 0, 0, 0),        # r = 0.48012; 480.12 -> 480; pad 160 mod 32 = 0
        (300, 420): (640, 457, 11, 12, 0, 0),        # r = 1.5238; 457.14 -> 457; pad 183 mod 32 = 23 -> 11.5 -> 11 / 12
    }
    for (h, w), want in cases.items():
        assert tuple(letterbox_geometry(h, w, 640)) == want, ((h, w), letterbox_geometry(h, w, 640))
        assert tuple(oracle_geometry(h, w, 640)) == want, ((h, w), oracle_geometry(h, w, 640))
