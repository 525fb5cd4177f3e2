"""CPU suite: the oracle against the reference's golden vectors / known-answer anchors, host logic,
and the C-ABI library's symbol table (no GPU compute)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from oracle import nms as onms
from oracle import preprocess as opre
from oracle.yolo11 import YOLO11, count_params

GOLD = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- stage-2 NMS pinned by the reference's own functions (src/utils.py:297-361) ------------------
def test_stage2_nms_matches_reference_golden():
    with open(os.path.join(GOLD, "nms_stage2.json")) as f:
        gold = json.load(f)
    assert len(gold["cases"]) >= 18
    for case in gold["cases"]:
        kept = onms.nms_by_confidence([dict(b) for b in case["boxes"]], case["iou_threshold"])
        assert [b["persistent_uid"] for b in kept] == case["kept_by_confidence"], case["name"]
        kept = onms.nms_by_area([dict(b) for b in case["boxes"]], case["iou_threshold"])
        assert [b["persistent_uid"] for b in kept] == case["kept_by_area"], case["name"]
    for pr in gold["iou_pairs"]:
        assert onms.calculate_iou(pr["a"], pr["b"]) == pr["iou"]


def test_product_stage2_nms_matches_reference_golden():
    from circuitvision_amd.detector import calculate_iou, non_max_suppression_by_confidence
    with open(os.path.join(GOLD, "nms_stage2.json")) as f:
        gold = json.load(f)
    for case in gold["cases"]:
        kept = non_max_suppression_by_confidence([dict(b) for b in case["boxes"]], case["iou_threshold"])
        assert [b["persistent_uid"] for b in kept] == case["kept_by_confidence"], case["name"]
    for pr in gold["iou_pairs"]:
        assert calculate_iou(pr["a"], pr["b"]) == pr["iou"]


# ---- YOLO11 known-answer anchors -------------------------------------------------
@pytest.mark.parametrize("scale,params", [("n", 2_624_080), ("l", 25_372_160)])
def test_yolo11_param_counts(scale, params):
    assert count_params(YOLO11(scale, 80)) == params


def test_yolo11_output_shape_and_decode_ranges():
    m = YOLO11("n", 62).eval()
    with torch.no_grad():
        y = m(torch.rand(1, 3, 96, 128))
    assert y.shape == (1, 66, 12 * 16 + 6 * 8 + 3 * 4)
    assert float(y[:, 4:].min()) >= 0 and float(y[:, 4:].max()) <= 1


def test_product_synthetic_checkpoint_loads_strictly_into_oracle():
    """Two independent statements of the architecture (product graph walker, oracle nn.Module) must
    agree on every parameter name and shape."""
    from circuitvision_amd._lib import F32
    from circuitvision_amd.yolo11 import SyntheticParams, Yolo11Weights
    for scale in ("n", "l"):
        p = SyntheticParams(seed=1, nc=62)
        wt = Yolo11Weights(scale, 62, p, F32, device="cpu")
        m = YOLO11(scale, 62)
        m.load_state_dict(p.state_dict(), strict=True)
        stem = wt.packed["model.0"]                      # stored space-to-depth: 16x2x2 taps of which 27 are real
        n_packed = sum(pc.param_bytes for pc in wt.packed.values()) // 4 - stem.N * (64 - 27)
        n_conv = sum(v.numel() for k, v in p.state_dict().items() if k.endswith("conv.weight") or re.search(r"cv[23]\.\d\.2\.weight$", k))
        assert n_packed == n_conv - 16        # everything but the DFL's constant arange(16) conv


# ---- ultralytics-semantics NMS restatement: self-consistency properties ---------------------------
def test_yolo_nms_properties():
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from synth import nms_stress_pred
    pred = nms_stress_pred(2, 20, (320, 320), seed=7)
    out, idx = onms.yolo_nms(pred, return_indices=True)
    for b, det in enumerate(out):
        assert det.shape[0] <= 300 and det.shape[0] > 0
        assert bool((det[:-1, 4] >= det[1:, 4]).all())              # sorted by confidence
        assert bool((det[:, 4] > 0.25).all())
        # idempotence: NMS of the survivors keeps all of them
        keep = onms.torchvision_nms(det[:, :4] + det[:, 5:6] * 7680, det[:, 4], 0.7)
        assert keep.numel() == det.shape[0]
        # the reported anchor reproduces the box
        a = idx[b]
        cx, cy, w, h = pred[b, 0, a], pred[b, 1, a], pred[b, 2, a], pred[b, 3, a]
        assert torch.equal(det[:, 0], cx - w / 2) and torch.equal(det[:, 3], cy + h / 2)


def test_scale_boxes_roundtrip():
    boxes = torch.tensor([[10.0, 20.0, 200.0, 300.0]])
    out = onms.scale_boxes((384, 640), boxes, (720, 1280))
    assert torch.allclose(out, torch.tensor([[20.0, 16.0, 400.0, 576.0]]))


# ---- letterbox restatement -------------------------------------------------------------------------
def test_letterbox_geometry_and_identity():
    assert opre.letterbox_geometry(720, 1280) == (640, 360, 12, 12, 0, 0)
    assert opre.letterbox_geometry(640, 640) == (640, 640, 0, 0, 0, 0)
    from circuitvision_amd.detector import letterbox_geometry
    for hw in ((720, 1280), (493, 712), (33, 900), (1000, 1000)):
        assert letterbox_geometry(*hw) == opre.letterbox_geometry(*hw)
    img = np.arange(64 * 64 * 3, dtype=np.uint8).reshape(64, 64, 3)
    assert np.array_equal(opre.resize_linear_u8(img, 64, 64), img)
    flat = np.full((50, 70, 3), 77, np.uint8)
    assert np.all(opre.resize_linear_u8(flat, 31, 23) == 77)       # constants survive the fixed-point path
    x = opre.yolo_preprocess(np.full((720, 1280, 3), 255, np.uint8))
    assert x.shape == (1, 3, 384, 640)
    assert x[0, 0, 0, 0] == np.float32(114 / 255) and x[0, 0, 100, 100] == 1.0


# ---- C ABI: the library loads and exports every symbol the header declares ------------------------
def test_library_exports_every_declared_symbol():
    from circuitvision_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "cvmi355.h")).read()
    declared = set(re.findall(r"\b(cvmi_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"cvmi_conv_desc", "cvmi_attn_desc"}
    assert declared, "no declarations parsed"
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libcvmi355.so not built (python -c 'import __graft_entry__ as g; g.build()')")
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in cvmi355.h but not exported"
    assert set(_lib.SIGNATURES) == declared
    assert lib.cvmi_version() >= 121                      # 121: cvmi_attn_desc.q_log2


def test_descriptor_mirrors_match_the_library_and_a_short_struct_is_rejected():
    """The ctypes mirrors of the four descriptor structs must have the size libcvmi355.so was compiled with (cvmi_desc_size), field
    order as in include/cvmi355.h; a mirror that is short by its trailing field (INTEGRATION.md once documented cvmi_conv_desc without
    `row_stats`) is refused at load instead of being read past its end."""
    from circuitvision_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libcvmi355.so not built")
    lib = _lib.load()                                     # runs check_abi
    for kind, cls in _lib.DESC_MIRRORS.items():
        assert lib.cvmi_desc_size(kind) == ctypes.sizeof(cls) > 0
    assert lib.cvmi_desc_size(99) == 0
    hdr = open(os.path.join(ROOT, "include", "cvmi355.h")).read()
    body = hdr[hdr.index("typedef struct cvmi_conv_desc {"):hdr.index("} cvmi_conv_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = [n for decl in re.findall(r"(?:const\s+)?(?:void|float|int)\s*\*?\s*([^;{]+);", body) for n in re.findall(r"\*?\s*([a-zA-Z_0-9]+)", decl)]
    assert fields == [n for n, _ in _lib.ConvDesc._fields_], "ConvDesc field order differs from cvmi_conv_desc"

    class ShortConvDesc(ctypes.Structure):                # the stale documented form: everything but the last field
        _fields_ = _lib.ConvDesc._fields_[:-1]
    with pytest.raises(_lib.CvmiError, match="ABI mismatch"):
        _lib.check_abi(lib, {0: ShortConvDesc})
    # the binding INTEGRATION.md documents is the full struct
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert '"row_stats"' in doc and "cvmi_desc_size" in doc


def test_product_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from circuitvision_amd import CvmiError
    from circuitvision_amd.detector import YOLO
    with pytest.raises(CvmiError, match="no CPU fallback"):
        YOLO("synthetic:n:62")


def test_row_stats_predicate_mirrors_the_gemm_dispatch():
    """`engine.row_stats_supported` decides on the host whether cvmi_conv2d will take the 256 x 192 GEMM -- the only kernel that writes
    cvmi_conv_desc.row_stats (LayerNorm statistics for the next launch).  It must say yes exactly for shapes igemm.hip::launch_typed sends
    there: N a multiple of 192 that 256-wide tiles would waste, K >= 1024, >= 256 tiles with >= 75 % of the last round used."""
    from circuitvision_amd.engine import row_stats_supported as ok
    assert ok(65536, 576, 2304)                  # Hiera-L stage-3 fc2 at B = 16: 256 x 3 tiles
    assert ok(32768, 576, 2304)                  # B = 8 (one rank's share of configs[3]): 384 tiles = 75 % of two rounds (r04: 119 us against 140 on the 128-row kernel)
    assert not ok(24576, 576, 2304)              # B = 6: 288 tiles = 56 % of two rounds -> the 128-row kernels take it
    assert not ok(8192, 576, 2304)               # B = 2: 96 tiles, less than one per CU
    assert not ok(65536, 1152, 4608)             # stage 4: 1152 = 4.5 x 256, 90 % column use -> the 256 x 256 kernel
    assert not ok(65536, 576, 576)               # K below the 256 x 192 kernel's threshold
    assert not ok(65536, 560, 2304)              # N not a multiple of 192


def test_counted_lds_rings_hold_no_scalar_memory_instruction():
    """the hand-counted `s_waitcnt lgkmcnt(N > 0)` rings of tok_linear16.hip / hiera_mlp.hip / tok_linear.hip are only valid while
    hipcc emits no scalar-memory instruction (same counter, out-of-order return) between a ring `ds_read_b128` and the MFMA that consumes it.
    tools/check_ring_asm.py disassembles the device code of both operand builds and fails on any; the checker itself is exercised on a
    synthetic stream first."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_ring_asm", os.path.join(ROOT, "tools", "check_ring_asm.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    fake = "\n".join(["_Zk:", ";;#ASMSTART", "ds_read_b128 v[0:3], v9 offset:0", ";;#ASMEND", "s_load_dwordx2 s[0:1], s[4:5], 0x0", ";;#ASMSTART",
                      "s_waitcnt lgkmcnt(0)", ";;#ASMEND", "s_load_dword s2, s[4:5], 0x8", ".end_amdhsa_kernel"])
    k, s, bad = m.check(fake)
    assert (k, s) == (1, 1) and [b[2] for b in bad] == ["s_load_dwordx2 s[0:1], s[4:5], 0x0"]      # inside the window: flagged; after the drain: not
    if not os.path.exists(m.HIPCC):
        pytest.skip("hipcc not available")
    assert m.main(["tok_linear16.hip", "hiera_mlp.hip", "tok_linear.hip"]) == 0


def test_attention_kernels_stay_inside_their_occupancy_register_budget():
    """The global / windowed head_dim-72 attention kernels are designed for four waves per SIMD (two 8-wave workgroups per CU): <= 128 registers,
    nothing spilled.  `__launch_bounds__` does not enforce that -- attn_dma72_kernel drifted to 139 registers in r03 and ran one workgroup per
    CU, correct and 13 % slower.  tools/check_occupancy.py reads the kernel descriptors of the device assembly; the extraction is exercised on
    a synthetic descriptor first."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_occupancy", os.path.join(ROOT, "tools", "check_occupancy.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    fake = "    .name:           _Zfoo_kernelILi8ELb0EE\n    .sgpr_count:     40\n    .vgpr_count:     139\n    .vgpr_spill_count: 0\n"
    t = m.parse_table(fake)
    assert t == {"_Zfoo_kernelILi8ELb0EE": (139, 0)}
    assert m.check(t, {"foo_kernelILi8ELb0EE": 128}) and not m.check(t, {"foo_kernelILi8ELb0EE": 168}) and m.check(t, {"bar": 128})
    if not os.path.exists(m.HIPCC):
        pytest.skip("hipcc not available")
    src, (extra, budgets) = next(iter(m.BUDGETS.items()))
    assert m.check(m.kernel_table(src, extra), budgets) == []


def test_asm_loads_of_the_persistent_fc2_kernel_are_not_touched_before_their_wait():
    """gemm256x192r_kernel loads its residual / bias by inline asm so that hipcc's waitcnt pass does not drain the DMA for them; the price is that the
    compiler believes the destination registers valid at once.  tools/check_asm_loads.py fails if any instruction between such a load and the next
    `s_waitcnt vmcnt` touches its destination (a copy, a spill, an accumulate) -- hipcc did that to another kernel in r03.  The checker is
    exercised on a synthetic stream first."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_asm_loads", os.path.join(ROOT, "tools", "check_asm_loads.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    good = [";;#ASMSTART", "global_load_dwordx4 v[4:7], v[2:3], off offset:64", ";;#ASMEND", "v_mov_b32_e32 v9, v8", ";;#ASMSTART",
            "s_waitcnt vmcnt(6)", ";;#ASMEND", "v_pk_add_f32 v[100:101], v[4:5], v[100:101]"]
    bad = good[:3] + ["v_mov_b32_e32 v89, v5"] + good[3:]
    assert m.check(good) == (1, [])
    seen, hits = m.check(bad)
    assert seen == 1 and [h[2] for h in hits] == [[5]]
    if not os.path.exists(m.HIPCC):
        pytest.skip("hipcc not available")
    body = m.kernel_body(m.device_asm("igemm.hip"), "gemm256x192r_kernel")
    assert body is not None
    seen, hits = m.check(body)
    assert seen == 30 and hits == [], hits[:3]


def test_asm_mfmas_of_the_pipelined_mlp_loop_keep_their_wait_states():
    """hiera_mlp_kernel<288, 2, 2> issues its MFMAs from inline asm, where hipcc's hazard recogniser sees nothing: a VALU result needs wait states
    before an MFMA reads it, an MFMA result before anything else reads it.  The source provides them by construction (`s_nop` statements the operands
    pass through); tools/check_asm_mfma.py verifies on the device assembly of both operand-type builds that hipcc left them intact -- it moved a
    LayerNorm conversion to 0 wait states in front of a consuming block in r04 (wrong results in whole waves).  The checker is exercised on
    synthetic streams first."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_asm_mfma", os.path.join(ROOT, "tools", "check_asm_mfma.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    mfma = "v_mfma_f32_32x32x16_f16 v[2:17], v[46:49], v[90:93], v[2:17]"
    good = ["v_cvt_pk_f16_f32 v93, v36, v37", "s_nop 1", mfma] + ["s_nop 3"] * 3 + ["v_add_f32_e32 v40, v2, v3"]
    assert m.check(good) == (1, [])
    n, bad = m.check(["v_cvt_pk_f16_f32 v93, v36, v37", "s_waitcnt lgkmcnt(5)", mfma])
    assert n == 1 and len(bad) == 1 and bad[0].startswith("VALU -> MFMA, 1 wait states")
    n, bad = m.check([mfma, "s_nop 7", "v_add_f32_e32 v40, v2, v3"])
    assert n == 1 and len(bad) == 1 and bad[0].startswith("MFMA -> read, 8 wait states")
    assert m.check([mfma, mfma.replace("v[46:49]", "v[50:53]"), "s_nop 15", "v_add_f32_e32 v40, v2, v3"]) == (2, [])       # the chain restarts the count
    if not os.path.exists(m.HIPCC):
        pytest.skip("hipcc not available")
    for extra in ((), ("-DCVMI_OPERAND_BF16",)):
        n, bad = m.check(m.kernel_body(m.device_asm("hiera_mlp.hip", extra), m.KERNEL))
        assert n == 148 and bad == [], bad[:3]
