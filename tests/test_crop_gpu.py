"""GPU side of the crop between the stages: the segmenter's transform reading WINDOWS of the u8 images the detector put in
HBM, masks returning to per-image window sizes, and the chained pipeline with the reference's data dependency
(/root/reference/src/analysis_pipeline.py:177 crop -> :206 segment_with_sam2) against the oracle chain."""
import numpy as np
import pytest
import torch

from circuitvision_amd import _lib
from circuitvision_amd._lib import BF16, F16, F32
from circuitvision_amd.detector import YOLO
from circuitvision_amd.pipeline import CircuitPipeline
from circuitvision_amd.sam2 import SamSyntheticParams
from circuitvision_amd.sam2_infer import SAM2Model, SAM2Transforms
from helpers import assert_same_detections, save_converted_yolo
from oracle import crop as ocrop
from oracle import preprocess as opre
from oracle import sam2_model as osam
from oracle.yolo11 import YOLO11
from synth import calibrated_yolo_params, circuit_image
from test_oracle_sam2_cpu import MINI, mini_oracle, mini_targets
from test_pipeline_gpu import _oracle_chain

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [F32, F16, BF16])
def test_transform_from_windows_is_bit_identical_to_transforming_crops(dtype):
    """`SAM2Transforms.forward_windows` (cvmi_sam2_transform_rects): each image's window of one u8 [B,H,W,3] device block, resized / normalised
    straight into the segmenter's input type == `forward_batch` of contiguous host crops, bit for bit -- windows at the borders, a one-pixel-high
    strip, the whole image (None), up-scaling and 6x down-scaling windows, with and without the channel swap; and vs the CPU oracle's transform."""
    R = 256
    tr = SAM2Transforms(resolution=R, mask_threshold=0, max_hole_area=0, max_sprinkle_area=0)
    H, W = 700, 1500
    imgs = [circuit_image(H, W, seed=50 + b) for b in range(6)]
    wins = [(0, 0, 300, 200), (1201, 401, 1500, 700), None, (10, 350, 1490, 351), (7, 3, 1493, 697), (640, 300, 700, 380)]
    src = torch.from_numpy(np.stack(imgs)).cuda()
    td = {F32: torch.float32, F16: torch.float16, BF16: torch.bfloat16}[dtype]
    for swap in (False, True):
        out = torch.empty(6, R, R, 3, dtype=td, device="cuda")
        tr.forward_windows(src, wins, swap_rb=swap, out=out, out_dtype=dtype)
        for b, (im, w) in enumerate(zip(imgs, wins)):
            crop = im if w is None else np.ascontiguousarray(im[w[1]:w[3], w[0]:w[2]])
            ref = torch.empty(1, R, R, 3, dtype=td, device="cuda")
            tr.forward_batch([crop], swap_rb=swap, out=ref, out_dtype=dtype)
            assert torch.equal(out[b], ref[0]), (b, swap)
            if dtype == F32:
                want = osam.sam2_transform(np.ascontiguousarray(crop[..., ::-1]) if swap else crop, R)
                torch.testing.assert_close(out[b].permute(2, 0, 1).cpu(), want, rtol=1e-5, atol=2e-5)
    x = tr.forward_windows(src, wins)                                    # the torch-op form: fresh f32 [B,3,R,R]
    assert x.shape == (6, 3, R, R) and x.dtype == torch.float32
    lib = _lib.load()
    bad = np.array([[1400, 0, 200, 100]], dtype=np.int32)                # leaves the image: refused, nothing launched
    rc = lib.cvmi_sam2_transform_rects(src.data_ptr(), H * W * 3, H, W, bad.ctypes.data, 1, x.data_ptr(), R, F32, 0, None)
    assert rc != 0 and b"leaves the" in lib.cvmi_last_error()
    with pytest.raises(TypeError):
        tr.forward_windows(src.cpu(), wins)


def test_masks_return_to_per_image_window_sizes_in_one_launch():
    """cvmi_mask_postprocess_sizes: N planes resized to N different sizes, thresholded, packed back to back, with their extents == N calls of
    cvmi_mask_postprocess; 70 planes (more than one argument table)."""
    tr = SAM2Transforms(resolution=256, mask_threshold=0, max_hole_area=0, max_sprinkle_area=0)
    g = torch.Generator().manual_seed(3)
    N = 70
    logits = (torch.randn(N, 1, 64, 64, generator=g) * 3 - 1.0).cuda()
    logits[5] = -4.0                                                     # an empty mask
    sizes = [(int(h), int(w)) for h, w in torch.randint(1, 400, (N, 2), generator=g).tolist()]
    sizes[0], sizes[1] = (1, 1), (399, 3)
    masks, ext = tr.postprocess_to_masks_sized(logits, sizes)
    boxes = tr.extents_to_boxes(ext)
    for n, (h, w) in enumerate(sizes):
        u8, ref_boxes = tr.postprocess_to_mask(logits[n:n + 1], (h, w))
        assert masks[n].shape == (h, w) and torch.equal(masks[n], u8[0, 0]), n
        assert boxes[n] == ref_boxes[0], n
    assert boxes[5] is None


# a label map under which the synthetic detector's (uniformly scattered) boxes give REAL crop windows: few component classes, some junction and
# text classes, the rest a class the clustering ignores (checked on the CPU oracle chain: windows such as (201, 29, 420, 300) of 420 x 300)
NAMES = {i: (f"component{i}" if i % 10 == 0 else "junction" if i % 10 == 1 else "text" if i % 10 == 2 else "explanatory") for i in range(62)}


def _setup(tmp_path, hw=(300, 420), n_images=5, dtype="f32"):
    images = [circuit_image(*hw, seed=300 + i) for i in range(n_images)]
    x = torch.cat([torch.from_numpy(opre.yolo_preprocess(im)) for im in images])
    yp = calibrated_yolo_params("n", 62, 4, x)
    det = YOLO(save_converted_yolo(str(tmp_path / "y.pt"), yp, "n", 62), dtype=dtype)
    det.names = det.model.names = dict(NAMES)                            # a label map with text / junction / crossover classes: the crop reads them
    yo = YOLO11("n", 62).eval()
    yo.load_state_dict(yp.state_dict(), strict=True)
    R = 256
    sp = SamSyntheticParams(seed=8, lora_targets=mini_targets(), std=0.05)
    seg = SAM2Model(MINI, R, dtype=dtype, use_refinement=True).load_params(sp)
    tr = SAM2Transforms(resolution=R, mask_threshold=0, max_hole_area=0, max_sprinkle_area=0)
    return images, det, yo, seg, tr, mini_oracle(sp, R), R


def test_cropped_pipeline_matches_the_oracle_chain(tmp_path):
    """detector -> stage-2 NMS -> crop window (padding 20 here so that the small test images do get cropped; the reference passes 80) ->
    segmenter on the window, f32: boxes, windows and shifted boxes equal the oracle chain's (oracle detector -> oracle NMS -> oracle crop), the
    mask is the oracle's mask of the oracle's crop (<= 1e-3 of the pixels differ), extent = bounding rectangle of the mask; the shard of a
    2-way split equals the unsharded run bit for bit."""
    images, det, yo, seg, tr, so, R = _setup(tmp_path)
    pipe = CircuitPipeline(det, seg, tr, crop=True, crop_padding=20, seg_batch=2)
    full = pipe.run_batch(images, "learned")
    assert any("detector chunk" in k for k in pipe.timings), "the cropped chain takes the chunked fast path on the package's own objects"
    cropped = 0
    for (i, r), im in zip(full, images):
        ref_b = _oracle_chain(yo, det.names, im)
        ref_img, ref_adj, plan = ocrop.crop_image_and_adjust_bboxes(im, ref_b, padding=20)
        nms_uids = [b["persistent_uid"] for b in ref_b]
        got_all = [b["persistent_uid"] for b in r["bboxes"]]
        assert_same_detections(f"cropped pipeline image {i}", got_all, [b["persistent_uid"] for b in ref_adj], top=8, min_overlap=0.9)
        assert r["crop_debug_info"]["crop_applied"] == plan["applied"] and r["window"] == (plan["window"] if plan["applied"] else None), (i, r["window"], plan)
        if [b["persistent_uid"] for b in r["bboxes"]] == [b["persistent_uid"] for b in ref_adj]:
            strip = lambda bs: [{k: v for k, v in b.items() if k != "confidence"} for b in bs]
            assert strip(r["bboxes"]) == strip(ref_adj), i                      # shifted, clipped integer boxes, classes, uids: identical
            assert max(abs(a_["confidence"] - b_["confidence"]) for a_, b_ in zip(r["bboxes"], ref_adj)) < 1e-3
        cropped += plan["applied"]
        assert r["image"].shape == ref_img.shape and np.array_equal(r["image"], ref_img)
        with torch.no_grad():
            rhi, _, _ = so(osam.sam2_transform(np.ascontiguousarray(ref_img[..., ::-1]), R)[None])
            rmask = (osam.postprocess_masks(rhi, ref_img.shape[:2]).squeeze() > 0.0).numpy().astype(np.uint8) * 255
        got = r["mask"].cpu().numpy()
        assert got.shape == ref_img.shape[:2] and (got != rmask).mean() < 1e-3, i
        ys, xs = np.nonzero(got)
        assert r["extent"] == ((int(xs.min()), int(ys.min()), int(xs.max()) + 1, int(ys.max()) + 1) if ys.size else None)
        assert len(nms_uids) >= 8
    assert cropped >= 3, "the test images must exercise real windows"
    sharded = pipe.run_batch(images, "learned", rank=0, world=2) + pipe.run_batch(images, "learned", rank=1, world=2)
    for (i, a), (_, b) in zip(sharded, full):
        assert a["bboxes"] == b["bboxes"] and torch.equal(a["mask"], b["mask"]) and a["extent"] == b["extent"] and a["window"] == b["window"], i


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_cropped_fast_path_equals_host_crop_path_bit_for_bit(tmp_path, dtype):
    """The chunked path (detector chunks, window-source transform, one sized post-process launch per chunk) == the generic path that crops on
    the host with the same function and transforms each crop (`crop_fn`), for equal-sized and ragged image sets: same boxes, windows, masks,
    extents, IoU predictions -- bit for bit."""
    from circuitvision_amd.crop import crop_image_and_adjust_bboxes
    images, det, yo, seg, tr, so, R = _setup(tmp_path, dtype=dtype)
    ragged = images[:3] + [circuit_image(260, 300, seed=77), circuit_image(340, 280, seed=78)]
    for imgs in (images, ragged):
        fast = CircuitPipeline(det, seg, tr, crop=True, crop_padding=20, seg_batch=2)
        slow = CircuitPipeline(det, seg, tr, seg_batch=2, crop_fn=lambda im, bb: crop_image_and_adjust_bboxes(im, bb, padding=20))
        a, b = fast.run_batch(imgs, "learned"), slow.run_batch(imgs, "learned")
        assert any("detector chunk" in k for k in fast.timings) and not any("enqueue" in k for k in slow.timings)
        for (i, ra), (j, rb) in zip(a, b):
            assert i == j and ra["bboxes"] == rb["bboxes"] and ra["crop_debug_info"] == rb["crop_debug_info"], (dtype, i)
            assert np.array_equal(ra["image"], rb["image"]) and torch.equal(ra["mask"], rb["mask"]) and ra["extent"] == rb["extent"], (dtype, i)
            assert torch.equal(ra["iou"], rb["iou"]), (dtype, i)
    # prompts="boxes" with the built-in crop takes the generic path: boxes are the SHIFTED boxes, masks have the window's size
    rb = CircuitPipeline(det, seg, tr, crop=True, crop_padding=20, max_prompts=4).run_batch(images[:2], "boxes")
    for (i, r) in rb:
        assert r["masks"].shape[1:] == r["image"].shape[:2] and len(r["bboxes"]) == r["masks"].shape[0] <= 4
