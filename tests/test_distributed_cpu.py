"""world_size-2 gloo tests of the N>1 path (sharding + one-time weight broadcast + result gather)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    from circuitvision_amd.distributed import shard_range
    for total in (0, 1, 7, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from circuitvision_amd._lib import F32
    from circuitvision_amd.distributed import broadcast_packed, gather_detections, packed_tensors
    from circuitvision_amd.yolo11 import BlankParams, SyntheticParams, Yolo11Weights
    # rank 0 reads the "checkpoint"; rank 1 holds NONE (blank buffers of the right shapes): after the broadcast all must equal rank 0's
    wt = Yolo11Weights("n", 62, SyntheticParams(seed=100, nc=62) if rank == 0 else BlankParams(), F32, device="cpu")
    ref = Yolo11Weights("n", 62, SyntheticParams(seed=100, nc=62), F32, device="cpu")
    sent = broadcast_packed(wt.packed, src=0, bucket_bytes=1 << 20)
    same = all(torch.equal(a, b) for a, b in zip(packed_tensors(wt.packed), packed_tensors(ref.packed)))
    # the SAM 2 side: a mini model packed from blank parameters on rank 1 receives everything a forward pass reads
    from circuitvision_amd.distributed import broadcast_weights, weight_tensors
    from circuitvision_amd.sam2 import Sam2Weights, SamBlankParams, SamSyntheticParams
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_oracle_sam2_cpu import MINI, mini_targets
    sp = SamSyntheticParams(seed=3, lora_targets=mini_targets(), std=0.05)
    sw = Sam2Weights(sp if rank == 0 else SamBlankParams(), MINI, 256, F32, device="cpu")
    sref = Sam2Weights(sp, MINI, 256, F32, device="cpu")
    broadcast_weights(sw, src=0, bucket_bytes=1 << 20)
    a_, b_ = weight_tensors(sw), weight_tensors(sref)
    same = same and len(a_) == len(b_) > 10 and all(torch.equal(x, y) for x, y in zip(a_, b_))
    det = torch.full((3, 4, 6), float(rank))
    cnt = torch.full((3,), rank, dtype=torch.int32)
    dets, cnts = gather_detections(det, cnt, dst=0)
    ok_gather = True
    if rank == 0:
        ok_gather = all(float(dets[r].mean()) == r and int(cnts[r][0]) == r for r in range(world))
    out[rank] = (same, sent > 0, ok_gather)
    dist.barrier()
    dist.destroy_process_group()


def _all_tensors(obj, seen=None, depth=0):
    """Every torch tensor reachable from a prepared-weights object (attributes, dict / list / tuple members), by storage pointer."""
    seen = {} if seen is None else seen
    if torch.is_tensor(obj):
        seen[obj.data_ptr()] = obj
    elif isinstance(obj, dict):
        for v in obj.values():
            _all_tensors(v, seen, depth + 1)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _all_tensors(v, seen, depth + 1)
    elif hasattr(obj, "__dict__") and depth < 4 and type(obj).__module__.startswith("circuitvision_amd"):
        for k, v in vars(obj).items():
            if k not in ("p", "params"):                              # the parameter SOURCE (checkpoint side), not the prepared model
                _all_tensors(v, seen, depth + 1)
    return seen


def test_broadcast_covers_every_tensor_of_the_prepared_models():
    """a rank without a checkpoint only works if the one-time broadcast carries EVERY device tensor a forward pass
    reads.  Everything reachable from Yolo11Weights / Sam2Weights must be in the broadcast list."""
    from circuitvision_amd._lib import F32
    from circuitvision_amd.distributed import packed_tensors, weight_tensors
    from circuitvision_amd.sam2 import Sam2Weights, SamSyntheticParams
    from circuitvision_amd.yolo11 import SyntheticParams, Yolo11Weights
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_oracle_sam2_cpu import MINI, mini_targets
    yw = Yolo11Weights("n", 62, SyntheticParams(seed=1, nc=62), F32, device="cpu")
    sent = {t.data_ptr() for t in packed_tensors(yw.packed)}
    missing = [tuple(t.shape) for p, t in _all_tensors(yw).items() if p not in sent and t.numel() > 0]
    assert not missing, f"YOLO11 tensors outside the broadcast: {missing}"
    sw = Sam2Weights(SamSyntheticParams(seed=3, lora_targets=mini_targets(), std=0.05), MINI, 256, F32, device="cpu")
    sent = {t.data_ptr() for t in weight_tensors(sw)}
    missing = [tuple(t.shape) for p, t in _all_tensors(sw).items() if p not in sent and t.numel() > 0]
    assert not missing, f"SAM 2 tensors outside the broadcast: {missing}"


def test_weight_broadcast_and_gather_gloo_world2():
    world, port = 2, 29000 + os.getpid() % 2000
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {0: (True, True, True), 1: (True, True, True)}


# ---- pipeline sharding (BASELINE configs[3]): sharded == unsharded, uneven shards, result gather -----------------------------
class _FakeResults:
    def __init__(self, det, names):
        from circuitvision_amd.detector import Boxes
        self.boxes, self.names = Boxes(det), names


class _FakeDetector:
    """Deterministic, per-image stand-in for `YOLO` (CPU): boxes derived from the image content only."""
    names = {i: f"c{i}" for i in range(5)}

    def predict(self, images, verbose=False, **_):
        import numpy as np
        out = []
        for im in images:
            rng = np.random.default_rng(int(im.astype(np.int64).sum()) % (1 << 31))
            n = int(rng.integers(3, 9))
            h, w = im.shape[:2]
            xy = rng.uniform(0, 0.6, (n, 2)) * (w, h)
            wh = rng.uniform(0.1, 0.4, (n, 2)) * (w, h)
            conf = np.sort(rng.uniform(0.3, 1.0, n))[::-1]
            det = np.concatenate((xy, xy + wh, conf[:, None], rng.integers(0, 5, (n, 1))), 1)
            out.append(_FakeResults(torch.from_numpy(det).float(), self.names))
        return out


class _FakeSegmenter:
    image_size = 32

    def infer_masks(self, x, boxes=None, return_high_res=True):
        hi = x.mean(1, keepdim=True) - x.mean((1, 2, 3), keepdim=True)          # per image: no cross-image term
        lo = hi[..., ::4, ::4]
        if boxes is None:
            return hi, lo, hi.flatten(1).mean(1, keepdim=True)
        B, P = boxes.shape[:2]
        lo = lo.expand(B, P, -1, -1) + boxes[..., :1, None] * 1e-3
        return None, lo, boxes.sum(-1)


class _FakeTransforms:
    resolution = 32

    def forward_batch(self, imgs):
        import torch.nn.functional as F
        return torch.stack([F.interpolate(torch.from_numpy(i.copy()).permute(2, 0, 1)[None].float() / 255, (32, 32), mode="bilinear")[0] for i in imgs])

    def transform_boxes(self, boxes, normalize=False, orig_hw=None):
        from circuitvision_amd.sam2_infer import SAM2Transforms
        return SAM2Transforms.transform_boxes(self, boxes, normalize, orig_hw)

    def transform_coords(self, coords, normalize=False, orig_hw=None):
        from circuitvision_amd.sam2_infer import SAM2Transforms
        return SAM2Transforms.transform_coords(self, coords, normalize, orig_hw)

    def postprocess_to_mask(self, masks, orig_hw):
        import torch.nn.functional as F
        m = F.interpolate(masks.float(), tuple(orig_hw), mode="bilinear", align_corners=False)
        u8 = (m > 0).to(torch.uint8) * 255
        ext = []
        for pl in u8.flatten(0, 1):
            ys, xs = torch.nonzero(pl, as_tuple=True)
            ext.append(None if ys.numel() == 0 else (int(xs.min()), int(ys.min()), int(xs.max()) + 1, int(ys.max()) + 1))
        return u8, ext


def _fake_pipeline():
    from circuitvision_amd.pipeline import CircuitPipeline
    return CircuitPipeline(_FakeDetector(), _FakeSegmenter(), _FakeTransforms(), max_prompts=4)


def _pipeline_images(n):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from synth import circuit_image
    return [circuit_image(48, 64, seed=70 + i) for i in range(n)]


def _pipeline_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from circuitvision_amd.pipeline import gather_results
    images = _pipeline_images(7)                                    # 7 images over 2 ranks: shards of 4 and 3 (uneven)
    pipe = _fake_pipeline()
    res = pipe.run_batch(images, "learned", rank, world)
    masks = gather_results(res, "mask", dst=0)
    resb = pipe.run_batch(images, "boxes", rank, world)
    out[rank] = ([i for i, _ in res], [[b["persistent_uid"] for b in r["bboxes"]] for _, r in res],
                 {k: v.clone() for k, v in masks.items()} if masks is not None else None,
                 [(i, r["masks"].clone(), r["iou"].clone()) for i, r in resb])
    dist.barrier()
    dist.destroy_process_group()


def test_pipeline_sharded_equals_unsharded_gloo_world2():
    """configs[3] data path: each rank runs detector -> stage-2 NMS -> segmenter on its contiguous share (7 images -> 4 + 3);
    the union of the ranks' results equals the single-process run image for image, and the uneven result gather works."""
    world, port = 2, 31000 + os.getpid() % 2000
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_pipeline_worker, args=(world, port, out), nprocs=world, join=True)
    images = _pipeline_images(7)
    pipe = _fake_pipeline()
    ref = pipe.run_batch(images, "learned")
    refb = pipe.run_batch(images, "boxes")
    got = dict(out)
    assert got[0][0] == [0, 1, 2, 3] and got[1][0] == [4, 5, 6]
    uids = got[0][1] + got[1][1]
    assert uids == [[b["persistent_uid"] for b in r["bboxes"]] for _, r in ref] and all(len(u) > 0 for u in uids)
    gathered = got[0][2]
    assert got[1][2] is None and sorted(gathered) == list(range(7))
    for i, r in ref:
        assert torch.equal(gathered[i], r["mask"])
    shard_b = got[0][3] + got[1][3]
    for (i, m, iou), (j, r) in zip(shard_b, refb):
        assert i == j and torch.equal(m, r["masks"]) and torch.equal(iou, r["iou"]) and m.shape[0] == min(4, len(r["bboxes"])) > 0


def _empty_shard_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from circuitvision_amd.pipeline import gather_results
    images = _pipeline_images(1)                                    # 1 image over 2 ranks: rank 1 holds nothing
    res = _fake_pipeline().run_batch(images, "learned", rank, world)
    masks = gather_results(res, "mask", dst=0)                      # must not hang or raise on the empty rank
    none_at_all = gather_results([], "mask", dst=0)                 # no rank has a result
    out[rank] = (len(res), None if masks is None else {k: v.clone() for k, v in masks.items()}, none_at_all)
    dist.barrier()
    dist.destroy_process_group()


def test_gather_results_with_an_empty_shard_gloo_world2():
    """a rank whose shard is empty (world > number of images) takes part in the gather with zero rows instead of raising
    before the collective (which left the other ranks hanging)."""
    world, port = 2, 35000 + os.getpid() % 2000
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_empty_shard_worker, args=(world, port, out), nprocs=world, join=True)
    ref = _fake_pipeline().run_batch(_pipeline_images(1), "learned")
    assert out[0][0] == 1 and out[1][0] == 0
    assert out[1][1] is None and sorted(out[0][1]) == [0] and torch.equal(out[0][1][0], ref[0][1]["mask"])
    assert out[0][2] == {} and out[1][2] is None


def _uneven_gather_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from circuitvision_amd.distributed import gather_detections, shard_range
    lo, hi = shard_range(5, rank, world)                            # 3 + 2
    det = torch.arange(lo, hi, dtype=torch.float32).view(-1, 1, 1).expand(-1, 4, 6).contiguous()
    cnt = torch.arange(lo, hi, dtype=torch.int32)
    dets, cnts = gather_detections(det, cnt, dst=0)
    out[rank] = None if dets is None else ([d.shape[0] for d in dets], torch.cat(cnts).tolist(), torch.cat(dets)[:, 0, 0].tolist())
    dist.barrier()
    dist.destroy_process_group()


def test_gather_detections_uneven_shards_gloo_world2():
    """dist.gather needs equal shapes; shards of 3 and 2 images must still gather (pad to the largest, trim on dst)."""
    world, port = 2, 33000 + os.getpid() % 2000
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_uneven_gather_worker, args=(world, port, out), nprocs=world, join=True)
    assert out[1] is None and out[0] == ([3, 2], [0, 1, 2, 3, 4], [0.0, 1.0, 2.0, 3.0, 4.0])


def _world8_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from circuitvision_amd.pipeline import CircuitPipeline, gather_results
    images = _pipeline_images(64)
    pipe = CircuitPipeline(_FakeDetector(), _FakeSegmenter(), _FakeTransforms(), crop=True, crop_padding=4)     # the reference's chain: detector -> crop -> segmenter
    res = pipe.run_batch(images, "learned", rank, world)
    # crops differ in size per image: gather the extents (same shape everywhere) and the window-sized masks padded into the page
    for i, r in res:
        full = torch.zeros(images[i].shape[:2], dtype=torch.uint8)
        w = r["crop_debug_info"]["final_crop_window_abs"] if r["crop_debug_info"]["crop_applied"] else (0, 0, images[i].shape[1], images[i].shape[0])
        full[w[1]:w[3], w[0]:w[2]] = r["mask"][0, 0] if r["mask"].dim() == 4 else r["mask"]
        r["page_mask"] = full
    masks = gather_results(res, "page_mask", dst=3)                 # (a destination other than rank 0)
    out[rank] = ([i for i, _ in res], [[b["persistent_uid"] for b in r["bboxes"]] for _, r in res], [r["crop_debug_info"]["final_crop_window_abs"] for _, r in res],
                 {k: v.clone() for k, v in masks.items()} if masks is not None else None)
    dist.barrier()
    dist.destroy_process_group()


def test_configs3_shape_64_images_over_8_ranks_gloo_world8():
    """BASELINE configs[3] as named: 64 circuit images, image-parallel over 8 ranks (8 each), the reference's chain per image (detector ->
    stage-2 NMS -> crop window -> segmenter on the window), no collective on the data path; one gather of the results on one rank.  The union
    of the 8 shards equals the single-process run image for image (boxes, crop windows, masks)."""
    from circuitvision_amd.distributed import shard_range
    from circuitvision_amd.pipeline import CircuitPipeline
    assert [shard_range(64, r, 8) for r in range(8)] == [(8 * r, 8 * r + 8) for r in range(8)]
    world, port = 8, 37000 + os.getpid() % 2000
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_world8_worker, args=(world, port, out), nprocs=world, join=True)
    images = _pipeline_images(64)
    ref = CircuitPipeline(_FakeDetector(), _FakeSegmenter(), _FakeTransforms(), crop=True, crop_padding=4).run_batch(images, "learned")
    got = dict(out)
    assert [got[r][0] for r in range(8)] == [list(range(8 * r, 8 * r + 8)) for r in range(8)]
    uids = [u for r in range(8) for u in got[r][1]]
    wins = [w for r in range(8) for w in got[r][2]]
    assert uids == [[b["persistent_uid"] for b in r["bboxes"]] for _, r in ref]
    assert wins == [r["crop_debug_info"]["final_crop_window_abs"] for _, r in ref] and sum(w is not None for w in wins) >= 32
    gathered = got[3][3]
    assert all(got[r][3] is None for r in range(8) if r != 3) and sorted(gathered) == list(range(64))
    for i, r in ref:
        w = r["crop_debug_info"]["final_crop_window_abs"] if r["crop_debug_info"]["crop_applied"] else (0, 0, images[i].shape[1], images[i].shape[0])
        m = r["mask"][0, 0] if r["mask"].dim() == 4 else r["mask"]
        assert torch.equal(gathered[i][w[1]:w[3], w[0]:w[2]], m), i
