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
    from circuitvision_amd.yolo11 import SyntheticParams, Yolo11Weights
    # every rank starts from DIFFERENT weights; after the broadcast all must equal rank 0's
    wt = Yolo11Weights("n", 62, SyntheticParams(seed=100 + rank, nc=62), F32, device="cpu")
    ref = Yolo11Weights("n", 62, SyntheticParams(seed=100, nc=62), F32, device="cpu")
    sent = broadcast_packed(wt.packed, src=0, bucket_bytes=1 << 20)
    same = all(torch.equal(a, b) for a, b in zip(packed_tensors(wt.packed), packed_tensors(ref.packed)))
    det = torch.full((3, 4, 6), float(rank))
    cnt = torch.full((3,), rank, dtype=torch.int32)
    dets, cnts = gather_detections(det, cnt, dst=0)
    ok_gather = True
    if rank == 0:
        ok_gather = all(float(dets[r].mean()) == r and int(cnts[r][0]) == r for r in range(world))
    out[rank] = (same, sent > 0, ok_gather)
    dist.barrier()
    dist.destroy_process_group()


def test_weight_broadcast_and_gather_gloo_world2():
    world, port = 2, 29000 + os.getpid() % 2000
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {0: (True, True, True), 1: (True, True, True)}
