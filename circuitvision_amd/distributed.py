"""Multi-GPU plumbing: image-parallel sharding, one RCCL weight broadcast, optional result gather.

The reference has no distributed code at all; every circuit image is
independent through both models, so a batch shards contiguously over ranks with NO per-step
collective (north_star).  One process per GPU; `torch.distributed` backend "nccl"
is RCCL on ROCm (xGMI), "gloo" is used by the CPU tests.
"""
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous [lo, hi) share of `total` images for `rank`; sizes differ by at most one."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def packed_tensors(packed):
    """Deterministic (sorted by key) list of the device tensors of a dict of PackedConv / PackedDW."""
    out = []
    for key in sorted(packed):
        out.append(packed[key].w)
        out.append(packed[key].bias)
    return out


def broadcast_packed(packed, src=0, bucket_bytes=64 << 20):
    """Broadcast every packed weight tensor from `src`.  Tensors are coalesced into flat buckets per
    dtype (a 1->7 xGMI fan-out is per-link bound, so few large messages beat hundreds of small ones)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    tensors = packed_tensors(packed)
    sent = 0
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    for dt, ts in by_dtype.items():
        bucket, size = [], 0
        for t in ts + [None]:
            if t is not None and (size == 0 or size + t.numel() * t.element_size() <= bucket_bytes):
                bucket.append(t)
                size += t.numel() * t.element_size()
                continue
            flat = torch.cat([b.reshape(-1) for b in bucket])
            dist.broadcast(flat, src=src)
            off = 0
            for b in bucket:
                b.copy_(flat[off:off + b.numel()].view_as(b))
                off += b.numel()
            sent += size
            bucket, size = ([t], t.numel() * t.element_size()) if t is not None else ([], 0)
    return sent


def gather_detections(det, count, dst=0):
    """Gather per-rank [B, max_det, 6] detections and [B] counts on `dst` (None elsewhere)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [det], [count]
    world = dist.get_world_size()
    dets = [torch.empty_like(det) for _ in range(world)] if dist.get_rank() == dst else None
    cnts = [torch.empty_like(count) for _ in range(world)] if dist.get_rank() == dst else None
    dist.gather(det, dets, dst=dst)
    dist.gather(count, cnts, dst=dst)
    return dets, cnts
