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
    return _broadcast_tensors(packed_tensors(packed), src, bucket_bytes)


def _broadcast_tensors(tensors, src, bucket_bytes):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
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


def weight_tensors(weights):
    """Every device tensor of a prepared model (Yolo11Weights / Sam2Weights): packed convs, LayerNorm affine pairs, folded
    constants, refinement parameters -- what a rank needs to run without reading the checkpoint itself."""
    out = []
    for name in ("packed", "pc", "mlp", "tl"):
        d = getattr(weights, name, None)
        if d:
            out += packed_tensors(d)
    for key in sorted(getattr(weights, "ln", {}) or {}):
        out += list(weights.ln[key])
    for key in sorted(getattr(weights, "const", {}) or {}):
        out.append(weights.const[key])
    rp = getattr(weights, "refine_params", None)
    if rp is not None:
        out.append(rp)
    return out


def broadcast_weights(weights, src=0, bucket_bytes=64 << 20):
    """One-time broadcast of a whole prepared model from `src` (RCCL over xGMI; gloo in the CPU tests)."""
    return _broadcast_tensors(weight_tensors(weights), src, bucket_bytes)


def gather_rows(t, dst=0):
    """Gather tensors whose FIRST dimension differs between ranks (uneven shards: shard_range hands out sizes that differ by
    one when total % world != 0) on `dst`: sizes are exchanged first, every rank pads to the largest shard, `dst` trims.
    Returns the list of per-rank tensors on `dst`, None elsewhere."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [t]
    world, rank = dist.get_world_size(), dist.get_rank()
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(x.item()) for x in sizes]
    m = max(sizes)
    pad = t if t.shape[0] == m else torch.cat((t, t.new_zeros((m - t.shape[0],) + tuple(t.shape[1:]))), 0)
    pad = pad.contiguous()
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    return [b[:k] for b, k in zip(bufs, sizes)] if rank == dst else None


def gather_detections(det, count, dst=0):
    """Gather per-rank [B_r, max_det, 6] detections and [B_r] counts on `dst` (None elsewhere); B_r may differ between ranks."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [det], [count]
    return gather_rows(det, dst), gather_rows(count, dst)
