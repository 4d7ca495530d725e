"""Batch-axis sharding of the quantized conv path over the GPUs of one node.

The reference has no distributed code at all.  Images are independent
units through the whole hot path, so the MI355X design is: one process per GPU
(torch.distributed, backend "nccl" == RCCL over xGMI), each rank owns a contiguous slice
of the batch, and the ONLY collective is a one-time broadcast of the quantized (prepared)
weight blobs from rank 0 -- all layers flattened into one bucket so that it is a single
large RCCL broadcast instead of 27-53 small ones.  There is no per-step exchange.
What travels is the weights' u8 codes (1 B per weight, SURVEY 8e: 3.19 MB for MobileNetV1,
23.5 MB for ResNet-50); every rank lays them out for its own kernels
(slfp_conv2d_prepare_weights_codes).  The same code runs on CPU tensors with the gloo backend (tests);
bench.py's N > 1 control flow is exactly these functions: per_rank_batch, broadcast_blobs, rank_times.
"""
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous [lo, hi) slice of `total` items owned by `rank` (sizes differ by <= 1)."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _single(group, force):
    """No process group, or a world of one: the collectives are the identity -- unless `force` asks for the call to go
    through the backend anyway (a one-rank RCCL collective is legal: tests/test_gpu_rccl.py runs the real code path)."""
    if not (dist.is_available() and dist.is_initialized()):
        return True
    return dist.get_world_size(group) == 1 and not force


def broadcast_blobs(blobs, src=0, group=None, force=False):
    """Broadcast a list of uint8 tensors (the per-layer prepared weight blobs) from `src`
    as ONE flat bucket; the tensors are overwritten in place on the other ranks."""
    if not blobs:
        return blobs
    if _single(group, force):
        return blobs
    sizes = [int(b.numel()) for b in blobs]
    flat = torch.empty(sum(sizes), dtype=torch.uint8, device=blobs[0].device)
    if dist.get_rank(group) == src:
        off = 0
        for b, n in zip(blobs, sizes):
            flat[off:off + n].copy_(b.reshape(-1))
            off += n
    dist.broadcast(flat, src=src, group=group)
    off = 0
    for b, n in zip(blobs, sizes):
        b.reshape(-1).copy_(flat[off:off + n])
        off += n
    return blobs


def gather_outputs(local, group=None, force=False):
    """Optional final all-gather of per-rank outputs (e.g. logits) along dim 0."""
    if _single(group, force):
        return local
    parts = [torch.empty_like(local) for _ in range(dist.get_world_size(group))]
    dist.all_gather(parts, local.contiguous(), group=group)
    return torch.cat(parts, dim=0)


def per_rank_batch(batch, global_batch, rank, world):
    """Images this rank processes per step: `batch` (weak scaling) or its contiguous slice of `global_batch`
    (strong scaling, BASELINE config 4: 1024 -> 128 per GPU on 8 GPUs).  Returns (lo, hi, strong)."""
    if global_batch and global_batch > 0:
        lo, hi = shard_range(global_batch, rank, world)
        return lo, hi, True
    return rank * batch, (rank + 1) * batch, False


def rank_times(seconds, device=None, group=None, force=False):
    """Every rank's timed-region seconds, on every rank: the job's time is the MAX (the slowest rank), and the
    rank-0 line reports all of them."""
    if _single(group, force):
        return [float(seconds)]
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    parts = [torch.zeros_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(parts, t, group=group)
    return [float(p.item()) for p in parts]
