"""Worker for tests/test_sharding_gloo.py: one rank of a world_size-N gloo job on CPU.
Exercises exactly the host logic bench.py uses for N > 1 (cnns_slfp_quantization_amd/sharding.py)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cnns_slfp_quantization_amd import layer_specs, sharding  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # 1. batch-axis sharding: contiguous, disjoint, covering
    total = 1024 + 3
    lo, hi = sharding.shard_range(total, rank, world)
    spans = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(spans, torch.tensor([lo, hi]))
    spans = [tuple(int(v) for v in s) for s in spans]
    assert spans[0][0] == 0 and spans[-1][1] == total
    assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
    assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
    # 2. one-bucket broadcast of per-layer "prepared weight" blobs (sizes of the real MobileNetV1 layers)
    specs = layer_specs.conv_layers("mobilenetv1_imagenet224")
    g = torch.Generator().manual_seed(7)  # same seed on every rank -> rank 0's payload is known everywhere
    want = [torch.randint(0, 256, (s.w_elems * 2 % 4099 + 17,), dtype=torch.uint8, generator=g) for s in specs]
    blobs = [w.clone() if rank == 0 else torch.zeros_like(w) for w in want]
    sharding.broadcast_blobs(blobs, src=0)
    assert all(torch.equal(a, b) for a, b in zip(blobs, want)), "broadcast payload differs from rank 0's"
    # 3. per-shard outputs gather back in rank order
    local = torch.arange(lo, hi, dtype=torch.float32).reshape(-1, 1).repeat(1, 3)
    if (hi - lo) * world == total:  # all_gather needs equal shapes
        full = sharding.gather_outputs(local)
        assert torch.equal(full[:, 0], torch.arange(total, dtype=torch.float32))
    else:
        pad = torch.full((max(h - l for l, h in spans) - (hi - lo), 3), -1.0)
        full = sharding.gather_outputs(torch.cat([local, pad]))
        assert torch.equal(full[full[:, 0] >= 0][:, 0], torch.arange(total, dtype=torch.float32))
    # 4. weak-scaling bookkeeping used by bench.py: value = images of ALL ranks / max time over ranks
    t = torch.tensor([0.5 + 0.25 * rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert abs(float(t) - (0.5 + 0.25 * (world - 1))) < 1e-12
    # 5. bench.py's N > 1 control flow, step by step, on CPU stand-ins: per-rank batch (weak and strong scaling), the
    #    weights as u8 CODES (1 B per weight) broadcast in one bucket from rank 0, every rank's time on every rank
    lo_w, hi_w, strong = sharding.per_rank_batch(256, 0, rank, world)
    assert (lo_w, hi_w, strong) == (rank * 256, (rank + 1) * 256, False)
    lo_s, hi_s, strong = sharding.per_rank_batch(256, 1024 * world, rank, world)   # BASELINE config 4's shape: G / world each
    assert strong and hi_s - lo_s == 1024 and lo_s == rank * 1024
    g2 = torch.Generator().manual_seed(11)
    codes_want = [torch.randint(0, 256, (s.w_elems,), dtype=torch.uint8, generator=g2) for s in specs]   # 1 B per weight
    assert sum(int(c.numel()) for c in codes_want) == sum(s.w_elems for s in specs) == 3185088        # 3.19 MB
    codes = [c.clone() if rank == 0 else torch.empty_like(c) for c in codes_want]
    sharding.broadcast_blobs(codes, src=0)
    assert all(torch.equal(a, b) for a, b in zip(codes, codes_want))
    times = sharding.rank_times(0.01 * (rank + 1))
    assert len(times) == world and all(abs(times[r] - 0.01 * (r + 1)) < 1e-12 for r in range(world))
    value = 256 * world * 10 / max(times)      # whole-job images / slowest rank's seconds, as bench.py reports it
    assert abs(value - 256 * world * 10 / (0.01 * world)) < 1e-6
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok")


if __name__ == "__main__":
    main()
