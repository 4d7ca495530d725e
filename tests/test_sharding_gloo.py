"""CPU: the N > 1 path (batch-axis sharding + one-time bucketed weight broadcast) with the
gloo backend, world_size 2 and 3, one process per rank as torch.distributed.run launches them."""
import os
import socket
import subprocess
import sys

import pytest

from cnns_slfp_quantization_amd import sharding

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_world(world):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1", PYTHONDONTWRITEBYTECODE="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_gloo_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {rank} ok" in out, out[-3000:]


def test_shard_range_edges():
    assert sharding.shard_range(256, 0, 1) == (0, 256)
    assert [sharding.shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert [sharding.shard_range(2, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]  # empty shards
    with pytest.raises(ValueError):
        sharding.shard_range(8, 4, 4)
    # without an initialised process group the collectives are identity (the N = 1 path of bench.py)
    import torch
    b = [torch.arange(5, dtype=torch.uint8)]
    assert sharding.broadcast_blobs(b) is b and sharding.gather_outputs(b[0]) is b[0]
