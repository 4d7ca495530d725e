"""GPU (MI355X): the multi-GPU path's collectives on the REAL backend.  The builder's box has one GPU, so this is a
world of one rank -- but every call bench.py makes for N > 1 goes through RCCL on device tensors (the gloo tests on the
CPU cover the N > 1 bookkeeping; the N = 2/4/8 numbers are the driver's)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_rccl_single_rank_runs_the_bench_collectives_on_device_tensors():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONDONTWRITEBYTECODE="1")
    p = subprocess.run([sys.executable, os.path.join(HERE, "_rccl_worker.py")], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=300)   # a fresh process: its own HIP / RCCL state
    assert p.returncode == 0 and "rccl rank 0 ok" in p.stdout, p.stdout[-3000:]
