"""The multi-rank stepper (ghost records, interior / halo ranges, one exchange per stage) with real kernels on a box that
has ONE GPU: four ranks share the device and talk through gloo.  gloo is not stream-aware for device tensors, so the
stepper synchronises before it posts the exchange (RMH_SYNC_EXCHANGE=1; with the nccl/RCCL backend the process group's
stream ordering does that).  The 2x2x1-partitioned run must equal the single-rank run bit for bit
(tools/two_ranks_one_gpu.py asserts it)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("lo", [5, 4])
def test_four_ranks_on_one_gpu(lo):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RMH_SYNC_EXCHANGE="1", LO=str(lo), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tools", "two_ranks_one_gpu.py")]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "bitwise equal to the single-rank run: True" in out.stdout
