"""The multi-GPU collectives on the real backend ("nccl" = RCCL), one rank per visible GPU, launched as fresh child
processes through torch.distributed.run — the same way bench.py --gpus N and the runner start their ranks.  On the
1-GPU test box this is world size 1 (every collective still goes through RCCL); on an 8-GPU node it is world size 8."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sharded_paths_over_rccl(gpu_device):
    import socket
    import torch
    n = torch.cuda.device_count()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "rccl_child.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "RCCL_CHILD_OK world=%d" % n in r.stdout
