"""rpgp_comm (one-shot / two-shot all-reduce over IPC-mapped peer buffers) with 2, 3 and 4 ranks that all use device 0:
the only multi-rank GPU collective that can run on the one-GPU test box (RCCL refuses several ranks per device)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(script, world, timeout=600, extra_env=None):
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", script)]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)


@pytest.mark.parametrize("world", [2, 3, 4])
def test_ipc_allreduce_multi_process_one_device(gpu_device, world):
    r = _launch("ipc_child.py", world)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "IPC_CHILD_OK world=%d" % world in r.stdout


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_native_executor_multi_process_one_device(gpu_device, world):
    """The native mBCG executor in both sharded modes (partial products of the pair- / J-sharded exact operator and of
    the pair-sharded packed cache; row-sharded SKI) with 2 and 3 ranks on device 0, against the unsharded native solve."""
    r = _launch("ipc_solve_child.py", world, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "IPC_SOLVE_CHILD_OK world=%d" % world in r.stdout
