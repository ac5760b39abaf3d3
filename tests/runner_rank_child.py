"""Rank program for the CPU test of the runner's one-command multi-device form (`--device cpu,cpu`): the same
`rpgp_amd.runner.main` a real rank runs, with the CPU test double installed first (the package itself has no CPU path).
Started by `runner.launch_ranks(..., entry=[this file])` under torch.distributed.run; flags arrive in RPGP_RUNNER_ARGV."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from rpgp_amd import backend, runner  # noqa: E402
from tests.oracle_backend import OracleBackend  # noqa: E402

backend.set_backend(OracleBackend())
runner.main()
import torch.distributed as dist  # noqa: E402

marker = os.environ.get("RPGP_TEST_MARKER_DIR")
if marker:
    open(os.path.join(marker, "rank%s_world%d" % (os.environ["RANK"], dist.get_world_size())), "w").write("ok")
if dist.is_initialized():
    dist.destroy_process_group()
