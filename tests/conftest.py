import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load


@pytest.fixture
def spawn_ranks():
    """Start `world` ranks of a script the way bench.py's parent does: the rendezvous port is bound here and the
    listening socket itself is inherited by rank 0 (VARSENS_RENDEZVOUS_FD), so that no port is probed, closed and
    raced for.  Returns the Popen objects (stdout + stderr piped together, text)."""
    import socket
    import subprocess

    def start(world, argv, extra_env=None):
        listener = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        listener.bind(("127.0.0.1", 0))
        listener.listen(max(world, 8))
        port = listener.getsockname()[1]
        procs = []
        for rank in range(world):
            env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), VARSENS_RENDEZVOUS="tcp")
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            for name in ("TORCHELASTIC_RUN_ID", "VARSENS_RENDEZVOUS_FD", "VARSENS_RENDEZVOUS_PORT"):
                env.pop(name, None)
            env.update(extra_env or {})
            if rank == 0:
                env["VARSENS_RENDEZVOUS_FD"] = str(listener.fileno())
            procs.append(subprocess.Popen([sys.executable] + list(argv), env=env, stdout=subprocess.PIPE,
                                          stderr=subprocess.STDOUT, text=True,
                                          pass_fds=(listener.fileno(),) if rank == 0 else ()))
        listener.close()
        return procs
    return start
