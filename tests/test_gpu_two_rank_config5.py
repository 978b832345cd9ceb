"""BASELINE config 5 (fp16 fwd + bwd, batch sharded across ranks, table replicated, gradient
combined across ranks) with TWO ranks running the HIP kernels -- see tests/two_rank_worker.py.
The ranks are fresh child processes sharing GPU 0; the exchange goes over gloo on host copies."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_config5_ranks_share_one_gpu(oracle, world):
    port = str(_free_port())
    worker = os.path.join(ROOT, "tests", "two_rank_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), port], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True, cwd=ROOT) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=600)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o[-4000:])
        assert "rank %d ok" % r in o
