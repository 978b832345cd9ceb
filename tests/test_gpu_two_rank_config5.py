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


@pytest.mark.parametrize("exchange", ["sparse", "sparse_fixed"])
def test_train_step_benchmark_under_torchrun_ranks_share_the_gpu(exchange):
    """benchmarks/train_step_benchmark.py (BASELINE configs[4]: fwd + bwd, batch sharded, sparse gradient exchange)
    launched exactly as on a multi-GPU node, with two ranks that have to share this box's GPU: the exchange then
    runs over gloo on host copies.  Sample blocks (an uncoalesced gradient) through the owner-partitioned exchange, with
    exact sizes and with the fixed capacities of SparseGradExchange (calibrated by a warm-up step; no overflow)."""
    import json
    port = str(_free_port())
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", port,
                        os.path.join(ROOT, "benchmarks", "train_step_benchmark.py"), "--exchange", exchange,
                        "--sparse_algorithm", "owner", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["exchange"] == exchange and line["sample_blocks"] == 2
    assert line["ms_per_step"] > 0 and line["breakdown_ms"]["exchange"] > 0
    assert line["fixed_capacity_overflowed"] is (False if exchange == "sparse_fixed" else None)
