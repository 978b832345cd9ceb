"""`python3 bench.py --gpus 2 --steps 20 --warmup 5`, typed exactly like that with no launcher around it:
bench.py starts its own two rank processes (before touching the GPU), they share the box's GPU(s), rank 0's
JSON line comes back through the parent.  This is the command a scaling run issues with N = 2, 4, 8."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True,
                       timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]   # stdout is the ONE JSON line, nothing else
    return json.loads(lines[0])


def test_gpus_2_as_typed():
    import torch
    line = _run(["--gpus", "2", "--steps", "20", "--warmup", "5"])
    assert line["n_gpus"] == 2 and line["steps"] == 20 and line["warmup"] == 5
    assert line["scaling"] == "weak" and line["config"]["global_batch"] == 2 * 65536
    assert line["config"]["ranks_share_gpus"] == (torch.cuda.device_count() < 2)
    assert line["config"]["rendezvous_backend"] in ("gloo", "nccl")
    assert line["value"] > 0 and line["roofline"]["achieved"] > 0
    # two ranks time-sharing one GPU deliver about one GPU's worth; two GPUs about twice that
    assert line["ms_per_step"] > 0.1
    # the SAME clock at every N: max over ranks of the rank's wall time between its synchronize() brackets; the
    # HIP-event figure and the wall time that includes the closing barrier sit next to it
    assert "same clock at every N" in line["timing"] and line["ms_per_step"] == line["wall_ms_per_step"]
    assert line["ms_per_step_events"] <= line["ms_per_step"] * 1.02 and line["value_events"] > 0
    assert line["wall_ms_per_step_incl_closing_barrier"] >= line["ms_per_step"] and line["value_wall"] == line["value"]
    assert line["config"]["index_batches_cycled"] == 4            # the same at every N
    # BASELINE config 5 ran on ALL ranks and the gradient really went through the exchange
    c5 = line["extras"]["c5_train_step"]
    assert c5["n_gpus"] == 2 and c5["backend"] in ("gloo", "nccl")
    assert c5["exchange_ms"] > 0 and c5["sparse"]["exchange_ms"] > 0 and c5["dense"]["exchange_ms"] > 0
    assert c5["sparse"]["algorithm"] == "owner" and c5["sparse"]["gradient_bytes_per_rank"] > 100e6
    assert c5["dense"]["gradient_bytes_per_rank"] == 10_000_000 * 256 * 2
    assert c5["sparse"]["step_ms"] > c5["compute_by_order"]["blocked_uncoalesced"]["compute_ms"]


def test_gpus_1_line_has_the_contract_fields():
    line = _run(["--gpus", "1", "--steps", "20", "--warmup", "5", "--no-c3"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "value_reference_protocol",
              "pct_of_hbm_peak", "preroll_ms", "steady_state", "timing"):
        assert k in line, k
    assert line["preroll_ms"] == 0 and line["preroll_launches"] == 0      # `value` is the caller's protocol, nothing else
    assert "same clock at every N" in line["timing"] and line["config"]["index_batches_cycled"] == 4
    assert line["ms_per_step"] == line["wall_ms_per_step"]
    # `metric` / `value` are BASELINE.json's: achieved HBM GB/s (measured fabric bytes / time), a fraction of the peak;
    # the reference benchmark's application bandwidth (which exceeds the peak at alpha = 1.15) sits beside it
    assert line["metric"].startswith("achieved HBM GB/s (% of peak), EmbeddingForward w=256 hot=64")
    assert 0 < line["value"] <= 8000.0 < line["application_GBps"] and "measured L2->fabric bytes" in line["value_basis"]
    assert abs(line["pct_of_hbm_peak_whole_job"] - 100 * line["value"] / 8000.0) < 0.02
    assert line["steady_state"]["preroll_ms"] == 100 and 0 < line["steady_state"]["ms_per_step"] < line["ms_per_step"] * 1.15
    rl = line["roofline"]
    assert rl["bound"] == "l2+fabric" and 0 < rl["frac"] <= 1.0 and rl["traffic"] > 0
    assert abs(line["pct_of_hbm_peak"] - 100 * rl["frac"]) < 0.02
    assert "protocol" in line["config"] and line["value_reference_protocol"]["value"] > 0
    for o in rl["other_kernels"]:
        assert 0 < o["frac"] <= 1.0 and o["frac"] <= o["traffic_frac"] + 1e-9
        assert o["frac"] == (o["algorithmic_frac"] if o["frac_convention"] == "algorithmic" else o["traffic_frac"])
    assert line["baseline_metric_value"] == rl["achieved"] and abs(line["baseline_metric_pct"] - 100 * rl["frac"]) < 0.02
    assert 0 < rl["compulsory_frac"] < rl["frac"]
    c3 = [o for o in rl["other_kernels"] if "C3" in o["kernel"]]
    assert not c3 or (c3[0]["compulsory_bytes_per_launch"] > 0 and c3[0]["hbm_bound_companion"]["row_loads_streaming"]["ms"] > 0)
    assert not c3 or (c3[0]["with_sample_order"]["bit_identical_to_default"] is True and c3[0]["with_sample_order"]["ms"] > 0)
    comp = rl["hbm_bound_companion"]
    assert comp["row_loads"] == "default" and comp["row_loads_streaming"]["ms"] > 0
    assert line["cpu_baseline"]["gpu_matches_oracle_bit_exact"] is True and line["cpu_baseline"]["cpu_model"]
    ex = line["extras"]
    assert ex["sample_blocks"] == 2 and ex["sample_blocks_gradient_equals_reference_order"] is True
    assert ex["backward_compressed_sample_blocks_ms"] < ex["backward_compressed_ms"]
    ts = ex["train_step_per_gpu"]
    assert 0 < ts["sample_blocks_2"]["ms"] < ts["sample_blocks_1"]["ms"] < 1.0
    # the reference's compressed gradient from the blocked order: same rows, same inverse mapping
    assert ex["blocked_coalesced_num_unique_equals_reference_order"] is True
    assert ex["blocked_coalesced_inverse_mapping_equals_reference_order"] is True
    assert ex["blocked_coalesced_max_rel_diff_vs_reference_order"] < 1e-2
    # (relations between measured times keep a margin: this is a functional test, the numbers are DESIGN.md's business)
    assert ex["backward_compressed_blocked_coalesced_ms"] < ex["backward_compressed_ms"] * 1.1
    c5 = ex["c5_train_step"]                                            # N = 1: no exchange, but the leg runs
    assert c5["n_gpus"] == 1 and c5["backend"] is None and c5["exchange_ms"] == 0
    assert c5["compute_by_order"]["blocked_uncoalesced"]["compute_ms"] < c5["compute_by_order"]["reference"]["compute_ms"]


def test_gpus_2_under_torch_distributed_run():
    """The driver's own multi-GPU launch: `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr
    127.0.0.1 --master-port P bench.py --gpus 2 --steps 20 --warmup 5` (local rank 0 walks the index stream for the node,
    every rank meets before the shared file goes)."""
    import glob
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    before = set(glob.glob("/dev/shm/cuembed_bench_idx_*"))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "20", "--warmup", "5", "--no-c3"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 2 * 65536 and line["value"] > 0
    assert set(glob.glob("/dev/shm/cuembed_bench_idx_*")) == before          # the node's stream file is gone
