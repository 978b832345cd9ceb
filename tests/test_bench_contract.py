"""bench.py's JSON contract, checked without a GPU: the roofline block is built from measured traffic
(profiles/traffic_c2.json, written by tools/traffic_from_pmc.py), every fraction of the HBM peak is
<= 1 by construction, and application (algorithmic) GB/s is never presented as an HBM rate."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_traffic_file_schema_and_calibration():
    with open(os.path.join(ROOT, "profiles", "traffic_c2.json")) as f:
        t = json.load(f)
    assert "tools/traffic_from_pmc.py" in t["generated_by"]
    assert t["fetch_correction"] == 2.0 and t["write_correction"] == 1.0
    k = t["kernels"]
    for name in ("forward_c2", "forward_c2_alpha0", "calibration_unique_rows", "backward_c4", "transpose_c4"):
        assert k[name]["hbm_bytes_per_launch"] > 0, name
    assert "backward_c4_run_aware" not in k          # the run-aware backward left the product in round 3
    with open(os.path.join(ROOT, "profiles", "traffic_c3.json")) as f:
        c3 = json.load(f)
    assert c3["kernels"]["forward_c3"]["hbm_bytes_per_launch"] > 0 and c3["kernel_sources_sha16"] == t["kernel_sources_sha16"]
    # the FETCH_SIZE x 2 correction is re-checked on a launch whose read volume is known exactly
    assert abs(k["calibration_unique_rows"]["measured_over_expected"] - 1.0) < 0.02
    # power-law: fabric traffic far below the algorithmic bytes; uniform: about equal to them
    alg = 2 * 65536 * 65 * 256
    assert k["forward_c2"]["hbm_bytes_per_launch"] < 0.5 * alg
    assert 0.9 * alg < k["forward_c2_alpha0"]["hbm_bytes_per_launch"] < 1.1 * alg


def test_roofline_entries_are_fractions():
    import bench
    traffic = bench.load_traffic("c2")
    assert traffic["forward_c2"]["hbm_bytes_per_launch"] > 0 and "_source" in traffic
    cfg = dict(bench.WORKLOADS["c2"])
    alg = bench.algorithmic_bytes(cfg, None)
    assert alg == 2181038080
    result = {"config": {"name": "c2"},
              "roofline": {"algorithmic_bytes_per_launch": alg},
              "extras": {"alpha0_uniform_back_to_back": {"ms": 0.36, "GBps": alg / 0.36e-3 / 1e9},
                         "backward_compressed_ms": 0.268, "backward_unique_rows": 572029,
                         "transpose_and_remap_ms": 0.129,
                         # the C3 forward as the driver's line carries it (algorithmic bytes ABOVE the measured traffic)
                         "c3_forward": {"ms": 0.1716, "algorithmic_bytes": 2184951808, "nnz": 4201805,
                                        "compulsory_bytes": 386000000},
                         "transpose_and_remap_small_shapes": {
                             "1024x16": {"pairs": 16384, "reference_sequence_ms": 0.038, "one_call_ms": 0.034}}}}
    bench.finish_roofline(result, traffic, cfg)
    rl = result["roofline"]
    comp = rl["hbm_bound_companion"]
    assert comp["bound"] == "hbm" and 0.5 < comp["frac"] <= 1.0
    assert abs(comp["achieved"] - comp["traffic"] / 0.36e-3 / 1e9) < 1.0
    kinds = [o["kernel"] for o in rl["other_kernels"]]
    assert any("EmbeddingBackward" in x for x in kinds) and any("Transpose" in x for x in kinds)
    assert any("C3" in x for x in kinds)
    for o in rl["other_kernels"]:
        assert 0.0 < o["traffic_frac"] <= 1.0 and o["peak"] == bench.HBM_PEAK_GBPS
        assert 0.0 < o["frac"] <= 1.0 and o["frac"] <= o["traffic_frac"] + 1e-9   # EVERY `frac` is a fraction
        if o["traffic"] is None:            # the small-shape index work: no counter pass, launch-latency bound
            assert "16,384 pairs" in o["kernel"] and o["one_call_ms"] > 0 and o["frac"] == o["algorithmic_frac"]
            continue
        if "C3" in o["kernel"]:
            # a gather with re-use: the formula's bytes exceed what the fabric carried, so `frac` follows the headline's
            # convention (measured traffic) and the algorithmic figure stays beside it
            assert o["frac_convention"] == "traffic" and o["algorithmic_frac"] > 1.0 and o["frac"] == o["traffic_frac"]
            assert abs(o["achieved"] - o["traffic"] / (o["ms"] * 1e-3) / 1e9) < 1.0
            continue
        assert o["traffic"] >= o["algorithmic_bytes_per_launch"] * 0.9      # traffic is measured, not assumed
        # `frac` is the fraction of the kernel's OWN roofline (reference formula bytes); the traffic-based one is
        # traffic_frac (a reader who takes `frac` at face value must not be misled)
        assert o["frac_convention"] == "algorithmic"
        assert abs(o["frac"] - o["algorithmic_bytes_per_launch"] / (o["ms"] * 1e-3) / 1e9 / 8000.0) < 1e-3
        assert o["frac"] == o["algorithmic_frac"]
        assert abs(o["achieved"] - o["algorithmic_bytes_per_launch"] / (o["ms"] * 1e-3) / 1e9) < 1.0
    assert abs(comp["algorithmic_frac"] - alg / 0.36e-3 / 1e9 / 8000.0) < 1e-3
    # the headline fraction: measured bytes over a plausible kernel time stays below 1
    fwd = traffic["forward_c2"]["hbm_bytes_per_launch"]
    assert fwd / 0.136e-3 / 1e9 / bench.HBM_PEAK_GBPS < 1.0
    # while the application figure exceeds the peak -- which is why it is not divided by it anywhere
    assert alg / 0.136e-3 / 1e9 > bench.HBM_PEAK_GBPS


def test_traffic_files_follow_from_the_committed_counter_rows():
    """profiles/rNN_traffic_rows.txt (the newest) holds the per-pass averages (KiB per dispatch) of the last refresh, written in
    the same run as the JSON files bench.py reads and committed next to the rocprofv3 pass summaries: every
    hbm_bytes_per_launch must be (FETCH_SIZE x 2 + WRITE_SIZE) x 1024 of its row, within 0.5 %."""
    import glob
    rows = {}
    latest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic_rows.txt")))[-1]     # the last refresh's
    with open(latest) as f:
        for ln in f:
            parts = ln.split()
            if parts and parts[0] == "traffic_row":
                kv = dict(p.split("=", 1) for p in parts[3:])
                rows[(parts[1], parts[2])] = (float(kv["FETCH_SIZE_KiB"]), float(kv["WRITE_SIZE_KiB"]), int(kv["dispatches"]))
    seen = 0
    for name in ("traffic_c2.json", "traffic_c3.json"):
        with open(os.path.join(ROOT, "profiles", name)) as f:
            t = json.load(f)
        for key, e in t["kernels"].items():
            fetch, write, n = rows[(name, key)]
            want = (fetch * t["fetch_correction"] + write * t["write_correction"]) * 1024
            assert abs(e["hbm_bytes_per_launch"] - want) <= 0.005 * want, (name, key)
            assert e["launches_averaged"] == n
            seen += 1
    assert seen >= 6      # forward c2 / alpha 0 / calibration, backward c4, transpose c4, forward c3
    # the same numbers are in the committed pass summaries (kernel-level averages over ALL dispatches of a pass)
    summary = open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_passes_forward_pipeline_c3.txt")))[-1]).read()
    for tag in ("#### pmc_fwd_fetch", "#### pmc_pipe_write", "#### pmc_c3_fetch", "SegmentedScatterAddKernel", "FETCH_SIZE"):
        assert tag in summary


def test_traffic_file_is_stamped_with_the_kernel_sources():
    import bench
    sha = bench.kernel_sources_sha16()
    assert len(sha) == 16 and int(sha, 16) >= 0
    t = bench.load_traffic("c2")
    assert "_sha16" in t          # None for files written before the stamp existed; bench.py then reports traffic_stale


def test_gpus_n_starts_its_own_ranks_and_fails_cleanly_without_a_gpu():
    """`python bench.py --gpus 2` typed as is: the parent starts two rank processes before touching any
    GPU API; here (no GPU) both ranks refuse to run, and the parent must relay that as a non-zero exit,
    leave no rank behind and remove the shared index stream."""
    import glob
    import subprocess
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("CPU-side launcher check")
    before = set(glob.glob("/dev/shm/cuembed_bench_idx_*"))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--workload", "c1"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0
    assert "rank" in r.stderr and "exited with code" in r.stderr
    assert "needs a GPU" in r.stderr                      # the ranks' own message reaches the caller
    assert r.stdout.strip() == ""                         # no JSON line from a failed job
    assert set(glob.glob("/dev/shm/cuembed_bench_idx_*")) == before
