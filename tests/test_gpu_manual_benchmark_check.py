"""benchmarks/manual_benchmark --check_result (reference: manual_benchmark.cu:85-90, :278-285, :373-386, :495-507):
the C++ benchmark on the header-only API validates forward, transpose (+ remap) and backward against the CPU checker
with exact equality -- the checker being oracle/libcuembed_oracle.so, loaded with dlopen only when the flag is given."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "benchmarks", "manual_benchmark")


@pytest.fixture(scope="module")
def exe(oracle):          # (the fixture builds oracle/libcuembed_oracle.so)
    from cuembed_amd import build
    return build.build_manual_benchmark()


CASES = {
    "fp32_fixed_compressed": ["--num_categories", "20000", "--embed_width", "64", "--batch_size", "2048", "--hotness", "16",
                              "--alpha", "1.15"],
    "fp16_csr_weighted": ["--num_categories", "5000", "--embed_width", "128", "--batch_size", "1000", "--hotness", "12",
                          "--alpha", "1.05", "--half_embedding_type=true", "--csr_input=true", "--weighted_sum=true"],
    "fp32_csr_weighted_bag_order": ["--num_categories", "8000", "--embed_width", "128", "--batch_size", "3001", "--hotness", "40",
                                    "--alpha", "1.15", "--csr_input=true", "--weighted_sum=true", "--bag_order=true"],
    "fp32_i64_dense_grad": ["--num_categories", "3000", "--embed_width", "32", "--batch_size", "4099", "--hotness", "7",
                            "--use_int64_indices=true", "--compressed_grad=false", "--skip_grad_init=false"],
    "fp16_math_fused_bounded": ["--num_categories", "100000", "--embed_width", "256", "--batch_size", "4096", "--hotness", "8",
                                "--alpha", "1.15", "--half_embedding_type=true", "--fp16_math=true", "--fused_row_ids=true",
                                "--bounded_sort=true"],
    "large_enough_for_slices": ["--num_categories", "200000", "--embed_width", "128", "--batch_size", "40000", "--hotness", "32",
                                "--alpha", "1.15"],
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_check_result_passes(exe, name):
    r = subprocess.run([exe] + CASES[name] + ["--check_result=true", "--iterations", "2"], capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "MISMATCH" not in r.stderr
    ok = [ln for ln in r.stderr.splitlines() if "matches the CPU result" in ln]
    kinds = " ".join(ok)
    assert "forward" in kinds and "transpose indices" in kinds and "transpose sample ids" in kinds and "backward" in kinds
    if "--compressed_grad=false" not in CASES[name]:
        assert "remapped indices" in kinds and "inverse mapping" in kinds


def test_check_result_needs_the_checker_and_is_off_by_default(exe):
    env = dict(os.environ, CUEMBED_ORACLE_LIB="/nonexistent/libcuembed_oracle.so")
    args = [exe] + CASES["fp32_fixed_compressed"] + ["--iterations", "1"]
    r = subprocess.run(args + ["--check_result=true"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 3 and "cannot load the CPU checker" in r.stderr
    r = subprocess.run(args, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)     # flag not given: never loaded
    assert r.returncode == 0 and "check_result" not in r.stderr
