"""Host sanitizer run (the reference's policy is "tests should pass with Compute
Sanitizer", CONTRIBUTING.md:11; GPU sanitizers are not available on this pool, so this covers the
host code): the CPU oracle and the harness's synthetic-input generator are rebuilt with
AddressSanitizer + UndefinedBehaviorSanitizer and the test files that exercise them are run against
those builds (oracle/Makefile target `asan-check`, oracle/run_asan.sh)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(os.environ.get("CUEMBED_ORACLE_LIB") is not None, reason="already inside the sanitizer run")
@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_oracle_and_harness_generator_are_clean_under_asan_ubsan():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan-check"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:]
    assert " passed" in r.stdout and "ERROR: AddressSanitizer" not in r.stdout and "runtime error" not in r.stdout


def test_sanitized_libraries_are_the_ones_loaded():
    """The override variables really switch the libraries (otherwise the run above proves nothing)."""
    lib = os.path.join(ROOT, "oracle", "_asan", "libcuembed_oracle_asan.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    code = ("import sys; sys.path.insert(0, %r); import oracle.oracle as O, cuembed_amd.harness as H; "
            "print(O._LIB_PATH); print(H._PATH)" % ROOT)
    env = dict(os.environ, CUEMBED_ORACLE_LIB=lib,
               CUEMBED_HARNESS_LIB=os.path.join(ROOT, "oracle", "_asan", "libcuembed_harness_asan.so"))
    out = subprocess.run(["python3", "-c", code], env=env, stdout=subprocess.PIPE, text=True, check=True).stdout.split()
    assert out[0].endswith("_asan/libcuembed_oracle_asan.so") and out[1].endswith("_asan/libcuembed_harness_asan.so")
