"""The one-launch passes over the high word of 64-bit keys (RadixHighPassesKernel / RadixHighPassesChainedKernel) order
their phases with a ticket queue instead of a grid barrier: they must finish -- with the right order -- whatever share
of their workgroups is resident.  (Round 5's grid-barrier form assumed that CU / 4 workgroups are always resident
together; nothing guarantees that under other streams, RCCL's kernels or other processes, and the failure mode was a
hang.  Reference: cub::DeviceRadixSort, index_transforms.cuh:108-136, never spins on residency.)

Each case runs in a child process under a timeout: a hang fails the test (the child, and only the child, is killed by
its pid) instead of taking the suite down."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import sys
sys.path.insert(0, {root!r})
import torch
import cuembed_amd as ce

def check(n, seed, contend):
    g = torch.Generator(device="cuda").manual_seed(seed)
    # keys that use all 64 bits, negative ones included; few distinct values per high word so that runs are long
    hi = torch.randint(-2**31, 2**31, (n,), device="cuda", generator=g, dtype=torch.int64)
    lo = torch.randint(0, 50, (n,), device="cuda", generator=g, dtype=torch.int64)
    cols = (hi << 32) | lo
    if seed % 3 == 0:
        cols = cols >> 20                      # only some of the high digits vary
    rows = torch.arange(n, device="cuda", dtype=torch.int64)
    want_keys, perm = torch.sort(cols, stable=True)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(3)] if contend else []
    a = torch.randn(6144, 6144, device="cuda") if contend else None
    outs = []
    for s in streams:                          # CU-filling work on other streams while the sort runs
        with torch.cuda.stream(s):
            for _ in range(6):
                outs.append(a @ a)
    results = [ce.transpose(rows, cols) for _ in range(3)]          # (reference signature: all 64 bits of the key)
    if contend:                                # ... and sorts racing each other on their own streams
        for s in streams:
            with torch.cuda.stream(s):
                results.append(ce.transpose(rows, cols))
    torch.cuda.synchronize()
    for t_cols, t_rows, _ in results:
        assert torch.equal(t_cols, want_keys), (n, seed, "keys")
        assert torch.equal(t_rows, perm), (n, seed, "stable order")

for seed, n in enumerate((5000, 70000, 200000, 229376, 229377, 1 << 20, (1 << 22) + 77)):
    check(n, seed, {contend})
torch.cuda.synchronize()
assert ce._lib.lib().cuembed_peek_last_error() == 0
print("sorted ok")
"""


def _run(env_extra, contend, timeout):
    env = dict(os.environ, **env_extra)
    child = subprocess.Popen([sys.executable, "-c", _CHILD.format(root=ROOT, contend=contend)], env=env,
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        out, err = child.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        child.kill()                    # this child, by its pid; a fresh process per case, never a re-exec
        child.communicate()
        pytest.fail("the high-word sort did not finish within %d s (workgroups waiting for each other?)" % timeout)
    assert child.returncode == 0 and "sorted ok" in out, out[-2000:] + err[-4000:]


@pytest.mark.parametrize("workgroups", ["1", "3", "20000"])
def test_high_word_passes_finish_with_any_number_of_resident_workgroups(workgroups):
    """A grid of ONE workgroup (everything in ticket order by a single workgroup), of three, and of 20,000 workgroups of
    1,024 threads -- some forty times what 256 compute units hold at once, so most of the grid is NOT resident while the
    first tickets are worked on (a grid barrier over such a grid never completes)."""
    _run({"CUEMBED_SORT_HIGH_WORD_WORKGROUPS": workgroups}, False, 600)


def test_high_word_passes_under_contention_from_other_streams():
    """Default grid; GEMMs that fill the chip on three other streams and more 64-bit sorts racing on those streams."""
    _run({}, True, 600)
