"""A short run of tools/fuzz_parity.py (randomised differential test of forward, row-id extraction,
Transpose, remap and backward against the CPU oracle).  The open-ended version is the tool itself."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fuzz_parity_smoke(oracle):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_parity
    n = fuzz_parity.run(seconds=20.0, seed=2024, max_cases=40)
    assert n >= 5
