"""Runs the C++ known-answer program built against the HEADER-ONLY API (tests/cpp/header_api_kat.hip):
every template instantiation the reference ships, through the reference-named include paths."""
import subprocess

import pytest

pytestmark = pytest.mark.gpu


def test_header_only_api_known_answers():
    from cuembed_amd import build
    exe = build.build_header_api_test()
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    assert "all known-answer checks passed" in r.stdout
