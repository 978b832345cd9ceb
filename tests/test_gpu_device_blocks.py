"""Unit tests of the gather / scatter building blocks ON THE DEVICE (tests/cpp/device_blocks_unit.hip): the counterpart
of the reference's tests/test_embedding_ops.cu:121-374 (Addresser, Combiner, IndexLoader, GradAddresser, GradCombiner unit
tests) for the blocks this design is made of -- WidenIndex / RowElems / RowPtr, ColumnSlice, Pack + Arith + RowPool (Add,
Gather around every unroll / pipelining boundary, both load kinds), FinishPooledRow (mean, empty bag), AccumulateRow --
each instantiated in a small kernel and compared bit for bit with the same single-rounding operations on the host."""
import subprocess

import pytest

pytestmark = pytest.mark.gpu


def test_device_building_blocks_against_host_recomputation():
    from cuembed_amd import build
    exe = build.build_device_blocks_test()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    assert "all device building-block checks passed" in r.stdout
