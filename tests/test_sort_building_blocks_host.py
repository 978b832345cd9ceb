"""Host-side unit tests of the radix sort's integer building blocks (tests/cpp/sort_route_unit.hip): the counterpart of
the reference's FOR_HOST_TEST suite (tests/test_embedding_ops.cu:121-374) for the pieces of this design that are pure
logic -- which buffer every pass reads and writes for every subset of skipped passes, the next working pass, narrow /
wide decisions, the multiply-shift division of the implicit sample ids, tile maps, workspace plans.  The program is
compiled by hipcc but makes no HIP call: it runs here, without a GPU."""
import subprocess


def test_sort_building_blocks_on_the_host():
    from cuembed_amd import build
    exe = build.build_sort_unit_test()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "all host-side checks passed" in r.stdout
