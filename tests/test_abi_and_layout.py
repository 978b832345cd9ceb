"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol that
include/cuembed_amd.h declares, the header compiles as plain C, the host-side argument contract
raises before any launch, and the product never reaches into oracle/."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "cuembed_amd.h")


@pytest.fixture(scope="module")
def lib_path():
    from cuembed_amd import build
    return build.build()


def declared_symbols():
    pre = subprocess.run(["gcc", "-E", "-P", HEADER], check=True, stdout=subprocess.PIPE, text=True).stdout
    names = set(re.findall(r"\b(cuembed_[a-z0-9_]+)\s*\(", pre))
    names.discard("cuembed_stream_t")
    return sorted(names)


def test_header_is_plain_c():
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", HEADER], check=True)


def test_every_declared_symbol_is_exported(lib_path):
    names = declared_symbols()
    assert len(names) >= 52
    for must in ["cuembed_embedding_forward_f16_i32_o32", "cuembed_embedding_backward_f32_i64",
                 "cuembed_transpose_i64_f32", "cuembed_compute_compressed_grad_indices_i32",
                 "cuembed_extract_row_ids_from_csr_i64_o64", "cuembed_embedding_forward"]:
        assert must in names
    L = ctypes.CDLL(lib_path)
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_cmake_project_builds_the_same_translation_units():
    """CMakeLists.txt (the C++ user's way in) and cuembed_amd/build.py (the Python package's) must not drift apart:
    the shared library is made of the same units, with the same float-atomics flag."""
    from cuembed_amd import build
    with open(os.path.join(ROOT, "CMakeLists.txt")) as f:
        cm = f.read()
    listed = set(re.findall(r"\$\{CUEMBED_AMD_CSRC\}/(c_api_[a-z_]+\.hip)", cm))
    assert listed == set(build.UNITS)
    assert "-munsafe-fp-atomics" in cm and "-munsafe-fp-atomics" in build.HIPCC_FLAGS
    assert "gfx950" in cm and build.ARCH == "gfx950"
    for src in ("tests/cpp/header_api_kat.hip", "benchmarks/manual_benchmark.hip"):
        assert src in cm and os.path.exists(os.path.join(ROOT, src))


def test_launch_shapes_of_baseline_configs():
    import torch
    import cuembed_amd as ce
    # C2: 512-byte fp16 rows -> 32 lanes x 16 B, two samples per wavefront, 8 per workgroup
    s = ce.forward_launch_shape(torch.float16, torch.int32, 256, 65536, 64)
    assert s == dict(elems_per_lane=8, lanes_per_row=32, samples_per_block=8, grid=8192,
                     lds_bytes=8 * 64 * 4, staged=True, wide_load=False)
    # C1: 128-byte fp32 rows -> 8 lanes, 32 samples per workgroup
    s = ce.forward_launch_shape(torch.float32, torch.int32, 32, 1024, 8)
    assert (s["elems_per_lane"], s["lanes_per_row"], s["samples_per_block"], s["grid"]) == (4, 8, 32, 32)
    # C3: CSR never stages
    s = ce.forward_launch_shape(torch.float32, torch.int32, 128, 65536, 0, is_csr=True, is_weighted=True)
    assert not s["staged"] and s["lds_bytes"] == 0 and s["lanes_per_row"] == 32
    # odd widths fall back to 8- and 4-byte lanes
    assert ce.forward_launch_shape(torch.float32, torch.int32, 514, 8, 4)["elems_per_lane"] == 2
    assert ce.forward_launch_shape(torch.float16, torch.int32, 514, 8, 4)["elems_per_lane"] == 2
    assert ce.forward_launch_shape(torch.float32, torch.int32, 3, 8, 4)["elems_per_lane"] == 1
    # small batches: the wide-load kernel (one sample per workgroup, 8 x slices rows parked in LDS = 32 KB with 16-byte
    # lanes) -- CSR up to 1,024 workgroups, fixed hotness from 32 lookups per sample while the
    # sequential mapping has at most min(256, 2 x hotness) wavefronts; never for concat, rows > 1 KiB or odd lane counts
    wide = ce.forward_launch_shape(torch.float32, torch.int32, 32, 1024, 64)
    assert wide["wide_load"] and (wide["samples_per_block"], wide["grid"], wide["lds_bytes"]) == (1, 1024, 32768)
    assert ce.forward_launch_shape(torch.float16, torch.int32, 256, 256, 64, is_weighted=True)["lds_bytes"] == 32768 + 64 * 2
    for args, kw, want in [((torch.float32, torch.int32, 32, 1024, 16), {}, False),        # too few lookups per sample
                           ((torch.float32, torch.int32, 32, 1024, 32), {}, False),        # 128 wavefronts > 2 x 32
                           ((torch.float32, torch.int32, 32, 512, 32), {}, True),
                           ((torch.float32, torch.int32, 32, 2048, 256), {}, True),        # 256 wavefronts
                           ((torch.float32, torch.int32, 32, 4096, 256), {}, False),
                           ((torch.float16, torch.int32, 256, 1024, 64), {}, False),       # 512 wavefronts
                           ((torch.float16, torch.int32, 256, 256, 64), {}, True),
                           ((torch.float16, torch.int32, 256, 1024, 0), dict(is_csr=True), True),
                           ((torch.float16, torch.int32, 256, 1025, 0), dict(is_csr=True), False),   # 512-byte rows: 1 sample per workgroup only
                           ((torch.float32, torch.int32, 32, 8192, 0), dict(is_csr=True), True),     # 128-byte rows: up to 8
                           ((torch.float32, torch.int32, 32, 8193, 0), dict(is_csr=True), False),
                           ((torch.float32, torch.int32, 64, 2048, 0), dict(is_csr=True), True),     # 256-byte rows: up to 2
                           ((torch.float32, torch.int32, 64, 2049, 0), dict(is_csr=True), False),
                           ((torch.float32, torch.int32, 512, 16, 64), {}, False),         # 2 KiB rows
                           ((torch.float32, torch.int32, 100, 16, 64), {}, False),         # 25 lanes
                           ((torch.float32, torch.int32, 32, 16, 64), dict(mode="concat"), False)]:
        assert ce.forward_launch_shape(*args, **kw)["wide_load"] == want, (args, kw)
    # CSR batches of narrow rows beyond 1,024 samples: several samples per workgroup while samples x row bytes <= 1024
    # (rows of up to 128 bytes) or 512 (wider ones)
    many = ce.forward_launch_shape(torch.float32, torch.int32, 32, 4000, 0, is_csr=True)
    assert many["wide_load"] and (many["samples_per_block"], many["grid"], many["lds_bytes"]) == (4, 1000, 32768)
    assert ce.forward_launch_shape(torch.float32, torch.int32, 8, 16384, 0, is_csr=True)["samples_per_block"] == 16
    assert ce.forward_launch_shape(torch.float32, torch.int32, 8, 32768, 0, is_csr=True)["samples_per_block"] == 32


def test_backward_column_slices_follow_the_lookup_count():
    """ChooseColumnSlices: no slices below 2^17 lookups, slices of >= 256 bytes from there, of >= 128 bytes from 2^20 on
    (measured: profiles/r05_backward_mid_size_slices.txt); never more than one per XCD, never on a single-XCD partition."""
    import torch
    import cuembed_amd as ce

    def slices(dtype, width, nnz, **kw):
        return ce.backward_launch_shape(dtype, torch.int32, width, nnz, compute_units=256, xcds=8, **kw)["column_slices"]
    for nnz, want_512, want_1k, want_256 in [((1 << 17) - 1, 1, 1, 1), (1 << 17, 2, 4, 1), ((1 << 20) - 1, 2, 4, 1),
                                             (1 << 20, 4, 8, 2), (1 << 24, 4, 8, 2)]:
        assert slices(torch.float16, 256, nnz) == want_512, nnz          # 512-byte rows
        assert slices(torch.float32, 128, nnz) == want_512, nnz
        assert slices(torch.float32, 256, nnz) == want_1k, nnz           # 1 KiB rows
        assert slices(torch.float32, 64, nnz) == want_256, nnz           # 256-byte rows
        assert slices(torch.float32, 32, nnz) == 1, nnz                  # 128-byte rows are one L2 line
    assert ce.backward_launch_shape(torch.float16, torch.int32, 256, 1 << 22, compute_units=32, xcds=1)["column_slices"] == 1
    assert ce.backward_launch_shape(torch.float32, torch.int32, 1024, 1 << 22, compute_units=128, xcds=4)["column_slices"] == 4


def test_backward_planner_follows_the_device_it_is_told_about():
    """The launch heuristics take the chip's shape from the device (cuembed::detail::DeviceShape; the reference asks
    the runtime per call, embedding_lookup.cuh:348-395) instead of assuming a full MI355X.  Host arithmetic only:
    a full chip (256 CUs, 8 XCDs), a CPX partition (32 CUs, 1 XCD) and a DPX half (128 CUs, 4 XCDs)."""
    import torch
    import cuembed_amd as ce
    nnz = 65536 * 64
    full = ce.backward_launch_shape(torch.float16, torch.int32, 256, nnz, compute_units=256, xcds=8)
    assert (full["column_slices"], full["lanes"], full["segment_len"], full["xcds"]) == (4, 8, 64, 8)
    assert full["grid"] % 8 == 0 and full["grid"] == full["nz_blocks"] * 4       # one workgroup per (nz block, slice)
    cpx = ce.backward_launch_shape(torch.float16, torch.int32, 256, nnz, compute_units=32, xcds=1)
    assert (cpx["column_slices"], cpx["lanes"], cpx["xcds"]) == (1, 32, 1)       # one L2: XCD slicing is meaningless
    assert cpx["grid"] == cpx["nz_blocks"] and cpx["segment_len"] == 128         # 1/8 of the lanes to fill: long segments
    dpx = ce.backward_launch_shape(torch.float16, torch.int32, 256, nnz, compute_units=128, xcds=4)
    assert dpx["column_slices"] == 4 and dpx["grid"] % 4 == 0 and dpx["xcds"] == 4
    wide = ce.backward_launch_shape(torch.float32, torch.int32, 256, nnz, compute_units=256, xcds=8)
    assert wide["column_slices"] == 8                                             # 1 KiB rows: one slice per XCD
    assert ce.backward_launch_shape(torch.float32, torch.int32, 256, nnz, compute_units=128, xcds=4)["column_slices"] == 4
    # few lookups: segments shrink until 40 % of the DEVICE's lanes are busy -- sooner on a small device
    few = 32768
    assert ce.backward_launch_shape(torch.float16, torch.int32, 256, few, compute_units=256, xcds=8)["segment_len"] < \
        ce.backward_launch_shape(torch.float16, torch.int32, 256, few, compute_units=32, xcds=1)["segment_len"]
    # sample blocks: what one L2 fronts while a block is scattered must fit it
    assert ce.recommended_sample_blocks(torch.float16, 256, 65536, nnz, 256, 8, 4 << 20) == 2
    assert ce.recommended_sample_blocks(torch.float16, 256, 65536, nnz, 32, 1, 4 << 20) == 8   # unsliced: 33.5 MB / 4 MiB
    assert ce.recommended_sample_blocks(torch.float16, 256, 65536, nnz, 256, 8, 16 << 20) == 1


def test_host_layer_refuses_cpu_tensors_and_bad_contracts():
    import torch
    import cuembed_amd as ce
    t = torch.zeros(4, 4)
    i = torch.zeros(4, dtype=torch.int64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ce.embedding_forward(t, i, num_hots=2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ce.transpose(i, i)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ce.embedding_backward(t, 4, i, i)
    with pytest.raises(ValueError):
        ce.embedding_forward(t, i, num_hots=2, mode="max")


def test_abort_on_contract_violation_in_c_abi(lib_path):
    """The C ABI keeps the reference's behaviour (CUEMBED_ASSERT, embedding_lookup.cuh:151-158):
    message on stderr + abort.  Checked in a child process; no GPU work is launched because the
    argument check precedes everything."""
    code = (
        "import ctypes,sys\n"
        "L=ctypes.CDLL(%r)\n"
        "L.cuembed_embedding_forward(None,0,4,None,0,None,0,None,2,0,0,0,None,None)\n" % lib_path)
    r = subprocess.run(["python3", "-c", code], stderr=subprocess.PIPE, text=True)
    assert r.returncode != 0
    assert "Check failed" in r.stderr and "num_hots" in r.stderr


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "cuembed_amd")
    offenders = []
    for dirpath, _, files in os.walk(pkg):
        if os.sep + "build" in dirpath or os.sep + "lib" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".cpp", ".h")):
                with open(os.path.join(dirpath, f), errors="ignore") as fh:
                    txt = fh.read()
                if re.search(r"\boracle\b", txt):
                    offenders.append(os.path.join(dirpath, f))
    assert not offenders, offenders
