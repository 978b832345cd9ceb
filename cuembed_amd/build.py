"""Builds libcuembed_amd.so (HIP kernels + C ABI) for gfx950 with hipcc.

    python -m cuembed_amd.build [--force]

The three translation units are compiled in parallel and linked into
cuembed_amd/lib/libcuembed_amd.so (in-tree, so the file travels with the repo
snapshot to the GPU box).  hipcc cross-compiles without a GPU present.
"""
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libcuembed_amd.so")
HARNESS_PATH = os.path.join(LIB_DIR, "libcuembed_harness.so")
OBJ_DIR = os.path.join(PKG, "build")
UNITS = ["c_api_forward.hip", "c_api_backward.hip", "c_api_transforms.hip", "c_api_exchange.hip"]
ARCH = "gfx950"

HIPCC_FLAGS = [
    "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC",
    # float atomics as hardware instructions (global_atomic_add_f32 / pk_add_f16)
    "-munsafe-fp-atomics",
    "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
] + os.environ.get("CUEMBED_HIPCC_EXTRA", "").split()   # experiments only (e.g. -DCUEMBED_TUNE_...): part of the build key


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; a ROCm toolchain is required to build cuembed_amd")
    return exe


def _digest_files(paths, extra=""):
    h = hashlib.sha256()
    h.update(extra.encode())
    for p in sorted(set(paths)):
        if os.path.exists(p):
            with open(p, "rb") as fh:
                h.update(os.path.relpath(p, ROOT).encode())
                h.update(fh.read())
        else:
            h.update(("missing:" + os.path.relpath(p, ROOT)).encode())
    return h.hexdigest()


def _deps_of(obj):
    """Dependencies recorded by the last compile of `obj` (hipcc -MD), repo files only."""
    dep = obj[:-2] + ".d"
    if not os.path.exists(dep):
        return None
    with open(dep) as f:
        txt = f.read().replace("\\\n", " ")
    files = txt.split(":", 1)[1].split() if ":" in txt else []
    out = []
    for x in files:  # keep repo files only, re-rooted (the tree may have moved since the compile)
        for marker in ("/cuembed_amd/csrc/", "/include/cuembed_amd.h"):
            k = x.find(marker)
            if k >= 0:
                out.append(os.path.join(ROOT, x[k + 1:]))
                break
    return out


def build(force=False, verbose=False):
    """Compile what changed (per translation unit, by content hash of the unit and the repo
    headers it includes).  Returns the path of the shared library."""
    os.makedirs(LIB_DIR, exist_ok=True)
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    flags_key = " ".join(HIPCC_FLAGS).replace(ROOT, "<root>")

    def unit_state(unit):
        obj = os.path.join(OBJ_DIR, unit.replace(".hip", ".o"))
        stamp = obj[:-2] + ".stamp"
        deps = _deps_of(obj)
        fresh = False
        if not force and deps is not None and os.path.exists(obj) and os.path.exists(stamp):
            with open(stamp) as f:
                fresh = f.read().strip() == _digest_files(deps, flags_key)
        return obj, stamp, fresh

    def compile_one(unit):
        obj, stamp, fresh = unit_state(unit)
        if fresh:
            return obj, False
        cmd = [hipcc] + HIPCC_FLAGS + ["-MD", "-MF", obj[:-2] + ".d", "-c", os.path.join(CSRC, unit),
                                       "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (unit, r.stdout))
        with open(stamp, "w") as f:
            f.write(_digest_files(_deps_of(obj) or [os.path.join(CSRC, unit)], flags_key))
        return obj, True

    harness_srcs = [os.path.join(CSRC, "utils", "synthetic_inputs.cpp"),
                    os.path.join(CSRC, "utils", "datagen.hpp")]
    harness_stamp = os.path.join(LIB_DIR, "libcuembed_harness.stamp")

    def compile_harness(_):
        # host-only synthetic-workload generator (libstdc++ <random>), plain g++
        digest = _digest_files(harness_srcs)
        if not force and os.path.exists(HARNESS_PATH) and os.path.exists(harness_stamp):
            with open(harness_stamp) as f:
                if f.read().strip() == digest:
                    return None
        cmd = [shutil.which("g++") or "g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I" + CSRC,
               harness_srcs[0], "-o", HARNESS_PATH]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("g++ failed for the harness library:\n" + r.stdout)
        with open(harness_stamp, "w") as f:
            f.write(digest)
        return None

    with ThreadPoolExecutor(max_workers=len(UNITS) + 1) as ex:
        harness_job = ex.submit(compile_harness, None)
        results = list(ex.map(compile_one, UNITS))
        harness_job.result()
    objs = [o for o, _ in results]
    if any(changed for _, changed in results) or not os.path.exists(LIB_PATH):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB_PATH] + objs
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout)
    build_torch_binding(force=force)
    return LIB_PATH


TORCH_BINDING_PATH = os.path.join(LIB_DIR, "libcuembed_pyt.so")


def build_torch_binding(force=False):
    """Compiles cuembed_amd/csrc/torch_binding.cpp (TORCH_LIBRARY / TORCH_LIBRARY_IMPL, no device code) with
    the host compiler against the installed torch and links it to libcuembed_amd.so.  Needs build() first."""
    import torch
    from torch.utils import cpp_extension
    src = os.path.join(CSRC, "torch_binding.cpp")
    stamp = os.path.join(LIB_DIR, "libcuembed_pyt.stamp")
    digest = _digest_files([src, os.path.join(ROOT, "include", "cuembed_amd.h")], torch.__version__)
    if not force and os.path.exists(TORCH_BINDING_PATH) and os.path.exists(stamp):
        with open(stamp) as f:
            if f.read().strip() == digest:
                return TORCH_BINDING_PATH
    torch_lib = cpp_extension.library_paths()[0]
    cmd = [shutil.which("g++") or "g++", "-O2", "-std=c++17", "-fPIC", "-shared",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",      # what torch's own headers key on under ROCm
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    cmd += ["-I" + p for p in cpp_extension.include_paths()] + ["-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include")]
    cmd += [src, "-o", TORCH_BINDING_PATH, "-L" + torch_lib, "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip",
            "-ltorch_hip", "-L" + LIB_DIR, "-lcuembed_amd", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + torch_lib]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("g++ failed for the torch binding:\n" + r.stdout)
    with open(stamp, "w") as f:
        f.write(digest)
    return TORCH_BINDING_PATH


def build_header_api_test(force=False):
    """Compiles tests/cpp/header_api_kat.hip against the header-only API (the C++ side of the
    drop-in boundary).  Returns the path of the executable."""
    src = os.path.join(ROOT, "tests", "cpp", "header_api_kat.hip")
    exe = os.path.join(ROOT, "tests", "cpp", "header_api_kat")
    stamp = exe + ".stamp"
    deps = [src]
    for dirpath, _, files in os.walk(os.path.join(CSRC, "cuembed", "include")):
        deps += [os.path.join(dirpath, f) for f in files]
    digest = _digest_files(deps)
    if not force and os.path.exists(exe) and os.path.exists(stamp):
        with open(stamp) as f:
            if f.read().strip() == digest:
                return exe
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-O2", "-std=c++17", "-munsafe-fp-atomics", "-I" + CSRC,
           src, "-o", exe]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for the header-only API test:\n" + r.stdout)
    with open(stamp, "w") as f:
        f.write(digest)
    return exe


def build_sort_unit_test(force=False):
    """Compiles tests/cpp/sort_route_unit.hip: host-side unit tests of the sort's integer building blocks (buffer
    routing, implicit payload division, tile maps, workspace plans).  The program makes no HIP call: it runs on the CPU."""
    src = os.path.join(ROOT, "tests", "cpp", "sort_route_unit.hip")
    exe = os.path.join(ROOT, "tests", "cpp", "sort_route_unit")
    stamp = exe + ".stamp"
    deps = [src]
    for dirpath, _, files in os.walk(os.path.join(CSRC, "cuembed", "include")):
        deps += [os.path.join(dirpath, f) for f in files]
    digest = _digest_files(deps)
    if not force and os.path.exists(exe) and os.path.exists(stamp):
        with open(stamp) as f:
            if f.read().strip() == digest:
                return exe
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-O1", "-std=c++17", "-I" + CSRC, src, "-o", exe]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for tests/cpp/sort_route_unit.hip:\n" + r.stdout)
    with open(stamp, "w") as f:
        f.write(digest)
    return exe


def build_device_blocks_test(force=False):
    """Compiles tests/cpp/device_blocks_unit.hip: unit tests of the gather / scatter building blocks (addressing, column
    slices, Pack / Arith / RowPool, FinishPooledRow, AccumulateRow) instantiated in small kernels and compared with host
    recomputation, bit for bit.  Needs a GPU to RUN (tests/test_gpu_device_blocks.py)."""
    src = os.path.join(ROOT, "tests", "cpp", "device_blocks_unit.hip")
    exe = os.path.join(ROOT, "tests", "cpp", "device_blocks_unit")
    stamp = exe + ".stamp"
    deps = [src]
    for dirpath, _, files in os.walk(os.path.join(CSRC, "cuembed", "include")):
        deps += [os.path.join(dirpath, f) for f in files]
    digest = _digest_files(deps)
    if not force and os.path.exists(exe) and os.path.exists(stamp):
        with open(stamp) as f:
            if f.read().strip() == digest:
                return exe
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-munsafe-fp-atomics", "-I" + CSRC, src, "-o", exe]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for tests/cpp/device_blocks_unit.hip:\n" + r.stdout)
    with open(stamp, "w") as f:
        f.write(digest)
    return exe


def build_manual_benchmark(force=False):
    """Compiles benchmarks/manual_benchmark.hip (C++ harness on the header-only API)."""
    src = os.path.join(ROOT, "benchmarks", "manual_benchmark.hip")
    gen = os.path.join(CSRC, "utils", "synthetic_inputs.cpp")
    exe = os.path.join(ROOT, "benchmarks", "manual_benchmark")
    stamp = exe + ".stamp"
    deps = [src, gen, os.path.join(CSRC, "utils", "datagen.hpp")]
    for dirpath, _, files in os.walk(os.path.join(CSRC, "cuembed", "include")):
        deps += [os.path.join(dirpath, f) for f in files]
    digest = _digest_files(deps)
    if not force and os.path.exists(exe) and os.path.exists(stamp):
        with open(stamp) as f:
            if f.read().strip() == digest:
                return exe
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-munsafe-fp-atomics", "-I" + CSRC,
           src, gen, "-ldl", "-o", exe]     # (-ldl: --check_result loads the CPU checker with dlopen)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for benchmarks/manual_benchmark.hip:\n" + r.stdout)
    with open(stamp, "w") as f:
        f.write(digest)
    return exe


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
