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
UNITS = ["c_api_forward.hip", "c_api_backward.hip", "c_api_transforms.hip"]
ARCH = "gfx950"

HIPCC_FLAGS = [
    "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC",
    # float atomics as hardware instructions (global_atomic_add_f32 / pk_add_f16)
    "-munsafe-fp-atomics",
    "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; a ROCm toolchain is required to build cuembed_amd")
    return exe


def _source_digest():
    h = hashlib.sha256()
    h.update(" ".join(HIPCC_FLAGS).encode())
    for base in (CSRC, os.path.join(ROOT, "include")):
        for dirpath, _, files in sorted(os.walk(base)):
            for f in sorted(files):
                if f.endswith((".hpp", ".hip", ".h", ".cuh", ".cpp")):
                    with open(os.path.join(dirpath, f), "rb") as fh:
                        h.update(f.encode())
                        h.update(fh.read())
    return h.hexdigest()


def build(force=False, verbose=False):
    """Compile if sources changed.  Returns the path of the shared library."""
    os.makedirs(LIB_DIR, exist_ok=True)
    os.makedirs(OBJ_DIR, exist_ok=True)
    stamp = os.path.join(LIB_DIR, "libcuembed_amd.stamp")
    digest = _source_digest()
    if not force and os.path.exists(LIB_PATH) and os.path.exists(HARNESS_PATH) and \
            os.path.exists(stamp):
        with open(stamp) as f:
            if f.read().strip() == digest:
                return LIB_PATH
    hipcc = _hipcc()

    def compile_one(unit):
        obj = os.path.join(OBJ_DIR, unit.replace(".hip", ".o"))
        cmd = [hipcc] + HIPCC_FLAGS + ["-c", os.path.join(CSRC, unit), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (unit, r.stdout))
        return obj

    def compile_harness(_):
        # host-only synthetic-workload generator (libstdc++ <random>), plain g++
        cmd = [shutil.which("g++") or "g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I" + CSRC,
               os.path.join(CSRC, "utils", "synthetic_inputs.cpp"), "-o", HARNESS_PATH]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("g++ failed for the harness library:\n" + r.stdout)
        return None

    with ThreadPoolExecutor(max_workers=len(UNITS) + 1) as ex:
        harness_job = ex.submit(compile_harness, None)
        objs = list(ex.map(compile_one, UNITS))
        harness_job.result()
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB_PATH] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout)
    with open(stamp, "w") as f:
        f.write(digest)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
