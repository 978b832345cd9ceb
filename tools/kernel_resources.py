#!/usr/bin/env python3
"""Registers / LDS / scratch / occupancy of every kernel of one translation unit, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks (the ROCm counterpart of the reference's
`--ptxas-options=-v`, CMakeLists.txt:22).  No GPU needed.

    python tools/kernel_resources.py cuembed_amd/csrc/c_api_transforms.hip [substring ...]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    unit = sys.argv[1]
    want = sys.argv[2:]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "cuembed_amd", "csrc"),
           "-Rpass-analysis=kernel-resource-usage", "-c", unit, "-o", "/dev/null"]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    cur = None
    rows = {}
    for line in out.splitlines():
        m = re.search(r"remark: (?:[^:]+:\d+:\d+: )?\s*Function Name: (\S+)", line) or re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            rows[cur] = {}
            continue
        m = re.search(r"\s(VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|"
                      r"LDS Size \[bytes/block\]|VGPRs Spill|SGPRs Spill): (\d+)", line)
        if m and cur:
            rows[cur][m.group(1).split(" [")[0]] = int(m.group(2))
    demangle = subprocess.run(["c++filt"] + list(rows), stdout=subprocess.PIPE, text=True)
    names = demangle.stdout.splitlines() if demangle.returncode == 0 else list(rows)
    print("%-6s %-6s %-8s %-8s %-5s  %s" % ("VGPR", "SGPR", "scratch", "LDS", "occ", "kernel"))
    for mangled, name in zip(rows, names):
        if want and not any(w in name for w in want):
            continue
        r = rows[mangled]
        short = re.sub(r"\(.*$", "", name)
        print("%-6s %-6s %-8s %-8s %-5s  %s" % (r.get("VGPRs"), r.get("SGPRs"), r.get("ScratchSize"),
                                              r.get("LDS Size"), r.get("Occupancy"), short))


if __name__ == "__main__":
    main()
