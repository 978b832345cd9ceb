#!/usr/bin/env python3
"""Times the forward-kernel variants of tools/forward_variants.hip side by side (interleaved
rounds in one process, median/min reported) on four regimes of the C2 shape:
L2-resident table, Infinity-Cache-resident table, power-law alpha=1.15, uniform over 10M rows."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SO = os.path.join(ROOT, "tools", "forward_variants.so")


def build(so=SO, defines=()):
    src = os.path.join(ROOT, "tools", "forward_variants.hip")
    hdr = os.path.join(ROOT, "cuembed_amd", "csrc", "cuembed", "include", "gather_reduce_kernels.hpp")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                               "-munsafe-fp-atomics", "-I" + os.path.join(ROOT, "cuembed_amd", "csrc"),
                               "-I" + os.path.join(ROOT, "include"), src, "-o", so] + ["-D" + d for d in defines])


# A/B of compile-time policies of the product kernel (variant 0): each policy is its own .so.
POLICIES = {"plain": (), "row_nt": ("CUEMBED_TUNE_ROW_LOAD_NT",),
            "asm_plain": ('CUEMBED_TUNE_ROW_LOAD_ASM=""',), "asm_sc0": ('CUEMBED_TUNE_ROW_LOAD_ASM="sc0"',),
            "asm_sc1": ('CUEMBED_TUNE_ROW_LOAD_ASM="sc1"',), "asm_sc0_sc1": ('CUEMBED_TUNE_ROW_LOAD_ASM="sc0 sc1"',),
            "asm_nt": ('CUEMBED_TUNE_ROW_LOAD_ASM="nt"',)}


def policy_so(name):
    return SO if name == "plain" else SO.replace(".so", "_%s.so" % name)


def main():
    if "--build-only" in sys.argv:
        build()
        for name, defs in POLICIES.items():
            build(policy_so(name), defs)
        return
    if "--policies" in sys.argv:
        return policies_main()
    import numpy as np
    import torch
    from cuembed_amd import harness
    L = ctypes.CDLL(SO)
    L.variant_name.restype = ctypes.c_char_p
    nv = L.variant_count()
    dev = torch.device("cuda", 0)
    B, H, W = 65536, 64, 256
    out = torch.empty((B, W), dtype=torch.float16, device=dev)
    ref = torch.empty((B, W), dtype=torch.float16, device=dev)
    nbytes = 2 * B * (H + 1) * W
    g = torch.Generator(device=dev).manual_seed(0)
    big = torch.empty((10_000_000, W), dtype=torch.float16, device=dev).uniform_(-1, 1)
    regimes = []
    for name, rows in [("L2 (2 MiB table)", 4096), ("MALL (32 MiB table)", 65536)]:
        regimes.append((name, big[:rows], [torch.randint(0, rows, (B * H,), device=dev, dtype=torch.int32, generator=g)
                                            for _ in range(2)]))
    for name, alpha in [("C2 alpha=1.15", 1.15), ("alpha=0 (HBM)", 0.0)]:
        idx = harness.generate_indices(10_000_000, 2 * B, H, alpha=alpha).reshape(2, -1)
        regimes.append((name, big, [torch.from_numpy(np.ascontiguousarray(idx[i])).to(dev) for i in range(2)]))
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    only = None if "--variants" not in sys.argv else \
        [int(x) for x in sys.argv[sys.argv.index("--variants") + 1].split(",")]
    spbs = [8] if "--spb" not in sys.argv else [int(x) for x in sys.argv[sys.argv.index("--spb") + 1].split(",")]

    def launch(v, table, idx, spb, o):
        L.variant_launch(v, ctypes.c_void_p(table.data_ptr()), W, B, ctypes.c_void_p(idx.data_ptr()), H,
                         ctypes.c_void_p(o.data_ptr()), spb, stream)

    for name, table, idxs in regimes:
        print("== %s" % name)
        launch(0, table, idxs[0], 8, ref)
        results = {}
        rounds, inner = 7, 10
        for r in range(rounds):
            for v in range(nv):
                if only is not None and v not in only:
                    continue
                lb = int(L.variant_name(v).decode().split("lb")[1])
                for spb in spbs:
                    if spb * (W // 8) > lb:
                        continue
                    if r == 0:
                        launch(v, table, idxs[0], spb, out)
                        torch.cuda.synchronize()
                        assert torch.equal(out, ref), (v, spb)
                    a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for t in range(inner):
                        launch(v, table, idxs[t % 2], spb, out)
                    z.record()
                    z.synchronize()
                    results.setdefault((v, spb), []).append(a.elapsed_time(z) / inner)
        for (v, spb), ms in sorted(results.items()):
            ms = sorted(ms)
            med = ms[len(ms) // 2]
            print("  %-22s spb=%2d  median %.4f ms (%6.0f GB/s)  min %.4f ms" %
                  (L.variant_name(v).decode(), spb, med, nbytes / med / 1e6, ms[0]))


def policies_main():
    import numpy as np
    import torch
    from cuembed_amd import harness
    libs = {n: ctypes.CDLL(policy_so(n)) for n in POLICIES}
    dev = torch.device("cuda", 0)
    B, H, W = 65536, 64, 256
    out = torch.empty((B, W), dtype=torch.float16, device=dev)
    ref = torch.empty((B, W), dtype=torch.float16, device=dev)
    nbytes = 2 * B * (H + 1) * W
    big = torch.empty((10_000_000, W), dtype=torch.float16, device=dev).uniform_(-1, 1)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    regimes = [("table of 32 rows (L1)", 32), ("table of 4096 rows (L2)", 4096), (1.15, None), (1.05, None), (0.0, None)]
    for alpha, small_rows in regimes:
        if small_rows is not None:
            g = torch.Generator(device=dev).manual_seed(1)
            idxs = [torch.randint(0, small_rows, (B * H,), device=dev, dtype=torch.int32, generator=g) for _ in range(4)]
        else:
            idx = harness.generate_indices(10_000_000, 4 * B, H, alpha=alpha).reshape(4, -1)
            idxs = [torch.from_numpy(np.ascontiguousarray(idx[i])).to(dev) for i in range(4)]

        def launch(L, i, o):
            L.variant_launch(0, ctypes.c_void_p(big.data_ptr()), W, B, ctypes.c_void_p(idxs[i].data_ptr()), H,
                             ctypes.c_void_p(o.data_ptr()), 8, stream)
        launch(libs["plain"], 0, ref)
        res = {}
        for r in range(9):
            for n, L in libs.items():
                if r == 0:
                    launch(L, 0, out)
                    torch.cuda.synchronize()
                    assert torch.equal(out, ref), n
                a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for t in range(12):
                    launch(L, t % 4, out)
                z.record()
                z.synchronize()
                res.setdefault(n, []).append(a.elapsed_time(z) / 12)
        for n, ms in res.items():
            ms = sorted(ms)
            print("%-24s  %-12s median %.4f ms (%6.0f GB/s)  min %.4f" %
                  (("alpha=%.2f" % alpha) if small_rows is None else alpha, n, ms[len(ms) // 2],
                   nbytes / ms[len(ms) // 2] / 1e6, ms[0]))


if __name__ == "__main__":
    main()
