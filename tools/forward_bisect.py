#!/usr/bin/env python3
"""Times EmbeddingForward of several builds of the library side by side in ONE process (interleaved rounds,
same table, same index batches): tools/bisect/<sha>.so from tools/build_bisect_libs.sh plus the in-tree
library.  Used to decide whether a difference between two driver runs on different boxes belongs to the code.

    bash tools/build_bisect_libs.sh 46562d3 d4392df c833d28 HEAD      # here (cross-compiles)
    gpurun -- 'python tools/forward_bisect.py > gpurun_out/forward_bisect.txt'
"""
import ctypes
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    from cuembed_amd import harness
    libs = {os.path.basename(p)[:-3]: ctypes.CDLL(p) for p in sorted(glob.glob(os.path.join(ROOT, "tools", "bisect", "*.so")))}
    libs["in-tree"] = ctypes.CDLL(os.path.join(ROOT, "cuembed_amd", "lib", "libcuembed_amd.so"))
    dev = torch.device("cuda", 0)
    B, H, W = 65536, 64, 256
    nbytes = 2 * B * (H + 1) * W
    table = torch.empty((10_000_000, W), dtype=torch.float16, device=dev).uniform_(-1, 1)
    out = torch.empty((B, W), dtype=torch.float16, device=dev)
    ref = torch.empty((B, W), dtype=torch.float16, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    VP = ctypes.c_void_p

    def launch(L, idx, o):
        L.cuembed_embedding_forward(VP(table.data_ptr()), 1, W, VP(idx.data_ptr()), 0, None, 0, None, B, H, 0, 0,
                                    VP(o.data_ptr()), stream)

    for alpha in (1.15, 0.0):
        idx = harness.generate_indices(10_000_000, 4 * B, H, alpha=alpha).reshape(4, -1)
        idxs = [torch.from_numpy(np.ascontiguousarray(idx[i])).to(dev) for i in range(4)]
        launch(libs["in-tree"], idxs[0], ref)
        res = {}
        for steps in (20, 200):
            for r in range(9):
                for name, L in libs.items():
                    if r == 0:
                        launch(L, idxs[0], out)
                        torch.cuda.synchronize()
                        assert torch.equal(out, ref), name
                    a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for t in range(steps):
                        launch(L, idxs[t % 4], out)
                    z.record()
                    z.synchronize()
                    res.setdefault((name, steps), []).append(a.elapsed_time(z) / steps)
        for (name, steps), ms in res.items():
            ms = sorted(ms)
            print("alpha=%.2f  %-10s  %3d launches per timing  median %.4f ms (%6.0f GB/s)  min %.4f  max %.4f"
                  % (alpha, name, steps, ms[len(ms) // 2], nbytes / ms[len(ms) // 2] / 1e6, ms[0], ms[-1]), flush=True)


if __name__ == "__main__":
    main()
