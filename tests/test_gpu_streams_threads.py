"""Stream-ordered, reentrant entry points (SURVEY section 8b "Threading / async"): no globals, no
host synchronisation -- concurrent calls from several host threads, each on its own HIP stream,
must not disturb each other."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_concurrent_host_threads_on_separate_streams(oracle):
    import cuembed_amd as ce
    ncat, W, B, H = 20000, 128, 3000, 24
    cases = []
    for t in range(4):
        a = oracle.allocate_forward(ncat + t, W, B, H, alpha=[0.0, 1.05, 1.15, 1.3][t], elem=np.float16)
        sid = oracle.extract_row_ids_from_fixed(B, H)
        ti, ts, _ = oracle.transpose(sid, a["indices"])
        remap = oracle.compute_compressed_grad_indices(ti)
        gy = (np.mod(oracle.allocate_grad_y(B * W), 3) - 1).reshape(B, W).astype(np.float16)
        nu = int(remap[-1]) + 1
        want_g, want_inv = oracle.embedding_backward(gy.astype(np.float32), W, nu, ti, ts, remap)
        cases.append(dict(a=a, want_fwd=oracle.embedding_forward(a["table"], a["indices"], num_hots=H),
                          ti=ti, ts=ts, remap=remap, gy=gy, nu=nu, want_g=want_g, want_inv=want_inv))
    errors = []

    def worker(t):
        try:
            c = cases[t]
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                table = torch.from_numpy(c["a"]["table"]).cuda()
                idx = torch.from_numpy(c["a"]["indices"]).cuda()
                gy = torch.from_numpy(c["gy"]).cuda()
                for rep in range(20):
                    out = ce.embedding_forward(table, idx, num_hots=H)
                    sid = ce.extract_row_ids_from_fixed(B, H, torch.int32, "cuda")
                    ti, ts, _ = ce.transpose(sid, idx)
                    remap = ce.compute_compressed_grad_indices(ti)
                    g, inv = ce.embedding_backward(gy, c["nu"], ti, ts, remap)
                stream.synchronize()
                assert np.array_equal(out.cpu().numpy().view(np.uint16), c["want_fwd"].view(np.uint16))
                assert np.array_equal(ti.cpu().numpy(), c["ti"]) and np.array_equal(ts.cpu().numpy(), c["ts"])
                assert np.array_equal(remap.cpu().numpy(), c["remap"])
                assert np.array_equal(g.float().cpu().numpy(), c["want_g"])
                assert np.array_equal(inv.cpu().numpy(), c["want_inv"])
        except Exception as e:  # noqa: BLE001 -- reported by the main thread
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    torch.cuda.synchronize()
    assert not errors, errors
    assert ce._lib.lib().cuembed_peek_last_error() == 0
