"""BASELINE.json configs[0] -- fp32 sum, 1k x 32 table, batch 1,024, fixed hotness 8 -- on the HIP path through the
C ABI, at alpha 0 and 1.15: bit for bit against the oracle AND against the FNV digests of the reference's own CPU code
recorded in tests/golden/survey_digests.json (reference: tests/test_embedding_against_cpu.cu:153-163 compares
GPU and CPU results with exact equality for the unweighted sum).  The whole pipeline of the config is covered: forward,
Transpose, ComputeCompressedGradIndices, compressed EmbeddingBackward and its inverse mapping."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ce():
    import cuembed_amd
    assert torch.cuda.is_available()
    return cuembed_amd


@pytest.mark.parametrize("alpha_key", ["alpha_0", "alpha_1.15"])
def test_config1_forward_and_backward_match_oracle_and_reference_digests(ce, oracle, golden_dir, alpha_key):
    with open(os.path.join(golden_dir, "survey_digests.json")) as f:
        d = json.load(f)
    shape, want = d["shape"], d[alpha_key]
    rows, width, batch, hot = shape["num_categories"], shape["embed_width"], shape["batch_size"], shape["hotness"]
    a = oracle.allocate_forward(rows, width, batch, hot, alpha=want["alpha"])
    fnv = lambda x: "%016x" % oracle.fnv1a64(np.ascontiguousarray(x))
    assert fnv(a["indices"]) == want["fnv_idx"]                       # the inputs ARE the reference's
    assert a["indices"][:8].tolist() == want["idx_head"]
    table = torch.from_numpy(a["table"]).cuda()
    idx = torch.from_numpy(a["indices"]).cuda()
    # ---- forward
    out = ce.embedding_forward(table, idx, num_hots=hot)
    got = out.cpu().numpy()
    o_out = oracle.embedding_forward(a["table"], a["indices"], num_hots=hot)
    assert np.array_equal(got.view(np.uint32), o_out.view(np.uint32))
    assert fnv(got) == want["fnv_res"]
    np.testing.assert_allclose(got.reshape(-1)[:4], want["res_head"], rtol=1e-6)
    # ---- index work of the backward
    sid = ce.extract_row_ids_from_fixed(batch, hot, torch.int32, "cuda")
    t_idx, t_sid, _ = ce.transpose(sid, idx)
    remap = ce.compute_compressed_grad_indices(t_idx)
    assert fnv(t_idx.cpu().numpy()) == want["fnv_t_idx"]
    assert fnv(t_sid.cpu().numpy()) == want["fnv_t_sid"]
    assert fnv(remap.cpu().numpy()) == want["fnv_remap"]
    num_unique = int(remap[-1].item()) + 1
    assert num_unique == want["num_unique"]
    # ... and in ONE call (fixed hotness, remapped ids from the same launch): the same arrays
    f_idx, f_sid, _, f_remap = ce.transpose_fixed_hotness(idx, batch, hot, num_categories=rows, remapped=True)
    assert torch.equal(f_idx, t_idx) and torch.equal(f_sid, t_sid) and torch.equal(f_remap, remap)
    # ---- compressed backward (grad_y: the reference's recipe, seed 654321)
    gy = oracle.allocate_grad_y(batch * width).reshape(batch, width)
    grad, inv = ce.embedding_backward(torch.from_numpy(gy).cuda(), num_unique, t_idx, t_sid, remap)
    assert fnv(grad.cpu().numpy()) == want["fnv_grad"]
    assert fnv(inv.cpu().numpy()) == want["fnv_inv"]
    o_grad, o_inv = oracle.embedding_backward(gy, width, num_unique, t_idx.cpu().numpy(), t_sid.cpu().numpy(),
                                              remap.cpu().numpy())
    assert np.array_equal(grad.cpu().numpy(), o_grad) and np.array_equal(inv.cpu().numpy(), o_inv)
