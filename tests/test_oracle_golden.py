"""Pins the CPU oracle (oracle/cuembed_oracle.cpp) to the reference:
 * every known-answer vector of the reference's own tests (reference_kats.json),
 * the digests of the reference CPU code's outputs recorded in SURVEY.md 8(c),
 * the reference's own index generator (golden vectors + oracle/_ref when built),
 * numpy.float16 for the software binary16 arithmetic.
CPU only."""
import json
import os

import numpy as np
import pytest

ELEMS = [np.float32, np.float16]
IDXS = [np.int32, np.int64]


@pytest.fixture(scope="module")
def kats(golden_dir):
    with open(os.path.join(golden_dir, "reference_kats.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def digests(golden_dir):
    with open(os.path.join(golden_dir, "survey_digests.json")) as f:
        return json.load(f)


# ---- binary16 arithmetic ---------------------------------------------------
def test_half_conversions_match_numpy(oracle):
    L = oracle.lib()
    allh = np.arange(65536, dtype=np.uint16)
    ref = allh.view(np.float16).astype(np.float32)
    got = np.array([L.oracle_h2f(int(h)) for h in allh], dtype=np.float32)
    nan = np.isnan(ref)
    assert (np.isnan(got) == nan).all()
    assert (got[~nan].view(np.uint32) == ref[~nan].view(np.uint32)).all()
    rng = np.random.default_rng(7)
    f = np.concatenate([
        rng.standard_normal(20000).astype(np.float32) * 10.0 ** rng.integers(-9, 6, 20000),
        ref[~nan],
        # exact midpoints between consecutive halves (ties-to-even)
        ((ref[~nan][:-1].astype(np.float64) + ref[~nan][1:].astype(np.float64)) / 2).astype(np.float32),
        np.array([65504.0, 65519.99, 65520.0, 1e9, -1e9, 2.0 ** -25, 2.0 ** -24, 5.9e-8, 0.0, -0.0],
                 dtype=np.float32),
    ]).astype(np.float32)
    f = f[np.isfinite(f)]
    with np.errstate(over="ignore"):
        want = f.astype(np.float16).view(np.uint16)
    got = np.array([L.oracle_f2h(float(x)) for x in f], dtype=np.uint16)
    assert (got == want).all()


# ---- reference KATs ----------------------------------------------------------
@pytest.mark.parametrize("elem", ELEMS)
@pytest.mark.parametrize("idx", IDXS)
@pytest.mark.parametrize("csr", [False, True])
def test_forward_kat(oracle, kats, elem, idx, csr):
    k = kats["forward"]
    table = np.array(k["embedding"], dtype=elem).reshape(k["num_categories"], k["embed_width"])
    indices = np.array(k["indices"], dtype=idx)
    weights = np.array(k["weights"], dtype=elem)
    offsets = np.array(k["offsets"], dtype=np.int32) if csr else None
    hots = 0 if csr else k["hotness"]
    B = k["batch_size"]
    for mode, w, key in [("sum", None, "sum"), ("sum", weights, "sum_weighted"),
                         ("mean", None, "mean")] + ([] if csr else [("concat", None, "concat")]):
        out = oracle.embedding_forward(table, indices, offsets, w, batch_size=B, num_hots=hots, mode=mode)
        assert out.ravel().tolist() == k[key], (mode, key)
        if elem == np.float16:  # fp16 accumulation is exact on these values too
            out = oracle.embedding_forward(table, indices, offsets, w, batch_size=B, num_hots=hots,
                                           mode=mode, fp16_math=True)
            assert out.ravel().tolist() == k[key]


def test_forward_contract_violations(oracle, kats):
    k = kats["forward"]
    table = np.array(k["embedding"], dtype=np.float32).reshape(5, 4)
    idx = np.array(k["indices"], dtype=np.int32)
    w = np.array(k["weights"], dtype=np.float32)
    off = np.array(k["offsets"], dtype=np.int32)
    with pytest.raises(ValueError):  # weights with concat
        oracle.embedding_forward(table, idx, None, w, batch_size=2, num_hots=2, mode="concat")
    with pytest.raises(ValueError):  # CSR with concat
        oracle.embedding_forward(table, idx, off, None, batch_size=2, num_hots=0, mode="concat")
    with pytest.raises(ValueError):  # CSR and fixed at once
        oracle.embedding_forward(table, idx, off, None, batch_size=2, num_hots=2, mode="sum")


@pytest.mark.parametrize("elem", ELEMS)
@pytest.mark.parametrize("idx", IDXS)
@pytest.mark.parametrize("compressed", [False, True])
@pytest.mark.parametrize("skip_init", [False, True])
def test_backward_kat(oracle, kats, elem, idx, compressed, skip_init):
    k = kats["backward"]
    t_idx = np.array(k["transpose_indices"], dtype=idx)
    remap = np.array(k["transpose_remapped_indices"], dtype=idx) if compressed else None
    w = np.array(k["transpose_weights"], dtype=elem)
    rows = k["num_unique"] if compressed else k["num_categories"]
    pre = "compressed_grad_" if compressed else "grad_"
    for mode in ["sum", "concat"]:
        sid = np.array(k["transpose_sample_ids" + ("_concat" if mode == "concat" else "")], dtype=idx)
        gy = np.array(k["grad_y_" + mode], dtype=elem).reshape(-1, k["embed_width"])
        for weighted in [False, True]:
            buf = np.zeros((rows, k["embed_width"]), dtype=elem) if skip_init else \
                np.full((rows, k["embed_width"]), 77, dtype=elem)
            grad, inv = oracle.embedding_backward(gy, k["embed_width"], rows, t_idx, sid, remap,
                                                  w if weighted else None, skip_grad_init=skip_init,
                                                  grad_embedding=buf)
            assert grad.ravel().tolist() == k[pre + mode + ("_weighted" if weighted else "")]
            if compressed:
                assert inv[:k["num_unique"]].tolist() == k["inverse_mapping"]


@pytest.mark.parametrize("elem", ELEMS)
@pytest.mark.parametrize("idx", IDXS)
@pytest.mark.parametrize("stable", [True, False])
def test_transpose_kat(oracle, kats, elem, idx, stable):
    k = kats["transpose"]
    for weighted in [False, True]:
        w = np.array(k["weights"], dtype=elem) if weighted else None
        ti, ts, tw = oracle.transpose(np.array(k["sample_ids"], dtype=idx),
                                      np.array(k["indices"], dtype=idx), w, stable=stable)
        assert ti.tolist() == k["transpose_indices"]
        assert ts.tolist() == k["transpose_sample_ids"]
        if weighted:
            assert tw.tolist() == k["transpose_weights"]


@pytest.mark.parametrize("idx", IDXS)
def test_readme_examples(oracle, kats, idx):
    r = kats["readme_examples"]
    e = r["extract_fixed"]
    assert oracle.extract_row_ids_from_fixed(e["batch_size"], e["num_hots"], idx).tolist() == e["row_ids"]
    for off_t in [np.int32, np.int64]:
        assert oracle.extract_row_ids_from_csr(np.array(r["extract_csr"]["offsets"], dtype=off_t),
                                               idx).tolist() == r["extract_csr"]["row_ids"]
    assert oracle.extract_row_ids_for_concat(r["extract_concat"]["nnz"], idx).tolist() == \
        r["extract_concat"]["row_ids"]
    c = r["compressed"]
    assert oracle.compute_compressed_grad_indices(np.array(c["indices"], dtype=idx)).tolist() == c["remapped"]


# ---- digests of the reference CPU code's own outputs -----------
@pytest.mark.parametrize("key", ["alpha_0", "alpha_1.15"])
def test_survey_digests(oracle, digests, key):
    s, d = digests["shape"], digests[key]
    a = oracle.allocate_forward(s["num_categories"], s["embed_width"], s["batch_size"], s["hotness"],
                                alpha=d["alpha"])
    h = lambda x: "%016x" % oracle.fnv1a64(x)  # noqa: E731
    assert a["indices"][:8].tolist() == d["idx_head"]
    assert h(a["indices"]) == d["fnv_idx"]
    if "fnv_emb" in d:
        assert h(a["table"]) == d["fnv_emb"]
        assert h(a["weights"]) == d["fnv_weights"]
        np.testing.assert_allclose(a["table"].ravel()[:4], d["emb_head"], rtol=1e-7)
    res = oracle.embedding_forward(a["table"], a["indices"], num_hots=s["hotness"])
    assert h(res) == d["fnv_res"]
    np.testing.assert_allclose(res.ravel()[:4], d["res_head"], rtol=1e-6)
    sid = oracle.extract_row_ids_from_fixed(s["batch_size"], s["hotness"])
    for stable in [True, False]:  # no sample repeats an index -> both orders coincide
        t_idx, t_sid, _ = oracle.transpose(sid, a["indices"], stable=stable)
        assert h(t_idx) == d["fnv_t_idx"] and h(t_sid) == d["fnv_t_sid"]
    remap = oracle.compute_compressed_grad_indices(t_idx)
    assert h(remap) == d["fnv_remap"]
    nu = int(remap[-1]) + 1
    assert nu == d["num_unique"]
    gy = oracle.allocate_grad_y(s["batch_size"] * s["embed_width"]).reshape(-1, s["embed_width"])
    grad, inv = oracle.embedding_backward(gy, s["embed_width"], nu, t_idx, t_sid, remap)
    assert h(grad) == d["fnv_grad"] and h(inv) == d["fnv_inv"]


# ---- index generator -----------------------------------------------------------
def test_generator_golden_vectors(oracle, golden_dir, digests):
    with open(os.path.join(golden_dir, "psx_vectors.json")) as f:
        cases = json.load(f)["cases"]
    for c in cases:
        v = oracle.psx_samples(c["num_categories_arg"], c["hot"], c["alpha"], c["n_samples"],
                               index=np.dtype(c["index"]).type, shuffle=c["shuffle"],
                               permute=c["permute"])
        if "samples" in c:
            assert v.tolist() == c["samples"], c
        else:
            assert "%016x" % oracle.fnv1a64(v) == c["fnv"] and v[0][:8].tolist() == c["head"]
    g = digests["generator"]
    assert oracle.psx_samples(999, 8, 1.15, 1)[0].tolist() == g["psx_999_8_1.15_first_sample"]
    assert oracle.psx_samples(9999999, 64, 1.15, 1)[0][:8].tolist() == g["psx_9999999_64_1.15_first8"]
    assert oracle.psx_samples(9999999, 64, 0.0, 1)[0][:8].tolist() == g["psx_9999999_64_0_first8"]


def test_generator_against_reference_build(oracle):
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    for n, hot, alpha, dt in [(999, 8, 1.15, np.int32), (4095, 26, 0.0, np.int64),
                              (99999, 64, 1.05, np.int32), (12345, 7, 2.5, np.int64)]:
        for shuf in (False, True):
            for perm in (False, True):
                a = oracle.psx_samples(n, hot, alpha, 50, index=dt, shuffle=shuf, permute=perm)
                b = oracle.psx_samples(n, hot, alpha, 50, index=dt, shuffle=shuf, permute=perm,
                                       use_reference=True)
                assert (a == b).all()


def test_generator_properties(oracle):
    # tests/test_embedding_allocation.cu:101-126, tests/test_datagen.cpp:143-160
    a = oracle.allocate_forward(7 * 1024, 4, 2048, 32, alpha=1.5, elem=np.float16)
    idx = a["indices"].reshape(2048, 32)
    assert idx.min() >= 0 and idx.max() < 7 * 1024
    assert all(len(set(r.tolist())) == 32 for r in idx)
    assert set(np.unique(a["weights"]).tolist()) <= {0.25, 0.5}
    c = oracle.allocate_forward(1024, 4, 512, 16, alpha=0.0, is_csr=True)
    assert c["offsets"][0] == 0 and (np.diff(c["offsets"]) >= 0).all() and (np.diff(c["offsets"]) <= 16).all()
    assert c["indices"].shape[0] == c["offsets"][-1]


# ---- fp16 forward/backward vs an independent numpy restatement ----------------
def _np_forward(table, idx2d, w2d, mode, fp16_math):
    B, H = idx2d.shape
    acc_t = np.float16 if (fp16_math and table.dtype == np.float16) else np.float32
    acc = np.zeros((B, table.shape[1]), dtype=acc_t)
    for j in range(H):
        v = table[idx2d[:, j]].astype(acc_t)
        if w2d is not None:
            v = (v * w2d[:, j:j + 1].astype(acc_t)).astype(acc_t)
        acc = (acc + v).astype(acc_t)
    if mode == "mean":
        acc = (acc * acc_t(np.float32(1.0) / np.float32(H))).astype(acc_t)
    return acc.astype(table.dtype)


@pytest.mark.parametrize("elem", ELEMS)
@pytest.mark.parametrize("fp16_math", [False, True])
@pytest.mark.parametrize("mode,weighted", [("sum", False), ("sum", True), ("mean", False)])
def test_forward_against_numpy(oracle, elem, fp16_math, mode, weighted):
    rng = np.random.default_rng(3)
    table = rng.uniform(-1, 1, (500, 36)).astype(elem)
    idx = rng.integers(0, 500, (257, 26)).astype(np.int64)
    w = rng.uniform(0, 1, (257, 26)).astype(elem) if weighted else None
    got = oracle.embedding_forward(table, idx.ravel(), None, None if w is None else w.ravel(),
                                   num_hots=26, mode=mode, fp16_math=fp16_math)
    want = _np_forward(table, idx, w, mode, fp16_math)
    assert (got.view(np.uint16 if elem == np.float16 else np.uint32) ==
            want.view(np.uint16 if elem == np.float16 else np.uint32)).all()
    # multi-threaded oracle = single-threaded oracle
    got_mt = oracle.embedding_forward(table, idx.ravel(), None, None if w is None else w.ravel(),
                                      num_hots=26, mode=mode, fp16_math=fp16_math, threads=4)
    assert (got_mt == got).all()
