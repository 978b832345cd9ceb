"""The product's synthetic-workload generator (cuembed_amd/csrc/utils) against the oracle, the
reference-built generator and the golden vectors.  CPU only."""
import json
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def harness():
    from cuembed_amd import build, harness
    build.build()
    return harness


def test_indices_match_golden_and_oracle(harness, oracle, golden_dir):
    with open(os.path.join(golden_dir, "survey_digests.json")) as f:
        d = json.load(f)
    got = harness.generate_indices(1024, 1024, 8, alpha=1.15)
    assert "%016x" % oracle.fnv1a64(got) == d["alpha_1.15"]["fnv_idx"]
    got = harness.generate_indices(1024, 1024, 8, alpha=0.0)
    assert "%016x" % oracle.fnv1a64(got) == d["alpha_0"]["fnv_idx"]
    assert harness.generate_indices(10_000_000, 1, 64, alpha=1.15)[:8].tolist() == \
        d["generator"]["psx_9999999_64_1.15_first8"]
    for idx in (np.int32, np.int64):
        for shuf in (False, True):
            for perm in (False, True):
                a = harness.generate_indices(5000, 300, 17, alpha=1.05, index=idx, shuffle=shuf, permute=perm)
                b = oracle.generate_indices(5000, 300, 17, alpha=1.05, index=idx, shuffle=shuf, permute=perm)
                assert np.array_equal(a, b)


@pytest.mark.parametrize("elem", [np.float32, np.float16])
@pytest.mark.parametrize("csr", [False, True])
def test_allocate_forward_matches_oracle(harness, oracle, golden_dir, elem, csr):
    a = harness.allocate_forward(1024, 32, 1024, 8, alpha=1.15, is_csr=csr, elem=elem)
    b = oracle.allocate_forward(1024, 32, 1024, 8, alpha=1.15, is_csr=csr, elem=elem)
    for k in ("table", "offsets", "indices", "weights"):
        assert np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)), k
    c = harness.allocate_forward(1024, 32, 1024, 8, alpha=1.15, is_csr=csr, elem=elem, with_table=False)
    assert c["table"] is None
    for k in ("offsets", "indices", "weights"):
        assert np.array_equal(c[k].view(np.uint8), b[k].view(np.uint8)), k
    if elem == np.float32 and not csr:
        with open(os.path.join(golden_dir, "survey_digests.json")) as f:
            d = json.load(f)
        assert "%016x" % oracle.fnv1a64(a["table"]) == d["alpha_0"]["fnv_emb"]
        assert "%016x" % oracle.fnv1a64(a["weights"]) == d["alpha_0"]["fnv_weights"]


def test_grad_y_matches_oracle(harness, oracle):
    for elem in (np.float32, np.float16):
        assert np.array_equal(harness.allocate_grad_y(5000, elem), oracle.allocate_grad_y(5000, elem))


def test_one_hot_power_law_matches_analytical_distribution(harness):
    """tests/test_datagen.cpp:109-139: one-hot Psx generator, 9 categories, alpha 1.15, 4M draws:
    empirical frequencies within 1e-3 of the normalised integral of x^-alpha over [i, i+1)."""
    n_cat, alpha, draws = 9, 1.15, 4_000_000
    idx = harness.generate_indices(n_cat + 1, draws, 1, alpha=alpha, shuffle=False, permute=False)
    assert idx.min() >= 1 and idx.max() <= n_cat
    i = np.arange(1, n_cat + 1, dtype=np.float64)
    p = (-alpha) * i ** (1 - alpha) - (-alpha) * (i + 1) ** (1 - alpha)
    p /= p.sum()
    freq = np.bincount(idx, minlength=n_cat + 1)[1:] / draws
    assert np.abs(freq - p).max() < 1e-3


def test_multi_hot_no_repeats_in_range(harness):
    """tests/test_datagen.cpp:143-160."""
    idx = harness.generate_indices(1001, 40000, 64, alpha=1.15, shuffle=False, permute=False).reshape(40000, 64)
    assert idx.min() >= 1 and idx.max() <= 1000
    assert (np.diff(np.sort(idx, axis=1), axis=1) > 0).all()


def test_alpha_one_is_refused_instead_of_hanging(harness):
    """alpha = 1 makes the reference's inverse-CDF recipe draw the same id forever (span = 0, datagen.cpp:39-50), and a
    sample of several DISTINCT ids then never fills: the Python layer refuses it."""
    with pytest.raises(ValueError):
        harness.generate_indices(1000, 4, 8, alpha=1.0)
    with pytest.raises(ValueError):
        harness.allocate_forward(1000, 8, 4, 8, alpha=1.0)
    assert harness.generate_indices(1000, 4, 1, alpha=1.0).shape == (4,)      # one id per sample: the recipe terminates
