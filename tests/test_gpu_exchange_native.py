"""The device-side halves of the sparse gradient exchange (cuembed::PackRowsByOwner / FinishOwnerPiece and the
owner's merge around them, include/cuembed_amd.h "multi-GPU") against a plain numpy restatement of what
cuembed_amd/distributed.py documents: through the C ABI (cuembed_amd.ops) and through the torch ops of
libcuembed_pyt.so.  Integer-valued rows: every sum is exact in fp16 / fp32, so equality is bit-for-bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _backends():
    import cuembed_amd.cuembed_pyt  # noqa: F401  (loads libcuembed_pyt.so)
    from cuembed_amd import ops
    return {"c_abi": (ops.exchange_pack_rows, ops.exchange_merge),
            "torch_op": (torch.ops.cuembed_pyt.cuembed_exchange_pack, torch.ops.cuembed_pyt.cuembed_exchange_merge)}


def _cuts(num_categories, world):
    from cuembed_amd.distributed import owner_bounds
    return np.array([b[0] for b in owner_bounds(num_categories, world)] + [num_categories], dtype=np.int64)


def _pack_expected(ids, rows, count, cuts, slot, input_capacity, num_categories):
    n = ids.shape[0]
    k = n if count is None else max(0, min(int(count), n))
    world = cuts.shape[0] - 1
    send_ids = np.full((world * slot,), num_categories, dtype=np.int64)
    written = np.zeros((world * slot,), dtype=bool)
    send_rows = np.zeros((world * slot, rows.shape[1]), dtype=rows.dtype)
    starts = np.searchsorted(ids[:k], cuts, side="left")
    flag = int(count is not None and input_capacity > 0 and int(count) > input_capacity)
    for r in range(world):
        lo, hi = int(starts[r]), int(starts[r + 1])
        take = min(hi - lo, slot)
        flag |= int(hi - lo > slot)
        send_ids[r * slot: r * slot + take] = ids[lo: lo + take]
        send_rows[r * slot: r * slot + take] = rows[lo: lo + take]
        written[r * slot: r * slot + take] = True
    return send_ids, send_rows, written, starts.astype(np.int64), flag


@pytest.mark.parametrize("backend", ["c_abi", "torch_op"])
@pytest.mark.parametrize("index_dtype", [torch.int32, torch.int64])
@pytest.mark.parametrize("elem_dtype,width", [(torch.float16, 256), (torch.float16, 33), (torch.float32, 7),
                                              (torch.bfloat16, 64), (torch.float32, 1024), (torch.float16, 4)])
def test_pack_rows_by_owner(backend, index_dtype, elem_dtype, width):
    pack, _ = _backends()[backend]
    rng = np.random.default_rng(11 + width)
    num_categories = 100_000
    for world, n, slot, count, input_capacity in [(1, 5000, 6000, None, 0), (3, 5000, 2500, 4100, 0),
                                                  (8, 20000, 2600, 19000, 0), (8, 3000, 300, None, 0),
                                                  (5, 777, 200, 0, 0), (4, 1, 16, None, 0), (8, 9000, 1500, 9500, 9000),
                                                  (64, 4000, 90, 3999, 0), (2, 4096, 16, 4096, 0)]:
        # (ascending and distinct in the counted part; whatever behind it)
        ids = np.sort(rng.choice(num_categories, size=n, replace=False)).astype(np.int64)
        if count is not None and count < n:
            ids[count:] = rng.integers(0, num_categories, size=n - count)
        rows = rng.integers(-8, 9, size=(n, width)).astype(np.float32)
        cuts = _cuts(num_categories, world)
        d_ids = torch.from_numpy(ids).to(index_dtype).cuda()
        d_rows = torch.from_numpy(rows).to(elem_dtype).cuda()
        d_count = None if count is None else torch.tensor([count], dtype=index_dtype, device="cuda")
        d_cuts = torch.from_numpy(cuts).cuda()
        send_ids = torch.full((world * slot,), -7, dtype=torch.int64, device="cuda")
        send_rows = torch.full((world * slot, width), 3.0, dtype=elem_dtype, device="cuda")
        starts = torch.full((world + 1,), -1, dtype=torch.int64, device="cuda")
        flag = torch.zeros((1,), dtype=torch.int64, device="cuda")
        pack(d_ids, d_rows, d_count, d_cuts, slot, input_capacity, num_categories, send_ids, send_rows, starts, flag)
        torch.cuda.synchronize()
        e_ids, e_rows, written, e_starts, e_flag = _pack_expected(ids, rows, count, cuts, slot, input_capacity,
                                                                  num_categories)
        case = (world, n, slot, count, input_capacity)
        assert np.array_equal(starts.cpu().numpy(), e_starts), case
        assert np.array_equal(send_ids.cpu().numpy(), e_ids), case
        got_rows = send_rows.float().cpu().numpy()
        assert np.array_equal(got_rows[written], e_rows[written]), case
        assert np.all(got_rows[~written] == 3.0), case          # rows behind a slot's ids are not touched
        assert int(flag.item()) == e_flag, case


def test_pack_keeps_a_raised_flag_and_takes_an_empty_gradient():
    pack, _ = _backends()["torch_op"]
    world, slot, width, num_categories = 4, 8, 16, 1000
    cuts = torch.from_numpy(_cuts(num_categories, world)).cuda()
    send_ids = torch.zeros((world * slot,), dtype=torch.int64, device="cuda")
    send_rows = torch.zeros((world * slot, width), dtype=torch.float16, device="cuda")
    starts = torch.zeros((world + 1,), dtype=torch.int64, device="cuda")
    flag = torch.ones((1,), dtype=torch.int64, device="cuda")
    pack(torch.empty((0,), dtype=torch.int64, device="cuda"), torch.empty((0, width), dtype=torch.float16, device="cuda"),
         None, cuts, slot, 0, num_categories, send_ids, send_rows, starts, flag)
    assert int(flag.item()) == 1
    assert bool((send_ids == num_categories).all()) and bool((starts == 0).all())


@pytest.mark.parametrize("shift_elems,width", [(4, 256), (2, 256), (1, 256), (1, 33), (3, 8)])
def test_pack_takes_rows_at_any_alignment(shift_elems, width):
    """rows / send_rows that start 8, 4 or 2 bytes into an allocation: the copy falls back to narrower pieces."""
    pack, _ = _backends()["torch_op"]
    rng = np.random.default_rng(5)
    num_categories, world, n, slot = 10_000, 3, 2000, 900
    ids = np.sort(rng.choice(num_categories, size=n, replace=False)).astype(np.int64)
    rows = rng.integers(-8, 9, size=(n, width)).astype(np.float32)
    cuts = _cuts(num_categories, world)
    base = torch.zeros((n * width + 16,), dtype=torch.float16, device="cuda")
    d_rows = base[shift_elems: shift_elems + n * width].view(n, width)
    d_rows.copy_(torch.from_numpy(rows))
    out_base = torch.full((world * slot * width + 16,), 3.0, dtype=torch.float16, device="cuda")
    send_rows = out_base[shift_elems: shift_elems + world * slot * width].view(world * slot, width)
    send_ids = torch.zeros((world * slot,), dtype=torch.int64, device="cuda")
    starts = torch.zeros((world + 1,), dtype=torch.int64, device="cuda")
    flag = torch.zeros((1,), dtype=torch.int64, device="cuda")
    pack(torch.from_numpy(ids).cuda(), d_rows, None, torch.from_numpy(cuts).cuda(), slot, 0, num_categories, send_ids,
         send_rows, starts, flag)
    e_ids, e_rows, written, _, e_flag = _pack_expected(ids, rows, None, cuts, slot, 0, num_categories)
    assert np.array_equal(send_ids.cpu().numpy(), e_ids) and int(flag.item()) == e_flag
    got = send_rows.float().cpu().numpy()
    assert np.array_equal(got[written], e_rows[written]) and np.all(got[~written] == 3.0)
    edge = out_base.float().cpu().numpy()
    assert np.all(edge[:shift_elems] == 3.0) and np.all(edge[shift_elems + world * slot * width:] == 3.0)


def _merge_expected(ids, rows, num_categories, capacity, pad_lo, pad_len, before_ids, before_rows):
    keep = ids < num_categories
    uniq, inverse = np.unique(ids[keep], return_inverse=True)
    k = uniq.shape[0]
    if k > capacity:
        return before_ids.copy(), before_rows.copy(), k, 1
    sums = np.zeros((capacity + 1, rows.shape[1]), dtype=np.float64)
    np.add.at(sums, inverse, rows[keep].astype(np.float64))
    out_ids = pad_lo + np.arange(capacity + 1, dtype=np.int64) % pad_len
    out_ids[:k] = uniq
    return out_ids, sums.astype(np.float32), k, 0


@pytest.mark.parametrize("backend", ["c_abi", "torch_op"])
@pytest.mark.parametrize("elem_dtype,width", [(torch.float16, 256), (torch.float32, 33), (torch.bfloat16, 64)])
def test_owner_merge_fixed_capacity(backend, elem_dtype, width):
    _, merge = _backends()[backend]
    rng = np.random.default_rng(23 + width)
    num_categories = 500_000
    for n, distinct, padding, capacity, with_tail in [(6000, 1500, 700, 2000, True), (6000, 1500, 0, 1500, True),
                                                      (6000, 1500, 700, 1499, True), (300, 300, 0, 300, False),
                                                      (5000, 1, 4000, 4, True), (64, 10, 64, 16, True),
                                                      (200_000, 60_000, 30_000, 70_000, True)]:
        pool = rng.choice(num_categories, size=distinct, replace=False)
        real = n - padding if padding < n else 0
        ids = np.concatenate([pool[rng.integers(0, distinct, size=real)],
                              np.full((n - real,), num_categories, dtype=np.int64)]).astype(np.int64)
        rng.shuffle(ids)
        rows = rng.integers(-3, 4, size=(n, width)).astype(np.float32)
        pad_lo, pad_len = 1234, 777
        before_ids = rng.integers(0, num_categories, size=capacity + 1).astype(np.int64)
        before_rows = rng.integers(-2, 3, size=(capacity + 1, width)).astype(np.float32)
        out_ids = torch.from_numpy(before_ids).cuda()
        out_rows = torch.from_numpy(before_rows).to(elem_dtype).cuda()
        tail = torch.full((capacity + 2,), -5, dtype=torch.int64, device="cuda") if with_tail else None
        flag = torch.zeros((1,), dtype=torch.int64, device="cuda")
        count = torch.full((1,), -1, dtype=torch.int64, device="cuda")
        merge(torch.from_numpy(ids).cuda(), torch.from_numpy(rows).to(elem_dtype).cuda(), num_categories, pad_lo, pad_len,
              out_ids, out_rows, tail, flag, count)
        torch.cuda.synchronize()
        case = (n, distinct, padding, capacity)
        k = np.unique(ids[ids < num_categories]).shape[0]
        assert int(count.item()) == k, case
        if k > capacity:
            # nothing of this step was written; the ids keep what they held (all of them: count > capacity >= i)
            assert int(flag.item()) == 1, case
            assert np.array_equal(out_ids.cpu().numpy(), before_ids), case
            got = out_rows.float().cpu().numpy()
            spare = min(k if padding else capacity, capacity)
            keep = np.arange(capacity + 1) != spare      # (one row is zeroed whatever happened)
            assert np.array_equal(got[keep], before_rows[keep]), case
        else:
            e_ids, e_rows, _, _ = _merge_expected(ids, rows, num_categories, capacity, pad_lo, pad_len, before_ids,
                                                  before_rows)
            assert int(flag.item()) == 0, case
            assert np.array_equal(out_ids.cpu().numpy(), e_ids), case
            assert np.array_equal(out_rows.float().cpu().numpy(), e_rows), case
        if with_tail:
            t = tail.cpu().numpy()
            assert np.array_equal(t[:capacity], out_ids.cpu().numpy()[:capacity]), case
            assert t[capacity] == min(k, capacity) and t[capacity + 1] == int(k > capacity), case


def test_merge_carries_the_flag_word_into_the_tail():
    _, merge = _backends()["torch_op"]
    ids = torch.tensor([5, 3, 5, 9], dtype=torch.int64, device="cuda")
    rows = torch.ones((4, 8), dtype=torch.float32, device="cuda")
    out_ids = torch.zeros((5,), dtype=torch.int64, device="cuda")
    out_rows = torch.zeros((5, 8), dtype=torch.float32, device="cuda")
    tail = torch.zeros((6,), dtype=torch.int64, device="cuda")
    flag = torch.ones((1,), dtype=torch.int64, device="cuda")        # raised earlier in the step (a slot overflowed)
    merge(ids, rows, 10, 0, 10, out_ids, out_rows, tail, flag, None)
    assert tail.tolist() == [3, 5, 9, 3, 3, 1]
    assert out_rows[:, 0].tolist() == [1.0, 2.0, 1.0, 0.0, 0.0]
