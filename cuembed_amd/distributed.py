"""Batch-sharded data parallelism over the GPUs of one node (one process per GPU).

The reference is single-GPU (README.md:110).  The path shards by SAMPLE: a sample's pooled row
depends only on its own lookups and on the table, which every GPU holds in full (a 10M x 256
fp16 table is 5.12 GB of 288 GB).  Therefore

  forward   : rank r computes samples [r*B/G, (r+1)*B/G) -- no collective at all;
  transpose : per shard, sample ids local to the shard -- no collective;
  backward  : every rank produces a partial gradient of the replicated table; the partials are
              summed with ONE collective, an RCCL all-reduce (torch.distributed backend "nccl")
              over xGMI -- dense, or "sparse" on the compressed rows only.

Only plain torch.distributed calls are used, so the same code runs on gloo/CPU tensors in the
tests (with the compute injected) and on RCCL in production.  When the process group's backend
cannot take device tensors (gloo; e.g. several ranks sharing one GPU, where RCCL refuses to build
a communicator) device tensors are staged through host copies around each collective.
"""
import os

import torch


def _stage_on_host(t, group):
    """True when the collective has to run on a host copy of device tensor `t`."""
    import torch.distributed as dist
    return t.is_cuda and dist.get_backend(group) == "gloo"


class _CompletedWork:
    """What an async collective returns when it had to run synchronously (host-staged path)."""

    def wait(self, timeout=None):
        return True

    def is_completed(self):
        return True


def _all_reduce_sum(t, group=None, async_op=False):
    """In-place sum across ranks.  async_op=True returns a handle with wait() / is_completed(); on the
    host-staged path (gloo group, device tensor) the copies and the collective are blocking, so the
    handle is already complete and nothing overlaps."""
    import torch.distributed as dist
    if not _stage_on_host(t, group):
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    h = t.cpu()
    dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
    t.copy_(h)
    return _CompletedWork() if async_op else None


def _all_gather(t, group=None):
    """List of every rank's `t` (same shape on all ranks)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if not _stage_on_host(t, group):
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t, group=group)
        return out
    h = t.cpu()
    out = [torch.empty_like(h) for _ in range(world)]
    dist.all_gather(out, h, group=group)
    return [o.to(t.device) for o in out]


def _all_to_all_single(t, recv_rows, send_rows, group=None):
    """Rows of `t` cut by send_rows go to the ranks in order; returns the rows received."""
    import torch.distributed as dist
    src = t.cpu() if _stage_on_host(t, group) else t.contiguous()
    got = torch.empty((sum(recv_rows),) + tuple(t.shape[1:]), dtype=t.dtype, device=src.device)
    dist.all_to_all_single(got, src, output_split_sizes=recv_rows, input_split_sizes=send_rows, group=group)
    return got.to(t.device)


def shard_bounds(batch_size, rank, world):
    """Samples [lo, hi) owned by `rank`: contiguous, sizes differ by at most one."""
    base, rem = divmod(batch_size, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_fixed(indices, weights, batch_size, num_hots, rank, world):
    """Slice a fixed-hotness batch.  Returns (indices, weights, local_batch_size)."""
    lo, hi = shard_bounds(batch_size, rank, world)
    sl = slice(lo * num_hots, hi * num_hots)
    return indices[sl], (None if weights is None else weights[sl]), hi - lo


def shard_csr(offsets, indices, weights, rank, world):
    """Slice a CSR batch; offsets are rebased to start at 0.
    Returns (offsets, indices, weights, local_batch_size).  Reads two offsets on the host."""
    batch_size = offsets.numel() - 1
    lo, hi = shard_bounds(batch_size, rank, world)
    local = offsets[lo:hi + 1]
    begin, end = int(local[0]), int(local[-1])
    return (local - local[0], indices[begin:end],
            None if weights is None else weights[begin:end], hi - lo)


def allreduce_dense_grad(grad_embedding, group=None, async_op=False):
    """Sum the per-rank partial table gradients in place (RCCL all-reduce over xGMI)."""
    return _all_reduce_sum(grad_embedding, group=group, async_op=async_op)


def _merge_on_gpu(ids, vals, num_categories):
    """Sum rows that carry the same id with the library's own kernels: the merge of the gathered
    (id, row) pairs IS an embedding backward -- Transpose sorts the ids (carrying the position of
    each row; TransposeFixedHotness with hotness 1), ComputeCompressedGradIndices numbers the distinct ids (from the same call), and EmbeddingBackward adds
    the rows of each run (fp32 partial sums) into the compressed result.
    Returns (ids[capacity], rows[capacity, W], count) with count a 1-element device tensor: the
    number of distinct ids stays on the device (EmbeddingBackward's num_grad_embedding_rows=None
    extension), so the merge itself has no host read-back."""
    from . import ops
    m = ids.numel()
    # (hotness 1: the position of each row is generated by the first sorting pass, the compressed ids by the last)
    t_ids, t_pos, _, remap = ops.transpose_fixed_hotness(ids.contiguous(), m, 1, num_categories=num_categories,
                                                         remapped=True)
    cap = min(m, num_categories)
    merged = torch.empty((cap, vals.shape[1]), dtype=vals.dtype, device=vals.device)
    uniq = torch.empty((cap,), dtype=torch.int64, device=vals.device)
    ops.embedding_backward(vals.contiguous(), None, t_ids, t_pos, remap, grad_embedding=merged, inverse_mapping=uniq)
    return uniq, merged, remap[-1:] + 1


def _merge(ids, vals, num_categories):
    """Sum rows with equal id; returns (ascending unique ids, summed rows, count) -- on the GPU padded to a
    capacity with `count` a 1-element device tensor, on the CPU exact with count = None."""
    if ids.numel() == 0:                 # nothing was gathered (every rank's compressed gradient was empty)
        return ids, vals, None
    if vals.is_cuda:
        return _merge_on_gpu(ids, vals, num_categories)
    uniq, inverse = torch.unique(ids, sorted=True, return_inverse=True)
    summed = torch.zeros((uniq.numel(), vals.shape[1]), dtype=torch.float32, device=vals.device)
    summed.index_add_(0, inverse, vals.float())
    return uniq, summed.to(vals.dtype), None


def _gather_ragged(ids, vals, group, count=None):
    """all-gather of per-rank (ids[n_r], vals[n_r, W]) with different n_r (padded to the maximum).
    `count` (1-element tensor on the tensors' device, or None = all rows) is the number of valid
    leading rows of this rank.  Returns the concatenation in rank order.  ONE host read-back: all
    ranks' counts."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n = torch.tensor([ids.shape[0]], dtype=torch.int64, device=ids.device) if count is None \
        else count.reshape(1).to(torch.int64)
    counts = torch.cat(_all_gather(n, group)).tolist()          # one host read-back for all ranks' counts
    cap = max(max(counts), 1)
    have = min(cap, vals.shape[0])
    pad_vals = torch.zeros((cap, vals.shape[1]), dtype=vals.dtype, device=vals.device)
    pad_vals[:have] = vals[:have]
    pad_ids = torch.full((cap,), -1, dtype=torch.int64, device=ids.device)
    pad_ids[:have] = ids[:have]
    all_vals = _all_gather(pad_vals, group)
    all_ids = _all_gather(pad_ids, group)
    return (torch.cat([all_ids[r][: counts[r]] for r in range(world)]),
            torch.cat([all_vals[r][: counts[r]] for r in range(world)]))


def owner_bounds(num_categories, world):
    """Row-id range [lo, hi) owned by each rank in the owner-partitioned exchange."""
    return [shard_bounds(num_categories, r, world) for r in range(world)]


def allreduce_sparse_grad(rows, inverse_mapping, num_categories, group=None, algorithm="auto", num_unique=None,
                          coalesced=True):
    """Sum compressed gradients across ranks without materialising the dense table gradient.

    rows[num_unique_r, W] / inverse_mapping[num_unique_r] are this rank's compressed gradient
    (EmbeddingBackward with remapped indices; ids ascending).  `num_unique` (optional, a 1-element
    device tensor, e.g. remap[-1:] + 1): only that many leading rows are valid -- for buffers that
    were sized for the worst case because the count never left the device.  `coalesced=False`: the ids
    are not ascending / not unique -- the uncoalesced gradient of a batch that was transposed in sample
    blocks (ops.transpose(..., sample_blocks=)): the owner-partitioned exchange then merges the rank's own
    rows first (one more Transpose + EmbeddingBackward over its rows; it also sends fewer rows), the
    all-gather exchange takes them as they are.  (CUEMBED_DEBUG_CHECKS=1 makes the owner algorithm verify
    that ids passed as coalesced really ascend.)  Returns (unique_ids, summed_rows), identical on every
    rank.  Two algorithms:

      "allgather": every rank all-gathers all (id, row) pairs and merges them locally with one
                   sort + segmented sum.  Per rank ~ G * n * (W * elem + 8) bytes come in and
                   G * n pairs are merged (n = rows per rank).
      "owner"    : the row-id space is cut into G ranges; each rank sends the pairs of range r to
                   rank r (one all-to-all over xGMI's point-to-point links: n pairs out, ~n in),
                   the owner merges its range -- 1/G of the work -- and the merged pieces, which no
                   longer contain duplicates, are all-gathered.  Less traffic and G x less merge
                   work whenever ranks share rows (power-law batches: the hot rows are in every
                   rank's gradient).
      "auto"     : "owner" for more than 2 ranks, else "allgather".

    Either way the traffic is a few hundred MB per rank instead of num_categories * W * elem for
    the dense all-reduce (at the north-star shape 293 MB vs 5.12 GB of gradient per rank).
    Host read-backs: two per call (the ranks' row counts before the exchange of rows; the size of the
    result), whatever the algorithm -- the sizes of the tensors exchanged and returned are host values
    (three for coalesced=False with the owner algorithm and a device-side count)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if algorithm == "auto":
        algorithm = "owner" if world > 2 else "allgather"
    if algorithm not in ("allgather", "owner"):
        raise ValueError("algorithm must be 'auto', 'allgather' or 'owner'")
    ids = inverse_mapping.to(torch.int64)
    if not coalesced and algorithm == "owner":
        # the owner ranges are cut out of ASCENDING ids: coalesce this rank's rows first (the ids of a sample-block
        # order ascend only inside a block, and a row may be there once per block).  A device-side count is read
        # back here (one more read-back than for a coalesced gradient): merging a worst-case buffer would cost far more
        if num_unique is not None:
            k = int(num_unique.item())
            ids, rows = ids[:k], rows[:k]
        ids, rows, num_unique = _merge(ids, rows, num_categories)
    if algorithm == "allgather":
        all_ids, all_vals = _gather_ragged(ids, rows, group, count=num_unique)     # read-back 1: the ranks' counts
        uniq, summed, count = _merge(all_ids, all_vals, num_categories)
        if count is None:
            return uniq, summed
        k = int(count.item())                                                        # read-back 2: size of the result
        return uniq[:k], summed[:k]

    # ---- owner-partitioned: all-to-all, merge my range, all-gather the merged pieces ----
    bounds = owner_bounds(num_categories, world)
    cuts = torch.tensor([b[0] for b in bounds] + [num_categories], dtype=torch.int64, device=ids.device)
    if num_unique is not None:       # rows past the count hold nothing: give them an id beyond every range
        valid = torch.arange(ids.numel(), device=ids.device) < num_unique.reshape(1).to(ids.device)
        ids = torch.where(valid, ids, torch.full_like(ids, num_categories))
    if os.environ.get("CUEMBED_DEBUG_CHECKS") == "1" and ids.numel() > 1:
        # the owner ranges are cut out of ASCENDING ids; an uncoalesced gradient (sample blocks) passed with the
        # default coalesced=True would be split wrongly without any error (one host read-back, debug only)
        if not bool((ids[1:] >= ids[:-1]).all().item()):
            raise ValueError("allreduce_sparse_grad(algorithm='owner'): ids are not ascending; pass coalesced=False "
                             "for the gradient of a batch that was transposed in sample blocks")
    pos = torch.searchsorted(ids, cuts)                      # ids ascend: range r = [pos[r], pos[r+1])
    send = (pos[1:] - pos[:-1]).to(torch.int64)
    # every rank learns the whole world x world split matrix with ONE collective and ONE host read-back
    splits = torch.stack(_all_gather(send, group)).tolist()                          # read-back 1
    rank = dist.get_rank(group)
    send_l, recv_l = splits[rank], [splits[r][rank] for r in range(world)]
    total = sum(send_l)
    got_ids = _all_to_all_single(ids[:total], recv_l, send_l, group)
    got_vals = _all_to_all_single(rows[:total], recv_l, send_l, group)
    mine_ids, mine_vals, count = _merge(got_ids, got_vals, num_categories)
    # owners hold disjoint, ascending id ranges in rank order: the concatenation is already sorted
    return _gather_ragged(mine_ids.to(torch.int64), mine_vals, group, count=count)   # read-back 2


# ---- fixed-capacity sparse exchange: no host read-back in the steady state ---------------------------------------
# allreduce_sparse_grad() above sizes every tensor it exchanges and returns exactly, and sizes are host values: two to
# three read-backs per step, each a stall of the whole pipeline.  A training loop exchanges gradients of (nearly) the
# same size every step, so the sizes can be fixed ONCE -- from a warm-up step that is allowed to look (calibrate()) --
# and every later step runs with device-side counts only: fixed slots in the all-to-all, a fixed-capacity merge on the
# owner (EmbeddingBackward's device-side row count + pad_to_capacity), a fixed-size all-gather that can run behind the
# next step's forward (async_op), and a sticky overflow word instead of a read-back, like the library's own
# capacity_overflow.  The result is a valid uncoalesced COO gradient (zero rows fill the slack), identical on every
# rank; compact() turns it into allreduce_sparse_grad()'s (ids, rows) for whoever wants to pay for the read-back.

XGMI_LINKS_PER_GPU = 7          # MI355X: 7 xGMI links per GPU, fully connected inside a node of 8
XGMI_LINK_GBPS = 153.0          # ~ 153 GB/s per link and direction


def exchange_model_ms(world, rows_per_rank, merged_rows_total, width, elem_size, pair_capacity=None,
                      owner_capacity=None, link_GBps=XGMI_LINK_GBPS, id_bytes=8):
    """Link arithmetic of the owner-partitioned sparse exchange on a fully connected xGMI node: every pair of GPUs has
    its own link, so the all-to-all moves each (source, owner) slot over that pair's link, all pairs at once, and the
    all-gather of the owners' merged pieces brings (world - 1) pieces into every GPU, one per link.  Bytes that cross
    ONE link and direction / link rate = the floor of each phase; RCCL's protocol overhead, the merge kernels
    (Transpose + EmbeddingBackward over the received rows) and launch gaps come on top.  pair_capacity /
    owner_capacity: the fixed slot sizes of SparseGradExchange (default: the exact averages, i.e. the exact exchange).
    Returns ms per phase for the peak link rate and for 75 % of it (what large RCCL point-to-point transfers reach)."""
    if world <= 1:
        return {"all_to_all_ms": 0.0, "all_gather_ms": 0.0, "total_ms": 0.0, "total_ms_at_75pct_link": 0.0,
                "bytes_in_per_rank": 0}
    row = width * elem_size + id_bytes
    pair = float(pair_capacity) if pair_capacity else rows_per_rank / world
    piece = float(owner_capacity) if owner_capacity else merged_rows_total / world
    a2a = pair * row / (link_GBps * 1e9) * 1e3
    gather = piece * row / (link_GBps * 1e9) * 1e3
    return {"all_to_all_ms": round(a2a, 4), "all_gather_ms": round(gather, 4), "total_ms": round(a2a + gather, 4),
            "total_ms_at_75pct_link": round((a2a + gather) / 0.75, 4),
            "bytes_in_per_rank": int((world - 1) * (pair + piece) * row),
            "links_used": min(world - 1, XGMI_LINKS_PER_GPU), "link_GBps": link_GBps}


def _exchange_ops():
    """The native halves of the exchange for GPU tensors: the torch ops of libcuembed_pyt.so (least host time), or --
    CUEMBED_PYT_BACKEND=python -- the same C entry points through ctypes (cuembed_amd.ops)."""
    global _EXCHANGE_OPS
    if _EXCHANGE_OPS is None:
        from . import cuembed_pyt, ops

        class _Ops:
            pass
        o = _Ops()
        if cuembed_pyt.BACKEND == "native":
            o.pack = torch.ops.cuembed_pyt.cuembed_exchange_pack
            o.merge = torch.ops.cuembed_pyt.cuembed_exchange_merge
        else:
            o.pack, o.merge = ops.exchange_pack_rows, ops.exchange_merge
        _EXCHANGE_OPS = o
    return _EXCHANGE_OPS


_EXCHANGE_OPS = None


def _merge_fixed(ids, vals, num_categories, capacity, pad_lo, pad_len, out_ids=None, out_rows=None, tail=None,
                 flag=None):
    """Sum rows with equal id into buffers of a FIXED capacity, without a host read-back on the GPU.

    ids >= num_categories are padding of the caller's fixed-size input and are dropped.  Returns (uniq[capacity + 1],
    rows[capacity + 1, W], count[1], overflow[1] -- `flag` itself when one was given): the first `count` entries are the ascending distinct ids and their
    sums (the same bits as _merge gives), every entry past them a ZERO row whose id names a row of
    [pad_lo, pad_lo + pad_len) in turn -- harmless to whoever scatter-adds the lot.  overflow != 0: more than
    `capacity` distinct ids; the buffers then keep what they held (well-formed, but not this step's gradient).
    tail (int64[capacity + 2], optional): the all-gather's id buffer -- the ids, min(count, capacity), the flag word;
    flag (int64[1], optional): the step's overflow word so far, returned with this merge's overflow OR-ed in."""
    m = ids.numel()
    dev = vals.device
    width = vals.shape[1]
    if out_rows is None:
        out_rows = torch.zeros((capacity + 1, width), dtype=vals.dtype, device=dev)
        out_ids = torch.zeros((capacity + 1,), dtype=torch.int64, device=dev)
    if m and vals.is_cuda:
        # (the padding id num_categories sorts behind every real id and becomes ONE extra run at the end; capacity + 1
        # rows: that run needs a row too; too many distinct ids -> the kernels write nothing.  One native op: Transpose
        # with the remap from the same call, EmbeddingBackward with a device-side count, cuembed::FinishOwnerPiece)
        count = torch.empty((1,), dtype=torch.int64, device=dev)
        flag = torch.zeros((1,), dtype=torch.int64, device=dev) if flag is None else flag
        _exchange_ops().merge(ids.contiguous(), vals.contiguous(), num_categories, pad_lo, max(pad_len, 1), out_ids,
                              out_rows, tail, flag, count)
        return out_ids, out_rows, count, flag
    # ---- host tensors (the gloo tests), or nothing to merge: the same result from tensor operations
    pad_ids = pad_lo + torch.arange(capacity + 1, dtype=torch.int64, device=dev) % max(pad_len, 1)
    count = torch.zeros((1,), dtype=torch.int64, device=dev)
    overflow = torch.zeros((1,), dtype=torch.int64, device=dev)
    if m == 0:
        out_rows.zero_()
        out_ids.copy_(pad_ids)
    else:
        keep = ids < num_categories
        uniq, inverse = torch.unique(ids[keep], sorted=True, return_inverse=True)
        k = uniq.numel()
        count.fill_(k)
        if k > capacity:
            overflow.fill_(1)
        else:
            summed = torch.zeros((k, width), dtype=torch.float32, device=dev)
            summed.index_add_(0, inverse, vals[keep].float())
            out_rows.zero_()
            out_rows[:k] = summed.to(vals.dtype)
            out_ids.copy_(pad_ids)
            out_ids[:k] = uniq
    if flag is not None:
        flag |= overflow
        overflow = flag
    if tail is not None:
        tail[:capacity] = out_ids[:capacity]
        tail[capacity: capacity + 1] = torch.clamp(count, max=capacity)
        tail[capacity + 1:] = overflow
    return out_ids, out_rows, count, overflow


class SparseGradResult:
    """What SparseGradExchange.start() returns: wait() -> (ids[world * piece], rows[world * piece, W], counts[world]),
    identical on every rank.  Piece r (entries [r * piece, (r + 1) * piece)) is owner r's merged id range: `counts[r]`
    ascending distinct ids with their summed rows, then zero rows with valid ids.  The whole is a valid uncoalesced
    COO gradient of the table (scatter-add it as it is); SparseGradExchange.compact() gives the exact rows.
    The tensors are views of the exchange's two result buffers, used in turn: they stay as they are until the start()
    after the next one."""

    def __init__(self, works, ids, rows, piece, world):
        self._works, self._ids, self._rows, self._piece, self._world = works, ids, rows, piece, world

    def is_completed(self):
        return all(w.is_completed() for w in self._works)

    def wait(self):
        for w in self._works:
            w.wait()          # (RCCL: the current stream waits for the collective; the host does not)
        tail = self._ids.view(self._world, self._piece + 2)
        return (tail[:, : self._piece].reshape(-1), self._rows, tail[:, self._piece])

    def flags(self):
        """(after wait()) every rank's overflow word of this step, a device tensor [world]."""
        return self._ids.view(self._world, self._piece + 2)[:, self._piece + 1]


class SparseGradExchange:
    """Owner-partitioned sparse gradient exchange with fixed capacities (see the section comment above).

    pair_capacity : rows one rank may send to one owner per step (a slot of the all-to-all);
    piece_capacity: distinct rows one owner may hold after merging (a piece of the all-gather);
    local_capacity: distinct rows of this rank's own gradient after its local merge (coalesced=False only);
    input_capacity: rows of the caller's buffers that are looked at when a device-side `count` is given (the valid rows
                    are a prefix; buffers sized for the worst case are cut there).
    gather_group  : (optional) a second process group over the same ranks for the all-gathers: a group runs its
                    collectives in issue order, so with one group the next step's all-to-all queues behind this step's
                    all-gather; with two, all-to-all + merge of step i + 1 overlap the pieces of step i on the links.
    Sized by hand or by calibrate() from a warm-up step.  A step that does not fit raises the sticky overflow word on
    EVERY rank (it travels with the all-gather) and delivers a well-formed but incomplete gradient: look at
    overflowed() whenever a host wait is affordable (every few hundred steps, or at the step where the loss is read
    anyway) and calibrate again.

        ex = SparseGradExchange.calibrate(rows, ids, num_categories, count=count)      # warm-up: reads sizes back
        for batch in loader:
            ... forward / Transpose / EmbeddingBackward into (rows, ids, count) ...
            pending = ex.start(rows, ids, count)          # all-to-all + merge enqueued, all-gather in flight
            ... next batch's forward ...
            ids_all, rows_all, counts = pending.wait()    # stream-side wait
            table.index_add_(0, ids_all, rows_all, alpha=-lr)
    """

    def __init__(self, num_categories, width, dtype, device, pair_capacity, piece_capacity, local_capacity=0,
                 group=None, input_capacity=0, gather_group=None):
        import torch.distributed as dist
        self.group = group
        # A process group executes its collectives in issue order, so with ONE group the all-to-all of step i + 1 waits
        # for the all-gather of step i (the longest transfer of the step) although nothing in it depends on that.
        # gather_group: a second group over the same ranks (dist.new_group()) for the all-gathers -- the all-to-all and
        # the owner's merge of the next step then run while the pieces of this one are still on the links.
        self.gather_group = group if gather_group is None else gather_group
        if dist.get_world_size(self.gather_group) != dist.get_world_size(group):
            raise ValueError("gather_group must span the same ranks as group")
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.num_categories, self.width, self.dtype, self.device = int(num_categories), int(width), dtype, device
        self.pair_capacity, self.piece_capacity = int(pair_capacity), int(piece_capacity)
        self.local_capacity = int(local_capacity)
        # rows of the caller's (worst-case sized) buffers that are looked at: the valid rows are a prefix, and sorting
        # and merging millions of rows past the count would cost more than the exchange (0 = all of them)
        self.input_capacity = int(input_capacity)
        if self.pair_capacity < 1 or self.piece_capacity < 1:
            raise ValueError("capacities must be at least one row")
        bounds = owner_bounds(self.num_categories, self.world)
        self._cuts = torch.tensor([b[0] for b in bounds] + [self.num_categories], dtype=torch.int64, device=device)
        self._lo, hi = bounds[self.rank]
        self._range = max(hi - self._lo, 1)
        w, s, p = self.world, self.pair_capacity, self.piece_capacity
        on_gpu = torch.device(device).type == "cuda"
        self._slot = None if on_gpu else torch.arange(s, dtype=torch.int64, device=device)
        self._send_ids = torch.empty((w * s,), dtype=torch.int64, device=device)
        # (GPU: the pack kernel's own buffers; zero-initialised ONCE -- rows behind a slot's ids are never written and
        # must stay finite, their padding id drops them at the owner)
        self._send_rows = torch.zeros((w * s, self.width), dtype=dtype, device=device) if on_gpu else None
        self._starts = torch.zeros((w + 1,), dtype=torch.int64, device=device)
        self._flag = torch.zeros((1,), dtype=torch.int64, device=device)       # this step's overflow word
        self._recv_ids = torch.empty((w * s,), dtype=torch.int64, device=device)
        self._recv_rows = torch.empty((w * s, self.width), dtype=dtype, device=device)
        # (zero-initialised ONCE: a step that overflows leaves them as they were -- finite rows, valid ids)
        # two sets of piece buffers, used in turn like the result buffers: the all-gather of step i may still read one
        # while the merge of step i + 1 fills the other (rows, ids, [ids | count | overflow word])
        self._piece = [(torch.zeros((p + 1, self.width), dtype=dtype, device=device),
                        torch.zeros((p + 1,), dtype=torch.int64, device=device),
                        torch.zeros((p + 2,), dtype=torch.int64, device=device)) for _ in range(2)]
        self._local_rows = self._local_ids = None
        if self.local_capacity:
            self._local_rows = torch.zeros((self.local_capacity + 1, self.width), dtype=dtype, device=device)
            self._local_ids = torch.zeros((self.local_capacity + 1,), dtype=torch.int64, device=device)
        # two sets of result buffers: the all-gather of step i may still be read while step i + 1 gathers
        self._out = [(torch.zeros((w * (p + 2),), dtype=torch.int64, device=device),
                      torch.zeros((w * p, self.width), dtype=dtype, device=device)) for _ in range(2)]
        self._turn = 0
        self._pending = [None, None]   # per set of piece buffers: the result whose all-gather may still read it
        self._overflow = torch.zeros((1,), dtype=torch.int64, device=device)      # sticky, all ranks' words OR-ed

    # -- sizes ----------------------------------------------------------------------------------------------------
    @classmethod
    def calibrate(cls, rows, inverse_mapping, num_categories, count=None, coalesced=True, group=None, headroom=1.25,
                  gather_group=None):
        """Warm-up: looks at THIS step's sizes on the host (read-backs, an exchange of the ids alone) and returns an
        exchange whose capacities are `headroom` x the largest slot / piece / local gradient any rank needs for it."""
        import torch.distributed as dist
        world = dist.get_world_size(group)
        ids = inverse_mapping.to(torch.int64)
        k = ids.numel() if count is None else int(count.reshape(-1)[0].item())
        ids = ids[:k]
        local = 0
        if not coalesced:
            ids = torch.unique(ids)          # (what the rank's own merge will leave: ascending distinct ids)
            local = ids.numel()
        bounds = owner_bounds(num_categories, world)
        cuts = torch.tensor([b[0] for b in bounds] + [num_categories], dtype=torch.int64, device=ids.device)
        pos = torch.searchsorted(ids, cuts)
        pair = int((pos[1:] - pos[:-1]).max().item()) if ids.numel() else 0
        merged = torch.unique(torch.cat(_all_gather_ragged_ids(ids, group)))      # the owners' merged rows, all ranges
        got = torch.searchsorted(merged, cuts)
        piece = int((got[1:] - got[:-1]).max().item()) if merged.numel() else 0
        need = torch.tensor([pair, piece, local, k], dtype=torch.int64,
                            device=rows.device if dist.get_backend(group) == "nccl" else "cpu")
        dist.all_reduce(need, op=dist.ReduceOp.MAX, group=group)
        pair, piece, local, given = (int(x) for x in need.tolist())

        def grow(n):
            return max(int(n * headroom) + 16, 16)
        return cls(num_categories, rows.shape[1], rows.dtype, rows.device, grow(pair), grow(piece),
                   0 if coalesced else grow(local), group=group, input_capacity=grow(given), gather_group=gather_group)

    def overflowed(self, reset=False):
        """True when a step since the last reset did not fit the capacities on SOME rank (one host read-back)."""
        flag = bool(self._overflow.item())
        if reset:
            self._overflow.zero_()
        return flag

    def wire_bytes_per_step(self):
        """Bytes that enter this rank per step: (world - 1) slots of the all-to-all + (world - 1) pieces."""
        row = self.width * torch.empty((), dtype=self.dtype).element_size() + 8
        return (self.world - 1) * (self.pair_capacity * row + self.piece_capacity * row + 16)

    # -- the step -------------------------------------------------------------------------------------------------
    def start(self, rows, inverse_mapping, count=None, coalesced=True, async_op=True):
        """Enqueues the whole exchange of this rank's compressed gradient -- rows[n, W], inverse_mapping[n], of which
        the first `count` (1-element device tensor; None = all) are valid and, for coalesced=True, ascending -- and
        returns a SparseGradResult.  Nothing in here waits for the device."""
        import torch.distributed as dist
        dev, w, s, p = self.device, self.world, self.pair_capacity, self.piece_capacity
        turn = self._turn
        flag = self._flag
        flag.zero_()
        check = 0            # (input_capacity when the buffers are cut there: the pack then flags a count beyond it)
        if self.input_capacity and count is not None and inverse_mapping.numel() > self.input_capacity:
            check = self.input_capacity
            inverse_mapping, rows = inverse_mapping[: self.input_capacity], rows[: self.input_capacity]
        n = inverse_mapping.numel()
        native = rows.is_cuda
        if native:
            ids = inverse_mapping
            if n and not coalesced:
                # the owner ranges are cut out of ASCENDING ids: the rank's own rows are merged first (one more native
                # merge; it also sends fewer rows).  Afterwards `count` is the merged row count, on the device.
                if not self.local_capacity:
                    raise ValueError("coalesced=False needs local_capacity (the rank's own rows are merged first)")
                ids = ids.to(torch.int64)
                if count is not None:      # rows past the count hold nothing: give them the padding id
                    given = count.reshape(1).to(dev)
                    ids = torch.where(torch.arange(n, device=dev) < given, ids,
                                      torch.full((1,), self.num_categories, dtype=torch.int64, device=dev))
                    if check:
                        flag |= (given > check).to(torch.int64)
                        check = 0
                ids, rows, count, _ = _merge_fixed(ids, rows, self.num_categories, self.local_capacity,
                                                   self.num_categories, 1, self._local_ids, self._local_rows, None, flag)
            # ---- pack: owner r's rows into slot r of the send buffers (two launches, device-side counts)
            _exchange_ops().pack(ids.contiguous(), rows.contiguous(), None if count is None else count.reshape(-1),
                                 self._cuts, s, check, self.num_categories, self._send_ids, self._send_rows,
                                 self._starts, flag)
            send_rows = self._send_rows
        else:
            send_rows = self._pack_on_host(inverse_mapping, rows, count, coalesced, check, flag)
        # ---- all-to-all with equal splits: no sizes to agree on
        _all_to_all_equal(self._recv_ids, self._send_ids, self.group)
        _all_to_all_equal(self._recv_rows, send_rows, self.group)
        # ---- merge my range into the piece (fixed capacity, device-side count; the piece's id buffer carries the
        #      count and the overflow word in-band)
        if self._pending[turn] is not None:
            # the all-gather of two steps ago read the piece buffers this merge is about to rewrite: order behind it (a
            # stream-side wait; a caller that already waited pays nothing).  Only the merge: the pack and the all-to-all
            # above touch neither the pieces nor the results.
            self._pending[turn].wait()
            self._pending[turn] = None
        piece_rows, piece_ids, piece_tail = self._piece[turn]
        _merge_fixed(self._recv_ids, self._recv_rows, self.num_categories, p, self._lo, self._range, piece_ids,
                     piece_rows, piece_tail, flag)
        # ---- all-gather of the fixed-size pieces (ids carry count and overflow word in-band)
        out_tail, out_rows = self._out[turn]
        self._turn ^= 1
        works = [_all_gather_into(out_tail, piece_tail, self.gather_group, async_op),
                 _all_gather_into(out_rows, piece_rows[:p], self.gather_group, async_op)]
        result = SparseGradResult([x for x in works if x is not None], out_tail, out_rows, p, w)
        if not async_op:
            self.note_flags(result)
        else:
            self._pending[turn] = result
        return result

    def _pack_on_host(self, inverse_mapping, rows, count, coalesced, check, flag):
        """The pack of start() from tensor operations: host tensors (the gloo tests).  Same send buffers as
        cuembed::PackRowsByOwner, except that the rows of unused slot entries are row 0 again and again instead of
        whatever the buffer held (their id drops them either way)."""
        dev, w, s = self.device, self.world, self.pair_capacity
        ids = inverse_mapping if inverse_mapping.dtype == torch.int64 else inverse_mapping.to(torch.int64)
        n = ids.numel()
        if check:
            flag |= (count.reshape(1).to(dev) > check).to(torch.int64)
        if n and count is not None:      # rows past the count hold nothing: give them the padding id
            ids = torch.where(torch.arange(n, device=dev) < count.reshape(1).to(dev), ids,
                              torch.full_like(ids, self.num_categories))
        if n and not coalesced:
            if not self.local_capacity:
                raise ValueError("coalesced=False needs local_capacity (the rank's own rows are merged first)")
            ids, rows, mine, _ = _merge_fixed(ids, rows, self.num_categories, self.local_capacity, 0, 1,
                                              self._local_ids, self._local_rows, None, flag)
            # (behind the count the merged buffer holds zero rows named 0: give them the padding id again)
            ids = torch.where(torch.arange(ids.numel(), device=dev) < mine, ids, torch.full_like(ids, self.num_categories))
            n = ids.numel()
        if not n:
            self._send_ids.fill_(self.num_categories)
            return torch.zeros((w * s, self.width), dtype=self.dtype, device=dev)
        pos = torch.searchsorted(ids, self._cuts)                       # padding ids sort behind the last cut
        have = pos[1:] - pos[:-1]
        flag |= (have > s).any().reshape(1).to(torch.int64)
        valid = self._slot.unsqueeze(0) < have.unsqueeze(1)             # [world, slot]
        src = torch.where(valid, pos[:-1].unsqueeze(1) + self._slot.unsqueeze(0), torch.zeros_like(valid, dtype=torch.int64))
        src = src.reshape(-1)
        torch.where(valid.reshape(-1), ids[src], torch.full_like(src, self.num_categories), out=self._send_ids)
        return rows.index_select(0, src)

    def note_flags(self, result):
        """(after result.wait()) folds the step's overflow words of all ranks into the sticky one; device-side."""
        self._overflow |= result.flags().max().reshape(1)

    @staticmethod
    def compact(ids, rows, counts):
        """The exact (ascending unique ids, summed rows) of a waited result -- allreduce_sparse_grad()'s return value.
        Reads the owners' counts back (one host wait)."""
        world = counts.numel()
        piece = ids.numel() // world
        ks = counts.tolist()
        return (torch.cat([ids[r * piece: r * piece + ks[r]] for r in range(world)]),
                torch.cat([rows[r * piece: r * piece + ks[r]] for r in range(world)]))


def _all_gather_ragged_ids(ids, group):
    """(calibration only) every rank's id list, through a padded all-gather; reads the lengths back."""
    import torch.distributed as dist
    n = torch.tensor([ids.numel()], dtype=torch.int64, device=ids.device)
    counts = torch.cat(_all_gather(n, group)).tolist()
    cap = max(max(counts), 1)
    pad = torch.full((cap,), -1, dtype=torch.int64, device=ids.device)
    pad[: ids.numel()] = ids
    got = _all_gather(pad, group)
    return [got[r][: counts[r]] for r in range(dist.get_world_size(group))]


def _all_to_all_equal(out, src, group):
    """all_to_all_single with equal splits (no split sizes: nothing to read back or agree on)."""
    import torch.distributed as dist
    if not _stage_on_host(src, group):
        dist.all_to_all_single(out, src.contiguous(), group=group)
        return
    h_out = torch.empty(out.shape, dtype=out.dtype)
    dist.all_to_all_single(h_out, src.cpu(), group=group)
    out.copy_(h_out)


def _all_gather_into(out, piece, group, async_op):
    """out[world * n, ...] = every rank's piece[n, ...]; a work handle when async_op (None when it ran blocking)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if _stage_on_host(piece, group):
        h = piece.cpu()
        got = [torch.empty_like(h) for _ in range(world)]
        dist.all_gather(got, h, group=group)
        out.copy_(torch.cat(got))
        return None
    if dist.get_backend(group) == "nccl":
        return dist.all_gather_into_tensor(out, piece.contiguous(), group=group, async_op=async_op) if async_op \
            else dist.all_gather_into_tensor(out, piece.contiguous(), group=group)
    got = list(out.chunk(world))
    return dist.all_gather(got, piece.contiguous(), group=group, async_op=async_op) if async_op \
        else dist.all_gather(got, piece.contiguous(), group=group)
