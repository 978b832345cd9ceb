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
    each row), ComputeCompressedGradIndices numbers the distinct ids, and EmbeddingBackward adds
    the rows of each run (fp32 partial sums) into the compressed result.
    Returns (ids[capacity], rows[capacity, W], count) with count a 1-element device tensor: the
    number of distinct ids stays on the device (EmbeddingBackward's num_grad_embedding_rows=None
    extension), so the merge itself has no host read-back."""
    from . import ops
    m = ids.numel()
    pos = ops.extract_row_ids_for_concat(m, torch.int64, ids.device)
    t_ids, t_pos, _ = ops.transpose(pos, ids.contiguous(), num_categories=num_categories, num_rows=m)
    remap = ops.compute_compressed_grad_indices(t_ids)
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
