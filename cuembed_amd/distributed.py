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
tests (with the compute injected) and on RCCL in production.
"""
import torch


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
    import torch.distributed as dist
    return dist.all_reduce(grad_embedding, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def _merge_on_gpu(ids, vals, num_categories):
    """Sum rows that carry the same id with the library's own kernels: the merge of the gathered
    (id, row) pairs IS an embedding backward -- Transpose sorts the ids (carrying the position of
    each row), ComputeCompressedGradIndices numbers the distinct ids, and EmbeddingBackward adds
    the rows of each run (fp32 partial sums) into the compressed result."""
    from . import ops
    m = ids.numel()
    pos = ops.extract_row_ids_for_concat(m, torch.int64, ids.device)
    t_ids, t_pos, _ = ops.transpose(pos, ids.contiguous(), num_categories=num_categories)
    remap = ops.compute_compressed_grad_indices(t_ids)
    num_unique = int(remap[-1].item()) + 1
    merged, uniq = ops.embedding_backward(vals.contiguous(), num_unique, t_ids, t_pos, remap)
    return uniq, merged


def allreduce_sparse_grad(rows, inverse_mapping, num_categories, group=None):
    """Sum compressed gradients across ranks without materialising the dense table gradient.

    rows[num_unique_r, W] / inverse_mapping[num_unique_r] are this rank's compressed gradient
    (EmbeddingBackward with remapped indices).  Every rank all-gathers the (id, row) pairs --
    sizes differ per rank, so they are padded to the maximum -- and merges them locally with one
    sort + segmented sum.  Returns (unique_ids, summed_rows), identical on every rank.
    Traffic per rank is ~ G * max_r(num_unique_r) * (W * elem + 8) bytes instead of
    num_categories * W * elem for the dense all-reduce (at the north-star shape 293 MB vs 5.12 GB)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    cap = max(counts)
    width = rows.shape[1]
    pad_rows = torch.zeros((cap, width), dtype=rows.dtype, device=rows.device)
    pad_rows[: rows.shape[0]] = rows
    pad_ids = torch.full((cap,), -1, dtype=torch.int64, device=rows.device)
    pad_ids[: rows.shape[0]] = inverse_mapping.to(torch.int64)
    all_rows = [torch.empty_like(pad_rows) for _ in range(world)]
    all_ids = [torch.empty_like(pad_ids) for _ in range(world)]
    dist.all_gather(all_rows, pad_rows, group=group)
    dist.all_gather(all_ids, pad_ids, group=group)
    ids = torch.cat([all_ids[r][: counts[r]] for r in range(world)])
    vals = torch.cat([all_rows[r][: counts[r]] for r in range(world)])
    if vals.is_cuda:
        return _merge_on_gpu(ids, vals, num_categories)
    uniq, inverse = torch.unique(ids, sorted=True, return_inverse=True)
    summed = torch.zeros((uniq.numel(), width), dtype=torch.float32, device=rows.device)
    summed.index_add_(0, inverse, vals.float())
    return uniq, summed.to(rows.dtype)
