"""One training step of BASELINE config 5 on one rank (shared by bench.py's `c5_train_step` leg and
benchmarks/train_step_benchmark.py): EmbeddingForward on the rank's shard, the index work of the backward
(TransposeFixedHotness + ComputeCompressedGradIndices[Blocked]), EmbeddingBackward into a compressed gradient, and
the exchange of the partial gradients of the replicated table (cuembed_amd.distributed).

Orders of the transposed COO (`order`):
  reference          fully sorted (the reference's Transpose); one gradient row per table row
  blocked            transposed in sample blocks, ComputeCompressedGradIndicesBlocked + EmbeddingBackward(sample_blocks):
                     the reference's compressed gradient (same rows, same inverse_mapping), gathered block by block
  blocked_uncoalesced transposed in sample blocks, plain remap: one gradient row per (block, table row) -- the
                     fastest backward; the sparse exchange merges rows by id anyway
"""
import time

import torch

ORDERS = ("reference", "blocked", "blocked_uncoalesced")


class TrainStep:
    def __init__(self, ce, table, indices, grad_y, batch, hotness, order="blocked_uncoalesced", sample_blocks=None,
                 dense=False):
        if order not in ORDERS:
            raise ValueError("order must be one of %s" % (ORDERS,))
        self.ce, self.table, self.idx, self.gy = ce, table, indices, grad_y
        self.B, self.H, self.W = batch, hotness, table.shape[1]
        self.rows = table.shape[0]
        self.order = order
        dev = table.device
        nnz = batch * hotness
        self.nnz = nnz
        blocks = 1
        if order != "reference":
            blocks = sample_blocks or ce.recommended_sample_blocks(table.dtype, self.W, batch, nnz)
            if order == "blocked":
                blocks = min(blocks, ce.ops.MAX_COALESCED_BLOCKS)
        self.blocks = blocks
        self.out = torch.empty((batch, self.W), dtype=table.dtype, device=dev)
        self.work = torch.empty(max(ce.transpose_workspace_bytes(nnz, torch.int32), 1), dtype=torch.uint8, device=dev)
        # compressed gradient: buffers for the largest possible number of rows, allocated once like a trainer would;
        # the row count stays on the device -- no host read-back inside the step.  One row per distinct table row,
        # or per (block, table row) for the uncoalesced order (blocks * rows, not rows, bounds that).
        per_row = blocks if order == "blocked_uncoalesced" else 1
        cap = min(nnz, per_row * self.rows)
        self.dense = torch.zeros((self.rows, self.W), dtype=table.dtype, device=dev) if dense else None
        self.comp_rows = None if dense else torch.empty((cap, self.W), dtype=table.dtype, device=dev)
        self.comp_inv = None if dense else torch.empty((cap,), dtype=torch.int32, device=dev)
        if order == "blocked":
            self.remap_work = torch.empty(max(ce.compressed_grad_blocked_workspace_bytes(nnz, torch.int32, blocks), 1),
                                          dtype=torch.uint8, device=dev)
            self.pair_rows = torch.empty((nnz,), dtype=torch.int32, device=dev)
            self.num_unique = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.count = None

    def forward(self):
        self.ce.embedding_forward(self.table, self.idx, num_hots=self.H, out=self.out)

    def index_work(self):
        ce = self.ce
        self.t_idx, self.t_sid, _ = ce.transpose_fixed_hotness(self.idx, self.B, self.H, workspace=self.work,
                                                               num_categories=self.rows, sample_blocks=self.blocks)
        if self.order == "blocked":
            self.remap, _, _ = ce.compute_compressed_grad_indices_blocked(
                self.t_idx, self.blocks, workspace=self.remap_work, num_unique=self.num_unique,
                block_row_ids=self.pair_rows)
            self.count = self.num_unique
        else:
            self.remap = ce.compute_compressed_grad_indices(self.t_idx)
            self.count = self.remap[-1:] + 1

    def backward(self):
        ce = self.ce
        if self.dense is not None:
            ce.embedding_backward(self.gy, self.rows, self.t_idx, self.t_sid, skip_grad_init=False,
                                  grad_embedding=self.dense)
        elif self.order == "blocked":
            ce.embedding_backward(self.gy, None, self.t_idx, self.t_sid, self.remap, grad_embedding=self.comp_rows,
                                  inverse_mapping=self.comp_inv, sample_blocks=self.blocks, block_row_ids=self.pair_rows)
        else:
            ce.embedding_backward(self.gy, None, self.t_idx, self.t_sid, self.remap, grad_embedding=self.comp_rows,
                                  inverse_mapping=self.comp_inv)

    def compute(self):
        self.forward()
        self.index_work()
        self.backward()

    def exchange(self, D, algorithm="auto"):
        """Sum the ranks' partial gradients.  Returns the bytes of gradient this rank contributes."""
        es = self.table.element_size()
        if self.dense is not None:
            D.allreduce_dense_grad(self.dense)
            return self.rows * self.W * es
        D.allreduce_sparse_grad(self.comp_rows, self.comp_inv, self.rows, algorithm=algorithm, num_unique=self.count,
                                coalesced=self.order != "blocked_uncoalesced")
        return None   # (the row count lives on the device; see exchanged_bytes())

    def exchange_fixed(self, D, plan):
        """The same sum through the fixed-capacity exchange (cuembed_amd.distributed.SparseGradExchange): no host
        read-back; returns the pending result (its all-gather may still be in flight)."""
        return plan.start(self.comp_rows, self.comp_inv, count=self.count,
                          coalesced=self.order != "blocked_uncoalesced")

    def calibrate_exchange(self, D, headroom=1.25, gather_group=None):
        """(warm-up, after a compute()) capacities for exchange_fixed from this step's sizes; reads them back.
        gather_group: a second process group for the all-gathers (SparseGradExchange: the next step's all-to-all then
        does not queue behind this step's pieces)."""
        return D.SparseGradExchange.calibrate(self.comp_rows, self.comp_inv, self.rows, count=self.count,
                                              coalesced=self.order != "blocked_uncoalesced", headroom=headroom,
                                              gather_group=gather_group)

    def exchanged_bytes(self):
        """(after a step) gradient bytes this rank puts on the wire: rows x (W x elem + 8-byte id) -- one read-back"""
        if self.dense is not None:
            return self.rows * self.W * self.table.element_size()
        return int(self.count.item()) * (self.W * self.table.element_size() + 8)


def timed_steps(torch_mod, fn, steps, warmup, barrier=None):
    """ms per call of fn(), one HIP-event pair around `steps` calls (after `warmup` calls and a barrier)."""
    for _ in range(warmup):
        fn()
    if barrier is not None:
        barrier()
    torch_mod.cuda.synchronize()
    a, z = torch_mod.cuda.Event(enable_timing=True), torch_mod.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a.record()
    for _ in range(steps):
        fn()
    z.record()
    z.synchronize()
    wall = (time.perf_counter() - t0) * 1e3 / steps
    return a.elapsed_time(z) / steps, wall
