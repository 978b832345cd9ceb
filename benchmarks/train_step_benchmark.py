#!/usr/bin/env python3
"""BASELINE config 5: fp16 forward + backward on a batch sharded over the GPUs of one node, table
replicated, gradient combined over RCCL (xGMI).  One process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port 29500 benchmarks/train_step_benchmark.py --exchange sparse

One step on every rank = EmbeddingForward on its 65,536-sample shard, TransposeFixedHotness (row ids
derived inside the first radix pass, index_bits from the table size; --reference_api runs
ExtractRowIdsFromFixed + Transpose over all key bits instead, i.e. only what the reference's API
offers), ComputeCompressedGradIndices, EmbeddingBackward into a compressed gradient, then the exchange:
  sparse : all-gather of the compressed rows + local merge (~293 MB per rank at this shape)
  dense  : scatter into the dense table gradient + RCCL all-reduce (5.12 GB per rank)
  none   : no exchange (upper bound / single GPU)
Rank 0 prints one JSON line with the step time (max over ranks) and its breakdown."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--rows", type=int, default=10_000_000)
    p.add_argument("--width", type=int, default=256)
    p.add_argument("--batch", type=int, default=65536, help="samples per GPU")
    p.add_argument("--hotness", type=int, default=64)
    p.add_argument("--alpha", type=float, default=1.15)
    p.add_argument("--exchange", default="sparse", choices=["sparse", "dense", "none"])
    p.add_argument("--sparse_algorithm", default="auto", choices=["auto", "allgather", "owner"])
    p.add_argument("--sample_blocks", default="auto",
                   help="transpose the batch in this many blocks of samples (extension: an uncoalesced compressed gradient, "
                        "every L2 gathers from 1 / blocks of grad_y at a time); auto = cuembed_recommended_sample_blocks, "
                        "1 = the reference's fully sorted order.  Ignored with --reference_api and with --exchange dense")
    p.add_argument("--reference_api", action="store_true",
                   help="index work through the reference's entry points only (row-id kernel + unbounded Transpose)")
    a = p.parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist
    import cuembed_amd as ce
    from cuembed_amd import distributed as D
    from cuembed_amd import harness

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ngpu = torch.cuda.device_count()
    dev = torch.device("cuda", local % ngpu)
    torch.cuda.set_device(dev)
    if world > 1 or "MASTER_ADDR" in os.environ:
        # one GPU per rank over RCCL; ranks that have to share a GPU (a rehearsal on a small box: RCCL refuses two
        # ranks on one device) exchange over gloo on host copies, which cuembed_amd.distributed does by itself
        if world > ngpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    use_dist = dist.is_initialized()
    B, H, W = a.batch, a.hotness, a.width
    table = torch.empty((a.rows, W), dtype=torch.float16, device=dev).uniform_(-1, 1)
    idx_all = harness.generate_indices(a.rows, world * B, H, alpha=a.alpha)
    idx = torch.from_numpy(np.ascontiguousarray(idx_all.reshape(world, B * H)[rank])).to(dev)
    gy = torch.from_numpy(harness.allocate_grad_y(B * W, np.float16).reshape(B, W)).to(dev)
    out = torch.empty((B, W), dtype=torch.float16, device=dev)
    nnz = B * H
    work = torch.empty(max(ce.transpose_workspace_bytes(nnz, torch.int32), 1), dtype=torch.uint8, device=dev)
    dense = torch.zeros((a.rows, W), dtype=torch.float16, device=dev) if a.exchange == "dense" else None
    # compressed gradient: buffers for the largest possible number of unique rows, allocated once like a trainer
    # would; num_unique stays on the device (remap[-1] + 1) -- no host read-back inside the step
    cap = min(nnz, a.rows)
    comp_rows = torch.empty((cap, W), dtype=torch.float16, device=dev) if a.exchange != "dense" else None
    comp_inv = torch.empty((cap,), dtype=torch.int32, device=dev) if a.exchange != "dense" else None
    blocks = 1
    if not a.reference_api and a.exchange != "dense":
        blocks = ce.recommended_sample_blocks(torch.float16, W, B, nnz) if a.sample_blocks == "auto" else int(a.sample_blocks)
    names = ["forward", "transpose", "backward", "exchange"]

    def step(ev):
        ev[0].record()
        ce.embedding_forward(table, idx, num_hots=H, out=out)
        ev[1].record()
        if a.reference_api:
            sid = ce.extract_row_ids_from_fixed(B, H, torch.int32, dev)
            t_idx, t_sid, _ = ce.transpose(sid, idx, workspace=work)
        else:
            t_idx, t_sid, _ = ce.transpose_fixed_hotness(idx, B, H, workspace=work, num_categories=a.rows,
                                                         sample_blocks=blocks)
        remap = ce.compute_compressed_grad_indices(t_idx)
        ev[2].record()
        if a.exchange == "dense":
            ce.embedding_backward(gy, a.rows, t_idx, t_sid, skip_grad_init=False, grad_embedding=dense)
            ev[3].record()
            if use_dist:
                D.allreduce_dense_grad(dense)
        else:
            rows, inv = ce.embedding_backward(gy, None, t_idx, t_sid, remap, grad_embedding=comp_rows,
                                              inverse_mapping=comp_inv)
            ev[3].record()
            if a.exchange == "sparse" and use_dist:
                D.allreduce_sparse_grad(rows, inv, a.rows, algorithm=a.sparse_algorithm, num_unique=remap[-1:] + 1,
                                        coalesced=blocks == 1)
        ev[4].record()

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step([torch.cuda.Event(enable_timing=True) for _ in range(5)])
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(a.steps)]
    sync()
    t0 = time.perf_counter()
    for s in range(a.steps):
        step(events[s])
    sync()
    wall = time.perf_counter() - t0
    parts = {n: sum(e[i].elapsed_time(e[i + 1]) for e in events) / a.steps for i, n in enumerate(names)}
    t = torch.tensor([wall], dtype=torch.float64, device=dev if not use_dist or dist.get_backend() != "gloo" else "cpu")
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        ms = float(t.item()) / a.steps * 1e3
        print(json.dumps({"workload": "fp16 fwd+bwd, %dx%d table, batch %d per GPU x %d GPUs, hotness %d, alpha %g"
                                      % (a.rows, W, B, world, H, a.alpha),
                          "exchange": a.exchange, "index_path": "reference_api" if a.reference_api else "fixed_hotness_bounded",
                          "sample_blocks": blocks, "backend": dist.get_backend() if use_dist else None,
                          "ranks_share_gpus": world > ngpu,
                          "n_gpus": world, "ms_per_step": round(ms, 4),
                          "samples_per_s": round(world * B / (ms * 1e-3)),
                          "breakdown_ms": {k: round(v, 4) for k, v in parts.items()}}), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
