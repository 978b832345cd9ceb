#!/usr/bin/env python3
"""BASELINE config 5: fp16 forward + backward on a batch sharded over the GPUs of one node, table
replicated, gradient combined over RCCL (xGMI).  One process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port 29500 benchmarks/train_step_benchmark.py --exchange sparse

One step on every rank (benchmarks/c5_train_step.py, shared with bench.py's c5_train_step leg) =
EmbeddingForward on its 65,536-sample shard, TransposeFixedHotness (row ids derived inside the first radix
pass, index_bits from the table size; --reference_api runs ExtractRowIdsFromFixed + Transpose over all key bits
instead, i.e. only what the reference's API offers), ComputeCompressedGradIndices[Blocked], EmbeddingBackward
into a compressed gradient, then the exchange:
  sparse : all-gather of the compressed rows + local merge (~293 MB per rank at this shape)
  dense  : scatter into the dense table gradient + RCCL all-reduce (5.12 GB per rank)
  sparse_fixed: the same sum through SparseGradExchange -- capacities fixed by one warm-up step, then no host read-back
           per step and the all-gather of step i in flight behind the compute of step i + 1
  none   : no exchange (upper bound / single GPU)
Rank 0 prints one JSON line with the step time (max over ranks) and its breakdown."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--rows", type=int, default=10_000_000)
    p.add_argument("--width", type=int, default=256)
    p.add_argument("--batch", type=int, default=65536, help="samples per GPU")
    p.add_argument("--hotness", type=int, default=64)
    p.add_argument("--alpha", type=float, default=1.15)
    p.add_argument("--exchange", default="sparse", choices=["sparse", "sparse_fixed", "dense", "none"],
                   help="sparse = exact sizes (host read-backs per step); sparse_fixed = SparseGradExchange: capacities "
                        "from one warm-up step, no host read-back per step, the all-gather of step i behind the compute "
                        "of step i + 1")
    p.add_argument("--sparse_algorithm", default="auto", choices=["auto", "allgather", "owner"])
    p.add_argument("--order", default="blocked_uncoalesced", choices=["reference", "blocked", "blocked_uncoalesced"],
                   help="order of the transposed COO (benchmarks/c5_train_step.py): reference = fully sorted; blocked = "
                        "sample blocks + the reference's compressed gradient (ComputeCompressedGradIndicesBlocked); "
                        "blocked_uncoalesced = sample blocks, one gradient row per (block, table row) -- the fastest, and "
                        "the sparse exchange merges by id anyway.  --reference_api and --exchange dense use reference")
    p.add_argument("--sample_blocks", default="auto",
                   help="blocks of the blocked orders; auto = cuembed_recommended_sample_blocks")
    p.add_argument("--reference_api", action="store_true",
                   help="index work through the reference's entry points only (row-id kernel + unbounded Transpose)")
    a = p.parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist
    import cuembed_amd as ce
    from cuembed_amd import distributed as D
    from cuembed_amd import harness
    from c5_train_step import TrainStep

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
    idx = torch.from_numpy(shard_of_stream(np, harness, dist if use_dist else None, a, world, rank, local)).to(dev)
    gy = torch.from_numpy(harness.allocate_grad_y(B * W, np.float16).reshape(B, W)).to(dev)
    order = a.order
    if a.exchange == "dense" or a.reference_api:
        order = "reference"
    blocks = None if a.sample_blocks == "auto" else int(a.sample_blocks)
    ts = TrainStep(ce, table, idx, gy, B, H, order=order, sample_blocks=blocks, dense=a.exchange == "dense")
    if a.reference_api:            # only what the reference's API offers: row-id kernel + Transpose over all key bits
        def reference_index_work():
            sid = ce.extract_row_ids_from_fixed(B, H, torch.int32, dev)
            ts.t_idx, ts.t_sid, _ = ce.transpose(sid, idx, workspace=ts.work)
            ts.remap = ce.compute_compressed_grad_indices(ts.t_idx)
            ts.count = ts.remap[-1:] + 1
        ts.index_work = reference_index_work
    names = ["forward", "transpose", "backward", "exchange"]
    plan, state = None, {"pending": None}
    if use_dist and a.exchange == "sparse_fixed":
        ts.compute()
        plan = ts.calibrate_exchange(D)                 # (the warm-up step that may look at sizes)

    def step(ev):
        ev[0].record()
        ts.forward()
        ev[1].record()
        ts.index_work()
        ev[2].record()
        ts.backward()
        ev[3].record()
        if plan is not None:
            if state["pending"] is not None:            # the previous step's gradient has arrived (stream-side wait)
                state["pending"].wait()
                plan.note_flags(state["pending"])
            state["pending"] = ts.exchange_fixed(D, plan)
        elif use_dist and a.exchange != "none":
            ts.exchange(D, algorithm=a.sparse_algorithm)
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
    if state["pending"] is not None:
        state["pending"].wait()
        plan.note_flags(state["pending"])
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
                          "order": ts.order, "sample_blocks": ts.blocks, "backend": dist.get_backend() if use_dist else None,
                          "ranks_share_gpus": world > ngpu,
                          "n_gpus": world, "ms_per_step": round(ms, 4),
                          "samples_per_s": round(world * B / (ms * 1e-3)),
                          "breakdown_ms": {k: round(v, 4) for k, v in parts.items()},
                          "fixed_capacity_overflowed": None if plan is None else plan.overflowed()}), flush=True)
    if use_dist:
        dist.destroy_process_group()


def shard_of_stream(np, harness, dist, a, world, rank, local):
    """This rank's shard of ONE generator stream of world x batch samples.  The stream cannot be skipped ahead
    (rejection sampling), so it is walked once per node: local rank 0 walks it and leaves it in /dev/shm, the others
    read their shard (bench.py does the same); every rank meets before the file goes."""
    B, H = a.batch, a.hotness
    if world == 1:
        return harness.generate_indices(a.rows, B, H, alpha=a.alpha)
    base = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    path = os.path.join(base, "cuembed_train_idx_%d_%s.npy" % (os.getppid(), os.environ.get("MASTER_PORT", "0")))
    if local == 0:
        tmp = path + ".tmp.%d" % os.getpid()
        with open(tmp, "wb") as f:
            np.save(f, harness.generate_indices(a.rows, world * B, H, alpha=a.alpha))
        os.replace(tmp, path)
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > 900:
            raise SystemExit("train_step_benchmark.py: index stream %s did not appear" % path)
        time.sleep(0.05)
    mine = np.ascontiguousarray(np.load(path, mmap_mode="r").reshape(world, B * H)[rank])
    dist.barrier()
    if local == 0 and os.path.exists(path):
        os.remove(path)
    return mine


if __name__ == "__main__":
    main()
