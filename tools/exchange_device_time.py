#!/usr/bin/env python3
"""What the fixed-capacity sparse exchange costs ON THE DEVICE besides the links: one rank over real RCCL (world size 1:
the collectives degenerate to copies), the C4 gradient of one GPU (572 k compressed rows of 512 bytes).  The time of
start() + wait() is then the pack (cuembed::PackRowsByOwner: two launches), the owner's merge (TransposeFixedHotness
with the remap + EmbeddingBackward with a device-side count + cuembed::FinishOwnerPiece: one native op) and RCCL's
local copies -- the pack and the merge are the part that the link model of DESIGN section 6 does not contain, and are
timed on their own as well (`pack_ms`, `merge_ms`: what remains of the step is RCCL copying 366 MB to itself three
times, which the links replace on a real node).  With 8 ranks an owner merges 8 slots of ~89 k rows (about the same
number of rows as here).  One JSON line."""
import json
import os
import socket
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import cuembed_amd as ce
from cuembed_amd import distributed as D
from cuembed_amd import harness


def main():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    rows, B, H, W = 10_000_000, 65536, 64, 256
    idx = torch.from_numpy(harness.generate_indices(rows, B, H, alpha=1.15)).cuda()
    gy = torch.randint(-10, 11, (B, W), device="cuda").half()
    ti, ts, _, remap = ce.transpose_fixed_hotness(idx, B, H, num_categories=rows, remapped=True)
    cap = min(B * H, rows)
    grad = torch.empty((cap, W), dtype=torch.float16, device="cuda")
    inv = torch.empty((cap,), dtype=torch.int32, device="cuda")
    ce.embedding_backward(gy, None, ti, ts, remap, grad_embedding=grad, inverse_mapping=inv)
    count = remap[-1:] + 1
    plan = D.SparseGradExchange.calibrate(grad, inv, rows, count=count)
    for _ in range(3):
        plan.start(grad, inv, count=count).wait()
    torch.cuda.synchronize()
    a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    a.record()
    for _ in range(n):
        plan.start(grad, inv, count=count).wait()
    z.record()
    z.synchronize()
    fixed_ms = a.elapsed_time(z) / n
    # the two native halves alone, on the exchange's own buffers (no collective in between)
    ex = D._exchange_ops()

    def pack():
        ex.pack(inv, grad, count.reshape(-1), plan._cuts, plan.pair_capacity, 0, rows, plan._send_ids, plan._send_rows,
                plan._starts, plan._flag)

    def merge():
        piece_rows, piece_ids, piece_tail = plan._piece[0]
        D._merge_fixed(plan._send_ids, plan._send_rows, rows, plan.piece_capacity, plan._lo, plan._range,
                       piece_ids, piece_rows, piece_tail, plan._flag)
    parts = {}
    for name, fn in (("pack_ms", pack), ("merge_ms", merge)):
        for _ in range(3):
            fn()
        a.record()
        for _ in range(n):
            fn()
        z.record()
        z.synchronize()
        parts[name] = round(a.elapsed_time(z) / n, 4)
    for _ in range(2):
        D.allreduce_sparse_grad(grad, inv, rows, algorithm="owner", num_unique=count)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        D.allreduce_sparse_grad(grad, inv, rows, algorithm="owner", num_unique=count)
    z.record()
    z.synchronize()
    exact_ms = a.elapsed_time(z) / n
    print(json.dumps({"rows_per_rank": int(count.item()), "pair_capacity": plan.pair_capacity,
                      "piece_capacity": plan.piece_capacity, "fixed_capacity_exchange_device_ms": round(fixed_ms, 4),
                      "pack_ms": parts["pack_ms"], "merge_ms": parts["merge_ms"],
                      "exact_size_exchange_ms_with_its_read_backs": round(exact_ms, 4), "world": 1, "backend": "nccl",
                      "overflowed": plan.overflowed()}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
