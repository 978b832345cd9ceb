#!/usr/bin/env python3
"""What reading the device-side row-load decision costs the forward: EmbeddingForward at C2 (alpha 1.15) with and
without ForwardOptions::row_loads_device (decision word 0), 200 launches each, three repetitions.  (Nothing: ratio 1.001.)"""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
import cuembed_amd as ce
from cuembed_amd import harness
rows,B,H,W=10_000_000,65536,64,256
table=torch.empty((rows,W),dtype=torch.float16,device='cuda').uniform_(-1,1)
idx=[torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in harness.generate_indices(rows,4*B,H,alpha=1.15).reshape(4,-1)]
out=torch.empty((B,W),dtype=torch.float16,device='cuda')
dec=ce.new_row_loads_decision()
def timed(fn,n=200):
    for i in range(20): fn(i)
    a,z=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n): fn(i)
    z.record(); z.synchronize()
    return a.elapsed_time(z)/n
for rep in range(3):
    t0=timed(lambda i: ce.embedding_forward(table,idx[i%4],num_hots=H,out=out))
    t1=timed(lambda i: ce.embedding_forward(table,idx[i%4],num_hots=H,out=out,row_loads_device=dec))
    print("plain %.4f ms   with row_loads_device (word 0) %.4f ms   ratio %.4f" % (t0,t1,t1/t0), flush=True)
