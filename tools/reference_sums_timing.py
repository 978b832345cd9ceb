#!/usr/bin/env python3
"""EmbeddingBackward at the C4 shape (10M x 256, 65,536 x 64 lookups, alpha 1.15, compressed gradient) on
NON-representable grad_y: the default entry point (fp32 partial sums) against EmbeddingBackwardReferenceSums (the
reference's per-lookup GradT rounding), fp16 and fp32.  Prints one line per (type, path): ms per call."""
import sys, time
sys.path.insert(0, '.')
import torch
import cuembed_amd as ce
from cuembed_amd import harness
B,H,W,rows=65536,64,256,10_000_000
idx=torch.from_numpy(harness.generate_indices(rows,B,H,alpha=1.15)).cuda()
ti,ts,_,remap=ce.transpose_fixed_hotness(idx,B,H,num_categories=rows,remapped=True)
nu=int(remap[-1])+1
for dt in (torch.float16, torch.float32):
    gy=(torch.randn(B,W,device='cuda')*3).to(dt)
    g=torch.empty((nu,W),dtype=dt,device='cuda'); inv=torch.empty((nu,),dtype=torch.int32,device='cuda')
    for rs in (False, True):
        for _ in range(2): ce.embedding_backward(gy,nu,ti,ts,remap,grad_embedding=g,inverse_mapping=inv,reference_sums=rs)
        torch.cuda.synchronize(); t=time.time()
        for _ in range(5): ce.embedding_backward(gy,nu,ti,ts,remap,grad_embedding=g,inverse_mapping=inv,reference_sums=rs)
        torch.cuda.synchronize(); print(dt, 'reference_sums' if rs else 'default', round((time.time()-t)/5*1e3,4),'ms', flush=True)
