import json, os, sys
sys.path.insert(0, os.getcwd())
import torch, cuembed_amd as ce
from cuembed_amd import harness
dev=torch.device("cuda",0)
rows,W,B,H=10_000_000,256,65536,64
idx=torch.from_numpy(harness.generate_indices(rows,B,H,alpha=1.15)).to(dev)
ti,ts,_=ce.transpose_fixed_hotness(idx,B,H,num_categories=rows)
gy=torch.randint(-3,4,(B,W),device=dev).half()
def timed(ti,ts,n=20):
    remap=ce.compute_compressed_grad_indices(ti); nu=int(remap[-1].item())+1
    grad=torch.empty((nu,W),dtype=torch.float16,device=dev); inv=torch.empty((nu,),dtype=torch.int32,device=dev)
    for _ in range(3): ce.embedding_backward(gy,nu,ti,ts,remap,grad_embedding=grad,inverse_mapping=inv)
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(n): ce.embedding_backward(gy,nu,ti,ts,remap,grad_embedding=grad,inverse_mapping=inv)
    e.record(); e.synchronize(); return round(s.elapsed_time(e)/n,4), nu
pos=torch.arange(ti.numel(),device=dev,dtype=torch.int32)
folded=(ts%2048).contiguous()
out={}
out["real"]=timed(ti,ts)
out["gy_l2_resident"]=timed(ti,folded)
for run in (1,4,64,2048,65536):
    out["runs_of_%d_gy_l2_resident"%run]=timed((pos//run).contiguous(),folded)
    out["runs_of_%d_real_samples"%run]=timed((pos//run).contiguous(),ts)
for sl in (1, 2, 4):
    for seg in (32, 64, 128):
        ce.set_backward_tuning(segment_len=seg, column_slices=sl)
        out["runs_of_64_gy_l2_resident_slices%d_seg%d" % (sl, seg)] = timed((pos // 64).contiguous(), folded)
ce.set_backward_tuning()
print(json.dumps(out))
