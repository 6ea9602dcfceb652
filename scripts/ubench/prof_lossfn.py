import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lc_amd.config import AttrDict
from lc_amd.losses import Loss_fn
from tests.golden.gen_golden_lossfn import DENSE_CFG
from lc_amd.synth import dense_inputs
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
fn = Loss_fn(AttrDict(DENSE_CFG), AttrDict(), 0).to(dev)
gt, out = dense_inputs(B=32, H=64, W=64)
gt = {k: (v.to(dev).contiguous() if isinstance(v, torch.Tensor) else v) for k, v in gt.items()}  # loader tensors are contiguous (the generator hands out strided views)
out = {k: v.to(dev).contiguous() for k, v in out.items()}
def step(i):
    np.random.seed(i)
    leaves = {k: v.detach().requires_grad_(True) for k, v in out.items()}
    ld, wd = fn(gt, leaves, 1, 1000 + i, 10)
    total = sum(wd.values())
    torch.autograd.grad(total, list(leaves.values()), allow_unused=True)
for i in range(5): step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for i in range(10): step(i)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=35, max_name_column_width=60))
