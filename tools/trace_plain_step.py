"""Which op of the reference-signature step issues which device activity (kernels, memcpys, memsets): one eager plain step
under torch.profiler.  GPU box:  python tools/trace_plain_step.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-unsup-pc_amd")):
    sys.path.insert(0, p)
import torch
from torch.profiler import ProfilerActivity, profile

import dpc.render as R


class Cfg(dict):
    __getattr__ = dict.__getitem__


B, N, G = 32, 8000, 64
d = torch.device("cuda", 0)
cfg = Cfg(vox_size=G, vox_size_z=-1, pc_gauss_kernel_size=21)
g = torch.Generator().manual_seed(1)
pc = (torch.tanh(0.5 * torch.randn(B, N, 3, generator=g)) / 2).to(d).requires_grad_(True)
q = torch.randn(B, 4, generator=g).to(d).requires_grad_(True)
s = (0.5 + 0.5 * torch.rand(B, 1, generator=g)).to(d).requires_grad_(True)
gt = torch.rand(B, G, G, 1, generator=g).to(d)
one = torch.ones((), device=d)


def step():
    kern = R.smoothing_kernel(cfg, 0.64)
    pc.grad = q.grad = s.grad = None
    proj = R.pointcloud_project_fast(cfg, pc, q, None, None, kern, scaling_factor=s)["proj"]
    loss = ((proj - gt) ** 2).sum() / B
    loss.backward(gradient=one)


for _ in range(5):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
rows = []
for e in prof.events():
    if e.device_type.name == "CUDA" or "memcpy" in e.name.lower() or "memset" in e.name.lower():
        rows.append((e.time_range.start, e.name[:90], e.device_time if hasattr(e, "device_time") else e.cuda_time))
for t, name, us in sorted(rows):
    print("%10.1f  %8.2f us  %s" % (t, us, name))
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70))
