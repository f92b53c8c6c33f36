import sys, os, ctypes
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'pytorch-unsup-pc_amd'))
import torch
from dpc.render import _native
_native.LIB_PATH=os.path.join(ROOT,'scratch','abl','libdpc_render.so')
import dpc.render as R
from oracle.dpc_oracle import Cfg, synth_inputs
B,N,G=32,8000,64
cfg=Cfg(vox_size=G,pc_gauss_kernel_size=21)
kern=R.smoothing_kernel(cfg,0.64)
pc,q,s,gt,_,_=synth_inputs(B,N,G,1234)
pc,q,s,gt=[x.cuda().float() for x in (pc,q,s,gt)]
pc.requires_grad_(True); q.requires_grad_(True); s.requires_grad_(True)
L=_native.lib(); L.dpc_debug_set_ablate.argtypes=[ctypes.c_int]
def step():
    pc.grad=q.grad=s.grad=None
    proj=R.pointcloud_project_fast(cfg,pc,q,None,None,kern,scaling_factor=s)["proj"]
    loss,_=R.silhouette_loss(proj,gt); loss.backward()
for name,v in [("full",0),("int atomics",32),("plain stores",64),("no atomics",2)]:
    L.dpc_debug_set_ablate(v)
    for _ in range(3): step()
    prof=_native.profile_kernels(lambda:[step() for _ in range(20)], torch.device('cuda'))
    print("%-22s"%name, {k:"%.1f"%(1e3*sum(x[5:])/len(x[5:])) for k,x in prof.items() if k in ('k_splat_hw',)})
