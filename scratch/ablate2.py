import sys, os, ctypes
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'pytorch-unsup-pc_amd'))
import torch
from dpc.render import _native
_native.LIB_PATH=os.path.join(ROOT,'scratch','abl','libdpc_render_z4.so')
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
    loss,_,_=R.pointcloud_project_loss(cfg,pc,q,None,None,kern,scaling_factor=s,gt=gt); loss.backward()
VARIANTS=[("full",0),("D:nothing",15<<8),("D:nothing,ret before camera",(15<<8)|(1<<12)),("D:nothing,ret before blocksum",(15<<8)|(1<<13)),("D:nothing,ret after blocksum",(15<<8)|(1<<14)),
          ("D:all but epilogue(13)",1<<13),("D:ret before camera",1<<12)]
if __name__=="__main__":
    for name,v in VARIANTS:
        L.dpc_debug_set_ablate(v)
        for _ in range(20): step()
        torch.cuda.synchronize()
