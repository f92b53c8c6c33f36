import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'pytorch-unsup-pc_amd'))
import numpy as np, torch
import dpc.render as R
from dpc.render import _geometry
from dpc.render._ops import Transform, Splat, Smooth, Drc
from oracle import dpc_oracle as O
torch.set_printoptions(precision=9)
B,N,G,k,sigma,smooth,seed=1,8000,64,21,3.0,False,1234
cfg=O.Cfg(vox_size=G, pc_gauss_kernel_size=k)
pc,q,s,gt,_,_=O.synth_inputs(B,N,G,seed)
leaf=lambda x:x.clone().requires_grad_(True)
cp,cq,cs=leaf(pc),leaf(q),leaf(s)
ref=O.pointcloud_project_fast(cfg,cp,cq,None,None,O.smoothing_kernel(cfg,sigma),scaling_factor=cs,smooth=smooth)
ref['voxels_raw'].retain_grad(); ref['tr_pc'].retain_grad()
(((ref['proj']-gt)**2).sum()/B).backward()
geom=_geometry(cfg,None)
gp,gq,gs=[leaf(x.cuda()) for x in (pc,q,s)]
tr=Transform.apply(gp,gq,None,None,geom); tr.retain_grad()
raw=Splat.apply(tr,geom); raw.retain_grad()
vox=torch.clamp(raw,0,1)
vox=torch.clamp(vox*gs.reshape(-1,1,1,1),0,1)
proj,probs,_=Drc.apply(vox,geom)
proj=torch.flip(proj,[1]).unsqueeze(-1)
(((proj-gt.cuda().float())**2).sum()/B).backward()
print('raw err', (raw.double().cpu()-ref['voxels_raw']).abs().max().item())
e=(raw.grad.double().cpu()-ref['voxels_raw'].grad).abs()
print('draw err max', e.max().item(), 'n>1e-6', int((e>1e-6).sum()))
for r in torch.nonzero(e>1e-4)[:12]:
    b,z,y,x=r.tolist()
    print(' vox',b,z,y,x,'raw ref %.10e gpu %.10e'%(ref['voxels_raw'][b,z,y,x].item(),raw[b,z,y,x].item()),'s*raw ref %.10e'%(ref['voxels_raw'][b,z,y,x].item()*s[b].item()),'draw ref %.6e gpu %.6e'%(ref['voxels_raw'].grad[b,z,y,x].item(), raw.grad[b,z,y,x].item()))
e2=(tr.grad.double().cpu()-ref['tr_pc'].grad).abs()
print('dtr err max',e2.max().item())
i=241
print('pt241 dtr ref',ref['tr_pc'].grad[0,i].tolist(),'gpu',tr.grad[0,i].tolist())
print('pt241 tr ref',ref['tr_pc'][0,i].tolist(),'gpu',tr[0,i].tolist())
