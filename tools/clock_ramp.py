import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'pytorch-unsup-pc_amd'))
import torch, numpy as np
import dpc.render as R
from dpc.harness import chair_unsupervised
import bench
B,N,G,SIG,K=bench.CONFIGS['c2']
cfg=chair_unsupervised(vox_size=G,pc_gauss_kernel_size=21)
kern=R.smoothing_kernel(cfg,SIG)
pc,q,s,gt=[x.cuda().float() for x in bench.synthetic_inputs(B,N,G,1234)]
pc.requires_grad_(True); q.requires_grad_(True); s.requires_grad_(True)
one=torch.ones((),device='cuda')
def step():
    pc.grad=q.grad=s.grad=None
    loss,_,_=R.pointcloud_project_loss(cfg,pc,q,None,None,kern,scaling_factor=s,gt=gt); loss.backward(gradient=one)
side=torch.cuda.Stream()
with torch.cuda.stream(side):
    for _ in range(3): step()
    side.synchronize()
    g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g,stream=side): step()
    for idle_ms in (0, 1, 20):
        for rep in range(2):
            for _ in range(5): g.replay()
            torch.cuda.synchronize(); time.sleep(idle_ms*1e-3)
            evs=[torch.cuda.Event(enable_timing=True) for _ in range(41)]
            evs[0].record(side)
            for i in range(40):
                g.replay(); evs[i+1].record(side)
            torch.cuda.synchronize()
            ts=[evs[i].elapsed_time(evs[i+1])*1e3 for i in range(40)]
            print("idle %2d ms: first 10 steps %s | steps 10-20 mean %.1f | 30-40 mean %.1f" % (idle_ms, " ".join("%.0f"%t for t in ts[:10]), np.mean(ts[10:20]), np.mean(ts[30:40])))
    def window(n):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record(side)
        for _ in range(n): g.replay()
        b.record(side); b.synchronize()
        return a.elapsed_time(b)*1e3/n
    for pre in (5, 50, 200, 500, 2000):
        time.sleep(0.2); torch.cuda.synchronize()
        for _ in range(pre): g.replay()
        torch.cuda.synchronize()
        w1=window(20); w2=window(20)
        print("after 0.2 s idle + %4d replays + sync: 20-step window %.2f us, next %.2f us" % (pre, w1, w2))
