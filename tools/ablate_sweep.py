#!/usr/bin/env python3
"""Timing sweep over the diagnostic switches of a -DDPC_ABLATE build (results are WRONG with switches on; timing only):
   DPC_RENDER_LIB=scratch/abl/libdpc_render.so python tools/ablate_sweep.py [c2|c4|c5] name=value [name=value ...]
Per setting: median per-kernel time (library event profiler, eager) and the graph-replayed step time, interleaved rounds."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pytorch-unsup-pc_amd"))
import numpy as np
import torch

import dpc.render as R
from dpc.harness import chair_unsupervised
from dpc.render import _native
import bench

args = sys.argv[1:]
cfgname = "c2"
if args and args[0] in bench.CONFIGS:
    cfgname = args.pop(0)
variants = [("base", 0)] + [(a.split("=")[0], int(a.split("=")[1], 0)) for a in args]
B, N, G, SIG, K = bench.CONFIGS[cfgname]
cfg = chair_unsupervised(vox_size=G, pc_gauss_kernel_size=21)
kern = R.smoothing_kernel(cfg, SIG)
pc, q, s, gt = [x.cuda().float() for x in bench.synthetic_inputs(B, N, G, 1234)]
if K > 1:
    S = B // K
    pc, s, gt = pc[:S].contiguous(), s[:S].repeat_interleave(K, dim=0).contiguous(), gt[:S].contiguous()
pc.requires_grad_(True), q.requires_grad_(True), s.requires_grad_(True)
dev = pc.device
L = _native.lib()
L.dpc_debug_set_ablate.argtypes = [ctypes.c_int]
one = torch.ones((), device=dev)


def step():
    pc.grad = q.grad = s.grad = None
    loss, _, _ = R.pointcloud_project_loss(cfg, pc, q, None, None, kern, scaling_factor=s, gt=gt, num_candidates=K)
    loss.backward(gradient=one)


side = torch.cuda.Stream(dev)
with torch.cuda.stream(side):
    for _ in range(3):
        step()
    side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        step()

    def window(n=100):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(side)
        for _ in range(n):
            graph.replay()
        b.record(side)
        b.synchronize()
        return 1e3 * a.elapsed_time(b) / n

    ov = _native.event_pair_overhead_ms(dev) * 0.5
    res = {name: [] for name, _ in variants}
    kres = {name: {} for name, _ in variants}
    for rnd in range(5):
        for name, v in variants:
            L.dpc_debug_set_ablate(v)
            window(20)
            res[name].append(min(window(100) for _ in range(3)))
    for name, v in variants:
        L.dpc_debug_set_ablate(v)
        for _ in range(5):
            step()
        prof = _native.profile_kernels(lambda: [step() for _ in range(40)], dev)
        kres[name] = {k: (float(np.median(vv)) - ov) * 1e3 for k, vv in prof.items()}
    L.dpc_debug_set_ablate(0)
for name, v in variants:
    print("%-14s %#10x  step median %.2f us (min %.2f)   %s" % (name, v, float(np.median(res[name])), min(res[name]),
          "  ".join("%s %.2f" % (k.replace("k_", ""), t) for k, t in sorted(kres[name].items()))))
