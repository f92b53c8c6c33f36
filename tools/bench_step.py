#!/usr/bin/env python3
"""c2 step time of the ProjectLossStep plan (one native call per step).
   python tools/bench_step.py [steps]"""
import os, sys, time
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pytorch-unsup-pc_amd"))
import torch
import bench
import dpc.render as R
from dpc.harness import chair_unsupervised

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B, N, G, SIG, K = bench.CONFIGS["c2"]
cfg = chair_unsupervised(vox_size=G, pc_gauss_kernel_size=21)
kern = R.smoothing_kernel(cfg, SIG)
d = torch.device("cuda")
pc, q, s, gt = [x.to(d).float().contiguous() for x in bench.synthetic_inputs(B, N, G, 1234)]


def timed(fn, n):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6, t_host / n * 1e6


plan = R.project_loss_step(cfg, kern, B, N, d)
plan.bind(pc, q, s, gt)
us, host = timed(plan.run, steps)
print("native step plan: %.2f us per step (%.0f clouds/s), host %.1f us per call, loss %.6f" % (us, B / us * 1e6, host, float(plan.loss)))
from dpc.render import _native
prof = _native.profile_kernels(lambda: [plan.run() for _ in range(30)], d)
print("    per-kernel event times (us):", {k: round(1e3 * sum(v[10:]) / len(v[10:]), 2) for k, v in prof.items()})
print("status word:", R.check_status())
