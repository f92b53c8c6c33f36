#!/usr/bin/env python3
"""c2 step time of the ProjectLossStep plan: overlapped vs plain launch sequence vs HIP-graph replay of the autograd path.
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


res = {}
for name, ov in (("plan, plain sequence", False), ("plan, overlapped", True)):
    plan = R.project_loss_step(cfg, kern, B, N, d, overlap=ov)
    us, host = timed(lambda: plan.run(pc, q, s, gt), steps)
    print("%-24s %.2f us per step (%.0f clouds/s), host %.1f us per call, overlapped=%d, loss %.6f"
          % (name, us, B / us * 1e6, host, plan.overlapped.value, float(plan.loss)))
    res[name] = [x.clone() for x in (plan.loss, plan.proj, plan.dpc, plan.dq, plan.ds)]
    from dpc.render import _native
    prof = _native.profile_kernels(lambda: [plan.run(pc, q, s, gt) for _ in range(30)], d)
    print("    per-kernel event times (us):", {k: round(1e3 * sum(v[10:]) / len(v[10:]), 2) for k, v in prof.items()})
    plan.close()
print("bit-identical:", all(torch.equal(a, b) for a, b in zip(res["plan, plain sequence"], res["plan, overlapped"])))
print("status word:", R.check_status())
