#!/usr/bin/env python3
"""Soak test of the fused training step on MI355X: replay the captured step many times and check that the loss, the
per-cloud squared errors and all gradients stay bit-identical (integer splat, fixed reduction order) -- a cheap detector
for races in the ticket logic, the in-LDS passes and the atomics.  Exit code 1 on the first deviation."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-unsup-pc_amd")):
    sys.path.insert(0, p)
import torch

import dpc.render as R
from dpc.harness import chair_unsupervised

sys.path.insert(0, ROOT)
from bench import synthetic_inputs  # noqa: E402


def main():
    replays = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    which = sys.argv[2] if len(sys.argv) > 2 else "c2"     # c2 | c4 (128-wide rolling backward) | wide (tap radius 10)
    dev = torch.device("cuda")
    B, N, G, sigma = {"c2": (32, 8000, 64, 0.64), "c4": (8, 16000, 128, 1.28), "wide": (32, 8000, 64, 3.0)}[which]
    cfg = chair_unsupervised(vox_size=G, pc_gauss_kernel_size=21)
    kern = R.smoothing_kernel(cfg, sigma)
    pc, q, s, gt = [x.to(dev) for x in synthetic_inputs(B, N, G, 1234)]
    pc.requires_grad_(True), q.requires_grad_(True), s.requires_grad_(True)
    one = torch.ones((), device=dev)

    def step():
        pc.grad = q.grad = s.grad = None
        loss, out, _ = R.pointcloud_project_loss(cfg, pc, q, None, None, kern, scaling_factor=s, gt=gt)
        loss.backward(gradient=one)
        return loss, out["proj"]

    side = torch.cuda.Stream(dev)
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
            loss, proj = step()
        graph.replay()
        side.synchronize()
        ref = [t.detach().clone() for t in (loss, proj, pc.grad, q.grad, s.grad)]
        names = ["loss", "proj", "dpc", "dq", "ds"]
        # bit-exact, all of them: integer accumulation or fixed-order sums on every path (since round 2)
        exact = [True, True, True, True, True]
        worst = [0.0] * 5
        for i in range(replays):
            graph.replay()
            if i % 500 == 499:
                side.synchronize()
                for k, (a, b) in enumerate(zip((loss, proj, pc.grad, q.grad, s.grad), ref)):
                    d = float((a.detach() - b).abs().max())
                    worst[k] = max(worst[k], d)
                    if (exact[k] and d != 0.0) or not torch.isfinite(a).all() or d > 1e-4 * max(1.0, float(b.abs().max())):
                        print("DEVIATION at replay", i, names[k], d)
                        sys.exit(1)
        side.synchronize()
    print("soak ok (%s): %d replays; max deviations" % (which, replays), dict(zip(names, worst)))


if __name__ == "__main__":
    main()
