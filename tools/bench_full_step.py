#!/usr/bin/env python3
"""BASELINE config 3: the chair_unsupervised training step (CNN encoder + decoder + pose candidates + renderer + loss +
Adam) on one MI355X, 8 objects x 4 views = 32 images of 128x128, 4 pose candidates -> 128 clouds into 64^3 grids.
Synthetic images/masks, random-init weights, fp32.  One JSON line per variant:
  keep=0.07  the experiment's live setting at step 0 (560 of 8000 points survive the dropout), host RNG like the reference
  keep=0.07d the same with the dropout drawn on the device
  keep=1.0   all 8000 points (the end of the schedule)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-unsup-pc_amd")):
    sys.path.insert(0, p)
import torch

from dpc.harness import TrainStep, chair_unsupervised
from dpc.render import _native


def main():
    dev = torch.device("cuda")
    steps, warmup = 50, 5
    for label, keep, on_device in (("keep=0.07 (host RNG, reference protocol)", 0.07, False),
                                   ("keep=0.07 (device RNG)", 0.07, True), ("keep=1.0", 1.0, False)):
        cfg = chair_unsupervised(pc_point_dropout=keep)
        torch.manual_seed(0)
        step = TrainStep(cfg, dev, device_dropout=on_device)
        nimg = cfg.batch_size * cfg.step_size
        g = torch.Generator().manual_seed(3)
        images = torch.rand(nimg, 3, 128, 128, generator=g).to(dev)
        masks = (torch.rand(nimg, 1, 128, 128, generator=g) > 0.5).float().to(dev)
        run = step
        for _ in range(warmup):
            run(images, masks)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = run(images, masks)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / steps
        prof = _native.profile_kernels(lambda: [step(images, masks) for _ in range(5)], dev)
        floor = 0.5 * _native.event_pair_overhead_ms(dev)
        render_us = sum(1e3 * (sorted(v)[len(v) // 2] - floor) for v in prof.values())
        nparam = sum(p.numel() for p in step.nets.parameters())
        print(json.dumps({
            "metric": "chair_unsupervised train steps/sec (fwd + loss + bwd + Adam)", "value": 1.0 / wall, "unit": "steps/sec",
            "ms_per_step": 1e3 * wall, "images_per_sec": nimg / wall,
            "clouds_per_sec": nimg * cfg.pose_predict_num_candidates / wall, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: batch 8 objects x 4 views (32 images 128x128x3), K=4 pose candidates -> "
                                   "128 clouds, %s, 64^3 grid, 21 taps sigma_rel=3.0, eager launches" % label,
                       "parameters": nparam},
            "renderer_kernels_us_per_step": render_us, "renderer_kernels": {k: len(v) // 5 for k, v in prof.items()},
            "loss": float(loss)}))


if __name__ == "__main__":
    main()
