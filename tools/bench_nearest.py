#!/usr/bin/env python3
"""Throughput of dpc.render.point_cloud_distance (the Chamfer evaluation's kernel) on MI355X,
with the CPU oracle timed beside it.  One JSON line per case.  Compute-bound: 8 flops per (source, target) pair
(3 subtractions, 3 multiplications, 2 additions) priced against the VECTOR peak of the dtype (no MFMA: the pair
distance must be the reference's difference-of-coordinates form to give its indices bit for bit)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-unsup-pc_amd")):
    sys.path.insert(0, p)
import torch

import dpc.render as R
from dpc.render import _native
from oracle import dpc_oracle as O

PEAK_TFLOPS = {torch.float32: 157.3, torch.float64: 78.6}  # MI355X_MICROARCH.md, vector (non-MFMA) peaks


def main():
    dev = torch.device("cuda")
    for ns, nt, dt in ((8000, 100000, torch.float64), (8000, 100000, torch.float32), (8000, 8000, torch.float64)):
        g = torch.Generator().manual_seed(1)
        vs = (torch.rand(ns, 3, generator=g, dtype=dt) - 0.5).to(dev)
        vt = (torch.rand(nt, 3, generator=g, dtype=dt) - 0.5).to(dev)
        for _ in range(3):
            R.point_cloud_distance(vs, vt)
        torch.cuda.synchronize()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            R.point_cloud_distance(vs, vt)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps
        prof = _native.profile_kernels(lambda: [R.point_cloud_distance(vs, vt) for _ in range(10)], dev)
        floor = 0.5 * _native.event_pair_overhead_ms(dev)
        kern_us = {k: 1e3 * (sorted(v)[len(v) // 2] - floor) for k, v in prof.items()}
        pairs = ns * nt
        tflops = 8 * pairs / (kern_us["k_nearest_partial"] * 1e-6) / 1e12
        # CPU oracle on a bounded sample of the sources (same targets)
        sample = 256
        t0 = time.perf_counter()
        O.point_cloud_distance(vs[:sample].cpu(), vt.cpu())
        cpu = sample * nt / (time.perf_counter() - t0)
        print(json.dumps({
            "metric": "nearest-target pairs/sec (point_cloud_distance)", "value": pairs / wall, "unit": "pairs/sec",
            "config": {"workload": "Ns=%d sources x Nt=%d targets" % (ns, nt), "dtype": str(dt).split(".")[1]},
            "call_us": 1e6 * wall, "kernels_us": kern_us,
            "roofline": {"bound": "vector fp pipe (no MFMA)", "achieved": tflops, "peak": PEAK_TFLOPS[dt], "unit": "TFLOP/s",
                         "frac": tflops / PEAK_TFLOPS[dt], "flops_per_pair": 8},
            "cpu_baseline": {"value": cpu, "unit": "pairs/sec", "cores": torch.get_num_threads(), "kind": "port",
                             "sample": "%d of the %d sources against all targets (numpy oracle)" % (sample, ns)}}))


if __name__ == "__main__":
    main()
