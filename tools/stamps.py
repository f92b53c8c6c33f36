#!/usr/bin/env python3
"""Per-phase timeline of the slab kernels from a -DDPC_ABLATE build (tools/build_variant.sh abl -DDPC_ABLATE):
   DPC_RENDER_LIB=scratch/abl/libdpc_render.so python tools/stamps.py [c2|c4|c5]
Stamps are 100 MHz s_memrealtime values written by thread 0 of every workgroup (diagnostic build only)."""
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

cfgname = sys.argv[1] if len(sys.argv) > 1 else "c2"
B, N, G, SIG, K = bench.CONFIGS[cfgname]
cfg = chair_unsupervised(vox_size=G, pc_gauss_kernel_size=21)
kern = R.smoothing_kernel(cfg, SIG)
pc, q, s, gt = [x.cuda().float() for x in bench.synthetic_inputs(B, N, G, 1234)]
if K > 1:  # like bench.py: the K candidates of a sample share its point set, scale and mask
    pc, s, gt = pc[:B // K].contiguous(), s[:B // K].repeat_interleave(K, dim=0).contiguous(), gt[:B // K].contiguous()
pc.requires_grad_(True), q.requires_grad_(True), s.requires_grad_(True)
L = _native.lib()
L.dpc_debug_set_stamps.argtypes = [ctypes.c_void_p]
L.dpc_debug_set_ablate.argtypes = [ctypes.c_int]
L.dpc_debug_set_ablate(int(os.environ.get('DPC_ABL_BITS', '0'), 0))


def step():
    pc.grad = q.grad = s.grad = None
    loss, _, _ = R.pointcloud_project_loss(cfg, pc, q, None, None, kern, scaling_factor=s, gt=gt, num_candidates=K)
    loss.backward()


for _ in range(5):
    step()
NB = max(64 * B, 4096)   # one 16-slot row per workgroup of the largest grid (c4: 1024 forward slabs)
buf = torch.zeros(NB * 16, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
L.dpc_debug_set_stamps(ctypes.c_void_p(buf.data_ptr()))
step()
torch.cuda.synchronize()
L.dpc_debug_set_stamps(None)
st = buf.cpu().numpy().reshape(-1, 16).astype(np.float64) * 0.01  # us


def report(name, slots, labels):
    rows = st[(st[:, slots[0]] > 0) & (st[:, slots[-1]] > 0)]
    if not len(rows):
        print(name, ": no stamps")
        return
    a = rows[:, slots]
    t0 = a[:, 0].min()
    start, end = a[:, 0] - t0, a[:, -1] - t0
    dur = a[:, -1] - a[:, 0]
    print("%s: %d workgroups; last start %.2f us, span %.2f us, WG duration mean %.2f (min %.2f max %.2f)"
          % (name, len(rows), start.max(), end.max(), dur.mean(), dur.min(), dur.max()))
    d = np.diff(a, axis=1)
    for l, m, mx in zip(labels, d.mean(0), d.max(0)):
        print("    %-28s mean %6.2f us   max %6.2f" % (l, m, mx))
    print("    start deciles", np.round(np.percentile(start, [0, 10, 25, 50, 75, 90, 100]), 2),
          " end deciles", np.round(np.percentile(end, [0, 10, 25, 50, 75, 90, 100]), 2))


if (st[:, 14] > 0).any():  # x-in-lanes kernel staying for two slabs
    report("k_splat_xl (two slabs)", [0, 1, 2, 3, 4, 14, 6, 7, 5],
           ["slab 0: zero+table", "slab 0: scatter", "slab 0: window+mask+convert", "slab 0: passes+stores issued",
            "slab 1: table+records+zero", "slab 1: scatter", "slab 1: window+mask+convert", "slab 1: passes+stores issued"])
elif (st[:, 4] > 0).any():
    report("k_splat_hw", [0, 1, 2, 3, 4, 5], ["zero+table", "scatter", "W-load/convert", "W-compute/write", "H-pass+store"])
else:  # x-in-lanes kernel: no fp32 slab, no barrier between the passes
    report("k_splat_hw (xl)", [0, 1, 2, 3, 5], ["zero+table", "scatter", "window+mask+convert", "H+W passes+store"])
report("k_gather_hw", [8, 9, 11, 12, 13], ["pads+H-pass(global)", "gather", "block_sum", "epilogue"])

# backward slab kernel by slab index (clouds % 8 == 0: block L -> slab (L >> 3) % slabs per cloud, dpc_kernels.h block_coords):
# is the gather phase proportional to the slab's points, or does its first pass cost a fixed price?
rows = np.nonzero((st[:, 8] > 0) & (st[:, 13] > 0))[0]
if len(rows) and B % 8 == 0:
    nx = len(rows) // B
    print("k_gather_hw by slab (mean over clouds): slab  H-pass  gather  block_sum  epilogue  total")
    for x in range(nx):
        sel = rows[(rows >> 3) % nx == x]
        a = st[sel][:, [8, 9, 11, 12, 13]]
        d = np.diff(a, axis=1).mean(0)
        print("    %2d   %6.2f  %6.2f  %6.2f  %6.2f  %6.2f" % (x, d[0], d[1], d[2], d[3], (a[:, -1] - a[:, 0]).mean()))
