#!/usr/bin/env python3
"""Are two builds of the library bit-identical on the hot path?  Runs the fused step (and the plain staged path) of a few
configurations once per library variant, each in a child process (DPC_RENDER_LIB selects the build), and compares SHA-256
digests of every output and gradient.  For kernel rewrites that must not change a single bit (sort order, sum order).

    python tools/compare_variants.py main locb        # main = in-tree, else scratch/<name>/libdpc_render.so
"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [("c2", 32, 8000, 64, 0.64, 0), ("c4", 8, 16000, 128, 1.28, 0), ("c5", 128, 2000, 32, 0.32, 0), ("ragged", 3, 1237, 64, 0.64, 0),
         ("odd-grid", 2, 3001, 48, 0.5, 0), ("dropout", 4, 8000, 64, 0.64, 5000)]


def child():
    for p in (ROOT, os.path.join(ROOT, "pytorch-unsup-pc_amd")):
        sys.path.insert(0, p)
    import torch

    import dpc.render as R
    from bench import synthetic_inputs
    from dpc.harness import chair_unsupervised

    dev = torch.device("cuda")
    out = {}

    def digest(t):
        return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:16]

    for name, B, N, G, sigma, keep in CASES:
        cfg = chair_unsupervised(vox_size=G, pc_gauss_kernel_size=21)
        kern = R.smoothing_kernel(cfg, sigma)
        pc, q, s, gt = [x.to(dev) for x in synthetic_inputs(B, N, G, 4321)]
        pc[:, ::7] *= 1.6   # some points off the grid: the out-of-bounds bin is populated
        pc.requires_grad_(True), q.requires_grad_(True), s.requires_grad_(True)
        sel = torch.randperm(N, generator=torch.Generator().manual_seed(5))[:keep].to(dev) if keep else None
        pc_in = pc[:, sel] if keep else pc
        loss, o, _ = R.pointcloud_project_loss(cfg, pc_in, q, None, None, kern, scaling_factor=s, gt=gt)
        loss.backward()
        out[name] = {"loss": digest(loss), "proj": digest(o["proj"]), "dpc": digest(pc.grad), "dq": digest(q.grad), "ds": digest(s.grad)}
        pc.grad = q.grad = s.grad = None
        pc_in = pc[:, sel] if keep else pc
        o2 = R.pointcloud_project_fast(cfg, pc_in, q, None, None, kern, scaling_factor=s)
        (o2["proj"] * gt).sum().backward()
        out[name].update({"plain_proj": digest(o2["proj"]), "plain_voxels": digest(o2["voxels"]), "plain_dpc": digest(pc.grad), "plain_dq": digest(q.grad)})
    print("DIGESTS " + json.dumps(out))


def main():
    if os.environ.get("DPC_COMPARE_CHILD"):
        return child()
    res = {}
    for v in sys.argv[1:]:
        env = dict(os.environ, DPC_COMPARE_CHILD="1")
        if v != "main":
            env["DPC_RENDER_LIB"] = os.path.join(ROOT, "scratch", v, "libdpc_render.so")
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=600)
        lines = [l for l in r.stdout.splitlines() if l.startswith("DIGESTS ")]
        if r.returncode != 0 or not lines:
            print(r.stdout[-2000:], r.stderr[-4000:])
            sys.exit(2)
        res[v] = json.loads(lines[0][8:])
    first = sys.argv[1]
    bad = 0
    for v in sys.argv[2:]:
        for case, d in res[first].items():
            diff = [k for k in d if res[v][case][k] != d[k]]
            print("%-10s %s vs %s: %s" % (case, first, v, "identical (%d arrays)" % len(d) if not diff else "DIFFERENT: " + ", ".join(diff)))
            bad += bool(diff)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
