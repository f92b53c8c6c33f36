#!/usr/bin/env python3
"""Where do the last digits of the gradients go?  Error budget of the fused step against the fp64 oracle, stage by stage:
the device's fp32 grid after the W/H passes (forward slab kernel), its dT (ray-march kernel, column backward) and its final
gradients (backward slab kernel) are each swapped into an otherwise fp64 computation.

    python tools/diag_precision.py            (GPU box; the oracle runs on its CPU)

Test infrastructure only (uses oracle/)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pytorch-unsup-pc_amd"))
import torch
import torch.nn.functional as F

import dpc.render as R
from oracle import dpc_oracle as O

F64 = torch.float64


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max()), max(1.0, float(b.abs().max()))


def budget(B, N, G, ksz, sig, seed, inputs=None):
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=ksz)
    kern = O.smoothing_kernel(cfg, sig)
    pc, q, s, gt = inputs if inputs is not None else O.synth_inputs(B, N, G, seed)[:4]
    d = torch.device("cuda")
    # ---- device
    plan = R.project_loss_step(cfg, R.smoothing_kernel(cfg, sig), B, N, d)
    plan.run(pc.float().to(d).contiguous(), q.float().to(d).contiguous(), s.float().to(d).contiguous(), gt.float().to(d).contiguous())
    torch.cuda.synchronize()
    dev_gw = plan.grid_wh.double().cpu()
    dev_dT = plan.ws[:B * G * G * G * 4].view(torch.float32).view(B, G, G, G).double().cpu()
    dev = dict(dpc=plan.dpc.double().cpu(), dq=plan.dq.double().cpu(), ds=plan.ds.double().cpu(), proj=plan.proj.double().cpu())

    # ---- fp64 oracle, split at the same places
    def forward(gw_override=None):
        # fp32 leaves like the reference's inputs (its first transform ops run in fp32; the gradients come back rounded to
        # fp32 ONCE, half an ulp: compare at that resolution)
        cp, cq, cs = (x.float().clone().requires_grad_(True) for x in (pc, q, s))
        tr = O.pc_perspective_transform(cfg, cp, cq)
        raw, _ = O.pointcloud2voxels3d_fast(cfg, tr, None)
        vox = torch.clamp(raw.unsqueeze(1), 0.0, 1.0)
        for k in kern[:2]:   # W, then H
            vox = F.conv3d(vox, k.to(F64), stride=1, padding=tuple(int(x) // 2 for x in k.shape[2:]))
        gw = vox   # [B,1,D,H,W]: what the device calls grid_wh
        gw_used = gw if gw_override is None else gw + (gw_override.unsqueeze(1) - gw).detach()   # device values, fp64 graph
        gw_leaf = gw_used.detach().clone().requires_grad_(True)
        k = kern[2]
        v = F.conv3d(gw_leaf, k.to(F64), stride=1, padding=tuple(int(x) // 2 for x in k.shape[2:]))
        v = v.squeeze(1).unsqueeze(-1)
        v = torch.clamp(v * cs.reshape(-1, 1, 1, 1, 1), 0.0, 1.0)
        proj, _ = O.drc_projection(v, cfg)
        proj = torch.flip(proj, [1])
        loss = ((proj - gt) ** 2).sum() / B
        dgw, dcs = torch.autograd.grad(loss, [gw_leaf, cs])
        return dict(cp=cp, cq=cq, gw=gw, dgw=dgw.squeeze(1), ds=dcs, proj=proj.detach())

    def finish(fw, dgw):
        """fp64 backward of splat + clamp + W/H passes + transform from a given d(grid_wh)."""
        dpc, dq = torch.autograd.grad(fw["gw"], [fw["cp"], fw["cq"]], grad_outputs=dgw.unsqueeze(1), retain_graph=True)
        return dpc, dq

    ref = forward()
    ref_dpc, ref_dq = finish(ref, ref["dgw"])
    mixed = forward(dev_gw)                      # device forward grid, everything after it in fp64
    mix_dpc, mix_dq = finish(mixed, mixed["dgw"])
    fromdT_dpc, fromdT_dq = finish(ref, dev_dT)  # device dT, the backward slab kernel's work in fp64

    print("== B=%d N=%d G=%d taps=%d sigma=%.2f" % (B, N, G, ksz, sig))
    for name, (a, b_) in {
        "grid_wh   device vs fp64": (dev_gw, ref["gw"].squeeze(1)),
        "proj      device vs fp64": (dev["proj"].squeeze(-1), ref["proj"].squeeze(-1)),
        "dT        device vs fp64 (all of it)": (dev_dT, ref["dgw"]),
        "dT        device vs fp64 continued from the DEVICE grid (column kernel alone)": (dev_dT, mixed["dgw"]),
        "dT        fp64 from device grid vs fp64 (forward rounding, propagated)": (mixed["dgw"], ref["dgw"]),
        "dpc       device vs fp64 (all of it)": (dev["dpc"], ref_dpc),
        "dpc       fp64 from the device dT vs fp64 (everything before the backward slab kernel)": (fromdT_dpc, ref_dpc),
        "dpc       device vs fp64 from the device dT (backward slab kernel alone)": (dev["dpc"], fromdT_dpc),
        "dpc       fp64 behind the device grid vs fp64 (forward slab kernel alone)": (mix_dpc, ref_dpc),
        "dq        device vs fp64 (all of it)": (dev["dq"], ref_dq),
        "dq        fp64 from the device dT vs fp64 (everything before the backward slab kernel)": (fromdT_dq, ref_dq),
        "dq        device vs fp64 from the device dT (backward slab kernel alone)": (dev["dq"], fromdT_dq),
        "dq        fp64 behind the device grid vs fp64 (forward slab kernel alone)": (mix_dq, ref_dq),
        "ds        device vs fp64": (dev["ds"], ref["ds"]),
    }.items():
        e, sc = rel(a, b_)
        print("  %-92s err %.3e  scale %.3g  ratio to the rule %.2f" % (name, e, sc, e / (1e-5 * sc)))


def winners_of_the_shared_set_case():
    """The inputs of tests/test_gpu_parity.py::test_shared_point_sets_vs_oracle[8-8-2], reduced to the two winning clouds (the
    only ones with a backward): the case whose d(q) sits at the parity rule."""
    K = reps = 8
    S, N, G = 2, 1300, 32
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=11)
    pc = O.synth_inputs(S, N, G, 5100 + K)[0]
    _, q, s, _, _, _ = O.synth_inputs(S * reps, 4, G, 5200)
    gt = O.synth_inputs(S, 1, G, 5300)[3]
    ref = O.pointcloud_project_fast(cfg, pc.repeat_interleave(reps, dim=0), q, None, None, O.smoothing_kernel(cfg, 0.9), scaling_factor=s)
    _, win = O.proj_loss_pose_candidates(gt, ref["proj"], K)
    rows = torch.arange(S) * K + win
    print("winners", win.tolist())
    return pc, q[rows], s[rows], gt


def shared_set_case_both_paths():
    """d(q) of the K = reps = 8 case through the unfused column kernels (as the test runs it) and, for its two winners alone,
    through the fused K = 1 path -- against the same oracle gradients."""
    K = reps = 8
    S, N, G = 2, 1300, 32
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=11)
    pc = O.synth_inputs(S, N, G, 5100 + K)[0]
    _, q, s, _, _, _ = O.synth_inputs(S * reps, 4, G, 5200)
    gt = O.synth_inputs(S, 1, G, 5300)[3]
    leaf = lambda x: x.clone().requires_grad_(True)
    cp, cq, cs = leaf(pc), leaf(q), leaf(s)
    ref = O.pointcloud_project_fast(cfg, cp.repeat_interleave(reps, dim=0), cq, None, None, O.smoothing_kernel(cfg, 0.9), scaling_factor=cs)
    rloss, win = O.proj_loss_pose_candidates(gt, ref["proj"], K)
    rloss.backward()
    rows = torch.arange(S) * K + win
    d = torch.device("cuda")
    g = lambda x: x.float().to(d).clone().requires_grad_(True)
    gp, gq, gs = g(pc), g(q), g(s)
    loss, _, _ = R.pointcloud_project_loss(cfg, gp, gq, None, None, R.smoothing_kernel(cfg, 0.9), scaling_factor=gs, gt=gt.float().to(d), num_candidates=K)
    loss.backward()
    print("K=8 path: dq err %.3e dpc err %.3e (scale dq %.3g)" % (rel(gq.grad, cq.grad)[0], rel(gp.grad, cp.grad)[0], rel(gq.grad, cq.grad)[1]))
    hp, hq, hs = g(pc), g(q[rows]), g(s[rows])
    loss1, _, _ = R.pointcloud_project_loss(cfg, hp, hq, None, None, R.smoothing_kernel(cfg, 0.9), scaling_factor=hs, gt=gt.float().to(d), num_candidates=1)
    loss1.backward()
    print("K=1 on the winners: dq err %.3e dpc err %.3e" % (rel(hq.grad, cq.grad[rows])[0], rel(hp.grad, cp.grad)[0]))
    print("loss K=8 %.9f K=1 %.9f oracle %.9f" % (float(loss), float(loss1), float(rloss)))
    print("dq K=8", gq.grad[rows.to(d)].cpu().numpy(), "\ndq K=1", hq.grad.cpu().numpy(), "\ndq ref", cq.grad[rows].numpy())


if __name__ == "__main__":
    shared_set_case_both_paths()
    budget(2, 1300, 32, 11, 0.9, 0, winners_of_the_shared_set_case())
    budget(2, 1300, 32, 11, 0.9, 5101)
    budget(3, 8000, 64, 21, 0.64, 77)
    budget(2, 4000, 64, 21, 1.0, 78)
