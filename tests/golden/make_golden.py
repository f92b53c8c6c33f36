#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself on CPU.

This script is the only place that imports /root/reference.  It runs in the build container only
(the reference never travels to the GPU box); its outputs -- small .npz/.json fixtures holding inputs
and the reference's outputs -- are committed and are what the oracle (oracle/dpc_oracle.py) and the HIP
path are checked against.

Import recipe follows SURVEY.md section 8(c): PYTHONDONTWRITEBYTECODE, two numpy-2 shims needed by
dpc/util/quaternion.py:22-23, attribute-dict config built from the YAML files (no easydict).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

"CUDA-branch semantics" below means the chain the reference executes when torch.cuda.is_available()
(dpc/util/point_cloud_to.py:203-209): transform -> splat -> clamp -> smoothen -> scale/clamp -> DRC -> flip.
On a CPU-only box pointcloud_project_fast skips the smoothing (:210-212), so that chain is composed here
from the reference's OWN functions, called in the reference's order.  The literal CPU call is captured too.
"""
import contextlib
import io
import json
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np

np.float = float  # numpy-2 shims, reference evaluates np.maximum_sctype(np.float) at import
np.maximum_sctype = lambda t: np.float64

import torch
import yaml

REF = os.environ.get("DPC_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(REF, "dpc"))

import util.drc as ref_drc  # noqa: E402
import util.gauss_kernel as ref_gk  # noqa: E402
import util.point_cloud_to as ref_pc  # noqa: E402
import util.quaternion as ref_q  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


class AttrDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def make_cfg(**over):
    cfg = AttrDict(yaml.safe_load(open(os.path.join(REF, "dpc/resources/default_config.yaml"))))
    cfg.update(over)
    return cfg


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()):
        yield


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print("wrote", name, "%.1f KB" % (os.path.getsize(path) / 1024))


def synth_inputs(B, N, G, seed, with_t=False, with_f=False):
    """Synthetic inputs of SURVEY.md section 8(d)."""
    g = torch.Generator().manual_seed(seed)
    pc = (torch.tanh(0.5 * torch.randn(B, N, 3, generator=g)) / 2).float()
    q = torch.randn(B, 4, generator=g).float()
    s = (0.5 + 0.5 * torch.rand(B, 1, generator=g)).float()
    mask = (torch.rand(B, 1, 2 * G, 2 * G, generator=g) > 0.5).double()
    gt = torch.nn.AvgPool2d(2)(mask).permute(0, 2, 3, 1).contiguous()  # [B,G,G,1]
    t = (0.05 * torch.randn(B, 3, generator=g)).float() if with_t else None
    f = (1.875 + 0.2 * torch.randn(B, 1, generator=g)).float() if with_f else None
    return pc, q, s, gt, t, f


def ref_chain(cfg, pc, q, t, kernel, s, f, smooth):
    """CUDA-branch semantics composed from the reference's own functions (or the literal CPU branch
    when smooth=False, which is what pointcloud_project_fast does on a CPU-only host)."""
    with quiet():
        tr_pc = ref_pc.pc_perspective_transform(cfg, pc, q, t, f)
        voxels, _ = ref_pc.pointcloud2voxels3d_fast(cfg, tr_pc, None)
    voxels = voxels.unsqueeze(1)
    raw = voxels
    voxels = torch.clamp(voxels, 0.0, 1.0)
    if kernel is not None and smooth:
        voxels = ref_pc.smoothen_voxels3d(cfg, voxels, kernel)
    voxels = voxels.squeeze(1).unsqueeze(-1)
    if s is not None:
        voxels = torch.clamp(voxels * s.reshape(-1, 1, 1, 1, 1), 0.0, 1.0)
    proj, probs = ref_drc.drc_projection(voxels, cfg)
    probs = torch.flip(probs, [2])
    depth = ref_drc.drc_depth_projection(probs, cfg)
    proj = torch.flip(proj, [1])
    return dict(proj=proj, voxels=voxels, tr_pc=tr_pc, drc_probs=probs, proj_depth=depth, raw=raw)


def leaf(x):
    return None if x is None else x.clone().requires_grad_(True)


def grad_or_none(x):
    return None if x is None or x.grad is None else x.grad.clone()


# ----------------------------------------------------------------------------------------- F1
def f1_gauss():
    arrs = {}
    for l, sig in [(11, 1.0), (21, 3.0), (21, 0.64), (21, 0.2), (11, 0.2), (10, 1.5)]:
        arrs["k_%d_%s" % (l, str(sig).replace(".", "p"))] = ref_gk.gauss_kernel_1d(l, sig)
    cfg = make_cfg(pc_gauss_kernel_size=21, vox_size=64)
    ks = ref_gk.smoothing_kernel(cfg, 3.0)
    for i, k in enumerate(ks):
        arrs["sk64_%d" % i] = k
    # vox_size_z != -1 with a different z length crashes in the reference (gauss_kernel.py:49 reshapes
    # the z kernel with fsz instead of fsz_z); only the equal-length case runs.
    cfg = make_cfg(pc_gauss_kernel_size=11, vox_size=32, vox_size_z=32)
    ks = ref_gk.smoothing_kernel(cfg, 1.5)
    for i, k in enumerate(ks):
        arrs["skz_%d" % i] = k
    save("f1_gauss.npz", **arrs)


# ----------------------------------------------------------------------------------------- F2
def f2_transform():
    cfg = make_cfg()
    g = torch.Generator().manual_seed(11)
    B, N = 3, 64
    pc = (torch.rand(B, N, 3, generator=g) - 0.5).float()
    q = torch.randn(B, 4, generator=g).float()
    t = (0.1 * torch.randn(B, 3, generator=g)).float()
    f = (1.875 + 0.3 * torch.randn(B, 1, generator=g)).float()
    w = torch.randn(B, N, 3, generator=g).double()
    arrs = dict(pc=pc, q=q, t=t, f=f, w=w)
    for tag, (ut, uf) in dict(plain=(0, 0), t=(1, 0), f=(0, 1), tf=(1, 1)).items():
        pc_, q_, t_, f_ = leaf(pc), leaf(q), leaf(t) if ut else None, leaf(f) if uf else None
        out = ref_pc.pc_perspective_transform(cfg, pc_, q_, t_, f_)
        (out * w).sum().backward()
        arrs["out_" + tag] = out
        arrs["dpc_" + tag] = pc_.grad
        arrs["dq_" + tag] = q_.grad
        if ut:
            arrs["dt_" + tag] = t_.grad
        if uf:
            arrs["df_" + tag] = f_.grad
    # identity quaternion known answer
    one = ref_pc.pc_perspective_transform(cfg, torch.tensor([[[0.1, 0.2, 0.3]]]), torch.tensor([[1.0, 0, 0, 0]]))
    arrs["identity_out"] = one
    # quaternion_rotate alone
    arrs["rot"] = ref_q.quaternion_rotate(pc, q)
    save("f2_transform.npz", **arrs)


# ----------------------------------------------------------------------------------------- F3
def f3_splat():
    arrs = {}
    for tag, (G, Gz) in dict(g16=(16, -1), g32=(32, -1), g16z8=(16, 8)).items():
        cfg = make_cfg(vox_size=G, vox_size_z=Gz)
        D = G if Gz == -1 else Gz
        g = torch.Generator().manual_seed(100 + G + D)
        pc = (1.2 * torch.rand(2, 500, 3, generator=g) - 0.6).double()  # ~42% outliers
        w = torch.randn(2, D, G, G, generator=g).double()
        pc_ = leaf(pc)
        with quiet():
            vox, _ = ref_pc.pointcloud2voxels3d_fast(cfg, pc_, None)
        (vox * w).sum().backward()
        nvalid = int(((pc >= -0.5) & (pc <= 0.5)).all(-1).sum())
        arrs.update({tag + "_pc": pc, tag + "_w": w, tag + "_vox": vox, tag + "_dpc": pc_.grad,
                     tag + "_nvalid": nvalid})
    save("f3_splat.npz", **arrs)


# ----------------------------------------------------------------------------------------- F4
def aniso_kernels(cfg, sigma):
    """What gauss_kernel.py:38-51 intends for vox_size_z != vox_size (the reference's own reshape at :49
    is broken for that case): x/y kernel of length fsz, z kernel of length fsz_z and sigma*ratio.  Built
    from the reference's gauss_kernel_1d so the arithmetic is still the reference's."""
    fsz = cfg.pc_gauss_kernel_size
    ratio = cfg.vox_size_z / cfg.vox_size
    fsz_z = int(np.floor(fsz * ratio))
    if fsz_z % 2 == 0:
        fsz_z += 1
    k = ref_gk.gauss_kernel_1d(fsz, sigma)
    kz = ref_gk.gauss_kernel_1d(fsz_z, sigma * ratio)
    return [k.reshape(1, 1, 1, 1, fsz), k.reshape(1, 1, 1, fsz, 1), kz.reshape(1, 1, fsz_z, 1, 1)]


def f4_smooth():
    arrs = {}
    for tag, (k, sig) in dict(k11=(11, 1.0), k21=(21, 3.0), k21s=(21, 0.64)).items():
        cfg = make_cfg(vox_size=16, pc_gauss_kernel_size=k)
        g = torch.Generator().manual_seed(200 + k)
        x = torch.rand(2, 1, 16, 16, 16, generator=g).double()
        w = torch.randn(2, 1, 16, 16, 16, generator=g).double()
        x_ = leaf(x)
        y = ref_pc.smoothen_voxels3d(cfg, x_, ref_gk.smoothing_kernel(cfg, sig))
        (y * w).sum().backward()
        arrs.update({tag + "_x": x, tag + "_w": w, tag + "_y": y, tag + "_dx": x_.grad, tag + "_sigma": sig})
    # anisotropic grid (vox_size_z != -1): separate z kernel
    cfg = make_cfg(vox_size=16, vox_size_z=8, pc_gauss_kernel_size=11)
    g = torch.Generator().manual_seed(299)
    x = torch.rand(2, 1, 8, 16, 16, generator=g).double()
    kz = aniso_kernels(cfg, 1.5)
    y = ref_pc.smoothen_voxels3d(cfg, x, kz)
    arrs.update(dict(z8_x=x, z8_y=y, z8_kxy=kz[0].reshape(-1), z8_kz=kz[2].reshape(-1)))
    save("f4_smooth.npz", **arrs)


# ----------------------------------------------------------------------------------------- F5
def f5_drc():
    cfg = make_cfg(vox_size=8)
    g = torch.Generator().manual_seed(300)
    arrs = {}
    vols = dict(zeros=torch.zeros(2, 8, 8, 8, 1).double(), ones=torch.ones(2, 8, 8, 8, 1).double(),
                rand=torch.rand(2, 8, 8, 8, 1, generator=g).double(),
                # values outside [eps, 1-eps] exercise the clamp masks
                wide=(1.4 * torch.rand(2, 8, 8, 8, 1, generator=g) - 0.2).double())
    w1 = torch.randn(2, 8, 8, 1, generator=g).double()
    w2 = torch.randn(9, 2, 8, 8, 1, generator=g).double()
    w3 = torch.randn(2, 8, 8, 1, generator=g).double()
    arrs.update(w1=w1, w2=w2, w3=w3)
    for tag, v in vols.items():
        v_ = leaf(v)
        proj, p = ref_drc.drc_projection(v_, cfg)
        depth = ref_drc.drc_depth_projection(p, cfg)
        ((proj * w1).sum() + (p * w2).sum() + (depth * w3).sum()).backward()
        arrs.update({tag + "_v": v, tag + "_proj": proj, tag + "_p": p, tag + "_depth": depth, tag + "_dv": v_.grad})
        v2 = leaf(v)
        proj2, _ = ref_drc.drc_projection(v2, cfg)
        (proj2 * w1).sum().backward()
        arrs[tag + "_dv_projonly"] = v2.grad
    arrs["depth_grid_4"] = ref_drc.drc_depth_grid(make_cfg(), 4)
    arrs["depth_grid_64"] = ref_drc.drc_depth_grid(make_cfg(), 64)
    # 64-deep empty ray (SURVEY: 6.398e-4) and full ray
    cfg64 = make_cfg(vox_size=64)
    e, _ = ref_drc.drc_projection(torch.zeros(1, 64, 1, 1, 1).double(), cfg64)
    o, _ = ref_drc.drc_projection(torch.ones(1, 64, 1, 1, 1).double(), cfg64)
    arrs.update(empty_ray64=e, full_ray64=o)
    save("f5_drc.npz", **arrs)


# ----------------------------------------------------------------------------------------- F6
def chain_case(cfg, B, N, G, k, sigma_rel, seed, with_t, with_f, with_s=True, full=True):
    pc, q, s, gt, t, f = synth_inputs(B, N, G, seed, with_t, with_f)
    if not with_s:
        s = None
    if cfg.vox_size_z != -1 and cfg.vox_size_z != cfg.vox_size:
        kernel = aniso_kernels(cfg, sigma_rel)
    else:
        kernel = ref_gk.smoothing_kernel(cfg, sigma_rel)
    arrs = dict(pc=pc, q=q, gt=gt, sigma_rel=sigma_rel, ksize=k, kernel1d=kernel[0].reshape(-1),
                kernel1d_z=kernel[2].reshape(-1))
    if s is not None:
        arrs["s"] = s
    if t is not None:
        arrs["t"] = t
    if f is not None:
        arrs["f"] = f
    for tag, smooth in (("smooth", True), ("literal", False)):
        pc_, q_, s_, t_, f_ = leaf(pc), leaf(q), leaf(s), leaf(t), leaf(f)
        out = ref_chain(cfg, pc_, q_, t_, kernel, s_, f_, smooth)
        loss = ((out["proj"] - gt) ** 2).sum() / B
        loss.backward()
        arrs[tag + "_proj"] = out["proj"]
        arrs[tag + "_loss"] = loss
        arrs[tag + "_dpc"] = pc_.grad
        arrs[tag + "_dq"] = q_.grad
        for nm, x in (("ds", s_), ("dt", t_), ("df", f_)):
            if x is not None:
                arrs[tag + "_" + nm] = x.grad
        arrs[tag + "_proj_depth"] = out["proj_depth"]
        if full:
            arrs[tag + "_voxels"] = out["voxels"].float()  # fp32 storage is ample for a 1e-5 check
            arrs[tag + "_drc_probs"] = out["drc_probs"].float()
            arrs[tag + "_tr_pc"] = out["tr_pc"]
            arrs[tag + "_raw"] = out["raw"].float()
        else:
            v = out["voxels"]
            arrs[tag + "_voxels_zsum"] = v.sum((2, 3, 4))
            arrs[tag + "_voxels_sub"] = v[:, ::4, ::4, ::4, 0]
            arrs[tag + "_voxels_sum"] = v.sum()
            arrs[tag + "_drc_probs_sum"] = out["drc_probs"].sum()
            arrs[tag + "_drc_probs_sub"] = out["drc_probs"][::8, :, ::4, ::4, 0]
            arrs[tag + "_tr_pc"] = out["tr_pc"]
            arrs[tag + "_raw_zsum"] = out["raw"].sum((1, 3, 4))
    return arrs


def f6_chain():
    cfg = make_cfg(vox_size=32, pc_gauss_kernel_size=11)
    save("f6_chain_g32.npz", **chain_case(cfg, 4, 512, 32, 11, 1.5, 1234, False, False))
    save("f6_chain_g32_tf.npz", **chain_case(cfg, 4, 512, 32, 11, 1.5, 1235, True, True))
    save("f6_chain_g32_nos.npz", **chain_case(cfg, 2, 512, 32, 11, 0.8, 1236, False, False, with_s=False))
    cfgz = make_cfg(vox_size=32, vox_size_z=16, pc_gauss_kernel_size=11)
    save("f6_chain_g32z16.npz", **chain_case(cfgz, 2, 512, 32, 11, 1.5, 1237, False, False))
    cfg = make_cfg(vox_size=64, pc_gauss_kernel_size=21)
    save("f6_chain_c1_s3p0.npz", **chain_case(cfg, 1, 8000, 64, 21, 3.0, 1234, False, False, full=False))
    save("f6_chain_c1_s0p64.npz", **chain_case(cfg, 1, 8000, 64, 21, 0.64, 1234, False, False, full=False))
    # also the literal pointcloud_project_fast call on this CPU-only host, to pin "literal == composed(no smoothing)"
    pc, q, s, gt, _, _ = synth_inputs(4, 512, 32, 1234)
    cfg = make_cfg(vox_size=32, pc_gauss_kernel_size=11)
    with quiet():
        out = ref_pc.pointcloud_project_fast(cfg, pc, q, None, None, ref_gk.smoothing_kernel(cfg, 1.5), scaling_factor=s)
    save("f6_literal_call_g32.npz", proj=out["proj"], proj_depth=out["proj_depth"], keys=np.array(sorted(out.keys())))


# ----------------------------------------------------------------------------------------- F7
def f7_scripts():
    """Bodies of dpc/run/pc_project_test.py:48-60 and dpc/run/pc_full_proj_test.py:48-71 (the scripts
    themselves need easydict/tensorboard; their statements are replayed against the reference functions)."""
    vals = {}
    cfg = make_cfg(**yaml.safe_load(open(os.path.join(REF, "experiments/chair_unsupervised/config.yaml"))))
    np.random.seed(0)
    pc = torch.from_numpy(np.random.random((128, 140, 3)))
    pc.requires_grad = True
    with quiet():
        vx = ref_pc.pointcloud2voxels3d_fast(cfg, pc, None)[0]
    loss = torch.sum(vx ** 2) / 2.0
    vx.retain_grad()
    loss.backward()
    vals["pc_project_test"] = dict(loss=loss.item(), output_grads_sum=vx.grad.sum().item(),
                                   input_grads_sum=pc.grad.sum().item(), voxels_sum=vx.sum().item())
    np.random.seed(0)
    cam = torch.from_numpy(np.random.random((128, 4))).float()
    pc = torch.from_numpy(np.random.random((128, 140, 3))).float()
    sc = torch.from_numpy(np.random.random((128, 1))).float()
    kern = ref_gk.smoothing_kernel(cfg, 3.0)  # model.setup_sigma(None, 0) -> pc_relative_sigma = 3.0
    with quiet():
        out = ref_pc.pointcloud_project_fast(cfg, pc, cam, None, None, kern, scaling_factor=sc)
    vals["pc_full_proj_test_literal_cpu"] = {k: out[k].sum().item() for k in
                                             ("proj", "voxels", "tr_pc", "drc_probs", "proj_depth")}
    out = ref_chain(cfg, pc, cam, None, kern, sc, None, True)
    vals["pc_full_proj_test_cuda_semantics"] = {k: out[k].sum().item() for k in
                                                ("proj", "voxels", "tr_pc", "drc_probs", "proj_depth")}
    with open(os.path.join(HERE, "f7_scripts.json"), "w") as fh:
        json.dump(vals, fh, indent=1)
    print("wrote f7_scripts.json", vals["pc_project_test"])


# ----------------------------------------------------------------------------------------- F8 / F9
def f8_f9_model_side():
    from models import model_pc_to as m

    cfg = make_cfg(pose_predict_num_candidates=4)

    class Fake:
        def cfg(self):
            return cfg

    g = torch.Generator().manual_seed(800)
    S, K, G = 6, 4, 16
    gt = (torch.rand(S, G, G, 1, generator=g) > 0.5).double()
    pred = torch.rand(S * K, G, G, 1, generator=g).double().requires_grad_(True)
    loss, argmin = m.ModelPointCloud.proj_loss_pose_candidates(Fake(), gt, pred, {}, None)
    loss.backward()
    save("f8_candidates.npz", gt=gt, pred=pred, loss=loss, argmin=argmin, dpred=pred.grad, K=K)

    sched = {}
    c = make_cfg(**yaml.safe_load(open(os.path.join(REF, "experiments/chair_unsupervised/config.yaml"))))
    for step in (0, 100000, 300000, 600000):
        sched[str(step)] = dict(sigma_rel=float(m.get_smooth_sigma(c, step)), keep_prob=float(m.get_dropout_prob(c, step)))
    with open(os.path.join(HERE, "f9_schedules.json"), "w") as fh:
        json.dump(sched, fh, indent=1)
    print("wrote f9_schedules.json", sched)


def f11_nearest():
    """point_cloud_distance (dpc/util/point_cloud_distance.py:25-40), the kernel of the Chamfer evaluation
    (dpc/run/eval_chamfer_to.py:24-44, 119-123): nearest target point of every source point."""
    import util.point_cloud_distance as ref_pcd

    g = torch.Generator().manual_seed(4242)
    cases = {}

    def add(name, vs, vt):
        proj, dist, idx = ref_pcd.point_cloud_distance(vs, vt)
        cases[name + "_vs"], cases[name + "_vt"] = vs, vt
        cases[name + "_proj"], cases[name + "_dist"], cases[name + "_idx"] = proj, dist, idx

    add("f32", torch.rand(300, 3, generator=g) - 0.5, torch.rand(500, 3, generator=g) - 0.5)
    add("f64", torch.rand(257, 3, generator=g, dtype=torch.float64) - 0.5,
        torch.rand(1000, 3, generator=g, dtype=torch.float64) - 0.5)
    # exact ties: points on a coarse lattice (many equal distances, first minimum must win), duplicates, zero distances
    lat_t = torch.randint(-4, 5, (400, 3), generator=g).float() / 8
    lat_s = torch.randint(-4, 5, (200, 3), generator=g).float() / 8
    add("ties32", lat_s, torch.cat([lat_t, lat_t[:50]]))
    add("ties64", lat_s.double(), torch.cat([lat_t, lat_t[:50]]).double())
    add("one_target", torch.rand(17, 3, generator=g) - 0.5, torch.rand(1, 3, generator=g) - 0.5)
    add("one_source", torch.rand(1, 3, generator=g) - 0.5, torch.rand(33, 3, generator=g) - 0.5)
    # the evaluation's two directed means (eval_chamfer_to.py:119-123) on a pair of clouds
    pred, gt = torch.rand(800, 3, generator=g, dtype=torch.float64) - 0.5, torch.rand(1500, 3, generator=g, dtype=torch.float64) - 0.5
    p2g = ref_pcd.point_cloud_distance(pred, gt)[1]
    g2p = ref_pcd.point_cloud_distance(gt, pred)[1]
    cases["chamfer_pred"], cases["chamfer_gt"] = pred, gt
    cases["chamfer_pair"] = np.array([p2g.numpy().mean(), g2p.numpy().mean()])
    save("f11_nearest.npz", **cases)


def f10_full_step():
    """Tiny-config full training step through the reference's own ModelPointCloud (dpc/models/model_pc_to.py):
    encoder -> decoder / pose candidates / scale -> projection -> min-of-K silhouette loss + student loss -> backward.
    The renderer call inside the model is given the CUDA-branch semantics (ref_chain above: the reference's own
    functions composed in the reference's order), because on this CPU-only host pointcloud_project_fast would skip
    the Gaussian (point_cloud_to.py:206-212).  Stored: config, state_dict, inputs, the outputs a harness needs to
    be checked against (points, poses, scales, silhouettes, winners, loss) and every parameter's gradient."""
    sys.modules.setdefault("nets.net_factory", type(sys)("nets.net_factory")).get_network = None  # TF-era module, unused
    import models.model_pc_to as ref_model

    cfg = make_cfg(**yaml.safe_load(open(os.path.join(REF, "experiments/chair_unsupervised/config.yaml"))))
    tiny = dict(z_dim=64, fc_dim=64, f_dim=8, pc_num_points=256, vox_size=16, pc_gauss_kernel_size=11, batch_size=2, step_size=2,
                input_shape=[32, 32, 3], pc_point_dropout=1.0, pc_relative_sigma=1.5, align_to_canonical=False)
    cfg.update(tiny)

    def project(cfg_, pc, q, t, rgb, kernel, scaling_factor=None, focal_length=None):
        out = ref_chain(cfg_, pc, q, t, kernel, scaling_factor, focal_length, smooth=True)
        out["proj_rgb"] = None
        return out

    ref_model.pointcloud_project_fast = project
    torch.manual_seed(77)
    np.random.seed(77)
    model = ref_model.ModelPointCloud(cfg)
    with torch.no_grad():  # spread the pose candidates (at the reference's init they nearly coincide and candidate 0 always wins)
        for prm in model.poseNet.parameters():
            prm.add_(0.5 * torch.randn(prm.shape, generator=torch.Generator().manual_seed(prm.numel())))
        # ... and the decoded shape away from the origin (an anisotropic cloud, so that the pose matters)
        b = model.decoder.pts_raw_fc.bias
        b.add_((torch.randn(b.shape, generator=torch.Generator().manual_seed(5)).reshape(-1, 3) * torch.tensor([0.9, 0.5, 0.2])).reshape(-1))
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(78)
    nimg = cfg.batch_size * cfg.step_size
    images = torch.rand(nimg, 3, 32, 32, generator=g)
    masks = (torch.rand(nimg, 1, 32, 32, generator=g) > 0.5).float()
    inputs = dict(images=images, masks=masks.clone(), images_1=images[::cfg.step_size])
    with quiet():
        outputs = model(inputs, 0, is_training=True, run_projection=True)
        loss, min_loss = model.get_loss(inputs, outputs, add_summary=False)
    loss.backward()
    arrays = {"state/" + k: v for k, v in state.items()}
    arrays.update({"grad/" + k: p.grad for k, p in model.named_parameters() if p.grad is not None})
    arrays["no_grad"] = np.array([k for k, p in model.named_parameters() if p.grad is None])
    arrays.update(images=images, masks=masks, points_1=outputs["points_1"], poses=outputs["poses"],
                  pose_student=outputs["pose_student"], scaling_factor=outputs["scaling_factor"], projs=outputs["projs"],
                  pooled_masks=inputs["masks"], loss=loss.detach(), min_loss=min_loss)
    save("f10_full_step.npz", **arrays)
    json.dump({k: (v if not isinstance(v, np.generic) else v.item()) for k, v in cfg.items()},
              open(os.path.join(HERE, "f10_config.json"), "w"), indent=0, sort_keys=True, default=str)
    print("wrote f10_config.json; loss", float(loss), "winners", min_loss.tolist())


def f12_view_sampler():
    """ModelBase.preprocess (dpc/models/model_base_to.py:23-109): which views of which objects a step trains on, and the
    tensors it selects.  Three cases on one set of raw inputs (3 objects x 5 views): random views, the first views in
    order, and a variable number of views per object (one object has fewer views than the step needs: padded with view 0
    and marked invalid).  The numpy RNG is seeded right before every call; stored: the seed, the raw inputs, the outputs."""
    import models.model_base_to as ref_base

    g = torch.Generator().manual_seed(1201)
    raw = dict(image=torch.rand(3, 5, 3, 8, 8, generator=g), mask=(torch.rand(3, 5, 1, 8, 8, generator=g) > 0.5).float(),
               extrinsic=torch.randn(3, 5, 4, 4, generator=g), cam_pos=torch.randn(3, 5, 3, generator=g),  # torch: the
               # reference's select_2d (list-of-index-arrays) no longer means the same thing on numpy-2 arrays
               num_views=torch.tensor([[5.0], [1.0], [3.0]]))
    arrays = dict(image=raw["image"], mask=raw["mask"], extrinsic=raw["extrinsic"], num_views=raw["num_views"], seed=np.array(1202))
    for tag, var, rnd in (("random", False, True), ("ordered", False, False), ("variable", True, True)):
        cfg = make_cfg(batch_size=3, step_size=2, num_views_to_use=-1, variable_num_views=var, saved_depth=False, saved_camera=True)
        np.random.seed(1202)
        with quiet():
            out = ref_base.ModelBase(cfg).preprocess(raw, cfg.step_size, random_views=rnd)
        for k in ("images", "masks", "valid_samples", "images_1", "matrices"):
            arrays[tag + "/" + k] = out[k]
    save("f12_view_sampler.npz", **arrays)


if __name__ == "__main__":
    torch.set_num_threads(8)
    if len(sys.argv) > 1:  # regenerate only the named groups, e.g. `make_golden.py f11_nearest`
        for name in sys.argv[1:]:
            globals()[name]()
        assert not os.path.exists(os.path.join(REF, "dpc/util/__pycache__")), "left bytecode in the reference"
        sys.exit(0)
    f1_gauss()
    f2_transform()
    f3_splat()
    f4_smooth()
    f5_drc()
    f6_chain()
    f7_scripts()
    f8_f9_model_side()
    f11_nearest()
    f10_full_step()
    f12_view_sampler()
    assert not os.path.exists(os.path.join(REF, "dpc/util/__pycache__")), "left bytecode in the reference"
