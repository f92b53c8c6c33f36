"""Pin the CPU oracle (oracle/dpc_oracle.py) against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only; fp64; tolerances are rounding-level because the oracle uses the
same operator classes as the reference."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import dpc_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
T = lambda a: torch.from_numpy(np.asarray(a))


def close(a, b, tol=1e-12):
    a = a.detach().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b).max() if a.size else 0.0
    assert err <= tol, "max abs err %.3e > %.1e" % (err, tol)


def leaf(x):
    return None if x is None else x.clone().requires_grad_(True)


# ------------------------------------------------------------------------------------------ F1
@pytest.mark.parametrize("l,sig", [(11, 1.0), (21, 3.0), (21, 0.64), (21, 0.2), (11, 0.2), (10, 1.5)])
def test_gauss_kernel_1d(golden, l, sig):
    ref = golden("f1_gauss.npz")["k_%d_%s" % (l, str(sig).replace(".", "p"))]
    k = O.gauss_kernel_1d(l, sig)
    assert k.dtype == torch.float32
    assert np.array_equal(k.numpy(), ref)  # bit-exact: same fp32 op sequence


def test_gauss_known_answer():
    k = O.gauss_kernel_1d(11, 1.0).numpy()
    np.testing.assert_allclose(k[:6], [1.4867e-6, 1.3383e-4, 4.4318e-3, 5.3991e-2, 0.24197, 0.39894], rtol=2e-4)


def test_smoothing_kernel_shapes(golden):
    g = golden("f1_gauss.npz")
    ks = O.smoothing_kernel(O.Cfg(pc_gauss_kernel_size=21, vox_size=64), 3.0)
    for i, k in enumerate(ks):
        assert np.array_equal(k.numpy(), g["sk64_%d" % i])
    ks = O.smoothing_kernel(O.Cfg(pc_gauss_kernel_size=11, vox_size=32, vox_size_z=32), 1.5)
    for i, k in enumerate(ks):
        assert np.array_equal(k.numpy(), g["skz_%d" % i])


# ------------------------------------------------------------------------------------------ F2
@pytest.mark.parametrize("tag", ["plain", "t", "f", "tf"])
def test_transform(golden, tag):
    g = golden("f2_transform.npz")
    cfg = O.Cfg()
    pc, q = leaf(T(g["pc"])), leaf(T(g["q"]))
    t = leaf(T(g["t"])) if "t" in tag else None
    f = leaf(T(g["f"])) if "f" in tag else None
    out = O.pc_perspective_transform(cfg, pc, q, t, f)
    assert out.dtype == torch.float64
    close(out, g["out_" + tag], 1e-13)
    (out * T(g["w"])).sum().backward()
    close(pc.grad, g["dpc_" + tag], 1e-6)  # grads arrive in fp32 (input dtype)
    close(q.grad, g["dq_" + tag], 2e-5)
    if t is not None:
        close(t.grad, g["dt_" + tag], 2e-5)
    if f is not None:
        close(f.grad, g["df_" + tag], 2e-5)


def test_transform_identity_and_rotate(golden):
    g = golden("f2_transform.npz")
    out = O.pc_perspective_transform(O.Cfg(), torch.tensor([[[0.1, 0.2, 0.3]]]), torch.tensor([[1.0, 0, 0, 0]]))
    close(out, g["identity_out"], 1e-15)
    np.testing.assert_allclose(out.numpy().ravel(), [0.1, 0.178571, 0.267857], atol=1e-6)
    close(O.quaternion_rotate(T(g["pc"]), T(g["q"])), g["rot"], 1e-14)


# ------------------------------------------------------------------------------------------ F3
@pytest.mark.parametrize("tag,G,Gz", [("g16", 16, -1), ("g32", 32, -1), ("g16z8", 16, 8)])
def test_splat(golden, tag, G, Gz):
    g = golden("f3_splat.npz")
    cfg = O.Cfg(vox_size=G, vox_size_z=Gz)
    pc = leaf(T(g[tag + "_pc"]))
    vox, none = O.pointcloud2voxels3d_fast(cfg, pc, None)
    assert none is None
    close(vox, g[tag + "_vox"], 1e-13)
    assert abs(vox.sum().item() - int(g[tag + "_nvalid"])) < 1e-9  # mass == number of in-bounds points
    (vox * T(g[tag + "_w"])).sum().backward()
    close(pc.grad, g[tag + "_dpc"], 1e-11)


def test_splat_plus_half_raises():
    with pytest.raises(IndexError):
        O.pointcloud2voxels3d_fast(O.Cfg(vox_size=8), torch.tensor([[[0.5, 0.0, 0.0]]], dtype=torch.float64))
    v, _ = O.pointcloud2voxels3d_fast(O.Cfg(vox_size=8), torch.tensor([[[-0.5, 0.0, 0.0]]], dtype=torch.float64))
    assert abs(v.sum().item() - 1.0) < 1e-12


# ------------------------------------------------------------------------------------------ F4
@pytest.mark.parametrize("tag,k", [("k11", 11), ("k21", 21), ("k21s", 21)])
def test_smooth(golden, tag, k):
    g = golden("f4_smooth.npz")
    cfg = O.Cfg(vox_size=16, pc_gauss_kernel_size=k)
    x = leaf(T(g[tag + "_x"]))
    y = O.smoothen_voxels3d(cfg, x, O.smoothing_kernel(cfg, float(g[tag + "_sigma"])))
    close(y, g[tag + "_y"], 1e-14)
    (y * T(g[tag + "_w"])).sum().backward()
    close(x.grad, g[tag + "_dx"], 1e-13)


def test_smooth_anisotropic(golden):
    g = golden("f4_smooth.npz")
    cfg = O.Cfg(vox_size=16, vox_size_z=8, pc_gauss_kernel_size=11)
    ks = O.smoothing_kernel(cfg, 1.5)
    assert np.array_equal(ks[0].reshape(-1).numpy(), g["z8_kxy"]) and np.array_equal(ks[2].reshape(-1).numpy(), g["z8_kz"])
    close(O.smoothen_voxels3d(cfg, T(g["z8_x"]), ks), g["z8_y"], 1e-14)


# ------------------------------------------------------------------------------------------ F5
@pytest.mark.parametrize("tag", ["zeros", "ones", "rand", "wide"])
def test_drc(golden, tag):
    g = golden("f5_drc.npz")
    cfg = O.Cfg(vox_size=8)
    v = leaf(T(g[tag + "_v"]))
    proj, p = O.drc_projection(v, cfg)
    depth = O.drc_depth_projection(p, cfg)
    close(proj, g[tag + "_proj"], 1e-14)
    close(p, g[tag + "_p"], 1e-14)
    close(depth, g[tag + "_depth"], 1e-13)
    ((proj * T(g["w1"])).sum() + (p * T(g["w2"])).sum() + (depth * T(g["w3"])).sum()).backward()
    close(v.grad, g[tag + "_dv"], 1e-9)
    v2 = leaf(T(g[tag + "_v"]))
    (O.drc_projection(v2, cfg)[0] * T(g["w1"])).sum().backward()
    close(v2.grad, g[tag + "_dv_projonly"], 1e-10)


def test_drc_known_answers(golden):
    g = golden("f5_drc.npz")
    cfg = O.Cfg()
    close(O.drc_depth_grid(cfg, 4), [1.5, 1.75, 2.0, 2.25, 10.0], 1e-15)
    close(O.drc_depth_grid(cfg, 64), g["depth_grid_64"], 1e-15)
    e, _ = O.drc_projection(torch.zeros(1, 64, 1, 1, 1), cfg)
    o, _ = O.drc_projection(torch.ones(1, 64, 1, 1, 1), cfg)
    close(e, g["empty_ray64"], 1e-15)
    close(o, g["full_ray64"], 1e-15)
    assert abs(e.item() - 6.398e-4) < 1e-7 and abs(o.item() - 1.00001) < 1e-7


# ------------------------------------------------------------------------------------------ F6
def _chain_cfg(name):
    if "c1" in name:
        return O.Cfg(vox_size=64, pc_gauss_kernel_size=21)
    if "z16" in name:
        return O.Cfg(vox_size=32, vox_size_z=16, pc_gauss_kernel_size=11)
    return O.Cfg(vox_size=32, pc_gauss_kernel_size=11)


@pytest.mark.parametrize("name", ["f6_chain_g32.npz", "f6_chain_g32_tf.npz", "f6_chain_g32_nos.npz",
                                  "f6_chain_g32z16.npz", "f6_chain_c1_s3p0.npz", "f6_chain_c1_s0p64.npz"])
@pytest.mark.parametrize("sem", ["smooth", "literal"])
def test_chain(golden, name, sem):
    g = golden(name)
    cfg = _chain_cfg(name)
    kern = O.smoothing_kernel(cfg, float(g["sigma_rel"]))
    assert np.array_equal(kern[0].reshape(-1).numpy(), g["kernel1d"])
    assert np.array_equal(kern[2].reshape(-1).numpy(), g["kernel1d_z"])
    pc, q = leaf(T(g["pc"])), leaf(T(g["q"]))
    s = leaf(T(g["s"])) if "s" in g else None
    t = leaf(T(g["t"])) if "t" in g else None
    f = leaf(T(g["f"])) if "f" in g else None
    out = O.pointcloud_project_fast(cfg, pc, q, t, None, kern, scaling_factor=s, focal_length=f, smooth=(sem == "smooth"))
    close(out["proj"], g[sem + "_proj"], 1e-12)
    close(out["proj_depth"], g[sem + "_proj_depth"], 1e-11)
    close(out["tr_pc"], g[sem + "_tr_pc"], 1e-13)
    B = pc.shape[0]
    loss = ((out["proj"] - T(g["gt"])) ** 2).sum() / B
    close(loss, g[sem + "_loss"], 1e-11)
    loss.backward()
    close(pc.grad, g[sem + "_dpc"], 1e-7)
    close(q.grad, g[sem + "_dq"], 1e-5)
    for nm, x in (("ds", s), ("dt", t), ("df", f)):
        if x is not None:
            close(x.grad, g[sem + "_" + nm], 1e-5)
    if sem + "_voxels" in g:
        close(out["voxels"], g[sem + "_voxels"], 1e-6)  # fixture stored in fp32
        close(out["drc_probs"], g[sem + "_drc_probs"], 1e-6)
        close(out["voxels_raw"].unsqueeze(1), g[sem + "_raw"], 1e-6)
    else:
        close(out["voxels"].sum((2, 3, 4)), g[sem + "_voxels_zsum"], 1e-10)
        close(out["voxels"][:, ::4, ::4, ::4, 0], g[sem + "_voxels_sub"], 1e-13)
        close(out["drc_probs"][::8, :, ::4, ::4, 0], g[sem + "_drc_probs_sub"], 1e-13)
        close(out["voxels_raw"].sum((2, 3)), g[sem + "_raw_zsum"].reshape(B, -1), 1e-10)


@pytest.mark.parametrize("name", ["f6_chain_g32.npz", "f6_chain_g32_tf.npz", "f6_chain_g32_nos.npz",
                                  "f6_chain_g32z16.npz", "f6_chain_c1_s3p0.npz", "f6_chain_c1_s0p64.npz"])
def test_exact_pose_gradient_mode_is_pinned(golden, name):
    """EXACT_POSE_GRADIENT = True is the variant every fresh-seed d(q) check of tests/test_gpu_parity.py and smoke() hold
    the device to.  It is a custom autograd.Function, so it is pinned here against the SAME golden vectors as the default
    mode: (i) the forward is the default mode's bit for bit (and therefore the reference's, to the 1e-12 of test_chain);
    (ii) d(q) moves away from the reference's own by no more than the parity rule 1e-5 * max(1, max|dq|) -- the size of the
    reference's fp32 summation noise over N points (measured on these fixtures: 4.4e-6 at |dq| 1.2 and N = 512 ...
    9.2e-5 at |dq| 304 and N = 8000, i.e. 0.3-3.6 ppm of the scale); (iii) every other gradient (d(pc), d(s), d(t), d(f))
    stays within 1e-6 * max(1, max|.|) (fp32 rounding of d(pc) in a different order; ds, dt, df identical).
    Reference: dpc/util/quaternion.py:69-86, 110-132."""
    g = golden(name)
    cfg = _chain_cfg(name)
    kern = O.smoothing_kernel(cfg, float(g["sigma_rel"]))
    res = {}
    assert O.EXACT_POSE_GRADIENT is False, "another test left the oracle in exact-sum mode"
    try:
        for exact in (False, True):
            O.EXACT_POSE_GRADIENT = exact
            pc, q = leaf(T(g["pc"])), leaf(T(g["q"]))
            s = leaf(T(g["s"])) if "s" in g else None
            t = leaf(T(g["t"])) if "t" in g else None
            f = leaf(T(g["f"])) if "f" in g else None
            out = O.pointcloud_project_fast(cfg, pc, q, t, None, kern, scaling_factor=s, focal_length=f)
            (((out["proj"] - T(g["gt"])) ** 2).sum() / pc.shape[0]).backward()
            res[exact] = dict(proj=out["proj"].detach(), tr_pc=out["tr_pc"].detach(), depth=out["proj_depth"].detach(),
                              dpc=pc.grad, dq=q.grad, ds=None if s is None else s.grad, dt=None if t is None else t.grad,
                              df=None if f is None else f.grad)
    finally:
        O.EXACT_POSE_GRADIENT = False
    raw, exact = res[False], res[True]
    for k in ("proj", "tr_pc", "depth"):
        assert torch.equal(raw[k], exact[k]), "exact-sum mode changed the forward value of " + k
    close(exact["proj"], g["smooth_proj"], 1e-12)
    scale = lambda x: max(1.0, float(np.abs(np.asarray(x)).max()))
    close(exact["dq"], g["smooth_dq"], 1e-5 * scale(g["smooth_dq"]))
    close(exact["dpc"], g["smooth_dpc"], 1e-6 * scale(g["smooth_dpc"]))
    for nm in ("ds", "dt", "df"):
        if exact[nm] is not None:
            close(exact[nm], g["smooth_" + nm], 1e-6 * scale(g["smooth_" + nm]))
    # and the mode does something: the reference's own fp32 sum is visibly not the exact one at N = 8000
    if "c1" in name:
        assert (raw["dq"].double() - exact["dq"].double()).abs().max().item() > 1e-6


def test_literal_cpu_call_matches_no_smoothing(golden):
    """What pointcloud_project_fast returns on a CPU-only host == the chain with the Gaussian skipped."""
    g, h = golden("f6_literal_call_g32.npz"), golden("f6_chain_g32.npz")
    close(g["proj"], h["literal_proj"], 0.0)
    assert sorted(g["keys"].tolist()) == sorted(["proj", "voxels", "tr_pc", "voxels_rgb", "proj_rgb", "drc_probs", "proj_depth"])


# ------------------------------------------------------------------------------------------ F7
def test_reference_script_bodies():
    vals = json.load(open(os.path.join(GOLDEN, "f7_scripts.json")))
    cfg = O.Cfg(vox_size=64, pc_gauss_kernel_size=21, pc_relative_sigma=3.0)
    np.random.seed(0)
    pc = torch.from_numpy(np.random.random((128, 140, 3))).requires_grad_(True)
    vx = O.pointcloud2voxels3d_fast(cfg, pc, None)[0]
    vx.retain_grad()
    loss = torch.sum(vx ** 2) / 2.0
    loss.backward()
    ref = vals["pc_project_test"]
    assert abs(loss.item() - ref["loss"]) < 1e-9 and abs(loss.item() - 337.6612128968) < 1e-8
    assert abs(vx.grad.sum().item() - 2269.0) < 1e-9 and abs(pc.grad.sum().item() - ref["input_grads_sum"]) < 1e-8

    np.random.seed(0)
    cam = torch.from_numpy(np.random.random((128, 4))).float()
    pc = torch.from_numpy(np.random.random((128, 140, 3))).float()
    sc = torch.from_numpy(np.random.random((128, 1))).float()
    kern = O.smoothing_kernel(cfg, O.get_smooth_sigma(cfg, 0))
    for sem, key in ((False, "pc_full_proj_test_literal_cpu"), (True, "pc_full_proj_test_cuda_semantics")):
        out = O.pointcloud_project_fast(cfg, pc, cam, None, None, kern, scaling_factor=sc, smooth=sem)
        for k, v in vals[key].items():
            assert abs(out[k].sum().item() - v) < 1e-7 * max(1.0, abs(v)), (key, k)
    lit = vals["pc_full_proj_test_literal_cpu"]
    assert abs(lit["proj"] - 1683.9716113117) < 1e-6 and abs(lit["proj_depth"] - 5229753.9055522671) < 1e-3


# ------------------------------------------------------------------------------------------ F8 / F9
def test_candidate_loss(golden):
    g = golden("f8_candidates.npz")
    pred = leaf(T(g["pred"]))
    loss, win = O.proj_loss_pose_candidates(T(g["gt"]), pred, int(g["K"]))
    close(loss, g["loss"], 1e-12)
    assert np.array_equal(win.numpy(), g["argmin"])
    loss.backward()
    close(pred.grad, g["dpred"], 1e-14)


def test_schedules():
    sched = json.load(open(os.path.join(GOLDEN, "f9_schedules.json")))
    cfg = O.Cfg(pc_relative_sigma=3.0, pc_point_dropout=0.07)
    for step, v in sched.items():
        assert abs(O.get_smooth_sigma(cfg, int(step)) - v["sigma_rel"]) < 1e-12
        assert abs(O.get_dropout_prob(cfg, int(step)) - v["keep_prob"]) < 1e-12


NEAREST_CASES = ["f32", "f64", "ties32", "ties64", "one_target", "one_source"]


@pytest.mark.parametrize("name", NEAREST_CASES)
def test_nearest_point(golden, name):
    """F11: point_cloud_distance of the reference (point_cloud_distance.py:25-40).  Indices are index work: exact,
    including the lattice cases full of ties (first minimum).  Distances: the reference's torch-CPU sqrt is a
    vectorised approximation, the oracle's is correctly rounded -- at most 1 ulp apart."""
    g = golden("f11_nearest.npz")
    vs, vt = torch.from_numpy(g[name + "_vs"]), torch.from_numpy(g[name + "_vt"])
    proj, dist, idx = O.point_cloud_distance(vs, vt)
    assert idx.dtype == torch.int64 and dist.dtype == vs.dtype
    assert np.array_equal(idx.numpy(), g[name + "_idx"])
    assert np.array_equal(proj.numpy(), g[name + "_proj"])
    ulp = np.spacing(np.abs(g[name + "_dist"]).astype(g[name + "_dist"].dtype))
    assert (np.abs(dist.numpy() - g[name + "_dist"]) <= ulp).all()


def test_chamfer_pair(golden):
    """The evaluation's two directed means (eval_chamfer_to.py:119-123)."""
    g = golden("f11_nearest.npz")
    pred, gt = g["chamfer_pred"], g["chamfer_gt"]
    p2g = O.point_cloud_distance(pred, gt)[1].numpy().mean()
    g2p = O.point_cloud_distance(gt, pred)[1].numpy().mean()
    assert np.allclose([p2g, g2p], g["chamfer_pair"], rtol=1e-14, atol=0)
