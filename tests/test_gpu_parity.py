"""GPU parity: the HIP path (through the C ABI) against golden vectors produced by the reference and
against the CPU oracle on the same seeded inputs.

THE parity rule (one rule for every silhouette, loss and gradient in this file):

    max |device - reference|  <=  1e-5 * max(1, max |reference|)

i.e. BASELINE.json's "1e-5 fp32" as an absolute bound for O(1) quantities (silhouettes, image-loss gradients) and relative
to the largest reference entry where the values are O(10..1000) (losses summed over 4096 pixels, gradients under the
fixtures' synthetic weights): fp32 device results have 24 significant bits, the references are fp64.  Tighter bounds are
used where the arithmetic allows them (bit-exact cell records and indices, 2e-6 on transformed coordinates, 1e-6 between
two device paths).

The reference of the fresh-seed d(q) checks is the oracle with EXACT_POSE_GRADIENT (oracle/dpc_oracle.py): the reference's forward
bit for bit, its gradient w.r.t. the pose quaternion with the sum over the points taken in fp64.  torch takes that sum in fp32,
over a vector whose radial part (|.| ~ 100..1000) cancels afterwards: the reference's OWN d(q) scatters by up to the size of the
rule around its exact value (1.6e-5 at N = 1300, |d(q)| ~ 1), so a device value can be held to the rule only against the exact
sum.  That oracle variant is pinned on CPU against the golden vectors (tests/test_oracle_golden.py::
test_exact_pose_gradient_mode_is_pinned); the golden-vector tests here (F2, F6, F10) compare d(q) with the RAW reference's stored
values; and test_pose_gradient_against_the_reference_and_its_exact_sum measures all three distances in the case that sat at the
rule: device - exact (<= rule), device - raw (<= rule + the raw reference's own deviation), raw - exact."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-5
ERRORS = []  # (what, max abs err, scale) of every comparison, dumped by test_zz_error_report
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def R():
    import dpc.render as R

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return R


@pytest.fixture(scope="module")
def O():
    from oracle import dpc_oracle as O

    O.EXACT_POSE_GRADIENT = True   # see the module docstring; the golden-vector tests use stored reference values, not this
    yield O
    O.EXACT_POSE_GRADIENT = False


def dev(a, grad=False, dtype=torch.float32):
    if a is None:
        return None
    t = torch.as_tensor(np.asarray(a)).to(device="cuda", dtype=dtype)
    return t.requires_grad_(True) if grad else t


def close(a, b, tol=TOL, what=""):
    """max |a-b| <= tol * max(1, max|b|): absolute 1e-5 for O(1) quantities (silhouettes, image-loss gradients),
    relative to the largest reference entry where a fixture's synthetic weights make the values O(10..100)."""
    a = a.detach().double().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = b.detach().double().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.isfinite(a).all(), what + ": non-finite values"
    err = float(np.abs(a - b).max()) if a.size else 0.0
    scale = max(1.0, float(np.abs(b).max())) if b.size else 1.0
    ERRORS.append((what, err, scale))
    assert err <= tol * scale, "%s: max abs err %.3e > %.1e * %.2f" % (what, err, tol, scale)


def cfg_for(O, **kw):
    return O.Cfg(**kw)


# ------------------------------------------------------------------------------------------ stage: transform
@pytest.mark.parametrize("tag", ["plain", "t", "f", "tf"])
def test_transform_golden(R, O, golden, tag):
    g = golden("f2_transform.npz")
    pc, q = dev(g["pc"], True), dev(g["q"], True)
    t = dev(g["t"], True) if "t" in tag else None
    f = dev(g["f"], True) if "f" in tag else None
    out = R.pc_perspective_transform(cfg_for(O), pc, q, t, f)
    close(out, g["out_" + tag], 2e-6, "tr_pc")
    (out * dev(g["w"])).sum().backward()
    close(pc.grad, g["dpc_" + tag], TOL, "dpc")
    close(q.grad, g["dq_" + tag], TOL, "dq")  # sums of 64 O(1) terms in fp32
    if t is not None:
        close(t.grad, g["dt_" + tag], TOL, "dt")
    if f is not None:
        close(f.grad, g["df_" + tag], TOL, "df")


def test_transform_identity(R, O):
    out = R.pc_perspective_transform(cfg_for(O), dev([[[0.1, 0.2, 0.3]]]), dev([[1.0, 0, 0, 0]]))
    close(out, [[[0.1, 0.2 * 1.875 / 2.1, 0.3 * 1.875 / 2.1]]], 1e-7)


def oracle_records(O, cfg, pc, q, t, f):
    """Point records (cell code + encoded fractions) from the oracle's fp64 transformed cloud."""
    tr = O.pc_perspective_transform(cfg, pc, q, t, f).numpy()  # fp64, bit-identical to the reference on CPU
    D, H, W = O.grid_dims(cfg)
    valid = ((tr >= -0.5) & (tr <= 0.5)).all(-1)
    g = (tr + 0.5) * (np.array([D, H, W], dtype=np.float64) - 1.0)
    fl = np.floor(g)
    r = g - fl
    enc = np.where(r < 0.5, r, r - 1.0).astype(np.float32)
    cell = fl.astype(np.int64)
    code = np.where(valid, (cell[..., 0] << 20) | (cell[..., 1] << 10) | cell[..., 2], -1).astype(np.int32)
    enc[~valid] = 0.0
    return tr, code, enc


@pytest.mark.parametrize("with_t,with_f", [(False, False), (True, True)])
def test_locate_bit_exact(R, O, with_t, with_f):
    """Integer/bit-level parity of the first launch: cell indices, validity and the encoded fractional
    weights equal those derived from the reference's (oracle's) fp64 transform, for every point."""
    from dpc.render._ops import locate_points, decode_cells
    from dpc.render import _geometry

    cfg = O.Cfg(vox_size=64)
    pc, q, _, _, t, f = O.synth_inputs(8, 8000, 64, 2024, with_t, with_f)
    pc = pc * 1.3  # push ~10% of the points out of bounds
    tr_ref, code, enc = oracle_records(O, cfg, pc, q, t, f)
    tr, cells = locate_points(dev(pc), dev(q), dev(t), dev(f), _geometry(cfg))
    got_code, got_enc, got_pts = decode_cells(cells, 8, 8000, 64)  # also checks the per-chunk z sort and bin offsets
    assert np.array_equal(got_pts, pc.numpy()), "points carried next to the records differ from the input cloud"
    assert np.array_equal(got_code, code), "cell index / validity differs for %d points" % (got_code != code).sum()
    assert np.array_equal(got_enc, enc), "encoded fractions differ"
    assert np.array_equal(tr.cpu().numpy(), tr_ref.astype(np.float32)), "tr_pc is not the fp32 rounding of the reference's"
    assert 0.02 < (code < 0).mean() < 0.5


# ------------------------------------------------------------------------------------------ stage: splat
@pytest.mark.parametrize("tag,G,Gz", [("g16", 16, -1), ("g32", 32, -1), ("g16z8", 16, 8)])
def test_splat_golden(R, O, golden, tag, G, Gz):
    g = golden("f3_splat.npz")
    cfg = cfg_for(O, vox_size=G, vox_size_z=Gz)
    pc = dev(g[tag + "_pc"], True, torch.float64)  # the reference's direct callers pass fp64
    vox, none = R.pointcloud2voxels3d_fast(cfg, pc, None)
    assert none is None and vox.dtype == torch.float32
    close(vox, g[tag + "_vox"], TOL, "voxels")
    assert abs(vox.sum().item() - int(g[tag + "_nvalid"])) < 1e-2
    (vox * dev(g[tag + "_w"])).sum().backward()
    assert pc.grad.dtype == torch.float64
    close(pc.grad, g[tag + "_dpc"], TOL, "dpc")


def test_splat_script_body(R, O):
    """dpc/run/pc_project_test.py:48-60 on the device: loss 337.66, 2269 valid points."""
    vals = json.load(open(os.path.join(GOLDEN, "f7_scripts.json")))["pc_project_test"]
    np.random.seed(0)
    pc = torch.from_numpy(np.random.random((128, 140, 3))).cuda().requires_grad_(True)
    vx = R.pointcloud2voxels3d_fast(cfg_for(O, vox_size=64), pc, None)[0]
    vx.retain_grad()
    loss = torch.sum(vx ** 2) / 2.0
    loss.backward()
    assert abs(loss.item() - vals["loss"]) < 1e-3
    assert abs(vx.sum().item() - 2269.0) < 1e-2 and abs(vx.grad.sum().item() - 2269.0) < 1e-2
    assert abs(pc.grad.sum().item() - vals["input_grads_sum"]) < 2e-2


def test_splat_edges(R, O):
    cfg = cfg_for(O, vox_size=8)
    # +1/2 face: the reference raises IndexError; here the in-range corner keeps the whole weight
    v, _ = R.pointcloud2voxels3d_fast(cfg, dev([[[0.5, 0.0, 0.0], [-0.5, -0.5, -0.5], [0.6, 0.0, 0.0]]]), None)
    assert abs(v.sum().item() - 2.0) < 1e-6 and abs(v[0, 0, 0, 0].item() - 1.0) < 1e-6
    # all points outside -> empty grid; zero points -> empty grid
    v, _ = R.pointcloud2voxels3d_fast(cfg, dev(np.full((2, 5, 3), 0.7)), None)
    assert v.abs().sum().item() == 0.0
    v, _ = R.pointcloud2voxels3d_fast(cfg, torch.zeros(2, 0, 3, device="cuda"), None)
    assert v.shape == (2, 8, 8, 8) and v.abs().sum().item() == 0.0


# ------------------------------------------------------------------------------------------ stage: smoothing
@pytest.mark.parametrize("tag,k", [("k11", 11), ("k21", 21), ("k21s", 21)])
def test_smooth_golden(R, O, golden, tag, k):
    g = golden("f4_smooth.npz")
    cfg = cfg_for(O, vox_size=16, pc_gauss_kernel_size=k)
    x = dev(g[tag + "_x"], True)
    y = R.smoothen_voxels3d(cfg, x, R.smoothing_kernel(cfg, float(g[tag + "_sigma"])))
    close(y, g[tag + "_y"], 2e-6, "smoothed")
    (y * dev(g[tag + "_w"])).sum().backward()
    close(x.grad, g[tag + "_dx"], TOL, "dx")


def test_smooth_anisotropic(R, O, golden):
    g = golden("f4_smooth.npz")
    cfg = cfg_for(O, vox_size=16, vox_size_z=8, pc_gauss_kernel_size=11)
    close(R.smoothen_voxels3d(cfg, dev(g["z8_x"]), R.smoothing_kernel(cfg, 1.5)), g["z8_y"], 2e-6)


# ------------------------------------------------------------------------------------------ stage: DRC
@pytest.mark.parametrize("tag", ["zeros", "ones", "rand", "wide"])
def test_drc_golden(R, O, golden, tag):
    g = golden("f5_drc.npz")
    cfg = cfg_for(O, vox_size=8)
    v = dev(g[tag + "_v"], True)
    proj, p = R.drc_projection(v, cfg)
    depth = R.drc_depth_projection(p, cfg)
    close(proj, g[tag + "_proj"], 2e-6, "proj")
    close(p, g[tag + "_p"], 2e-6, "probs")
    close(depth, g[tag + "_depth"], TOL, "depth")  # values up to max_depth = 10
    ((proj * dev(g["w1"])).sum() + (p * dev(g["w2"])).sum() + (depth * dev(g["w3"])).sum()).backward()
    close(v.grad, g[tag + "_dv"], TOL, "dv (all outputs)")  # depth weights reach 10, gradients O(30)
    v2 = dev(g[tag + "_v"], True)
    (R.drc_projection(v2, cfg)[0] * dev(g["w1"])).sum().backward()
    close(v2.grad, g[tag + "_dv_projonly"], TOL, "dv (proj only)")
    close(R.drc_event_probabilities(dev(g[tag + "_v"]), cfg), g[tag + "_p"], 2e-6, "event probs")


def test_drc_rays(R, O, golden):
    g = golden("f5_drc.npz")
    cfg = cfg_for(O)
    e, _ = R.drc_projection(torch.zeros(1, 64, 1, 1, 1, device="cuda"), cfg)
    o, _ = R.drc_projection(torch.ones(1, 64, 1, 1, 1, device="cuda"), cfg)
    close(e, g["empty_ray64"], 1e-7)
    close(o, g["full_ray64"], 1e-6)


# ------------------------------------------------------------------------------------------ fused chain
def _chain_cfg(O, name):
    if "c1" in name:
        return O.Cfg(vox_size=64, pc_gauss_kernel_size=21)
    if "z16" in name:
        return O.Cfg(vox_size=32, vox_size_z=16, pc_gauss_kernel_size=11)
    return O.Cfg(vox_size=32, pc_gauss_kernel_size=11)


CHAINS = ["f6_chain_g32.npz", "f6_chain_g32_tf.npz", "f6_chain_g32_nos.npz", "f6_chain_g32z16.npz",
          "f6_chain_c1_s3p0.npz", "f6_chain_c1_s0p64.npz"]


@pytest.mark.parametrize("name", CHAINS)
@pytest.mark.parametrize("sem", ["smooth", "literal"])
def test_chain_golden(R, O, golden, name, sem):
    """pointcloud_project_fast (fused kernels) vs the reference: silhouette and d(pc), d(q), d(s), d(t), d(f)."""
    g = golden(name)
    cfg = _chain_cfg(O, name)
    kern = R.smoothing_kernel(cfg, float(g["sigma_rel"]))
    pc, q = dev(g["pc"], True), dev(g["q"], True)
    s = dev(g["s"], True) if "s" in g else None
    t = dev(g["t"], True) if "t" in g else None
    f = dev(g["f"], True) if "f" in g else None
    out = R.pointcloud_project_fast(cfg, pc, q, t, None, kern, scaling_factor=s, focal_length=f, smooth=(sem == "smooth"))
    assert sorted(out.keys()) == sorted(["proj", "voxels", "tr_pc", "voxels_rgb", "proj_rgb", "drc_probs", "proj_depth"])
    proj = out["proj"]
    assert proj.shape == g[sem + "_proj"].shape and proj.dtype == torch.float32
    close(proj, g[sem + "_proj"], TOL, "proj")
    B = pc.shape[0]
    loss = ((proj - dev(g["gt"])) ** 2).sum() / B
    close(loss, float(g[sem + "_loss"]), TOL, "chain loss")
    loss.backward()
    close(pc.grad, g[sem + "_dpc"], TOL, "dpc")
    close(q.grad, g[sem + "_dq"], TOL, "dq")  # a sum over N points
    for nm, x in (("ds", s), ("dt", t), ("df", f)):
        if x is not None:
            close(x.grad, g[sem + "_" + nm], TOL, nm)
    # lazily produced outputs (stage kernels)
    close(out["tr_pc"], g[sem + "_tr_pc"], 2e-6, "tr_pc")
    close(out["proj_depth"], g[sem + "_proj_depth"], TOL, "proj_depth")
    if sem + "_voxels" in g:
        close(out["voxels"], g[sem + "_voxels"], TOL, "voxels")
        close(out["drc_probs"], g[sem + "_drc_probs"], TOL, "drc_probs")
    else:
        close(out["voxels"][:, ::4, ::4, ::4, 0], g[sem + "_voxels_sub"], TOL, "voxels (sub-sample)")
        close(out["drc_probs"][::8, :, ::4, ::4, 0], g[sem + "_drc_probs_sub"], TOL, "drc_probs (sub-sample)")
        close(out["voxels"].double().sum((2, 3, 4)), g[sem + "_voxels_zsum"], TOL, "voxels per-slice sums")


@pytest.mark.parametrize("name", ["f6_chain_g32_tf.npz", "f6_chain_c1_s0p64.npz"])
def test_fused_equals_staged(R, O, golden, name):
    """Two independent device implementations (fused LDS kernels vs one kernel per reference function)."""
    from dpc.render import _project_staged, _geometry

    g = golden(name)
    cfg = _chain_cfg(O, name)
    kern = R.smoothing_kernel(cfg, float(g["sigma_rel"]))
    args = [dev(g["pc"]), dev(g["q"]), dev(g["t"]) if "t" in g else None, dev(g["f"]) if "f" in g else None,
            dev(g["s"]) if "s" in g else None]
    grads = []
    for mode in ("fused", "staged"):
        a = [None if x is None else x.clone().requires_grad_(True) for x in args]
        if mode == "fused":
            proj = R.pointcloud_project_fast(cfg, a[0], a[1], a[2], None, kern, scaling_factor=a[4], focal_length=a[3])["proj"]
        else:
            proj = _project_staged(cfg, _geometry(cfg, kern), a[0], a[1], a[2], a[3], a[4], True)["proj"]
        ((proj - dev(g["gt"])) ** 2).sum().backward()
        grads.append([proj] + [x.grad for x in a if x is not None])
    for u, v in zip(*grads):
        close(u, v, TOL, "fused vs staged")


@pytest.mark.parametrize("G,ksz,sig,with_s", [(32, 11, 1.2, True), (64, 21, 0.64, True), (32, 11, 1.2, False)])
def test_dict_entries_from_the_saved_grid(R, O, G, ksz, sig, with_s):
    """What ModelPointCloud.compute_projection does with the renderer's dict (dpc/models/model_pc_to.py:262-269): it reads
    proj, drc_probs and proj_depth every step.  Here those entries (and voxels) come from the fused forward's grid_wh --
    D pass, scale/clamp, DRC -- and a loss over ALL of them sends its gradients back through that grid into the one fused
    backward.  Values and gradients against the oracle running the reference's full chain."""
    B, N = 3, 1500
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=ksz)
    pc, q, s, _, t, _ = O.synth_inputs(B, N, G, 6100 + G, with_t=True)
    rs = np.random.RandomState(11)
    w_proj, w_depth = rs.rand(B, G, G, 1), rs.rand(B, G, G, 1) * 0.1
    w_vox, w_probs = rs.rand(B, G, G, G, 1) * 0.01, rs.rand(G + 1, B, G, G, 1) * 0.01
    leaf = lambda x: x.clone().requires_grad_(True)
    cp, cq, ct = leaf(pc), leaf(q), leaf(t)
    cs = leaf(s) if with_s else None
    ref = O.pointcloud_project_fast(cfg, cp, cq, ct, None, O.smoothing_kernel(cfg, sig), scaling_factor=cs)
    total = lambda o, f: ((o["proj"] * f(w_proj)).sum() + (o["proj_depth"] * f(w_depth)).sum() + (o["voxels"] * f(w_vox)).sum()
                          + (o["drc_probs"] * f(w_probs)).sum())
    total(ref, lambda a: torch.from_numpy(a)).backward()
    gp, gq, gt_ = dev(pc, True), dev(q, True), dev(t, True)
    gs = dev(s, True) if with_s else None
    out = R.pointcloud_project_fast(cfg, gp, gq, gt_, None, R.smoothing_kernel(cfg, sig), scaling_factor=gs)
    for key in ("proj", "voxels", "drc_probs", "proj_depth"):
        close(out[key], ref[key], TOL, "dict entry " + key)
    total(out, dev).backward()
    close(gp.grad, cp.grad, TOL, "dpc through the dict entries")
    close(gq.grad, cq.grad, TOL, "dq through the dict entries")
    close(gt_.grad, ct.grad, TOL, "dt through the dict entries")
    if with_s:
        close(gs.grad, cs.grad, TOL, "ds through the dict entries")


def test_script_body_full_projection(R, O):
    """dpc/run/pc_full_proj_test.py:48-71 on the device (CUDA-branch semantics, and the literal CPU branch)."""
    vals = json.load(open(os.path.join(GOLDEN, "f7_scripts.json")))
    cfg = O.Cfg(vox_size=64, pc_gauss_kernel_size=21, pc_relative_sigma=3.0)
    np.random.seed(0)
    cam = torch.from_numpy(np.random.random((128, 4))).float().cuda()
    pc = torch.from_numpy(np.random.random((128, 140, 3))).float().cuda()
    sc = torch.from_numpy(np.random.random((128, 1))).float().cuda()
    kern = R.smoothing_kernel(cfg, R.get_smooth_sigma(cfg, 0))
    for smooth, key in ((True, "pc_full_proj_test_cuda_semantics"), (False, "pc_full_proj_test_literal_cpu")):
        out = R.pointcloud_project_fast(cfg, pc, cam, None, None, kern, scaling_factor=sc, smooth=smooth)
        for k, ref in vals[key].items():
            got = out[k].double().sum().item()
            assert abs(got - ref) <= TOL * max(1.0, abs(ref)), (key, k, got, ref)


# ------------------------------------------------------------------------------------------ oracle on fresh inputs
@pytest.mark.parametrize("B,N,G,k,sigma,with_t,with_f", [(3, 700, 32, 11, 2.0, True, False), (2, 3000, 64, 21, 0.64, False, False),
                                                          (2, 2000, 64, 21, 3.0, False, True), (2, 300, 16, 5, 0.7, False, False),
                                                          (1, 1000, 128, 21, 1.28, False, False), (2, 500, 24, 7, 1.0, True, True)])
def test_chain_vs_oracle(R, O, B, N, G, k, sigma, with_t, with_f):
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=k)
    pc, q, s, gt, t, f = O.synth_inputs(B, N, G, 4321 + G + N, with_t, with_f)
    leaf = lambda x: None if x is None else x.clone().requires_grad_(True)
    cp, cq, cs, ct, cf = leaf(pc), leaf(q), leaf(s), leaf(t), leaf(f)
    ref = O.pointcloud_project_fast(cfg, cp, cq, ct, None, O.smoothing_kernel(cfg, sigma), scaling_factor=cs, focal_length=cf)
    (((ref["proj"] - gt) ** 2).sum() / B).backward()
    gp, gq, gs, gt_, gf = dev(pc, True), dev(q, True), dev(s, True), dev(t, t is not None), dev(f, f is not None)
    out = R.pointcloud_project_fast(cfg, gp, gq, gt_, None, R.smoothing_kernel(cfg, sigma), scaling_factor=gs, focal_length=gf)
    (((out["proj"] - gt.cuda().float()) ** 2).sum() / B).backward()
    close(out["proj"], ref["proj"], TOL, "proj")
    close(gp.grad, cp.grad, TOL, "dpc")
    close(gq.grad, cq.grad, TOL, "dq")
    close(gs.grad, cs.grad, TOL, "ds")
    if t is not None:
        close(gt_.grad, ct.grad, TOL, "dt")
    if f is not None:
        close(gf.grad, cf.grad, TOL, "df")


@pytest.mark.parametrize("B,sigma,Gz,with_tf", [(9, 0.3, -1, False), (8, 0.45, -1, True), (3, 0.75, -1, False), (10, 1.0, -1, True),
                                                (8, 1.4, -1, False), (5, 0.64, 32, True), (16, 0.64, 128, False)])
def test_x_in_lanes_kernel_every_radius_bucket(R, O, B, sigma, Gz, with_tf):
    """The 64-wide forward slab kernel (k_splat_xl: W pass over DPP wave shifts) at every effective radius it is built for
    (sigma 0.3 -> 1 ... 1.4 -> 6 voxels), with batch sizes that do and do not take the XCD-aware workgroup map (multiples of 8
    or not), with translation / focal length, and with depths other than the width (vox_size_z): plain API and the fused
    one-candidate loss against the oracle."""
    G, N = 64, 1200
    cfg = O.Cfg(vox_size=G, vox_size_z=Gz, pc_gauss_kernel_size=21)
    pc, q, s, gt, t, f = O.synth_inputs(B, N, G, 9000 + B, with_tf, with_tf)
    leaf = lambda x: None if x is None else x.clone().requires_grad_(True)
    cp, cq, cs, ct, cf = leaf(pc), leaf(q), leaf(s), leaf(t), leaf(f)
    ref = O.pointcloud_project_fast(cfg, cp, cq, ct, None, O.smoothing_kernel(cfg, sigma), scaling_factor=cs, focal_length=cf)
    rloss = ((ref["proj"] - gt) ** 2).sum() / B
    rloss.backward()
    kern = R.smoothing_kernel(cfg, sigma)
    for fused in (False, True):
        gp, gq, gs, gt_, gf = dev(pc, True), dev(q, True), dev(s, True), dev(t, t is not None), dev(f, f is not None)
        if fused:
            loss, out, _ = R.pointcloud_project_loss(cfg, gp, gq, gt_, None, kern, scaling_factor=gs, focal_length=gf, gt=dev(gt))
        else:
            out = R.pointcloud_project_fast(cfg, gp, gq, gt_, None, kern, scaling_factor=gs, focal_length=gf)
            loss = ((out["proj"] - dev(gt)) ** 2).sum() / B
        loss.backward()
        tag = "xl sigma %g B %d %s: " % (sigma, B, "fused" if fused else "plain")
        close(loss, rloss, TOL, tag + "loss")
        close(out["proj"], ref["proj"], TOL, tag + "proj")
        close(gp.grad, cp.grad, TOL, tag + "dpc")
        close(gq.grad, cq.grad, TOL, tag + "dq")
        close(gs.grad, cs.grad, TOL, tag + "ds")
        if with_tf:
            close(gt_.grad, ct.grad, TOL, tag + "dt")
            close(gf.grad, cf.grad, TOL, tag + "df")


@pytest.mark.parametrize("G,Gz,sigma,B", [(64, 70, 0.64, 32), (64, -1, 3.0, 32), (32, -1, 0.8, 64), (64, 24, 1.0, 88), (64, -1, 0.64, 33)])
def test_forward_slab_workgroups_that_stay_for_several_slabs(R, O, G, Gz, sigma, B):
    """Batches large enough that a forward slab workgroup walks several slabs of its cloud (k_splat_xl and k_splat_hw keep
    one workgroup per CU): a depth whose last slab has fewer planes than the others (70 = 17 x 4 + 2: waves without a plane
    still have to reach the barriers), the radius-10 kernel at 64^3, the 32^3 kernel, and a slab count that only halves
    once (24 planes = 6 slabs, 88 clouds), and a batch off the XCD-aware workgroup map (33 clouds).  Fused one-candidate loss
    against the oracle."""
    N = 500
    cfg = O.Cfg(vox_size=G, vox_size_z=Gz, pc_gauss_kernel_size=21)
    pc, q, s, gt, _, _ = O.synth_inputs(B, N, G, 9700 + B + G)
    leaf = lambda x: x.clone().requires_grad_(True)
    cp, cq, cs = leaf(pc), leaf(q), leaf(s)
    ref = O.pointcloud_project_fast(cfg, cp, cq, None, None, O.smoothing_kernel(cfg, sigma), scaling_factor=cs)
    rloss = ((ref["proj"] - gt) ** 2).sum() / B
    rloss.backward()
    gp, gq, gs = dev(pc, True), dev(q, True), dev(s, True)
    loss, out, _ = R.pointcloud_project_loss(cfg, gp, gq, None, None, R.smoothing_kernel(cfg, sigma), scaling_factor=gs, gt=dev(gt))
    loss.backward()
    tag = "staying G %d Gz %d sigma %g B %d: " % (G, Gz, sigma, B)
    close(loss, rloss, TOL, tag + "loss")
    close(out["proj"], ref["proj"], TOL, tag + "proj")
    close(gp.grad, cp.grad, TOL, tag + "dpc")
    close(gq.grad, cq.grad, TOL, tag + "dq")
    close(gs.grad, cs.grad, TOL, tag + "ds")


@pytest.mark.parametrize("B,N,G,sigma", [(9, 1200, 64, 0.64), (33, 2000, 64, 0.3), (3, 3000, 128, 1.0)])
def test_write_through_grids_are_never_read_stale(R, O, B, N, G, sigma):
    """The W/H-filtered grid and its gradient are written THROUGH the L2s (csrc/dpc_kernels.h, store_through) and read by
    the next kernel with ordinary loads -- on workgroups of other XCDs too when the batch is off the XCD map.  One step
    plan (static buffers, so every call reuses the same addresses) alternates between three input sets while other kernels
    churn through the caches in between: every result must stay bit-identical to the first one of its input set."""
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=21)
    kern = R.smoothing_kernel(cfg, sigma)
    d = torch.device("cuda")
    sets = [[dev(x) for x in O.synth_inputs(B, N, G, 4200 + k)[:4]] for k in range(3)]
    plan = R.project_loss_step(cfg, kern, B, N, d)
    first = []
    for k in range(3):
        plan.run(*sets[k])
        torch.cuda.synchronize()
        first.append([t.clone() for t in (plan.loss, plan.dpc, plan.dq, plan.ds, plan.proj)])
    junk = torch.empty(48 << 20, device=d)
    for it in range(30):
        k = it % 3
        if it % 4 == 0:
            junk.normal_()
        plan.run(*sets[k])
        if it % 5 == 0:
            (junk[: 1 << 20] * 2).sum()
        torch.cuda.synchronize()
        for name, a, b_ in zip(("loss", "dpc", "dq", "ds", "proj"), (plan.loss, plan.dpc, plan.dq, plan.ds, plan.proj), first[k]):
            assert torch.equal(a, b_), "call %d (input set %d): %s differs from the first run" % (it, k, name)


def test_long_kernel_falls_back_to_staged(R, O):
    """Effective radius > 15 voxels exceeds the fused kernels' register window; same answer via stage kernels."""
    cfg = O.Cfg(vox_size=32, pc_gauss_kernel_size=41)
    pc, q, s, gt, _, _ = O.synth_inputs(2, 400, 32, 99)
    ref = O.pointcloud_project_fast(cfg, pc, q, None, None, O.smoothing_kernel(cfg, 8.0), scaling_factor=s)
    out = R.pointcloud_project_fast(cfg, dev(pc), dev(q), None, None, R.smoothing_kernel(cfg, 8.0), scaling_factor=dev(s))
    close(out["proj"], ref["proj"], TOL)


@pytest.mark.parametrize("G", [64, 32, 48])
def test_asymmetric_smoothing_kernel(R, O, G):
    """The API takes any separable 1-D kernel (the reference's smoothen_voxels3d is a plain conv3d with whatever it is given,
    point_cloud_to.py:90-103), not only Gaussians: a skewed 7-tap kernel through the x-in-lanes forward kernel (64-wide: its
    W pass has a shorter form for symmetric kernels, which the launcher must NOT pick here), the LDS-window kernels (32-wide)
    and the generic ones (48-wide), forward and backward -- the backward correlates with the mirrored kernel -- against the
    oracle."""
    B, N = 3, 2500
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=7)
    k1 = torch.tensor([0.02, 0.08, 0.25, 0.35, 0.18, 0.09, 0.03])
    kernel = [k1.reshape(1, 1, 1, 1, 7), k1.reshape(1, 1, 1, 7, 1), k1.reshape(1, 1, 7, 1, 1)]
    pc, q, s, gt, _, _ = O.synth_inputs(B, N, G, 6200 + G)
    gp, gq, gs = dev(pc, True), dev(q, True), dev(s, True)
    out = R.pointcloud_project_fast(cfg, gp, gq, None, None, kernel, scaling_factor=gs)
    (((out["proj"] - dev(gt)) ** 2).sum() / B).backward()
    cp, cq, cs = (x.clone().requires_grad_(True) for x in (pc, q, s))
    ref = O.pointcloud_project_fast(cfg, cp, cq, None, None, kernel, scaling_factor=cs)
    (((ref["proj"] - gt) ** 2).sum() / B).backward()
    close(out["proj"], ref["proj"], TOL, "asymmetric kernel, %d-wide: proj" % G)
    close(out["voxels"], ref["voxels"], TOL, "asymmetric kernel, %d-wide: voxels" % G)
    close(gp.grad, cp.grad, TOL, "asymmetric kernel, %d-wide: dpc" % G)
    close(gq.grad, cq.grad, TOL, "asymmetric kernel, %d-wide: dq" % G)
    close(gs.grad, cs.grad, TOL, "asymmetric kernel, %d-wide: ds" % G)
    # the fused loss and its one-call plan take the same kernel
    hp, hq, hs = dev(pc, True), dev(q, True), dev(s, True)
    loss, o2, _ = R.pointcloud_project_loss(cfg, hp, hq, None, None, kernel, scaling_factor=hs, gt=dev(gt))
    loss.backward()
    close(o2["proj"], ref["proj"], TOL, "asymmetric kernel, %d-wide, fused loss: proj" % G)
    close(hp.grad, cp.grad, TOL, "asymmetric kernel, %d-wide, fused loss: dpc" % G)


@pytest.mark.parametrize("G,Gz", [(48, -1), (24, 40), (96, 12)])
def test_generic_grid_widths_are_bit_reproducible_too(R, O, G, Gz):
    """Grids without kernels of their own (any width but 32 / 64 / 128) take the generic slab kernels.  Their splat accumulates in
    64-bit fixed point as well (it used float LDS atomics, whose arrival order showed in the last bits): five runs of the same
    call give identical bits for the raw grid, the silhouettes and every gradient, and the values are the oracle's."""
    B, N = 5, 4000
    cfg = O.Cfg(vox_size=G, vox_size_z=Gz, pc_gauss_kernel_size=11)
    kern = R.smoothing_kernel(cfg, 1.1)
    pc, q, s, gt, _, _ = O.synth_inputs(B, N, G, 6100 + G)
    pc = (pc * 0.35).contiguous()       # many points per voxel: long accumulation chains, where arrival order would show
    runs = []
    for _ in range(5):
        gp, gq, gs = dev(pc, True), dev(q, True), dev(s, True)
        out = R.pointcloud_project_fast(cfg, gp, gq, None, None, kern, scaling_factor=gs)
        (((out["proj"] - dev(gt)) ** 2).sum() / B).backward()
        raw, _ = R.pointcloud2voxels3d_fast(cfg, out["tr_pc"].detach(), None)
        runs.append([x.detach().clone() for x in (raw, out["proj"], gp.grad, gq.grad, gs.grad)])
    for r in runs[1:]:
        for name, a, b_ in zip(("raw", "proj", "dpc", "dq", "ds"), r, runs[0]):
            assert torch.equal(a, b_), "%s differs between two runs on a %d-wide grid" % (name, G)
    cp, cq, cs = (x.clone().requires_grad_(True) for x in (pc, q, s))
    ref = O.pointcloud_project_fast(cfg, cp, cq, None, None, O.smoothing_kernel(cfg, 1.1), scaling_factor=cs)
    (((ref["proj"] - gt) ** 2).sum() / B).backward()
    raw, proj, dpc, dq, ds = runs[0]
    assert float(ref["voxels_raw"].max()) > 3.0, "the test wants voxels that collect many points"
    close(raw, ref["voxels_raw"], TOL, "generic %d-wide grid: raw" % G)
    close(proj, ref["proj"], TOL, "generic %d-wide grid: proj" % G)
    close(dpc, cp.grad, TOL, "generic %d-wide grid: dpc" % G)
    close(dq, cq.grad, TOL, "generic %d-wide grid: dq" % G)
    close(ds, cs.grad, TOL, "generic %d-wide grid: ds" % G)


def test_grids_wider_than_the_lds_tile_are_refused_loudly(R, O):
    """The slab kernels keep whole H x W planes in LDS (160 KiB per CU): one for the forward (planes up to 199 x 199), a cell
    layer plus its halo for the backward (up to 141 x 141 besides the 32 / 64 / 128-wide kernels).  Beyond the forward's limit
    every entry point raises DPC_ERR_LDS ('an H x W plane does not fit the 160 KiB LDS tile'); between the two limits a
    forward-only call (no_grad, or inputs without gradients: prediction, evaluation) is served and checked against the oracle,
    while a call whose inputs require gradients is refused UP FRONT (dpc_check_grid) rather than in its backward -- no fault,
    no garbage, at any size.  (The reference takes any vox_size; its experiments use 32, 64 and 128.)"""
    from dpc.render import _native
    pc, q, s, _, _, _ = O.synth_inputs(2, 500, 16, 77)
    for G in (256, 192, 144):
        cfg = O.Cfg(vox_size=G, vox_size_z=8, pc_gauss_kernel_size=11)
        kern = R.smoothing_kernel(cfg, 1.0)
        gtG = O.synth_inputs(2, 1, G, 79)[3]
        with_grad = (lambda: R.pointcloud_project_fast(cfg, dev(pc, True), dev(q), None, None, kern, scaling_factor=dev(s)),
                     lambda: R.pointcloud_project_loss(cfg, dev(pc), dev(q, True), None, None, kern, scaling_factor=dev(s, True), gt=dev(gtG)))
        for call in with_grad:
            with pytest.raises(_native.DpcError) as err:
                call()
            assert err.value.code == _native.DPC_ERR_LDS and "LDS" in str(err.value)
        forward_only = (lambda: R.pointcloud_project_fast(cfg, dev(pc), dev(q), None, None, kern, scaling_factor=dev(s))["proj"],
                        lambda: R.pointcloud_project_loss(cfg, dev(pc), dev(q), None, None, kern, scaling_factor=dev(s), gt=dev(gtG))[1]["proj"],
                        lambda: R.pointcloud2voxels3d_fast(cfg, dev(pc, dtype=torch.float64), None)[0])
        if G > 199:
            for call in forward_only:
                with pytest.raises(_native.DpcError) as err:
                    call()
                assert err.value.code == _native.DPC_ERR_LDS
        else:
            ref = O.pointcloud_project_fast(cfg, pc, q, None, None, O.smoothing_kernel(cfg, 1.0), scaling_factor=s)
            with torch.no_grad():
                hard = R.pointcloud_project_fast(cfg, dev(pc, True), dev(q, True), None, None, kern, scaling_factor=dev(s, True))["proj"]
            close(hard, ref["proj"], TOL, "%d-wide forward under no_grad: proj" % G)
            close(forward_only[0](), ref["proj"], TOL, "%d-wide forward-only call: proj" % G)
            close(forward_only[1](), ref["proj"], TOL, "%d-wide forward-only loss call: proj" % G)
            close(forward_only[2](), O.pointcloud2voxels3d_fast(cfg, pc.double(), None)[0], TOL, "%d-wide forward-only splat" % G)
    torch.cuda.synchronize()
    # the widest grid of the differentiable path's generic kernels: 136 x 136 planes, forward and backward against the oracle
    G = 136
    cfg = O.Cfg(vox_size=G, vox_size_z=8, pc_gauss_kernel_size=11)
    pc, q, s, gt, _, _ = O.synth_inputs(1, 3000, G, 78)
    gp, gq, gs = dev(pc, True), dev(q, True), dev(s, True)
    out = R.pointcloud_project_fast(cfg, gp, gq, None, None, R.smoothing_kernel(cfg, 1.0), scaling_factor=gs)
    (((out["proj"] - dev(gt)) ** 2).sum()).backward()
    cp, cq, cs = (x.clone().requires_grad_(True) for x in (pc, q, s))
    ref = O.pointcloud_project_fast(cfg, cp, cq, None, None, O.smoothing_kernel(cfg, 1.0), scaling_factor=cs)
    (((ref["proj"] - gt) ** 2).sum()).backward()
    close(out["proj"], ref["proj"], TOL, "136-wide planes: proj")
    close(gp.grad, cp.grad, TOL, "136-wide planes: dpc")
    close(gq.grad, cq.grad, TOL, "136-wide planes: dq")
    close(gs.grad, cs.grad, TOL, "136-wide planes: ds")


def test_dead_branches_raise(R, O):
    pc, q = torch.zeros(1, 4, 3, device="cuda"), torch.ones(1, 4, device="cuda")
    with pytest.raises(NotImplementedError, match="all_rgb"):
        R.pointcloud_project_fast(O.Cfg(), pc, q, None, torch.zeros(1, 4, 3, device="cuda"))
    with pytest.raises(NotImplementedError, match="pose_quaternion"):
        R.pointcloud_project_fast(O.Cfg(pose_quaternion=False), pc, q, None, None)
    with pytest.raises(NotImplementedError, match="ptn_max_projection"):
        R.pointcloud_project_fast(O.Cfg(ptn_max_projection=True), pc, q, None, None)
    with pytest.raises(NotImplementedError, match="drc_logsum"):
        R.pointcloud_project_fast(O.Cfg(drc_logsum=False), pc, q, None, None)
    with pytest.raises(RuntimeError, match="MI355X only"):
        R.pointcloud_project_fast(O.Cfg(), pc.cpu(), q.cpu(), None, None)


# ------------------------------------------------------------------------------------------ full-size properties
def test_full_size_properties(R, O):
    """BASELINE config 2 (B=32, N=8000, 64^3, sigma=0.01 -> sigma_rel 0.64): size-independent properties."""
    B, N, G = 32, 8000, 64
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=21)
    kern = R.smoothing_kernel(cfg, 0.64)
    pc, q, s, gt, _, _ = O.synth_inputs(B, N, G, 1234)
    pc, q, s, gt = dev(pc, True), dev(q, True), dev(s, True), dev(gt)
    out = R.pointcloud_project_fast(cfg, pc, q, None, None, kern, scaling_factor=s)
    proj = out["proj"]
    assert proj.shape == (B, G, G, 1) and torch.isfinite(proj).all()
    empty = 1.0 - (1.0 - 1e-5) ** 64
    assert proj.min().item() >= empty - 1e-6 and proj.max().item() <= 1.0 + 2e-5
    # mass conservation of the splat: sum of the raw grid == number of in-bounds points
    tr = out["tr_pc"]
    raw, _ = R.pointcloud2voxels3d_fast(cfg, tr, None)
    nvalid = ((tr >= -0.5) & (tr <= 0.5)).all(-1).sum().item()
    assert abs(raw.double().sum().item() - nvalid) < 1e-3 * nvalid ** 0.5 + 0.5
    loss = ((proj - gt) ** 2).sum() / B
    loss.backward()
    g1 = [pc.grad.clone(), q.grad.clone(), s.grad.clone()]
    pc.grad = q.grad = s.grad = None
    proj2 = R.pointcloud_project_fast(cfg, pc, q, None, None, kern, scaling_factor=s)["proj"]
    (((proj2 - gt) ** 2).sum() / B).backward()
    # run to run: every sum of the path is taken in a fixed order or in exact integer arithmetic -- identical bits
    assert torch.equal(proj2, proj), "silhouettes differ between two runs"
    for name, a, b in zip(("dpc", "dq", "ds"), g1, [pc.grad, q.grad, s.grad]):
        assert torch.equal(a, b), "%s differs between two runs" % name
    # batch independence: clouds 0..3 alone give the same silhouettes and gradients
    sub = R.pointcloud_project_fast(cfg, pc[:4].detach(), q[:4].detach(), None, None, kern, scaling_factor=s[:4].detach())["proj"]
    close(sub, proj[:4], 1e-6, "batch independence")
    # the oracle on two of the 32 clouds (seconds on CPU)
    idx = [0, 17]
    cp, cq, cs = (x[idx].detach().cpu().clone().requires_grad_(True) for x in (pc, q, s))
    ref = O.pointcloud_project_fast(cfg, cp, cq, None, None, O.smoothing_kernel(cfg, 0.64), scaling_factor=cs)
    (((ref["proj"] - gt[idx].cpu().double()) ** 2).sum() / B).backward()
    close(proj[idx], ref["proj"], TOL, "proj vs oracle at full size")
    close(g1[0][idx], cp.grad, TOL, "dpc vs oracle at full size")
    close(g1[1][idx], cq.grad, TOL, "dq vs oracle at full size")
    close(g1[2][idx], cs.grad, TOL, "ds vs oracle at full size")


@pytest.mark.parametrize("path,B,chunk", [("plain", 4352, 128), ("fused", 8192, 128)])
def test_batches_whose_grids_pass_4_gib(R, O, path, B, chunk):
    """Maximum sizes: a batch whose intermediate grids (B x 64^3 fp32: 4.6 GB / 8.6 GB each) lie beyond 32-bit byte AND, at
    8192 clouds, 32-bit element offsets.  The oracle cannot run at that size; the property is batch independence, exact: every
    cloud's silhouette and gradients are bit for bit what the same cloud gives in a batch of 128 (fixed-order sums and integer
    accumulation make results independent of the company a cloud keeps; the fused loss's 1/B is a power of two in both runs)."""
    free, _ = torch.cuda.mem_get_info()
    if free < 60 << 30:
        pytest.skip("needs 60 GB of free device memory")
    N, G = 300, 64
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=21)
    kern = R.smoothing_kernel(cfg, 0.64)
    g = torch.Generator().manual_seed(77)
    pc = (torch.tanh(0.5 * torch.randn(B, N, 3, generator=g)) / 2).cuda().requires_grad_(True)
    q = torch.randn(B, 4, generator=g).cuda().requires_grad_(True)
    s = (0.5 + 0.5 * torch.rand(B, 1, generator=g)).cuda().requires_grad_(True)
    gt = (torch.rand(B, G, G, 1, generator=g) > 0.5).float().cuda()

    def run(lo, hi):
        a, b, c = (x[lo:hi].detach().clone().requires_grad_(True) for x in (pc, q, s))
        if path == "plain":
            proj = R.pointcloud_project_fast(cfg, a, b, None, None, kern, scaling_factor=c)["proj"]
            (proj * (gt[lo:hi] - 0.5)).sum().backward()
        else:
            loss, out, _ = R.pointcloud_project_loss(cfg, a, b, None, None, kern, scaling_factor=c, gt=gt[lo:hi])
            loss.backward()
            proj = out["proj"]
        return proj.detach(), a.grad, b.grad, c.grad

    whole = run(0, B)
    torch.cuda.synchronize()
    scale = 1.0 if path == "plain" else float(B) / chunk     # d(loss) carries 1/B: 1/8192 against 1/128
    assert all(torch.isfinite(t).all() for t in whole)
    assert float(whole[1].abs().max()) > 0 and float(whole[2].abs().max()) > 0
    for lo in list(range(0, B, chunk))[::7] + [B - chunk]:    # every seventh batch of 128 and the last one
        part = run(lo, lo + chunk)
        for name, w, p_, k in zip(("proj", "dpc", "dq", "ds"), whole, part, (1.0, scale, scale, scale)):
            assert torch.equal(w[lo:lo + chunk] * k, p_), "%s of clouds %d..%d depends on the batch (%s path, B=%d)" % (name, lo, lo + chunk, path, B)
    del whole
    torch.cuda.empty_cache()


@pytest.mark.parametrize("B,chunk", [(32, 8), (33, 3)])
@pytest.mark.parametrize("sigma_rel", [0.64, 0.9, 1.2, 3.0])
def test_backward_slab_thickness_is_not_visible_in_the_results(R, O, sigma_rel, B, chunk):
    """The 64-wide backward picks its slab per call: 8 cell layers (narrow row pads beyond tap radius 4, taps that reach over a
    row's end masked) when the clouds fill the chip, 4 or 3 layers when few clouds have backward work.  A batch of 32 takes the
    thick slabs, the same clouds in batches of 8 the thin ones: gradients equal to the rule against each other AND both against
    the oracle; d(points) -- per-point arithmetic in the same order whatever the slab -- bit for bit.  (33 clouds: thick slabs
    off the XCD-aware workgroup map, which wants a multiple of 8; the loss' 1/33 against 1/3 is not a power of two, so
    that case compares to rounding.)"""
    N, G = 3000 if B == 32 else 1200, 64
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=21)
    kern = R.smoothing_kernel(cfg, sigma_rel)
    pc, q, s, gt, _, _ = O.synth_inputs(B, N, G, 515)
    q[::4] = torch.tensor([1.0, 0.0, 0.0, 0.0])       # every fourth cloud unrotated: its third coordinate is the grid's x ...
    pc[:, ::5, 2] = pc[:, ::5, 2].sign() * 0.49      # ... and a fifth of its points sit 3 voxels from the x faces, where windows leave the row

    def run(lo, hi):
        a, b, c = dev(pc[lo:hi], True), dev(q[lo:hi], True), dev(s[lo:hi], True)
        loss, out, _ = R.pointcloud_project_loss(cfg, a, b, None, None, kern, scaling_factor=c, gt=dev(gt[lo:hi]))
        loss.backward()
        return out["proj"].detach(), a.grad, b.grad, c.grad

    whole = run(0, B)
    k = B / chunk
    for lo in (0, chunk, 3 * chunk):
        part = run(lo, lo + chunk)
        assert torch.equal(whole[0][lo:lo + chunk], part[0]), "silhouettes depend on the batch"
        if B == 32:
            assert torch.equal(whole[1][lo:lo + chunk] * k, part[1]), "d(points) depends on the slab thickness"   # 1/32 against 1/8
        else:
            close(whole[1][lo:lo + chunk] * k, part[1], 1e-6, "dpc thick vs thin slabs")
        sums = 1e-6 if B == 32 else 3e-6   # per-slab fp32 wave sums grouped 8 ways or 22; x 11 instead of x 4 on top for 33 clouds
        close(whole[2][lo:lo + chunk] * k, part[2], sums, "dq thick vs thin slabs")
        close(whole[3][lo:lo + chunk] * k, part[3], sums, "ds thick vs thin slabs")
    idx = [4, 30]
    cp, cq, cs = (x[idx].clone().requires_grad_(True) for x in (pc, q, s))
    ref = O.pointcloud_project_fast(cfg, cp, cq, None, None, O.smoothing_kernel(cfg, sigma_rel), scaling_factor=cs)
    (((ref["proj"] - gt[idx].double()) ** 2).sum() / B).backward()
    close(whole[0][idx], ref["proj"], TOL, "proj vs oracle (thick slabs, sigma_rel %g)" % sigma_rel)
    close(whole[1][idx], cp.grad, TOL, "dpc vs oracle (thick slabs, sigma_rel %g)" % sigma_rel)
    close(whole[2][idx], cq.grad, TOL, "dq vs oracle (thick slabs, sigma_rel %g)" % sigma_rel)
    close(whole[3][idx], cs.grad, TOL, "ds vs oracle (thick slabs, sigma_rel %g)" % sigma_rel)


def test_benchmarked_call_at_full_size(R, O):
    """The call bench.py times, at the size it times it: pointcloud_project_loss with one pose candidate per sample (the
    ray-march kernel runs the column backward inside the forward, the loss is summed in 64-bit fixed point), B=32, N=8000,
    64^3, sigma_rel 0.64.  Oracle on 3 of the 32 clouds; the loss against an fp64 sum over the device silhouettes; two runs
    bit for bit."""
    B, N, G = 32, 8000, 64
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=21)
    kern = R.smoothing_kernel(cfg, 0.64)
    pc, q, s, gt, _, _ = O.synth_inputs(B, N, G, 1234)
    gp, gq, gs, ggt = dev(pc, True), dev(q, True), dev(s, True), dev(gt)
    runs = []
    for _ in range(2):
        gp.grad = gq.grad = gs.grad = None
        loss, out, win = R.pointcloud_project_loss(cfg, gp, gq, None, None, kern, scaling_factor=gs, gt=ggt, num_candidates=1)
        loss.backward()
        runs.append([loss.detach().clone(), out["proj"].detach().clone(), gp.grad.clone(), gq.grad.clone(), gs.grad.clone()])
    for name, a, b in zip(("loss", "proj", "dpc", "dq", "ds"), *runs):
        assert torch.equal(a, b), "%s differs between two runs of the same step" % name
    loss, proj, dpc, dq, ds = runs[0]
    assert int(win.abs().max()) == 0
    close(loss, ((proj.double().cpu() - gt.double()) ** 2).sum() / B, 1e-6, "benchmarked call: loss vs fp64 sum of its silhouettes")
    idx = [0, 13, 31]
    cp, cq, cs = (x[idx].clone().requires_grad_(True) for x in (pc, q, s))
    ref = O.pointcloud_project_fast(cfg, cp, cq, None, None, O.smoothing_kernel(cfg, 0.64), scaling_factor=cs)
    (((ref["proj"] - gt[idx].double()) ** 2).sum() / B).backward()
    close(proj[idx], ref["proj"], TOL, "benchmarked call: proj vs oracle")
    close(dpc[idx], cp.grad, TOL, "benchmarked call: dpc vs oracle")
    close(dq[idx], cq.grad, TOL, "benchmarked call: dq vs oracle")
    close(ds[idx], cs.grad, TOL, "benchmarked call: ds vs oracle")


def test_fused_loss_does_not_hide_divergence(R, O):
    """A non-finite occupancy scale (a diverged network) or a `gt` that is no mask must surface as a NaN loss, as the
    reference's float sum would report it -- the fused loss sums in 64-bit fixed point, where such a share is flagged in a
    field of its own instead of being converted to an integer.  The next (clean) call is unaffected."""
    B, N, G = 4, 900, 32
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=11)
    pc, q, s, gt, _, _ = O.synth_inputs(B, N, G, 8800)
    kern = R.smoothing_kernel(cfg, 1.0)

    def run(scale, mask):
        gp, gq, gs = dev(pc, True), dev(q, True), dev(scale, True)
        loss, _, _ = R.pointcloud_project_loss(cfg, gp, gq, None, None, kern, scaling_factor=gs, gt=dev(mask), num_candidates=1)
        loss.backward()
        return float(loss.detach())

    clean = run(s, gt)
    assert np.isfinite(clean)
    bad_s = s.clone()
    bad_s[1] = float("nan")
    assert np.isnan(run(bad_s, gt))
    assert np.isnan(run(s, gt * 200.0))        # squared errors beyond the range of the fixed-point words
    assert run(s, gt) == clean                 # the words start from zero again
    # a NaN pose: the reference DROPS the cloud's points (masked_select on comparisons that are all false), so its loss stays
    # finite and only that cloud's gradients are NaN (oracle: d(pc), d(q) of the cloud); same here, the other clouds untouched
    bad_q = q.clone()
    bad_q[2, 1] = float("nan")
    cp, cq, cs = (x.clone().requires_grad_(True) for x in (pc, bad_q, s))
    ref = O.pointcloud_project_fast(cfg, cp, cq, None, None, O.smoothing_kernel(cfg, 1.0), scaling_factor=cs)
    rloss = ((ref["proj"] - gt) ** 2).sum() / B
    rloss.backward()
    gp, gq, gs = dev(pc, True), dev(bad_q, True), dev(s, True)
    loss, out, _ = R.pointcloud_project_loss(cfg, gp, gq, None, None, kern, scaling_factor=gs, gt=dev(gt), num_candidates=1)
    loss.backward()
    assert np.isfinite(float(rloss)) and bool(torch.isnan(cq.grad[2]).all())
    close(loss, rloss, TOL, "NaN pose: loss")
    close(out["proj"], ref["proj"], TOL, "NaN pose: proj")
    assert not bool(torch.isfinite(gq.grad[2]).any()), "the gradient of a NaN pose must not look healthy"
    keep = [0, 1, 3]
    close(gq.grad[keep], cq.grad[keep], TOL, "NaN pose: dq of the other clouds")
    close(gp.grad[keep], cp.grad[keep], TOL, "NaN pose: dpc of the other clouds")


@pytest.mark.parametrize("indexed", [False, True])
def test_shared_point_sets_do_not_hide_divergence(R, O, indexed):
    """The same for clouds that ADD into a shared point set's gradient (BASELINE config 3's layout): their contributions are
    summed in 64-bit fixed point, and a NaN / Inf contribution has no integer to stand for it.  It must not come back as a
    healthy-looking number: the set's gradient is NaN (the reference's float sums give NaN at the points the diverged cloud
    touches; here the whole set is flagged -- the decoder above mixes every point into every weight either way), the other
    sets keep the oracle's gradient, and the next clean call starts from zero again."""
    S, reps, N, G = 2, 4, 700, 32
    B = S * reps
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=11)
    pc = O.synth_inputs(S, N, G, 8810)[0]
    _, q, s, _, _, _ = O.synth_inputs(B, 4, G, 8811)
    gt = O.synth_inputs(B, 1, G, 8812)[3]
    kern = R.smoothing_kernel(cfg, 1.0)
    idx = R.point_dropout_indices(B, N, 0.6, torch.device("cuda"), torch.Generator(device="cuda").manual_seed(3)) if indexed else None

    def run(scale):
        gp, gq, gs = dev(pc, True), dev(q, True), dev(scale, True)
        loss, _, _ = R.pointcloud_project_loss(cfg, gp, gq, None, None, kern, scaling_factor=gs, gt=dev(gt), num_candidates=1,
                                               point_index=idx)
        loss.backward()
        torch.cuda.synchronize()
        return loss.detach().clone(), gp.grad.clone(), gq.grad.clone()

    clean = run(s)
    assert torch.isfinite(clean[0]) and torch.isfinite(clean[1]).all()
    bad_s = s.clone()
    bad_s[1] = float("nan")                       # cloud 1 of point set 0
    cp, cq, cs = (x.clone().requires_grad_(True) for x in (pc, q, bad_s))
    mat = cp.repeat_interleave(reps, dim=0)
    if indexed:
        mat = mat.gather(1, idx.long().cpu().unsqueeze(-1).expand(-1, -1, 3))
    ref = O.pointcloud_project_fast(cfg, mat, cq, None, None, O.smoothing_kernel(cfg, 1.0), scaling_factor=cs)
    (((ref["proj"] - gt) ** 2).sum() / B).backward()
    assert bool(torch.isnan(cp.grad[0]).any()) and bool(torch.isfinite(cp.grad[1]).all())
    loss, dpc, dq = run(bad_s)
    assert bool(torch.isnan(loss))
    assert bool(torch.isnan(dpc[0][torch.isnan(cp.grad[0]).cuda()]).all()), "a NaN contribution came back as a finite gradient"
    close(dpc[1], cp.grad[1], TOL, "diverged neighbour set: dpc of the healthy set")
    again = run(s)
    assert all(torch.equal(a, b) for a, b in zip(again, clean)), "the poison word must start from zero again"


def test_silhouette_loss_candidates(R, O, golden):
    """Fused min-of-K loss + gradient vs proj_loss_pose_candidates of the reference (fixture F8), and K=1."""
    g = golden("f8_candidates.npz")
    pred = dev(g["pred"], True)
    loss, win = R.silhouette_loss(pred, dev(g["gt"]), int(g["K"]))
    close(loss, g["loss"], TOL, "candidate loss")
    assert np.array_equal(win.cpu().numpy(), g["argmin"])
    (3.0 * loss).backward()
    close(pred.grad, 3.0 * g["dpred"], TOL, "candidate dpred")
    pred1 = dev(g["pred"][::4], True)
    gt = dev(g["gt"])
    loss1, _ = R.silhouette_loss(pred1, gt)
    ref = ((gt - pred1.detach()) ** 2).sum() / gt.shape[0]
    close(loss1, ref, TOL, "K=1 loss")
    loss1.backward()
    close(pred1.grad, 2 * (pred1.detach() - gt) / gt.shape[0], TOL, "K=1 dpred")


@pytest.mark.parametrize("K", [1, 4])
def test_project_loss_fused(R, O, K):
    """pointcloud_project_loss (loss folded into the ray-march kernels, losing candidates skipped) vs the oracle's
    projection followed by the reference's min-of-K loss."""
    S, N, G = 3, 900, 32
    B = S * K
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=11)
    pc, q, s, _, t, _ = O.synth_inputs(B, N, G, 555 + K, with_t=True)
    gtS = O.synth_inputs(S, 1, G, 77)[3]
    leaf = lambda x: x.clone().requires_grad_(True)
    cp, cq, cs, ct = leaf(pc), leaf(q), leaf(s), leaf(t)
    ref = O.pointcloud_project_fast(cfg, cp, cq, ct, None, O.smoothing_kernel(cfg, 1.2), scaling_factor=cs)
    rloss, rwin = O.proj_loss_pose_candidates(gtS, ref["proj"], K)
    (2.5 * rloss).backward()
    gp, gq, gs, gtt = dev(pc, True), dev(q, True), dev(s, True), dev(t, True)
    loss, out, win = R.pointcloud_project_loss(cfg, gp, gq, gtt, None, R.smoothing_kernel(cfg, 1.2), scaling_factor=gs,
                                               gt=dev(gtS), num_candidates=K)
    (2.5 * loss).backward()
    assert np.array_equal(win.cpu().numpy(), rwin.numpy())
    close(loss, rloss, TOL, "fused loss")
    close(out["proj"], ref["proj"], TOL, "proj")
    close(gp.grad, cp.grad, TOL, "dpc (fused loss)")
    close(gq.grad, cq.grad, TOL, "dq (fused loss)")
    close(gs.grad, cs.grad, TOL, "ds (fused loss)")
    close(gtt.grad, ct.grad, TOL, "dt (fused loss)")
    if K > 1:  # losing candidates: exact zeros
        lose = np.ones(B, bool)
        lose[np.arange(S) * K + rwin.numpy()] = False
        assert gp.grad[torch.from_numpy(lose).cuda()].abs().max().item() == 0.0
        assert gq.grad[torch.from_numpy(lose).cuda()].abs().max().item() == 0.0


@pytest.mark.parametrize("B,N,G,ksz,sig", [(37, 700, 32, 11, 1.2), (19, 1500, 64, 21, 0.64), (3, 2500, 64, 21, 3.0)])
def test_project_loss_fused_shapes(R, O, B, N, G, ksz, sig):
    """The one-candidate fused step (column backward inside the forward, 64-bit sum-and-count words per cloud) on batch
    sizes and kernels that are not the benchmark's: odd B, both grid sizes, a short and a full-length Gaussian; called
    twice on the same buffers (the per-cloud words must come back to zero), backward with a non-unit upstream factor."""
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=ksz)
    pc, q, s, gt, _, _ = O.synth_inputs(B, N, G, 900 + B)
    leaf = lambda x: x.clone().requires_grad_(True)
    cp, cq, cs = leaf(pc), leaf(q), leaf(s)
    ref = O.pointcloud_project_fast(cfg, cp, cq, None, None, O.smoothing_kernel(cfg, sig), scaling_factor=cs)
    rloss = ((ref["proj"] - gt) ** 2).sum() / B
    (0.5 * rloss).backward()
    gp, gq, gs = dev(pc, True), dev(q, True), dev(s, True)
    for rep in range(2):
        gp.grad = gq.grad = gs.grad = None
        loss, out, win = R.pointcloud_project_loss(cfg, gp, gq, None, None, R.smoothing_kernel(cfg, sig), scaling_factor=gs,
                                                   gt=dev(gt), num_candidates=1)
        (0.5 * loss).backward()
        assert int(win.abs().max()) == 0
        close(loss, rloss, TOL, "fused K=1 loss B=%d G=%d rep %d" % (B, G, rep))
        close(out["proj"], ref["proj"], TOL, "fused K=1 proj")
        close(gp.grad, cp.grad, TOL, "fused K=1 dpc")
        close(gq.grad, cq.grad, TOL, "fused K=1 dq")
        close(gs.grad, cs.grad, TOL, "fused K=1 ds")


@pytest.mark.parametrize("K,reps,sig,G", [(1, 4, 0.64, 32), (4, 8, 1.1, 32), (2, 2, 0.64, 32), (1, 2, 1.1, 32), (2, 4, 0.8, 24),
                                            (1, 3, 0.64, 64)])
def test_shared_point_sets(R, O, K, reps, sig, G):
    """[B/R,N,3] point sets shared by R consecutive clouds (views x pose candidates of one object)
    give the very same silhouettes, loss and winners as the materialised tf_repeat_0 copy, and a point gradient that is
    the sum over the replicas.  Both entry points (with and without the fused loss)."""
    S_obj, N = 3, 1100
    B = S_obj * reps
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=11)
    kern = R.smoothing_kernel(cfg, sig)  # G = 24: the generic (runtime-dimension) kernels; 32 and 64: the specialised ones
    pc, _, _, _, _, _ = O.synth_inputs(S_obj, N, G, 61)
    _, q, s, _, _, _ = O.synth_inputs(B, 4, G, 62)
    gt = O.synth_inputs(B // K, 1, G, 63)[3]
    full = pc.repeat_interleave(reps, dim=0)
    a_pc, a_q, a_s = dev(full, True), dev(q, True), dev(s, True)
    b_pc, b_q, b_s = dev(pc, True), dev(q, True), dev(s, True)
    la, oa, wa = R.pointcloud_project_loss(cfg, a_pc, a_q, None, None, kern, scaling_factor=a_s, gt=dev(gt), num_candidates=K)
    lb, ob, wb = R.pointcloud_project_loss(cfg, b_pc, b_q, None, None, kern, scaling_factor=b_s, gt=dev(gt), num_candidates=K)
    (1.7 * la).backward()
    (1.7 * lb).backward()
    # (not bit for bit: the generic slab kernels accumulate the splat with fp32 LDS atomics, whose order varies run to run)
    close(ob["proj"], oa["proj"], 1e-6, "shared points: proj")
    close(lb, la, 1e-6, "shared points: loss")
    assert torch.equal(wa, wb)
    assert b_pc.grad.shape == (S_obj, N, 3)
    close(b_pc.grad, a_pc.grad.reshape(S_obj, reps, N, 3).sum(1), 2e-6, "shared points: dpc summed over replicas (K=%d)" % K)
    close(b_q.grad, a_q.grad, 1e-6, "shared points: dq")
    close(b_s.grad, a_s.grad, 1e-6, "shared points: ds")
    close(ob["voxels"], oa["voxels"], 1e-7, "shared points: lazy voxels")
    # the plain entry point, external loss
    a_pc.grad = b_pc.grad = None
    pa = R.pointcloud_project_fast(cfg, a_pc, a_q, None, None, kern, scaling_factor=a_s)["proj"]
    pb = R.pointcloud_project_fast(cfg, b_pc, b_q, None, None, kern, scaling_factor=b_s)["proj"]
    w = dev(torch.rand(pa.shape, generator=torch.Generator().manual_seed(3)))
    (pa * w).sum().backward()
    (pb * w).sum().backward()
    close(pb, pa, 1e-6, "shared points: proj (plain entry point)")
    close(b_pc.grad, a_pc.grad.reshape(S_obj, reps, N, 3).sum(1), 2e-6, "shared points: dpc (plain entry point)")
    with pytest.raises(ValueError):
        R.pointcloud_project_fast(cfg, dev(pc[:2]), dev(q[:3]), None, None, kern)


@pytest.mark.parametrize("K,reps,G,sig", [(1, 4, 32, 1.1), (2, 4, 32, 0.64), (4, 8, 64, 0.64), (1, 1, 32, 1.1),
                                          (4, 4, 32, 0.9), (2, 2, 64, 0.64)])   # K == reps: one writer per point set
def test_point_index_vs_oracle(R, O, K, reps, G, sig):
    """Per-replica point dropout inside the kernels (reference: tf_repeat_0 then pc_point_dropout,
    dpc/models/model_pc_to.py:254-258, 302-306): cloud b projects point_cloud[b // R][point_index[b]].  Against the ORACLE
    on the materialised clouds -- silhouettes, winners, loss, d(pc) summed over the replicas into the stored sets (zeros
    where no replica kept the point), d(q), d(s) -- for the plain projection and for the fused min-of-K loss."""
    S, Nsrc, keep = 3, 1100, 0.3
    B = S * reps
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=11)
    pc, _, _, _, _, _ = O.synth_inputs(S, Nsrc, G, 4400 + K)
    _, q, s, _, _, _ = O.synth_inputs(B, 4, G, 4500 + reps)
    gt = O.synth_inputs(B // K, 1, G, 4600)[3]
    gen = torch.Generator(device="cuda").manual_seed(99)
    idx = R.point_dropout_indices(B, Nsrc, keep, torch.device("cuda"), gen)
    assert idx.shape == (B, int(Nsrc * keep)) and idx.dtype == torch.int32
    rows = idx.long().cpu()
    leaf = lambda x: x.clone().requires_grad_(True)
    # oracle: the reference's own order of operations -- replicate, then every replica drops its points
    cp, cq, cs = leaf(pc), leaf(q), leaf(s)
    mat = cp.repeat_interleave(reps, dim=0).gather(1, rows.unsqueeze(-1).expand(-1, -1, 3))
    ref = O.pointcloud_project_fast(cfg, mat, cq, None, None, O.smoothing_kernel(cfg, sig), scaling_factor=cs)
    rloss, rwin = O.proj_loss_pose_candidates(gt, ref["proj"], K)
    (1.5 * rloss).backward()
    # fused loss path
    gp, gq, gs = dev(pc, True), dev(q, True), dev(s, True)
    loss, out, win = R.pointcloud_project_loss(cfg, gp, gq, None, None, R.smoothing_kernel(cfg, sig), scaling_factor=gs, gt=dev(gt),
                                               num_candidates=K, point_index=idx)
    (1.5 * loss).backward()
    assert np.array_equal(win.cpu().numpy(), rwin.numpy())
    close(loss, rloss, TOL, "point_index: fused loss")
    close(out["proj"], ref["proj"], TOL, "point_index: proj")
    assert gp.grad.shape == pc.shape
    close(gp.grad, cp.grad, TOL, "point_index: dpc (fused loss)")
    close(gq.grad, cq.grad, TOL, "point_index: dq (fused loss)")
    close(gs.grad, cs.grad, TOL, "point_index: ds (fused loss)")
    kept = torch.zeros(S, Nsrc, dtype=torch.bool)
    kept[torch.arange(B).unsqueeze(1) // reps, rows] = True
    assert float(gp.grad.cpu()[~kept].abs().max()) == 0.0, "a point no cloud kept has a gradient"
    # plain projection path (external loss), same inputs
    hp, hq, hs = dev(pc, True), dev(q, True), dev(s, True)
    o2 = R.pointcloud_project_fast(cfg, hp, hq, None, None, R.smoothing_kernel(cfg, sig), scaling_factor=hs, point_index=idx)
    w = dev(np.random.RandomState(5).rand(B, G, G, 1))
    (o2["proj"] * w).sum().backward()
    cp2, cq2, cs2 = leaf(pc), leaf(q), leaf(s)
    mat2 = cp2.repeat_interleave(reps, dim=0).gather(1, rows.unsqueeze(-1).expand(-1, -1, 3))
    ref2 = O.pointcloud_project_fast(cfg, mat2, cq2, None, None, O.smoothing_kernel(cfg, sig), scaling_factor=cs2)
    (ref2["proj"] * w.cpu().double()).sum().backward()
    close(o2["proj"], ref2["proj"], TOL, "point_index: proj (plain)")
    close(hp.grad, cp2.grad, TOL, "point_index: dpc (plain)")
    close(hq.grad, cq2.grad, TOL, "point_index: dq (plain)")
    close(o2["tr_pc"], ref2["tr_pc"], 2e-6, "point_index: tr_pc of the kept points")


@pytest.mark.parametrize("K,reps,S", [(1, 4, 2), (4, 8, 2), (4, 4, 2), (2, 2, 2), (8, 8, 2), (8, 8, 16)])
def test_shared_point_sets_vs_oracle(R, O, K, reps, S):
    """Shared point sets without dropout (point_cloud [B/R,N,3], B poses) against the ORACLE on the tf_repeat_0 copies
    (dpc/models/model_pc_to.py:47-56, 302-306): silhouettes, winners, loss, d(pc) summed over the replicas, d(q), d(s).
    K == reps: one writer per point set (plain stores in the backward); K = reps = 8, S = 16 is the layout of
    `bench.py --config c5`."""
    N, G = 1300, 32
    B = S * reps
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=11)
    pc = O.synth_inputs(S, N, G, 5100 + K)[0]
    _, q, s, _, _, _ = O.synth_inputs(B, 4, G, 5200)
    gt = O.synth_inputs(B // K, 1, G, 5300)[3]
    leaf = lambda x: x.clone().requires_grad_(True)
    cp, cq, cs = leaf(pc), leaf(q), leaf(s)
    ref = O.pointcloud_project_fast(cfg, cp.repeat_interleave(reps, dim=0), cq, None, None, O.smoothing_kernel(cfg, 0.9), scaling_factor=cs)
    rloss, rwin = O.proj_loss_pose_candidates(gt, ref["proj"], K)
    rloss.backward()
    gp, gq, gs = dev(pc, True), dev(q, True), dev(s, True)
    loss, out, win = R.pointcloud_project_loss(cfg, gp, gq, None, None, R.smoothing_kernel(cfg, 0.9), scaling_factor=gs, gt=dev(gt),
                                               num_candidates=K)
    loss.backward()
    assert np.array_equal(win.cpu().numpy(), rwin.numpy())
    close(loss, rloss, TOL, "shared sets: loss")
    close(out["proj"], ref["proj"], TOL, "shared sets: proj")
    close(gp.grad, cp.grad, TOL, "shared sets: dpc summed over replicas")
    close(gq.grad, cq.grad, TOL, "shared sets: dq")
    close(gs.grad, cs.grad, TOL, "shared sets: ds")


def test_point_dropout_indices_properties(R):
    """dpc.render.point_dropout_indices against the contract of pc_point_dropout (dpc/util/point_cloud_to.py:269-295): per
    cloud int(N * keep) indices, all distinct, in range; clouds independent; a seeded generator reproduces the draw; every
    point is kept with probability keep (uniform without replacement)."""
    B, N, keep = 256, 1000, 0.07
    d = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(7)
    a = R.point_dropout_indices(B, N, keep, d, g)
    n = int(N * keep)
    assert a.shape == (B, n) and a.dtype == torch.int32 and int(a.min()) >= 0 and int(a.max()) < N
    srt = a.long().sort(dim=1).values
    assert bool((srt[:, 1:] != srt[:, :-1]).all()), "a cloud kept the same point twice"
    assert not torch.equal(srt[0], srt[1]) and len({tuple(r.tolist()) for r in srt[:32]}) == 32, "clouds share their draws"
    again = R.point_dropout_indices(B, N, keep, d, torch.Generator(device="cuda").manual_seed(7))
    other = R.point_dropout_indices(B, N, keep, d, torch.Generator(device="cuda").manual_seed(8))
    assert torch.equal(a, again) and not torch.equal(a, other)
    counts = torch.bincount(a.long().flatten(), minlength=N).double().cpu().numpy()   # ~ Binomial(B, keep) per point
    mean, sd = B * n / N, (B * keep * (1 - keep)) ** 0.5
    assert abs(counts.mean() - mean) < 1e-9 and counts.max() < mean + 6 * sd and counts.min() > max(0.0, mean - 6 * sd)
    assert abs(counts.std() - sd) < 0.25 * sd
    # keep = 1: a permutation of all points; the harness helper materialises the same selection
    full = R.point_dropout_indices(3, 50, 1.0, d)
    assert torch.equal(full.long().sort(dim=1).values.cpu(), torch.arange(50).repeat(3, 1))
    from dpc.harness import device_point_dropout

    pts = torch.randn(4, 100, 3, device=d)
    sub = device_point_dropout(pts, 0.25, torch.Generator(device="cuda").manual_seed(3))
    ind = R.point_dropout_indices(4, 100, 0.25, d, torch.Generator(device="cuda").manual_seed(3))
    assert torch.equal(sub, pts.gather(1, ind.long().unsqueeze(-1).expand(-1, -1, 3)))


@pytest.mark.parametrize("case", ["fused K=1", "K=4 shared sets + point_index", "plain call + torch loss", "64-wide x-in-lanes"])
def test_results_do_not_depend_on_what_fresh_buffers_hold(R, O, case, monkeypatch):
    """Every output buffer, workspace and record store the host layer allocates is handed over UNINITIALISED.  Run each entry
    point twice -- once with those allocations pre-filled with 0xFF bytes (NaN floats, huge indices and counters), once
    pre-filled with zeros -- and demand bit-identical losses, silhouettes and gradients: nothing may read what it did not
    write.  (A captured graph's private pool hands out memory nobody has touched; eager runs mostly recycle the previous
    step's buffers, which hides such reads.)"""
    import dpc.render._ops as ops

    real_empty = torch.empty
    fill = [0xFF]

    def poisoned_empty(*a, **k):
        t = real_empty(*a, **k)
        if t.is_cuda and t.numel():
            t.view(-1).view(torch.uint8).fill_(fill[0])
        return t

    monkeypatch.setattr(ops.torch, "empty", poisoned_empty)
    G = 64 if case == "64-wide x-in-lanes" else 32
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=21 if G == 64 else 11)
    kern = R.smoothing_kernel(cfg, 0.64 if G == 64 else 0.9)
    if case == "K=4 shared sets + point_index":
        S, reps, K, N = 2, 8, 4, 1500
        pc = O.synth_inputs(S, N, G, 6100)[0]
        _, q, s, _, _, _ = O.synth_inputs(S * reps, 4, G, 6200)
        gt = O.synth_inputs(S * reps // K, 1, G, 6300)[3]
        idx = torch.stack([torch.randperm(N, generator=torch.Generator().manual_seed(b))[:400] for b in range(S * reps)]).int().cuda()
    else:
        B, N, K, idx = 8, 3000, 1, None
        pc, q, s, gt, _, _ = O.synth_inputs(B, N, G, 6400)

    def run(byte):
        fill[0] = byte
        gp, gq, gs = dev(pc, True), dev(q, True), dev(s, True)
        if case == "plain call + torch loss":
            proj = R.pointcloud_project_fast(cfg, gp, gq, None, None, kern, scaling_factor=gs)["proj"]
            loss = ((proj - dev(gt)) ** 2).sum() / proj.shape[0]
        else:
            loss, out, _ = R.pointcloud_project_loss(cfg, gp, gq, None, None, kern, scaling_factor=gs, gt=dev(gt),
                                                     num_candidates=K, point_index=idx)
            proj = out["proj"]
        loss.backward()
        torch.cuda.synchronize()
        return [x.detach().clone() for x in (loss, proj, gp.grad, gq.grad, gs.grad)]

    a, b = run(0xFF), run(0x00)
    for name, x, y in zip(("loss", "proj", "dpc", "dq", "ds"), a, b):
        assert torch.isfinite(x).all(), name
        assert torch.equal(x, y), "%s depends on the contents of a freshly allocated buffer (%s)" % (name, case)


@pytest.mark.parametrize("K,shared", [(1, False), (4, False), (4, True)])
def test_backward_twice_through_one_forward(R, O, K, shared):
    """loss.backward(retain_graph=True) twice: the second pass over the saved buffers (record store, grid, workspace with
    its arrival counters and sum-and-count words) gives the same gradients again -- .grad ends at exactly twice the single
    pass -- also with a different upstream factor."""
    S, N, G = 2, 1200, 32
    reps = K if shared else 1
    B = S * K
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=11)
    kern = R.smoothing_kernel(cfg, 0.9)
    pc, q, s, _, _, _ = O.synth_inputs(B, N, G, 7100 + K)
    gt = O.synth_inputs(S, 1, G, 7200)[3]
    if shared:
        pc = pc[:S]

    def grads(passes):
        gp, gq, gs = dev(pc, True), dev(q, True), dev(s, True)
        loss, _, _ = R.pointcloud_project_loss(cfg, gp, gq, None, None, kern, scaling_factor=gs, gt=dev(gt), num_candidates=K)
        for i, up in enumerate(passes):
            (up * loss).backward(retain_graph=i + 1 < len(passes))
        return gp.grad, gq.grad, gs.grad

    once = grads([1.0])
    twice = grads([1.0, 1.0])
    mixed = grads([0.5, 1.5])
    for name, a, b_, c in zip(("dpc", "dq", "ds"), once, twice, mixed):
        assert torch.equal(2.0 * a, b_), name + ": the second backward through the same forward differs"
        close(c, 2.0 * a, TOL, name + " with upstream factors 0.5 + 1.5")


def test_plain_call_backward_twice_and_two_forwards_alive(R, O):
    """pointcloud_project_fast: backward twice through one forward (retain_graph), also through the lazily derived dict
    entries, and two forwards of different inputs alive at the same time before either backward runs."""
    B, N, G = 5, 1400, 32
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=11)
    kern = R.smoothing_kernel(cfg, 0.9)
    pc, q, s, gt, _, _ = O.synth_inputs(B, N, G, 7300)
    pc2, q2, s2, gt2, _, _ = O.synth_inputs(B, N, G, 7301)

    def leaves(a, b_, c):
        return dev(a, True), dev(b_, True), dev(c, True)

    def loss_of(out, g):
        return ((out["proj"] - dev(g)) ** 2).sum() / B + 0.1 * out["proj_depth"].sum() / B

    a = leaves(pc, q, s)
    loss_of(R.pointcloud_project_fast(cfg, a[0], a[1], None, None, kern, scaling_factor=a[2]), gt).backward()
    b_ = leaves(pc, q, s)
    l2 = loss_of(R.pointcloud_project_fast(cfg, b_[0], b_[1], None, None, kern, scaling_factor=b_[2]), gt)
    l2.backward(retain_graph=True)
    l2.backward()
    for name, x, y in zip(("dpc", "dq", "ds"), a, b_):
        assert torch.equal(2.0 * x.grad, y.grad), name
    # two forwards in flight, backwards in the opposite order
    c, d = leaves(pc, q, s), leaves(pc2, q2, s2)
    lc = loss_of(R.pointcloud_project_fast(cfg, c[0], c[1], None, None, kern, scaling_factor=c[2]), gt)
    ld = loss_of(R.pointcloud_project_fast(cfg, d[0], d[1], None, None, kern, scaling_factor=d[2]), gt2)
    ld.backward()
    lc.backward()
    e = leaves(pc2, q2, s2)
    loss_of(R.pointcloud_project_fast(cfg, e[0], e[1], None, None, kern, scaling_factor=e[2]), gt2).backward()
    for name, x, y in zip(("dpc", "dq", "ds"), a, c):
        assert torch.equal(x.grad, y.grad), name + " (first of two forwards alive)"
    for name, x, y in zip(("dpc", "dq", "ds"), e, d):
        assert torch.equal(x.grad, y.grad), name + " (second of two forwards alive)"


def test_strided_and_fp64_inputs_of_the_fused_call(R, O):
    """The fused call on inputs that are not fp32-contiguous: points as a strided view of a wider tensor, poses as every
    second row of a longer one, scales in fp64.  Same silhouettes and loss; gradients arrive in the inputs' own dtype and
    shape and land in the right rows of the tensors the views were cut from."""
    B, N, G = 6, 1100, 32
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=11)
    kern = R.smoothing_kernel(cfg, 0.9)
    pc, q, s, gt, _, _ = O.synth_inputs(B, N, G, 7400)
    a = dev(pc, True), dev(q, True), dev(s, True)
    la, oa, _ = R.pointcloud_project_loss(cfg, a[0], a[1], None, None, kern, scaling_factor=a[2], gt=dev(gt), num_candidates=1)
    la.backward()
    wide = torch.zeros(B, N, 5, device="cuda")
    wide[:, :, 1:4] = dev(pc)
    wide.requires_grad_(True)
    long_q = torch.zeros(2 * B, 4, device="cuda")
    long_q[::2] = dev(q)
    long_q.requires_grad_(True)
    s64 = dev(s, True, torch.float64)
    lb, ob, _ = R.pointcloud_project_loss(cfg, wide[:, :, 1:4], long_q[::2], None, None, kern, scaling_factor=s64, gt=dev(gt),
                                          num_candidates=1)
    lb.backward()
    assert torch.equal(la, lb) and torch.equal(oa["proj"], ob["proj"])
    assert torch.equal(wide.grad[:, :, 1:4], a[0].grad) and float(wide.grad[:, :, 0].abs().max()) == 0.0 and float(wide.grad[:, :, 4].abs().max()) == 0.0
    assert torch.equal(long_q.grad[::2], a[1].grad) and float(long_q.grad[1::2].abs().max()) == 0.0
    assert s64.grad.dtype == torch.float64 and torch.equal(s64.grad.float(), a[2].grad)


def _dropout_keys_numpy(seed, clouds, N):
    """Host restatement of the library's key function (csrc/dpc_stages.hip::dropout_key): uint32 [clouds, N]."""
    M = (1 << 64) - 1
    s0, s1 = int(seed[0]) & M, int(seed[1]) & M
    c = (np.arange(clouds, dtype=np.uint64)[:, None] * np.uint64(0x9E3779B97F4A7C15))
    i = (np.arange(N, dtype=np.uint64)[None, :] * np.uint64(0xD1B54A32D192ED03))
    with np.errstate(over="ignore"):
        x = np.uint64(s0) ^ c ^ i

        def mix(x):
            x = x ^ (x >> np.uint64(30)); x = x * np.uint64(0xBF58476D1CE4E5B9)
            x = x ^ (x >> np.uint64(27)); x = x * np.uint64(0x94D049BB133111EB)
            return x ^ (x >> np.uint64(31))
        x = mix(mix(x) + np.uint64(s1))
    return (x >> np.uint64(32)).astype(np.uint32)


@pytest.mark.parametrize("clouds,N,n", [(5, 8000, 560), (3, 1000, 70), (2, 1024, 1024), (2, 1025, 1), (1, 37, 36),
                                        (4, 20000, 19999), (2, 3000, 0), (3, 64, 64)])
def test_point_dropout_kernel_is_the_n_smallest_keys(clouds, N, n):
    """dpc_point_dropout_indices through the C ABI against a host restatement: the kept set is exactly the n points with
    the smallest hashed keys (ties broken by index), written in ascending order."""
    import ctypes

    from dpc.render import _native as Nat

    d = torch.device("cuda")
    seed = torch.tensor([0x1234567890ABCDEF - (1 << 63), 987654321], dtype=torch.int64, device=d)
    out = torch.full((clouds, n), -7, dtype=torch.int32, device=d)
    rc = Nat.lib().dpc_point_dropout_indices(clouds, N, n, Nat.ptr(seed), Nat.ptr(out), Nat.stream_ptr(d))
    assert rc == 0
    torch.cuda.synchronize()
    keys = _dropout_keys_numpy(seed.cpu().numpy(), clouds, N)
    order = np.lexsort((np.broadcast_to(np.arange(N), keys.shape), keys), axis=1)   # by key, then by index
    want = np.sort(order[:, :n], axis=1)
    assert np.array_equal(out.cpu().numpy(), want)
    assert Nat.lib().dpc_point_dropout_indices(1, 10, 11, Nat.ptr(seed), Nat.ptr(out), Nat.stream_ptr(d)) == Nat.DPC_ERR_SHAPE


def test_point_dropout_indices_in_a_replayed_graph(R):
    """The draw at the size of BASELINE config 3 (128 clouds x 8000 points, keep 0.07) captured in a HIP graph: every replay
    gives valid, distinct, ascending indices and a NEW draw.  (Round 2 found torch.topk returning float bit patterns as
    indices from the second replay on at this size -- the cause of the captured training step's memory fault.)"""
    d = torch.device("cuda")
    side = torch.cuda.Stream(d)
    with torch.cuda.stream(side):
        for _ in range(2):
            R.point_dropout_indices(128, 8000, 0.07, d)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        idx = R.point_dropout_indices(128, 8000, 0.07, d)
    seen = []
    for _ in range(4):
        g.replay()
        torch.cuda.synchronize()
        a = idx.cpu().numpy().astype(np.int64)
        assert a.shape == (128, 560) and a.min() >= 0 and a.max() < 8000
        assert (np.diff(a, axis=1) > 0).all(), "not ascending / not distinct"
        seen.append(a)
    assert all(not np.array_equal(seen[0], s) for s in seen[1:]) and not np.array_equal(seen[1], seen[2])


def test_graphed_project_loss(R, O):
    """The graph-captured step for eager loops: same loss and gradients as the eager call, on the sample data and on new
    data of the same shapes, called repeatedly."""
    import warnings

    B, N, G = 6, 1500, 32
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=11)
    kern = R.smoothing_kernel(cfg, 0.64)
    pc, q, s, gt, _, _ = O.synth_inputs(B, N, G, 808)
    leafs = lambda scale=1.0: [dev(pc * scale, True), dev(q, True), dev(s, True), dev(gt)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        step = R.graphed_project_loss(cfg, kern, *leafs())
        for scale in (1.0, 0.8, 0.8):
            a, b = leafs(scale), leafs(scale)
            le = R.pointcloud_project_loss(cfg, a[0], a[1], None, None, kern, scaling_factor=a[2], gt=a[3])[0]
            le.backward()
            lg = step(*b)
            lg.backward()
            assert float(le.detach()) == float(lg.detach())
            assert torch.equal(a[0].grad, b[0].grad)
            close(b[1].grad, a[1].grad, 1e-6, "graphed step: dq")
            close(b[2].grad, a[2].grad, 1e-6, "graphed step: ds")


def test_config4_full_size(R, O):
    """BASELINE config 4 per-GPU shard: 8 clouds x 16000 pts -> 128^3, sigma = 0.01 (sigma_rel 1.28), 21 taps.
    Size-independent properties on all 8 clouds + the oracle on one of them (128^3 fp64 on CPU takes seconds)."""
    B, N, G = 8, 16000, 128
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=21)
    kern = R.smoothing_kernel(cfg, 1.28)
    pc, q, s, gt, _, _ = O.synth_inputs(B, N, G, 404)
    gp, gq, gs, ggt = dev(pc, True), dev(q, True), dev(s, True), dev(gt)
    loss, out, win = R.pointcloud_project_loss(cfg, gp, gq, None, None, kern, scaling_factor=gs, gt=ggt)
    loss.backward()
    proj = out["proj"]
    assert proj.shape == (B, G, G, 1) and torch.isfinite(proj).all() and torch.isfinite(gp.grad).all()
    assert proj.min().item() >= 1.0 - (1.0 - 1e-5) ** G - 1e-6 and proj.max().item() <= 1.0 + 2e-5
    assert (win == 0).all()
    close(loss, ((proj - ggt) ** 2).sum() / B, TOL, "c4 loss")
    raw, _ = R.pointcloud2voxels3d_fast(cfg, out["tr_pc"], None)
    nvalid = ((out["tr_pc"] >= -0.5) & (out["tr_pc"] <= 0.5)).all(-1).sum().item()
    assert abs(raw.double().sum().item() - nvalid) < 1.0
    i = 3
    cp, cq, cs = (x[i:i + 1].clone().requires_grad_(True) for x in (pc, q, s))
    ref = O.pointcloud_project_fast(cfg, cp, cq, None, None, O.smoothing_kernel(cfg, 1.28), scaling_factor=cs)
    (((ref["proj"] - gt[i:i + 1]) ** 2).sum() / B).backward()
    close(proj[i:i + 1], ref["proj"], TOL, "c4 proj vs oracle")
    close(gp.grad[i:i + 1], cp.grad, TOL, "c4 dpc vs oracle")
    close(gq.grad[i:i + 1], cq.grad, TOL, "c4 dq vs oracle")
    close(gs.grad[i:i + 1], cs.grad, TOL, "c4 ds vs oracle")


def test_config5_full_size(R, O):
    """BASELINE config 5: 16 samples x K=8 candidate rotations of the SAME cloud (tf_repeat_0), 8000 pts, 64^3,
    min-of-K loss.  Winners and loss vs the oracle's silhouettes for 3 samples; losers get exact-zero gradients."""
    S, K, N, G = 16, 8, 8000, 64
    B = S * K
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=21)
    kern = R.smoothing_kernel(cfg, 0.64)
    base, _, sb, gtS, _, _ = O.synth_inputs(S, N, G, 505)
    pc = base.repeat_interleave(K, dim=0)          # same cloud for the K candidates of a sample
    s = sb.repeat_interleave(K, dim=0)
    q = O.synth_inputs(B, 1, G, 506)[1]            # K different candidate quaternions per sample
    gp, gq, gs = dev(pc, True), dev(q, True), dev(s, True)
    loss, out, win = R.pointcloud_project_loss(cfg, gp, gq, None, None, kern, scaling_factor=gs, gt=dev(gtS), num_candidates=K)
    loss.backward()
    proj = out["proj"]
    # the loss kernel's argmin/loss agree with the reference formula applied to the device silhouettes
    rloss, rwin = O.proj_loss_pose_candidates(gtS, proj.detach().double().cpu(), K)
    assert np.array_equal(win.cpu().numpy(), rwin.numpy())
    close(loss, rloss, TOL, "c5 loss")
    lose = torch.ones(B, dtype=torch.bool)
    lose[torch.arange(S) * K + rwin] = False
    assert gp.grad[lose.cuda()].abs().max().item() == 0.0 and gq.grad[lose.cuda()].abs().max().item() == 0.0
    assert gp.grad[(~lose).cuda()].abs().max().item() > 0.0
    # oracle on the candidates of 2 samples (16 clouds would take too long: take sample 5's 8 candidates only)
    smp = 5
    sl = slice(smp * K, (smp + 1) * K)
    cp, cq, cs = (x[sl].clone().requires_grad_(True) for x in (pc, q, s))
    ref = O.pointcloud_project_fast(cfg, cp, cq, None, None, O.smoothing_kernel(cfg, 0.64), scaling_factor=cs)
    l1, w1 = O.proj_loss_pose_candidates(gtS[smp:smp + 1], ref["proj"], K)
    (l1 / S).backward()                            # the batch loss divides by S samples
    assert w1.item() == rwin[smp].item()
    close(proj[sl], ref["proj"], TOL, "c5 proj vs oracle")
    close(gp.grad[sl], cp.grad, TOL, "c5 dpc vs oracle")
    close(gq.grad[sl], cq.grad, TOL, "c5 dq vs oracle")
    # a second run: the K-candidate path (unfused ray march, per-tile partials added in order by the finalize launch) gives
    # the same bits -- loss, winners and every gradient (this test materialises the clouds: no shared-set atomics)
    first = [loss.detach().clone(), win.clone(), gp.grad.clone(), gq.grad.clone(), gs.grad.clone()]
    gp.grad = gq.grad = gs.grad = None
    loss2, _, win2 = R.pointcloud_project_loss(cfg, gp, gq, None, None, kern, scaling_factor=gs, gt=dev(gtS), num_candidates=K)
    loss2.backward()
    for name, a, b in zip(("loss", "winner", "dpc", "dq", "ds"), first, [loss2.detach(), win2, gp.grad, gq.grad, gs.grad]):
        assert torch.equal(a, b), "c5: %s differs between two runs" % name


def test_many_chunks_fallback_iteration(R, O):
    """N > 16384 points = more than 64 sorted chunks per cloud: the slab kernels take the wave-per-chunk loop."""
    B, N, G = 2, 20000, 32
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=11)
    pc, q, s, gt, _, _ = O.synth_inputs(B, N, G, 808)
    leaf = lambda x: x.clone().requires_grad_(True)
    cp, cq, cs = leaf(pc), leaf(q), leaf(s)
    ref = O.pointcloud_project_fast(cfg, cp, cq, None, None, O.smoothing_kernel(cfg, 1.0), scaling_factor=cs)
    (((ref["proj"] - gt) ** 2).sum() / B).backward()
    gp, gq, gs = dev(pc, True), dev(q, True), dev(s, True)
    out = R.pointcloud_project_fast(cfg, gp, gq, None, None, R.smoothing_kernel(cfg, 1.0), scaling_factor=gs)
    (((out["proj"] - dev(gt)) ** 2).sum() / B).backward()
    close(out["proj"], ref["proj"], TOL, "proj (N=20000)")
    close(gp.grad, cp.grad, TOL, "dpc (N=20000)")
    close(gq.grad, cq.grad, TOL, "dq (N=20000)")  # a sum over 20000 points in fp32
    close(gs.grad, cs.grad, TOL, "ds (N=20000)")


def test_the_most_points_a_cloud_may_have(R, O):
    """N = DPC_MAX_POINTS = 2^20 - 1 (one more could wrap a voxel's 64-bit fixed-point sum; the library refuses it).  All of them
    in ONE voxel: the sum is exact.  A random cloud of that size (4096 sorted chunks) against the oracle, forward and backward."""
    from dpc.render import _native
    N, G = _native.DPC_MAX_POINTS, 64
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=21)
    corner = torch.full((1, N, 3), -0.5, device="cuda")
    raw, _ = R.pointcloud2voxels3d_fast(cfg, corner, None)
    assert float(raw[0, 0, 0, 0]) == float(N) and float(raw.double().sum()) == float(N)
    proj = R.pointcloud_project_fast(cfg, corner.requires_grad_(True), torch.tensor([[1.0, 0, 0, 0]], device="cuda"), None, None,
                                     R.smoothing_kernel(cfg, 0.64))["proj"]
    assert torch.isfinite(proj).all()
    with pytest.raises(RuntimeError, match="out of range"):
        R.pointcloud2voxels3d_fast(cfg, torch.zeros(1, N + 1, 3, device="cuda"), None)
    del corner, raw, proj

    g = torch.Generator().manual_seed(4)
    pc = torch.tanh(0.5 * torch.randn(1, N, 3, generator=g)) / 2
    pc[:, ::3] *= 1.9          # a third of the points far out: sparse rims whose voxels stay below the clamp
    q, s = torch.randn(1, 4, generator=g), torch.full((1, 1), 0.001)
    gt = (torch.rand(1, G, G, 1, generator=g) > 0.5).float()
    leaf = lambda x: x.clone().requires_grad_(True)
    cp, cq, cs = leaf(pc), leaf(q), leaf(s)
    ref = O.pointcloud_project_fast(cfg, cp, cq, None, None, O.smoothing_kernel(cfg, 0.64), scaling_factor=cs)
    ((ref["proj"] - gt) ** 2).sum().backward()
    gp, gq, gs = dev(pc, True), dev(q, True), dev(s, True)
    loss, out, _ = R.pointcloud_project_loss(cfg, gp, gq, None, None, R.smoothing_kernel(cfg, 0.64), scaling_factor=gs, gt=dev(gt))
    loss.backward()
    close(out["proj"], ref["proj"], TOL, "proj (N=2^20-1)")
    close(gs.grad, cs.grad, TOL, "ds (N=2^20-1)")
    close(gq.grad, cq.grad, TOL, "dq (N=2^20-1)")
    close(gp.grad, cp.grad, TOL, "dpc (N=2^20-1)")
    assert float(cp.grad.abs().max()) > 0 and float(cq.grad.abs().max()) > 0


def test_empty_clouds_through_fused_path(R, O):
    cfg = O.Cfg(vox_size=32, pc_gauss_kernel_size=11)
    pc = torch.zeros(2, 0, 3, device="cuda", requires_grad=True)
    q = torch.ones(2, 4, device="cuda", requires_grad=True)
    s = torch.ones(2, 1, device="cuda", requires_grad=True)
    out = R.pointcloud_project_fast(cfg, pc, q, None, None, R.smoothing_kernel(cfg, 1.0), scaling_factor=s)
    empty = 1.0 - (1.0 - 1e-5) ** 32
    assert out["proj"].shape == (2, 32, 32, 1) and abs(out["proj"].max().item() - empty) < 1e-7
    out["proj"].sum().backward()
    assert pc.grad.shape == (2, 0, 3) and q.grad.abs().max().item() == 0.0 and s.grad.abs().max().item() == 0.0
    # no clouds at all (an empty shard)
    pc0 = torch.zeros(0, 40, 3, device="cuda", requires_grad=True)
    q0 = torch.ones(0, 4, device="cuda", requires_grad=True)
    out0 = R.pointcloud_project_fast(cfg, pc0, q0, None, None, R.smoothing_kernel(cfg, 1.0))
    assert out0["proj"].shape == (0, 32, 32, 1)
    out0["proj"].sum().backward()
    assert pc0.grad.shape == (0, 40, 3) and q0.grad.shape == (0, 4)


@pytest.mark.parametrize("case", ["no points, K=1", "no points, K=4", "every point dropped", "no clouds"])
def test_empty_inputs_through_the_fused_loss(R, O, case):
    """Zero points per cloud, a dropout that keeps nothing, zero clouds: the fused loss is the loss of empty silhouettes (or
    0 for an empty batch), gradients are zeros of the inputs' shapes, nothing faults."""
    G = 32
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=11)
    kern = R.smoothing_kernel(cfg, 1.0)
    K = 4 if "K=4" in case else 1
    B, N = (0, 50) if case == "no clouds" else (4, 0 if "no points" in case else 50)
    pc = torch.rand(B, N, 3, device="cuda") - 0.5
    pc.requires_grad_(True)
    q = torch.randn(B, 4, device="cuda", requires_grad=True)
    s = torch.ones(B, 1, device="cuda", requires_grad=True)
    gt = (torch.rand(B // K, G, G, 1, device="cuda") > 0.5).float()
    idx = torch.zeros(B, 0, dtype=torch.int32, device="cuda") if case == "every point dropped" else None
    loss, out, win = R.pointcloud_project_loss(cfg, pc, q, None, None, kern, scaling_factor=s, gt=gt, num_candidates=K, point_index=idx)
    loss.backward()
    torch.cuda.synchronize()
    assert out["proj"].shape == (B, G, G, 1) and win.shape == (B // K,)
    if B:
        empty = 1.0 - (1.0 - 1e-5) ** G
        assert abs(float(out["proj"].max()) - empty) < 1e-7 and abs(float(out["proj"].min()) - empty) < 1e-7
        want = float(((empty - gt.double()) ** 2).sum() / (B // K))
        assert abs(float(loss) - want) <= 1e-5 * max(1.0, want)
    else:
        assert float(loss) == 0.0
    assert pc.grad.shape == pc.shape and float(pc.grad.abs().sum()) == 0.0
    assert q.grad.shape == q.shape and float(q.grad.abs().sum()) == 0.0


@pytest.mark.parametrize("case", ["only q requires grad", "K=4, one sample, one shared set", "K=2, one sample", "vox_size_z=16",
                                  "63 taps sigma_rel 8 (staged fallback)", "1 tap", "N=70000 at 64^3"])
def test_edge_configurations_of_the_fused_loss_vs_oracle(R, O, case):
    """Corners of the fused call against the oracle: a single sample with K candidates (with and without a shared point
    set), only the pose needing a gradient, a grid that is shallower than wide, a Gaussian too long for the fused kernels
    (the call falls back to the staged chain) and a single tap, and a cloud far beyond the flat record table (274 chunks)."""
    G, ksz, sig, B, N, K, reps, only_q, vz = 32, 11, 0.9, 4, 900, 1, 1, False, None
    if case == "only q requires grad":
        only_q = True
    elif case.startswith("K=4"):
        K, reps = 4, 4
    elif case.startswith("K=2"):
        B, N, K = 2, 700, 2
    elif case == "vox_size_z=16":
        B, N, vz = 3, 800, 16
    elif case.startswith("63 taps"):
        ksz, sig, B, N = 63, 8.0, 2, 600
    elif case == "1 tap":
        ksz, sig, B, N = 1, 0.5, 2, 600
    else:
        G, ksz, sig, B, N = 64, 21, 0.64, 1, 70000
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=ksz)
    if vz is not None:
        cfg.vox_size_z = vz
    S = B // reps
    pc = O.synth_inputs(S, N, G, 11)[0]
    _, q, s, _, _, _ = O.synth_inputs(B, 4, G, 12)
    gt = O.synth_inputs(B // K, 1, G, 13)[3]
    cp, cq, cs = (x.clone().requires_grad_(True) for x in (pc, q, s))
    ref = O.pointcloud_project_fast(cfg, cp.repeat_interleave(reps, 0) if reps > 1 else cp, cq, None, None,
                                    O.smoothing_kernel(cfg, sig), scaling_factor=cs)
    rloss = O.proj_loss_pose_candidates(gt, ref["proj"], K)[0] if K > 1 else ((ref["proj"] - gt) ** 2).sum() / B
    rloss.backward()
    gp, gq, gs = dev(pc, not only_q), dev(q, True), dev(s, not only_q)
    loss, out, _ = R.pointcloud_project_loss(cfg, gp, gq, None, None, R.smoothing_kernel(cfg, sig), scaling_factor=gs, gt=dev(gt),
                                             num_candidates=K)
    loss.backward()
    close(loss, rloss, TOL, case + ": loss")
    close(out["proj"], ref["proj"], TOL, case + ": proj")
    close(gq.grad, cq.grad, TOL, case + ": dq")
    if only_q:
        assert gp.grad is None and gs.grad is None
    else:
        close(gp.grad, cp.grad, TOL, case + ": dpc")
        close(gs.grad, cs.grad, TOL, case + ": ds")


def test_optional_outputs_of_the_c_abi(R, O, golden):
    """dpc_project_fwd called directly: the optional `tr_pc`, `raw` (unclamped splat) and `smoothed` (grid after the
    full Gaussian) outputs against the golden chain, and grid_wh == W/H-smoothed clamp(raw) via the stage kernels."""
    import ctypes
    from dpc.render import _native as N, _geometry
    from dpc.render._ops import _new_cells

    g = golden("f6_chain_g32.npz")
    cfg = O.Cfg(vox_size=32, pc_gauss_kernel_size=11)
    geom = _geometry(cfg, R.smoothing_kernel(cfg, float(g["sigma_rel"])))
    pc, q, s = dev(g["pc"]), dev(g["q"]), dev(g["s"])
    B, Np, G = pc.shape[0], pc.shape[1], 32
    P = geom.params(B, Np)
    L = N.lib()
    f = lambda *sh: torch.empty(sh, dtype=torch.float32, device="cuda")
    tr, raw, wh, sm, proj, trans = f(B, Np, 3), f(B, G, G, G), f(B, G, G, G), f(B, G, G, G), f(B, G, G), f(B, G, G)
    mask = torch.empty((B, G, L.dpc_mask_words_per_plane(ctypes.byref(P))), dtype=torch.int64, device="cuda")
    cells = _new_cells(P, pc.device)
    kxy, kz = geom.kern_ptrs()
    rc = L.dpc_project_fwd(ctypes.byref(P), N.ptr(pc), N.ptr(q), None, None, N.ptr(s), kxy, kz, N.ptr(tr), N.ptr(cells),
                           N.ptr(raw), N.ptr(wh), N.ptr(sm), N.ptr(mask), N.ptr(proj), N.ptr(trans), N.stream_ptr(pc.device))
    assert rc == 0
    torch.cuda.synchronize()
    close(tr, g["smooth_tr_pc"], 2e-6, "tr_pc (C ABI)")
    close(raw.unsqueeze(1), g["smooth_raw"], TOL, "raw (C ABI)")
    close(proj.unsqueeze(-1), g["smooth_proj"], TOL, "proj (C ABI)")
    vox = torch.clamp(sm * s.reshape(-1, 1, 1, 1), 0.0, 1.0)
    close(vox.unsqueeze(-1), g["smooth_voxels"], TOL, "voxels from `smoothed` (C ABI)")
    # clamp mask bits == (raw <= 1)
    bits = mask.cpu().numpy().view(np.uint64)
    unpacked = np.unpackbits(bits.view(np.uint8), bitorder="little").reshape(B, G, G * G).astype(bool)
    assert np.array_equal(unpacked, (raw.cpu().numpy().reshape(B, G, G * G) <= 1.0))
    # transmittance: proj = 1 - T + (e^eps - 1) y0  =>  T <= 1 and consistent with proj where the first voxel is empty
    assert trans.max().item() <= 1.0 and trans.min().item() >= 0.0


def test_point_dropout_matches_reference_rng(R):
    pts = torch.arange(2 * 10 * 3, dtype=torch.float32, device="cuda").reshape(2, 10, 3)
    np.random.seed(7)
    out, rgb = R.pc_point_dropout(pts, None, 0.5)
    np.random.seed(7)
    idx = [np.random.choice(10, 5, replace=False) for _ in range(2)]
    assert rgb is None and out.shape == (2, 5, 3)
    for b in range(2):
        assert torch.equal(out[b].cpu(), pts[b].cpu()[idx[b]])


@pytest.mark.parametrize("name", ["f32", "f64", "ties32", "ties64", "one_target", "one_source"])
def test_nearest_point_golden(R, name):
    """point_cloud_distance against the reference's outputs (F11): indices exact (first minimum, lattice ties
    included), nearest points exact, distances within 1 ulp (the reference's torch-CPU sqrt is not correctly rounded)."""
    g = np.load(os.path.join(GOLDEN, "f11_nearest.npz"))
    dt = torch.float64 if g[name + "_vs"].dtype == np.float64 else torch.float32
    proj, dist, idx = R.point_cloud_distance(dev(g[name + "_vs"], dtype=dt), dev(g[name + "_vt"], dtype=dt))
    assert idx.dtype == torch.int64 and dist.dtype == dt and proj.dtype == dt
    assert np.array_equal(idx.cpu().numpy(), g[name + "_idx"])
    assert np.array_equal(proj.cpu().numpy(), g[name + "_proj"])
    ref = g[name + "_dist"]
    assert (np.abs(dist.cpu().numpy() - ref) <= np.spacing(np.abs(ref))).all()


@pytest.mark.parametrize("ns,nt,dt", [(1000, 3000, torch.float32), (777, 5001, torch.float64), (8000, 16000, torch.float32),
                                      (5, 70000, torch.float64), (3000, 1, torch.float32)])
def test_nearest_point_oracle(R, O, ns, nt, dt):
    """Fresh seeds, sizes that exercise several target slices and ragged tiles: bit-exact against the oracle
    (indices AND distances: both use a correctly rounded sqrt), plus properties that need no oracle."""
    g = torch.Generator().manual_seed(ns * 7 + nt)
    vs = (torch.rand(ns, 3, generator=g, dtype=dt) - 0.5)
    vt = (torch.rand(nt, 3, generator=g, dtype=dt) - 0.5)
    vt[nt // 2] = vt[0]  # a duplicate target: the first one must win whenever it is the nearest
    vs[0] = vt[0]
    proj, dist, idx = R.point_cloud_distance(vs.cuda(), vt.cuda())
    i = idx.cpu()
    assert int(i.min()) >= 0 and int(i.max()) < nt and int(i[0]) == 0 and float(dist[0]) == 0.0
    assert torch.equal(proj.cpu(), vt[i])
    if ns * nt <= 2e7:
        _, od, oi = O.point_cloud_distance(vs, vt)
        assert torch.equal(i, oi)
        assert torch.equal(dist.cpu(), od)
    else:  # full size: the returned neighbour is at the returned distance, and no sampled target is closer
        d = (vt[i] - vs)
        rec = torch.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])
        close(dist, rec, tol=1e-6, what="nearest: distance of the returned neighbour")
        probe = vt[torch.randint(0, nt, (256,), generator=g)]
        dd = torch.cdist(vs.double(), probe.double())
        assert bool((dd.min(dim=1).values >= dist.cpu().double() * (1 - 1e-6)).all())


def test_chamfer_pair_golden(R):
    g = np.load(os.path.join(GOLDEN, "f11_nearest.npz"))
    pair = R.chamfer_distances(dev(g["chamfer_pred"], dtype=torch.float64), dev(g["chamfer_gt"], dtype=torch.float64))
    assert np.allclose(pair.cpu().numpy(), g["chamfer_pair"], rtol=1e-13, atol=0)
    md, ix = R.compute_distance(None, g["chamfer_pred"], g["chamfer_gt"])
    assert md.dtype == np.float64 and ix.dtype == np.float64 and md.shape == (g["chamfer_pred"].shape[0],)
    assert np.isclose(md.mean(), g["chamfer_pair"][0], rtol=1e-13)


def test_chamfer_of_prediction_files(R, tmp_path):
    """The evaluation's model loop (dpc/run/eval_chamfer_to.py:108-130) on a prediction file: per view the two directed mean
    nearest distances, with the per-view point counts and the reference rotation of the unsupervised evaluation, against
    brute-force numpy."""
    rs = np.random.RandomState(3)
    pts = rs.rand(3, 300, 3).astype(np.float32) - 0.5
    gt = (rs.rand(500, 3) - 0.5).astype(np.float64)
    nums = np.array([300, 120, 250])
    quat = np.array([[0.9, 0.1, -0.3, 0.2]])
    path = str(tmp_path / "m_pc.pkl")
    R.save_predictions(path, pts, num_points=nums)
    p, _, n = R.load_predictions(path)
    got = R.chamfer_of_predictions(p, gt, reference_rotation=quat, num_points=n)
    for i in range(3):
        pr = R.quaternion_rotate(torch.from_numpy(p[i, :nums[i]]).unsqueeze(0), torch.from_numpy(quat)).squeeze(0).numpy()
        d = np.sqrt(((pr[:, None, :].astype(np.float64) - gt[None]) ** 2).sum(-1))
        assert abs(got[i, 0] - d.min(1).mean()) < 1e-6 and abs(got[i, 1] - d.min(0).mean()) < 1e-6


def test_nearest_point_errors(R):
    with pytest.raises(IndexError):
        R.point_cloud_distance(torch.zeros(3, 3, device="cuda"), torch.zeros(0, 3, device="cuda"))
    with pytest.raises(ValueError):
        R.point_cloud_distance(torch.zeros(3, 2, device="cuda"), torch.zeros(4, 3, device="cuda"))
    with pytest.raises(RuntimeError):
        R.point_cloud_distance(torch.zeros(3, 3), torch.zeros(4, 3))
    p, d, i = R.point_cloud_distance(torch.zeros(0, 3, device="cuda"), torch.zeros(4, 3, device="cuda"))
    assert p.shape == (0, 3) and d.shape == (0,) and i.shape == (0,)


# ------------------------------------------------------------------------------------------ round 3: indices, schedules
def test_bad_point_index_is_an_error_not_a_fault(R, O):
    """An index row holding N_src and -1 (the reference's fancy indexing raises IndexError for the first, wraps the second,
    dpc/util/point_cloud_to.py:266-295): nothing out of range is read or written, the entry is dropped, the status word
    turns it into IndexError at check_status(), the clouds with clean rows are bit-identical to a clean run, and the debug
    mode raises at the call."""
    S, reps, Nsrc, n, G, K = 2, 4, 900, 300, 32, 2
    B = S * reps
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=11)
    kern = R.smoothing_kernel(cfg, 0.9)
    pc = O.synth_inputs(S, Nsrc, G, 8100)[0]
    _, q, s, _, _, _ = O.synth_inputs(B, 4, G, 8200)
    gt = O.synth_inputs(B // K, 1, G, 8300)[3]
    idx = R.point_dropout_indices(B, Nsrc, (n + 0.5) / Nsrc, torch.device("cuda"), torch.Generator(device="cuda").manual_seed(5))
    assert idx.shape == (B, n)
    R.check_status()   # clean slate

    def run(index):
        gp, gq, gs = dev(pc, True), dev(q, True), dev(s, True)
        loss, out, win = R.pointcloud_project_loss(cfg, gp, gq, None, None, kern, scaling_factor=gs, gt=dev(gt), num_candidates=K,
                                                   point_index=index)
        loss.backward()
        o2 = R.pointcloud_project_fast(cfg, dev(pc), dev(q), None, None, kern, scaling_factor=dev(s), point_index=index)
        torch.cuda.synchronize()
        return out["proj"].clone(), gp.grad.clone(), gq.grad.clone(), o2["proj"].clone()

    clean = run(idx)
    assert R.check_status() == 0
    bad = idx.clone()
    bad[3, 7], bad[3, 100], bad[3, 299] = Nsrc, -1, 2 ** 31 - 1    # cloud 3 (point set 0)
    dirty = run(bad)
    with pytest.raises(IndexError):
        R.check_status()
    assert R.check_status() == 0, "the status word is cleared by the check that raised"
    others = [b for b in range(B) if b != 3]
    assert torch.equal(dirty[0][others], clean[0][others]) and torch.equal(dirty[3][others], clean[3][others])
    assert torch.isfinite(dirty[0]).all() and torch.isfinite(dirty[1]).all() and torch.isfinite(dirty[2]).all()
    # cloud 3 itself: exactly the projection of its row without the three bad entries
    keep = torch.ones(n, dtype=torch.bool)
    keep[[7, 100, 299]] = False
    sub = pc[0:1].gather(1, idx[3].long().cpu()[keep].view(1, -1, 1).expand(1, -1, 3))
    want = R.pointcloud_project_fast(cfg, dev(sub), dev(q[3:4]), None, None, kern, scaling_factor=dev(s[3:4]))["proj"]
    assert torch.equal(dirty[3][3:4], want)
    R.set_debug_checks(True)
    try:
        with pytest.raises(IndexError):
            R.pointcloud_project_fast(cfg, dev(pc), dev(q), None, None, kern, scaling_factor=dev(s), point_index=bad)
        R.pointcloud_project_fast(cfg, dev(pc), dev(q), None, None, kern, scaling_factor=dev(s), point_index=idx)
    finally:
        R.set_debug_checks(False)
    assert R.check_status() == 0


@pytest.mark.parametrize("K,reps", [(4, 4), (2, 2), (2, 4)])
def test_repeated_point_indices_vs_oracle(R, O, K, reps):
    """`point_index` rows may repeat an index (include/dpc_render.h): both contributions are summed into the point's
    gradient -- also where K == reps, the layout whose single winner per point set otherwise stores plainly."""
    S, Nsrc, n, G = 3, 500, 260, 32
    B = S * reps
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=11)
    pc = O.synth_inputs(S, Nsrc, G, 8400 + K)[0]
    _, q, s, _, _, _ = O.synth_inputs(B, 4, G, 8500 + reps)
    gt = O.synth_inputs(B // K, 1, G, 8600)[3]
    rows = torch.randint(0, Nsrc, (B, n), generator=torch.Generator().manual_seed(11))   # with replacement: many repeats
    rows[:, 1] = rows[:, 0]
    leaf = lambda x: x.clone().requires_grad_(True)
    cp, cq, cs = leaf(pc), leaf(q), leaf(s)
    mat = cp.repeat_interleave(reps, dim=0).gather(1, rows.unsqueeze(-1).expand(-1, -1, 3))
    ref = O.pointcloud_project_fast(cfg, mat, cq, None, None, O.smoothing_kernel(cfg, 0.9), scaling_factor=cs)
    rloss, rwin = O.proj_loss_pose_candidates(gt, ref["proj"], K)
    rloss.backward()
    gp, gq, gs = dev(pc, True), dev(q, True), dev(s, True)
    loss, out, win = R.pointcloud_project_loss(cfg, gp, gq, None, None, R.smoothing_kernel(cfg, 0.9), scaling_factor=gs, gt=dev(gt),
                                               num_candidates=K, point_index=rows.int().cuda())
    loss.backward()
    assert np.array_equal(win.cpu().numpy(), rwin.numpy())
    close(out["proj"], ref["proj"], TOL, "repeated indices: proj")
    close(gp.grad, cp.grad, TOL, "repeated indices: dpc")
    close(gq.grad, cq.grad, TOL, "repeated indices: dq")


def test_device_schedule_follows_sigma_and_keep_count_under_replay(R, O):
    """A call captured ONCE in a HIP graph, with its Gaussian taps and its live-point count in device memory
    (DeviceSchedule), replayed while sigma and the keep-count move along their schedules
    (dpc/models/model_pc_to.py:59-87, 171-179, 254-258): every replay equals the eager call made with that step's kernel and
    that step's number of points."""
    S, reps, Nsrc, G, K, cap = 2, 4, 2000, 64, 2, 600
    B = S * reps
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=21)
    pc = O.synth_inputs(S, Nsrc, G, 8700)[0]
    _, q, s, _, _, _ = O.synth_inputs(B, 4, G, 8800)
    gt = O.synth_inputs(B // K, 1, G, 8900)[3]
    d = torch.device("cuda")
    idx = R.point_dropout_indices(B, Nsrc, (cap + 0.5) / Nsrc, d, torch.Generator(device="cuda").manual_seed(6))   # ascending rows
    assert idx.shape == (B, cap)
    sig0 = 1.3
    k0 = R.smoothing_kernel(cfg, sig0)
    sched = R.DeviceSchedule(d, k0[0], k0[2], n_live=cap, capacity=cap)
    assert sched.buckets == (R.taps_bucket(k0[0]), R.taps_bucket(k0[2])) and sched.buckets[0] >= 4
    gp, gq, gs, ggt = dev(pc, True), dev(q, True), dev(s, True), dev(gt)

    def step():
        gp.grad = gq.grad = gs.grad = None
        loss, out, _ = R.pointcloud_project_loss(cfg, gp, gq, None, None, k0, scaling_factor=gs, gt=ggt, num_candidates=K,
                                                 point_index=idx, schedule=sched)
        loss.backward()
        return loss, out["proj"]

    side = torch.cuda.Stream(d)
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            loss, proj = step()
        for sig, n in [(1.3, 600), (1.1, 560), (0.9, 301), (0.64, 256), (0.5, 1), (1.3, 600)]:
            kern = R.smoothing_kernel(cfg, sig)
            assert sched.fits(kern[0], kern[2], n)
            sched.update(kern[0], kern[2], n)
            graph.replay()
            side.synchronize()
            got = (loss.clone(), proj.clone(), gp.grad.clone(), gq.grad.clone(), gs.grad.clone())
            ep, eq, es = dev(pc, True), dev(q, True), dev(s, True)
            el, eo, _ = R.pointcloud_project_loss(cfg, ep, eq, None, None, kern, scaling_factor=es, gt=ggt, num_candidates=K,
                                                  point_index=idx[:, :n].contiguous())
            el.backward()
            tag = "schedule sigma %.2f n %d: " % (sig, n)
            close(got[0], el, 1e-6, tag + "loss")
            close(got[1], eo["proj"], 1e-6, tag + "proj")
            close(got[2], ep.grad, 2e-6, tag + "dpc")
            close(got[3], eq.grad, 2e-6, tag + "dq")
            close(got[4], es.grad, 2e-6, tag + "ds")
    big = R.smoothing_kernel(cfg, 3.0)
    assert not sched.fits(big[0], big[2], 100) and not sched.fits(k0[0], k0[2], cap + 1)
    small = R.smoothing_kernel(cfg, 0.3)
    assert sched.fits(small[0], small[2], 500) and not sched.tight(small[0], small[2], 500)
    # the index draw with a device-side count: the first n_live slots of every row are the n_live smallest keys' points
    sched.update(k0[0], k0[2], 123)
    gen = lambda: torch.Generator(device="cuda").manual_seed(21)
    part = R.point_dropout_indices(B, Nsrc, (cap + 0.5) / Nsrc, d, gen(), n_live=sched.n_live)
    full = R.point_dropout_indices(B, Nsrc, 123.5 / Nsrc, d, gen())
    assert full.shape == (B, 123) and torch.equal(part[:, :123], full)
    # the STAGE-LEVEL path under the same schedule (the lazy entries of the fused loss's output dict, and the fallback for
    # Gaussians too long for the fused kernels): the dead slots of a row splat nothing there either
    with torch.no_grad():
        _, lazy, _ = R.pointcloud_project_loss(cfg, dev(pc), dev(q), None, None, k0, scaling_factor=dev(s), gt=ggt, num_candidates=K,
                                               point_index=idx, schedule=sched)
        fast = R.pointcloud_project_fast(cfg, dev(pc), dev(q), None, None, k0, scaling_factor=dev(s), point_index=idx, schedule=sched)
        want = R.pointcloud_project_fast(cfg, dev(pc), dev(q), None, None, k0, scaling_factor=dev(s),
                                         point_index=idx[:, :123].contiguous())
        for name, got in (("staged (loss dict)", lazy), ("from the saved grid (plain dict)", fast)):
            close(got["voxels"], want["voxels"], 1e-6, "n_live < capacity, %s: voxels" % name)
            close(got["proj_depth"], want["proj_depth"], 1e-6, "n_live < capacity, %s: proj_depth" % name)
            assert torch.equal(got["tr_pc"][:, :123], want["tr_pc"]) and bool((got["tr_pc"][:, 123:] == 2.0).all())
        close(lazy["proj"], want["proj"], 1e-6, "n_live < capacity: proj")
    with pytest.raises(ValueError):          # values that no longer fit the captured tap windows / rows are refused
        sched.update(big[0], big[2], 100)
    with pytest.raises(ValueError):
        sched.update(k0[0], k0[2], cap + 1)


@pytest.mark.parametrize("B,N,G,ksz,sig,K,reps", [(8, 3000, 64, 21, 0.64, 1, 1), (5, 700, 32, 11, 1.3, 1, 1), (3, 1500, 64, 21, 3.0, 1, 1),
                                                  (12, 900, 32, 11, 0.9, 4, 1), (16, 900, 64, 21, 0.64, 8, 8)])
def test_step_plan_is_bit_identical_to_the_autograd_path(R, O, B, N, G, ksz, sig, K, reps):
    """dpc_project_loss_step / ProjectLossStep (forward + backward of the fused loss as one native call on static buffers)
    against pointcloud_project_loss + backward, bit for bit, over back-to-back runs with inputs that change from run to run
    -- one pose candidate per sample (four launches), K candidates (six), K candidates sharing their sample's point set."""
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=ksz)
    kern = R.smoothing_kernel(cfg, sig)
    d = torch.device("cuda")
    sets, want = [], []
    for k in range(3):
        pc, q, s, _, _, _ = O.synth_inputs(B, N, G, 9100 + k)
        gt = O.synth_inputs(B // K, 1, G, 9200 + k)[3]
        sets.append([dev(x) for x in (pc[:B // reps].contiguous(), q, s, gt)])
        a, b_, c = (x.clone().requires_grad_(True) for x in sets[-1][:3])
        loss, out, win = R.pointcloud_project_loss(cfg, a, b_, None, None, kern, scaling_factor=c, gt=sets[-1][3], num_candidates=K)
        (0.5 * loss).backward()
        want.append([x.clone() for x in (loss.detach(), out["proj"], a.grad, b_.grad, c.grad, win)])
    plan = R.project_loss_step(cfg, kern, B, N, d, num_candidates=K, point_replicas=reps)
    half = torch.full((), 0.5, device=d)
    for it in range(12):
        k = (it * 7 + it // 5) % 3
        pc, q, s, gt = sets[k]
        plan.run(pc, q, s, gt, dloss=half)
        if it % 4 != 1:
            torch.cuda.synchronize()
            for g, w, name in zip((plan.loss, plan.proj, plan.dpc, plan.dq, plan.ds, plan.winner), want[k],
                                  ("loss", "proj", "dpc", "dq", "ds", "winner")):
                assert torch.equal(g, w), "step plan, run %d: %s differs" % (it, name)
    with pytest.raises(ValueError):
        plan.run(sets[0][0].double(), sets[0][1], sets[0][2], sets[0][3])


@pytest.mark.parametrize("K,reps,indexed", [(4, 16, True), (1, 4, False), (2, 8, True), (1, 4, True)])
def test_shared_sets_with_several_writers_are_bit_reproducible(R, O, K, reps, indexed):
    """Several LIVE clouds adding into one point set's gradient (BASELINE config 3: 4 views x 4 pose candidates per object =
    4 winners per set) used to do so with float atomics, whose arrival order changed the last bits from run to run -- the
    one documented exception to bit-reproducibility.  They add 64-bit fixed point now (exact, so order-free): five runs of
    the same call give identical bits, for replicas with and without per-cloud point subsets, and the sums agree with the
    oracle as before."""
    S, N, G = 3, 2500, 64
    B = S * reps
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=21)
    kern = R.smoothing_kernel(cfg, 0.64)
    # (seed of the un-indexed case: 9301 puts one smoothed voxel within 1e-6 RELATIVE of the DRC clamp at eps = 1e-5, where
    # the fp32 Gaussian and the reference's fp64 one decide differently whether that voxel passes a gradient -- 3.7e-4 on one
    # point, in the shared AND in the materialised call alike; DESIGN.md section 2 on hard thresholds)
    pc = O.synth_inputs(S, N, G, (9300 if indexed else 9350) + K)[0]
    _, q, s, _, _, _ = O.synth_inputs(B, 4, G, 9400 + reps)
    gt = O.synth_inputs(B // K, 1, G, 9500)[3]
    idx = R.point_dropout_indices(B, N, 0.3, torch.device("cuda"), torch.Generator(device="cuda").manual_seed(12)) if indexed else None
    runs = []
    for _ in range(5):
        gp, gq, gs = dev(pc, True), dev(q, True), dev(s, True)
        loss, out, win = R.pointcloud_project_loss(cfg, gp, gq, None, None, kern, scaling_factor=gs, gt=dev(gt), num_candidates=K,
                                                   point_index=idx)
        loss.backward()
        runs.append((loss.detach().clone(), gp.grad.clone(), gq.grad.clone(), gs.grad.clone()))
        hp = dev(pc, True)
        o2 = R.pointcloud_project_fast(cfg, hp, dev(q), None, None, kern, scaling_factor=dev(s), point_index=idx)
        (o2["proj"] * o2["proj"]).sum().backward()
        runs[-1] += (hp.grad.clone(),)
    for r in runs[1:]:
        assert all(torch.equal(a, b_) for a, b_ in zip(r, runs[0])), "two runs of the same call differ"
    # and the values are the oracle's
    leaf = lambda x: x.clone().requires_grad_(True)
    cp, cq, cs = leaf(pc), leaf(q), leaf(s)
    mat = cp.repeat_interleave(reps, dim=0)
    if indexed:
        mat = mat.gather(1, idx.long().cpu().unsqueeze(-1).expand(-1, -1, 3))
    ref = O.pointcloud_project_fast(cfg, mat, cq, None, None, O.smoothing_kernel(cfg, 0.64), scaling_factor=cs)
    rloss, _ = O.proj_loss_pose_candidates(gt, ref["proj"], K)
    rloss.backward()
    close(runs[0][1], cp.grad, TOL, "several writers: dpc (K=%d reps=%d)" % (K, reps))
    close(runs[0][2], cq.grad, TOL, "several writers: dq")


@pytest.mark.parametrize("config", ["c2", "c4", "c5"])
def test_benchmarked_step_plan_vs_oracle_at_full_size(R, O, config):
    """THE entry point bench.py times by default -- dpc.render.project_loss_step(...).run(), one native call
    (dpc_project_loss_step) on static buffers -- at the sizes it times, against the oracle directly (not through its
    bit-identity with the autograd path at smaller sizes):
      c2  B=32, N=8000, 64^3, sigma_rel 0.64, one pose candidate per sample: oracle on 3 of the 32 clouds
      c4  the per-GPU shard of BASELINE configs[3]: 8 clouds x 16000 pts -> 128^3, sigma_rel 1.28: oracle on one cloud
      c5  16 samples x K=8 candidates SHARING their sample's point set (point_replicas = 8, the layout bench.py --config c5
          runs), min-of-K loss: winners from the device silhouettes, the oracle on one sample's 8 candidates
    loss, proj, d(pc), d(q), d(s) under the parity rule; two runs of the plan bit for bit.
    Reference call sequence: dpc/models/model_pc_to.py:239-282, 339-385, 410-440; dpc/run/train_to.py:122."""
    S, K, N, G, sig = {"c2": (32, 1, 8000, 64, 0.64), "c4": (8, 1, 16000, 128, 1.28), "c5": (16, 8, 8000, 64, 0.64)}[config]
    B = S * K
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=21)
    kern = R.smoothing_kernel(cfg, sig)
    d = torch.device("cuda")
    pcS, _, sS, gtS, _, _ = O.synth_inputs(S, N, G, {"c2": 1234, "c4": 404, "c5": 505}[config])
    q = O.synth_inputs(B, 1, G, 506)[1] if K > 1 else O.synth_inputs(S, N, G, {"c2": 1234, "c4": 404}[config])[1]
    s = sS.repeat_interleave(K, dim=0)
    plan = R.project_loss_step(cfg, kern, B, N, d, num_candidates=K, point_replicas=K)
    args = [dev(pcS), dev(q), dev(s), dev(gtS)]
    runs = []
    for _ in range(2):
        plan.run(*args)
        torch.cuda.synchronize()
        runs.append([x.clone() for x in (plan.loss, plan.proj, plan.dpc, plan.dq, plan.ds, plan.winner)])
    for name, a, b in zip(("loss", "proj", "dpc", "dq", "ds", "winner"), *runs):
        assert torch.equal(a, b), "%s step plan: %s differs between two runs" % (config, name)
    loss, proj, dpc, dq, ds, win = runs[0]
    assert torch.isfinite(proj).all() and torch.isfinite(dpc).all()
    # loss and winners from the device's own silhouettes with the reference's formula (all samples)
    rloss, rwin = O.proj_loss_pose_candidates(gtS, proj.double().cpu(), K)
    assert np.array_equal(win.cpu().numpy(), rwin.numpy())
    close(loss, rloss, TOL, "%s step plan: loss vs the reference formula on its silhouettes" % config)
    # the oracle itself on a few samples (fp64 CPU: seconds per cloud)
    for smp in {"c2": [0, 13, 31], "c4": [3], "c5": [5]}[config]:
        sl = slice(smp * K, (smp + 1) * K)
        cp = pcS[smp:smp + 1].clone().requires_grad_(True)
        cq, cs = q[sl].clone().requires_grad_(True), s[sl].clone().requires_grad_(True)
        ref = O.pointcloud_project_fast(cfg, cp.repeat_interleave(K, dim=0), cq, None, None, O.smoothing_kernel(cfg, sig),
                                        scaling_factor=cs)
        l1, w1 = O.proj_loss_pose_candidates(gtS[smp:smp + 1], ref["proj"], K)
        (l1 / S).backward()                        # the batch loss divides by the S samples
        assert w1.item() == int(win[smp])
        tag = "%s step plan, sample %d: " % (config, smp)
        close(proj[sl], ref["proj"], TOL, tag + "proj vs oracle")
        close(dpc[smp:smp + 1], cp.grad, TOL, tag + "dpc vs oracle")
        close(dq[sl], cq.grad, TOL, tag + "dq vs oracle")
        close(ds[sl], cs.grad, TOL, tag + "ds vs oracle")
    if K > 1:   # losing candidates: exact zeros in the small gradients
        lose = torch.ones(B, dtype=torch.bool)
        lose[torch.arange(S) * K + rwin] = False
        assert dq[lose.cuda()].abs().max().item() == 0.0 and ds[lose.cuda()].abs().max().item() == 0.0


def clamp_flip_analysis(O, cfg, sigma, pc, q, s, gt, reps, rel=1e-5, reach=4):
    """Oracle side of test_drc_clamp_threshold_flip_is_bounded_and_explained (K = 1, shared point sets of `reps` clouds).
    Returns a dict with
      near      [k,4] (cloud, z, y, x) of the voxels whose occupancy lies within `rel` RELATIVE of eps or 1 - eps (drc.py:57)
      touched   [S,N] bool: points of a set with a trilinear corner within `reach` voxels (Chebyshev) of such a voxel in one of
                the set's clouds -- the only points the decision about that voxel can reach (the Gaussian's taps beyond
                +-3 weigh < 1e-8 at sigma_rel 0.64, a point spreads over cell .. cell+1)
      base      the un-nudged oracle: per-cloud d(points) [B,N,3], dq [B,4], ds [B,1], proj
      variants  {cloud: [(bits, dpc_b [N,3], dq_b [4], ds_b [1]), ...]}: cloud b re-run alone with every assignment `bits`
                of pass (1) / block (0) to its near voxels, forced through O.DRC_CLAMP_NUDGE
    """
    import itertools

    B, S, N = q.shape[0], pc.shape[0], pc.shape[1]
    G = cfg.vox_size
    eps = cfg.drc_logsum_clip_val
    kern = O.smoothing_kernel(cfg, sigma)
    mat = pc.repeat_interleave(reps, dim=0).clone().requires_grad_(True)   # a leaf per cloud: every cloud's own contribution
    cq, cs = q.clone().requires_grad_(True), s.clone().requires_grad_(True)
    ref = O.pointcloud_project_fast(cfg, mat, cq, None, None, kern, scaling_factor=cs)
    (((ref["proj"] - gt) ** 2).sum() / B).backward()
    v = ref["voxels"].detach()[..., 0]
    near = (((v - eps).abs() <= rel * eps) | ((v - (1.0 - eps)).abs() <= rel * (1.0 - eps))).nonzero()
    tr = ref["tr_pc"].detach()
    cell = torch.floor((tr + 0.5) * (G - 1.0)).long()
    inside = ((tr >= -0.5) & (tr <= 0.5)).all(-1)
    touched = torch.zeros(S, N, dtype=torch.bool)
    for b, z, y, x in near.tolist():
        c = cell[b] - torch.tensor([z, y, x])
        hit = inside[b] & ((c >= -(reach + 1)) & (c <= reach)).all(-1)
        touched[b // reps] |= hit
    variants = {}
    for b in sorted(set(near[:, 0].tolist())):
        mine = [n for n in near.tolist() if n[0] == b]
        variants[b] = []
        for bits in itertools.product((0, 1), repeat=len(mine)):
            nudge = torch.zeros(1, G if cfg.vox_size_z == -1 else cfg.vox_size_z, G, G, 1, dtype=torch.float64)
            for (_, z, y, x), keep in zip(mine, bits):
                val = v[b, z, y, x].item()
                low = abs(val - eps) <= abs(val - (1.0 - eps))
                th = eps if low else 1.0 - eps
                inward = (1.0 + 1e-9) if low else (1.0 - 1e-9)          # just inside [eps, 1-eps] / just outside
                nudge[0, z, y, x, 0] = th * (inward if keep else 2.0 - inward) - val
            cp1, cq1, cs1 = (t[b:b + 1].detach().clone().requires_grad_(True) for t in (mat, q, s))
            O.DRC_CLAMP_NUDGE = nudge
            try:
                r1 = O.pointcloud_project_fast(cfg, cp1, cq1, None, None, kern, scaling_factor=cs1)
            finally:
                O.DRC_CLAMP_NUDGE = None
            (((r1["proj"] - gt[b:b + 1]) ** 2).sum() / B).backward()
            assert (r1["proj"].detach() - ref["proj"].detach()[b:b + 1]).abs().max().item() < 1e-9   # the nudge is invisible forward
            variants[b].append((bits, cp1.grad[0].double(), cq1.grad[0].double(), cs1.grad[0].double()))
    return dict(near=near, touched=touched, variants=variants, proj=ref["proj"].detach(),
                base=(mat.grad.double(), cq.grad.double(), cs.grad.double()))


def test_drc_clamp_threshold_flip_is_bounded_and_explained(R, O):
    """The reference's DRC clamp (dpc/util/drc.py:57, clamp(v, eps, 1 - eps)) switches a voxel's gradient on or off.  A voxel
    whose fp64 occupancy lies within the fp32 Gaussian's rounding of eps can be decided the other way on the device; this input
    (seed 9301, the one test_shared_sets_with_several_writers_are_bit_reproducible steers around) has two voxels within 1e-5
    RELATIVE of eps, one of them at +7e-7.  Instead of avoiding the input, the event is pinned down:
      * the near-threshold voxels are few and listed (oracle, fp64);
      * every point OUTSIDE their reach (no trilinear corner within 4 voxels) has the oracle's gradient, by the rule;
      * for each cloud holding such voxels, the device's d(points), d(q), d(s) equal -- by the rule -- the oracle re-run with
        ONE assignment of pass / block to those voxels (forced through O.DRC_CLAMP_NUDGE, a 1e-10 nudge no forward value sees).
    What stays outside the rule is reported, not hidden: the distance to the un-nudged oracle on the touched points."""
    S, N, G, reps = 3, 2500, 64, 4
    B = S * reps
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=21)
    pc = O.synth_inputs(S, N, G, 9301)[0]
    _, q, s, _, _, _ = O.synth_inputs(B, 4, G, 9400 + reps)
    gt = O.synth_inputs(B, 1, G, 9500)[3]
    A = clamp_flip_analysis(O, cfg, 0.64, pc, q, s, gt, reps)
    near, touched = A["near"], A["touched"]
    assert 1 <= len(near) <= 4, "expected a handful of near-threshold voxels in this input, found %d" % len(near)
    assert 0 < int(touched.sum()) <= 0.03 * S * N, "the voxels' reach is a small neighbourhood (%d points)" % int(touched.sum())
    gp, gq, gs = dev(pc, True), dev(q, True), dev(s, True)
    loss, out, _ = R.pointcloud_project_loss(cfg, gp, gq, None, None, R.smoothing_kernel(cfg, 0.64), scaling_factor=gs, gt=dev(gt),
                                             num_candidates=1)
    loss.backward()
    close(out["proj"], A["proj"], TOL, "clamp flip: proj")
    dpc, dq, ds = gp.grad.double().cpu(), gq.grad.double().cpu(), gs.grad.double().cpu()
    base_dpc, base_dq, base_ds = A["base"]
    set_sum = lambda per_cloud: per_cloud.reshape(S, reps, N, 3).sum(1)
    ref_dpc = set_sum(base_dpc)
    scale = max(1.0, float(ref_dpc.abs().max()))
    # (1) outside the reach of the near-threshold voxels: the oracle's gradient
    close(dpc[~touched], ref_dpc[~touched], TOL, "clamp flip: dpc outside the near-threshold voxels' reach")
    clean = [b for b in range(B) if b not in A["variants"]]
    close(dq[clean], base_dq[clean], TOL, "clamp flip: dq of the clouds without such a voxel")
    close(ds[clean], base_ds[clean], TOL, "clamp flip: ds of the clouds without such a voxel")
    # (2) inside: one assignment of the voxels' pass bits explains the device, for every gradient of the cloud
    import itertools
    sets = sorted(set(b // reps for b in A["variants"]))
    chosen = {}
    for j in sets:
        clouds = [b for b in A["variants"] if b // reps == j]
        best = None
        for combo in itertools.product(*[range(len(A["variants"][b])) for b in clouds]):
            per_cloud = base_dpc[j * reps:(j + 1) * reps].clone()
            errs = []
            for b, vi in zip(clouds, combo):
                bits, dpc_b, dq_b, ds_b = A["variants"][b][vi]
                per_cloud[b - j * reps] = dpc_b
                errs.append(float((dq[b] - dq_b).abs().max()) / max(1.0, float(base_dq.abs().max())))
                errs.append(float((ds[b] - ds_b).abs().max()) / max(1.0, float(base_ds.abs().max())))
            errs.append(float((dpc[j] - per_cloud.sum(0)).abs().max()) / scale)
            if best is None or max(errs) < best[0]:
                best = (max(errs), [A["variants"][b][vi][0] for b, vi in zip(clouds, combo)], per_cloud.sum(0))
        chosen[j] = best
        ERRORS.append(("clamp flip: gradients of point set %d under the explaining assignment %s (relative to the rule's scale)"
                       % (j, best[1]), best[0], 1.0))
        assert best[0] <= TOL, "no assignment of the near-threshold voxels explains the device gradients of set %d: %.3e" % (j, best[0])
    # what the flip is worth, reported (NOT held to the rule: it is the explained event)
    unexplained = float((dpc[touched] - ref_dpc[touched]).abs().max())
    ERRORS.append(("clamp flip: dpc vs the UN-NUDGED oracle on the %d touched points (the explained event; not a bound)"
                   % int(touched.sum()), unexplained, scale))
    flipped = {j: c[1] for j, c in chosen.items() if any(0 in bits for bits in c[1])}
    print("clamp flip: near-threshold voxels", near.tolist(), "assignments", {j: c[1] for j, c in chosen.items()},
          "flipped on the device:", flipped, "distance to the un-nudged oracle on touched points %.3e" % unexplained)


def test_pose_gradient_against_the_reference_and_its_exact_sum(R, O):
    """The case that sat AT the parity rule in round 2 (K = reps = 8 shared sets, N = 1300, |d(q)| = 1.67).  The reference's own
    d(q) -- torch sums the per-point terms of the first Hamilton product in fp32 (dpc/util/quaternion.py:69-86, 119-131) --
    deviates from the same gradient summed exactly by about the size of the rule.  Both references stay in the picture:
      * device vs the exact sum (oracle, EXACT_POSE_GRADIENT = True, pinned on CPU by
        tests/test_oracle_golden.py::test_exact_pose_gradient_mode_is_pinned):  <= rule
      * device vs the RAW reference (oracle default mode = the golden-pinned restatement):  <= rule + the raw reference's
        measured own deviation from its exact sum
    and all three numbers go into gpurun_out/parity_errors.json."""
    K = reps = 8
    S, N, G = 2, 1300, 32
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=11)
    pc = O.synth_inputs(S, N, G, 5100 + K)[0]
    _, q, s, _, _, _ = O.synth_inputs(S * reps, 4, G, 5200)
    gt = O.synth_inputs(S, 1, G, 5300)[3]
    refs = {}
    try:
        for exact in (False, True):
            O.EXACT_POSE_GRADIENT = exact
            cp, cq, cs = (x.clone().requires_grad_(True) for x in (pc, q, s))
            out = O.pointcloud_project_fast(cfg, cp.repeat_interleave(reps, dim=0), cq, None, None, O.smoothing_kernel(cfg, 0.9),
                                            scaling_factor=cs)
            loss, win = O.proj_loss_pose_candidates(gt, out["proj"], K)
            loss.backward()
            refs[exact] = (out["proj"].detach(), cq.grad.double(), cp.grad.double(), win, cs.grad.double())
    finally:
        O.EXACT_POSE_GRADIENT = True   # what the module fixture set
    assert torch.equal(refs[False][0], refs[True][0]), "the exact-sum mode must not change the reference's forward"
    own = float((refs[False][1] - refs[True][1]).abs().max())
    scale = max(1.0, float(refs[True][1].abs().max()))
    gp, gq, gs = dev(pc, True), dev(q, True), dev(s, True)
    loss, _, win = R.pointcloud_project_loss(cfg, gp, gq, None, None, R.smoothing_kernel(cfg, 0.9), scaling_factor=gs, gt=dev(gt),
                                             num_candidates=K)
    loss.backward()
    assert np.array_equal(win.cpu().numpy(), refs[True][3].numpy())
    err_exact = float((gq.grad.double().cpu() - refs[True][1]).abs().max())
    err_raw = float((gq.grad.double().cpu() - refs[False][1]).abs().max())
    ERRORS.append(("pose gradient: dq, device vs the reference's exact sum", err_exact, scale))
    ERRORS.append(("pose gradient: dq, device vs the RAW reference (bound: rule + the reference's own deviation)", err_raw, scale))
    ERRORS.append(("pose gradient: dq, the raw reference vs its own exact sum (not a device error)", own, scale))
    assert err_exact <= TOL * scale, (err_exact, scale)
    assert err_raw <= TOL * scale + own, (err_raw, own, scale)
    assert own > 0.3 * TOL * scale, "the reference's fp32 summation noise was expected to be visible in this case (%.2e)" % own
    close(gp.grad, refs[True][2], TOL, "pose gradient case: dpc")
    close(gp.grad, refs[False][2], TOL, "pose gradient case: dpc vs the raw reference")
    close(gs.grad, refs[False][4], TOL, "pose gradient case: ds vs the raw reference")


def test_randomised_shapes_vs_oracle(R, O):
    """Forty random configurations out of tools/fuzz_parity.py (grid side 16..64, a z side of its own, 1..21 taps, sigma 0.25..3.2,
    1/2/4 pose candidates, shared point sets, per-cloud point dropout, translation / focal-length inputs, up to 48 clouds and
    30 000 points): the fused loss call AND the reference-signature call against the oracle by the rule.  A case in which the device
    decided a voxel the other way at the DRC clamp's threshold is judged against the oracle with that voxel on the device's side
    (test_drc_clamp_threshold_flip_is_bounded_and_explained; the tool's sweeps of 2000 cases: profiles/r04_fuzz_parity.txt)."""
    import importlib.util

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(root, "tools", "fuzz_parity.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    worst = {}
    for idx in range(40):
        label, rs, notes = fuzz.one_case(7, idx)
        top = max(rs, key=rs.get)
        assert rs[top] <= 1.0, "%s: %s %.3f x the rule %s" % (label, top, rs[top], notes)
        for k, v in rs.items():
            worst[k] = max(worst.get(k, 0.0), v)
    for k, v in sorted(worst.items()):
        ERRORS.append(("randomised shapes (40 cases), worst error / bound: " + k, v * TOL, 1.0))


def test_zz_error_report():
    """Not a check: writes the worst observed error per quantity to gpurun_out/ for DESIGN.md."""
    worst = {}
    for what, err, scale in ERRORS:
        if err / scale >= worst.get(what, (0, 1, -1))[2]:
            worst[what] = (err, scale, err / scale)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_errors.json", "w") as fh:
        json.dump({k: dict(max_abs_err=v[0], ref_scale=v[1]) for k, v in sorted(worst.items())}, fh, indent=1)
