"""dpc.render -- MI355X-native differentiable point-cloud projection.

Drop-in for the projection path of NiteshBharadwaj/pytorch-unsup-pc: the functions below keep the
reference's names, argument order and defaults (reference file:line cited per function) and return tensors
of the same shapes.  They are also importable under the names the reference's caller really uses
(`util.point_cloud_to`, `util.drc`, `util.gauss_kernel`, `util.quaternion`) when this package's parent
directory is on sys.path -- the same arrangement as the reference's dpc/run/startup.py.

Differences from the reference, all deliberate (SURVEY.md section 8a "quirks"):
  * fp32 tensors on the device of the INPUTS (the reference promotes to fp64 and picks the device globally);
  * the Gaussian smoothing is applied (the reference's CUDA branch); `smooth=False` reproduces its CPU branch;
  * dead branches of the reference raise NotImplementedError naming the config key;
  * no `print` in the hot path; a point exactly on the +1/2 face contributes its in-range corners instead of
    raising IndexError.
"""
import numpy as np
import torch

from . import _native
from ._ops import (DeviceSchedule, Drc, Geometry, ProjectFused, ProjectLossFused, ProjectLossStep, SilhouetteLoss, Smooth, Splat,
                   Transform, status_word, taps_bucket)
from .predictions import chamfer_of_predictions, load_predictions, save_predictions  # noqa: F401

__all__ = [
    "pointcloud_project_fast", "pointcloud_project", "pc_perspective_transform", "pointcloud2voxels3d_fast",
    "smoothen_voxels3d", "smooth_voxels3d", "smoothing_kernel", "gauss_kernel_1d", "separable_kernels",
    "drc_projection", "drc_event_probabilities", "drc_depth_projection", "drc_depth_grid", "pc_point_dropout",
    "quaternion_rotate", "quaternion_multiply", "quaternion_conjugate", "quaternion_normalise",
    "get_smooth_sigma", "get_dropout_prob", "ProjectionOutputs", "silhouette_loss", "pointcloud_project_loss",
    "point_cloud_distance", "compute_distance", "chamfer_distances", "graphed_project_loss", "prefer_direct_graph_launch", "point_dropout_indices", "save_predictions", "load_predictions", "chamfer_of_predictions",
    "DeviceSchedule", "check_status", "set_debug_checks", "taps_bucket", "project_loss_step",
]


# ------------------------------------------------------------------------------------------------------
# config helpers (duck-typed cfg: any object with attribute access; fields of SURVEY.md section 5)
# ------------------------------------------------------------------------------------------------------
def _get(cfg, key, default):
    try:
        return getattr(cfg, key)
    except (AttributeError, KeyError):
        return default


def _grid(cfg):
    G = int(cfg.vox_size)
    vz = int(_get(cfg, "vox_size_z", -1))
    return (G if vz == -1 else vz), G, G


def _check_live_branches(cfg):
    if not _get(cfg, "pose_quaternion", True):
        raise NotImplementedError("pose_quaternion: false is a broken branch of the reference "
                                  "(dpc/util/point_cloud_to.py:153 UnboundLocalError)")
    if _get(cfg, "ptn_max_projection", False):
        raise NotImplementedError("ptn_max_projection: true returns a tuple in the reference (point_cloud_to.py:234)")
    if not _get(cfg, "drc_logsum", True) or not _get(cfg, "drc_tf_cumulative", True):
        raise NotImplementedError("drc_logsum: false / drc_tf_cumulative: false are dead branches of the reference "
                                  "(dpc/util/drc.py:45,84-92)")


_plain_geometries = {}   # Geometry objects of calls without a Gaussian, by their constants


def _geometry(cfg, kernel=None, schedule=None):
    """Per-call constants as a Geometry.  Geometry objects are immutable after construction and cache their own DpcParams
    blocks and buffer sizes per call shape, so they are reused: one per (kernel list, grid, camera constants) -- kept on the
    KernelList smoothing_kernel returned (a list of bare tensors, as a caller other than this package's smoothing_kernel
    may pass, is converted on every call as before)."""
    D, H, W = _grid(cfg)
    key = (D, H, W, _get(cfg, "camera_distance", 2.0), _get(cfg, "focal_length", 1.875), _get(cfg, "drc_logsum_clip_val", 1e-5),
           _get(cfg, "max_depth", 10.0), id(schedule))
    cache = (kernel.geometries if isinstance(kernel, KernelList) and kernel.untouched()
             else (_plain_geometries if kernel is None else None))
    if cache is not None:
        hit = cache.get(key)
        if hit is not None and hit.schedule is schedule:
            if kernel is not None and not _get(cfg, "pc_separable_gauss_filter", True):
                _kernel_taps(cfg, kernel)   # raises
            return hit
    kxy = kz = None
    if kernel is not None:
        kxy, kz = _kernel_taps(cfg, kernel)
    geom = Geometry(D, H, W, kxy, kz, key[3], key[4], key[5], key[6], schedule=schedule)
    if cache is not None:
        if len(cache) >= 16:
            cache.clear()
        cache[key] = geom
    return geom


# ------------------------------------------------------------------------------------------------------
# Bad point indices: an error, never a GPU fault         reference: fancy indexing raises IndexError
# ------------------------------------------------------------------------------------------------------
import os as _os

_debug_checks = _os.environ.get("DPC_RENDER_DEBUG", "0") not in ("", "0")


def set_debug_checks(on=True):
    """Debug mode (also DPC_RENDER_DEBUG=1): every call with a `point_index` checks its range on the HOST before anything
    is launched (one device-to-host synchronisation per call) and raises IndexError like the reference's fancy indexing
    (dpc/util/point_cloud_to.py:266-295).  Off, the default: the kernels drop an out-of-range entry (it never becomes an
    address) and flag it in the device status word, which check_status() turns into the same IndexError later."""
    global _debug_checks
    _debug_checks = bool(on)


def _validate_point_index(point_index, point_cloud):
    if point_index is None or not _debug_checks or point_index.numel() == 0:
        return
    lo, hi = int(point_index.min()), int(point_index.max())
    n = point_cloud.shape[1]
    if lo < 0 or hi >= n:
        raise IndexError("point_index holds %d (valid: 0 .. %d): index out of range for a point set of %d points"
                         % (lo if lo < 0 else hi, n - 1, n))


def check_status(device=None):
    """Read and clear the device status word (one synchronisation): raises IndexError when a `point_index` entry of any
    call since the last check was outside its point set -- the reference raises at the call itself; here the kernels dropped
    the point, flagged it and went on, and the error surfaces where the caller synchronises anyway (reading the loss, a
    checkpoint)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    word = status_word(dev)
    bits = int(word.item())
    if bits:
        word.zero_()
    if bits & _native.DPC_STATUS_BAD_INDEX:
        raise IndexError("dpc.render: a point_index entry was out of range for its point set (the point was dropped); "
                         "set DPC_RENDER_DEBUG=1 to find the call")
    return bits


def _kernel_taps(cfg, kernel):
    """Host tap arrays (x/y, z) from what smoothing_kernel returns: a list of three 5-D kernels
    [1,1,1,1,k], [1,1,1,k,1], [1,1,kz,1,1] applied in that order; a bare 1-D tensor is accepted too."""
    if not _get(cfg, "pc_separable_gauss_filter", True):
        raise NotImplementedError("pc_separable_gauss_filter: false leaves `kernel` unbound in the reference "
                                  "(dpc/util/gauss_kernel.py:52-55)")
    if isinstance(kernel, KernelList) and kernel.untouched():   # what smoothing_kernel returned: its host taps travel with it
        return kernel.taps
    host = lambda k: np.ascontiguousarray(k.detach().cpu().numpy() if isinstance(k, torch.Tensor) else k,
                                          dtype=np.float32).reshape(-1)
    if isinstance(kernel, (list, tuple)):
        if len(kernel) != 3:
            raise ValueError("kernel must be the list of 3 separable kernels returned by smoothing_kernel")
        kx, ky, kz = host(kernel[0]), host(kernel[1]), host(kernel[2])
        if kx.shape != ky.shape or not np.array_equal(kx, ky):
            raise NotImplementedError("different x and y kernels: smoothing_kernel never produces them")
        return kx, kz
    k = host(kernel)
    return k, k


# ------------------------------------------------------------------------------------------------------
# Gaussian kernels                                   reference: dpc/util/gauss_kernel.py
# ------------------------------------------------------------------------------------------------------
def gauss_kernel_1d(l, sig):
    """1-D Gaussian of side length l, fp32, normalised (dpc/util/gauss_kernel.py:5-11).  Host tensor: the
    weights travel to the GPU as kernel arguments, not as a device tensor."""
    x = torch.arange(float((-l) // 2) + 1.0, l // 2 + 1)
    k = torch.exp(-x ** 2 / (2.0 * sig ** 2))
    return k / k.sum()


class KernelList(list):
    """The list of three separable kernels smoothing_kernel returns -- a plain list to every caller (the reference passes it
    straight on to pointcloud_project_fast, dpc/models/model_pc_to.py:171-179, 262-265) that also carries what this package
    derives from it on every call: the host tap arrays (`taps`) and the per-grid Geometry objects built from them
    (`geometries`), so that a step pays for them once per kernel instead of once per call."""

    __slots__ = ("taps", "geometries", "_made_of")

    def __init__(self, items):
        super().__init__(items)
        kx, kz = (np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.float32).reshape(-1) for t in (items[0], items[2]))
        self.taps = (kx, kz)
        self.geometries = {}
        self._made_of = tuple(id(t) for t in items)

    def untouched(self):
        """Still the three tensors it was built from (a caller that replaced an element gets the slow, re-reading path)."""
        return len(self) == 3 and tuple(id(t) for t in self) == self._made_of


def separable_kernels(kernel):
    """[1,1,1,1,k], [1,1,1,k,1], [1,1,k,1,1] views of a 1-D kernel (dpc/util/gauss_kernel.py:27-32)."""
    n = kernel.shape[0]
    return KernelList([kernel.reshape((1, 1, 1, 1, n)), kernel.reshape((1, 1, 1, n, 1)), kernel.reshape((1, 1, n, 1, 1))])


_kernel_cache = {}     # (taps, sigma, z taps, z sigma) -> KernelList; a training loop asks for the same sigma_rel every step
                       # until the schedule moves it (model_pc_to.py:59-63), a benchmark loop for ever


def smoothing_kernel(cfg, sigma):
    """The three separable kernels for pc_gauss_kernel_size and sigma (in voxels)
    (dpc/util/gauss_kernel.py:35-55).  With vox_size_z != vox_size the z kernel has length
    floor(fsz*ratio)|1 and sigma*ratio -- what the reference intends; its own reshape at :49 raises.
    The result for a given (size, sigma) is memoised (the last few): the tensors are never modified by this package."""
    fsz = int(cfg.pc_gauss_kernel_size)
    vz = int(_get(cfg, "vox_size_z", -1))
    fz, ratio = fsz, 1.0
    if vz != -1:
        ratio = vz / int(cfg.vox_size)
        fz = int(np.floor(fsz * ratio))
        if fz % 2 == 0:
            fz += 1
    elif not _get(cfg, "pc_separable_gauss_filter", True):
        raise NotImplementedError("pc_separable_gauss_filter: false leaves `kernel` unbound in the reference "
                                  "(dpc/util/gauss_kernel.py:52-55)")
    key = (fsz, float(sigma), fz, float(ratio))
    hit = _kernel_cache.get(key)
    if hit is not None:
        return hit
    k = gauss_kernel_1d(fsz, sigma)
    if vz != -1:
        kz = k if (fz == fsz and ratio == 1.0) else gauss_kernel_1d(fz, sigma * ratio)
        out = KernelList([k.reshape((1, 1, 1, 1, fsz)), k.reshape((1, 1, 1, fsz, 1)), kz.reshape((1, 1, fz, 1, 1))])
    else:
        out = separable_kernels(k)
    if len(_kernel_cache) >= 8:
        _kernel_cache.pop(next(iter(_kernel_cache)))
    _kernel_cache[key] = out
    return out


# ------------------------------------------------------------------------------------------------------
# Quaternion helpers (host-side mirror; the fused kernels rotate in-register)   dpc/util/quaternion.py
# ------------------------------------------------------------------------------------------------------
def quaternion_multiply(a, b):
    """Hamilton product, (w,x,y,z) order (dpc/util/quaternion.py:69-86); 3-vectors get a leading 0."""
    pad = lambda v: torch.nn.functional.pad(v, (1, 0)) if v.shape[-1] == 3 else v
    a, b = pad(a), pad(b)
    if a.shape[-1] != 4 or b.shape[-1] != 4:
        raise ValueError("Can't create a quaternion: the last dimension must be 3 or 4.")
    aw, ax, ay, az = a.unbind(-1)
    bw, bx, by, bz = b.unbind(-1)
    return torch.stack((aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                        aw * by + ay * bw + az * bx - ax * bz, aw * bz + az * bw + ax * by - ay * bx), dim=-1)


def quaternion_conjugate(q):
    """[w, -x, -y, -z] (dpc/util/quaternion.py:89-92)."""
    return torch.cat((q[..., :1], -q[..., 1:]), dim=-1)  # no host constant: usable while a HIP graph is being captured


def quaternion_normalise(q):
    """q / |q| (dpc/util/quaternion.py:100-107)."""
    return q / q.norm(p=2, dim=-1, keepdim=True)


def quaternion_rotate(pc, q, inverse=False):
    """Rotate [B,N,3] points by [B,4] quaternions, norm not detached (dpc/util/quaternion.py:110-132)."""
    qn = quaternion_normalise(q).unsqueeze(1)
    qc = quaternion_conjugate(qn)
    w = quaternion_multiply(quaternion_multiply(qc, pc), qn) if inverse else quaternion_multiply(quaternion_multiply(qn, pc), qc)
    return w[:, :, 1:4]


# ------------------------------------------------------------------------------------------------------
# Stage functions                                    reference: dpc/util/point_cloud_to.py, dpc/util/drc.py
# ------------------------------------------------------------------------------------------------------
def pc_perspective_transform(cfg, point_cloud, transform, predicted_translation=None, focal_length=None):
    """[B,N,3] xyz -> [B,N,3] (z, y, x) camera-space coordinates (dpc/util/point_cloud_to.py:118-178)."""
    _check_live_branches(cfg)
    return Transform.apply(point_cloud, transform, predicted_translation, focal_length, _geometry(cfg))


def pointcloud2voxels3d_fast(cfg, pc, rgb=None):
    """Trilinear splat of transformed points into [B,D,H,W] (dpc/util/point_cloud_to.py:10-87).
    Returns (voxels, None) like the reference."""
    if rgb is not None:
        raise NotImplementedError("rgb splatting is a dead branch of the reference (point_cloud_to.py:64 AttributeError)")
    return Splat.apply(pc, _geometry(cfg)), None


def smoothen_voxels3d(cfg, voxels, kernel):
    """Separable zero-padded Gaussian over [B,1,D,H,W] in W,H,D order (dpc/util/point_cloud_to.py:90-103)."""
    D, H, W = voxels.shape[-3:]
    kxy, kz = _kernel_taps(cfg, kernel)
    return Smooth.apply(voxels, Geometry(D, H, W, kxy, kz))


smooth_voxels3d = smoothen_voxels3d


def _drc_geometry(voxels, cfg):
    B, D, H, W = voxels.shape[:4]
    return Geometry(D, H, W, None, None, _get(cfg, "camera_distance", 2.0), _get(cfg, "focal_length", 1.875),
                    _get(cfg, "drc_logsum_clip_val", 1e-5), _get(cfg, "max_depth", 10.0))


def drc_event_probabilities(voxels, cfg):
    """Ray-termination probabilities [D+1,B,H,W,1] of a [B,D,H,W,1] occupancy grid (dpc/util/drc.py:109-111)."""
    _check_live_branches(cfg)
    _, probs, _ = Drc.apply(voxels.squeeze(-1), _drc_geometry(voxels, cfg))
    return probs.unsqueeze(-1)


def drc_projection(voxels, cfg):
    """(silhouette [B,H,W,1], probabilities [D+1,B,H,W,1]) (dpc/util/drc.py:114-129)."""
    _check_live_branches(cfg)
    proj, probs, _ = Drc.apply(voxels.squeeze(-1), _drc_geometry(voxels, cfg))
    return proj.unsqueeze(-1), probs.unsqueeze(-1)


def drc_depth_grid(cfg, z_size):
    """psi_k = k/D - 1/2 + camera_distance, last = max_depth (dpc/util/drc.py:145-149)."""
    i = torch.arange(0, z_size, 1, dtype=torch.float64)
    return torch.cat([i / z_size - 0.5 + cfg.camera_distance, torch.tensor([cfg.max_depth], dtype=torch.float64)])


def drc_depth_projection(p, cfg):
    """Expected depth sum_k p_k psi_k of probabilities [D+1,B,H,W,1] (dpc/util/drc.py:152-160)."""
    psi = drc_depth_grid(cfg, p.shape[0] - 1).to(device=p.device, dtype=p.dtype).reshape(-1, 1, 1, 1, 1)
    return (p * psi).sum(0)


# ------------------------------------------------------------------------------------------------------
# The projection                                     reference: dpc/util/point_cloud_to.py:191-263
# ------------------------------------------------------------------------------------------------------
class ProjectionOutputs(dict):
    """The reference's output dict.  `proj` (what the training loss consumes) comes from the fused kernels; the other
    entries are produced on first access, differentiably, so a step that never reads them never pays for them -- and a
    caller that reads them every step (ModelPointCloud.compute_projection does, dpc/models/model_pc_to.py:266-269) pays for
    what it reads only: `builder` is one callable for all lazy keys, or a dict key -> callable; a callable returns a dict
    and may fill several keys at once."""

    _LAZY = ("voxels", "tr_pc", "drc_probs", "proj_depth")

    def __init__(self, proj, builder):
        super().__init__(proj=proj, voxels_rgb=None, proj_rgb=None)
        self._builders = builder if isinstance(builder, dict) else {k: builder for k in self._LAZY}
        for k in self._LAZY:
            dict.__setitem__(self, k, None)
        self._pending = set(self._LAZY)

    def _materialise(self, key):
        if key in self._pending:
            for k, v in self._builders[key]().items():
                if k in self._pending:
                    dict.__setitem__(self, k, v)
                    self._pending.discard(k)
            self._pending.discard(key)

    def __getitem__(self, key):
        self._materialise(key)
        return dict.__getitem__(self, key)

    def get(self, key, default=None):
        self._materialise(key)
        return dict.get(self, key, default)

    def _materialise_all(self):
        for k in list(self._pending):
            self._materialise(k)

    def values(self):
        self._materialise_all()
        return dict.values(self)

    def items(self):
        self._materialise_all()
        return dict.items(self)


_OUTSIDE = 2.0   # a coordinate outside the unit cube [-1/2, 1/2]^3: such a point splats nothing (point_cloud_to.py:26-27)


def _mask_dead_slots(tr, schedule):
    """Under a DeviceSchedule with a live-point count, a cloud's row has `capacity` slots of which only the first *n_live hold
    points (k_locate turns the others into out-of-bounds records).  The stage-level path does the same without reading the
    count on the host (so it stays capturable): the transformed coordinates of the dead slots are replaced by a constant
    outside the unit cube -- no splat, no gradient -- which is also what `tr_pc` shows in those slots."""
    if schedule is None or schedule.n_live is None:
        return tr
    dead = (torch.arange(tr.shape[1], device=tr.device, dtype=torch.int32) >= schedule.n_live).view(1, -1, 1)
    return torch.where(dead, torch.full_like(tr, _OUTSIDE), tr)


def _outputs_from_grid(cfg, geom, grid_wh, pc, q, t, f, s, point_index):
    """Lazy entries of the output dict derived from what the fused forward left behind: `tr_pc` is one transform launch;
    `voxels`, `drc_probs`, `proj_depth` start from grid_wh (the grid after clamp and the W, H passes, a differentiable
    output of the fused node): D pass, occupancy scale + clamp, DRC probabilities and depth -- no second transform, splat
    or W/H smoothing (reference: point_cloud_to.py:95-97 third pass, :218-222, :228-247)."""
    def tr_pc():
        pts = pc
        if pts.shape[0] != q.shape[0]:
            pts = pts.repeat_interleave(q.shape[0] // pts.shape[0], dim=0)
        if point_index is not None:
            pts = pts.gather(1, point_index.long().unsqueeze(-1).expand(-1, -1, 3))
        return {"tr_pc": _mask_dead_slots(Transform.apply(pts, q, t, f, geom), geom.schedule)}

    cache = {}

    def voxels():
        if "vox" not in cache:
            vox = grid_wh
            if geom.kz is not None:
                vox = Smooth.apply(vox, Geometry(geom.D, geom.H, geom.W, None, geom.kz, schedule=geom.schedule))
            if s is not None:
                vox = torch.clamp(vox * s.reshape(-1, 1, 1, 1).to(vox.dtype), 0.0, 1.0)
            cache["vox"] = vox
        return cache["vox"]

    def vox_entry():
        return {"voxels": voxels().unsqueeze(-1)}

    def drc_entries():
        _, probs, _ = Drc.apply(voxels(), geom)
        probs = torch.flip(probs, [2]).unsqueeze(-1)
        return {"drc_probs": probs, "proj_depth": drc_depth_projection(probs, cfg)}

    return {"tr_pc": tr_pc, "voxels": vox_entry, "drc_probs": drc_entries, "proj_depth": drc_entries}


def _project_staged(cfg, geom, pc, q, t, f, s, smooth, point_index=None):
    """The same chain composed from the stage-level kernels (all differentiable)."""
    if pc.shape[0] != q.shape[0]:  # shared point sets: the stage kernels want one cloud per pose
        pc = pc.repeat_interleave(q.shape[0] // pc.shape[0], dim=0)
    if point_index is not None:    # every cloud's own subset, materialised
        pc = pc.gather(1, point_index.long().unsqueeze(-1).expand(-1, -1, 3))
    tr = _mask_dead_slots(Transform.apply(pc, q, t, f, geom), geom.schedule)
    raw = Splat.apply(tr, geom)
    vox = torch.clamp(raw, 0.0, 1.0)
    if smooth and geom.kxy is not None:
        vox = Smooth.apply(vox, geom)
    if s is not None:
        vox = torch.clamp(vox * s.reshape(-1, 1, 1, 1).to(vox.dtype), 0.0, 1.0)
    proj, probs, _ = Drc.apply(vox, geom)
    probs = torch.flip(probs, [2]).unsqueeze(-1)
    depth = drc_depth_projection(probs, cfg)
    return {"proj": torch.flip(proj, [1]).unsqueeze(-1), "voxels": vox.unsqueeze(-1), "tr_pc": tr,
            "drc_probs": probs, "proj_depth": depth}


def pointcloud_project_fast(cfg, point_cloud, transform, predicted_translation, all_rgb, kernel=None,
                            scaling_factor=None, focal_length=None, smooth=True, point_index=None, schedule=None):
    """Project [B,N,3] point clouds to [B,H,W,1] silhouettes (dpc/util/point_cloud_to.py:191-263).

    Same positional signature as the reference; returns a dict with the reference's keys
    (proj, voxels, tr_pc, voxels_rgb, proj_rgb, drc_probs, proj_depth).  `smooth=False` reproduces the
    reference's CPU branch, which skips the Gaussian (:210-212).

    Shared point sets: `point_cloud` may be [B/R,N,3] while `transform` (and the other per-cloud
    inputs) have B rows -- clouds b*R .. b*R+R-1 then use point set b, the layout tf_repeat_0 produces for the views and
    pose candidates of one object (dpc/models/model_pc_to.py:302-306), without materialising the B copies; the
    gradient comes back as [B/R,N,3], summed over the replicas inside the backward kernel.

    Per-cloud point subsets (point dropout without copies): `point_index` [B,n] integer tensor -- cloud b projects the points
    point_cloud[b // R][point_index[b]] only.  This is pc_point_dropout applied AFTER tf_repeat_0 as the reference does
    (model_pc_to.py:254-258: every replica drops its own points) with neither the replicated [B,N,3] tensor nor the
    gathered [B,n,3] one; the gradient has the shape of `point_cloud` (zeros at points no cloud kept).  Indices from
    dpc.render.point_dropout_indices (device RNG) or any other source; they may repeat.  An entry outside the point set is
    an IndexError: at once with set_debug_checks() / DPC_RENDER_DEBUG=1, otherwise from check_status() (the kernels drop the
    point and flag it; nothing out of range is ever read or written).

    `schedule` (a DeviceSchedule): the Gaussian's tap values and the number of live points are read from device memory at
    run time -- for a call captured in a HIP graph whose sigma / keep-count follow a schedule; `kernel` then only fixes the
    compiled tap windows (see DeviceSchedule)."""
    if all_rgb is not None:
        raise NotImplementedError("all_rgb: the rgb branch of the reference is dead (point_cloud_to.py:64 AttributeError)")
    _check_live_branches(cfg)
    _validate_point_index(point_index, point_cloud)
    geom = _geometry(cfg, kernel if smooth else None, schedule if smooth else None)
    staged = lambda: _project_staged(cfg, geom, point_cloud, transform, predicted_translation, focal_length,
                                     scaling_factor, smooth, point_index)
    try:
        proj, grid_wh = ProjectFused.apply(point_cloud, transform, predicted_translation, focal_length, scaling_factor, geom,
                                           point_index, torch.is_grad_enabled())
    except _native.DpcError as e:
        if e.code != _native.DPC_ERR_TAPS:
            raise
        # effective Gaussian radius > 15 voxels: beyond the fused kernels' register window; same math, staged
        out = staged()
        return ProjectionOutputs(out["proj"], lambda: out)
    return ProjectionOutputs(proj, _outputs_from_grid(cfg, geom, grid_wh, point_cloud, transform, predicted_translation,
                                                      focal_length, scaling_factor, point_index))


pointcloud_project = pointcloud_project_fast


def pointcloud_project_loss(cfg, point_cloud, transform, predicted_translation, all_rgb, kernel=None,
                            scaling_factor=None, focal_length=None, gt=None, num_candidates=1, smooth=True, point_index=None,
                            schedule=None):
    """pointcloud_project_fast followed by the model's projection loss, as ONE autograd node.

    What ModelPointCloud does in two steps -- compute_projection (dpc/models/model_pc_to.py:239-282) then
    add_proj_loss / proj_loss_pose_candidates (:339-385, 410-440) -- with the loss folded into the ray-march
    kernels: `gt` [S,H,W,1] is the mask already pooled to the silhouette size, `point_cloud` holds
    S*num_candidates clouds (candidate-minor, like tf_repeat_0).  Returns (loss, outputs, winner): the scalar
    loss sum_s min_k sum (gt-pred)^2 / S, the usual output dict (`proj` from this pass, the rest lazy), and the
    winning candidate per sample.  Falls back to pointcloud_project_fast + silhouette_loss when the Gaussian is
    too long for the fused kernels.  `point_cloud` may hold shared point sets ([B/R,N,3]) and `point_index` per-cloud
    subsets of them (see pointcloud_project_fast)."""
    if all_rgb is not None:
        raise NotImplementedError("all_rgb: the rgb branch of the reference is dead (point_cloud_to.py:64 AttributeError)")
    if gt is None:
        raise ValueError("gt (pooled masks [S,H,W,1]) is required")
    _check_live_branches(cfg)
    _validate_point_index(point_index, point_cloud)
    geom = _geometry(cfg, kernel if smooth else None, schedule if smooth else None)
    staged = lambda: _project_staged(cfg, geom, point_cloud, transform, predicted_translation, focal_length,
                                     scaling_factor, smooth, point_index)
    try:
        loss, proj, winner = ProjectLossFused.apply(point_cloud, transform, predicted_translation, focal_length,
                                                    scaling_factor, gt, geom, num_candidates, point_index,
                                                    torch.is_grad_enabled())
    except _native.DpcError as e:
        if e.code != _native.DPC_ERR_TAPS:
            raise
        out = pointcloud_project_fast(cfg, point_cloud, transform, predicted_translation, None, kernel, scaling_factor,
                                      focal_length, smooth, point_index, schedule)
        loss, winner = silhouette_loss(out["proj"], gt, num_candidates)
        return loss, out, winner
    return loss, ProjectionOutputs(proj, staged), winner


def project_loss_step(cfg, kernel, num_clouds, num_points, device, schedule=None, num_candidates=1, point_replicas=1):
    """A ProjectLossStep plan for pointcloud_project_loss + backward at fixed shapes (`num_candidates` pose candidates per
    sample, `point_replicas` clouds per shared point set): static buffers, ONE native call per step that enqueues the four kernels -- instead of capturing the autograd path into a HIP
    graph (same kernels, same bits, 2-3 us per step faster than the replay on MI355X).  See dpc.render._ops.ProjectLossStep;
    reference call sequence: compute_projection + add_proj_loss + loss.backward() (dpc/models/model_pc_to.py:239-282, 339-385;
    dpc/run/train_to.py:122)."""
    _check_live_branches(cfg)
    return ProjectLossStep(_geometry(cfg, kernel, schedule), num_clouds, num_points, device, num_candidates, point_replicas)


def graphed_project_loss(cfg, kernel, point_cloud, transform, scaling_factor, gt, num_candidates=1):
    """pointcloud_project_loss for an eager training loop, captured into HIP graphs once.

    An eager call costs ~0.25-0.4 ms of Python, ctypes and allocator work for ~65 us of GPU time; this returns
    `step(point_cloud, transform, scaling_factor, gt) -> loss` built with torch.cuda.make_graphed_callables: forward and
    backward are each one graph launch, autograd works as usual (`loss.backward()`), results are the eager ones
    (bench: 379 -> 137 us per fwd+bwd at B=32, N=8000, 64^3).  The arguments are SAMPLE tensors fixing shapes, dtypes,
    device and requires_grad; what is frozen into the graphs: cfg, the Gaussian kernel (re-create the step when
    get_smooth_sigma has moved noticeably), num_candidates and the shapes.  Translation / focal-length inputs and the lazy
    output dict are not part of this shortcut -- use pointcloud_project_loss for those."""
    def step(pc, q, s, g):
        return pointcloud_project_loss(cfg, pc, q, None, None, kernel, scaling_factor=s, gt=g, num_candidates=num_candidates)[0]

    sample = tuple(t.detach().clone().requires_grad_(t.requires_grad) for t in (point_cloud, transform, scaling_factor, gt))
    return torch.cuda.make_graphed_callables(step, sample)


def pc_point_dropout(points, rgb, keep_prob):
    """Keep int(N*keep_prob) random points per cloud (dpc/util/point_cloud_to.py:269-295).  Same host RNG
    protocol as the reference (one np.random.choice(N, n, replace=False) per cloud, in batch order), so a
    seeded run selects the same points; the gather itself runs on the device."""
    B, Npts = points.shape[0], points.shape[1]
    n_out = int(Npts * keep_prob)
    idx = np.stack([np.random.choice(Npts, n_out, replace=False) for _ in range(B)]).astype(np.int64)
    idx_t = torch.from_numpy(idx).to(points.device)
    rows = torch.arange(B, device=points.device).unsqueeze(1)
    out_points = points[rows, idx_t]
    out_rgb = rgb[rows, idx_t] if rgb is not None else None
    return out_points, out_rgb


def prefer_direct_graph_launch():
    """Ask the ROCm runtime to replay HIP graphs as ordinary dispatches instead of pre-built AQL packets ("graph packet
    capture", the default of ROCm 7): on MI355X every kernel boundary inside a replayed graph is about 1 us shorter that way
    (the four-kernel fwd+bwd step: 61.2 -> 57.2 us).  The runtime reads the setting once, when HIP initialises, so this
    must run before the first CUDA call of the process (importing torch is fine); an explicit
    DEBUG_CLR_GRAPH_PACKET_CAPTURE in the environment wins.  Returns True when the setting is in force for this process."""
    import os

    if not torch.cuda.is_initialized():
        os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
    # once HIP is up the runtime has read the variable: what the process started with stays
    return os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") == "0"


def point_dropout_indices(num_clouds, num_points, keep_prob, device, generator=None, n_live=None):
    """Indices of pc_point_dropout (point_cloud_to.py:269-295) drawn on the device: for each of `num_clouds` clouds,
    int(num_points * keep_prob) DISTINCT point indices, a uniformly random subset (the same distribution as the reference's
    np.random.choice(replace=False), a different random stream), ascending.  int32 [num_clouds, n] for the `point_index`
    argument of pointcloud_project_fast / pointcloud_project_loss.

    Two 64-bit seed words come from torch's generator (`generator`, or the device's default one) and stay on the device;
    the choice itself is one kernel of this library (dpc_point_dropout_indices: hashed keys + radix select).  No host work,
    no upload, no sync, and safe inside HIP-graph capture: torch advances the generator at every replay, so every replay
    draws anew.  (torch.topk / sort are NOT used here: replayed from a graph they returned out-of-range indices at
    [128, 8000] on this ROCm build -- tests/test_gpu_parity.py::test_point_dropout_indices_in_a_replayed_graph.)

    `n_live` (int32 device tensor [1], e.g. DeviceSchedule.n_live): the rows keep their int(num_points * keep_prob) slots
    (the capacity) but only the first min(n_live, capacity) of each are filled, the count being read on the device -- a
    captured graph then follows the keep-probability schedule from replay to replay (hand the same tensor to the
    projection through its DeviceSchedule, which skips the unfilled slots)."""
    N = _native
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("dpc.render: point_dropout_indices draws on a GPU, got device %s" % device)
    keep = int(num_points * keep_prob)
    if not 0 <= keep <= num_points:
        raise ValueError("keep_prob %r leaves %d of %d points" % (keep_prob, keep, num_points))
    seed = torch.randint(-2 ** 62, 2 ** 62, (2,), dtype=torch.int64, device=device, generator=generator)
    out = torch.empty((num_clouds, keep), dtype=torch.int32, device=device)
    if n_live is not None:
        out.zero_()   # slots beyond the live count are never read by the kernels; zeros keep a debug range check quiet
    with torch.cuda.device(device):
        rc = N.lib().dpc_point_dropout_indices_live(int(num_clouds), int(num_points), keep, N.ptr(n_live), N.ptr(seed),
                                                    N.ptr(out), N.stream_ptr(device))
    N.check(rc, "dpc_point_dropout_indices_live")
    return out


# ------------------------------------------------------------------------------------------------------
# The caller's silhouette loss                       reference: dpc/models/model_pc_to.py:339-385, 410-440
# ------------------------------------------------------------------------------------------------------
def silhouette_loss(pred, gt, num_candidates=1):
    """Projection loss of ModelPointCloud.add_proj_loss, fused with its gradient in one kernel.

    pred [S*K,H,W,1] candidate silhouettes, gt [S,H,W,1] masks already pooled to H x W (the reference pools with
    AvgPool2d and permutes first, model_pc_to.py:348-369).  K = 1: sum (gt-pred)^2 / S.  K > 1
    (proj_loss_pose_candidates): per sample the candidate with the smallest sum of squared differences wins
    and only winners contribute.  Returns (loss, winner [S] int32)."""
    return SilhouetteLoss.apply(pred, gt, num_candidates)


# ------------------------------------------------------------------------------------------------------
# Schedules read by the caller                       reference: dpc/models/model_pc_to.py:59-87
# ------------------------------------------------------------------------------------------------------
def get_smooth_sigma(cfg, global_step):
    """sigma_rel(step), linear from pc_relative_sigma to pc_relative_sigma_end (model_pc_to.py:59-63)."""
    return cfg.pc_relative_sigma + global_step / cfg.max_number_of_steps * (cfg.pc_relative_sigma_end - cfg.pc_relative_sigma)


def get_dropout_prob(cfg, global_step):
    """Point keep-probability schedule (model_pc_to.py:68-87)."""
    if not cfg.pc_point_dropout_scheduled:
        return cfg.pc_point_dropout
    if cfg.pc_point_dropout_exponential_schedule:
        raise NotImplementedError("pc_point_dropout_exponential_schedule: true calls torch.log on floats in the reference")
    k0 = cfg.pc_point_dropout
    slope = (1.0 - k0) / (cfg.pc_point_dropout_end_step - cfg.pc_point_dropout_start_step)
    keep = slope * (global_step / cfg.max_number_of_steps) + (k0 - slope * cfg.pc_point_dropout_start_step)
    return max(min(keep, 1.0), k0)


# ------------------------------------------------------------------------------------------------------
# Evaluation side: nearest target point / Chamfer      reference: dpc/util/point_cloud_distance.py:25-40,
#                                           dpc/run/eval_chamfer_to.py:24-44, 119-123
# ------------------------------------------------------------------------------------------------------
def point_cloud_distance(Vs, Vt):
    """For each point of Vs [Ns,3] the closest point of Vt [Nt,3] (point_cloud_distance.py:25-40).

    Returns (proj [Ns,3] = Vt[idx], minDist [Ns], idx [Ns] int64) like the reference; fp32 or fp64 after the inputs.
    Nothing of size Ns x Nt is materialised.  No gradient (the reference only evaluates with it)."""
    dev = _native.require_device(Vs, Vt)
    if Vs.dim() != 2 or Vt.dim() != 2 or Vs.shape[1] != 3 or Vt.shape[1] != 3:
        raise ValueError("point_cloud_distance expects [Ns,3] and [Nt,3], got %s and %s" % (tuple(Vs.shape), tuple(Vt.shape)))
    dtype = torch.float64 if torch.float64 in (Vs.dtype, Vt.dtype) else torch.float32
    vs, vt = Vs.detach().to(dtype).contiguous(), Vt.detach().to(dtype).contiguous()
    ns, nt = vs.shape[0], vt.shape[0]
    if nt == 0 and ns > 0:
        raise IndexError("point_cloud_distance: empty target cloud (argmin of an empty sequence)")
    L = _native.lib()
    is64 = int(dtype == torch.float64)
    proj = torch.empty((ns, 3), dtype=dtype, device=dev)
    dist = torch.empty((ns,), dtype=dtype, device=dev)
    idx = torch.empty((ns,), dtype=torch.int64, device=dev)
    ws = torch.empty((max(L.dpc_nearest_workspace_bytes(ns, nt, is64), 16),), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = L.dpc_point_cloud_distance(_native.ptr(vs), _native.ptr(vt), ns, nt, is64, _native.ptr(proj), _native.ptr(dist),
                                        _native.ptr(idx), _native.ptr(ws), _native.stream_ptr(dev))
    _native.check(rc, "dpc_point_cloud_distance")
    return proj, dist, idx


def compute_distance(cfg, source_np, target_np, device=None):
    """eval_chamfer_to.py:24-44: numpy in, (min_dist, idx) numpy out (both float64 arrays, like the reference's
    np.concatenate onto float64 zeros).  The reference walks the source in cfg.pc_eval_chamfer_num_parts pieces to
    bound its [Ns,Nt,3] temporary; nothing of that size exists here, so the cloud goes through in one call."""
    device = torch.device("cuda") if device is None else device
    src = torch.from_numpy(np.ascontiguousarray(source_np)).to(device)
    tgt = torch.from_numpy(np.ascontiguousarray(target_np)).to(device)
    _, dist, idx = point_cloud_distance(src, tgt)
    return dist.cpu().numpy().astype(np.float64), idx.cpu().numpy().astype(np.float64)


def chamfer_distances(pred, gt):
    """The two directed means the evaluation reports per view (eval_chamfer_to.py:119-123):
    (mean_i min_j |pred_i - gt_j|, mean_j min_i |gt_j - pred_i|) as a float64 tensor [2] on the inputs' device."""
    p2g = point_cloud_distance(pred, gt)[1]
    g2p = point_cloud_distance(gt, pred)[1]
    return torch.stack([p2g.double().mean(), g2p.double().mean()])
