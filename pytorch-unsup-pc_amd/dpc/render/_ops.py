"""torch.autograd.Function wrappers over the C ABI: tensors in, tensors out, hand-written backward.

PyTorch is plumbing here (device memory, the current HIP stream, the autograd tape); every kernel is in
csrc/.  All work is enqueued on torch's current stream without synchronising, so a whole
forward+backward can be captured in a HIP graph.
"""
import ctypes

import numpy as np
import torch

from . import _native as N


class Geometry:
    """Per-call constants: grid, camera, tap weights (host side)."""

    __slots__ = ("D", "H", "W", "kxy", "kz", "camera_distance", "focal_length", "clip_val", "max_depth", "schedule",
                 "_params", "_kptrs")

    def __init__(self, D, H, W, kxy=None, kz=None, camera_distance=2.0, focal_length=1.875, clip_val=1e-5,
                 max_depth=10.0, schedule=None):
        """schedule: a DeviceSchedule (below) whose device-resident tap values / live-point count the kernels read at run
        time instead of this object's host arrays -- for captured HIP graphs that follow a sigma / dropout schedule."""
        self.schedule = schedule
        self.D, self.H, self.W = int(D), int(H), int(W)
        self.kxy = None if kxy is None else np.ascontiguousarray(kxy, dtype=np.float32).reshape(-1)
        self.kz = None if kz is None else np.ascontiguousarray(kz, dtype=np.float32).reshape(-1)
        if self.kxy is not None and self.kz is None:
            raise ValueError("an x/y kernel needs its z kernel (a z kernel alone is the D pass on a grid that has been through W and H)")
        for k in (self.kxy, self.kz):
            if k is not None and (k.size % 2 == 0 or k.size > N.DPC_MAX_TAPS):
                raise ValueError("smoothing kernels must have odd length <= %d, got %d" % (N.DPC_MAX_TAPS, k.size))
        self.camera_distance, self.focal_length = float(camera_distance), float(focal_length)
        self.clip_val, self.max_depth = float(clip_val), float(max_depth)
        self._params = {}     # (B, Npts, replicas) -> Sized DpcParams of the calls without a point_index
        self._kptrs = (None if self.kxy is None else self.kxy.ctypes.data, None if self.kz is None else self.kz.ctypes.data)

    def sized(self, B, Npts, point_replicas=1, point_index=None, n_src=0):
        """The DpcParams block of a call together with the buffer sizes the library derives from it (three native calls),
        as a `Sized`; cached per call shape when there is no per-call pointer in it (no point_index)."""
        if point_index is not None:
            return Sized(self.params(B, Npts, point_replicas, point_index, n_src))
        key = (B, Npts, point_replicas, n_src)
        hit = self._params.get(key)
        if hit is None:
            if len(self._params) >= 32:
                self._params.clear()
            hit = self._params[key] = Sized(self.params(B, Npts, point_replicas, None, n_src))
        return hit

    def params(self, B, Npts, point_replicas=1, point_index=None, n_src=0):
        """point_index: int32 device tensor [B,Npts] (kept alive by the caller for the duration of the call) selecting
        every cloud's points out of a stored set of n_src points."""
        sch = self.schedule
        dev = None if point_index is None else point_index.device
        return N.DpcParams(int(B), int(Npts), self.D, self.H, self.W,
                           0 if self.kxy is None else self.kxy.size, 0 if self.kz is None else self.kz.size,
                           self.camera_distance, self.focal_length, self.clip_val, self.max_depth, int(point_replicas),
                           int(n_src), None if point_index is None else point_index.data_ptr(),
                           None if dev is None or dev.type != "cuda" else status_word(dev).data_ptr(),
                           None if sch is None or sch.n_live is None else sch.n_live.data_ptr(),
                           None if sch is None or self.kxy is None else sch.taps_xy.data_ptr(),
                           None if sch is None or self.kz is None else sch.taps_z.data_ptr())

    def kern_ptrs(self):
        """Host addresses of the tap arrays (plain ints / None: the argtypes of the bindings take them as void*)."""
        return self._kptrs


class Sized:
    """A DpcParams block, a reference to it for the calls, and the sizes of the buffers a call with it needs."""

    __slots__ = ("P", "ref", "wpp", "cells_bytes", "ws_bytes", "bwd_rc")

    def __init__(self, P):
        L = N.lib()
        self.P, self.ref = P, ctypes.byref(P)
        self.wpp = L.dpc_mask_words_per_plane(self.ref)
        self.cells_bytes = max(L.dpc_cells_bytes(self.ref), 1)
        self.ws_bytes = max(L.dpc_workspace_bytes(self.ref), 1)
        self.bwd_rc = L.dpc_check_grid(self.ref, 1)   # 0, or why a backward could not follow a forward of this shape

    def require_backward(self, where):
        """A call whose inputs require gradients is refused BEFORE anything is launched when its backward could not run (grids
        wider than the backward's LDS tile): a forward that succeeds and a backward that then raises is the worse surprise."""
        if self.bwd_rc != 0:
            raise N.DpcError(self.bwd_rc, where + " (the inputs require gradients and the backward of this grid could not run)")


_status = {}


def status_word(device):
    """The int32 status word of `device` that every call with a point_index hands to the kernels (DpcParams.status): the
    DPC_STATUS_* bits are OR-ed into it on the device; read (and cleared) by dpc.render.check_status at a point where the
    caller synchronises anyway."""
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    w = _status.get(device)
    if w is None:
        w = _status[device] = torch.zeros(1, dtype=torch.int32, device=device)
    return w


class DeviceSchedule:
    """Per-step schedule values in DEVICE memory, so that a captured HIP graph follows them from replay to replay
    (dpc/models/model_pc_to.py:59-87, 171-179, 254-258: sigma_rel(step) and the dropout keep-probability are recomputed every
    step; a graph freezes kernel arguments, not memory).  taps_xy / taps_z hold the Gaussian's current values, n_live the
    current number of kept points per cloud.  update() is one tiny launch on the current stream (the values travel as its
    arguments) -- enqueue it in front of a replay.  `buckets` remembers the compiled tap windows the graph was captured
    for: fits() says whether new values still fit them."""

    def __init__(self, device, kxy, kz, n_live=None, capacity=None):
        kxy = np.ascontiguousarray(kxy, dtype=np.float32).reshape(-1)
        kz = np.ascontiguousarray(kz, dtype=np.float32).reshape(-1)
        self.device = torch.device(device)
        self.taps_xy = torch.zeros(N.DPC_MAX_TAPS, dtype=torch.float32, device=device)
        self.taps_z = torch.zeros(N.DPC_MAX_TAPS, dtype=torch.float32, device=device)
        self.n_live = None if n_live is None else torch.zeros(1, dtype=torch.int32, device=device)
        self.capacity = capacity
        self.sizes = (kxy.size, kz.size)
        self.buckets = (taps_bucket(kxy), taps_bucket(kz))
        self.update(kxy, kz, n_live)

    def fits(self, kxy, kz, n_live=None):
        kxy = np.ascontiguousarray(kxy, dtype=np.float32).reshape(-1)
        kz = np.ascontiguousarray(kz, dtype=np.float32).reshape(-1)
        if (kxy.size, kz.size) != self.sizes:
            return False
        bx, bz = taps_bucket(kxy), taps_bucket(kz)
        if bx < 0 or bz < 0 or bx > self.buckets[0] or bz > self.buckets[1]:
            return False
        return n_live is None or self.capacity is None or n_live <= self.capacity

    def tight(self, kxy, kz, n_live=None):
        """fits(), and a graph captured for the new values would be no cheaper (same tap windows, capacity not twice the
        live count)."""
        if not self.fits(kxy, kz, n_live):
            return False
        kxy = np.ascontiguousarray(kxy, dtype=np.float32).reshape(-1)
        kz = np.ascontiguousarray(kz, dtype=np.float32).reshape(-1)
        same = (taps_bucket(kxy), taps_bucket(kz)) == self.buckets
        return same and (n_live is None or self.capacity is None or 2 * n_live > self.capacity or self.capacity <= 512)

    def update(self, kxy, kz, n_live=None):
        kxy = np.ascontiguousarray(kxy, dtype=np.float32).reshape(-1)
        kz = np.ascontiguousarray(kz, dtype=np.float32).reshape(-1)
        if (kxy.size, kz.size) != self.sizes:
            raise ValueError("the schedule was built for kernels of %s taps, got %s" % (self.sizes, (kxy.size, kz.size)))
        if not self.fits(kxy, kz, n_live):
            # the kernels that read these values were compiled / captured for the tap windows `buckets` and rows of `capacity`
            # slots: taps outside the window would be dropped silently, points beyond the rows never read
            raise ValueError("schedule values do not fit what the schedule was built for: tap windows %s (needed %s), capacity %s "
                             "(n_live %s) -- build a new DeviceSchedule (and capture again)"
                             % (self.buckets, (taps_bucket(kxy), taps_bucket(kz)), self.capacity, n_live))
        with torch.cuda.device(self.device):
            rc = N.lib().dpc_schedule_update(kxy.ctypes.data_as(ctypes.c_void_p), kxy.size, kz.ctypes.data_as(ctypes.c_void_p),
                                             kz.size, 0 if n_live is None else int(n_live), N.ptr(self.taps_xy),
                                             N.ptr(self.taps_z), N.ptr(self.n_live), N.stream_ptr(self.device))
        N.check(rc, "dpc_schedule_update")


def taps_bucket(k):
    """Compiled tap-window radius a 1-D kernel needs in the fused kernels (-1: beyond them)."""
    k = np.ascontiguousarray(k, dtype=np.float32).reshape(-1)
    return N.lib().dpc_taps_bucket(k.ctypes.data_as(ctypes.c_void_p), k.size)


def _f32(t):
    if t is None:
        return None
    if t.dtype is torch.float32 and t.is_contiguous():
        return t.detach()
    return t.detach().to(torch.float32).contiguous()


def _dp(t):
    """Device address of a tensor as a plain int (None -> NULL): the bindings' argtypes turn it into a void*."""
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(dev):
    """torch's current HIP stream of `dev` as an address (the raw getter skips building a Stream object per call)."""
    if _raw_stream is not None:
        return _raw_stream(dev.index if dev.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(dev).cuda_stream


class _on:
    """`with torch.cuda.device(dev)` only when `dev` is not the current device already (the guard costs ~10 us of host time
    per call; a training process sits on its one device)."""

    __slots__ = ("guard",)

    def __init__(self, dev):
        self.guard = None if (dev.index is None or dev.index == torch.cuda.current_device()) else torch.cuda.device(dev)

    def __enter__(self):
        if self.guard is not None:
            self.guard.__enter__()

    def __exit__(self, *exc):
        if self.guard is not None:
            return self.guard.__exit__(*exc)
        return False


def _replicas(pc, q):
    """Clouds per point set: pc [S,N,3] shared by consecutive groups of B/S rows of q [B,4] (tf_repeat_0 order)."""
    B, S = q.shape[0], pc.shape[0]
    if S == B:
        return 1
    if S == 0 or B % S:
        raise ValueError("%d poses cannot share %d point sets: the number of clouds must be a multiple" % (B, S))
    return B // S


def _index32(point_index, pc, q):
    """Per-cloud point selection [B,n] -> contiguous int32 on the inputs' device (None passes through)."""
    if point_index is None:
        return None
    if point_index.dim() != 2 or point_index.shape[0] != q.shape[0]:
        raise ValueError("point_index must be [%d, n] (one row of point indices per cloud), got %s"
                         % (q.shape[0], tuple(point_index.shape)))
    if point_index.device != pc.device:
        raise RuntimeError("point_index lives on %s, the points on %s" % (point_index.device, pc.device))
    return point_index.detach().to(torch.int32).contiguous()


def _meta(t):
    """(dtype, shape) of an input -- kept on ctx instead of the tensor itself (no reference cycles)."""
    return None if t is None else (t.dtype, tuple(t.shape))


def _small(dsmall, col, width, B):
    """Dense [B,width] view of one gradient block of the small-gradient buffer (contiguous: autograd takes it
    without cloning)."""
    return dsmall.as_strided((B, width), (width, 1), col * B)   # one op instead of a slice and a view


def _like_input(grad, meta):
    """Gradients arrive in the input's dtype and shape, like autograd's would."""
    if grad is None or meta is None:
        return None
    if grad.dtype is meta[0] and tuple(grad.shape) == meta[1]:
        return grad
    return grad.to(meta[0]).reshape(meta[1])


def _zero_grads(metas, dev):
    """Backward of an empty batch (no clouds on this rank): zeros in every input's own dtype and shape."""
    return tuple(None if m is None else torch.zeros(m[1], dtype=m[0], device=dev) for m in metas)


def _new_cells(P, dev):
    return torch.empty((max(N.lib().dpc_cells_bytes(ctypes.byref(P)), 1),), dtype=torch.uint8, device=dev)


def decode_cells(cells, B, Npts, D):
    """Test helper: binned point records -> (code [B,N] int32, frac [B,N,3] float32, pts [B,N,3] float32) in
    original point order; also checks each chunk's z sort and bin offsets."""
    nblk = (Npts + 255) // 256
    chunk = 256 * 32 + ((2 * (D + 2) + 15) // 16) * 16
    raw = cells.cpu().numpy().reshape(B, nblk, chunk)
    code = np.full((B, Npts), -2, dtype=np.int32)
    frac = np.zeros((B, Npts, 3), dtype=np.float32)
    pts = np.zeros((B, Npts, 3), dtype=np.float32)
    for b in range(B):
        for k in range(nblk):
            n = min(256, Npts - 256 * k)
            rec = raw[b, k, :256 * 16].view(np.int32).reshape(256, 4)[:n]
            aux = raw[b, k, 256 * 16:256 * 32].view(np.int32).reshape(256, 4)[:n]
            perm = aux[:, 3]
            offs = raw[b, k, 256 * 32:256 * 32 + 2 * (D + 2)].view(np.uint16)
            assert offs[D + 1] == n and np.all(np.diff(offs.astype(np.int64)) >= 0)
            bins = np.where(rec[:, 0] < 0, D, rec[:, 0] >> 20)
            assert np.array_equal(np.sort(bins), bins), "records are not sorted by z bin"
            assert np.array_equal(np.searchsorted(bins, np.arange(D + 2)), offs), "bin offsets do not match the records"
            code[b, perm] = rec[:, 0]
            frac[b, perm] = rec[:, 1:].view(np.float32)
            pts[b, perm] = aux[:, :3].view(np.float32)
    assert (code != -2).all(), "some points are missing from the bins"
    return code, frac, pts


def locate_points(pc, q, t, f, geom):
    """First launch of the fused forward alone: (tr_pc [B,N,3] fp32, cells [B,N,4] int32 point records)."""
    dev = N.require_device(pc, q, t, f)
    pc32, q32, t32, f32 = _f32(pc), _f32(q), _f32(t), _f32(f)
    B, Npts = pc32.shape[0], pc32.shape[1]
    P = geom.params(B, Npts)
    tr = torch.empty_like(pc32)
    cells = _new_cells(P, dev)
    with torch.cuda.device(dev):
        rc = N.lib().dpc_locate(ctypes.byref(P), N.ptr(pc32), N.ptr(q32), N.ptr(t32), N.ptr(f32), N.ptr(tr), N.ptr(cells),
                                N.stream_ptr(dev))
    N.check(rc, "dpc_locate")
    return tr, cells


# ------------------------------------------------------------------------------------------------------
# Fused hot path
# ------------------------------------------------------------------------------------------------------
class ProjectFused(torch.autograd.Function):
    """pointcloud_project_fast as three launches forward, two backward (csrc/dpc_entry.hip and the kernel files it names).

    forward(pc [B,N,3], q [B,4], t [B,3]|None, f [B,1]|None, s [B,1]|None, geom) -> proj [B,H,W,1], grid_wh [B,D,H,W]
    Saved for backward: the binned point records, the grid after clamp + W/H passes, the clamp mask and the per-ray
    transmittance.  grid_wh (the grid after clamp and the W, H passes) is a differentiable output too: the other entries
    of the reference's output dict are derived from it on demand (D pass, scale/clamp, DRC probabilities) and their
    gradients come back through here, added to the silhouette's -- the chain never runs a second time.
    """

    @staticmethod
    def forward(ctx, pc, q, t, f, s, geom, point_index=None, want_grad=False):
        dev = N.require_device(pc, q, t, f, s)
        L = N.lib()
        pc32, q32, t32, f32, s32 = _f32(pc), _f32(q), _f32(t), _f32(f), _f32(s)
        idx = _index32(point_index, pc32, q32)
        B, reps = q32.shape[0], _replicas(pc32, q32)
        Npts = pc32.shape[1] if idx is None else idx.shape[1]
        Z = geom.sized(B, Npts, reps, idx, pc32.shape[1])
        # grad mode is always off INSIDE forward(): the caller passes what it saw outside
        if want_grad and B > 0 and any(x is not None and x.requires_grad for x in (pc, q, t, f, s)):
            Z.require_backward("dpc_project_fwd")
        grid_wh = torch.empty((B, geom.D, geom.H, geom.W), dtype=torch.float32, device=dev)
        mask = torch.empty((B, geom.D, Z.wpp), dtype=torch.int64, device=dev)
        cells = torch.empty((Z.cells_bytes,), dtype=torch.uint8, device=dev)
        proj = torch.empty((B, geom.H, geom.W, 1), dtype=torch.float32, device=dev)
        trans = torch.empty((B, geom.H, geom.W), dtype=torch.float32, device=dev)
        kxy, kz = geom.kern_ptrs()
        with _on(dev):
            rc = L.dpc_project_fwd(Z.ref, _dp(pc32), _dp(q32), _dp(t32), _dp(f32), _dp(s32), kxy, kz,
                                   None, _dp(cells), None, _dp(grid_wh), None, _dp(mask), _dp(proj), _dp(trans), _stream(dev))
        if rc != 0:
            N.check(rc, "dpc_project_fwd")
        ctx.geom = geom
        ctx.inputs = tuple(_meta(x) for x in (pc, q, t, f, s))
        empty = pc32.new_empty(0)
        ctx.save_for_backward(pc32, q32, t32 if t32 is not None else empty, f32 if f32 is not None else empty,
                              s32 if s32 is not None else empty, grid_wh, mask, cells, trans)
        ctx.has = (t is not None, f is not None, s is not None)
        ctx.npts, ctx.indexed = Npts, idx is not None
        ctx.set_materialize_grads(False)
        return proj, grid_wh

    @staticmethod
    def backward(ctx, dproj, dgrid):
        if dproj is None and dgrid is None:
            return None, None, None, None, None, None, None, None
        pc32, q32, t32, f32, s32, grid_wh, mask, cells, trans = ctx.saved_tensors
        has_t, has_f, has_s = ctx.has
        t32 = t32 if has_t else None
        f32 = f32 if has_f else None
        s32 = s32 if has_s else None
        geom = ctx.geom
        dev = pc32.device
        L = N.lib()
        B, reps = q32.shape[0], _replicas(pc32, q32)
        Z = geom.sized(B, ctx.npts, reps, cells if ctx.indexed else None, pc32.shape[1])  # the backward reads the source
        # index out of the binned records; any non-NULL pointer says "dpc is per stored set, accumulate"
        dproj32 = torch.zeros((B, geom.H, geom.W), dtype=torch.float32, device=dev) if dproj is None else _f32(dproj)
        dgrid32 = _f32(dgrid)
        dpc = torch.zeros_like(pc32) if (reps > 1 or ctx.indexed) else torch.empty_like(pc32)  # clouds add into a shared gradient
        dsmall = torch.empty((N.DPC_SMALL_COLS * B,), dtype=torch.float32, device=dev)
        ws = torch.empty((Z.ws_bytes,), dtype=torch.uint8, device=dev)
        kxy, kz = geom.kern_ptrs()
        with _on(dev):
            rc = L.dpc_project_bwd(Z.ref, _dp(pc32), _dp(q32), _dp(t32), _dp(f32), _dp(s32), kxy, kz,
                                   _dp(cells), _dp(grid_wh), _dp(mask), _dp(trans), _dp(dproj32), _dp(dgrid32),
                                   _dp(dpc), _dp(dsmall), _dp(ws), _stream(dev))
        if rc != 0:
            N.check(rc, "dpc_project_bwd")
        pc, q, t, f, s = ctx.inputs
        return (_like_input(dpc, pc), _like_input(_small(dsmall, N.COL_DQ, 4, B), q),
                _like_input(_small(dsmall, N.COL_DT, 3, B), t) if has_t else None,
                _like_input(_small(dsmall, N.COL_DF, 1, B), f) if has_f else None,
                _like_input(_small(dsmall, N.COL_DS, 1, B), s) if has_s else None, None, None, None)


class ProjectLossFused(torch.autograd.Function):
    """pointcloud_project_fast + the caller's silhouette loss (add_proj_loss / proj_loss_pose_candidates) in one
    autograd node.  With one pose candidate per sample and gradients required, the forward's ray-march kernel also runs
    the column half of the backward (DRC backward + adjoint D pass, for dloss = 1), so the whole step is 3 launches
    forward + 1 backward; otherwise 3 (+ finalize) forward + 2 backward.  The silhouette gradient is never
    materialised; losing pose candidates skip their backward.

    forward(pc, q, t, f, s, gt [S,H,W,1], geom, K) -> (loss [], proj [B,H,W,1], winner [S] int32)
    """

    @staticmethod
    def forward(ctx, pc, q, t, f, s, gt, geom, num_candidates, point_index=None, want_grad=True):
        dev = N.require_device(pc, q, t, f, s, gt)
        L = N.lib()
        pc32, q32, t32, f32, s32, gt32 = _f32(pc), _f32(q), _f32(t), _f32(f), _f32(s), _f32(gt)
        idx = _index32(point_index, pc32, q32)
        B, reps = q32.shape[0], _replicas(pc32, q32)
        Npts = pc32.shape[1] if idx is None else idx.shape[1]
        K = int(num_candidates)
        if K < 1 or B % K:
            raise ValueError("%d clouds is not a multiple of %d pose candidates" % (B, K))
        S = B // K
        if gt32.shape[0] != S or gt32.numel() != S * geom.H * geom.W:
            raise ValueError("gt must be [%d,%d,%d,1] (masks pooled to the silhouette size), got %s"
                             % (S, geom.H, geom.W, tuple(gt32.shape)))
        if B == 0:   # an empty shard (more ranks than samples): nothing to launch, loss 0, zero gradients
            ctx.empty, ctx.dev = True, dev
            ctx.inputs = tuple(_meta(x) for x in (pc, q, t, f, s))
            ctx.set_materialize_grads(False)
            proj = torch.zeros((0, geom.H, geom.W, 1), dtype=torch.float32, device=dev)
            winner = torch.zeros((0,), dtype=torch.int32, device=dev)
            ctx.mark_non_differentiable(proj, winner)
            return torch.zeros((), dtype=torch.float32, device=dev), proj, winner
        ctx.empty = False
        Z = geom.sized(B, Npts, reps, idx, pc32.shape[1])
        f32e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        grid_wh, proj, trans = f32e(B, geom.D, geom.H, geom.W), f32e(B, geom.H, geom.W, 1), f32e(B, geom.H, geom.W)
        sse, loss = f32e(B), torch.empty((), dtype=torch.float32, device=dev)
        sse_tiles = f32e(B, (geom.H * geom.W + 255) // 256)   # per-tile partials of the unfused ray march (fixed-order sum)
        mask = torch.empty((B, geom.D, Z.wpp), dtype=torch.int64, device=dev)
        winner = torch.empty((S,), dtype=torch.int32, device=dev)
        cells = torch.empty((Z.cells_bytes,), dtype=torch.uint8, device=dev)
        # backward buffers handed to the forward so it can run the column half of the backward right away -- only when a
        # backward can follow (grad mode is always off INSIDE forward(): the caller passes what it saw outside)
        want_grad = want_grad and any(x is not None and x.requires_grad for x in (pc, q, t, f, s))
        if want_grad:
            Z.require_backward("dpc_project_loss_fwd")
        ws = dsmall = None
        if want_grad and K == 1:
            ws = torch.empty((Z.ws_bytes,), dtype=torch.uint8, device=dev)
            dsmall = f32e(N.DPC_SMALL_COLS * B)
        fused = ctypes.c_int(0)
        kxy, kz = geom.kern_ptrs()
        with _on(dev):
            rc = L.dpc_project_loss_fwd(Z.ref, _dp(pc32), _dp(q32), _dp(t32), _dp(f32), _dp(s32), kxy, kz,
                                        _dp(gt32), K, None, _dp(cells), _dp(grid_wh), _dp(mask), _dp(proj),
                                        _dp(trans), _dp(sse), _dp(sse_tiles), _dp(loss), _dp(winner), _dp(ws), _dp(dsmall),
                                        ctypes.byref(fused), _stream(dev))
        if rc != 0:
            N.check(rc, "dpc_project_loss_fwd")
        ctx.geom, ctx.K, ctx.fused = geom, K, bool(fused.value)
        ctx.inputs = tuple(_meta(x) for x in (pc, q, t, f, s))
        empty = pc32.new_empty(0)
        ctx.save_for_backward(pc32, q32, t32 if t32 is not None else empty, f32 if f32 is not None else empty,
                              s32 if s32 is not None else empty, gt32, grid_wh, mask, cells, proj, trans, winner,
                              ws if ctx.fused else empty, dsmall if ctx.fused else empty)
        ctx.has = (t is not None, f is not None, s is not None)
        ctx.npts, ctx.indexed = Npts, idx is not None
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(proj, winner)
        return loss, proj, winner

    @staticmethod
    def backward(ctx, dloss, _dproj, _dwinner):
        if dloss is None:
            return (None,) * 10
        if ctx.empty:
            return _zero_grads(ctx.inputs, ctx.dev) + (None,) * 5
        pc32, q32, t32, f32, s32, gt32, grid_wh, mask, cells, proj, trans, winner, ws, dsmall = ctx.saved_tensors
        has_t, has_f, has_s = ctx.has
        t32, f32, s32 = (t32 if has_t else None), (f32 if has_f else None), (s32 if has_s else None)
        geom, dev, L = ctx.geom, pc32.device, N.lib()
        B, reps = q32.shape[0], _replicas(pc32, q32)
        Z = geom.sized(B, ctx.npts, reps, cells if ctx.indexed else None, pc32.shape[1])  # see ProjectFused.backward
        dl = dloss.detach().to(torch.float32).reshape(())
        dpc = torch.zeros_like(pc32) if (reps > 1 or ctx.indexed) else torch.empty_like(pc32)  # clouds add into a shared gradient
        # a FRESH block for dq / ds / dt / df on every call: k_gather_hw writes (never accumulates) them, and a block kept
        # across calls would alias the .grad tensors autograd took over from an earlier backward through this same forward
        out_small = torch.empty((N.DPC_SMALL_COLS * B,), dtype=torch.float32, device=dev)
        if not ctx.fused:
            ws = torch.empty((Z.ws_bytes,), dtype=torch.uint8, device=dev)
        kxy, kz = geom.kern_ptrs()
        with _on(dev):
            rc = L.dpc_project_loss_bwd(Z.ref, _dp(pc32), _dp(q32), _dp(t32), _dp(f32), _dp(s32), kxy, kz,
                                        _dp(cells), _dp(grid_wh), _dp(mask), _dp(proj), _dp(trans), _dp(gt32),
                                        ctx.K, _dp(winner), _dp(dl), int(ctx.fused), _dp(dpc), _dp(out_small),
                                        _dp(ws), _stream(dev))
        if rc != 0:
            N.check(rc, "dpc_project_loss_bwd")
        pc, q, t, f, s = ctx.inputs
        return (_like_input(dpc, pc), _like_input(_small(out_small, N.COL_DQ, 4, B), q),
                _like_input(_small(out_small, N.COL_DT, 3, B), t) if has_t else None,
                _like_input(_small(out_small, N.COL_DF, 1, B), f) if has_f else None,
                _like_input(_small(out_small, N.COL_DS, 1, B), s) if has_s else None, None, None, None, None, None)


class ProjectLossStep:
    """One training step's worth of the renderer -- pointcloud_project_loss and its backward -- as a PLAN: every buffer is
    allocated once and a call is ONE native call that enqueues the kernels on torch's current stream
    (dpc_project_loss_step, include/dpc_render.h: four launches with one pose candidate per sample, six with K).  What a loop
    uses instead of capturing the autograd path into a HIP graph: the same kernels and bits, no Python between the
    launches, and on MI355X 2-3 us per step faster than the replayed graph.  Gradients land in the plan's static tensors.

        plan = ProjectLossStep(geom, B, N, device)           # geom: dpc.render._geometry(cfg, kernel)
        loss = plan.run(pc, q, s, gt)                         # fp32 contiguous device tensors [B/R,N,3], [B,4], [B,1]|None, [B/K,H,W,1]
        plan.dpc, plan.dq, plan.ds (plan.dt, plan.df)         # d loss / d input, overwritten by every run; plan.proj, plan.winner

    num_candidates = K pose candidates per sample (min-of-K loss); point_replicas = R clouds share a point set (pc is
    [B/R,N,3], the gradient [B/R,N,3] summed over the replicas)."""

    def __init__(self, geom, B, Npts, device, num_candidates=1, point_replicas=1):
        L = N.lib()
        self.geom, self.B, self.N, self.device = geom, int(B), int(Npts), torch.device(device)
        self.K, self.R = int(num_candidates), int(point_replicas)
        if self.K < 1 or self.B % self.K or self.R < 1 or self.B % self.R:
            raise ValueError("%d clouds: not a multiple of %d candidates / %d replicas" % (self.B, self.K, self.R))
        dev = self.device
        self.P = geom.params(self.B, self.N, self.R)
        P = self.P
        f32e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        wpp = L.dpc_mask_words_per_plane(ctypes.byref(P))
        self.cells = _new_cells(P, dev)
        self.grid_wh = f32e(self.B, geom.D, geom.H, geom.W)
        self.mask = torch.empty((self.B, geom.D, wpp), dtype=torch.int64, device=dev)
        self.proj, self.trans, self.sse = f32e(self.B, geom.H, geom.W, 1), f32e(self.B, geom.H, geom.W), f32e(self.B)
        self.sse_tiles = f32e(self.B, (geom.H * geom.W + 255) // 256)
        self.loss = torch.zeros((), dtype=torch.float32, device=dev)
        self.winner = torch.empty((self.B // self.K,), dtype=torch.int32, device=dev)
        self.ws = torch.empty((max(L.dpc_workspace_bytes(ctypes.byref(P)), 1),), dtype=torch.uint8, device=dev)
        self.fwd_dsmall, self.dsmall = f32e(N.DPC_SMALL_COLS * self.B), f32e(N.DPC_SMALL_COLS * self.B)
        self.dpc = torch.zeros((self.B // self.R, self.N, 3), dtype=torch.float32, device=dev)
        self.dq, self.ds = _small(self.dsmall, N.COL_DQ, 4, self.B), _small(self.dsmall, N.COL_DS, 1, self.B)
        self.dt, self.df = _small(self.dsmall, N.COL_DT, 3, self.B), _small(self.dsmall, N.COL_DF, 1, self.B)
        self._kxy, self._kz = geom.kern_ptrs()
        self._fixed = (N.ptr(self.cells), N.ptr(self.grid_wh), N.ptr(self.mask), N.ptr(self.proj), N.ptr(self.trans), N.ptr(self.sse),
                       N.ptr(self.sse_tiles), N.ptr(self.loss), N.ptr(self.winner), N.ptr(self.ws), N.ptr(self.fwd_dsmall))
        self._tail = (N.ptr(self.dpc), N.ptr(self.dsmall))
        self._fn = L.dpc_project_loss_step
        self._pref = ctypes.byref(self.P)

    @staticmethod
    def _arg(t, shape, name):
        if t is None:
            return None
        if t.dtype != torch.float32 or not t.is_contiguous() or tuple(t.shape) != shape or not t.is_cuda:
            raise ValueError("%s must be a contiguous float32 device tensor of shape %s, got %s %s"
                             % (name, shape, t.dtype, tuple(t.shape)))
        return ctypes.c_void_p(t.data_ptr())

    def bind(self, pc, q, s, gt, t=None, f=None, dloss=None):
        """Fix the input tensors (static buffers that are refilled in place): run() without arguments then skips the
        per-call checks and pointer conversions."""
        B, g = self.B, self.geom
        self._bound = (self._pref, self._arg(pc, (B // self.R, self.N, 3), "pc"), self._arg(q, (B, 4), "q"),
                       self._arg(t, (B, 3), "t"), self._arg(f, (B, 1), "f"), self._arg(s, (B, 1), "s"), self._kxy, self._kz,
                       self._arg(gt, (B // self.K, g.H, g.W, 1), "gt"), self.K) + self._fixed \
            + (None if dloss is None else ctypes.c_void_p(dloss.data_ptr()),) + self._tail
        self._keep = (pc, q, s, gt, t, f, dloss)
        return self

    def run(self, pc=None, q=None, s=None, gt=None, t=None, f=None, dloss=None):
        """Enqueue forward + backward on torch's current stream; returns the loss tensor (static, no sync)."""
        if pc is not None:
            self.bind(pc, q, s, gt, t, f, dloss)
        if self.R > 1:
            self.dpc.zero_()    # the replicas ADD into the shared gradient (include/dpc_render.h, point_replicas)
        rc = self._fn(*self._bound, ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        if rc != 0:
            N.check(rc, "dpc_project_loss_step")
        return self.loss


# ------------------------------------------------------------------------------------------------------
# Stage-level functions (one per reference function)
# ------------------------------------------------------------------------------------------------------
class Transform(torch.autograd.Function):
    """pc_perspective_transform, quaternion branch."""

    @staticmethod
    def forward(ctx, pc, q, t, f, geom):
        dev = N.require_device(pc, q, t, f)
        pc32, q32, t32, f32 = _f32(pc), _f32(q), _f32(t), _f32(f)
        P = geom.params(pc32.shape[0], pc32.shape[1])
        out = torch.empty_like(pc32)
        with torch.cuda.device(dev):
            rc = N.lib().dpc_transform_fwd(ctypes.byref(P), N.ptr(pc32), N.ptr(q32), N.ptr(t32), N.ptr(f32), N.ptr(out),
                                           N.stream_ptr(dev))
        N.check(rc, "dpc_transform_fwd")
        ctx.geom, ctx.inputs, ctx.saved = geom, tuple(_meta(x) for x in (pc, q, t, f)), (pc32, q32, t32, f32)
        return out

    @staticmethod
    def backward(ctx, dout):
        pc32, q32, t32, f32 = ctx.saved
        dev = pc32.device
        P = ctx.geom.params(pc32.shape[0], pc32.shape[1])
        dout32 = dout.detach().to(torch.float32).contiguous()
        dpc = torch.empty_like(pc32)
        dsmall = torch.empty((N.DPC_SMALL_COLS * pc32.shape[0],), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = N.lib().dpc_transform_bwd(ctypes.byref(P), N.ptr(pc32), N.ptr(q32), N.ptr(t32), N.ptr(f32),
                                           N.ptr(dout32), N.ptr(dpc), N.ptr(dsmall), N.stream_ptr(dev))
        N.check(rc, "dpc_transform_bwd")
        B = pc32.shape[0]
        pc, q, t, f = ctx.inputs
        return (_like_input(dpc, pc), _like_input(_small(dsmall, N.COL_DQ, 4, B), q),
                _like_input(_small(dsmall, N.COL_DT, 3, B), t), _like_input(_small(dsmall, N.COL_DF, 1, B), f), None)


class Splat(torch.autograd.Function):
    """pointcloud2voxels3d_fast: tr [B,N,3] (z,y,x) -> voxels [B,D,H,W]."""

    @staticmethod
    def forward(ctx, tr, geom):
        dev = N.require_device(tr)
        is64 = tr.dtype == torch.float64  # the reference's direct callers pass fp64 coordinates; keep them
        trc = tr.detach().contiguous() if is64 else _f32(tr)
        P = geom.params(trc.shape[0], trc.shape[1])
        vox = torch.empty((trc.shape[0], geom.D, geom.H, geom.W), dtype=torch.float32, device=dev)
        cells = _new_cells(P, dev)
        with torch.cuda.device(dev):
            rc = N.lib().dpc_splat_fwd(ctypes.byref(P), N.ptr(trc), int(is64), N.ptr(cells), N.ptr(vox), N.stream_ptr(dev))
        N.check(rc, "dpc_splat_fwd")
        ctx.geom, ctx.tr, ctx.trc, ctx.is64 = geom, _meta(tr), trc, is64
        return vox

    @staticmethod
    def backward(ctx, dvox):
        trc = ctx.trc
        dev = trc.device
        P = ctx.geom.params(trc.shape[0], trc.shape[1])
        dvox32 = dvox.detach().to(torch.float32).contiguous()
        dtr = torch.empty(trc.shape, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = N.lib().dpc_splat_bwd(ctypes.byref(P), N.ptr(trc), int(ctx.is64), N.ptr(dvox32), N.ptr(dtr),
                                       N.stream_ptr(dev))
        N.check(rc, "dpc_splat_bwd")
        return _like_input(dtr, ctx.tr), None


def _smooth_call(x32, geom, transpose):
    """geom.kxy None: the D pass alone."""
    dev = x32.device
    B = x32.shape[0]
    P = geom.params(B, 0)
    out, tmp = torch.empty_like(x32), torch.empty_like(x32)
    kxy, kz = geom.kern_ptrs()
    with torch.cuda.device(dev):
        rc = N.lib().dpc_smooth(ctypes.byref(P), kxy, kz, int(transpose), N.ptr(x32), N.ptr(out), N.ptr(tmp),
                                N.stream_ptr(dev))
    N.check(rc, "dpc_smooth")
    return out


class Smooth(torch.autograd.Function):
    """smoothen_voxels3d on a [B,D,H,W] (or [B,1,D,H,W]) grid; the backward is the adjoint correlation."""

    @staticmethod
    def forward(ctx, vox, geom):
        N.require_device(vox)
        ctx.geom, ctx.vox = geom, _meta(vox)
        return _smooth_call(_f32(vox).reshape(-1, geom.D, geom.H, geom.W), geom, False).reshape(vox.shape)

    @staticmethod
    def backward(ctx, dout):
        geom = ctx.geom
        d32 = dout.detach().to(torch.float32).contiguous().reshape(-1, geom.D, geom.H, geom.W)
        return _like_input(_smooth_call(d32, geom, True), ctx.vox), None


class Drc(torch.autograd.Function):
    """drc_projection + drc_depth_projection: vox [B,D,H,W] -> proj [B,H,W], probs [D+1,B,H,W], depth [B,H,W]
    (no flips; the caller applies the reference's flips)."""

    @staticmethod
    def forward(ctx, vox, geom):
        dev = N.require_device(vox)
        vox32 = _f32(vox)
        B = vox32.shape[0]
        P = geom.params(B, 0)
        proj = torch.empty((B, geom.H, geom.W), dtype=torch.float32, device=dev)
        probs = torch.empty((geom.D + 1, B, geom.H, geom.W), dtype=torch.float32, device=dev)
        depth = torch.empty((B, geom.H, geom.W), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = N.lib().dpc_drc_fwd(ctypes.byref(P), N.ptr(vox32), N.ptr(proj), N.ptr(probs), N.ptr(depth),
                                     N.stream_ptr(dev))
        N.check(rc, "dpc_drc_fwd")
        ctx.geom, ctx.vox, ctx.vox32 = geom, _meta(vox), vox32
        ctx.set_materialize_grads(False)  # unused outputs arrive as None, not as zero-filled [D+1,B,H,W] tensors
        return proj, probs, depth

    @staticmethod
    def backward(ctx, dproj, dprobs, ddepth):
        vox32 = ctx.vox32
        dev = vox32.device
        P = ctx.geom.params(vox32.shape[0], 0)
        c = lambda g: None if g is None else g.detach().to(torch.float32).contiguous()
        dproj, dprobs, ddepth = c(dproj), c(dprobs), c(ddepth)
        dvox = torch.empty_like(vox32)
        with torch.cuda.device(dev):
            rc = N.lib().dpc_drc_bwd(ctypes.byref(P), N.ptr(vox32), N.ptr(dproj), N.ptr(dprobs), N.ptr(ddepth),
                                     N.ptr(dvox), N.stream_ptr(dev))
        N.check(rc, "dpc_drc_bwd")
        return _like_input(dvox, ctx.vox), None


class SilhouetteLoss(torch.autograd.Function):
    """add_proj_loss / proj_loss_pose_candidates fused with its gradient (one launch, gradient precomputed in
    the forward; backward scales it by the incoming scalar)."""

    @staticmethod
    def forward(ctx, pred, gt, num_candidates):
        dev = N.require_device(pred, gt)
        p32, g32 = _f32(pred), _f32(gt)
        S = g32.shape[0]
        K = int(num_candidates)
        if p32.shape[0] != S * K:
            raise ValueError("pred has %d silhouettes, expected %d samples x %d candidates" % (p32.shape[0], S, K))
        n_pix = g32[0].numel() if S else 1
        if S and p32[0].numel() != n_pix:
            raise ValueError("gt and pred silhouettes differ in size: %s vs %s" % (tuple(g32.shape), tuple(p32.shape)))
        part = torch.empty((S,), dtype=torch.float32, device=dev)
        winner = torch.empty((S,), dtype=torch.int32, device=dev)
        dpred = torch.empty_like(p32)
        with torch.cuda.device(dev):
            rc = N.lib().dpc_silhouette_loss(N.ptr(g32), N.ptr(p32), S, K, n_pix, N.ptr(part), N.ptr(winner), N.ptr(dpred),
                                             N.stream_ptr(dev))
        N.check(rc, "dpc_silhouette_loss")
        ctx.dpred, ctx.meta = dpred, _meta(pred)
        ctx.mark_non_differentiable(winner)
        return part.sum(), winner

    @staticmethod
    def backward(ctx, dloss, _dwinner):
        return _like_input(ctx.dpred * dloss, ctx.meta), None, None
