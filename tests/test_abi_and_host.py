"""CPU-side checks (no GPU, no compute calls): the C-ABI library loads and exports every symbol that
include/dpc_render.h declares; host-side logic of the Python mirror (kernels, config validation, lazy outputs,
sharding helpers) behaves like the reference's."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "dpc_render.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dpc_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    from dpc.render import _native

    lib = ctypes.CDLL(_native.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libdpc_render.so does not export %s" % n
    assert sorted(_native.SYMBOLS) == names, "binding list and header disagree: %s" % (set(_native.SYMBOLS) ^ set(names))


def test_abi_version_and_error_strings():
    from dpc.render import _native

    L = _native.lib()
    text = open(HEADER).read()
    assert L.dpc_abi_version() == int(re.search(r"#define DPC_ABI_VERSION (\d+)", text).group(1)) == _native.ABI_VERSION
    for code in range(0, -7, -1):
        assert _native.strerror(code) and "unknown" not in _native.strerror(code)
    assert "unknown" in _native.strerror(-99)


def test_size_queries_and_validation_without_gpu():
    from dpc.render import _native

    L = _native.lib()
    P = _native.DpcParams(32, 8000, 64, 64, 64, 21, 21, 2.0, 1.875, 1e-5, 10.0, 1)
    assert L.dpc_mask_words_per_plane(ctypes.byref(P)) == 64
    assert L.dpc_cells_bytes(ctypes.byref(P)) == 32 * 32 * (256 * 32 + 144)
    ws = L.dpc_workspace_bytes(ctypes.byref(P))
    assert ws >= 32 * 64 ** 3 * 4 and ws % 256 == 0
    bad = _native.DpcParams(1, 10, 2048, 64, 64, 0, 0, 2.0, 1.875, 1e-5, 10.0, 1)  # D beyond the 10-bit cell index
    assert L.dpc_workspace_bytes(ctypes.byref(bad)) == 0
    # ... and a cloud of 2^20 points could wrap one voxel's 64-bit fixed-point sum: refused, 2^20 - 1 is the most
    many = _native.DpcParams(1, _native.DPC_MAX_POINTS + 1, 64, 64, 64, 0, 0, 2.0, 1.875, 1e-5, 10.0, 1)
    assert L.dpc_cells_bytes(ctypes.byref(many)) == 0 and L.dpc_check_grid(ctypes.byref(many), 1) == _native.DPC_ERR_SHAPE
    most = _native.DpcParams(1, _native.DPC_MAX_POINTS, 64, 64, 64, 0, 0, 2.0, 1.875, 1e-5, 10.0, 1)
    assert L.dpc_cells_bytes(ctypes.byref(most)) == 4096 * (256 * 32 + 144) and L.dpc_check_grid(ctypes.byref(most), 1) == 0
    assert "1048575" in _native.strerror(_native.DPC_ERR_SHAPE)
    header = open(os.path.join(ROOT, "include", "dpc_render.h")).read()
    assert "#define DPC_MAX_POINTS %d" % _native.DPC_MAX_POINTS in header
    # argument validation happens before any launch: NULL pointers / even tap counts are refused on CPU too
    assert L.dpc_project_fwd(ctypes.byref(P), *([None] * 16)) == -1  # 16 pointer arguments
    even = _native.DpcParams(1, 10, 16, 16, 16, 4, 4, 2.0, 1.875, 1e-5, 10.0, 1)
    assert L.dpc_transform_fwd(ctypes.byref(even), *([None] * 6)) == -3


def test_cpu_tensors_are_refused_loudly():
    import dpc.render as R
    from oracle.dpc_oracle import Cfg

    with pytest.raises(RuntimeError, match="MI355X only"):
        R.pointcloud_project_fast(Cfg(vox_size=16), torch.zeros(1, 4, 3), torch.ones(1, 4), None, None)
    with pytest.raises(RuntimeError, match="MI355X only"):
        R.pointcloud2voxels3d_fast(Cfg(vox_size=16), torch.zeros(1, 4, 3), None)


def test_missing_library_fails_loudly(monkeypatch):
    from dpc.render import _native

    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", "/nonexistent/libdpc_render.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _native.lib()


def test_kernels_match_oracle_bitwise():
    import dpc.render as R
    from oracle import dpc_oracle as O

    for l, sig in [(11, 1.0), (21, 3.0), (21, 0.64), (10, 1.5)]:
        assert torch.equal(R.gauss_kernel_1d(l, sig), O.gauss_kernel_1d(l, sig))
    for cfg in (O.Cfg(vox_size=64, pc_gauss_kernel_size=21), O.Cfg(vox_size=32, vox_size_z=16, pc_gauss_kernel_size=11)):
        for a, b in zip(R.smoothing_kernel(cfg, 1.5), O.smoothing_kernel(cfg, 1.5)):
            assert a.shape == b.shape and torch.equal(a, b)
    with pytest.raises(NotImplementedError, match="pc_separable_gauss_filter"):
        R.smoothing_kernel(O.Cfg(pc_separable_gauss_filter=False), 1.0)


def test_schedules_and_quaternion_helpers():
    import dpc.render as R
    from oracle import dpc_oracle as O

    cfg = O.Cfg(pc_relative_sigma=3.0, pc_point_dropout=0.07)
    for step in (0, 100000, 600000):
        assert R.get_smooth_sigma(cfg, step) == O.get_smooth_sigma(cfg, step)
        assert R.get_dropout_prob(cfg, step) == O.get_dropout_prob(cfg, step)
    g = torch.Generator().manual_seed(3)
    pc, q = torch.rand(2, 5, 3, generator=g, dtype=torch.float64), torch.randn(2, 4, generator=g, dtype=torch.float64)
    assert torch.allclose(R.quaternion_rotate(pc, q), O.quaternion_rotate(pc, q), atol=1e-14)
    assert torch.allclose(R.quaternion_rotate(R.quaternion_rotate(pc, q), q, inverse=True), pc, atol=1e-13)


def test_projection_outputs_are_lazy():
    from dpc.render import ProjectionOutputs

    calls = []

    def build():
        calls.append(1)
        return {"voxels": "V", "tr_pc": "T", "drc_probs": "P", "proj_depth": "D", "proj": "ignored"}

    out = ProjectionOutputs("PROJ", build)
    assert sorted(out.keys()) == sorted(["proj", "voxels", "tr_pc", "voxels_rgb", "proj_rgb", "drc_probs", "proj_depth"])
    assert out["proj"] == "PROJ" and out["proj_rgb"] is None and not calls
    assert out["drc_probs"] == "P" and out["voxels"] == "V" and len(calls) == 1


def test_dead_branches_named_before_any_launch():
    import dpc.render as R
    from oracle.dpc_oracle import Cfg

    z = torch.zeros(1, 4, 3)
    for kw, key in ((dict(pose_quaternion=False), "pose_quaternion"), (dict(ptn_max_projection=True), "ptn_max_projection"),
                    (dict(drc_tf_cumulative=False), "drc_tf_cumulative")):
        with pytest.raises(NotImplementedError, match=key):
            R.pointcloud_project_fast(Cfg(**kw), z, torch.ones(1, 4), None, None)
    with pytest.raises(NotImplementedError, match="all_rgb"):
        R.pointcloud_project_fast(Cfg(), z, torch.ones(1, 4), None, z)


def test_util_shims_expose_reference_names():
    import util.drc
    import util.gauss_kernel
    import util.point_cloud_to as P
    import util.quaternion

    for name in ("pc_point_dropout", "pointcloud_project_fast", "pointcloud2voxels3d_fast", "smoothen_voxels3d",
                 "pc_perspective_transform"):
        assert callable(getattr(P, name))
    for name in ("drc_projection", "drc_event_probabilities", "drc_depth_projection", "drc_depth_grid"):
        assert callable(getattr(util.drc, name))
    assert callable(util.gauss_kernel.smoothing_kernel) and callable(util.quaternion.quaternion_rotate)
    np.testing.assert_allclose(util.drc.drc_depth_grid(__import__("oracle.dpc_oracle", fromlist=["Cfg"]).Cfg(), 4).numpy(),
                               [1.5, 1.75, 2.0, 2.25, 10.0])


def test_prediction_files_are_the_reference_format(tmp_path):
    """`<model>_pc.pkl` (dpc/run/predict_to.py:331-336 writes it, dpc/run/eval_chamfer_to.py:95-107 reads it): a file
    written the reference's way loads here, a file written here is what the reference's reader expects."""
    import pickle

    import dpc.render as R

    rs = np.random.RandomState(0)
    pts, cam = rs.rand(5, 40, 3).astype(np.float32), rs.rand(5, 4).astype(np.float32)
    ref_file = tmp_path / "ref_pc.pkl"
    with open(ref_file, "wb") as handle:   # the writer of predict_to.py, verbatim in effect
        pickle.dump({"points": pts, "camera_pose": cam}, handle, protocol=pickle.HIGHEST_PROTOCOL)
    p, c, n = R.load_predictions(str(ref_file))
    assert np.array_equal(p, pts) and np.array_equal(c, cam) and n is None
    ours = tmp_path / "ours_pc.pkl"
    R.save_predictions(str(ours), torch.from_numpy(pts), torch.from_numpy(cam), num_points=np.array([40, 30, 40, 10, 40]))
    with open(ours, "rb") as handle:       # the reader of eval_chamfer_to.py
        data = pickle.load(handle)
    assert set(data) == {"points", "camera_pose", "num_points"} and isinstance(data["points"], np.ndarray)
    assert np.array_equal(np.squeeze(data["points"]), pts) and np.array_equal(np.squeeze(data["num_points"]), [40, 30, 40, 10, 40])


def test_prefer_direct_graph_launch_sets_the_runtime_flag(monkeypatch):
    """dpc.render.prefer_direct_graph_launch(): sets DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 while HIP is not initialised (never in
    this CPU test process), leaves an explicit value alone, and reports what is in force."""
    import dpc.render as R

    monkeypatch.delenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE", raising=False)
    assert R.prefer_direct_graph_launch() is True and os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] == "0"
    monkeypatch.setenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "1")
    assert R.prefer_direct_graph_launch() is False and os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] == "1"


def test_camera_gradient_hand_off_carries_sc1_in_the_isa():
    """camgrad_publish (csrc/dpc_kernels.h) hands 13 doubles per slab from workgroup to workgroup inside one launch with
    relaxed agent-scope atomics -- the form MI355X_MICROARCH.md lists as measured valid on gfx950 ONLY when every handed-off
    byte is stored and loaded with sc1 (write-through / L2-served) and the storing wave drains its stores before the ticket
    add.  That is a property of the generated code, so it is checked on the generated code: in the backward slab kernel of
    the benchmark configuration the 8-byte hand-off stores and loads carry sc1, an s_waitcnt vmcnt(0) sits between the
    stores and the returning ticket add."""
    import re
    import subprocess

    obj = os.path.join(ROOT, "pytorch-unsup-pc_amd", "csrc", "dpc_slab_bwd.o")
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not (os.path.exists(obj) and os.path.exists(objdump)):
        pytest.skip("needs the built object file and llvm-objdump")
    import tempfile

    with tempfile.TemporaryDirectory() as tmp:
        # the device code object sits in the host object's .hip_fatbin section: dump it, unbundle, disassemble
        subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objcopy", "--dump-section", ".hip_fatbin=" + tmp + "/fb.bin", obj],
                       check=True, capture_output=True)
        subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--type=o", "--unbundle",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + tmp + "/fb.bin", "--output=" + tmp + "/dev.o"],
                       check=True, capture_output=True)
        asm = subprocess.run([objdump, "-d", tmp + "/dev.o"], check=True, capture_output=True, text=True).stdout
    m = re.search(r"<_ZN4dpck\S*k_gather_hwILi64ELi8ELi3E\S*>:\n(.*?)\n\n", asm, re.S)
    assert m, "k_gather_hw<64,8,3> not found in the disassembly"
    body = m.group(1)
    stores = [l for l in body.splitlines() if "global_store_dwordx2" in l and "sc1" in l]
    loads = [l for l in body.splitlines() if "global_load_dwordx2" in l and "sc1" in l]
    assert stores and len(loads) >= 8, (len(stores), len(loads))
    lines = body.splitlines()
    first_store = next(i for i, l in enumerate(lines) if "global_store_dwordx2" in l and "sc1" in l)
    ticket = next(i for i, l in enumerate(lines) if i > first_store and "global_atomic_add" in l and "sc0" in l)   # returning add
    assert any("s_waitcnt vmcnt(0)" in l for l in lines[first_store:ticket]), "no vmcnt(0) between the hand-off stores and the ticket"


def _device_asm(obj_name):
    import subprocess
    import tempfile

    obj = os.path.join(ROOT, "pytorch-unsup-pc_amd", "csrc", obj_name)
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not (os.path.exists(obj) and os.path.exists(objdump)):
        pytest.skip("needs the built object file and llvm-objdump")
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objcopy", "--dump-section", ".hip_fatbin=" + tmp + "/fb.bin", obj],
                       check=True, capture_output=True)
        subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--type=o", "--unbundle",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + tmp + "/fb.bin", "--output=" + tmp + "/dev.o"],
                       check=True, capture_output=True)
        return subprocess.run([objdump, "-d", tmp + "/dev.o"], check=True, capture_output=True, text=True).stdout


def _kernel_body(asm, mangled_fragment):
    import re

    m = re.search(r"<_ZN4dpck\S*" + mangled_fragment + r"\S*>:\n(.*?)\n\n", asm, re.S)
    assert m, mangled_fragment + " not found in the disassembly"
    return m.group(1).splitlines()


def test_grid_sized_store_streams_are_written_through_in_the_isa():
    """T (forward slab kernels) and dT (column kernels) leave with the sc1 bit -- store_through / kAuxThrough in
    csrc/dpc_kernels.h: without it the L2s hold a whole grid dirty until the kernel ends (4 us of the c2 step, DESIGN.md
    section 4).  The bit is a property of the generated code: every grid store of the benchmark configuration's kernels
    carries it, and the small scattered stores that measured slower written through do not."""
    xl = _kernel_body(_device_asm("dpc_slab_xl.o"), "k_splat_xlILi4ELi3E")
    rows = [l for l in xl if "global_store_dword " in l]
    plain = [l for l in rows if "sc1" not in l]   # the per-cloud words the kernel zeroes for the ray-march kernel
    assert len(rows) - len(plain) >= 16 and len(plain) <= 4, (len(rows), plain[:5])
    col = _kernel_body(_device_asm("dpc_column.o"), "k_zcol_fwdbwdILi64ELi3ELi1E")
    dts = [l for l in col if "buffer_store_dword " in l]
    assert len(dts) >= 64 and all("sc1" in l for l in dts), (len(dts), [l for l in dts if "sc1" not in l][:3])
    hw = _kernel_body(_device_asm("dpc_slab_fwd.o"), "k_splat_hwILi128ELi1ELi8E")
    assert any("global_store_dwordx2" in l and "sc1" in l for l in hw), "k_splat_hw<128,1,8> stores T without sc1"
    loc = _kernel_body(_device_asm("dpc_slab_fwd.o"), "k_locateILi0E")
    assert not any("global_store" in l and "sc1" in l for l in loc), "k_locate's record stores are meant to be ordinary"


def test_kernel_lists_carry_their_taps_and_geometry():
    """smoothing_kernel returns a list (what the reference's caller passes straight on) that also carries its host taps and the
    Geometry objects built from it; the result is memoised per (size, sigma); a list whose elements a caller replaced is
    re-read instead of trusted."""
    import dpc.render as R

    class Cfg(dict):
        __getattr__ = dict.__getitem__

    cfg = Cfg(vox_size=32, vox_size_z=-1, pc_gauss_kernel_size=11)
    k = R.smoothing_kernel(cfg, 1.5)
    assert isinstance(k, list) and len(k) == 3 and [tuple(t.shape) for t in k] == [(1, 1, 1, 1, 11), (1, 1, 1, 11, 1), (1, 1, 11, 1, 1)]
    assert R.smoothing_kernel(cfg, 1.5) is k and R.smoothing_kernel(cfg, 1.25) is not k
    assert np.array_equal(k.taps[0], k[0].reshape(-1).numpy()) and np.array_equal(k.taps[1], k[2].reshape(-1).numpy())
    g1, g2 = R._geometry(cfg, k), R._geometry(cfg, k)
    assert g1 is g2 and np.array_equal(g1.kxy, k.taps[0]) and (g1.D, g1.H, g1.W) == (32, 32, 32)
    cfg64 = Cfg(cfg, vox_size=64)
    assert R._geometry(cfg64, k) is not g1 and R._geometry(cfg64, k).W == 64
    # a caller that swaps an element: the list is read again, not trusted
    k2 = R.smoothing_kernel(cfg, 0.75)
    mine = type(k2)(list(k2))
    mine[2] = k[2]
    g3 = R._geometry(cfg, mine)
    assert np.array_equal(g3.kz, k.taps[1]) and np.array_equal(g3.kxy, k2.taps[0]) and not mine.untouched()
    # plain lists of tensors and bare 1-D kernels keep working
    g4 = R._geometry(cfg, [t.clone() for t in k])
    assert np.array_equal(g4.kxy, g1.kxy) and np.array_equal(g4.kz, g1.kz)
    # DpcParams blocks and buffer sizes are cached per call shape (needs the library, not a GPU)
    z1, z2 = g1.sized(4, 1000), g1.sized(4, 1000)
    assert z1 is z2 and z1.cells_bytes > 0 and z1.ws_bytes >= 4 * 32 ** 3 * 4 and g1.sized(5, 1000) is not z1
