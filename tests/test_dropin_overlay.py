"""The Python drop-in next to a second `util` package (reference imports: dpc/models/model_pc_to.py:15-24,
path set-up dpc/run/startup.py:4-6).  Runs in a child interpreter so that the `util` of this process is untouched."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "pytorch-unsup-pc_amd")


def _run(code, *paths):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    proc = subprocess.run([sys.executable, "-c", textwrap.dedent(code)] + list(paths), capture_output=True, text=True, env=env)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    return proc.stdout


def test_overlay_next_to_a_second_util_package(tmp_path):
    """A stand-in for the reference's dpc/ directory: a regular package `util` with modules this build replaces
    (point_cloud_to, quaternion) and modules it does not (app_config, camera)."""
    ref = tmp_path / "dpc"
    (ref / "util").mkdir(parents=True)
    (ref / "util" / "__init__.py").write_text("")
    (ref / "util" / "app_config.py").write_text("config = 'the reference config module'\n")
    (ref / "util" / "camera.py").write_text("from util.quaternion import quaternion_multiply, as_rotation_matrix\nWHO = 'ref camera'\n")
    (ref / "util" / "point_cloud_to.py").write_text("WHO = 'ref'\ndef pointcloud_project_fast(*a, **k):\n    raise AssertionError('shadowed')\n"
                                                    "def select_3d(data, indices):\n    return 'ref select_3d'\n")
    (ref / "util" / "quaternion.py").write_text("def quaternion_multiply(a, b):\n    raise AssertionError('shadowed')\n"
                                                 "def as_rotation_matrix(q):\n    return 'ref as_rotation_matrix'\n")
    (ref / "util" / "gauss_kernel.py").write_text("def gauss_smoothen_image(cfg, img, sigma_rel):\n    return 'ref gauss_smoothen_image'\n")
    out = _run("""
        import sys
        sys.path.insert(0, sys.argv[1])      # the line a maintainer adds to dpc/run/startup.py
        sys.path.append(sys.argv[2])         # startup.py's own line
        from util.app_config import config
        from util.point_cloud_to import pc_point_dropout, pointcloud_project_fast
        from util.gauss_kernel import smoothing_kernel
        from util.quaternion import quaternion_multiply as q_mul, quaternion_normalise as q_norm, \\
            quaternion_rotate as q_rotate, quaternion_conjugate as q_conj
        import util.camera, util.point_cloud_to, util.gauss_kernel, util.drc, util.point_cloud_distance
        import dpc.render as R
        assert config == 'the reference config module'
        assert pointcloud_project_fast is R.pointcloud_project_fast and pc_point_dropout is R.pc_point_dropout
        assert smoothing_kernel is R.smoothing_kernel and q_mul is R.quaternion_multiply and q_rotate is R.quaternion_rotate
        assert q_norm is R.quaternion_normalise and q_conj is R.quaternion_conjugate
        assert util.drc.drc_projection is R.drc_projection and util.point_cloud_distance.point_cloud_distance is R.point_cloud_distance
        # modules this build does not replace come from the second package, and THEIR imports resolve through the overlay
        assert util.camera.WHO == 'ref camera' and util.camera.quaternion_multiply is R.quaternion_multiply
        # names the replaced modules do not define fall through to the module they overlay
        assert util.camera.as_rotation_matrix(None) == 'ref as_rotation_matrix'
        assert util.point_cloud_to.select_3d(0, 0) == 'ref select_3d'
        assert util.gauss_kernel.gauss_smoothen_image(0, 0, 0) == 'ref gauss_smoothen_image'
        try:
            util.point_cloud_to.no_such_name
        except AttributeError as e:
            assert 'no_such_name' in str(e)
        else:
            raise AssertionError('missing attribute did not raise')
        print('ok')
        """, PKG, str(ref))
    assert out.strip().endswith("ok")


@pytest.mark.skipif(not os.path.isdir("/root/reference/dpc/util"), reason="the reference checkout exists in the build container only")
def test_overlay_next_to_the_reference_itself():
    """With the reference's real dpc/ on the path: its caller's import lines (model_pc_to.py:15-24) bind to this build,
    its other util modules are found where they were."""
    out = _run("""
        import sys, importlib.util
        sys.path.insert(0, sys.argv[1])
        sys.path.append('/root/reference/dpc')
        from util.point_cloud_to import pc_point_dropout, pointcloud_project_fast
        from util.gauss_kernel import smoothing_kernel
        from util.quaternion import quaternion_multiply as q_mul, quaternion_normalise as q_norm, \\
            quaternion_rotate as q_rotate, quaternion_conjugate as q_conj
        import dpc.render as R
        assert pointcloud_project_fast is R.pointcloud_project_fast and smoothing_kernel is R.smoothing_kernel
        for name in ('app_config', 'camera', 'fs', 'system', 'voxel', 'losses_to'):
            spec = importlib.util.find_spec('util.' + name)
            assert spec is not None and spec.origin.startswith('/root/reference/dpc/util/'), (name, spec)
        for name in ('point_cloud_to', 'drc', 'gauss_kernel', 'quaternion', 'point_cloud_distance'):
            assert importlib.util.find_spec('util.' + name).origin.startswith(sys.argv[1]), name
        import util.fs                      # a reference module with no third-party dependency: really importable
        assert util.fs.__file__.startswith('/root/reference/dpc/util/')
        print('ok')
        """, PKG)
    assert out.strip().endswith("ok")


@pytest.mark.skipif(not os.path.isdir("/root/reference/dpc/models"), reason="the reference checkout exists in the build container only")
def test_the_references_own_caller_through_the_overlay():
    """north_star: "so dpc/run/train_eval_to.py can import it as a drop-in".  The reference's caller module
    (dpc/models/model_pc_to.py, import block :15-24) is imported with the overlay in front of the reference's dpc/: the
    names it binds ARE this build's functions, ModelPointCloud(cfg) constructs from the experiment's yaml, and its own
    forward (compute_projection, :239-282) reaches dpc.render.pointcloud_project_fast with arguments that bind -- on this
    GPU-less host that call ends in dpc.render's loud "MI355X only" error, raised after the config, the kernel list and the
    tensors were accepted (there is no CPU path to fall into)."""
    out = _run("""
        import sys
        sys.path.insert(0, sys.argv[1])            # the line a maintainer adds to dpc/run/startup.py
        sys.path.append('/root/reference/dpc')     # startup.py's own line (dpc/run/startup.py:4-6)
        import torch, yaml
        from models import model_pc_to
        import dpc.render as R
        assert model_pc_to.pointcloud_project_fast is R.pointcloud_project_fast
        assert model_pc_to.smoothing_kernel is R.smoothing_kernel and model_pc_to.pc_point_dropout is R.pc_point_dropout
        assert model_pc_to.q_mul is R.quaternion_multiply and model_pc_to.q_norm is R.quaternion_normalise
        assert model_pc_to.q_rotate is R.quaternion_rotate and model_pc_to.q_conj is R.quaternion_conjugate
        assert model_pc_to.__file__.startswith('/root/reference/dpc/models/')

        class Cfg(dict):
            __getattr__ = dict.__getitem__
            __setattr__ = dict.__setitem__
        cfg = Cfg(yaml.safe_load(open('/root/reference/dpc/resources/default_config.yaml')))
        cfg.update(yaml.safe_load(open('/root/reference/experiments/chair_unsupervised/config.yaml')))
        full = model_pc_to.ModelPointCloud(cfg)
        assert sum(p.numel() for p in full.parameters()) == 61042134      # SURVEY 8(c)
        del full
        cfg.update(z_dim=64, fc_dim=64, f_dim=8, pc_num_points=256, vox_size=16, pc_gauss_kernel_size=11, batch_size=2,
                   step_size=2, input_shape=[32, 32, 3], pc_point_dropout=0.5)
        model = model_pc_to.ModelPointCloud(cfg)
        g = torch.Generator().manual_seed(3)
        images = torch.rand(4, 3, 32, 32, generator=g)
        inputs = dict(images=images, masks=(torch.rand(4, 1, 32, 32, generator=g) > 0.5).float(), images_1=images[::2])
        seen = {}
        real = R._native.require_device
        def spy(*tensors):
            seen['tensors'] = [None if t is None else tuple(t.shape) for t in tensors]
            return real(*tensors)
        R._ops.N.require_device = spy
        try:
            model(inputs, 0, is_training=True, run_projection=True)
        except RuntimeError as e:
            assert 'MI355X' in str(e), e
        else:
            raise AssertionError('a CPU call went through: there must be no CPU path')
        # what arrived at the renderer: 2 objects x 2 views x 4 candidates = 16 clouds of 128 kept points, poses, scales
        assert seen['tensors'][:2] == [(16, 128, 3), (16, 4)] and seen['tensors'][4] == (16, 1), seen
        print('ok')
        """, PKG)
    assert out.strip().endswith("ok")
