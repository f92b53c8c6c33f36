"""Overlay of the reference's `util` package.

The reference's caller imports its hot path as `from util.point_cloud_to import ...`, `from util.gauss_kernel import ...`,
`from util.quaternion import ...` (dpc/models/model_pc_to.py:15-24) because dpc/run/startup.py:4-6 puts the reference's
`dpc/` directory -- which holds its own regular package `util` (an empty __init__.py and 24 modules) -- on sys.path.
Put THIS package's parent directory on sys.path BEFORE that one and the same imports resolve here, while every module this
build does not replace (`util.app_config`, `util.camera`, `util.fs`, ...) keeps resolving to the reference: the package's
search path is extended with every other `util` directory found on sys.path (pkgutil.extend_path), this directory first.
The five modules that ARE replaced (point_cloud_to, drc, gauss_kernel, quaternion, point_cloud_distance) export the
MI355X implementation of the hot-path functions and fall through to the reference's module of the same name for every
other attribute (`util.quaternion.as_rotation_matrix`, `util.gauss_kernel.gauss_smoothen_image`, ...): _overlay.py.
"""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
