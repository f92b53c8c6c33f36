"""Drop-in for the reference's dpc/util/point_cloud_to.py (names as imported at dpc/models/model_pc_to.py:15)."""
from dpc.render import (pc_perspective_transform, pc_point_dropout, pointcloud2voxels3d_fast,  # noqa: F401
                        pointcloud_project, pointcloud_project_fast, smooth_voxels3d, smoothen_voxels3d)

from ._overlay import fall_through as _fall_through  # noqa: E402

__getattr__ = _fall_through(__name__, __file__)   # everything else: the module of the same name that this one overlays
