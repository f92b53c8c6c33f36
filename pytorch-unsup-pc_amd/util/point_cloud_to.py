"""Drop-in for the reference's dpc/util/point_cloud_to.py (names as imported at dpc/models/model_pc_to.py:15)."""
from dpc.render import (pc_perspective_transform, pc_point_dropout, pointcloud2voxels3d_fast,  # noqa: F401
                        pointcloud_project, pointcloud_project_fast, smooth_voxels3d, smoothen_voxels3d)
