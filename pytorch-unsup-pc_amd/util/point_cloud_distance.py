"""Drop-in for the reference's dpc/util/point_cloud_distance.py."""
from dpc.render import point_cloud_distance  # noqa: F401
