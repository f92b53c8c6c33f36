"""Drop-in for the reference's dpc/util/point_cloud_distance.py."""
from dpc.render import point_cloud_distance  # noqa: F401

from ._overlay import fall_through as _fall_through  # noqa: E402

__getattr__ = _fall_through(__name__, __file__)   # everything else: the module of the same name that this one overlays
