"""Drop-in for the hot-path part of the reference's dpc/util/quaternion.py (:69-132)."""
from dpc.render import quaternion_conjugate, quaternion_multiply, quaternion_normalise, quaternion_rotate  # noqa: F401

from ._overlay import fall_through as _fall_through  # noqa: E402

__getattr__ = _fall_through(__name__, __file__)   # everything else: the module of the same name that this one overlays
