"""Drop-in for the hot-path part of the reference's dpc/util/quaternion.py (:69-132)."""
from dpc.render import quaternion_conjugate, quaternion_multiply, quaternion_normalise, quaternion_rotate  # noqa: F401
