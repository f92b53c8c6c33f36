"""Drop-in for the reference's dpc/util/drc.py."""
from dpc.render import drc_depth_grid, drc_depth_projection, drc_event_probabilities, drc_projection  # noqa: F401

from ._overlay import fall_through as _fall_through  # noqa: E402

__getattr__ = _fall_through(__name__, __file__)   # everything else: the module of the same name that this one overlays
