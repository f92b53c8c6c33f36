"""Drop-in for the reference's dpc/util/drc.py."""
from dpc.render import drc_depth_grid, drc_depth_projection, drc_event_probabilities, drc_projection  # noqa: F401
