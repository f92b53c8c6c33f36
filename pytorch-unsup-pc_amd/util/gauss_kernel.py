"""Drop-in for the reference's dpc/util/gauss_kernel.py (imported at dpc/models/model_pc_to.py:17)."""
from dpc.render import gauss_kernel_1d, separable_kernels, smoothing_kernel  # noqa: F401

from ._overlay import fall_through as _fall_through  # noqa: E402

__getattr__ = _fall_through(__name__, __file__)   # everything else: the module of the same name that this one overlays
