"""Drop-in for the reference's dpc/util/gauss_kernel.py (imported at dpc/models/model_pc_to.py:17)."""
from dpc.render import gauss_kernel_1d, separable_kernels, smoothing_kernel  # noqa: F401
