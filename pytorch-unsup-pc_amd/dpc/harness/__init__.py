"""Full-step harness: the networks around the renderer and one training step.

Not part of the drop-in renderer: this is what BASELINE config 3 ("chair_unsupervised full train step") is measured
with, and a worked example of dpc.render inside the reference's training graph.  The networks are plain torch.nn
(dense contractions: they stay on PyTorch-ROCm / MIOpen / hipBLASLt); parameter names follow the reference's modules
so that its checkpoints load unchanged.
"""
from .config import chair_unsupervised  # noqa: F401
from .nets import Decoder, Encoder, PoseNet, ScalePredictor, StepNets  # noqa: F401
from .step import TrainStep, device_point_dropout, pooled_masks, student_loss  # noqa: F401
from .views import camera_from_blender, pool_single_view, sample_view_indices, sample_views  # noqa: F401
