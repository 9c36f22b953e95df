"""graingraphnn_amd -- MI355X-native GrainGNN rollout hot path (HIP kernels behind the
reference's `models.py` API).  See DESIGN.md."""
from .models import GrainNN_classifier, GrainNN_regressor  # noqa: F401
from .rollout import GrainRollout  # noqa: F401

__all__ = ["GrainNN_regressor", "GrainNN_classifier", "GrainRollout"]
