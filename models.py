"""Drop-in for the reference's top-level `models` module.

The reference's drivers import the two model classes by this module name
(test.py:16, train.py:16, dist_train.py:15: `from models import GrainNN_regressor,
GrainNN_classifier`).  With this repository ahead of the reference checkout on `sys.path`
that line resolves here, and the classes are the MI355X HIP implementations of
`graingraphnn_amd.models` -- same constructors (test.py:177,182), same `forward` /
`update` signatures (test.py:382-383, 400, 426), same 284-key `state_dict` layout
(test.py:178,183 load the reference's `.pt` files), same `threshold` attributes
(test.py:187-188 set them after construction).  test.py itself is not edited.
"""
from graingraphnn_amd.models import GrainNN_classifier, GrainNN_regressor

__all__ = ["GrainNN_regressor", "GrainNN_classifier"]
