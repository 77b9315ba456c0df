"""cvml_goalnet_amd — MI355X-native implementation of CVML-GoalNet's one hot path: the `AVM`
frame-importance model (forward, broadcast-MSE, backward, Adam) behind the reference's own Python
surface. See DESIGN.md / INTEGRATION.md.

    from cvml_goalnet_amd import AVM          # drop-in for /root/reference/utils.py:229 `AVM`
"""
from . import synth  # noqa: F401
from ._lib import GoalnetError, LIB_PATH  # noqa: F401
from .avm import AVM  # noqa: F401
from . import optim  # noqa: F401

__all__ = ["AVM", "GoalnetError", "synth", "LIB_PATH", "optim"]
