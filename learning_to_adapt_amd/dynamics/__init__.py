from .mlp_dynamics import MLPDynamicsModel  # noqa: F401
from .meta_mlp_dynamics import MetaMLPDynamicsModel  # noqa: F401
