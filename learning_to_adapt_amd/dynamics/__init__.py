from .mlp_dynamics import MLPDynamicsModel  # noqa: F401
from .meta_mlp_dynamics import MetaMLPDynamicsModel  # noqa: F401
from .rnn_dynamics import RNNDynamicsModel, LSTMStateTuple  # noqa: F401
