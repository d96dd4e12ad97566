from .mpc_controller import MPCController  # noqa: F401
from .rnn_mpc_controller import RNNMPCController  # noqa: F401
