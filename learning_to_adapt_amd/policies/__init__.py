from .mpc_controller import MPCController  # noqa: F401
