"""learning_to_adapt_amd - MI355X-native MPC planner hot path of iclavera/learning_to_adapt.

Python surface mirrors the reference's for this path only:

* ``learning_to_adapt_amd.policies.MPCController``      (policies/mpc_controller.py)
* ``learning_to_adapt_amd.dynamics.MLPDynamicsModel``    (dynamics/mlp_dynamics.py)
* ``learning_to_adapt_amd.dynamics.MetaMLPDynamicsModel`` (dynamics/meta_mlp_dynamics.py)
* ``learning_to_adapt_amd.policies.RNNMPCController``   (policies/rnn_mpc_controller.py)
* ``learning_to_adapt_amd.dynamics.RNNDynamicsModel``    (dynamics/rnn_dynamics.py)

The compute is ``libl2a_hip.so`` (C ABI in ``include/l2a.h``, HIP sources in ``csrc/``).
Importing this package never touches HIP; the device is initialised on the first plan step.
"""

__version__ = "0.1.0"
