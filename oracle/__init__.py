"""CPU oracle for the MPC random-shooting / CEM hot path.  TEST INFRASTRUCTURE ONLY.

This package is a NumPy restatement of the reference algorithm
(``learning_to_adapt/policies/mpc_controller.py:59-129`` and the model /
reward code it calls).  It exists so that the HIP path can be checked against
something that follows the reference line by line.

Rules (enforced by ``tests/test_layout.py``):

* only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
  ``cpu_baseline`` leg may import anything from here;
* nothing under ``learning_to_adapt_amd/`` imports it - the product path has
  no CPU fallback and fails loudly when the HIP extension is missing.

Parity status
-------------
* planner (``oracle/planner.py``): PINNED.  Checked in this container against
  the *real* reference ``MPCController`` (imported from ``/root/reference``
  with a stub ``tensorflow`` module, see ``tools/gen_golden.py``) and against
  the committed golden vectors under ``tests/golden/``.
* rewards (``oracle/rewards.py``): restated from the reference env files; the
  reference envs need the proprietary MuJoCo 1.31 binary at import time and
  cannot be executed here.  The formulas are three lines each and are cited.
* MLP arithmetic (``oracle/dynamics.py``): **parity unpinned at the
  TensorFlow boundary**.  The arithmetic lives in third-party
  ``tensorflow==1.13.1`` (``docker/environment.yml:57``), which is not under
  ``/root/reference`` and not installed; the reference has no test pinning
  it.  The restatement follows the call sites
  (``dynamics/core/utils.py:111-142``, ``:264-296``) - dense = ``x @ W + b``,
  kernels ``[in, out]`` - and is cross-checked against float64 NumPy and
  torch CPU fp32 in ``tests/test_oracle.py``.
* inner adaptation (``oracle/adapt.py``): restates ``meta_mlp_dynamics.py:321-345,96-120,409-421`` (zero-row
  padding, pre/post split, mean-square loss, one SGD step).  ``tf.gradients`` is TensorFlow's: **unpinned at the
  TensorFlow boundary**; the hand-written backward pass is pinned against float64 central finite differences
  (``tools/gen_adapt_golden.py``, ``tests/test_adapt_oracle.py``) and by the fixture ``tests/golden/adapt_cases.npz``.
* training loop (``oracle/fit.py``): restates ``mlp_dynamics.py:140-197`` (epoch / batch structure, loss, TensorFlow's
  documented Adam update, rolling-average early stop) with the batch order as an input (the ``tf.data`` shuffle cannot
  be reproduced): **unpinned at the TensorFlow boundary**; ``tests/test_fit_oracle.py`` drives the drop-in's ``fit``
  with the same split and batch orders.
* recurrent planner (``oracle/rnn_planner.py``): PINNED against the real
  ``RNNMPCController`` (``policies/rnn_mpc_controller.py``) the same way.
* LSTM cell arithmetic (``oracle/rnn_dynamics.py``): unpinned at the
  TensorFlow boundary (``tf.nn.rnn_cell.LSTMCell`` of tensorflow==1.13.1),
  restated from its published algorithm; cross-checked against
  ``torch.nn.LSTMCell`` in ``tests/test_oracle.py``.
"""

from .dynamics import OracleMLPDynamics, mlp_forward_f32  # noqa: F401
from .rewards import make_reward  # noqa: F401
from .planner import rs_plan, cem_plan  # noqa: F401
from .rnn_dynamics import OracleLSTMDynamics, LSTMStateTuple, lstm_step_f32  # noqa: F401
from .rnn_planner import rnn_rs_plan, rnn_cem_plan, repeat_hidden  # noqa: F401
from .rnn_cells import OracleRNNStackDynamics, gru_step_f32, basic_rnn_step_f32  # noqa: F401
