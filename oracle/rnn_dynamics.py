"""NumPy restatement of the recurrent (ReBAL) dynamics ``predict`` path.  TEST INFRASTRUCTURE ONLY.

Follows:

* ``learning_to_adapt/dynamics/rnn_dynamics.py:233-252`` (``predict``): add a time axis of
  length 1, normalise obs / act (``:309-318``), run the cell + output layer, denormalise the
  delta (``:247``), ``obs + delta``, return ``(pred_obs, next_hidden_state)``;
* ``rnn_dynamics.py:273-293`` (``get_initial_hidden``): the cell's zero state tiled to the
  batch - for the single-layer LSTM of ``run_scripts/run_rebal.py:98-99`` an
  ``LSTMStateTuple(c, h)`` of two ``[batch, units]`` arrays;
* ``dynamics/core/utils.py:192-236`` (``create_rnn``): ``tf.nn.rnn_cell.LSTMCell(units,
  activation=hidden_nonlinearity)`` + ``tf.nn.dynamic_rnn`` + ``tf.layers.dense(outputs,
  output_dim, name='output')``; placeholders are float32 (``rnn_dynamics.py:56-59``).

The cell arithmetic lives in third-party ``tensorflow==1.13.1`` (``docker/environment.yml:57``,
not under ``/root/reference``).  Restated from its published algorithm
(``tensorflow/python/ops/rnn_cell_impl.py``, class ``LSTMCell.call``, no peepholes, no
projection, ``forget_bias=1.0``):

    z = concat([x, h_prev], 1) @ kernel + bias              kernel [in + units, 4 * units]
    i, j, f, o = split(z, 4, axis=1)
    c = sigmoid(f + forget_bias) * c_prev + sigmoid(i) * act(j)
    h = sigmoid(o) * act(c)

Parity status: the planner around it is pinned (``tools/gen_golden.py`` drives the real
``RNNMPCController`` with this class); the cell arithmetic is **unpinned at the TensorFlow
boundary**, like the MLP (``oracle/__init__``).
"""

from collections import namedtuple

import numpy as np

from .dynamics import _act, normalize, denormalize

# same field order as tf.nn.rnn_cell.LSTMStateTuple
LSTMStateTuple = namedtuple("LSTMStateTuple", ("c", "h"))

FORGET_BIAS = np.float32(1.0)


def _sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x))).astype(x.dtype)


def lstm_step_f32(x, c_prev, h_prev, kernel, bias, activation="tanh", dtype=np.float32):
    """One ``LSTMCell.call``.  Returns ``(c, h)``."""
    act = _act(activation)
    x = np.asarray(x, dtype=dtype)
    c_prev = np.asarray(c_prev, dtype=dtype)
    h_prev = np.asarray(h_prev, dtype=dtype)
    z = np.concatenate([x, h_prev], axis=1) @ np.asarray(kernel, dtype=dtype)
    z = z + np.asarray(bias, dtype=dtype)
    i, j, f, o = np.split(z, 4, axis=1)
    c = _sigmoid(f + dtype(FORGET_BIAS)) * c_prev + _sigmoid(i) * act(j)
    h = _sigmoid(o) * act(c)
    return c.astype(dtype), h.astype(dtype)


class OracleLSTMDynamics(object):
    """Duck-typed recurrent ``dynamics_model`` for the reference / oracle RNN planner.

    ``params`` = ``[kernel [in + units, 4 units], bias [4 units], out_kernel [units, obs_dim],
    out_bias [obs_dim]]`` - the order of the reference's trainable variables
    (``rnn/lstm_cell/kernel``, ``rnn/lstm_cell/bias``, ``output/kernel``, ``output/bias``).
    """

    recurrent = True

    def __init__(self, obs_dim, act_dim, params, norm, hidden_nonlinearity="tanh",
                 output_nonlinearity=None, dtype=np.float32):
        self.obs_space_dims = obs_dim
        self.action_space_dims = act_dim
        self.params = list(params)
        self.units = int(self.params[1].shape[0]) // 4
        assert self.params[0].shape == (obs_dim + act_dim + self.units, 4 * self.units)
        assert self.params[2].shape == (self.units, obs_dim)
        self.normalization = norm
        self.hidden_nonlinearity = hidden_nonlinearity
        self.output_nonlinearity = output_nonlinearity
        self.dtype = dtype

    def get_initial_hidden(self, batch_size):
        z = np.zeros((batch_size, self.units), dtype=np.float32)        # rnn_dynamics.py:286-289
        return LSTMStateTuple(z.copy(), z.copy())

    def predict(self, obs, act, hidden_state):
        assert obs.shape[0] == act.shape[0]                             # :234-236
        assert obs.ndim == 2 and obs.shape[1] == self.obs_space_dims
        assert act.ndim == 2 and act.shape[1] == self.action_space_dims
        obs = np.asarray(obs, dtype=np.float64)
        act = np.asarray(act, dtype=np.float64)
        c_prev, h_prev = hidden_state                                   # LSTMStateTuple or [c, h]
        nm = self.normalization
        o = normalize(obs, nm["obs"][0], nm["obs"][1])
        a = normalize(act, nm["act"][0], nm["act"][1])
        x = np.concatenate([o, a], axis=1)                              # :62, seq_len 1
        c, h = lstm_step_f32(x, c_prev, h_prev, self.params[0], self.params[1],
                             self.hidden_nonlinearity, self.dtype)
        d = h @ np.asarray(self.params[2], dtype=self.dtype) + np.asarray(self.params[3], dtype=self.dtype)
        d = _act(self.output_nonlinearity)(d)
        delta = denormalize(d, nm["delta"][0], nm["delta"][1])          # :247
        return obs + delta, LSTMStateTuple(c, h)                        # :250-252
