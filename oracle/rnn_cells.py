"""NumPy restatement of the OTHER recurrent models ``create_rnn`` can build.  TEST INFRASTRUCTURE ONLY.

``learning_to_adapt/dynamics/core/utils.py:192-236``: ``cell_type`` selects ``tf.nn.rnn_cell.LSTMCell`` /
``GRUCell`` / ``RNNCell`` (``:199-213``), more than one entry in ``hidden_sizes`` wraps the cells in
``tf.nn.rnn_cell.MultiRNNCell`` (``:217-220``: layer l feeds its new h to layer l + 1, the state is the tuple of the
layers' states), ``tf.nn.dynamic_rnn`` runs them and ``tf.layers.dense(outputs, output_dim, name='output')`` maps
the top layer's h to the delta (``:231-238``).

Cell arithmetic - third-party ``tensorflow==1.13.1`` (``tensorflow/python/ops/rnn_cell_impl.py``), restated from its
published algorithm, **unpinned at the TensorFlow boundary** like the LSTM cell (``oracle/rnn_dynamics.py``):

    GRUCell.call       value = sigmoid([x | h] @ gates/kernel + gates/bias);  r, u = split(value, 2, axis=1)
                       c = act([x | r * h] @ candidate/kernel + candidate/bias);  new_h = u * h + (1 - u) * c
    BasicRNNCell.call  new_h = act([x | h] @ kernel + bias)

``cell_type='rnn'`` names the ABSTRACT ``tf.nn.rnn_cell.RNNCell`` in the reference (``:209``), which TensorFlow
cannot instantiate with ``(hidden_size, activation=...)``; ``BasicRNNCell`` is the evident intent and what is
restated here - build-defined, not a measured reference behaviour.

Hidden-state structures (``rnn_dynamics.py:273-293``): one LSTM layer -> ``LSTMStateTuple(c, h)``; one GRU / RNN
layer -> ``ndarray [batch, units]``; a stack -> a list (``get_initial_hidden``) / tuple (``predict``) of those.
The planner around these models is the reference's own (``tools/gen_golden.py`` drives the real
``RNNMPCController``; note that its ``reset`` (``:139-163``) indexes a single ndarray state row-wise and therefore
only works for LSTM layers and for stacks - the golden cases stay inside that).
"""

import numpy as np

from .dynamics import _act, normalize, denormalize
from .rnn_dynamics import LSTMStateTuple, lstm_step_f32, _sigmoid


def gru_step_f32(x, h_prev, gate_kernel, gate_bias, cand_kernel, cand_bias, activation="tanh", dtype=np.float32):
    act = _act(activation)
    x = np.asarray(x, dtype=dtype)
    h_prev = np.asarray(h_prev, dtype=dtype)
    value = _sigmoid(np.concatenate([x, h_prev], axis=1) @ np.asarray(gate_kernel, dtype=dtype)
                     + np.asarray(gate_bias, dtype=dtype))
    r, u = np.split(value, 2, axis=1)
    c = act(np.concatenate([x, r * h_prev], axis=1) @ np.asarray(cand_kernel, dtype=dtype)
            + np.asarray(cand_bias, dtype=dtype))
    return (u * h_prev + (dtype(1) - u) * c).astype(dtype)


def basic_rnn_step_f32(x, h_prev, kernel, bias, activation="tanh", dtype=np.float32):
    act = _act(activation)
    z = np.concatenate([np.asarray(x, dtype=dtype), np.asarray(h_prev, dtype=dtype)], axis=1) @ np.asarray(kernel, dtype=dtype)
    return act(z + np.asarray(bias, dtype=dtype)).astype(dtype)


PARAMS_PER_LAYER = {"lstm": 2, "gru": 4, "rnn": 2}


class OracleRNNStackDynamics(object):
    """Duck-typed recurrent ``dynamics_model`` (reference ``RNNDynamicsModel`` surface the planner touches).

    ``params``: per layer the cell's variables in ``get_params()`` order (lstm: kernel, bias; gru: gates/kernel,
    gates/bias, candidate/kernel, candidate/bias; rnn: kernel, bias), then ``output/kernel``, ``output/bias``."""

    recurrent = True

    def __init__(self, obs_dim, act_dim, hidden_sizes, cell_type, params, norm, hidden_nonlinearity="tanh",
                 output_nonlinearity=None, dtype=np.float32):
        assert cell_type in PARAMS_PER_LAYER
        self.obs_space_dims, self.action_space_dims = obs_dim, act_dim
        self.hidden_sizes = tuple(int(u) for u in hidden_sizes)
        self.cell_type = cell_type
        self.params = [np.asarray(p) for p in params]
        assert len(self.params) == PARAMS_PER_LAYER[cell_type] * len(self.hidden_sizes) + 2
        self.normalization = norm
        self.hidden_nonlinearity, self.output_nonlinearity = hidden_nonlinearity, output_nonlinearity
        self.dtype = dtype

    # ---- state structures (rnn_dynamics.py:273-293) ----------------------------------------------------------
    def _zero_layer(self, batch, units):
        z = np.zeros((batch, units), dtype=np.float32)
        return LSTMStateTuple(z.copy(), z.copy()) if self.cell_type == "lstm" else z

    def get_initial_hidden(self, batch_size):
        layers = [self._zero_layer(batch_size, u) for u in self.hidden_sizes]
        return layers if len(layers) > 1 else layers[0]

    def _layers_of(self, hidden):
        return list(hidden) if len(self.hidden_sizes) > 1 else [hidden]

    def predict(self, obs, act, hidden_state):
        assert obs.shape[0] == act.shape[0]
        assert obs.ndim == 2 and obs.shape[1] == self.obs_space_dims
        assert act.ndim == 2 and act.shape[1] == self.action_space_dims
        obs = np.asarray(obs, dtype=np.float64)
        act = np.asarray(act, dtype=np.float64)
        nm = self.normalization
        x = np.concatenate([normalize(obs, nm["obs"][0], nm["obs"][1]),
                            normalize(act, nm["act"][0], nm["act"][1])], axis=1)
        new_states, pi = [], 0
        for units, st in zip(self.hidden_sizes, self._layers_of(hidden_state)):        # MultiRNNCell.call
            if self.cell_type == "lstm":
                c, h = lstm_step_f32(x, st[0], st[1], self.params[pi], self.params[pi + 1],
                                     self.hidden_nonlinearity, self.dtype)
                new_states.append(LSTMStateTuple(c, h))
            elif self.cell_type == "gru":
                h = gru_step_f32(x, st, self.params[pi], self.params[pi + 1], self.params[pi + 2], self.params[pi + 3],
                                 self.hidden_nonlinearity, self.dtype)
                new_states.append(h)
            else:
                h = basic_rnn_step_f32(x, st, self.params[pi], self.params[pi + 1], self.hidden_nonlinearity, self.dtype)
                new_states.append(h)
            pi += PARAMS_PER_LAYER[self.cell_type]
            x = h
        d = x @ np.asarray(self.params[pi], dtype=self.dtype) + np.asarray(self.params[pi + 1], dtype=self.dtype)
        d = _act(self.output_nonlinearity)(d)
        delta = denormalize(d, nm["delta"][0], nm["delta"][1])
        nxt = tuple(new_states) if len(new_states) > 1 else new_states[0]
        return obs + delta, nxt
