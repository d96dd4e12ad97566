"""Recurrent cell stacks of ``create_rnn`` (reference ``dynamics/core/utils.py:192-236``): parameter layout, initial
values, the stock-PyTorch forward pass used by ``fit`` and the conversion between the reference's hidden-state
structures and the flat ``[rows, sum(units)]`` arrays the HIP kernels take.

Cell types: ``'lstm'`` (``tf.nn.rnn_cell.LSTMCell``), ``'gru'`` (``GRUCell``), ``'rnn'`` (``BasicRNNCell`` - the
reference names the abstract ``RNNCell`` there, ``:209``, which TensorFlow cannot instantiate).  More than one entry in
``hidden_sizes`` = ``MultiRNNCell``.  Variable names follow TensorFlow's scopes (``rnn/<cell>/...`` for one layer,
``rnn/multi_rnn_cell/cell_<i>/<cell>/...`` for a stack), in ``get_params()`` order.
"""

from collections import namedtuple

import numpy as np
import torch

from . import core

LSTMStateTuple = namedtuple("LSTMStateTuple", ("c", "h"))      # field order of tf.nn.rnn_cell.LSTMStateTuple
FORGET_BIAS = 1.0                                               # tf.nn.rnn_cell.LSTMCell default
CELL_SCOPE = {"lstm": "lstm_cell", "gru": "gru_cell", "rnn": "basic_rnn_cell"}
CELL_TYPES = tuple(CELL_SCOPE)


def layer_param_shapes(cell_type, k_in, units):
    """[(suffix, shape)] of one layer's variables, in creation order."""
    if cell_type == "lstm":
        return [("kernel", (k_in + units, 4 * units)), ("bias", (4 * units,))]
    if cell_type == "gru":
        return [("gates/kernel", (k_in + units, 2 * units)), ("gates/bias", (2 * units,)),
                ("candidate/kernel", (k_in + units, units)), ("candidate/bias", (units,))]
    return [("kernel", (k_in + units, units)), ("bias", (units,))]


def param_spec(obs_dim, act_dim, hidden_sizes, cell_type):
    """[(name, shape)] of every trainable variable incl. the output layer."""
    assert cell_type in CELL_TYPES
    spec, k_in = [], obs_dim + act_dim
    for i, units in enumerate(hidden_sizes):
        scope = "rnn/%s" % CELL_SCOPE[cell_type] if len(hidden_sizes) == 1 else \
            "rnn/multi_rnn_cell/cell_%d/%s" % (i, CELL_SCOPE[cell_type])
        spec += [("%s/%s" % (scope, suffix), shape) for suffix, shape in layer_param_shapes(cell_type, k_in, units)]
        k_in = units
    return spec + [("output/kernel", (k_in, obs_dim)), ("output/bias", (obs_dim,))]


def init_params(obs_dim, act_dim, hidden_sizes, cell_type, rng):
    """TF defaults: glorot-uniform kernels, zero biases - except the GRU gate bias, which starts at 1.0
    (``GRUCell``'s ``constant_initializer(1.0)``)."""
    out = []
    for name, shape in param_spec(obs_dim, act_dim, hidden_sizes, cell_type):
        if len(shape) == 2:
            lim = np.sqrt(6.0 / (shape[0] + shape[1]))
            out.append(torch.from_numpy(rng.uniform(-lim, lim, size=shape).astype(np.float32)))
        else:
            out.append(torch.ones(shape) if name.endswith("gates/bias") else torch.zeros(shape))
    return out


def zero_state(cell_type, hidden_sizes, batch, device):
    """Flat training-time state: list of per-layer tensors ((c, h) pairs for LSTM)."""
    z = lambda u: torch.zeros((batch, u), dtype=torch.float32, device=device)  # noqa: E731
    return [(z(u), z(u)) if cell_type == "lstm" else z(u) for u in hidden_sizes]


def detach_state(state):
    return [tuple(t.detach() for t in s) if isinstance(s, tuple) else s.detach() for s in state]


def stack_forward(x_seq, state, params, hidden_sizes, cell_type, cell_act, output_act):
    """Stock-op forward over ``x_seq [B, T, in]`` (used by ``fit``).  Returns ``(deltas [B, T, obs], state)``."""
    act, out = core.torch_act(cell_act), core.torch_act(output_act)
    per = {"lstm": 2, "gru": 4, "rnn": 2}[cell_type]
    state = list(state)
    ys = []
    for t in range(x_seq.shape[1]):
        x = x_seq[:, t]
        for li in range(len(hidden_sizes)):
            p = params[per * li:per * (li + 1)]
            if cell_type == "lstm":
                c, h = state[li]
                z = torch.cat([x, h], dim=1) @ p[0] + p[1]
                i, j, f, o = torch.chunk(z, 4, dim=1)
                c = torch.sigmoid(f + FORGET_BIAS) * c + torch.sigmoid(i) * act(j)
                h = torch.sigmoid(o) * act(c)
                state[li] = (c, h)
            elif cell_type == "gru":
                h = state[li]
                r, u = torch.chunk(torch.sigmoid(torch.cat([x, h], dim=1) @ p[0] + p[1]), 2, dim=1)
                cand = act(torch.cat([x, r * h], dim=1) @ p[2] + p[3])
                h = u * h + (1 - u) * cand
                state[li] = h
            else:
                h = act(torch.cat([x, state[li]], dim=1) @ p[0] + p[1])
                state[li] = h
            x = h
        ys.append(out(x @ params[-2] + params[-1]))
    return torch.stack(ys, dim=1), state


# ---- hidden-state structures (reference rnn_dynamics.py:273-293) <-> flat [rows, sum(units)] arrays ------------
def initial_hidden(cell_type, hidden_sizes, batch_size):
    def layer(u):
        z = np.zeros((batch_size, u), dtype=np.float32)
        return LSTMStateTuple(z.copy(), z.copy()) if cell_type == "lstm" else z
    layers = [layer(u) for u in hidden_sizes]
    return layers if len(layers) > 1 else layers[0]


def pack_hidden(cell_type, hidden_sizes, hidden):
    """Reference structure -> ``(c [rows, W], h [rows, W])`` float32 (``c`` all zero unless LSTM)."""
    layers = list(hidden) if len(hidden_sizes) > 1 else [hidden]
    assert len(layers) == len(hidden_sizes)
    hs = [np.asarray(st[1] if cell_type == "lstm" else st, dtype=np.float32) for st in layers]
    h = np.concatenate(hs, axis=1)
    c = np.concatenate([np.asarray(st[0], dtype=np.float32) for st in layers], axis=1) if cell_type == "lstm" \
        else np.zeros_like(h)
    return np.ascontiguousarray(c), np.ascontiguousarray(h)


def unpack_hidden(cell_type, hidden_sizes, c, h, as_tuple=False):
    """Flat arrays -> the reference structure (``as_tuple``: what ``predict`` returns for a stack)."""
    layers, off = [], 0
    for u in hidden_sizes:
        hl = np.array(h[:, off:off + u])
        layers.append(LSTMStateTuple(np.array(c[:, off:off + u]), hl) if cell_type == "lstm" else hl)
        off += u
    if len(layers) == 1:
        return layers[0]
    return tuple(layers) if as_tuple else layers
