"""NumPy restatement of ``RNNDynamicsModel.fit``'s training loop.  TEST INFRASTRUCTURE ONLY.

Follows ``learning_to_adapt/dynamics/rnn_dynamics.py``:

* ``:118-140`` normalisation of the paths (``compute_normalization`` ``:295-311``: mean / std over paths AND time) and
  the split of WHOLE paths into a training and a validation set (``train_test_split``, ``dynamics/utils.py``);
* ``:156-186`` one epoch = one pass over the batches of ``batch_size`` paths; every batch starts from the zero hidden
  state (``get_initial_hidden``, ``:162``) and is cut into chunks of ``backprop_steps`` time steps (``:167``); each chunk is
  one ``sess.run([self.loss, self._gradients_vars, self.next_hidden_state_var])`` (``:175-176``) with ``loss =
  reduce_mean(square(delta_pred - delta))`` over the chunk (``:88``) - the gradient stops at the chunk's first hidden
  state, which is FED as a value (truncated back-propagation through time), the hidden state after the chunk is carried
  into the next one; the chunk gradients are averaged (``np.mean``, ``:181``) and applied by ONE
  ``optimizer.apply_gradients`` step (``:91-92``, ``:183``; ``tf.train.AdamOptimizer``);
* ``:185-201`` after the last batch the validation loss: the whole held-out set from the zero state in one pass, the
  rolling average (``1.5 x`` / ``2 x`` start values) and the stop rule ``prev < avg or epoch == epochs - 1`` (``:214``).

The batch order is TensorFlow's (``tf.data`` ``batch`` then ``shuffle``, ``:255-268``: the paths are batched in data-set
order and the BATCHES are shuffled) and cannot be reproduced; it is an input here (``orders[epoch]`` = the batch start
rows in visiting order) - the product is driven with the same orders in ``tests/test_fit_oracle.py``.

The gradient is a hand-written backward pass through the cells of ``oracle/rnn_cells.py`` /
``oracle/rnn_dynamics.py`` (``LSTMCell`` with forget bias 1, ``GRUCell``, ``BasicRNNCell``, stacks as ``MultiRNNCell``
wires them, the dense output layer), in float64, pinned by central finite differences of the forward pass in
``tests/test_fit_oracle.py``.  Adam: ``oracle/fit.py``.  Parity status: **unpinned at the TensorFlow boundary**.
"""

import numpy as np

from .fit import adam_step
from .rnn_cells import PARAMS_PER_LAYER

FORGET_BIAS = 1.0


def _sig(x):
    return 1.0 / (1.0 + np.exp(-x))


def _act_pair(name):
    """``(f, f')`` with ``f'`` as a function of the pre-activation."""
    if name is None or name == "identity":
        return (lambda x: x), (lambda x: np.ones_like(x))
    if name == "relu":
        return (lambda x: np.maximum(x, 0.0)), (lambda x: (x > 0.0).astype(np.float64))
    if name == "tanh":
        return np.tanh, (lambda x: 1.0 - np.tanh(x) ** 2)
    if name == "sigmoid":
        return _sig, (lambda x: _sig(x) * (1.0 - _sig(x)))
    if name == "swish":
        return (lambda x: x * _sig(x)), (lambda x: _sig(x) * (1.0 + x * (1.0 - _sig(x))))
    raise ValueError("unsupported activation %r" % (name,))


def zero_state(cell_type, hidden_sizes, batch):
    """Per layer ``(c, h)`` for an LSTM layer, ``h`` otherwise (``:273-293``)."""
    z = lambda u: np.zeros((batch, u))  # noqa: E731
    return [(z(u), z(u)) if cell_type == "lstm" else z(u) for u in hidden_sizes]


def _layer_forward(cell_type, x, st, p, act):
    """One cell call.  Returns ``(h, new_state, cache)``."""
    if cell_type == "lstm":
        c_prev, h_prev = st
        inp = np.concatenate([x, h_prev], axis=1)
        i, j, f, o = np.split(inp @ p[0] + p[1], 4, axis=1)
        si, aj, sf, so = _sig(i), act(j), _sig(f + FORGET_BIAS), _sig(o)
        c = sf * c_prev + si * aj
        h = so * act(c)
        return h, (c, h), (inp, j, si, aj, sf, so, c_prev, c)
    if cell_type == "gru":
        h_prev = st
        inp = np.concatenate([x, h_prev], axis=1)
        r, u = np.split(_sig(inp @ p[0] + p[1]), 2, axis=1)
        inp2 = np.concatenate([x, r * h_prev], axis=1)
        pc = inp2 @ p[2] + p[3]
        c = act(pc)
        h = u * h_prev + (1.0 - u) * c
        return h, h, (inp, inp2, r, u, pc, c, h_prev)
    h_prev = st
    inp = np.concatenate([x, h_prev], axis=1)
    z = inp @ p[0] + p[1]
    h = act(z)
    return h, h, (inp, z)


def _layer_backward(cell_type, dh, dstate, cache, p, g, act, dact, nx):
    """Adds the layer's parameter gradients to ``g``; returns ``(dx, dstate_prev)``; ``dstate`` is the gradient that
    arrives from the next time step (``(dc, dh)`` for LSTM, ``dh`` otherwise, ``None`` at the chunk's last step)."""
    if cell_type == "lstm":
        inp, j, si, aj, sf, so, c_prev, c = cache
        dc_in, dh_in = dstate if dstate is not None else (0.0, 0.0)
        dh = dh + dh_in
        do = dh * act(c) * so * (1.0 - so)
        dc = dc_in + dh * so * dact(c)
        dz = np.concatenate([dc * aj * si * (1.0 - si), dc * si * dact(j), dc * c_prev * sf * (1.0 - sf), do], axis=1)
        g[0] += inp.T @ dz
        g[1] += dz.sum(axis=0)
        dinp = dz @ p[0].T
        return dinp[:, :nx], (dc * sf, dinp[:, nx:])
    if cell_type == "gru":
        inp, inp2, r, u, pc, c, h_prev = cache
        dh = dh + (dstate if dstate is not None else 0.0)
        du = dh * (h_prev - c)
        dpc = dh * (1.0 - u) * dact(pc)
        dh_prev = dh * u
        g[2] += inp2.T @ dpc
        g[3] += dpc.sum(axis=0)
        dinp2 = dpc @ p[2].T
        dx = dinp2[:, :nx]
        drh = dinp2[:, nx:]
        dh_prev = dh_prev + drh * r
        dv = np.concatenate([drh * h_prev * r * (1.0 - r), du * u * (1.0 - u)], axis=1)
        g[0] += inp.T @ dv
        g[1] += dv.sum(axis=0)
        dinp = dv @ p[0].T
        return dx + dinp[:, :nx], dh_prev + dinp[:, nx:]
    inp, z = cache
    dh = dh + (dstate if dstate is not None else 0.0)
    dz = dh * dact(z)
    g[0] += inp.T @ dz
    g[1] += dz.sum(axis=0)
    dinp = dz @ p[0].T
    return dinp[:, :nx], dinp[:, nx:]


def chunk_forward(params, x, state, hidden_sizes, cell_type, hidden_nonlinearity="tanh", output_nonlinearity=None,
                  keep=False):
    """``tf.nn.dynamic_rnn`` over ``x [batch, steps, in]`` from ``state`` + the dense output layer
    (``dynamics/core/utils.py:231-238``).  Returns ``(pred [batch, steps, out], new_state, caches)``."""
    act, _ = _act_pair(hidden_nonlinearity)
    oact, _ = _act_pair(output_nonlinearity)
    ppl = PARAMS_PER_LAYER[cell_type]
    state = list(state)
    preds, caches = [], []
    for t in range(x.shape[1]):
        inp, step_cache = x[:, t], []
        for l in range(len(hidden_sizes)):
            inp, state[l], cache = _layer_forward(cell_type, inp, state[l], params[ppl * l:ppl * (l + 1)], act)
            step_cache.append(cache)
        pre = inp @ params[-2] + params[-1]
        preds.append(oact(pre))
        if keep:
            caches.append((step_cache, inp, pre))
    return np.stack(preds, axis=1), state, caches


def chunk_loss(params, x, y, state, hidden_sizes, cell_type, hidden_nonlinearity="tanh", output_nonlinearity=None):
    pred, _, _ = chunk_forward(params, x, state, hidden_sizes, cell_type, hidden_nonlinearity, output_nonlinearity)
    return float(np.mean((pred - y) ** 2))


def chunk_gradients(params, x, y, state, hidden_sizes, cell_type, hidden_nonlinearity="tanh", output_nonlinearity=None):
    """``sess.run([loss, tf.gradients(loss, params), next_hidden_state])`` of one chunk (``:175-176``): the state the chunk
    starts from is a constant.  Returns ``(loss, grads, new_state)``."""
    params = [np.asarray(p, dtype=np.float64) for p in params]
    act, dact = _act_pair(hidden_nonlinearity)
    _, doact = _act_pair(output_nonlinearity)
    ppl = PARAMS_PER_LAYER[cell_type]
    n_layers = len(hidden_sizes)
    pred, new_state, caches = chunk_forward(params, x, state, hidden_sizes, cell_type, hidden_nonlinearity,
                                            output_nonlinearity, keep=True)
    diff = pred - y
    loss = float(np.mean(diff ** 2))
    grads = [np.zeros_like(p) for p in params]
    dstate = [None] * n_layers
    in_dims = [x.shape[2]] + list(hidden_sizes[:-1])
    for t in reversed(range(x.shape[1])):
        step_cache, h_top, pre = caches[t]
        dd = 2.0 * diff[:, t] / diff.size * doact(pre)
        grads[-2] += h_top.T @ dd
        grads[-1] += dd.sum(axis=0)
        dh = dd @ params[-2].T
        for l in reversed(range(n_layers)):
            dh, dstate[l] = _layer_backward(cell_type, dh, dstate[l], step_cache[l], params[ppl * l:ppl * (l + 1)],
                                            grads[ppl * l:ppl * (l + 1)], act, dact, in_dims[l])
    return loss, grads, new_state


def rnn_fit_loop(params, train, test, orders, batch_size, backprop_steps, learning_rate, rolling_average_persitency,
                 hidden_sizes, cell_type, hidden_nonlinearity="tanh", output_nonlinearity=None):
    """``rnn_dynamics.py:146-217`` on normalised float64 path sets ``dict(obs, act, delta)`` of shape ``[paths, len, dim]``.
    ``orders``: per epoch the batch start rows in visiting order (at most ``len(orders)`` epochs are run).  Returns
    ``(params, last_epoch, history)`` with ``history = [(mean chunk loss, valid loss, rolling average)]``."""
    params = [np.array(p, dtype=np.float64) for p in params]
    m = [np.zeros_like(p) for p in params]
    v = [np.zeros_like(p) for p in params]
    step = 0
    kw = dict(hidden_sizes=hidden_sizes, cell_type=cell_type, hidden_nonlinearity=hidden_nonlinearity,
              output_nonlinearity=output_nonlinearity)
    x_tr = np.concatenate([train["obs"], train["act"]], axis=2)
    y_tr = train["delta"]
    x_te = np.concatenate([test["obs"], test["act"]], axis=2)
    y_te = test["delta"]
    epochs = len(orders)
    rolling = rolling_prev = None
    history, last_epoch = [], 0
    for epoch in range(epochs):
        losses = []
        for s in orders[epoch]:
            xb, yb = x_tr[s:s + batch_size], y_tr[s:s + batch_size]
            state = zero_state(cell_type, hidden_sizes, xb.shape[0])                       # :162
            all_grads = []
            for i in range(0, xb.shape[1], backprop_steps):                                # :167-179
                loss, grads, state = chunk_gradients(params, xb[:, i:i + backprop_steps], yb[:, i:i + backprop_steps],
                                                     state, **kw)
                all_grads.append(grads)
                losses.append(loss)
            mean_grads = [np.mean(g, axis=0) for g in zip(*all_grads)]                     # :181
            step += 1
            adam_step(params, mean_grads, m, v, step, learning_rate)                       # :183
        valid = chunk_loss(params, x_te, y_te, zero_state(cell_type, hidden_sizes, x_te.shape[0]), **kw)   # :186-196
        if rolling is None:                                                                # :198-203
            rolling, rolling_prev = 1.5 * valid, 2 * valid
            if valid < 0:
                rolling, rolling_prev = valid / 1.5, valid / 2
        rolling = rolling_average_persitency * rolling + (1.0 - rolling_average_persitency) * valid
        history.append((float(np.mean(losses)), valid, rolling))
        last_epoch = epoch
        if rolling_prev < rolling or epoch == epochs - 1:                                  # :214
            break
        rolling_prev = rolling
    return params, last_epoch, history
