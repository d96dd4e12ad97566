"""NumPy restatement of GrBAL's inner adaptation step.  TEST INFRASTRUCTURE ONLY.

Follows, line by line, ``learning_to_adapt/dynamics/meta_mlp_dynamics.py``:

* ``adapt`` (``:321-345``): every task's batch ``[B, dim]`` is followed by B zero rows
  (``np.concatenate([ob, np.zeros_like(ob)])``, ``:324-326``), the tasks are concatenated, ``_pad_inputs``
  (``:308-319``) appends all-zero tasks up to ``meta_batch_size``, then ALL rows - the zero rows too - go through
  ``_normalize_data`` (``:333-336``; ``delta = obs_next - obs`` is normalised with the delta statistics);
* the meta-training graph (``:96-120``): ``nn_input = concat([obs, act])`` is split into ``meta_batch_size`` task
  blocks (``tf.split``, ``:96-97``) and every block into a *pre* and a *post* half (``:99-100``): the pre half is
  exactly the B real rows, the post half the B (normalised) zero rows, which never enter the adapted parameters;
  ``pre_loss = reduce_mean(square(pre_delta - pre_mlp(pre_input)))`` (``:118``) - the mean runs over all
  ``B * obs_dim`` elements;
* ``_adapt_sym`` (``:409-421``): ``theta' = theta - inner_learning_rate * d pre_loss / d theta`` for every
  parameter of the MLP (``hidden_i/kernel``, ``hidden_i/bias``, ``output/kernel``, ``output/bias``).

The MLP is ``oracle/dynamics.py``'s (``core/utils.py:264-296``: ``x @ W + b``, nonlinearity).  TensorFlow's
``tf.gradients`` is the exact reverse-mode derivative of that graph in fp32; this file writes the same chain rule
by hand in fp32 (``dtype=np.float32``) or float64.  Parity status: **unpinned at the TensorFlow boundary** like the
forward pass (``oracle/__init__``); pinned instead against float64 central finite differences of the restated loss
(``tests/test_adapt_oracle.py``) - a derivation error in the backward pass cannot hide there.
"""

import numpy as np

from .dynamics import mlp_forward_f32, normalize


def _act_and_grad(name, dtype):
    """(activation, derivative as a function of the pre-activation z and the output y)."""
    one = dtype(1)
    if name is None or name == "identity":
        return (lambda z: z), (lambda z, y: np.ones_like(z))
    if name == "relu":
        return (lambda z: np.maximum(z, dtype(0))), (lambda z, y: (z > 0).astype(z.dtype))
    if name == "tanh":
        return np.tanh, (lambda z, y: one - y * y)
    if name == "sigmoid":
        return (lambda z: (one / (one + np.exp(-z))).astype(z.dtype)), (lambda z, y: y * (one - y))
    raise ValueError("unsupported activation %r" % (name,))


def build_adapt_batch(obs, act, obs_next, meta_batch_size, norm):
    """``adapt`` ``:321-336``: lists of per-task arrays -> the normalised ``(nn_input, delta)`` of the whole
    meta-batch, ``[meta_batch_size * 2B, .]`` float64, zero rows and zero tasks included."""
    assert len(obs) == len(act) == len(obs_next)                                     # :323
    n_adapt = len(obs)
    ob = np.concatenate([np.concatenate([o, np.zeros_like(o)], axis=0) for o in obs], axis=0)          # :324
    ac = np.concatenate([np.concatenate([a, np.zeros_like(a)], axis=0) for a in act], axis=0)          # :325
    on = np.concatenate([np.concatenate([o, np.zeros_like(o)], axis=0) for o in obs_next], axis=0)     # :326
    if n_adapt < meta_batch_size:                                                    # _pad_inputs :308-319
        pad = int(ob.shape[0] / n_adapt * (meta_batch_size - n_adapt))
        ob = np.concatenate([ob, np.zeros((pad,) + ob.shape[1:])], axis=0)
        ac = np.concatenate([ac, np.zeros((pad,) + ac.shape[1:])], axis=0)
        on = np.concatenate([on, np.zeros((pad,) + on.shape[1:])], axis=0)
    if norm is not None:                                                             # :333-336
        delta = normalize(on - ob, norm["delta"][0], norm["delta"][1])
        ob = normalize(ob, norm["obs"][0], norm["obs"][1])
        ac = normalize(ac, norm["act"][0], norm["act"][1])
    else:
        delta = on - ob                                                              # :338
    return np.concatenate([ob, ac], axis=1), delta                                   # nn_input: :61-68


def pre_split(nn_input, delta, meta_batch_size):
    """``:96-100``: task blocks, then the pre half of each."""
    x_tasks = np.split(nn_input, meta_batch_size, axis=0)
    d_tasks = np.split(delta, meta_batch_size, axis=0)
    return [np.split(x, 2, axis=0)[0] for x in x_tasks], [np.split(d, 2, axis=0)[0] for d in d_tasks]


def pre_loss(params, x, y, hidden_nonlinearity="relu", output_nonlinearity=None, dtype=np.float32):
    """``:118``: ``tf.reduce_mean(tf.square(pre_delta - pre_delta_pred))``."""
    pred = mlp_forward_f32(x, params, hidden_nonlinearity, output_nonlinearity, dtype=dtype)
    diff = np.asarray(y, dtype=dtype) - pred
    return np.mean(np.square(diff), dtype=dtype)


def loss_gradients(params, x, y, hidden_nonlinearity="relu", output_nonlinearity=None, dtype=np.float32):
    """Reverse-mode derivative of ``pre_loss`` w.r.t. ``[W0, b0, ..., Wout, bout]`` (what ``tf.gradients`` computes
    in ``_adapt_sym``, ``:412``)."""
    dt = np.dtype(dtype).type
    hid, dhid = _act_and_grad(hidden_nonlinearity, dt)
    out, dout = _act_and_grad(output_nonlinearity, dt)
    n_layers = len(params) // 2
    a = [np.asarray(x, dtype=dtype)]
    zs = []
    for li in range(n_layers):                                      # core/utils.py:273-292
        w = np.asarray(params[2 * li], dtype=dtype)
        b = np.asarray(params[2 * li + 1], dtype=dtype)
        z = a[-1] @ w + b
        zs.append(z)
        a.append(hid(z) if li < n_layers - 1 else out(z))
    y = np.asarray(y, dtype=dtype)
    # L = mean((y - pred)^2) over all B * obs_dim elements  ->  dL/dpred = 2 (pred - y) / (B * obs_dim)
    g = (dt(2) * (a[-1] - y) / dt(y.size)).astype(dtype)
    g = g * dout(zs[-1], a[-1])
    grads = [None] * len(params)
    for li in range(n_layers - 1, -1, -1):
        grads[2 * li] = (a[li].T @ g).astype(dtype)                 # d/dW of a @ W
        grads[2 * li + 1] = g.sum(axis=0).astype(dtype)             # d/db
        if li > 0:
            g = (g @ np.asarray(params[2 * li], dtype=dtype).T) * dhid(zs[li - 1], a[li])
    return grads


def adapt_sets(params, obs, act, obs_next, meta_batch_size, inner_learning_rate, norm,
               hidden_nonlinearity="relu", output_nonlinearity=None, dtype=np.float32):
    """``MetaMLPDynamicsModel.adapt``: the ``_adapted_param_values`` of the first ``len(obs)`` tasks (``:343-345``),
    one list ``[W0', b0', ...]`` per task."""
    nn_input, delta = build_adapt_batch(obs, act, obs_next, meta_batch_size, norm)
    pre_x, pre_y = pre_split(nn_input, delta, meta_batch_size)
    lr = np.dtype(dtype).type(inner_learning_rate)
    out = []
    for i in range(len(obs)):                                       # self._adapted_params[:num_adapted] :344
        grads = loss_gradients(params, pre_x[i], pre_y[i], hidden_nonlinearity, output_nonlinearity, dtype)
        out.append([np.asarray(p, dtype=dtype) - lr * g for p, g in zip(params, grads)])      # :415-417
    return out
