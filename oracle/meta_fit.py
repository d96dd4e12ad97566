"""NumPy restatement of ``MetaMLPDynamicsModel.fit``'s meta-training loop.  TEST INFRASTRUCTURE ONLY.

Follows ``learning_to_adapt/dynamics/meta_mlp_dynamics.py``:

* ``:353-390`` ``_get_batch``: ``meta_batch_size`` windows of ``2 * batch_size`` consecutive transitions, path and position
  drawn with ``np.random.randint`` from the GLOBAL generator (two calls per batch, path indices first) - reproduced call
  for call, so the restatement and the product, seeded alike, train on the same windows;
* ``:96-141`` the meta-training graph: every window is split in two (``tf.split(., 2)``, ``:99-100``), the pre half gives
  ``pre_loss = reduce_mean(square(delta - pred))`` (``:118``) and the adapted parameters ``theta' = theta - alpha *
  grad(pre_loss)`` (``_adapt_sym``, ``:409-421``: ``tf.gradients``, NOT stopped - the outer gradient flows through the
  inner step), the post half ``post_loss`` at ``theta'`` (``:133``); ``train_op`` minimises the MEAN post loss over the tasks
  (``:139-141``) with ``tf.train.AdamOptimizer`` (the constructor's default ``optimizer``);
* ``:209-262`` epochs: ``num_steps_per_epoch = max(int(paths * len / (meta_batch * batch * 2)), 1)`` training steps, then
  ``num_steps_test`` windows from the held-out set through the PLAIN loss of the un-adapted network (``self.loss``,
  ``:88``), their mean, the rolling average (``1.5 x`` / ``2 x`` start values) and the stop rule.

The outer gradient: ``d post(theta - alpha g(theta)) / d theta = v - alpha * H(theta) v`` with ``v = grad post (theta')`` and
``H`` the Hessian of the pre loss.  ``g`` and ``v`` are ``oracle/adapt.py``'s hand-written backward pass (pinned by finite
differences there); ``H v`` is the directional derivative of ``g`` along ``v``, taken as a float64 central difference of that
analytic gradient (relative error ~1e-9 - six orders below what the comparison with the fp32 product resolves).  Adam:
``oracle/fit.py`` (TensorFlow 1.13's documented update).  Parity status: **unpinned at the TensorFlow boundary**.
"""

import numpy as np

from .adapt import loss_gradients, pre_loss
from .fit import adam_step


def get_batch(dataset, meta_batch_size, batch_size):
    """``_get_batch`` (``:353-390``): list of ``(x_window, y_window)`` with ``x = [obs | act]``."""
    num_paths, len_path = dataset["obs"].shape[:2]
    idx_path = np.random.randint(0, num_paths, size=meta_batch_size)
    idx_batch = np.random.randint(batch_size, len_path - batch_size, size=meta_batch_size)
    out = []
    for ip, ib in zip(idx_path, idx_batch):
        sl = slice(ib - batch_size, ib + batch_size)
        out.append((np.concatenate([dataset["obs"][ip, sl], dataset["act"][ip, sl]], axis=1), dataset["delta"][ip, sl]))
    return out


def maml_gradients(params, windows, batch_size, inner_lr, hidden_nonlinearity, output_nonlinearity):
    """Mean pre / post loss over the tasks and the gradient of the mean post loss w.r.t. ``params`` (through the inner step)."""
    kw = dict(hidden_nonlinearity=hidden_nonlinearity, output_nonlinearity=output_nonlinearity, dtype=np.float64)
    total = [np.zeros_like(p) for p in params]
    pre_sum = post_sum = 0.0
    for xw, yw in windows:
        xa, ya, xb, yb = xw[:batch_size], yw[:batch_size], xw[batch_size:], yw[batch_size:]       # :99-100
        g = loss_gradients(params, xa, ya, **kw)
        fast = [p - inner_lr * gi for p, gi in zip(params, g)]                                      # :409-421
        v = loss_gradients(fast, xb, yb, **kw)
        vnorm = np.sqrt(sum(float(np.sum(vi * vi)) for vi in v))
        if vnorm > 0.0:
            eps = 1e-5 / vnorm
            gp = loss_gradients([p + eps * vi for p, vi in zip(params, v)], xa, ya, **kw)
            gm = loss_gradients([p - eps * vi for p, vi in zip(params, v)], xa, ya, **kw)
            hv = [(a - b) / (2.0 * eps) for a, b in zip(gp, gm)]
        else:
            hv = [np.zeros_like(p) for p in params]
        for t, vi, hi in zip(total, v, hv):
            t += vi - inner_lr * hi
        pre_sum += float(pre_loss(params, xa, ya, **kw))
        post_sum += float(pre_loss(fast, xb, yb, **kw))
    n = float(len(windows))
    return pre_sum / n, post_sum / n, [t / n for t in total]


def meta_fit_loop(params, dataset_train, dataset_test, meta_batch_size, batch_size, inner_lr, learning_rate,
                  rolling_average_persitency, epochs, hidden_nonlinearity="relu", output_nonlinearity=None):
    """``:205-262`` on normalised float64 data sets ``dict(obs, act, delta)`` of shape ``[paths, len, dim]``; draws its windows
    from ``np.random`` exactly as the reference does.  Returns ``(params, last_epoch, history)``."""
    params = [np.array(p, dtype=np.float64) for p in params]
    m = [np.zeros_like(p) for p in params]
    v = [np.zeros_like(p) for p in params]
    t = 0
    steps_per_epoch = max(int(np.prod(dataset_train["obs"].shape[:2]) / (meta_batch_size * batch_size * 2)), 1)
    steps_test = max(int(np.prod(dataset_test["obs"].shape[:2]) / (meta_batch_size * batch_size * 2)), 1)
    rolling = rolling_prev = None
    history = []
    last_epoch = 0
    for epoch in range(epochs):
        pre_l, post_l = [], []
        for _ in range(steps_per_epoch):
            windows = get_batch(dataset_train, meta_batch_size, batch_size)
            pre, post, grads = maml_gradients(params, windows, batch_size, inner_lr, hidden_nonlinearity, output_nonlinearity)
            t += 1
            adam_step(params, grads, m, v, t, learning_rate)
            pre_l.append(pre)
            post_l.append(post)
        valid_losses = []
        for _ in range(steps_test):                                                                 # :227-236
            ws = get_batch(dataset_test, meta_batch_size, batch_size)
            xv = np.concatenate([w[0] for w in ws], axis=0)
            yv = np.concatenate([w[1] for w in ws], axis=0)
            valid_losses.append(float(pre_loss(params, xv, yv, hidden_nonlinearity, output_nonlinearity, dtype=np.float64)))
        valid = float(np.mean(valid_losses))
        if rolling is None:                                                                         # :238-243
            rolling, rolling_prev = 1.5 * valid, 2 * valid
            if valid < 0:
                rolling, rolling_prev = valid / 1.5, valid / 2
        rolling = rolling_average_persitency * rolling + (1.0 - rolling_average_persitency) * valid
        history.append((float(np.mean(post_l)), float(np.mean(pre_l)), valid, rolling))
        last_epoch = epoch
        if rolling_prev < rolling or epoch == epochs - 1:                                            # :258
            break
        rolling_prev = rolling
    return params, last_epoch, history
