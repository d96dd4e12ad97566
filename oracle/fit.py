"""NumPy restatement of ``MLPDynamicsModel.fit``'s training loop.  TEST INFRASTRUCTURE ONLY.

Follows ``learning_to_adapt/dynamics/mlp_dynamics.py``:

* ``:106-117`` normalisation of the data set (``compute_normalization`` ``:224-240`` = per-column mean / std of obs,
  act and ``obs_next - obs``; ``_normalize_data`` ``:242-251`` = ``(x - mean) / (std + 1e-10)``) and the train /
  validation split (``train_test_split``, ``dynamics/utils.py``: a permutation of the rows, the first
  ``1 - valid_split_ratio`` of it trains);
* ``:140-165`` one epoch = one pass over the shuffled training set in batches of ``batch_size`` (the last one ragged),
  each batch one ``sess.run([self.loss, self.train_op])`` with ``loss = reduce_mean(square(delta - delta_pred))``
  (``:83``) and ``train_op = optimizer(learning_rate).minimize(loss)`` (``:84-85``; ``tf.train.AdamOptimizer``);
* ``:166-197`` the validation loss on the whole held-out set, its rolling average
  (``1.5 x`` / ``2 x`` the first value as start / previous, then ``p * avg + (1 - p) * loss``) and the stop rule
  ``prev < avg or epoch == epochs - 1``.

The shuffle itself is TensorFlow's (``tf.data`` ``shuffle`` buffer, ``:255-268``) and cannot be reproduced; the
batch ORDER is therefore an input (``orders[epoch]`` = a permutation of the training rows) - the product is driven with
the same permutations in ``tests/test_fit_oracle.py``.

Adam is restated from TensorFlow 1.13's documented update (``python/training/adam.py``):
``lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t); m = b1 m + (1 - b1) g; v = b2 v + (1 - b2) g^2;
theta -= lr_t * m / (sqrt(v) + eps)`` with ``b1 = 0.9, b2 = 0.999, eps = 1e-8``.  Parity status: **unpinned at the
TensorFlow boundary** (TensorFlow is absent here); the gradients are ``oracle/adapt.py``'s, pinned by finite differences.
"""

import numpy as np

from .adapt import loss_gradients, pre_loss


def adam_step(params, grads, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """One ``AdamOptimizer`` update, in place on the float64 lists ``params / m / v``; ``t`` counts from 1."""
    lr_t = lr * np.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    for i, g in enumerate(grads):
        g = np.asarray(g, dtype=np.float64)
        m[i] = beta1 * m[i] + (1.0 - beta1) * g
        v[i] = beta2 * v[i] + (1.0 - beta2) * g * g
        params[i] = params[i] - lr_t * m[i] / (np.sqrt(v[i]) + eps)


def fit_loop(params, x_train, y_train, x_test, y_test, orders, batch_size, learning_rate, rolling_average_persitency,
             hidden_nonlinearity="relu", output_nonlinearity=None):
    """``mlp_dynamics.py:140-197`` on normalised float64 data.  ``orders``: one permutation of the training rows per
    epoch (at most ``len(orders)`` epochs are run).  Returns ``(params, last_epoch, history)`` with ``history`` =
    ``[(mean batch loss, valid loss, rolling average)]`` per epoch."""
    params = [np.array(p, dtype=np.float64) for p in params]
    m = [np.zeros_like(p) for p in params]
    v = [np.zeros_like(p) for p in params]
    t = 0
    epochs = len(orders)
    rolling = rolling_prev = None
    history = []
    last_epoch = 0
    for epoch in range(epochs):
        order = np.asarray(orders[epoch])
        losses = []
        for s in range(0, len(order), batch_size):                               # :153-163
            idx = order[s:s + batch_size]
            xb, yb = x_train[idx], y_train[idx]
            losses.append(float(pre_loss(params, xb, yb, hidden_nonlinearity, output_nonlinearity, dtype=np.float64)))
            grads = loss_gradients(params, xb, yb, hidden_nonlinearity, output_nonlinearity, dtype=np.float64)
            t += 1
            adam_step(params, grads, m, v, t, learning_rate)
        valid = float(pre_loss(params, x_test, y_test, hidden_nonlinearity, output_nonlinearity, dtype=np.float64))  # :166-175
        if rolling is None:                                                      # :177-182
            rolling, rolling_prev = 1.5 * valid, 2 * valid
            if valid < 0:
                rolling, rolling_prev = valid / 1.5, valid / 2
        rolling = rolling_average_persitency * rolling + (1.0 - rolling_average_persitency) * valid      # :184-185
        history.append((float(np.mean(losses)), valid, rolling))
        last_epoch = epoch
        if rolling_prev < rolling or epoch == epochs - 1:                         # :194
            break
        rolling_prev = rolling
    return params, last_epoch, history
