"""NumPy restatement of the recurrent planner.  TEST INFRASTRUCTURE ONLY.

Follows ``learning_to_adapt/policies/rnn_mpc_controller.py``:

* ``get_rs_action`` (``:112-134``): as ``mpc_controller.py:108-129`` plus a hidden state that
  is repeated ``n`` times per env (``repeat_hidden``, ``:165-187``) and threaded through
  ``dynamics_model.predict(obs, act, hidden)``;
* ``get_cem_action`` (``:71-110``): the MLP controller's CEM without the ``alpha`` smoothing
  (``mean = np.mean(elites)``, ``:107``) and with ``percent_elites=0.05`` by default (``:19``);
* ``get_actions`` (``:57-65``): after planning, ONE more ``predict`` on the real observations
  with the chosen actions advances the controller's own hidden state.

PINNED against the real ``RNNMPCController`` by ``tools/gen_golden.py`` (bit for bit).
"""

import numpy as np

from .planner import sample_rs_actions
from .rnn_dynamics import LSTMStateTuple


def repeat_hidden(hidden, n):
    """``rnn_mpc_controller.py:165-187``: every row n times, env-major - for an LSTM state, a plain array
    (GRU / RNN layer) or a list / tuple of those (stacked cells)."""
    if isinstance(hidden, LSTMStateTuple):
        return LSTMStateTuple(np.repeat(hidden.c, n, axis=0), np.repeat(hidden.h, n, axis=0))
    if isinstance(hidden, (list, tuple)):
        return [repeat_hidden(h, n) for h in hidden]
    return np.repeat(hidden, n, axis=0)


def rnn_rollout_returns(dynamics_model, reward_fn, observations, hidden, actions, n, discount):
    h = actions.shape[0]
    total = np.zeros((actions.shape[1],))
    state = np.repeat(np.asarray(observations, dtype=np.float64), n, axis=0)   # :119 / :95
    hid = repeat_hidden(hidden, n)                                              # :120 / :96
    for t in range(h):
        nxt, hid = dynamics_model.predict(state, actions[t], hid)               # :123
        total += discount ** t * reward_fn(state, actions[t], nxt)
        state = nxt
    return total


def rnn_rs_plan(dynamics_model, reward_fn, observations, hidden, low, high, n, h, discount=1.0,
                actions=None):
    """Returns ``(chosen[m, act_dim], best_idx[m], returns[m, n], next_hidden)``."""
    observations = np.asarray(observations, dtype=np.float64)
    m = len(observations)
    if actions is None:
        actions = sample_rs_actions(low, high, n, m, h)                         # :117
    first = actions[0].reshape((m, n, -1))
    returns = rnn_rollout_returns(dynamics_model, reward_fn, observations, hidden, actions, n,
                                  discount).reshape(m, n)
    best = np.argmax(returns, axis=1)
    chosen = first[np.arange(m), best]
    _, nxt_hidden = dynamics_model.predict(observations, chosen, hidden)        # :63
    return chosen, best, returns, nxt_hidden


def rnn_cem_plan(dynamics_model, reward_fn, observations, hidden, low, high, n, h, discount=1.0,
                 num_cem_iters=8, percent_elites=0.05, trace=None):
    """Returns ``(chosen, best_idx, returns of the last iteration, next_hidden)``."""
    observations = np.asarray(observations, dtype=np.float64)
    low = np.asarray(low, dtype=np.float64)
    high = np.asarray(high, dtype=np.float64)
    m = len(observations)
    act_dim = low.shape[0]
    num_elites = max(int(n * percent_elites), 1)                                # :78
    mean = np.zeros((m, h * act_dim))
    std = np.ones((m, h * act_dim))
    clip_low = np.concatenate([low] * h)
    clip_high = np.concatenate([high] * h)
    returns = first = None
    for _ in range(num_cem_iters):
        z = np.random.normal(size=(n, m, h * act_dim))                          # :85
        raw = mean + z * std
        clipped = np.clip(raw, clip_low, clip_high)
        seq = np.transpose(raw.reshape((n * m, h, act_dim)), (1, 0, 2))         # :88-89
        first = seq[0].reshape((m, n, -1))
        returns = rnn_rollout_returns(dynamics_model, reward_fn, observations, hidden, seq, n,
                                      discount).reshape(m, n)
        elite_mask = ((-returns).argsort(axis=-1) < num_elites).T               # :105
        elites = clipped[elite_mask]
        mean = np.mean(elites, axis=0)                                          # :107 (no alpha)
        std = np.std(elites, axis=0)
        if trace is not None:
            trace.append(dict(mean=np.array(mean), std=np.array(std), returns=np.array(returns)))
    best = np.argmax(returns, axis=1)
    chosen = first[np.arange(m), best]
    _, nxt_hidden = dynamics_model.predict(observations, chosen, hidden)
    return chosen, best, returns, nxt_hidden
