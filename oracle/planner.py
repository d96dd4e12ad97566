"""NumPy restatement of the MPC planner.  TEST INFRASTRUCTURE ONLY.

Restates ``learning_to_adapt/policies/mpc_controller.py``:

* ``get_random_action`` (``:67-69``) - one ``np.random.uniform(low, high,
  (h*n*m, act_dim))`` draw from NumPy's *legacy global* MT19937, reshaped to
  ``[h, n*m, act_dim]`` (``:114``).  Row ``r`` of a horizon slice belongs to
  env ``r // n`` and candidate ``r % n`` because the observations are laid out
  with ``np.repeat(observations, n, axis=0)`` (``:119``).
* ``get_rs_action`` (``:108-129``) - horizon loop, ``returns += discount**t *
  reward`` (``:126``), ``reshape(m, n)``, ``argmax`` (first maximum wins),
  gather of the first action (``:128-129``).
* ``get_cem_action`` (``:71-106``) - including its three quirks (SURVEY.md
  section 3.3): rollouts use the UNCLIPPED samples, the "elite" mask is
  ``((-returns).argsort(-1) < num_elites).T`` (a rank test on the argsort
  *values*, not a top-k), and for m > 1 the action rows are candidate-major
  while the observation rows are env-major.

Unlike the reference these functions also hand back the full returns table so
that tests can compare more than the chosen action.  Pinned against the real
reference class by ``tools/gen_golden.py`` (fixtures under ``tests/golden``).
"""

import numpy as np


def sample_rs_actions(low, high, n, m, h):
    """mpc_controller.py:67-69 + :114.  Consumes h*n*m*act_dim doubles of the global stream."""
    low = np.asarray(low, dtype=np.float64)
    high = np.asarray(high, dtype=np.float64)
    flat = np.random.uniform(low=low, high=high, size=(h * n * m,) + low.shape)
    return flat.reshape((h, n * m, -1))


def rollout_returns(dynamics_model, reward_fn, observations, actions, n, discount):
    """The shared horizon loop (``:116-127`` for RS, ``:92-99`` for CEM).

    ``actions``: float64 ``[h, n*m, act_dim]``.  Returns float64 ``[n*m]``.
    """
    h = actions.shape[0]
    total = np.zeros((actions.shape[1],))
    state = np.repeat(np.asarray(observations, dtype=np.float64), n, axis=0)
    for t in range(h):
        nxt = dynamics_model.predict(state, actions[t])
        total += discount ** t * reward_fn(state, actions[t], nxt)
        state = nxt
    return total


def rs_plan(dynamics_model, reward_fn, observations, low, high, n, h, discount=1.0,
            actions=None):
    """Random shooting.  Returns ``(chosen[m, act_dim], best_idx[m], returns[m, n], actions)``."""
    observations = np.asarray(observations, dtype=np.float64)
    m = len(observations)
    if actions is None:
        actions = sample_rs_actions(low, high, n, m, h)
    first = actions[0].reshape((m, n, -1))                       # :118
    returns = rollout_returns(dynamics_model, reward_fn, observations, actions, n, discount)
    returns = returns.reshape(m, n)                              # :128
    best = np.argmax(returns, axis=1)
    return first[np.arange(m), best], best, returns, actions


def cem_plan(dynamics_model, reward_fn, observations, low, high, n, h, discount=1.0,
             num_cem_iters=8, percent_elites=0.1, alpha=0.1, trace=None):
    """CEM in reference mode.  Returns ``(chosen[m, act_dim], best_idx[m], returns[m, n])``
    of the LAST iteration.  ``trace`` (a list) receives per-iteration
    ``dict(mean, std, returns)`` copies when given.
    """
    observations = np.asarray(observations, dtype=np.float64)
    low = np.asarray(low, dtype=np.float64)
    high = np.asarray(high, dtype=np.float64)
    m = len(observations)
    act_dim = low.shape[0]

    num_elites = max(int(n * percent_elites), 1)                 # :78
    mean = np.zeros((m, h * act_dim))
    std = np.ones((m, h * act_dim))
    clip_low = np.concatenate([low] * h)
    clip_high = np.concatenate([high] * h)

    returns = None
    first = None
    for _ in range(num_cem_iters):
        z = np.random.normal(size=(n, m, h * act_dim))           # :85
        raw = mean + z * std
        clipped = np.clip(raw, clip_low, clip_high)              # :87 (only the elites use it)
        # :88-89  rows are candidate-major here: row = j*m + i
        seq = np.transpose(raw.reshape((n * m, h, act_dim)), (1, 0, 2))
        first = seq[0].reshape((m, n, -1))                       # :94
        returns = rollout_returns(dynamics_model, reward_fn, observations, seq, n, discount)
        returns = returns.reshape(m, n)                          # :100
        elite_mask = ((-returns).argsort(axis=-1) < num_elites).T  # :101  [n, m] bool
        elites = clipped[elite_mask]                             # [num_elites*m, h*act_dim]
        mean = mean * alpha + (1 - alpha) * np.mean(elites, axis=0)
        std = np.std(elites, axis=0)
        if trace is not None:
            trace.append(dict(mean=np.array(mean), std=np.array(std), returns=np.array(returns)))

    best = np.argmax(returns, axis=1)
    return first[np.arange(m), best], best, returns
