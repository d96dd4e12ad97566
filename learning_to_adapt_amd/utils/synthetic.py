"""Seeded synthetic inputs for the planner hot path (SURVEY.md section 8(d) recipe).

There is no network for datasets or checkpoints, so benchmarks, golden vectors and parity
tests all use random-init weights of the reference architecture:

* weights of member ``e``: ``RandomState(1000 + e)``, Xavier-uniform kernels
  ``U(+-sqrt(6 / (fan_in + fan_out)))`` shaped ``[in, out]`` in fp32 - the reference's
  initialiser (``dynamics/core/utils.py:81``).  The reference initialises biases to zero
  (``:82``); a trained model has non-zero biases, and a zero bias would hide indexing bugs,
  so biases here are ``bias_std * randn`` (default 0.05; pass 0.0 for the init-time values).
* normalisation of member ``e``: ``RandomState(2000 + e)``: ``mu_o = 0.1 randn``,
  ``sigma_o = 1 + rand``, ``mu_a = 0``, ``sigma_a = (high - low) / sqrt(12)``,
  ``mu_d = 0.01 randn``, ``sigma_d = 0.1 + 0.1 rand`` (float64, as
  ``compute_normalization`` produces, ``dynamics/mlp_dynamics.py:253-262``).
* ``obs0 = RandomState(1).randn(m, obs_dim)``.
* GrBAL-style adapted sets (config 3): base set 0 plus ``1e-3 * randn`` from
  ``RandomState(3000 + i)`` for block ``i``.
"""

from collections import OrderedDict

import numpy as np


def layer_sizes(obs_dim, act_dim, hidden_sizes):
    return [obs_dim + act_dim] + list(hidden_sizes) + [obs_dim]


def make_weight_set(obs_dim, act_dim, hidden_sizes, seed, bias_std=0.05):
    rs = np.random.RandomState(seed)
    sizes = layer_sizes(obs_dim, act_dim, hidden_sizes)
    params = []
    for fan_in, fan_out in zip(sizes[:-1], sizes[1:]):
        lim = np.sqrt(6.0 / (fan_in + fan_out))
        w = rs.uniform(-lim, lim, size=(fan_in, fan_out)).astype(np.float32)
        b = (bias_std * rs.randn(fan_out)).astype(np.float32)
        params += [w, b]
    return params


def make_norm(obs_dim, act_dim, low, high, seed):
    rs = np.random.RandomState(seed)
    mu_o = 0.1 * rs.randn(obs_dim)
    sig_o = 1.0 + rs.rand(obs_dim)
    mu_a = np.zeros(act_dim)
    sig_a = (np.asarray(high, dtype=np.float64) - np.asarray(low, dtype=np.float64)) / np.sqrt(12.0)
    mu_d = 0.01 * rs.randn(obs_dim)
    sig_d = 0.1 + 0.1 * rs.rand(obs_dim)
    norm = OrderedDict()
    norm["obs"] = (mu_o, sig_o)
    norm["delta"] = (mu_d, sig_d)
    norm["act"] = (mu_a, sig_a)
    return norm


def make_members(env, hidden_sizes, n_members, bias_std=0.05, seed_offset=0):
    """``n_members`` independent weight sets + normalisations (mean-ensemble / single)."""
    obs_dim = env.observation_space.shape[0]
    act_dim = env.action_space.shape[0]
    low, high = env.action_space.low, env.action_space.high
    sets = [make_weight_set(obs_dim, act_dim, hidden_sizes, 1000 + seed_offset + e, bias_std)
            for e in range(n_members)]
    norms = [make_norm(obs_dim, act_dim, low, high, 2000 + seed_offset + e) for e in range(n_members)]
    return sets, norms


def make_adapted_sets(env, hidden_sizes, n_blocks, bias_std=0.05, scale=1e-3):
    """One base set perturbed per block (stand-in for GrBAL's inner-adapted weights);
    a single normalisation shared by all blocks, as in ``meta_mlp_dynamics.py:276-294``."""
    obs_dim = env.observation_space.shape[0]
    act_dim = env.action_space.shape[0]
    base = make_weight_set(obs_dim, act_dim, hidden_sizes, 1000, bias_std)
    sets = []
    for i in range(n_blocks):
        rs = np.random.RandomState(3000 + i)
        sets.append([(p + scale * rs.randn(*p.shape)).astype(np.float32) for p in base])
    norm = make_norm(obs_dim, act_dim, env.action_space.low, env.action_space.high, 2000)
    return sets, norm


def make_lstm_set(obs_dim, act_dim, units, seed, bias_std=0.05):
    """Single-layer LSTM + output layer in the reference's variable order
    (``dynamics/core/utils.py:192-236``): ``rnn/lstm_cell/kernel [in + units, 4 units]`` (gate
    order i, j, f, o), ``rnn/lstm_cell/bias``, ``output/kernel [units, obs_dim]``, ``output/bias``.
    Glorot-uniform kernels (TF's default for ``LSTMCell`` and the ``w_init`` of the output layer)."""
    rs = np.random.RandomState(seed)
    k_in = obs_dim + act_dim + units
    lim = np.sqrt(6.0 / (k_in + 4 * units))
    kernel = rs.uniform(-lim, lim, size=(k_in, 4 * units)).astype(np.float32)
    bias = (bias_std * rs.randn(4 * units)).astype(np.float32)
    lim = np.sqrt(6.0 / (units + obs_dim))
    wout = rs.uniform(-lim, lim, size=(units, obs_dim)).astype(np.float32)
    bout = (bias_std * rs.randn(obs_dim)).astype(np.float32)
    return [kernel, bias, wout, bout]


def make_rnn_stack_set(obs_dim, act_dim, hidden_sizes, cell_type, seed, bias_std=0.05):
    """Variables of a stack of recurrent cells + output layer in ``get_params()`` order
    (``dynamics/rnn_cells.param_spec``): glorot-uniform kernels, small random biases (the GRU gate bias around its
    TensorFlow initial value 1.0) so that bias handling is exercised."""
    from ..dynamics import rnn_cells
    rs = np.random.RandomState(seed)
    out = []
    for name, shape in rnn_cells.param_spec(obs_dim, act_dim, hidden_sizes, cell_type):
        if len(shape) == 2:
            lim = np.sqrt(6.0 / (shape[0] + shape[1]))
            out.append(rs.uniform(-lim, lim, size=shape).astype(np.float32))
        else:
            base = 1.0 if name.endswith("gates/bias") else 0.0
            out.append((base + bias_std * rs.randn(*shape)).astype(np.float32))
    return out


def named_lstm_params(params):
    out = OrderedDict()
    for name, p in zip(("rnn/lstm_cell/kernel", "rnn/lstm_cell/bias", "output/kernel", "output/bias"), params):
        out[name] = p
    return out


def make_obs_sequence(m, obs_dim, steps):
    """Observations of ``steps`` consecutive controller steps (recurrent planner cases)."""
    return [np.random.RandomState(1 + k).randn(m, obs_dim) for k in range(steps)]


def make_obs0(m, obs_dim):
    return np.random.RandomState(1).randn(m, obs_dim)


def named_params(params):
    """``[W0, b0, ...]`` -> the reference's ``network_params`` OrderedDict
    (``dynamics/core/layers.py:160-163``: ``hidden_0/kernel, hidden_0/bias, ..., output/kernel,
    output/bias``)."""
    out = OrderedDict()
    n_layers = len(params) // 2
    for li in range(n_layers):
        prefix = "hidden_%d" % li if li < n_layers - 1 else "output"
        out[prefix + "/kernel"] = params[2 * li]
        out[prefix + "/bias"] = params[2 * li + 1]
    return out
