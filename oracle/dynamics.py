"""NumPy restatement of the learned-dynamics ``predict`` path.  TEST INFRASTRUCTURE ONLY.

Follows, line by line:

* ``learning_to_adapt/dynamics/mlp_dynamics.py:204-222`` (``predict``):
  normalise -> fp32 MLP -> denormalise -> ``obs + delta``;
* ``mlp_dynamics.py:242-251`` (``_normalize_data``), ``:265-266``
  (``normalize``: ``(x - mean) / (std + 1e-10)``), ``:269-270``
  (``denormalize``: ``x * (std + 1e-10) + mean``);
* ``dynamics/core/utils.py:111-142`` (``create_mlp``: ``tf.layers.dense``
  stack) and ``:264-296`` (``forward_mlp``: ``tf.matmul`` / ``tf.add`` /
  nonlinearity) - kernels are ``[in, out]`` row-major, parameter order
  ``hidden_0/kernel, hidden_0/bias, ..., output/kernel, output/bias``
  (``dynamics/core/layers.py:160-163``);
* ``mlp_dynamics.py:63-68``: placeholders are float32 and the network input is
  ``concat([obs, act], axis=1)``; the float64 host arrays are cast to fp32 on
  feed (``utils/tensor_utils.py:6-11``);
* ``meta_mlp_dynamics.py:296-306,143-163`` (``_predict`` with adapted weights):
  the batch is split into equal row blocks (``tf.split``), block *i* runs
  through weight set *i*, results are concatenated.  The zero-row padding of
  ``_pad_inputs`` (``:308-319``) only adds rows that are dropped again
  (``post_update_delta[:num_adapted]``), so it is a no-op here.

Mean-ensemble mode (BASELINE.json ``ens=5``) is NOT in the reference
(SURVEY.md section 0 / section 8 row A9).  Build-defined semantics:
``delta = mean_e denorm_e(MLP_e(norm_e(obs, act)))`` with the state shared
across members.

Parity status: unpinned at the TensorFlow boundary (see ``oracle/__init__``).
"""

import numpy as np

_EPS = 1e-10


def _act(name):
    # mlp_dynamics.py:16-23
    if name is None or name == "identity":
        return lambda x: x
    if name == "relu":
        return lambda x: np.maximum(x, np.float32(0))
    if name == "tanh":
        return np.tanh
    if name == "sigmoid":
        return lambda x: (1.0 / (1.0 + np.exp(-x))).astype(x.dtype)
    if name == "swish":
        return lambda x: (x * (1.0 / (1.0 + np.exp(-x)))).astype(x.dtype)
    raise ValueError("unsupported activation %r" % (name,))


def mlp_forward_f32(x, params, hidden_nonlinearity="relu", output_nonlinearity=None,
                    dtype=np.float32):
    """core/utils.py:111-142 / :264-296.  ``params`` = [W0, b0, W1, b1, ..., Wout, bout]."""
    assert len(params) % 2 == 0 and len(params) >= 2
    hid = _act(hidden_nonlinearity)
    out = _act(output_nonlinearity)
    x = np.asarray(x, dtype=dtype)
    n_layers = len(params) // 2
    for li in range(n_layers):
        w = np.asarray(params[2 * li], dtype=dtype)
        b = np.asarray(params[2 * li + 1], dtype=dtype)
        assert w.shape[0] == x.shape[-1]            # core/utils.py:277
        x = x @ w                                   # tf.matmul      :278
        x = x + b                                   # tf.add         :281
        x = hid(x) if li < n_layers - 1 else out(x)  # :286-292
    return x


def normalize(data, mean, std):
    return (data - mean) / (std + _EPS)             # mlp_dynamics.py:265-266


def denormalize(data, mean, std):
    return data * (std + _EPS) + mean               # mlp_dynamics.py:269-270


class OracleMLPDynamics(object):
    """Duck-typed ``dynamics_model`` for the reference / oracle planner.

    ``weight_sets``: list of E parameter lists ``[W0, b0, ..., Wout, bout]``.
    ``norms``: list of E dicts ``{'obs': (mean, std), 'act': ..., 'delta': ...}``
    (float64 vectors, as produced by ``compute_normalization``,
    ``mlp_dynamics.py:253-262``) or a single dict shared by all sets.
    ``mode``: ``'single'`` (E == 1), ``'per_block'`` (GrBAL adapted sets,
    row block i <-> set i) or ``'mean'`` (ensemble mean of deltas).
    """

    def __init__(self, obs_dim, act_dim, weight_sets, norms, mode="single",
                 hidden_nonlinearity="relu", output_nonlinearity=None,
                 mlp_dtype=np.float32):
        self.obs_space_dims = obs_dim
        self.action_space_dims = act_dim
        self.weight_sets = [list(ws) for ws in weight_sets]
        if isinstance(norms, dict):
            norms = [norms] * len(self.weight_sets)
        assert len(norms) == len(self.weight_sets)
        self.norms = norms
        assert mode in ("single", "per_block", "mean")
        if mode == "single":
            assert len(self.weight_sets) == 1
        self.mode = mode
        self.hidden_nonlinearity = hidden_nonlinearity
        self.output_nonlinearity = output_nonlinearity
        self.mlp_dtype = mlp_dtype

    def _delta(self, e, obs, act):
        nm = self.norms[e]
        o = normalize(obs, nm["obs"][0], nm["obs"][1])      # float64
        a = normalize(act, nm["act"][0], nm["act"][1])
        x = np.concatenate([o, a], axis=1)                   # mlp_dynamics.py:68
        d = mlp_forward_f32(x, self.weight_sets[e], self.hidden_nonlinearity,
                            self.output_nonlinearity, dtype=self.mlp_dtype)
        return denormalize(d, nm["delta"][0], nm["delta"][1])  # fp32 * f64 -> f64

    def predict(self, obs, act):
        assert obs.shape[0] == act.shape[0]                  # mlp_dynamics.py:205-207
        assert obs.ndim == 2 and obs.shape[1] == self.obs_space_dims
        assert act.ndim == 2 and act.shape[1] == self.action_space_dims
        obs = np.asarray(obs, dtype=np.float64)
        act = np.asarray(act, dtype=np.float64)
        if self.mode == "single":
            delta = self._delta(0, obs, act)
        elif self.mode == "per_block":
            nset = len(self.weight_sets)
            assert obs.shape[0] % nset == 0
            ob = np.split(obs, nset, axis=0)                 # tf.split, meta_mlp_dynamics.py:147
            ab = np.split(act, nset, axis=0)
            delta = np.concatenate([self._delta(i, ob[i], ab[i]) for i in range(nset)], axis=0)
        else:
            acc = np.zeros((obs.shape[0], self.obs_space_dims), dtype=np.float64)
            for e in range(len(self.weight_sets)):
                acc += self._delta(e, obs, act)
            delta = acc / len(self.weight_sets)
        assert delta.ndim == 2
        return obs + delta                                   # mlp_dynamics.py:220
