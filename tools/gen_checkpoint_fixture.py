#!/usr/bin/env python
"""Write snapshots in the reference trainer's on-disk format - `dict(itr, policy, env, dynamics_model)` dumped with
`joblib.dump(..., compress=3)` (`trainers/mb_trainer.py:118-122`, `logger/logger.py:376-396`) - WITHOUT this
package's classes: throw-away classes are registered under the reference's module paths
(`learning_to_adapt.dynamics.meta_mlp_dynamics.MetaMLPDynamicsModel`, ...) whose `__getstate__` returns exactly the
state the reference classes return (`utils/serializable.py:44-45`; `dynamics/meta_mlp_dynamics.py:434-440`,
`dynamics/rnn_dynamics.py:319-324`, `dynamics/core/layers.py:103-108`; env wrapped by `normalize(...)`,
`envs/normalized_env.py:116-120`), so the byte stream names the reference's classes and carries the reference's
state layout.  The weights are the seeded recipe of the matching golden plan case, so that a controller rebuilt
from the file must reproduce that case's golden vector (tests/test_checkpoint.py).

    python tools/gen_checkpoint_fixture.py      # rewrites tests/golden/ref_snapshot_*.pkl
"""
import os
import sys
import types
from collections import OrderedDict

import joblib
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402


def ref_class(module, name, extra_state=None):
    """A class that pickles as `module.name` with the reference's Serializable state (+ extra_state(self))."""
    if module not in sys.modules:
        parts = module.split(".")
        for i in range(1, len(parts) + 1):
            sys.modules.setdefault(".".join(parts[:i]), types.ModuleType(".".join(parts[:i])))

    def __init__(self, *args, **kwargs):
        self._a, self._k = args, kwargs

    def __getstate__(self):
        base = {"__args": self._a, "__kwargs": self._k}                 # utils/serializable.py:44-45
        return extra_state(self, base) if extra_state else base
    cls = type(name, (object,), {"__init__": __init__, "__getstate__": __getstate__, "__module__": module})
    setattr(sys.modules[module], name, cls)
    return cls


def ref_function(module, name):
    """A function that pickles as the global `module.name` (how `tf.nn.tanh` & co. appear among recorded arguments)."""
    if module not in sys.modules:
        parts = module.split(".")
        for i in range(1, len(parts) + 1):
            sys.modules.setdefault(".".join(parts[:i]), types.ModuleType(".".join(parts[:i])))

    def fn(*args, **kwargs):
        raise RuntimeError("fixture stand-in")
    fn.__name__ = fn.__qualname__ = name
    fn.__module__ = module
    setattr(sys.modules[module], name, fn)
    return fn


def dynamics_state(self, base):         # meta_mlp_dynamics.py:434-440 / mlp_dynamics.py / rnn_dynamics.py:319-324
    return {"init_args": base, "normalization": self.normalization,
            "networks": [{"network_params": self.network_params}]}       # core/layers.py:103-108


def normalized_state(self, base):       # envs/normalized_env.py:116-120
    return dict(base, _obs_mean=self._obs_mean, _obs_var=self._obs_var)


ENV_CLS = {"half_cheetah": ("learning_to_adapt.envs.half_cheetah_env", "HalfCheetahEnv"),
           "ant": ("learning_to_adapt.envs.ant_env", "AntEnv")}


def make_env(kind, obs_dim):
    Env = ref_class(*ENV_CLS[kind])
    Norm = ref_class("learning_to_adapt.envs.normalized_env", "NormalizedEnv", normalized_state)
    inner = Env(None, True)                                              # task=None, reset_every_episode=True
    env = Norm(inner, 1., False, False, 0.001, 0.001, 1.)
    env._obs_mean, env._obs_var = np.zeros(obs_dim), np.ones(obs_dim)
    return env


def named(params, names):
    return OrderedDict((k, np.asarray(p, dtype=np.float32)) for k, p in zip(names, params))


def mlp_names(n_hidden):
    out = []
    for i in range(n_hidden):
        out += ["hidden_%d/kernel" % i, "hidden_%d/bias" % i]
    return out + ["output/kernel", "output/bias"]


def write(name, snapshot):
    path = os.path.join(ROOT, "tests", "golden", name)
    joblib.dump(snapshot, path, compress=3)
    print("wrote", path, os.path.getsize(path), "bytes")


def main():
    Adam = ref_class("tensorflow.python.training.adam", "AdamOptimizer")     # the default `optimizer=` argument
    # ---- run_mb_mpc.py shape: MLPDynamicsModel + MPCController (random shooting) -----------------------------
    case = cases.CASES["hc_rs_h128_1layer"]
    env_obj, sets, norms = cases.recipe(case)
    env = make_env("half_cheetah", 20)
    Dyn = ref_class("learning_to_adapt.dynamics.mlp_dynamics", "MLPDynamicsModel", dynamics_state)
    # positional args in the order of the reference constructor (mlp_dynamics.py:28-44)
    dyn = Dyn("dyn", env, tuple(case["hidden"]), "relu", None, 500, 0.001, True, Adam, 0.2, 0.99)
    dyn.normalization = OrderedDict((k, (np.asarray(v[0]), np.asarray(v[1]))) for k, v in norms[0].items())
    dyn.network_params = named(sets[0], mlp_names(len(case["hidden"])))
    Pol = ref_class("learning_to_adapt.policies.mpc_controller", "MPCController")
    # mpc_controller.py:7-21: name, env, dynamics_model, reward_model, discount, use_cem, n_candidates, horizon,
    # num_cem_iters, percent_elites, use_reward_model, alpha
    pol = Pol("policy", env, dyn, None, 1, False, case["n"], case["h"], 8, 0.1, False, 0.1)
    write("ref_snapshot_mb_mpc.pkl", dict(itr=3, policy=pol, env=env, dynamics_model=dyn))

    # ---- run_grbal.py shape: MetaMLPDynamicsModel -----------------------------------------------------------
    case = cases.CASES["hc_rs_sigmoid_3x128"]
    env_obj, sets, norms = cases.recipe(case)
    Meta = ref_class("learning_to_adapt.dynamics.meta_mlp_dynamics", "MetaMLPDynamicsModel", dynamics_state)
    # meta_mlp_dynamics.py:22-40: name, env, hidden_sizes, meta_batch_size, hidden_nonlinearity,
    # output_nonlinearity, batch_size, learning_rate, inner_learning_rate, normalize_input, optimizer,
    # valid_split_ratio, rolling_average_persitency
    meta = Meta("dyn", env, tuple(case["hidden"]), 10, "sigmoid", None, 16, 0.001, 0.01, True, Adam, 0.2, 0.99)
    meta.normalization = OrderedDict((k, (np.asarray(v[0]), np.asarray(v[1]))) for k, v in norms[0].items())
    meta.network_params = named(sets[0], mlp_names(len(case["hidden"])))
    pol = Pol("policy", env, meta, None, 1, False, case["n"], case["h"], 8, 0.1, False, 0.1)
    write("ref_snapshot_grbal.pkl", dict(itr=11, policy=pol, env=env, dynamics_model=meta))

    # ---- run_rebal.py shape: RNNDynamicsModel + RNNMPCController --------------------------------------------
    case = cases.CASES["hc_rnn_rs_u128_n40_h3"]
    env_obj, params, norm = cases.rnn_recipe(case)
    Rnn = ref_class("learning_to_adapt.dynamics.rnn_dynamics", "RNNDynamicsModel", dynamics_state)
    # rnn_dynamics.py:16-30: name, env, hidden_sizes, cell_type, hidden_nonlinearity, output_nonlinearity,
    # batch_size, learning_rate, normalize_input, optimizer, valid_split_ratio, rolling_average_persitency,
    # backprop_steps
    # run_rebal.py:77-99 never passes hidden_nonlinearity, so quick_init records the DEFAULT - the function object
    # tf.nn.tanh (rnn_dynamics.py:21,32) - which pickles as a global of TensorFlow's generated op module
    tf_tanh = ref_function("tensorflow.python.ops.gen_math_ops", "tanh")
    rnn = Rnn("dyn", env, (case["units"],), "lstm", tf_tanh, None, 10, 0.001, True, Adam, 0.2, 0.99, 50)
    rnn.normalization = OrderedDict((k, (np.asarray(v[0]), np.asarray(v[1]))) for k, v in norm.items())
    rnn.network_params = named(params, ["rnn/lstm_cell/kernel", "rnn/lstm_cell/bias", "output/kernel", "output/bias"])
    RPol = ref_class("learning_to_adapt.policies.rnn_mpc_controller", "RNNMPCController",
                     lambda self, base: {"init_args": base})            # rnn_mpc_controller.py:189-192
    # rnn_mpc_controller.py:8-21: name, env, dynamics_model, reward_model, discount, use_cem, n_candidates, horizon,
    # num_cem_iters, percent_elites, use_reward_model
    rpol = RPol("policy", env, rnn, None, 1, False, case["n"], case["h"], 8, 0.05, False)
    write("ref_snapshot_rebal.pkl", dict(itr=5, policy=rpol, env=env, dynamics_model=rnn))


if __name__ == "__main__":
    main()
