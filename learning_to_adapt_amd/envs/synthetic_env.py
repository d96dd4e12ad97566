"""Simulator-free env stand-ins with the shapes of the reference's MuJoCo envs.

The planner consumes exactly four things from an env (``policies/mpc_controller.py:34-39``,
``policies/base.py:28-34``): ``reward(obs, act, next_obs)``, ``action_space``,
``observation_space`` and (through the reward) ``dt``.  The physics needs the proprietary
MuJoCo 1.31 binary and is out of scope.

Shapes:

* ``half_cheetah``: obs 20, act 6, ctrl range +-1, dt 0.01
  (``envs/half_cheetah_env.py:32-37``, ``assets/half_cheetah.xml:40,43,88-93``)
* ``ant``: obs 41, act 8, ctrl range +-150, dt 0.02 (``envs/ant_env.py:31-37``,
  ``assets/ant.xml:3,71-78``)
* ``arm_7dof``: obs 23 (qpos 10 + qvel 10 + object-target 3), act 7, +-1, dt 0.02
  (``envs/arm_7dof_env.py:85-89``, ``assets/arm_7dof.xml:4,83-89``)
"""

import numpy as np

from ..spaces import Box
from .reward_spec import RewardSpec

_SHAPES = {
    "half_cheetah": dict(obs_dim=20, act_dim=6, bound=1.0, dt=0.01),
    "ant": dict(obs_dim=41, act_dim=8, bound=150.0, dt=0.02),
    "arm_7dof": dict(obs_dim=23, act_dim=7, bound=1.0, dt=0.02),
}


class SyntheticEnv(object):
    def __init__(self, kind="half_cheetah", obs_dim=None, act_dim=None, bound=None, dt=None):
        cfg = dict(_SHAPES[kind])
        if obs_dim is not None:
            cfg["obs_dim"] = obs_dim
        if act_dim is not None:
            cfg["act_dim"] = act_dim
        if bound is not None:
            cfg["bound"] = bound
        if dt is not None:
            cfg["dt"] = dt
        self.kind = kind
        self.dt = cfg["dt"]
        self._obs_space = Box(-np.inf * np.ones(cfg["obs_dim"]), np.inf * np.ones(cfg["obs_dim"]))
        self._act_space = Box(-cfg["bound"] * np.ones(cfg["act_dim"]), cfg["bound"] * np.ones(cfg["act_dim"]))
        if kind == "half_cheetah":
            self.reward_spec = RewardSpec.half_cheetah(cfg["obs_dim"], self.dt)
        elif kind == "ant":
            self.reward_spec = RewardSpec.ant(cfg["obs_dim"], self.dt)
        else:
            self.reward_spec = RewardSpec.arm_7dof(cfg["obs_dim"])

    @property
    def observation_space(self):
        return self._obs_space

    @property
    def action_space(self):
        return self._act_space

    def reward(self, obs, action, next_obs):
        assert obs.ndim == 2
        assert obs.shape == next_obs.shape
        assert obs.shape[0] == action.shape[0]
        return self.reward_spec.evaluate(obs, action, next_obs)

    def reset(self):
        self._state = np.zeros(self._obs_space.shape)
        return self._state.copy()

    # A toy transition function so that closed-loop tests and demos can run the whole fit -> plan -> act cycle
    # without MuJoCo: a damped linear system whose velocity coordinate (the one the reward reads) is driven by a
    # fixed random mix of the actions.  Not a physics model.
    def _toy_matrices(self):
        if getattr(self, "_toy", None) is None:
            rs = np.random.RandomState(7)
            od, ad = self._obs_space.shape[0], self._act_space.shape[0]
            a = 0.9 * np.eye(od) + 0.02 * rs.randn(od, od)
            b = 0.05 * rs.randn(ad, od) / np.maximum(np.abs(self._act_space.high), 1e-6)[:, None]
            self._toy = (a, b)
        return self._toy

    def toy_dynamics(self, obs, act):
        """Vectorised ``next_obs = obs @ A + clip(act) @ B`` (float64)."""
        a, b = self._toy_matrices()
        act = np.clip(act, self._act_space.low, self._act_space.high)
        return np.asarray(obs, dtype=np.float64) @ a + act @ b

    def step(self, action):
        obs = getattr(self, "_state", None)
        if obs is None:
            obs = self.reset()
        nxt = self.toy_dynamics(obs[None], np.asarray(action, dtype=np.float64)[None])[0]
        rew = float(self.reward(obs[None], np.asarray(action, dtype=np.float64)[None], nxt[None])[0])
        self._state = nxt
        return nxt.copy(), rew, False, {}

    def log_diagnostics(self, paths, prefix=""):
        pass
