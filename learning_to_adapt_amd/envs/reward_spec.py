"""Closed-form planning rewards as data, so the rollout kernel can fuse them.

The reference evaluates ``env.reward(obs, act, next_obs)`` on the host once per horizon
step (``policies/mpc_controller.py:125``).  All five reference rewards are instances of

    r = w_vel * (next[vel_index] - obs[vel_index]) / dt + alive
        - ctrl_coef * sum(act**2) - dist_coef * ||next[dist_index : dist_index+3]||

* HalfCheetah (+Blocks, +HField): ``envs/half_cheetah_env.py:58-65`` - w_vel 1,
  vel_index obs_dim-3, ctrl_coef 0.05.
* Ant: ``envs/ant_env.py:56-66`` - w_vel 1, alive 0.05, ctrl_coef 0.
* Arm7Dof: ``envs/arm_7dof_env.py:91-99`` - dist_coef 1 over next[-3:], ctrl_coef 0.005.

``RewardSpec`` mirrors ``struct l2a_reward`` in ``include/l2a.h`` field for field.
"""

import ctypes

import numpy as np


class RewardSpec(ctypes.Structure):
    _fields_ = [
        ("w_vel", ctypes.c_float),
        ("inv_dt", ctypes.c_float),
        ("alive", ctypes.c_float),
        ("ctrl_coef", ctypes.c_float),
        ("dist_coef", ctypes.c_float),
        ("vel_index", ctypes.c_int),
        ("dist_index", ctypes.c_int),
        ("reserved", ctypes.c_int),
    ]

    @classmethod
    def make(cls, w_vel=0.0, dt=1.0, alive=0.0, ctrl_coef=0.0, dist_coef=0.0, vel_index=0, dist_index=0):
        """Build a spec; the float64 coefficients are kept beside the fp32 C fields so that the
        host-side ``evaluate`` reproduces the reference's float64 arithmetic exactly."""
        spec = cls(w_vel, (1.0 / dt) if w_vel != 0.0 else 0.0, alive, ctrl_coef, dist_coef,
                   int(vel_index), int(dist_index), 0)
        spec.exact = dict(w_vel=float(w_vel), dt=float(dt), alive=float(alive), ctrl_coef=float(ctrl_coef),
                          dist_coef=float(dist_coef))
        return spec

    @classmethod
    def half_cheetah(cls, obs_dim, dt):
        return cls.make(w_vel=1.0, dt=dt, ctrl_coef=1e-1 * 0.5, vel_index=obs_dim - 3)

    @classmethod
    def ant(cls, obs_dim, dt):
        return cls.make(w_vel=1.0, dt=dt, alive=0.05, vel_index=obs_dim - 3)

    @classmethod
    def arm_7dof(cls, obs_dim):
        return cls.make(ctrl_coef=0.01 * 0.5, dist_coef=1.0, dist_index=obs_dim - 3)

    @classmethod
    def none(cls):
        """All-zero reward (used by ``predict``-only launches)."""
        return cls.make()

    def evaluate(self, obs, act, next_obs):
        """Host NumPy evaluation of the same closed form (float64)."""
        ex = getattr(self, "exact", None) or dict(
            w_vel=float(self.w_vel), dt=(1.0 / float(self.inv_dt)) if self.inv_dt else 1.0,
            alive=float(self.alive), ctrl_coef=float(self.ctrl_coef), dist_coef=float(self.dist_coef))
        r = np.zeros((obs.shape[0],))
        if ex["w_vel"] != 0.0:
            r = r + ex["w_vel"] * (next_obs[:, self.vel_index] - obs[:, self.vel_index]) / ex["dt"]
        if ex["dist_coef"] != 0.0:
            d = self.dist_index
            r = r - ex["dist_coef"] * np.linalg.norm(next_obs[:, d:d + 3], axis=1)
        if ex["ctrl_coef"] != 0.0:
            r = r - ex["ctrl_coef"] * np.sum(np.square(act), axis=1)
        return r + ex["alive"]


_BY_CLASS_NAME = {
    "HalfCheetahEnv": "half_cheetah",
    "HalfCheetahBlocksEnv": "half_cheetah",
    "HalfCheetahHFieldEnv": "half_cheetah",
    "AntEnv": "ant",
    "Arm7DofEnv": "arm_7dof",
}


def _innermost(env):
    seen = 0
    while seen < 16:
        inner = None
        for attr in ("wrapped_env", "_wrapped_env"):
            inner = env.__dict__.get(attr) if hasattr(env, "__dict__") else None
            if inner is not None:
                break
        if inner is None:
            return env
        env, seen = inner, seen + 1
    return env


def reward_spec_for_env(env):
    """Find the fusable closed form for ``env`` or return ``None``.

    1. an explicit ``env.reward_spec`` attribute wins;
    2. otherwise the class name of the innermost wrapped env is matched against the
       reference's env classes (note N1 of SURVEY.md: ``NormalizedEnv`` keeps the real env
       in ``_wrapped_env`` and forwards attribute reads).
    """
    spec = getattr(env, "reward_spec", None)
    if spec is not None:
        return spec
    core = _innermost(env)
    kind = _BY_CLASS_NAME.get(type(core).__name__)
    if kind is None:
        return None
    obs_dim = int(env.observation_space.shape[0])
    if kind == "half_cheetah":
        return RewardSpec.half_cheetah(obs_dim, float(env.dt))
    if kind == "ant":
        return RewardSpec.ant(obs_dim, float(env.dt))
    return RewardSpec.arm_7dof(obs_dim)
