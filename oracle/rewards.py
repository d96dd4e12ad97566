"""NumPy restatement of the closed-form planning rewards.  TEST INFRASTRUCTURE ONLY.

* HalfCheetah  - ``learning_to_adapt/envs/half_cheetah_env.py:58-65`` (identical in
  ``half_cheetah_blocks_env.py:56-63`` and ``half_cheetah_hfield_env.py:59-66``):
  ``(next[:, -3] - obs[:, -3]) / dt - 1e-1 * 0.5 * sum(a**2)``; dt = 0.01
  (``assets/half_cheetah.xml:43``).
* Ant          - ``envs/ant_env.py:56-66``:
  ``(next[:, -3] - obs[:, -3]) / dt - 0 + 0.05``; dt = 0.02 (``assets/ant.xml:3``).
* Arm 7-DoF    - ``envs/arm_7dof_env.py:91-99``:
  ``-||next[:, -3:]|| + 0.01 * 0.5 * (-sum(a**2))``.
"""

import numpy as np


def half_cheetah_reward(dt=0.01):
    def reward(obs, action, next_obs):
        assert obs.ndim == 2
        assert obs.shape == next_obs.shape
        assert obs.shape[0] == action.shape[0]
        ctrl_cost = 1e-1 * 0.5 * np.sum(np.square(action), axis=1)
        forward_reward = (next_obs[:, -3] - obs[:, -3]) / dt
        return forward_reward - ctrl_cost
    return reward


def ant_reward(dt=0.02):
    def reward(obs, action, next_obs):
        assert obs.ndim == 2
        assert obs.shape == next_obs.shape
        assert obs.shape[0] == action.shape[0]
        ctrl_cost = 0
        vel = (next_obs[:, -3] - obs[:, -3]) / dt
        survive_reward = 0.05
        return vel - ctrl_cost + survive_reward
    return reward


def arm_7dof_reward():
    def reward(obs, action, next_obs):
        assert obs.ndim == 2
        assert obs.shape == next_obs.shape
        assert obs.shape[0] == action.shape[0]
        vec = next_obs[:, -3:]
        reward_dist = -np.linalg.norm(vec, axis=1)
        reward_ctrl = -np.sum(np.square(action), axis=1)
        return reward_dist + 0.01 * 0.5 * reward_ctrl
    return reward


def make_reward(kind, dt=None):
    if kind == "half_cheetah":
        return half_cheetah_reward(0.01 if dt is None else dt)
    if kind == "ant":
        return ant_reward(0.02 if dt is None else dt)
    if kind == "arm_7dof":
        return arm_7dof_reward()
    raise ValueError(kind)
