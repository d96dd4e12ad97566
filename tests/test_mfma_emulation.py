"""CPU check of the MFMA kernel's index algebra (no GPU): the lane-level emulator in
``tests/mfma_emulator.py`` consumes weights packed by the library's own host packer and must
reproduce the oracle's returns / final states."""

import numpy as np
import pytest

from learning_to_adapt_amd import _lib
from learning_to_adapt_amd.envs import SyntheticEnv
from learning_to_adapt_amd.utils import synthetic
from oracle import OracleMLPDynamics, make_reward
from oracle.planner import rollout_returns

import mfma_emulator as emu


def _reward_dict(spec):
    return {k: getattr(spec, k) for k in ("w_vel", "inv_dt", "alive", "ctrl_coef", "dist_coef",
                                           "vel_index", "dist_index")}


CASES = [
    # kind, hidden, E, mode, m, n, h, act, discount
    ("half_cheetah", [128, 128], 2, "mean", 1, 20, 3, "relu", 1.0),
    ("ant", [128], 2, "per_block", 2, 16, 2, "tanh", 0.9),
    ("arm_7dof", [128, 128, 128], 1, "single", 1, 9, 2, "relu", 1.0),
]


@pytest.mark.parametrize("kind,hidden,E,mode,m,n,h,act,discount", CASES)
def test_emulator_matches_oracle(kind, hidden, E, mode, m, n, h, act, discount):
    lib = _lib.load()
    env = SyntheticEnv(kind)
    obs_dim = env.observation_space.shape[0]
    act_dim = env.action_space.shape[0]
    if mode == "per_block":
        sets, norm = synthetic.make_adapted_sets(env, hidden, E)
        norms = [norm] * E
    else:
        sets, norms = synthetic.make_members(env, hidden, E)
    dyn = OracleMLPDynamics(obs_dim, act_dim, sets, norms, mode=mode, hidden_nonlinearity=act)
    obs0 = synthetic.make_obs0(m, obs_dim)
    rs = np.random.RandomState(5)
    actions = rs.uniform(env.action_space.low, env.action_space.high, size=(h, m * n, act_dim))
    want = rollout_returns(dyn, make_reward(kind, env.dt), obs0, actions, n, discount).reshape(m, n)

    packed = [emu.PackedSet(lib, sets[e], norms[e], obs_dim, act_dim) for e in range(E)]
    a32 = actions.astype(np.float32)
    got = np.full((m, n), np.nan)
    for env_i in range(m):
        for tb in range((n + 15) // 16):
            ret, valid, _ = emu.rollout_workgroup(packed, mode, obs0[env_i], a32, env_i, tb, n, m,
                                                  obs_dim, act_dim, discount,
                                                  _reward_dict(env.reward_spec), hidden_act=act)
            for j in range(16):
                if valid[0, j]:
                    got[env_i, tb * 16 + j] = ret[0, j]
    assert not np.isnan(got).any()
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-4)
    assert np.array_equal(np.argmax(got, axis=1), np.argmax(want, axis=1))


def test_emulator_final_state_matches_predict():
    lib = _lib.load()
    env = SyntheticEnv("half_cheetah")
    obs_dim, act_dim = 20, 6
    sets, norms = synthetic.make_members(env, [128, 128], 1)
    dyn = OracleMLPDynamics(obs_dim, act_dim, sets, norms, mode="single")
    obs0 = synthetic.make_obs0(1, obs_dim)
    rs = np.random.RandomState(9)
    actions = rs.uniform(-1, 1, size=(1, 16, act_dim))
    want = dyn.predict(np.repeat(obs0, 16, axis=0), actions[0])
    packed = [emu.PackedSet(lib, sets[0], norms[0], obs_dim, act_dim)]
    _, valid, state = emu.rollout_workgroup(packed, "single", obs0[0], actions.astype(np.float32), 0, 0,
                                            16, 1, obs_dim, act_dim, 1.0,
                                            _reward_dict(env.reward_spec))
    assert valid.all()
    np.testing.assert_allclose(state[0], want, rtol=1e-5, atol=1e-5)
