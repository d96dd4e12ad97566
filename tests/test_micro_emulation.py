"""CPU check of the micro-tile kernels' index algebra (no GPU): ``tests/micro_emulator.py`` consumes the wave-stream
weight layout produced by the library's own host packer and must give the BITS of the 16-candidate kernel's lane-level
emulation (``tests/mfma_emulator.py``) - the claim `test_micro_tiles_are_bit_identical` then confirms on the chip."""

import numpy as np
import pytest

from learning_to_adapt_amd import _lib
from learning_to_adapt_amd.envs import SyntheticEnv
from learning_to_adapt_amd.utils import synthetic

import mfma_emulator as emu
import micro_emulator as micro

CASES = [
    # env, hidden, E, mode, m: HalfCheetah two hidden layers = the O4 quarter sums; one hidden layer and the Ant = without;
    # hidden 512 = two 64-unit tiles per wave (chunks side by side), 256 = one (chunks one after the other)
    ("half_cheetah", [256, 256], 3, "mean", 1),
    ("half_cheetah", [256], 2, "mean", 1),
    ("ant", [256, 256], 2, "per_block", 2),
    ("half_cheetah", [512, 512], 1, "single", 1),
]


@pytest.mark.parametrize("kind,hidden,E,mode,m", CASES)
def test_micro_tile_step_is_bit_identical_to_the_16_candidate_emulation(kind, hidden, E, mode, m):
    lib = _lib.load()
    env = SyntheticEnv(kind)
    od, ad = env.observation_space.shape[0], env.action_space.shape[0]
    if mode == "per_block":
        sets, norm = synthetic.make_adapted_sets(env, hidden, E)
        norms = [norm] * E
    else:
        sets, norms = synthetic.make_members(env, hidden, E)
    n = 16
    obs0 = synthetic.make_obs0(m, od)
    rs = np.random.RandomState(7)
    actions = rs.uniform(env.action_space.low, env.action_space.high, size=(1, m * n, ad)).astype(np.float32)
    reward = {k: getattr(env.reward_spec, k) for k in ("w_vel", "inv_dt", "alive", "ctrl_coef", "dist_coef", "vel_index", "dist_index")}
    packed = [emu.PackedSet(lib, sets[e], norms[e], od, ad) for e in range(E)]
    msets = [micro.MicroSet(lib, sets[e], norms[e], od, ad) for e in range(E)]
    env_i = m - 1
    _, valid, want = emu.rollout_workgroup(packed, mode, obs0[env_i], actions, env_i, 0, n, m, od, ad, 1.0, reward)
    assert valid.all()
    state = np.repeat(obs0[env_i][None, :].astype(np.float32), 4, axis=0)
    for q in (0, 3):            # two of the tile's four micro tiles
        got = micro.micro_step(msets, mode, env_i, state, actions[0, env_i * n + 4 * q:env_i * n + 4 * q + 4])
        assert np.array_equal(got.view(np.uint32), np.asarray(want[0, 4 * q:4 * q + 4], dtype=np.float32).view(np.uint32)), q


def test_micro_layout_has_no_instance_for_other_shapes():
    lib = _lib.load()
    assert lib.l2a_micro_layout_floats(20, 6, 2, 128) == 0
    assert lib.l2a_micro_layout_floats(20, 6, 2, 200) == 0
    assert lib.l2a_micro_layout_floats(70, 6, 2, 512) == 0
    assert lib.l2a_micro_layout_floats(20, 6, 2, 512) == 8 * (8 + 128 + 16) * 256
    assert lib.l2a_micro_layout_floats(41, 8, 3, 512) == 8 * (16 + 256 + 16) * 256
