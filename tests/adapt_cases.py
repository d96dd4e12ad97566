"""Seeded inputs of the GrBAL adaptation cases shared by tools/gen_adapt_golden.py and the tests."""

import numpy as np

from learning_to_adapt_amd.envs import SyntheticEnv
from learning_to_adapt_amd.utils import synthetic

# name -> (env, hidden sizes, hidden nonlinearity, tasks adapted, meta_batch_size, rows per task)
CASES = {
    "ant_2x512_relu_m5_b16": ("ant", (512, 512), "relu", 5, 5, 16),          # run_grbal.py batch (adapt_batch_size=16)
    "ant_3x512_relu_m5_b16": ("ant", (512, 512, 512), "relu", 5, 5, 16),     # run_grbal.py:100 default network
    "ant_2x512_tanh_m3_b7": ("ant", (512, 512), "tanh", 3, 5, 7),            # ragged rows, fewer tasks than the meta batch
    "hc_2x128_sigmoid_m2_b16": ("half_cheetah", (128, 128), "sigmoid", 2, 2, 16),
    "hc_1x64_relu_m1_b3": ("half_cheetah", (64,), "relu", 1, 4, 3),          # non-MFMA shape, one task
    "arm_200x72_relu_m2_b5": ("arm_7dof", (200, 72), "relu", 2, 3, 5),       # widths off every tile size (16 / 64 / k-step 4)
    "hc_3x96_tanh_m4_b16": ("half_cheetah", (96, 96, 96), "tanh", 4, 4, 16),
}


def build(name):
    env_name, hidden, act, m, mbs, rows = CASES[name]
    env = SyntheticEnv(env_name)
    od, ad = env.observation_space.shape[0], env.action_space.shape[0]
    norm = synthetic.make_norm(od, ad, env.action_space.low, env.action_space.high, 2000)
    params = synthetic.make_weight_set(od, ad, list(hidden), 1000)
    rs = np.random.RandomState(len(name) * 131 + rows)
    obs = [rs.randn(rows, od) for _ in range(m)]
    act_ = [rs.uniform(env.action_space.low, env.action_space.high, (rows, ad)) for _ in range(m)]
    nxt = [o + 0.3 * rs.randn(rows, od) for o in obs]
    return dict(env=env, hidden=hidden, hidden_nonlinearity=act, meta_batch_size=mbs, inner_learning_rate=0.01,
                norm=norm, params=[np.asarray(p) for p in params], obs=obs, act=act_, obs_next=nxt)
