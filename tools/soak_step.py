#!/usr/bin/env python
"""Soak of the controller step as one C call (developer aid, GPU): three planners - config-1 random shooting, the run_grbal.py default
on per-block adapted sets (adapt before every plan), the run_rebal.py default LSTM - stepped in runs of 25; EVERY step is run twice
from the same NumPy generator state, through `l2a_controller_step` and through the Python path, and must give the same action,
index, return, hidden state and generator state afterwards.  Random foreign draws between steps exercise miss -> synchronous
draw -> re-arm -> back-off.

    python tools/soak_step.py [steps per planner, default 3000] > profiles/rNN_soak_step.txt
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
from learning_to_adapt_amd import _lib  # noqa: E402
from learning_to_adapt_amd.envs import SyntheticEnv  # noqa: E402
cases.SyntheticEnv = SyntheticEnv

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
rs = np.random.RandomState(123)


def pair(make):
    a, b = make(), make()
    b.native_step = False
    return a, b


planners = []
case = cases.CASES["c1_hc_rs_n500_h10_e1"]
env1, model1 = cases.product_model(case)
planners.append(("c1 random shooting", pair(lambda: cases.product_controller(case, model=model1, env=env1)), 1, 20, None))

from learning_to_adapt_amd.dynamics import MetaMLPDynamicsModel  # noqa: E402
from learning_to_adapt_amd.policies import MPCController  # noqa: E402
from learning_to_adapt_amd.utils import synthetic  # noqa: E402
enva = SyntheticEnv("ant")
gm = MetaMLPDynamicsModel(name="dyn", env=enva, hidden_sizes=(512, 512, 512), inner_learning_rate=0.01, init_seed=0)
gm.set_normalization(synthetic.make_norm(41, 8, enva.action_space.low, enva.action_space.high, 2000))


def adapt():
    ob = [rs.randn(16, 41) for _ in range(5)]
    ac = [rs.uniform(-150, 150, (16, 8)) for _ in range(5)]
    nx = [o + 0.1 * rs.randn(16, 41) for o in ob]
    gm.switch_to_pre_adapt()
    gm.adapt(ob, ac, nx)


planners.append(("run_grbal.py default (adapt + plan)", pair(lambda: MPCController(name="p", env=enva, dynamics_model=gm, n_candidates=500, horizon=10)), 5, 41, adapt))

case6 = cases.CASES["c6_hc_rnn_rs_n500_h10_m5"]
env6, model6 = cases.product_rnn_model(case6)


def make_rnn():
    c = cases.product_rnn_controller(case6, model=model6, env=env6)
    c.reset(dones=[True] * 5)
    return c


planners.append(("run_rebal.py default (LSTM 256)", pair(make_rnn), 5, 20, None))

t0 = time.time()
mism = 0
np.random.seed(7)
RUN = 25            # consecutive steps of one planner (the others consume the shared global generator in between: a stale block)
for step, (name, (nat, py), m, od, pre) in ((r * RUN + k, pl) for r in range(N // RUN) for pl in planners for k in range(RUN)):
    if True:
        obs = rs.randn(m, od)
        if pre:
            pre()
        if rs.rand() < 0.05:
            np.random.uniform(size=int(rs.randint(1, 5)))          # a foreign consumer of the global generator
        st = np.random.get_state()
        a1, _ = nat.get_actions(obs)
        s1 = np.random.get_state()
        p1 = (np.array(nat.last_plan["best_index"]), np.array(nat.last_plan["best_return"]))
        np.random.set_state(st)
        a2, _ = py.get_actions(obs)
        s2 = np.random.get_state()
        p2 = (np.array(py.last_plan["best_index"]), np.array(py.last_plan["best_return"]))
        ok = np.array_equal(a1, a2) and np.array_equal(p1[0], p2[0]) and np.array_equal(p1[1], p2[1]) and s1[2] == s2[2] and np.array_equal(s1[1], s2[1])
        if hasattr(nat, "_pack"):
            c1, h1 = nat._pack(nat._hidden_state)
            c2, h2 = py._pack(py._hidden_state)
            ok = ok and np.array_equal(c1, c2) and np.array_equal(h1, h2)
            if step % 97 == 96:
                d = [bool(rs.rand() < 0.3) for _ in range(m)]
                nat.reset(dones=d)
                py.reset(dones=d)
        if not ok:
            mism += 1
            print("MISMATCH", name, step, flush=True)
torch.cuda.synchronize()
st = _lib.Context.get(0).launch_status_value()
print("soak: %d steps x %d planners in %.1f s, mismatches %d, launch status word %d" % (N, len(planners), time.time() - t0, mism, st))
for name, (nat, py), *_ in planners:
    s = nat._cstep.stats()
    print("  %-40s C steps %d  hits %d  stale blocks %d  synchronous draws %d  blocks produced %d  relaunches %d" %
          (name, s["steps"], s["hits"], s["misses"], s["sync_draws"], s["produced"], s["relaunches"]))
    nat._cstep.close()
    nat._cstep = None
    if py._ahead is not None:
        py._ahead.stop()
assert mism == 0 and st == 0
