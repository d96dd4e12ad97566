#!/usr/bin/env python
"""Soak: a minute of randomly mixed launches (split / unsplit / tail-split MLP plans, blocking launches, recurrent plans
with and without the unit-tile split, the micro-tile kernels of all three families) - the status word must stay clean and every plan must return the
bits of its first run.  Developer aid, needs a GPU:  python tools/soak.py [seconds]"""
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

C = cases.CASES
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
plans = []
for name, over in [("c2_hc_rs_n2000_h30_e5", {}), ("c3_ant_rs_n2000_h20_pb5", dict(h=5)), ("c1_hc_rs_n500_h10_e1", {}),
                   ("c2_hc_rs_n2000_h30_e5", dict(n=4800, h=4)), ("hc_rs_m2_n100_h7_e2", {}),
                   ("c2_hc_rs_n2000_h30_e5", dict(h=1)), ("c3b_ant_rs_n500_h10_pb5_3x512", dict(h=4)),
                   ("ant_rs_n300_h6_e3", {}),                                           # (member fan: 2 envs x 19 tiles x 3)
                   ("c5_hc_cem_n4000_h30_e5", dict(n=500, h=8)),                        # member fan: one rank's config-5 shard
                   ("c5_hc_cem_n4000_h30_e5", dict(n=1000, h=4)),                       # member fan, two tiles per workgroup
                   ("c1_hc_rs_n500_h10_e1", dict(n=8192, h=2)),                         # double rounds: all on two-tile workgroups
                   ("c2_hc_rs_n2000_h30_e5", dict(E=1, mode="single", m=10, h=2))]:     # double rounds + a whole round (run_mb_mpc.py default)
    case = dict(C[name], **over)
    env, model = cases.product_model(case)
    nat = model.planner_model()
    dev = nat.device
    m, n, h = case["m"], case["n"], case["h"]
    od, ad = env.observation_space.shape[0], env.action_space.shape[0]
    obs = np.random.RandomState(len(plans)).randn(m, od).astype(np.float32)
    a = (torch.rand((h, m * n, ad), device=dev) * 2 - 1) * float(env.action_space.high[0])
    best = torch.zeros((m,), dtype=torch.int64, device=dev)
    rets = torch.empty((m, n), dtype=torch.float32, device=dev)
    obs_d = torch.from_numpy(obs).to(dev)

    def run(nat=nat, obs_d=obs_d, a=a, m=m, n=n, h=h, env=env, best=best, rets=rets):
        nat.plan_rs(obs_d, a, m, n, h, 1.0, env.reward_spec, returns_out=rets, best_key=best)
        return best, rets

    def run_sync(nat=nat, obs=obs, a=a, m=m, n=n, h=h, env=env):
        return nat.plan_rs_sync(obs, a, m, n, h, 1.0, env.reward_spec)
    k, r = run()
    torch.cuda.synchronize()
    plans.append(dict(run=run, run_sync=run_sync, key=k.clone(), rets=r.clone(), keep=(model, env)))
for over in (dict(n=2000, h=30, m=1), dict(), dict(n=4096, h=6, m=1),
             dict(cell_type="gru", hidden_sizes=[96, 40], units=136, n=300, h=5, m=3),          # generic matrix-core kernel
             dict(cell_type="lstm", hidden_sizes=[128, 64], units=192, n=700, h=4, m=2),
             # the generic micro-tile kernel (csrc/l2a_rnn_micro.h): three-tile workgroups, a GRU stack, several rounds of four-tile ones
             dict(cell_type="gru", hidden_sizes=[256], units=256, n=500, h=10, m=5),
             dict(cell_type="lstm", hidden_sizes=[256, 256], units=512, n=333, h=4, m=3),
             dict(cell_type="rnn", hidden_sizes=[256], units=256, n=5000, h=3, m=1)):
    case = dict(C["c6_hc_rnn_rs_n500_h10_m5"], **over)
    env, model = cases.product_rnn_model(case)
    nat = model.planner_model()
    dev = nat.device
    m, n, h, U = case["m"], case["n"], case["h"], case["units"]
    obs = np.random.RandomState(7).randn(m, 20).astype(np.float32)
    c0 = torch.randn((m, U), device=dev) * (0.0 if case.get("cell_type", "lstm") != "lstm" else 1.0)
    h0 = torch.tanh(torch.randn((m, U), device=dev))
    a = torch.rand((h, m * n, 6), device=dev) * 2 - 1
    best = torch.zeros((m,), dtype=torch.int64, device=dev)
    rets = torch.empty((m, n), dtype=torch.float32, device=dev)
    obs_d = torch.from_numpy(obs).to(dev)
    c1, h1 = torch.empty_like(c0), torch.empty_like(h0)

    def run(nat=nat, obs_d=obs_d, c0=c0, h0=h0, a=a, m=m, n=n, h=h, env=env, best=best, rets=rets):
        nat.plan_rs(obs_d, c0, h0, a, m, n, h, 1.0, env.reward_spec, returns_out=rets, best_key=best)
        return best, rets

    def run_sync(nat=nat, obs=obs, c0=c0, h0=h0, a=a, m=m, n=n, h=h, env=env, c1=c1, h1=h1):
        return nat.plan_rs_sync(obs, c0, h0, a, m, n, h, 1.0, env.reward_spec, c_next=c1, h_next=h1)
    k, r = run()
    torch.cuda.synchronize()
    plans.append(dict(run=run, run_sync=run_sync, key=k.clone(), rets=r.clone(), keep=(model, env)))
ctx = _lib.Context.get(0)
rs = np.random.RandomState(0)
t0 = time.time()
count = sync_count = 0
while time.time() - t0 < seconds:
    for _ in range(200):
        p = plans[rs.randint(len(plans))]
        if rs.rand() < 0.3:
            keys = p["run_sync"]()
            assert keys is not None and np.array_equal(keys.view(np.int64), p["key"].cpu().numpy())
            sync_count += 1
        else:
            p["run"]()
        count += 1
    torch.cuda.synchronize()
    ctx.launch_status()
    for p in plans:
        k, r = p["run"]()
        torch.cuda.synchronize()
        assert torch.equal(k, p["key"]) and torch.equal(r, p["rets"])
print("soak ok: %d mixed launches (%d of them blocking) over %d plan shapes in %.1f s, status word clean, every plan "
      "bit-stable" % (count, sync_count, len(plans), time.time() - t0))
