#!/usr/bin/env python
"""What a multi-round plan's last round costs: config 3 whole, its two full rounds alone, and its left-over candidates as one
micro-tile launch (policy 2) - developer aid that priced the hybrid tail."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import cases  # noqa: E402
import bench_configs as bc  # noqa: E402
from learning_to_adapt_amd import _lib  # noqa: E402

NAME = "c3_ant_rs_n2000_h20_pb5"
for label, over, pol in (("whole", {}, 1), ("two_rounds", dict(n=1632), 1), ("tail_micro", dict(n=1808, m=1), 2), ("tail_split", dict(n=1808, m=1), 0),
                         ("whole", {}, 1), ("two_rounds", dict(n=1632), 1), ("tail_micro", dict(n=1808, m=1), 2)):
    case = dict(cases.CASES[NAME], **over)
    env, model = cases.product_model(case)
    native = model.planner_model()
    dev = native.device
    ctx = _lib.Context.get(0)
    m, n, h = case["m"], case["n"], case["h"]
    od, ad = env.observation_space.shape[0], env.action_space.shape[0]
    obs0 = torch.randn((m, od), device=dev)
    a = torch.rand((h, m * n, ad), device=dev) * 2 - 1
    best = torch.zeros((m,), dtype=torch.int64, device=dev)
    ctx.set_micro(pol)
    ms = min(bc.time_launches(lambda: native.plan_rs(obs0, a, m, n, h, 1.0, env.reward_spec, best_key=best), 30) for _ in range(3))
    torch.cuda.synchronize()
    ctx.launch_status()
    ctx.set_micro(1)
    print(json.dumps({"leg": label, "m": m, "n": n, "h": h, "policy": pol, "ms": round(ms, 4)}), flush=True)
