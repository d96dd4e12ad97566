#!/usr/bin/env python
"""Kernel time of the recurrent rollout on the split (n = 2000) and the chip-filling (n = 4096) plan - developer A/B aid."""
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

out = {"lib": os.path.basename(os.environ.get("L2A_LIB_PATH", "libl2a_hip.so"))}
for n, h, m in ((2000, 30, 1), (4096, 30, 1), (500, 10, 5)):
    case = dict(cases.CASES["c6_hc_rnn_rs_n500_h10_m5"], n=n, h=h, m=m)
    env, model = cases.product_rnn_model(case)
    native = model.planner_model()
    dev = native.device
    U = case["units"]
    obs0 = torch.randn((m, 20), device=dev)
    c0 = torch.randn((m, U), device=dev)
    h0 = torch.tanh(torch.randn((m, U), device=dev))
    a = torch.rand((h, m * n, 6), device=dev) * 2 - 1
    best = torch.zeros((m,), dtype=torch.int64, device=dev)
    ms = min(bc.time_launches(lambda: native.plan_rs(obs0, c0, h0, a, m, n, h, 1.0, env.reward_spec, best_key=best), 30) for _ in range(2))
    fl = 2.0 * ((26 + U) * 4 * U + U * 20) * n * m * h
    out["lstm_n%d_m%d" % (n, m)] = {"ms": round(ms, 4), "frac": round(fl / ms / 1e9 / bc.PEAK, 4)}
print(json.dumps(out), flush=True)
