#!/usr/bin/env python
"""Micro-tile kernels (csrc/l2a_micro.h) against the 16-candidate kernels on the plan sizes they are meant for: kernel time by
HIP events with the clocks up, policy by policy (l2a_set_micro 0 / 2) - developer A/B aid, feeds profiles/r04_ab_micro.jsonl."""
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

WHAT = sys.argv[1:] or ["lstm"]
ctx = None


def lstm_rows():
    global ctx
    for units, n, h, m in ((256, 500, 10, 5), (256, 2000, 30, 1), (256, 500, 10, 1), (256, 3000, 10, 1), (256, 1000, 10, 2), (256, 1000, 10, 1),
                           (256, 250, 10, 5), (512, 500, 10, 5)):
        case = dict(cases.CASES["c6_hc_rnn_rs_n500_h10_m5"], n=n, h=h, m=m, units=units)
        env, model = cases.product_rnn_model(case)
        native = model.planner_model()
        dev = native.device
        ctx = _lib.Context.get(0)
        U = units
        obs0 = torch.randn((m, 20), device=dev)
        c0 = torch.randn((m, U), device=dev)
        h0 = torch.tanh(torch.randn((m, U), device=dev))
        a = torch.rand((h, m * n, 6), device=dev) * 2 - 1
        best = torch.zeros((m,), dtype=torch.int64, device=dev)
        fl = 2.0 * ((26 + U) * 4 * U + U * 20) * n * m * h
        row = {"kernel": "lstm", "units": U, "n": n, "h": h, "m": m}
        for pol in (0, 2, 1, 0, 2, 1):
            ctx.set_micro(pol)
            ms = bc.time_launches(lambda: native.plan_rs(obs0, c0, h0, a, m, n, h, 1.0, env.reward_spec, best_key=best), 30)
            torch.cuda.synchronize()
            ctx.launch_status()
            key = ("tile16", "auto", "micro")[pol]
            if key + "_ms" in row:
                ms = min(ms, row[key + "_ms"])
            row[key + "_ms"] = round(ms, 4)
            row[key + "_frac"] = round(fl / ms / 1e9 / bc.PEAK, 4)
        ctx.set_micro(1)
        row["ratio"] = round(row["micro_ms"] / row["tile16_ms"], 3)
        print(json.dumps(row), flush=True)


def mlp_rows():
    import numpy as np
    shapes = (("c3b_ant_rs_n500_h10_pb5_3x512", {}), ("c1_hc_rs_n500_h10_e1", {}), ("c3b_ant_rs_n500_h10_pb5_3x512", dict(hidden=[512, 512])),
              ("c2_hc_rs_n2000_h30_e5", {}), ("c2_hc_rs_n2000_h30_e5", dict(n=2500, h=10)), ("c2_hc_rs_n2000_h30_e5", dict(n=500, h=10)),
              ("c2_hc_rs_n2000_h30_e5", dict(n=3000, h=10)), ("c1_hc_rs_n500_h10_e1", dict(n=2500, m=1)), ("c1_hc_rs_n500_h10_e1", dict(n=1000)),
              ("c3_ant_rs_n2000_h20_pb5", dict(n=600, h=10)), ("c1_hc_rs_n500_h10_e1", dict(n=1600)), ("c1_hc_rs_n500_h10_e1", dict(n=250, m=5)),
              ("c3b_ant_rs_n500_h10_pb5_3x512", dict(m=3, E=3)), ("c2_hc_rs_n2000_h30_e5", dict(hidden=[256, 256], n=2500, h=10)))
    for name, over in shapes:
        case = dict(cases.CASES[name], **over)
        env, model = cases.product_model(case)
        native = model.planner_model()
        dev = native.device
        ctx = _lib.Context.get(0)
        m, n, h = case["m"], case["n"], case["h"]
        od, ad = env.observation_space.shape[0], env.action_space.shape[0]
        obs0 = torch.randn((m, od), device=dev)
        a = torch.rand((h, m * n, ad), device=dev) * 2 - 1
        best = torch.zeros((m,), dtype=torch.int64, device=dev)
        dims = [od + ad] + list(case["hidden"]) + [od]
        mac = sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1))
        fl = 2.0 * mac * (case["E"] if case["mode"] == "mean" else 1) * n * m * h
        row = {"kernel": "mlp", "case": name, "hidden": case["hidden"], "mode": case["mode"], "E": case["E"], "n": n, "h": h, "m": m}
        for pol in (0, 2, 1, 0, 2, 1):
            ctx.set_micro(pol)
            ms = bc.time_launches(lambda: native.plan_rs(obs0, a, m, n, h, 1.0, env.reward_spec, best_key=best), 30)
            torch.cuda.synchronize()
            ctx.launch_status()
            key = ("tile16", "auto", "micro")[pol]
            if key + "_ms" in row:
                ms = min(ms, row[key + "_ms"])
            row[key + "_ms"] = round(ms, 4)
            row[key + "_frac"] = round(fl / ms / 1e9 / bc.PEAK, 4)
        ctx.set_micro(1)
        row["ratio"] = round(row["micro_ms"] / row["tile16_ms"], 3)
        print(json.dumps(row), flush=True)


if "lstm" in WHAT:
    lstm_rows()
if "mlp" in WHAT:
    mlp_rows()
