#!/usr/bin/env python
"""Kernel-only timings of the recurrent plan shapes with whatever library L2A_LIB_PATH selects (A/B of LSTM kernel
variants built by tools/build_variant.py).  One JSON line."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402

PEAK = 157.3


def main():
    out = {"lib": os.path.basename(os.environ.get("L2A_LIB_PATH", "libl2a_hip.so"))}
    for tag, over in (("n2000_h30", {"n": 2000, "h": 30, "m": 1}), ("c6", {}), ("n4096_h30", {"n": 4096, "h": 30, "m": 1})):
        case = dict(cases.CASES["c6_hc_rnn_rs_n500_h10_m5"], **over)
        env, model = cases.product_rnn_model(case)
        native = model.planner_model()
        dev = native.device
        m, n, h, U = case["m"], case["n"], case["h"], case["units"]
        obs0 = torch.randn((m, 20), device=dev)
        c0 = torch.randn((m, U), device=dev)
        h0 = torch.tanh(torch.randn((m, U), device=dev))
        a = torch.rand((h, m * n, 6), device=dev) * 2 - 1
        best = torch.zeros((m,), dtype=torch.int64, device=dev)
        for _ in range(5):
            native.plan_rs(obs0, c0, h0, a, m, n, h, 1.0, env.reward_spec, best_key=best)
        torch.cuda.synchronize()
        best_ms = 1e9
        for _ in range(3):
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
            for s_, e_ in evs:
                s_.record()
                native.plan_rs(obs0, c0, h0, a, m, n, h, 1.0, env.reward_spec, best_key=best)
                e_.record()
            torch.cuda.synchronize()
            best_ms = min(best_ms, float(np.median([s_.elapsed_time(e_) for s_, e_ in evs])))
        fl = 2.0 * ((26 + U) * 4 * U + U * 20) * n * m * h
        out[tag] = {"ms": round(best_ms, 4), "frac": round(fl / best_ms / 1e9 / PEAK, 4)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
