#!/usr/bin/env python
"""Kernel time against the horizon for one plan shape (developer aid): the slope is the cost of a horizon step, the
intercept what a launch pays once (launch, per-workgroup prologue: constants to LDS, first operand fetches; epilogue).

    [L2A_LIB_PATH=...] python tools/probe_horizon.py [case] [h ...]
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import cases  # noqa: E402
import bench_configs as bc  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c3b_ant_rs_n500_h10_pb5_3x512"
hs = [int(x) for x in sys.argv[2:]] or [1, 2, 5, 10, 20, 40]
base = cases.CASES[name]
env, model = cases.product_model(base)
native = model.planner_model()
ms = []
for h in hs:
    case = dict(base, h=h)
    ms.append(min(bc.time_plan(native, case, env, reps=40) for _ in range(3)))
slope, icpt = np.polyfit(np.array(hs, dtype=np.float64), np.array(ms), 1)
print(json.dumps({"lib": os.path.basename(os.environ.get("L2A_LIB_PATH", "libl2a_hip.so")), "case": name,
                  "h": hs, "ms": [round(x, 4) for x in ms], "us_per_step": round(1e3 * slope, 3),
                  "us_per_launch": round(1e3 * icpt, 2)}), flush=True)
