#!/usr/bin/env python
"""NT = 1 against NT = 2 (two candidate tiles per workgroup) on multi-round plans, shape by shape - run once per value of
L2A_FORCE_NT (the library reads it at the first launch): feeds the NT choice of launch_rollout (csrc/l2a_api.hip), in particular
for the instances whose NT = 2 form spills registers (wide observations at hidden width 512)."""
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

SHAPES = (("c3_ant_rs_n2000_h20_pb5", dict(n=8000, h=10)), ("c3_ant_rs_n2000_h20_pb5", dict(n=8000, h=10, hidden=[512, 512, 512])),
          ("ant_rs_n300_h6_e3", dict(n=16000, h=10, m=1)), ("c4_hc_rs_n16000_h30_e5", dict(h=10)),
          ("arm_rs_n256_h8", dict(n=16000, h=10)), ("c1_hc_rs_n500_h10_e1", dict(n=16000, h=10)),
          ("c2_hc_rs_n2000_h30_e5", dict(n=16000, h=10, hidden=[256, 256])))
for name, over in SHAPES:
    case = dict(cases.CASES[name], **over)
    env, model = cases.product_model(case)
    ms = bc.time_plan(model.planner_model(), case, env)
    print(json.dumps({"force_nt": os.environ.get("L2A_FORCE_NT", "auto"), "case": name, "over": over, "env": case["env"], "kernel_ms": round(ms, 4),
                      "frac_fp32_peak": round(bc.flops(case, env) / ms / 1e9 / bc.PEAK, 4)}), flush=True)
