#!/usr/bin/env python
"""What the REST launch of a double-round plan costs under the geometries it could take (developer A/B, one JSON line per run):
config 3's rest (5 envs x 368 candidates = 5 x 23 tiles) and run_mb_mpc.py default's (10 x 368), as plans of their own.
L2A_MICRO=0 / 1 in the environment selects tile split / micro tiles."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import cases  # noqa: E402
import bench_configs as bc  # noqa: E402

out = {"micro": os.environ.get("L2A_MICRO", "auto")}
for tag, name, over in (("c3_rest", "c3_ant_rs_n2000_h20_pb5", dict(n=368)),
                        ("mbmpc_rest", "c2_hc_rs_n2000_h30_e5", dict(E=1, mode="single", m=10, n=368, h=20))):
    case = dict(cases.CASES[name], **over)
    env, model = cases.product_model(case)
    ms = min(bc.time_plan(model.planner_model(), case, env, reps=40) for _ in range(3))
    out[tag] = round(ms, 4)
print(json.dumps(out), flush=True)
