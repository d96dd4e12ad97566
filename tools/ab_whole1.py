"""Whole single tiles of one set per candidate: whole-tiles-only instances (L2A_DOUBLE=1, default) against the general ones (L2A_DOUBLE=0)
on plans of under two rounds of tiles, plus config 5's iteration (an ensemble: general instance either way) and run_mb_mpc.py's default."""
import json, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import cases
import bench_configs as bc
out = {"double_policy": os.environ.get("L2A_DOUBLE", "1")}      # (plans of under two rounds: the switch only picks the instance)
for tag, name, over in (("c5_iter", "c5_hc_cem_n4000_h30_e5", dict(planner="rs")), ("mbmpc", "c2_hc_rs_n2000_h30_e5", dict(E=1, mode="single", m=10, n=2000, h=20)),
                        ("c1_4096", "c1_hc_rs_n500_h10_e1", dict(n=4096)), ("ant_3000", "c3_ant_rs_n2000_h20_pb5", dict(n=800, h=10))):
    case = dict(cases.CASES[name], **over)
    env, model = cases.product_model(case)
    ms = min(bc.time_plan(model.planner_model(), case, env, reps=30) for _ in range(3))
    out[tag] = round(ms, 4)
print(json.dumps(out), flush=True)
