import os, sys, time, json
import numpy as np, torch
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
case = cases.CASES["c2_hc_rs_n2000_h30_e5"]
env, model = cases.product_model(case)
gold = cases.load_golden("c2_hc_rs_n2000_h30_e5_s0")
obs = np.array(gold["obs0"])
for rep in range(2):
    ctrl = cases.product_controller(case, model=model, env=env, rng="numpy")
    np.random.seed(0)
    ts = []
    for i in range(60):
        t0 = time.perf_counter(); ctrl.get_actions(obs); ts.append((time.perf_counter() - t0) * 1e3)
    print("controller %d: per-call ms:" % rep, " ".join("%.2f" % t for t in ts[:40]))
    print("   mean calls 6..25: %.4f   mean calls 40..59: %.4f   hits %s" % (np.mean(ts[6:26]), np.mean(ts[40:]), getattr(ctrl._ahead, "hits", None)))
    if ctrl._ahead is not None: ctrl._ahead.stop()
