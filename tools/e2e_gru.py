import sys, os, time, json
import numpy as np, torch
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
import cases
base = cases.CASES["hc_rnn_rs_gru2_n48_h4"]
case = dict(base, n=500, h=10, m=5, cell_type="gru", hidden_sizes=[256], units=256)
case.pop("reset_after", None)
for mode in ("numpy", "device"):
    ctrl = cases.product_rnn_controller(case, rng=mode)
    obs = np.random.RandomState(0).randn(5, 20)
    ctrl.reset(dones=[True]*5)
    np.random.seed(0)
    for _ in range(20): ctrl.get_actions(obs)
    torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(300): ctrl.get_actions(obs)
    torch.cuda.synchronize()
    print(json.dumps({"gru256_controller_step_ms": round(1e3*(time.perf_counter()-t0)/300,4), "rng": mode}))
