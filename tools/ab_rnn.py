#!/usr/bin/env python
"""Generic recurrent cells (GRU / BasicRNN / stacks) at the ReBAL default plan size: matrix-core kernel against the VALU one
(time, agreement of the returns) - developer aid, needs a GPU."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import cases  # noqa: E402
import bench_configs as bc  # noqa: E402
from learning_to_adapt_amd import _lib  # noqa: E402

ctx = _lib.Context.get(0)
base = cases.CASES["hc_rnn_rs_gru2_n48_h4"]
for label, over in (("gru 256", dict(cell_type="gru", hidden_sizes=[256])),
                    ("gru 2x128", dict(cell_type="gru", hidden_sizes=[128, 128])),
                    ("lstm 2x128", dict(cell_type="lstm", hidden_sizes=[128, 128])),
                    ("lstm 2x256", dict(cell_type="lstm", hidden_sizes=[256, 256])),
                    ("rnn 256", dict(cell_type="rnn", hidden_sizes=[256])),
                    ("gru 56+40 (golden shape)", dict(cell_type="gru", hidden_sizes=[56, 40]))):
    case = dict(base, n=500, h=10, m=5, **over)
    case["units"] = sum(case["hidden_sizes"])
    case.pop("reset_after", None)
    env, model = cases.product_rnn_model(case)
    native = model.planner_model()
    dev = native.device
    m, n, h, U = case["m"], case["n"], case["h"], case["units"]
    g = torch.Generator(device="cpu").manual_seed(1)
    obs0 = torch.randn((m, 20), generator=g).to(dev)
    c0 = torch.randn((m, U), generator=g).to(dev) * (1.0 if case["cell_type"] == "lstm" else 0.0)
    h0 = torch.tanh(torch.randn((m, U), generator=g)).to(dev)
    a = (torch.rand((h, m * n, 6), generator=g) * 2 - 1).to(dev)
    best = torch.zeros((m,), dtype=torch.int64, device=dev)
    out = {"model": label}
    rets = {}
    for kernel in ("mfma", "valu"):
        ctx.set_kernel(kernel)
        r = torch.empty((m, n), dtype=torch.float32, device=dev)
        native.plan_rs(obs0, c0, h0, a, m, n, h, 1.0, env.reward_spec, returns_out=r, best_key=best)
        torch.cuda.synchronize()
        rets[kernel] = r.cpu().numpy()
        out[kernel + "_ms"] = round(bc.time_launches(lambda: native.plan_rs(obs0, c0, h0, a, m, n, h, 1.0, env.reward_spec, best_key=best), 20), 4)
    ctx.set_kernel("auto")
    macs = 0
    kin = 26
    for u in case["hidden_sizes"]:
        macs += (kin + u) * u * {"lstm": 4, "gru": 3, "rnn": 1}[case["cell_type"]]
        kin = u
    macs += kin * 20
    out["mfma_frac_fp32_peak"] = round(2.0 * macs * n * m * h / out["mfma_ms"] / 1e9 / bc.PEAK, 4)
    out["speedup"] = round(out["valu_ms"] / out["mfma_ms"], 1)
    out["max_rel_diff_returns"] = float(np.max(np.abs(rets["mfma"] - rets["valu"])) / max(1.0, float(np.max(np.abs(rets["valu"])))))
    out["same_argmax"] = bool(np.array_equal(rets["mfma"].argmax(1), rets["valu"].argmax(1)))
    print(json.dumps(out), flush=True)
