#!/usr/bin/env python
"""Generic recurrent cells (GRU / BasicRNN / LSTM stacks of 256-unit layers): micro-tile kernel (csrc/l2a_rnn_micro.h) against the
16-candidate kernel (csrc/l2a_rnn_mfma.h) - time, fraction of the fp32 matrix peak, agreement of the returns.  Needs a GPU.
    python tools/ab_rnn_micro.py [n m h]"""
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

PLANS = [(500, 5, 10), (2000, 1, 30), (250, 5, 10), (1000, 1, 10), (4000, 1, 30), (2000, 5, 10)]
if len(sys.argv) > 3:
    PLANS = [tuple(int(v) for v in sys.argv[1:4])]
ctx = _lib.Context.get(0)
base = cases.CASES["hc_rnn_rs_gru2_n48_h4"]
for n, m, h in PLANS:
    for label, over in (("gru 256", dict(cell_type="gru", hidden_sizes=[256])),
                        ("lstm 2x256", dict(cell_type="lstm", hidden_sizes=[256, 256])),
                        ("rnn 256", dict(cell_type="rnn", hidden_sizes=[256])),
                        ("gru 2x256", dict(cell_type="gru", hidden_sizes=[256, 256])),
                        ("rnn 3x256", dict(cell_type="rnn", hidden_sizes=[256, 256, 256])),
                        ("gru 512", dict(cell_type="gru", hidden_sizes=[512])),
                        ("lstm 3x256", dict(cell_type="lstm", hidden_sizes=[256, 256, 256])),
                        # (the tuned kernels' own shape: l2a_lstm.h / l2a_lstm_micro_k - or, with L2A_FORCE_GENERIC=1, the generic ones)
                        ("lstm 256" + (" (generic kernels)" if os.environ.get("L2A_FORCE_GENERIC") else ""), dict(cell_type="lstm", hidden_sizes=[256]))):
        case = dict(base, n=n, h=h, m=m, **over)
        case["units"] = sum(case["hidden_sizes"])
        case.pop("reset_after", None)
        env, model = cases.product_rnn_model(case)
        native = model.planner_model()
        dev = native.device
        U = case["units"]
        g = torch.Generator(device="cpu").manual_seed(1)
        obs0 = torch.randn((m, 20), generator=g).to(dev)
        c0 = torch.randn((m, U), generator=g).to(dev) * (1.0 if case["cell_type"] == "lstm" else 0.0)
        h0 = torch.tanh(torch.randn((m, U), generator=g)).to(dev)
        a = (torch.rand((h, m * n, 6), generator=g) * 2 - 1).to(dev)
        best = torch.zeros((m,), dtype=torch.int64, device=dev)
        out = {"model": label, "n": n, "m": m, "h": h}
        rets = {}
        for name, policy in (("tiles16", 0), ("micro", 2)):
            ctx.set_micro(policy)
            r = torch.empty((m, n), dtype=torch.float32, device=dev)
            try:
                native.plan_rs(obs0, c0, h0, a, m, n, h, 1.0, env.reward_spec, returns_out=r, best_key=best)
            except _lib.L2AError as e:          # (two 512-unit layers: neither 16-candidate kernel has the LDS for them)
                out[name + "_error"] = str(e)[-80:]
                continue
            torch.cuda.synchronize()
            ctx.launch_status()
            rets[name] = r.cpu().numpy()
            out[name + "_ms"] = round(bc.time_launches(lambda: native.plan_rs(obs0, c0, h0, a, m, n, h, 1.0, env.reward_spec, best_key=best), 20), 4)
        ctx.set_micro(1)
        macs, kin = 0, 26
        for u in case["hidden_sizes"]:
            macs += (kin + u) * u * {"lstm": 4, "gru": 3, "rnn": 1}[case["cell_type"]]
            kin = u
        macs += kin * 20
        for name in rets:
            out[name + "_frac_fp32_peak"] = round(2.0 * macs * n * m * h / out[name + "_ms"] / 1e9 / bc.PEAK, 4)
        if len(rets) == 2:
            out["micro_over_tiles16"] = round(out["micro_ms"] / out["tiles16_ms"], 3)
            out["max_rel_diff_returns"] = float(np.max(np.abs(rets["micro"] - rets["tiles16"]) / np.maximum(1.0, np.abs(rets["tiles16"]))))
            out["same_argmax"] = bool(np.array_equal(rets["micro"].argmax(1), rets["tiles16"].argmax(1)))
        print(json.dumps(out), flush=True)
