#!/usr/bin/env python
"""Worst-case error of the HIP path against the golden vectors, per case (needs a GPU).
Feeds the numbers quoted in DESIGN.md section 2."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
from oracle.planner import sample_rs_actions  # noqa: E402

print("%-36s %10s %10s %8s %s" % ("case", "max rel", "max abs", "argmax", "top-2 margin / |ret|"))
for cid in cases.case_ids(planner="rs"):
    case, seed = cases.split_id(cid)
    gold = cases.load_golden(cid)
    env, model = cases.product_model(case)
    native = model.planner_model()
    np.random.seed(seed)
    a = sample_rs_actions(env.action_space.low, env.action_space.high, case["n"], case["m"], case["h"])
    dev = native.device
    rets = torch.empty((case["m"], case["n"]), dtype=torch.float32, device=dev)
    best = torch.zeros((case["m"],), dtype=torch.int64, device=dev)
    native.plan_rs(torch.from_numpy(gold["obs0"].astype(np.float32)).to(dev), torch.from_numpy(a.astype(np.float32)).to(dev),
                   case["m"], case["n"], case["h"], case.get("discount", 1.0), env.reward_spec, returns_out=rets, best_key=best)
    got = rets.cpu().numpy().astype(np.float64)
    want = gold["returns"]
    rel = np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want)))
    ok = np.array_equal(np.argmax(got, axis=1), gold["best"])
    marg = np.min(gold["margin"] / np.maximum(1.0, np.abs(want[np.arange(case["m"]), gold["best"]])))
    print("%-36s %10.2e %10.2e %8s %.2e" % (cid, rel, np.max(np.abs(got - want)), "equal" if ok else "DIFF", marg))

# ---- recurrent planner: the controller replayed over the recorded steps (first step of every case: the
#      candidate draw is the reference's, so every candidate's return can be compared) ----------------------
from oracle.rnn_planner import rnn_rollout_returns  # noqa: E402
from oracle import LSTMStateTuple, make_reward  # noqa: E402

print("\n%-36s %10s %10s %8s %s" % ("recurrent case (step)", "max rel", "max abs", "argmax", "hidden-state max abs err"))
for cid in cases.rnn_case_ids():
    case, seed = cases.split_id(cid)
    if case["planner"] != "rnn_rs":
        continue
    gold = cases.load_golden(cid)
    env, model = cases.product_rnn_model(case)
    native = model.planner_model()
    dev = native.device
    ctrl = cases.product_rnn_controller(case, model=model, env=env)
    ctrl.reset(dones=[True] * case["m"])
    np.random.seed(seed)
    resets = {int(k): v for k, v in case.get("reset_after", {}).items()}
    for k in range(case["steps"]):
        c_prev, h_prev = (np.array(x) for x in model.pack_hidden(ctrl._hidden_state))       # flat [m, sum(units)]
        state = np.random.get_state()
        a = sample_rs_actions(env.action_space.low, env.action_space.high, case["n"], case["m"], case["h"])
        np.random.set_state(state)
        up = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev)  # noqa: E731
        rets = torch.empty((case["m"], case["n"]), dtype=torch.float32, device=dev)
        best = torch.zeros((case["m"],), dtype=torch.int64, device=dev)
        native.plan_rs(up(gold["obs"][k]), up(c_prev), up(h_prev), up(a), case["m"], case["n"], case["h"],
                       case.get("discount", 1.0), env.reward_spec, returns_out=rets, best_key=best)
        got = rets.cpu().numpy().astype(np.float64)
        want = gold["returns_%d" % k]
        rel = np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want)))
        ok = np.array_equal(np.argmax(got, axis=1), gold["best_%d" % k])
        ctrl.get_actions(gold["obs"][k])
        c_now, h_now = model.pack_hidden(ctrl._hidden_state)
        herr = max(float(np.max(np.abs(c_now - gold["hidden_c_%d" % k]))),
                   float(np.max(np.abs(h_now - gold["hidden_h_%d" % k]))))
        print("%-36s %10.2e %10.2e %8s %.2e" % ("%s (%d)" % (cid, k), rel, np.max(np.abs(got - want)),
                                               "equal" if ok else "DIFF", herr))
        if k in resets:
            ctrl.reset(dones=np.array(resets[k], dtype=bool))
