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

# ---- CEM: every iteration teacher-forced from the reference's mean / std (what test_cem_iterations_teacher_forced asserts):
#      error of the returns, rank swaps against the reference's ranking, elite-mask flips (reference :101), and the
#      largest gap - in the REFERENCE's returns - between the two candidates of a swapped rank (its witness pair) --------
import cem_ties  # noqa: E402

print("\n%-30s %3s %10s %10s %6s %6s %12s %12s" % ("CEM case (teacher-forced)", "it", "max rel", "max abs", "swaps", "flips",
                                                   "worst gap", "2 x max abs"))
for cid in cases.case_ids(planner="cem"):
    case, seed = cases.split_id(cid)
    gold = cases.load_golden(cid)
    ctrl = cases.product_controller(case)
    env = ctrl.env
    m, n, h = case["m"], case["n"], case["h"]
    act_dim = env.action_space.shape[0]
    k = max(int(n * 0.1), 1)
    clip_low = np.concatenate([env.action_space.low] * h)
    clip_high = np.concatenate([env.action_space.high] * h)
    np.random.seed(seed)
    mean, std = np.zeros((m, h * act_dim)), np.ones((m, h * act_dim))
    for it in range(case["num_cem_iters"]):
        _, _, got, _ = ctrl._cem_iteration(gold["obs0"], mean, std, k, clip_low, clip_high, 0, n, 1)
        want = gold["cem_returns"][it]
        flips = cem_ties.rank_flips(got, want, k)
        gaps = [f[4] for f in flips]
        print("%-30s %3d %10.2e %10.2e %6d %6d %12.3e %12.3e"
              % (cid, it, np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want))), np.max(np.abs(got - want)),
                 len(flips), sum(1 for f in flips if f[6]), max(gaps) if gaps else 0.0, 2 * np.max(np.abs(got - want))))
        mean, std = gold["cem_mean"][it], gold["cem_std"][it]
    # end to end (no teacher forcing): where does the first elite-mask flip happen, if any
    ctrl2 = cases.product_controller(case)
    np.random.seed(seed)
    ctrl2.get_actions(gold["obs0"])
    first = None
    for it, tr in enumerate(ctrl2.last_plan["cem_trace"]):
        fl = cem_ties.rank_flips(tr["returns"], gold["cem_returns"][it], k)
        if any(f[6] for f in fl):
            first = (it, sum(1 for f in fl if f[6]), max(f[4] for f in fl if f[6]))
            break
    print("%-30s end to end: %s" % (cid, "no elite-mask flip, bit-equal plan expected" if first is None else
                                    "first mask flip at iteration %d (%d positions, widest witness gap %.3e)" % first))

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
