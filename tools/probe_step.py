#!/usr/bin/env python
"""Where every microsecond of a parity-mode controller step goes (round 5): the four default workloads of the reference's run
scripts through the drop-in classes, un-instrumented wall time per call beside the C controller's own stage table
(`l2a_controller_stats`: take | stage obs | launch | kick | wait | decode), the rollout kernel's duration by HIP events measured
in a separate pass, and the call-time distribution (p50 / p95 / p99 / p99.9).

    python tools/probe_step.py [c2|rebal|grbal|mbmpc ...] [--calls N] > profiles/r05_probe_steps.jsonl
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
from learning_to_adapt_amd.envs import SyntheticEnv  # noqa: E402
cases.SyntheticEnv = SyntheticEnv


def build(which):
    C = cases.CASES
    pre = None
    if which == "c2":
        case = C["c2_hc_rs_n2000_h30_e5"]
        env, model = cases.product_model(case)
        ctrl = cases.product_controller(case, model=model, env=env)
        obs = np.random.RandomState(1).randn(case["m"], 20)
        label = "config 2 (n=2000, h=30, E=5 mean, m=1)"
    elif which == "mbmpc":
        case = dict(C["c1_hc_rs_n500_h10_e1"], n=2000, h=20, m=10)
        env, model = cases.product_model(case)
        ctrl = cases.product_controller(case, model=model, env=env)
        obs = np.random.RandomState(3).randn(10, 20)
        label = "run_mb_mpc.py default (one 2x512 model, n=2000, h=20, m=10)"
    elif which == "rebal":
        case = C["c6_hc_rnn_rs_n500_h10_m5"]
        ctrl = cases.product_rnn_controller(case)
        model = ctrl.dynamics_model
        obs = np.random.RandomState(0).randn(5, 20)
        ctrl.reset(dones=[True] * 5)
        label = "run_rebal.py default (LSTM 256, n=500, h=10, m=5): plan + state advance"
    elif which == "grbal":
        from learning_to_adapt_amd.dynamics import MetaMLPDynamicsModel
        from learning_to_adapt_amd.policies import MPCController
        from learning_to_adapt_amd.utils import synthetic
        env = SyntheticEnv("ant")
        model = MetaMLPDynamicsModel(name="dyn", env=env, hidden_sizes=(512, 512, 512), inner_learning_rate=0.01, init_seed=0)
        model.set_normalization(synthetic.make_norm(41, 8, env.action_space.low, env.action_space.high, 2000))
        rs = np.random.RandomState(0)
        ob = [rs.randn(16, 41) for _ in range(5)]
        ac = [rs.uniform(-150, 150, (16, 8)) for _ in range(5)]
        nx = [o + 0.1 * rs.randn(16, 41) for o in ob]
        ctrl = MPCController(name="p", env=env, dynamics_model=model, n_candidates=500, horizon=10)
        obs = rs.randn(5, 41)

        def pre():      # samplers/sampler.py:81-91: adapt, then get_actions
            model.switch_to_pre_adapt()
            model.adapt(ob, ac, nx)
        label = "run_grbal.py default (3x512, n=500, h=10, m=5): adapt + plan"
    else:
        raise SystemExit("unknown workload " + which)
    return label, ctrl, obs, pre


def pct(a, qs=(5, 50, 95, 99, 99.9)):
    return {("p%g" % q): round(float(np.percentile(a, q)), 1) for q in qs}


def run(which, calls):
    label, ctrl, obs, pre = build(which)
    np.random.seed(0)
    for _ in range(30):
        if pre:
            pre()
        ctrl.get_actions(obs)
    torch.cuda.synchronize()
    st = ctrl._cstep
    # pass 1: un-instrumented calls (the number), the C stage table read after each call (outside the timed span)
    tot, pre_us, stages = [], [], []
    for _ in range(calls):
        t0 = time.perf_counter()
        if pre:
            pre()
        t1 = time.perf_counter()
        ctrl.get_actions(obs)
        t2 = time.perf_counter()
        tot.append(1e6 * (t2 - t0))
        pre_us.append(1e6 * (t1 - t0))
        if st is not None:
            stages.append(list(st.stats()["stage_us"].values()))
    tot = np.array(tot)
    row = dict(workload=label, calls=calls, native_step=st is not None, ms_per_call=round(float(np.mean(tot)) / 1e3, 4),
               call_us=pct(tot), over_1p15_p50=int(np.sum(tot > 1.15 * np.median(tot))))
    if pre:
        row["adapt_call_us"] = pct(np.array(pre_us), (50, 99))
    if stages:
        S = np.array(stages)
        names = list(st.stats()["stage_us"].keys())
        row["c_stage_us_median"] = {k: round(float(np.median(S[:, i])), 1) for i, k in enumerate(names)}
        row["c_stage_us_p99"] = {k: round(float(np.percentile(S[:, i], 99)), 1) for i, k in enumerate(names)}
        row["python_outside_c_call_us_median"] = round(float(np.median(tot - np.array(pre_us) - S[:, names.index("call")])), 1)
        s = st.stats()
        row["chain"] = {k: s[k] for k in ("hits", "misses", "produced", "producer_us_per_block", "consumer_wait_us_per_take")}
        # every call slower than 1.15 x the median, attributed to the stage with the largest excess over its own median
        med = np.median(S, axis=0)
        blame = {}
        for i in np.nonzero(tot > 1.15 * np.median(tot))[0]:
            ex = S[i, :6] - med[:6]
            j = int(np.argmax(ex))
            outside = tot[i] - pre_us[i] - S[i, names.index("call")]
            k = names[j] if ex[j] >= outside else "python_outside_c_call"
            blame[k] = blame.get(k, 0) + 1
        row["slow_calls_by_stage"] = blame
        slow = np.argsort(tot)[-5:]
        row["slowest_calls"] = [dict(call_us=round(float(tot[i]), 1), **{k: round(float(S[i, j]), 1) for j, k in enumerate(names)})
                                for i in slow]
    # pass 2: GPU time of a step by events around the call (adds two event records to the stream; not the number above)
    ev = []
    for _ in range(min(calls, 300)):
        if pre:
            pre()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        ctrl.get_actions(obs)
        e.record()
        ev.append((s, e))
    torch.cuda.synchronize()
    g = np.array([1e3 * s.elapsed_time(e) for s, e in ev])
    row["gpu_us_between_events_around_get_actions"] = pct(g, (50, 95, 99))
    print(json.dumps(row), flush=True)
    if st is not None:
        st.close()
        ctrl._cstep = None
    if getattr(ctrl, "_ahead", None) is not None:
        ctrl._ahead.stop()


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    calls = 1000
    for a in sys.argv[1:]:
        if a.startswith("--calls="):
            calls = int(a.split("=")[1])
        if a.startswith("--switch="):       # CPython's GIL switch interval (seconds; default 0.005) - for the Python path's tail
            sys.setswitchinterval(float(a.split("=")[1]))
    for w in (args or ["c2", "rebal", "grbal", "mbmpc"]):
        run(w, calls)
