#!/usr/bin/env python
"""Where a parity-mode controller step of the PYTHON path spends its host time (config 2; CEM: config 5): wraps the controller's
stages with timers.  Since round 5 random shooting on one GPU is one C call (tools/probe_step.py has its stage table); this probe
forces the Python path (L2A_NATIVE_STEP=0), which CEM, sharded plans and the fallback still take."""
import json
import os
os.environ.setdefault("L2A_NATIVE_STEP", "0")
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402


def main():
    case = cases.CASES[sys.argv[1] if len(sys.argv) > 1 else "c2_hc_rs_n2000_h30_e5"]
    if os.environ.get("L2A_PROBE_OVER"):        # e.g. '{"n": 2000, "h": 20, "m": 10}' on c1: run_mb_mpc.py's own defaults
        case = dict(case, **json.loads(os.environ["L2A_PROBE_OVER"]))
    rnn = case["planner"].startswith("rnn")
    env, model = cases.product_rnn_model(case) if rnn else cases.product_model(case)
    obs = np.random.RandomState(1).randn(case["m"], env.observation_space.shape[0])
    for mode, ahead in (("numpy", True), ("numpy", False), ("device", True)):
        if rnn:
            ctrl = cases.product_rnn_controller(case, model=model, env=env, rng=mode)
            ctrl.draw_ahead = ahead
            ctrl.reset(dones=[True] * case["m"])
        else:
            ctrl = cases.product_controller(case, model=model, env=env, rng=mode, draw_ahead=ahead)
        acc = {}

        def wrap(obj, name, label):
            fn = getattr(obj, name)

            def timed(*a, _acc=acc, **k):
                t = time.perf_counter()
                r = fn(*a, **k)
                _acc[label] = _acc.get(label, 0.0) + time.perf_counter() - t
                return r
            setattr(obj, name, timed)
        np.random.seed(0)
        for _ in range(5):
            ctrl.get_actions(obs)
        native = model.planner_model()
        if hasattr(native, "plan_rs_sync"):
            wrap(native, "plan_rs_sync", "plan_rs_sync (launch + wait)")
        if rnn:
            wrap(ctrl, "_advance_hidden", "_advance_hidden")      # (wrapping `_rollout` would switch the blocking launch off)
            wrap(ctrl, "_combine_keys", "_combine_keys (sync + read-back)")
        wrap(ctrl, "_plan_pipelined", "_plan_pipelined")
        wrap(ctrl, "_draw_rows", "_draw_rows")
        if ctrl._ahead is not None:
            wrap(ctrl._ahead, "take", "ahead.take")
            wrap(ctrl._ahead, "start", "ahead.start")
            wrap(ctrl._ahead, "active_for", "ahead.active_for")
        wrap(ctrl, "_rs_parity_plan", "_rs_parity_plan (total)")
        wrap(ctrl, "_cem_draw", "_cem_draw")
        wrap(ctrl, "_cem_rollout_pipelined", "_cem_rollout_pipelined")
        wrap(ctrl, "_cem_refit", "_cem_refit")
        wrap(ctrl, "_cem_iteration", "_cem_iteration (total)")
        wrap(ctrl, "_upload_obs", "_upload_obs")
        wrap(ctrl, "_to_device", "_to_device")
        from learning_to_adapt_amd.utils import fast_rng
        if not hasattr(fast_rng, "_orig_cem_samples"):
            fast_rng._orig_cem_samples = fast_rng.cem_samples
        _cs = fast_rng._orig_cem_samples

        def timed_cs(*a, _acc=acc, **k):
            t = time.perf_counter()
            r = _cs(*a, **k)
            _acc["cem_samples"] = _acc.get("cem_samples", 0.0) + time.perf_counter() - t
            return r
        fast_rng.cem_samples = timed_cs
        torch.cuda.synchronize()
        K = 200 if case["planner"].endswith("rs") else 20
        t0 = time.perf_counter()
        for _ in range(K):
            ctrl.get_actions(obs)
        torch.cuda.synchronize()
        tot = time.perf_counter() - t0
        row = {"mode": mode, "draw_ahead": ahead, "ms_per_call": round(1e3 * tot / K, 4),
               "stages_us": {k: round(1e6 * v / K, 1) for k, v in acc.items()},
               "hits": None if ctrl._ahead is None else ctrl._ahead.hits,
               "worker_ms_per_block": None if (ctrl._ahead is None or not ctrl._ahead.produced) else
               round(1e3 * ctrl._ahead.produce_s / ctrl._ahead.produced, 3)}
        print(json.dumps(row), flush=True)
        if ctrl._ahead is not None:
            ctrl._ahead.stop()


def ab(attr, values, case_name="c2_hc_rs_n2000_h30_e5", rounds=4, calls=200):
    """Same-process A/B of a controller attribute: alternating blocks of un-instrumented get_actions calls."""
    case = cases.CASES[case_name]
    env, model = cases.product_model(case)
    obs = np.random.RandomState(1).randn(case["m"], env.observation_space.shape[0])
    ctrl = cases.product_controller(case, model=model, env=env, rng="numpy", draw_ahead=True)
    np.random.seed(0)
    for _ in range(5):
        ctrl.get_actions(obs)
    out = {str(v): [] for v in values}
    for _ in range(rounds):
        for v in values:
            setattr(ctrl, attr, v)
            for _ in range(5):
                ctrl.get_actions(obs)
            t0 = time.perf_counter()
            for _ in range(calls):
                ctrl.get_actions(obs)
            out[str(v)].append(round(1e3 * (time.perf_counter() - t0) / calls, 4))
    print(json.dumps({"ab": attr, "case": case_name, "ms_per_call": out}), flush=True)
    ctrl._ahead.stop()


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--ab":
        if len(sys.argv) > 3:       # --ab <attr> <case> <value> <value> ...
            ab(sys.argv[2], [int(v) for v in sys.argv[4:]], case_name=sys.argv[3], rounds=5, calls=20)
        else:
            ab(sys.argv[2], [True, False])
    else:
        main()
