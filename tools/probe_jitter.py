#!/usr/bin/env python
"""Per-call distribution of a parity-mode controller step (config 2): total, blocking launch, everything else."""
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

case = cases.CASES["c2_hc_rs_n2000_h30_e5"]
env, model = cases.product_model(case)
obs = np.random.RandomState(1).randn(1, 20)
ctrl = cases.product_controller(case, model=model, env=env, rng="numpy", draw_ahead=True)
native = model.planner_model()
sync_t = []
orig = native.plan_rs_sync


def timed(*a, **k):
    t = time.perf_counter()
    r = orig(*a, **k)
    sync_t.append(time.perf_counter() - t)
    return r


native.plan_rs_sync = timed
stages = {}


def wrap(obj, name, label):
    fn = getattr(obj, name)

    def w(*a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        stages.setdefault(label, []).append(time.perf_counter() - t)
        return r
    setattr(obj, name, w)


if os.environ.get("STAGES"):
    chain = ctrl._ahead_chain()
    wrap(chain, "take", "take")
    wrap(chain, "active_for", "active_for")
    wrap(ctrl, "_rs_parity_plan", "parity_plan")
    wrap(ctrl, "_combine_keys", "combine")
    wrap(ctrl, "_plan_keys", "plan_keys")
    _we = torch.cuda.Stream.wait_event

    def we(self, ev):
        t = time.perf_counter()
        r = _we(self, ev)
        stages.setdefault("wait_event", []).append(time.perf_counter() - t)
        return r
    torch.cuda.Stream.wait_event = we
np.random.seed(0)
for _ in range(30):
    ctrl.get_actions(obs)
sync_t.clear()
for v in stages.values():
    v.clear()
if os.environ.get("PIN"):       # keep the calling thread on the core it is on (threads created earlier are unaffected)
    import ctypes
    os.sched_setaffinity(0, {ctypes.CDLL(None).sched_getcpu()})
if os.environ.get("NOGC"):
    import gc
    gc.collect()
    gc.disable()
tot = []
for _ in range(1000):
    t = time.perf_counter()
    ctrl.get_actions(obs)
    tot.append(time.perf_counter() - t)
tot, sy = 1e6 * np.array(tot), 1e6 * np.array(sync_t)
rest = tot - sy
pc = lambda a: [round(float(np.percentile(a, q)), 1) for q in (5, 25, 50, 75, 95, 99)]  # noqa: E731
print(json.dumps({"pinned": bool(os.environ.get("PIN")), "gc_disabled": bool(os.environ.get("NOGC")), "calls_over_2000us": int((tot > 2000).sum()),
                  "percentiles": [5, 25, 50, 75, 95, 99], "total_us": pc(tot), "plan_rs_sync_us": pc(sy), "python_rest_us": pc(rest),
                  "mean_total_us": round(float(tot.mean()), 1),
                  "by_block_of_100_mean_sync_us": [round(float(sy[i:i + 100].mean()), 1) for i in range(0, 1000, 100)],
                  "by_block_of_100_mean_rest_us": [round(float(rest[i:i + 100].mean()), 1) for i in range(0, 1000, 100)]}))
if stages:
    slow = rest > np.median(rest) + 15
    out = {}
    for k, v in stages.items():
        v = 1e6 * np.array(v[:1000])
        out[k] = {"fast": round(float(v[~slow].mean()), 1), "slow": round(float(v[slow].mean()), 1) if slow.any() else None}
    print(json.dumps({"n_slow": int(slow.sum()), "stages_us": out}))
    big = tot > 2000
    if big.any():
        print(json.dumps({"outliers": int(big.sum()), "mean_us_in_outliers": {k: round(float((1e6 * np.array(v[:1000]))[big].mean()), 1)
                                                                               for k, v in stages.items() if len(v) >= 1000},
                          "worker_ms_per_block": round(1e3 * ctrl._ahead.produce_s / max(ctrl._ahead.produced, 1), 3)}))
ctrl._ahead.stop()
