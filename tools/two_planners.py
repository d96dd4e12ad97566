#!/usr/bin/env python
"""Two planner PROCESSES on one GPU (config 2 each): what the exchange traffic of the tile split (151 MB per launch against 7 MB
algorithmic) costs when the chip is shared.  Kernel-only loops (candidates resident), HIP events per launch.

    python tools/two_planners.py            parent: runs the single-process references, then two concurrent children
    python tools/two_planners.py child TAG  one planner (L2A_SPLIT from the environment)
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def child(tag):
    import numpy as np
    import torch
    import cases
    from learning_to_adapt_amd import _lib
    case = cases.CASES["c2_hc_rs_n2000_h30_e5"]
    env, model = cases.product_model(case)
    native = model.planner_model()
    dev = native.device
    obs0 = torch.randn((1, 20), device=dev)
    a = torch.rand((30, 2000, 6), device=dev) * 2 - 1
    best = torch.zeros((1,), dtype=torch.int64, device=dev)
    launch = lambda: native.plan_rs(obs0, a, 1, 2000, 30, 1.0, env.reward_spec, best_key=best)  # noqa: E731
    for _ in range(150):
        launch()
    torch.cuda.synchronize()
    # rendezvous through the file system so that both children time the same window
    open("/tmp/l2a_two_%s.ready" % tag, "w").close()
    t_end = time.time() + 20
    while time.time() < t_end and len([f for f in os.listdir("/tmp") if f.startswith("l2a_two_") and f.endswith(".ready")]) < int(os.environ.get("L2A_TWO_N", "1")):
        time.sleep(0.001)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(400)]
    t0 = time.perf_counter()
    for s, e in evs:
        s.record()
        launch()
        e.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = np.array([s.elapsed_time(e) for s, e in evs])
    flagged = _lib.Context.get(0).launch_status_value()
    print(json.dumps({"tag": tag, "split": os.environ.get("L2A_SPLIT", "1"), "kernel_ms_p50": round(float(np.median(ms)), 4),
                      "kernel_ms_p95": round(float(np.percentile(ms, 95)), 4), "wall_ms_per_launch": round(1e3 * wall / len(evs), 4),
                      "launches_flagged": int(flagged)}), flush=True)


def child_sync(tag):
    """The product's blocking step path (`l2a_plan_rs_sync`, default policy): planner p0 runs for the whole window, planner p1
    joins a third of the way in and leaves two thirds in.  Wall time per plan in windows of 100 plans.  Round 6 measured a host-side
    geometry tuner with this (switch a shape to the unsplit launch when its split launches run slower than the unsplit one, probe
    the split again every 256 launches; L2A_GEO_AUTO): it changed nothing - the planner that arrives second has never seen itself
    alone (no baseline to be slow against), and ONE planner going unsplit while the other keeps its 250 workgroups gains nothing;
    only both unsplit (L2A_SPLIT=0 on both: 2.59 against 2.81 ms) does.  The tuner was removed again; this scenario stays as the
    measurement (profiles/r06_two_planners_sync.jsonl)."""
    import numpy as np
    import torch
    import cases
    case = cases.CASES["c2_hc_rs_n2000_h30_e5"]
    env, model = cases.product_model(case)
    native = model.planner_model()
    dev = native.device
    obs0 = np.random.RandomState(1).randn(1, 20)
    a = torch.rand((30, 2000, 6), device=dev) * 2 - 1
    plan = lambda: native.plan_rs_sync(obs0, a, 1, 2000, 30, 1.0, env.reward_spec)  # noqa: E731
    for _ in range(150):
        plan()
    open("/tmp/l2a_two_%s.ready" % tag, "w").close()
    while len([f for f in os.listdir("/tmp") if f.startswith("l2a_two_") and f.endswith(".ready")]) < 2:
        time.sleep(0.001)
    t_start = time.perf_counter()
    total = 6.0                                     # seconds
    lo, hi = (0.0, total) if tag == "p0" else (total / 3, 2 * total / 3)
    while time.perf_counter() - t_start < lo:
        time.sleep(0.001)
    windows = []
    while time.perf_counter() - t_start < hi:
        t0 = time.perf_counter()
        for _ in range(100):
            plan()
        t1 = time.perf_counter()
        windows.append((round(t0 - t_start, 2), round(1e3 * (t1 - t0) / 100, 4)))
    print(json.dumps({"tag": tag, "path": "l2a_plan_rs_sync, L2A_SPLIT=%s" % os.environ.get("L2A_SPLIT", "1 (default)"), "active_s": [lo, hi],
                      "ms_per_plan_by_window_of_100 (window start s, ms)": windows}), flush=True)


def run_sync(split):
    for f in os.listdir("/tmp"):
        if f.startswith("l2a_two_"):
            os.remove(os.path.join("/tmp", f))
    env = dict(os.environ)
    env.pop("L2A_SPLIT", None)
    if split is not None:
        env["L2A_SPLIT"] = str(split)
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "child_sync", "p%d" % i], env=env, stdout=subprocess.PIPE) for i in range(2)]
    for p in procs:
        print(p.communicate()[0].decode().strip().splitlines()[-1], flush=True)


def run(n, split):
    for f in os.listdir("/tmp"):
        if f.startswith("l2a_two_"):
            os.remove(os.path.join("/tmp", f))
    env = dict(os.environ, L2A_SPLIT=str(split), L2A_TWO_N=str(n))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "child", "p%d" % i], env=env, stdout=subprocess.PIPE) for i in range(n)]
    outs = [json.loads(p.communicate()[0].decode().strip().splitlines()[-1]) for p in procs]
    print(json.dumps({"processes": n, "split_policy": split, "planners": outs}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        child(sys.argv[2])
    elif len(sys.argv) > 2 and sys.argv[1] == "child_sync":
        child_sync(sys.argv[2])
    elif len(sys.argv) > 1 and sys.argv[1] == "sync":
        run_sync(None)  # round 6: the blocking path, default policy - a co-tenant comes and goes
        run_sync(0)     # ... and with the split off on both
    else:
        run(1, 1)       # the product's default: tile split, 250 workgroups
        run(1, 0)       # one workgroup per tile (125 workgroups, no exchange)
        run(2, 0)       # two planners sharing the chip, no exchange: 2 x 125 workgroups
        run(2, 1)       # two planners, each WANTING the split: partners are not co-resident -> flagged launches (the controller degrades)
