#!/usr/bin/env python
"""BASELINE config 5 as ONE rank of eight sees it (CEM, 5 iterations x n = 4000, h = 30, E = 5 mean; shard = 500 candidates).

    python tools/probe_c5_shard.py > profiles/rNN_probe_c5_shard.jsonl

Two measurements on one MI355X:
 1. the rollout launch of the shard (n = 500, h = 30, E = 5 mean) by HIP events with the clocks up, under every launch
    geometry the library has for it: member fan (one workgroup per candidate tile and member, l2a_set_fan), tile split
    (two workgroups per tile), unsplit, micro tiles - and, beside it, the whole n = 4000 iteration one GPU runs alone;
 2. the rank's CEM plan step end to end: the drop-in controller with `rng="device"`, told it is rank r of 8 (its shard
    range lo .. hi), sample (all 4000 rows - every rank keeps every sample) / rollout (its 500) / refit (all 4000), five
    iterations + pick.  The all-gather is replaced by a local copy of this rank's returns into every rank's slot (no second
    GPU here): the elite statistics are then meaningless, their cost is not.
From these DESIGN.md section 6 derives the predicted 8-GPU speed-up of config 5.
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
sys.path.insert(0, os.path.join(ROOT, "tools"))
import cases  # noqa: E402
from bench_configs import PEAK, flops, time_launches  # noqa: E402
from learning_to_adapt_amd import _lib  # noqa: E402

WORLD = 8


def time_geometry(native, case, env, fan, split, micro, reps=40):
    ctx = _lib.Context.get(0)
    ctx.set_fan(fan)
    ctx.set_split(split)
    ctx.set_micro(micro)
    dev = native.device
    m, n, h = case["m"], case["n"], case["h"]
    obs0 = torch.randn((m, 20), device=dev)
    a = torch.rand((h, m * n, 6), device=dev) * 2 - 1
    best = torch.zeros((m,), dtype=torch.int64, device=dev)
    rets = torch.zeros((m, n), dtype=torch.float32, device=dev)
    try:
        ms = time_launches(lambda: native.plan_rs(obs0, a, m, n, h, 1.0, env.reward_spec, returns_out=rets, best_key=best), reps)
        ctx.launch_status()
    finally:
        ctx.set_fan(1)
        ctx.set_split(1)
        ctx.set_micro(1)
    return ms


def main():
    full = cases.CASES["c5_hc_cem_n4000_h30_e5"]
    env, model = cases.product_model(full)
    native = model.planner_model()
    rows = []
    who = {500: "shard of one rank of 8 (n=500)", 1000: "shard of one rank of 4 (n=1000)", 2000: "shard of one rank of 2 (n=2000)",
           4000: "on one GPU alone (n=4000)"}
    for n in (500, 1000, 2000, 4000):
        case = dict(full, n=n)
        for label, fan, split, micro in (("member fan", 1, 1, 0), ("tile split (r5 default)", 0, 1, 0), ("unsplit", 0, 0, 0),
                                         ("micro tiles", 0, 1, 2), ("library default", 1, 1, 1)):
            if (n >= 2000 and label == "member fan") or (n >= 1000 and label == "micro tiles"):
                continue
            ms = time_geometry(native, case, env, fan, split, micro)
            fl = flops(case, env)
            row = dict(what="config 5 %s: CEM rollout, one iteration" % who[n],
                       geometry=label, n=n, h=case["h"], E=case["E"], kernel_ms=round(ms, 4), tflops=round(fl / ms / 1e9, 2),
                       frac_fp32_peak=round(fl / ms / 1e9 / PEAK, 4))
            rows.append(row)
            print(json.dumps(row), flush=True)

    # ---- the rank's plan step: controller with a fake world of eight ------------------------------------------------
    from learning_to_adapt_amd.policies import MPCController

    class OneRankOfEight(MPCController):
        fake_rank = 0

        def _dist(self):
            return self.fake_rank, WORLD

        @staticmethod
        def _all_gather(mine, world):
            return [mine for _ in range(world)]

        def _agree(self, flag, world):
            return bool(flag)

    gold = cases.load_golden("c5_hc_cem_n4000_h30_e5_s0")
    obs = np.array(gold["obs0"])
    out = {}
    for label, world_cls, fan in (("one GPU alone (n=4000 per iteration)", MPCController, 1),
                                  ("rank 0 of 8 (500 of 4000), member fan", OneRankOfEight, 1),
                                  ("rank 0 of 8 (500 of 4000), tile split (r5 geometry)", OneRankOfEight, 0)):
        _lib.Context.get(0).set_fan(fan)
        ctrl = world_cls(name="policy", env=env, dynamics_model=model, discount=1.0, n_candidates=full["n"], horizon=full["h"],
                         use_cem=True, num_cem_iters=full["num_cem_iters"], rng="device")
        torch.manual_seed(0)
        for _ in range(8):
            ctrl.get_actions(obs)
        torch.cuda.synchronize()
        ts = []
        for _ in range(40):
            t0 = time.perf_counter()
            ctrl.get_actions(obs)
            ts.append(1e3 * (time.perf_counter() - t0))
        _lib.Context.get(0).set_fan(1)
        out[label] = float(np.median(ts))
        print(json.dumps(dict(what="config 5 CEM plan step end to end, rng=device (5 x sample / rollout / refit + pick)", who=label,
                              plan_step_ms_p50=round(float(np.median(ts)), 3), plan_step_ms_min=round(float(np.min(ts)), 3),
                              plan_step_ms_p90=round(float(np.percentile(ts, 90)), 3))), flush=True)
    one = out["one GPU alone (n=4000 per iteration)"]
    for k, v in out.items():
        if k.startswith("rank"):
            print(json.dumps(dict(what="predicted 8-GPU speed-up of a config-5 plan step (one GPU alone / one rank's step; the "
                                       "all-gathers of 5 x 2 KB per rank over xGMI not included: ~5 x 20-30 us)", geometry=k,
                                  speedup=round(one / v, 2), with_5_collectives_of_30us=round(one / (v + 0.15), 2))), flush=True)


if __name__ == "__main__":
    main()
