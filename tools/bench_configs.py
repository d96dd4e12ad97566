#!/usr/bin/env python
"""Kernel-only timings of every BASELINE.json config (and kernel / split variants of config 2)
on ONE MI355X.  Developer report that feeds DESIGN.md; the contract benchmark is bench.py.

    python tools/bench_configs.py > profiles/rNN_configs.jsonl
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
from learning_to_adapt_amd import _lib  # noqa: E402

PEAK = 157.3


def flops(case, env):
    od, ad = env.observation_space.shape[0], env.action_space.shape[0]
    sizes = [od + ad] + list(case["hidden"]) + [od]
    mac = sum(a * b for a, b in zip(sizes[:-1], sizes[1:]))
    e_eff = case["E"] if case["mode"] == "mean" else 1
    return 2.0 * mac * e_eff * case["n"] * case["m"] * case["h"]


def time_plan(native, case, env, reps=30, split=None, kernel="auto"):
    ctx = _lib.Context.get(0)
    ctx.set_kernel(kernel)
    if split is not None:
        ctx.set_split(split)
    dev = native.device
    m, n, h = case["m"], case["n"], case["h"]
    od, ad = env.observation_space.shape[0], env.action_space.shape[0]
    obs0 = torch.randn((m, od), device=dev)
    low = torch.as_tensor(env.action_space.low, dtype=torch.float32, device=dev)
    high = torch.as_tensor(env.action_space.high, dtype=torch.float32, device=dev)
    a = torch.rand((h, m * n, ad), device=dev) * (high - low) + low
    best = torch.zeros((m,), dtype=torch.int64, device=dev)
    ms = time_launches(lambda: native.plan_rs(obs0, a, m, n, h, case.get("discount", 1.0), env.reward_spec, best_key=best), reps)
    ctx.launch_status()
    ctx.set_kernel("auto")
    ctx.set_split(1)
    return ms


def time_launches(launch, reps=30, warm_ms=80.0):
    """Median duration of one launch by HIP events, measured with the shader clock UP: after idle time the MI355X needs
    some tens of milliseconds of back-to-back work to reach its ~2.39 GHz (tools/timeline*.py read s_memtime against
    s_memrealtime: 2.0-2.15 GHz over the first ~10 launches of a 0.4 ms kernel, 2.37-2.39 GHz from ~100 launches on), and
    the fp32 matrix peak the fractions are quoted against is the peak-clock number.  Rounds 1-2 and the first profiles
    of round 3 warmed with 3 launches: their figures for the SHORT kernels (LSTM plans, c1, c3b, c6) are 5-12 % low."""
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record()
    for _ in range(3):
        launch()
    e0.record()
    torch.cuda.synchronize()
    per = max(s0.elapsed_time(e0) / 3.0, 1e-3)
    for _ in range(int(min(max(warm_ms / per, 3), 2000))):
        launch()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in evs:
        s.record()
        launch()
        e.record()
    torch.cuda.synchronize()
    return float(np.median([s.elapsed_time(e) for s, e in evs]))


def _step_stats(ctrl):
    """hits / synchronous draws of the C controller behind a drop-in controller (None: the Python path ran)."""
    st = getattr(ctrl, "_cstep", None)
    if st is None:
        return None
    s = st.stats()
    return {"steps": s["steps"], "hits": s["hits"], "sync_draws": s["sync_draws"], "stale_blocks": s["misses"]}


def _retire(ctrl):
    """Stop a finished section's draw-ahead worker: an idle chain of an earlier section still costs the later sections'
    parity-mode steps tens of microseconds each (measured: ReBAL 0.36 ms after the GrBAL section, 0.29 ms alone)."""
    if getattr(ctrl, "_ahead", None) is not None:
        ctrl._ahead.stop()
    if getattr(ctrl, "_cstep", None) is not None:
        ctrl._cstep.close()
        ctrl._cstep = None


def report(tag, case, env, ms, **extra):
    fl = flops(case, env)
    row = dict(config=tag, n=case["n"], h=case["h"], m=case["m"], E=case["E"], mode=case["mode"],
               hidden=case["hidden"], kernel_ms=round(ms, 4), plan_steps_per_s=round(1e3 / ms, 2),
               tflops=round(fl / ms / 1e9, 2), frac_fp32_peak=round(fl / ms / 1e9 / PEAK, 4),
               mlp_steps_per_ms=round(case["n"] * case["m"] * case["h"] *
                                      (case["E"] if case["mode"] == "mean" else 1) / ms, 1))
    row.update(extra)
    print(json.dumps(row), flush=True)


def main():
    C = cases.CASES
    # config 2 under every policy / kernel
    case = C["c2_hc_rs_n2000_h30_e5"]
    env, model = cases.product_model(case)
    native = model.planner_model()
    for split, name in ((1, "shared-set split (default)"), (2, "whole-set split"), (0, "one workgroup per tile")):
        report("config 2, MFMA kernel, " + name, case, env, time_plan(native, case, env, split=split))
    report("config 2, VALU kernel (fp32 FMA, no matrix cores)", case, env, time_plan(native, case, env, reps=5, kernel="valu"))
    # config 1 shape on the GPU (the reference's CPU-runnable case)
    for name in ("c1_hc_rs_n500_h10_e1", "c3_ant_rs_n2000_h20_pb5", "c3b_ant_rs_n500_h10_pb5_3x512",
                 "c4_hc_rs_n16000_h30_e5"):
        case = C[name]
        env, model = cases.product_model(case)
        report(name + " (single GPU)", case, env, time_plan(model.planner_model(), case, env))
    # the same default plan sizes on the 16-candidate kernels alone (l2a_set_micro(0): round 3's geometry)
    _lib.Context.get(0).set_micro(0)
    for name in ("c1_hc_rs_n500_h10_e1", "c3b_ant_rs_n500_h10_pb5_3x512"):
        case = C[name]
        env, model = cases.product_model(case)
        report(name + " (single GPU), 16-candidate tiles only", case, env, time_plan(model.planner_model(), case, env))
    _lib.Context.get(0).set_micro(1)
    # run_mb_mpc.py's own defaults (:77-78,85,97): ONE 2 x 512 model, n = 2000, h = 20, and the Sampler hands get_actions the
    # observations of all num_rollouts = 10 envs at once - a 1250-tile plan
    case = dict(C["c1_hc_rs_n500_h10_e1"], n=2000, h=20, m=10)
    env, model = cases.product_model(case)
    report("run_mb_mpc.py default (single model, n=2000, h=20, m=10 rollouts)", case, env, time_plan(model.planner_model(), case, env))
    for mode in ("numpy", "device"):
        ctrl = cases.product_controller(case, model=model, env=env, rng=mode)
        obs_np = np.random.RandomState(3).randn(10, 20)
        np.random.seed(0)
        # 30 calls of warm-up: the first draws of a new plan size compute the generator's jump polynomials for their slice
        # offsets (~100 ms once per process and size, csrc/l2a_rng.c) - inside 50 timed calls that was +0.4 ms per call
        # (r05: 2.288 here against tools/probe_step.py's 1.899 for the same loop)
        for _ in range(30):
            ctrl.get_actions(obs_np)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            ctrl.get_actions(obs_np)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / 200
        print(json.dumps(dict(config="run_mb_mpc.py default end to end through MPCController.get_actions (10 envs), rng=" + mode,
                              ms_per_call=round(ms, 3), calls_per_s=round(1e3 / ms, 1), env_steps_per_s=round(1e4 / ms, 1))), flush=True)
        _retire(ctrl)
    # GrBAL default shape: 3 x 512, n = 2000, h = 20, 5 adapted sets
    case = dict(C["c3_ant_rs_n2000_h20_pb5"], hidden=[512, 512, 512])
    env, model = cases.product_model(case)
    report("config 3 with the run_grbal.py default 3x512 network", case, env, time_plan(model.planner_model(), case, env))
    # GrBAL controller step (samplers/sampler.py:81-91): switch_to_pre_adapt + adapt (5 envs x 16
    # transitions, one SGD step each) + re-upload of the 5 adapted sets + plan (config 3, 3x512)
    from learning_to_adapt_amd.dynamics import MetaMLPDynamicsModel
    from learning_to_adapt_amd.policies import MPCController
    from learning_to_adapt_amd.utils import synthetic
    env = cases.SyntheticEnv("ant")
    gm = MetaMLPDynamicsModel(name="dyn", env=env, hidden_sizes=(512, 512, 512), inner_learning_rate=0.01, init_seed=0)
    gm.set_normalization(synthetic.make_norm(41, 8, env.action_space.low, env.action_space.high, 2000))
    rs = np.random.RandomState(0)
    ob = [rs.randn(16, 41) for _ in range(5)]
    ac = [rs.uniform(-150, 150, (16, 8)) for _ in range(5)]
    nx = [o + 0.1 * rs.randn(16, 41) for o in ob]
    for mode, n_c, h_c in (("numpy", 2000, 20), ("device", 2000, 20), ("numpy", 500, 10), ("device", 500, 10)):
        gc = MPCController(name="p", env=env, dynamics_model=gm, n_candidates=n_c, horizon=h_c, rng=mode)
        obs0 = rs.randn(5, 41)
        t_adapt, t_all = [], []
        for it in range(12):        # the adaptation on its own (with a synchronisation the real loop does not have)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            gm.switch_to_pre_adapt()
            gm.adapt(ob, ac, nx)
            gm.planner_model()
            torch.cuda.synchronize()
            if it > 1:
                t_adapt.append(time.perf_counter() - t0)
        np.random.seed(0)
        for it in range(45):        # the controller step as samplers/sampler.py:81-91 runs it: adapt, then get_actions
            t0 = time.perf_counter()
            gm.switch_to_pre_adapt()
            gm.adapt(ob, ac, nx)
            gc.get_actions(obs0)
            if it >= 5:
                t_all.append(time.perf_counter() - t0)
        label = "config 3 / 3x512" if n_c == 2000 else "run_grbal.py default n=500 h=10 / 3x512"
        print(json.dumps(dict(config="GrBAL controller step (adapt 5 envs + plan %s), rng=%s" % (label, mode),
                              adapt_and_upload_ms=round(1e3 * float(np.median(t_adapt)), 3), step_ms=round(1e3 * float(np.median(t_all)), 3),
                              step_ms_mean=round(1e3 * float(np.mean(t_all)), 3), step_ms_max=round(1e3 * float(np.max(t_all)), 3))), flush=True)
        _retire(gc)
    # recurrent planner (ReBAL, run_scripts/run_rebal.py defaults: LSTM(256), n=500, h=10, 5 envs) + larger plans
    for label, over in (("c6 ReBAL default (LSTM 256, n=500, h=10, m=5)", {}),
                        ("ReBAL, LSTM 256, n=2000, h=30, m=1", {"n": 2000, "h": 30, "m": 1}),
                        ("ReBAL, LSTM 256, n=4096, h=30, m=1 (256 tiles)", {"n": 4096, "h": 30, "m": 1}),
                        ("ReBAL, LSTM 512, n=4096, h=20, m=1", {"n": 4096, "h": 20, "m": 1, "units": 512})):
        case = dict(C["c6_hc_rnn_rs_n500_h10_m5"], **over)
        env, model = cases.product_rnn_model(case)
        native = model.planner_model()
        dev = native.device
        m, n, h, U = case["m"], case["n"], case["h"], case["units"]
        obs0 = torch.randn((m, 20), device=dev)
        c0 = torch.randn((m, U), device=dev)
        h0 = torch.tanh(torch.randn((m, U), device=dev))
        a = torch.rand((h, m * n, 6), device=dev) * 2 - 1
        best = torch.zeros((m,), dtype=torch.int64, device=dev)
        for kernel in ("mfma", "mfma, 16-candidate tiles only", "valu"):
            if kernel == "valu" and n * m * h > 30000:
                continue
            if kernel.endswith("only") and not label.startswith("c6"):
                continue
            _lib.Context.get(0).set_kernel(kernel.split(",")[0])
            _lib.Context.get(0).set_micro(0 if kernel.endswith("only") else 1)
            ms = time_launches(lambda: native.plan_rs(obs0, c0, h0, a, m, n, h, 1.0, env.reward_spec, best_key=best), 20)
            _lib.Context.get(0).set_micro(1)
            fl = 2.0 * ((26 + U) * 4 * U + U * 20) * n * m * h
            print(json.dumps(dict(config=label + ", " + kernel + " kernel", n=n, h=h, m=m, units=U,
                                  kernel_ms=round(ms, 4), tflops=round(fl / ms / 1e9, 2),
                                  frac_fp32_peak=round(fl / ms / 1e9 / PEAK, 4),
                                  lstm_steps_per_ms=round(n * m * h / ms, 1))), flush=True)
        _lib.Context.get(0).set_kernel("auto")
    # the other cells create_rnn builds, at the ReBAL plan size: matrix-core kernel (l2a_rnn_mfma.h) and VALU kernel
    for label, ctype, sizes in (("GRU 256", "gru", [256]), ("LSTM 2 x 256", "lstm", [256, 256]), ("BasicRNN 256", "rnn", [256])):
        case = dict(C["hc_rnn_rs_gru2_n48_h4"], n=500, h=10, m=5, cell_type=ctype, hidden_sizes=sizes, units=sum(sizes))
        case.pop("reset_after", None)
        env, model = cases.product_rnn_model(case)
        native = model.planner_model()
        dev = native.device
        m, n, h, U = case["m"], case["n"], case["h"], case["units"]
        obs0 = torch.randn((m, 20), device=dev)
        c0 = torch.randn((m, U), device=dev) * (1.0 if ctype == "lstm" else 0.0)
        h0 = torch.tanh(torch.randn((m, U), device=dev))
        a = torch.rand((h, m * n, 6), device=dev) * 2 - 1
        best = torch.zeros((m,), dtype=torch.int64, device=dev)
        macs, kin = 0, 26
        for u in sizes:
            macs += (kin + u) * u * {"lstm": 4, "gru": 3, "rnn": 1}[ctype]
            kin = u
        fl = 2.0 * (macs + kin * 20) * n * m * h
        for kernel in ("mfma", "valu"):
            _lib.Context.get(0).set_kernel(kernel)
            ms = time_launches(lambda: native.plan_rs(obs0, c0, h0, a, m, n, h, 1.0, env.reward_spec, best_key=best), 10)
            print(json.dumps(dict(config="ReBAL plan size with %s (generic recurrent kernels), %s kernel" % (label, kernel), n=n, h=h, m=m,
                                  units=U, kernel_ms=round(ms, 4), tflops=round(fl / ms / 1e9, 2),
                                  frac_fp32_peak=round(fl / ms / 1e9 / PEAK, 4))), flush=True)
        _lib.Context.get(0).set_kernel("auto")
    case = C["c6_hc_rnn_rs_n500_h10_m5"]
    for mode in ("numpy", "device"):
        ctrl = cases.product_rnn_controller(case, rng=mode)
        obs = np.random.RandomState(0).randn(5, 20)
        ctrl.reset(dones=[True] * 5)
        np.random.seed(0)
        for _ in range(10):         # steady state: the draw-ahead chain is running, buffers exist
            ctrl.get_actions(obs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            ctrl.get_actions(obs)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / 200
        print(json.dumps(dict(config="c6 ReBAL controller step end to end (plan + hidden-state advance), rng=" + mode,
                              ms_per_call=round(ms, 3), calls_per_s=round(1e3 / ms, 1), native_step=_step_stats(ctrl))), flush=True)
        _retire(ctrl)
    # config 5: one CEM plan step (5 iterations x 4000 candidates) through the drop-in controller
    case = C["c5_hc_cem_n4000_h30_e5"]
    ctrl = cases.product_controller(case)
    gold = cases.load_golden("c5_hc_cem_n4000_h30_e5_s0")
    np.random.seed(0)
    ctrl.get_actions(gold["obs0"])
    t0 = time.perf_counter()
    k = 10
    for _ in range(k):
        ctrl.get_actions(gold["obs0"])
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / k
    _retire(ctrl)
    env, model = cases.product_model(case)
    one = dict(case, n=4000)
    k_ms = time_plan(model.planner_model(), one, env)
    report("config 5: CEM rollout kernel, one iteration (n=4000)", one, env, k_ms,
           cem_plan_step_ms_end_to_end_host_rng=round(ms, 2), cem_iters=5,
           note="end-to-end includes 5 x np.random.normal(720k) + clip + elite statistics on the host")
    # config 5 as one of eight ranks runs it: the 500-candidate shard of an iteration (member fan: one workgroup per candidate
    # tile and member; round 5 ran it on the tile split: 64 workgroups), tools/probe_c5_shard.py has the rank's whole plan step
    shard = dict(case, n=500)
    for label, fan in (("member fan (default)", 1), ("tile split, fan off (round 5's geometry)", 0)):
        _lib.Context.get(0).set_fan(fan)
        report("config 5 shard (n=500, h=30, E=5 mean) = one rank of 8, " + label, shard, env, time_plan(model.planner_model(), shard, env))
    _lib.Context.get(0).set_fan(1)
    for cem_mode in ("reference", "fixed"):
        ctrl = cases.product_controller(case, rng="device", cem_mode=cem_mode)
        ctrl.get_actions(gold["obs0"])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            ctrl.get_actions(gold["obs0"])
        torch.cuda.synchronize()
        print(json.dumps(dict(config="config 5: CEM plan step end to end, rng=device, cem_mode=%s (sampling / elites / refit on the GPU)" % cem_mode,
                              ms_per_call=round(1e3 * (time.perf_counter() - t0) / 10, 3))), flush=True)
    # host<->device inclusive: get_actions through the controller (parity mode: host MT19937 + H2D per step)
    case = C["c2_hc_rs_n2000_h30_e5"]
    for mode in ("numpy", "device"):
        ctrl = cases.product_controller(case, rng=mode)
        obs_np = np.array(cases.load_golden("c2_hc_rs_n2000_h30_e5_s0")["obs0"])
        np.random.seed(0)
        for _ in range(10):
            ctrl.get_actions(obs_np)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            ctrl.get_actions(obs_np)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / 200
        print(json.dumps(dict(config="config 2 end to end through MPCController.get_actions, rng=" + mode,
                              ms_per_call=round(ms, 3), calls_per_s=round(1e3 / ms, 1), native_step=_step_stats(ctrl))), flush=True)
        _retire(ctrl)


if __name__ == "__main__":
    main()
