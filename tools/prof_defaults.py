#!/usr/bin/env python
"""The reference's own default plan sizes, launched back to back for a rocprofv3 --kernel-trace --stats run (profiles/
r04_defaults_kernel_stats.csv): run_mb_mpc.py's config-1 shape (one 2 x 512 model, n = 500, h = 10), run_grbal.py's default
(5 adapted 3 x 512 sets, n = 500, h = 10), run_rebal.py's default (LSTM 256, 5 envs, n = 500, h = 10) - the launches the
micro-tile kernels of csrc/l2a_micro.h take; `gru` / `lstm2`: a 256-unit GRU and a 2 x 256 LSTM stack at the ReBAL plan size
(csrc/l2a_rnn_micro.h).  L2A_MICRO=0 in the environment gives the 16-candidate kernels instead."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402

WHICH = sys.argv[1] if len(sys.argv) > 1 else "all"          # c1 | c3b | c6 | gru | lstm2 | c5shard | c3 | c4 | all: one shape per rocprof run keeps its row apart
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 400
plans = []
for name in ("c1_hc_rs_n500_h10_e1", "c3b_ant_rs_n500_h10_pb5_3x512"):
    if WHICH not in ("all", name.split("_")[0]):
        continue
    case = cases.CASES[name]
    env, model = cases.product_model(case)
    native = model.planner_model()
    dev = native.device
    m, n, h = case["m"], case["n"], case["h"]
    od, ad = env.observation_space.shape[0], env.action_space.shape[0]
    obs0 = torch.randn((m, od), device=dev)
    a = torch.rand((h, m * n, ad), device=dev) * 2 - 1
    best = torch.zeros((m,), dtype=torch.int64, device=dev)
    plans.append((model, lambda native=native, obs0=obs0, a=a, m=m, n=n, h=h, env=env, best=best:
                  native.plan_rs(obs0, a, m, n, h, 1.0, env.reward_spec, best_key=best)))
if WHICH == "c5shard":
    # one rank's shard of BASELINE config 5 (CEM iteration: n = 4000 over 8 ranks -> 500 candidates, h = 30, E = 5 mean): the
    # member-fan launch of csrc/l2a_mfma.h (L2A_FAN=0 in the environment: round 5's tile split)
    case = dict(cases.CASES["c5_hc_cem_n4000_h30_e5"], n=500)
    env, model = cases.product_model(case)
    native = model.planner_model()
    dev = native.device
    obs0 = torch.randn((1, 20), device=dev)
    a = torch.rand((30, 500, 6), device=dev) * 2 - 1
    best = torch.zeros((1,), dtype=torch.int64, device=dev)
    rets = torch.zeros((1, 500), dtype=torch.float32, device=dev)
    plans.append((model, lambda: native.plan_rs(obs0, a, 1, 500, 30, 1.0, env.reward_spec, returns_out=rets, best_key=best)))
if WHICH in ("c3", "c4"):
    # BASELINE config 3 (5 x 2000 candidates, 5 adapted sets: 625 tiles) / config 4's shape on one GPU (16 000 candidates, E = 5):
    # the double rounds of csrc/l2a_api.hip - config 3 = a launch of 255 double-tile workgroups + a launch of 115 shared tiles,
    # config 4 = one launch of 500 double tiles (L2A_DOUBLE=0 in the environment: the one-launch geometries before round 6)
    case = cases.CASES["c3_ant_rs_n2000_h20_pb5" if WHICH == "c3" else "c4_hc_rs_n16000_h30_e5"]
    env, model = cases.product_model(case)
    native = model.planner_model()
    dev = native.device
    m, n, h = case["m"], case["n"], case["h"]
    od, ad = env.observation_space.shape[0], env.action_space.shape[0]
    obs0 = torch.randn((m, od), device=dev)
    a = (torch.rand((h, m * n, ad), device=dev) * 2 - 1) * float(env.action_space.high[0])
    best = torch.zeros((m,), dtype=torch.int64, device=dev)
    plans.append((model, lambda: native.plan_rs(obs0, a, m, n, h, 1.0, env.reward_spec, best_key=best)))
if WHICH in ("all", "c6"):
    case = cases.CASES["c6_hc_rnn_rs_n500_h10_m5"]
    env, model = cases.product_rnn_model(case)
    native = model.planner_model()
    dev = native.device
    m, n, h, U = case["m"], case["n"], case["h"], case["units"]
    obs0 = torch.randn((m, 20), device=dev)
    c0 = torch.randn((m, U), device=dev)
    h0 = torch.tanh(torch.randn((m, U), device=dev))
    a = torch.rand((h, m * n, 6), device=dev) * 2 - 1
    best = torch.zeros((m,), dtype=torch.int64, device=dev)
    plans.append((model, lambda: native.plan_rs(obs0, c0, h0, a, m, n, h, 1.0, env.reward_spec, best_key=best)))
for tag, cell, hidden in (("gru", "gru", [256]), ("lstm2", "lstm", [256, 256])):
    # ... and the other cells of create_rnn at that plan size (csrc/l2a_rnn_micro.h): GRU 256, a 2 x 256 LSTM stack
    if WHICH not in ("all", tag):
        continue
    case = dict(cases.CASES["hc_rnn_rs_gru2_n48_h4"], units=sum(hidden), n=500, m=5, h=10, cell_type=cell, hidden_sizes=hidden)
    case.pop("reset_after", None)
    env, model = cases.product_rnn_model(case)
    native = model.planner_model()
    dev = native.device
    U = sum(hidden)
    obs0 = torch.randn((5, 20), device=dev)
    c0 = torch.randn((5, U), device=dev) * (1.0 if cell == "lstm" else 0.0)
    h0 = torch.tanh(torch.randn((5, U), device=dev))
    a = torch.rand((10, 2500, 6), device=dev) * 2 - 1
    best = torch.zeros((5,), dtype=torch.int64, device=dev)
    plans.append((model, lambda native=native, obs0=obs0, c0=c0, h0=h0, a=a, env=env, best=best:
                  native.plan_rs(obs0, c0, h0, a, 5, 500, 10, 1.0, env.reward_spec, best_key=best)))
for _, plan in plans:       # one shape after the other, back to back: the clocks are up after the first few dozen launches
    for _ in range(REPS):
        plan()
    torch.cuda.synchronize()
print("done", WHICH, REPS)
