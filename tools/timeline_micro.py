#!/usr/bin/env python
"""Phase timeline of the micro-tile kernels (csrc/l2a_micro.h; developer aid, needs a GPU and a library built with
`python tools/build_variant.py timeline -DL2A_TIMELINE`).  Workgroup 0, four waves, shader clocks per phase.

    python tools/timeline_micro.py lstm [n] [m] [h]        stamps: 0 step start | 1 gate GEMM done | 2 gates done | 3 output partials
                                                            written | 4 past the barrier | 5 reduce / reward done | 6 next inputs written
    python tools/timeline_micro.py mlp [case] [n] [m] [h]   stamps: see NAMES_MLP
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_TL = os.path.join(ROOT, "learning_to_adapt_amd", "libl2a_hip_timeline.so")
if not os.path.exists(_TL):
    raise SystemExit("build the timeline library first: python tools/build_variant.py timeline -DL2A_TIMELINE")
os.environ["L2A_LIB_PATH"] = _TL

import numpy as np   # noqa: E402
import torch   # noqa: E402

sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
from learning_to_adapt_amd import _lib  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "lstm"
NAMES_LSTM = ["gate GEMM", "gates", "out", "barrier", "reduce+reward", "next x"]
# MLP stamps per step: 0 start | 3 layer 0 of the first set done | 4 its rows written | 5 hidden layers done | 6 output partials written |
# 7 past the barrier | 11 every set reduced | 12 mean / reward / state done | 13 next inputs written
MLP_SLOTS = [0, 3, 4, 5, 6, 7, 11, 12, 13]
NAMES_MLP = ["set 0: layer 0", "write", "hidden", "out", "barrier", "reduce + other sets", "mean+reward", "next x"]


def run(plan, h, names, slots):
    ctx = _lib.Context.get(0)
    ctx.set_micro(2)
    for _ in range(3):
        plan()
    torch.cuda.synchronize()
    dbg = torch.zeros(h * 4 * 16 + 64 + 4 * 1024, dtype=torch.int64, device="cuda")
    ctx.check(ctx.lib.l2a_set_debug_buffer(ctx.handle, ctypes.c_void_p(dbg.data_ptr())), "dbg")
    for _ in range(int(os.environ.get("L2A_TL_WARM", "200"))):
        plan()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    plan()
    ev1.record()
    torch.cuda.synchronize()
    ctx.check(ctx.lib.l2a_set_debug_buffer(ctx.handle, ctypes.c_void_p(0)), "dbg")
    ctx.set_micro(1)
    d = dbg[:h * 4 * 16].view(h, 4, 16).cpu().numpy().astype(np.int64)
    print("launch %.4f ms (stamped build)" % ev0.elapsed_time(ev1))
    lo = 2 if h > 4 else 0
    for w in range(4):
        seg = np.median(np.diff(d[lo:, w][:, slots], axis=1), axis=0)
        print("  wave %d: " % w + "  ".join("%s %6.0f" % (nm, v) for nm, v in zip(names, seg)))
    if d[0, 0, 8]:
        print("  prologue (wave 0): entry -> constants in LDS %d, -> state / first operands requested %d, -> first step %d clocks"
              % (d[0, 0, 9] - d[0, 0, 8], d[0, 0, 10] - d[0, 0, 9], d[0, 0, 0] - d[0, 0, 10]))
    step = np.diff(d[:, 0, 0])
    print("  step period: median %d clocks (min %d max %d); first stamp -> last stamp of the launch %d"
          % (np.median(step), step.min(), step.max(), d[:, :, slots].max() - d[0, :, 0].min()))
    wg = dbg[h * 4 * 16 + 64:].view(1024, 4).cpu().numpy().astype(np.int64)
    ran = wg[:, 1] != 0
    if ran.any():
        life = (wg[:, 1] - wg[:, 0]) * 10        # ns
        if d[0, 0, 8] and life[0] > 0:
            print("  shader clock over workgroup 0's lifetime: %.0f MHz (%d clocks from entry to the last stamp, %.1f us of real time)"
                  % ((d[:, 0, slots].max() - d[0, 0, 8]) / (life[0] / 1e3), d[:, 0, slots].max() - d[0, 0, 8], life[0] / 1e3))
        print("  per-workgroup records: %d workgroups; launch (first start -> last end) %.1f us; lifetimes by XCD / micro tiles (us):"
              % (ran.sum(), (wg[ran, 1].max() - wg[ran, 0].min()) / 100.0))
        for x in range(8):
            sel = ran & (wg[:, 2] == x)
            if not sel.any():
                continue
            parts = []
            for mt in (3, 2, 1):
                s2 = sel & ((wg[:, 3] & 255) == mt)
                if s2.any():
                    parts.append("MT %d: %3d wgs median %.1f max %.1f" % (mt, s2.sum(), np.median(life[s2]) / 1e3, life[s2].max() / 1e3))
            envs = sorted(set((wg[sel, 3] >> 8).tolist()))
            print("    XCD %d envs %s: %s" % (x, envs, "; ".join(parts)))
    return d


if what == "lstm":
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    m = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    h = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    units = 256
    case = dict(cases.CASES["c6_hc_rnn_rs_n500_h10_m5"], units=units, n=n, m=m, h=h)
    env, model = cases.product_rnn_model(case)
    native = model.planner_model()
    dev = native.device
    obs0 = torch.randn((m, 20), device=dev)
    c0 = torch.randn((m, units), device=dev)
    h0 = torch.tanh(torch.randn((m, units), device=dev))
    a = torch.rand((h, m * n, 6), device=dev) * 2 - 1
    best = torch.zeros(m, dtype=torch.int64, device=dev)
    print("micro-tile LSTM rollout, units %d, n %d, m %d, h %d" % (units, n, m, h))
    run(lambda: native.plan_rs(obs0, c0, h0, a, m, n, h, 1.0, env.reward_spec, best_key=best), h, NAMES_LSTM, list(range(7)))
elif what == "rnn":
    # generic cells (csrc/l2a_rnn_micro.h): python tools/timeline_micro.py rnn gru|lstm|rnn 256[,256..] [n m h]
    # stamps: 0 step start | per layer l: 6 + 3 l product 0 done, 15 (GRU, last layer's survives) r * h written + barrier passed,
    # 7 + 3 l product 1 done, 8 + 3 l gates done | 1 past the layers | 2 output partials written | 3 past the barrier |
    # 4 reduce / reward done | 5 next inputs written
    cell = sys.argv[2] if len(sys.argv) > 2 else "gru"
    hidden = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "256").split(",")]
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 500
    m = int(sys.argv[5]) if len(sys.argv) > 5 else 5
    h = int(sys.argv[6]) if len(sys.argv) > 6 else 10
    case = dict(cases.CASES["hc_rnn_rs_gru2_n48_h4"], units=sum(hidden), n=n, m=m, h=h, cell_type=cell, hidden_sizes=hidden)
    case.pop("reset_after", None)
    env, model = cases.product_rnn_model(case)
    native = model.planner_model()
    dev = native.device
    U = sum(hidden)
    obs0 = torch.randn((m, 20), device=dev)
    c0 = torch.randn((m, U), device=dev) * (1.0 if cell == "lstm" else 0.0)
    h0 = torch.tanh(torch.randn((m, U), device=dev))
    a = torch.rand((h, m * n, 6), device=dev) * 2 - 1
    best = torch.zeros(m, dtype=torch.int64, device=dev)
    print("micro-tile generic recurrent rollout, %s %s, n %d, m %d, h %d" % (cell, hidden, n, m, h))
    slots, names = [0], []
    for l in range(len(hidden)):
        slots.append(6 + 3 * l); names.append("L%d product 0" % l)
        if cell == "gru":
            if l == len(hidden) - 1:
                slots.append(15); names.append("r*h + barrier")
            slots.append(7 + 3 * l); names.append("L%d product 1" % l if l == len(hidden) - 1 else "L%d r*h + barrier + product 1" % l)
        slots.append(8 + 3 * l); names.append("L%d gates" % l)
    slots += [1, 2, 3, 4, 5]
    names += ["(layer barrier)", "out", "barrier", "reduce+reward", "next x"]
    run(lambda: native.plan_rs(obs0, c0, h0, a, m, n, h, 1.0, env.reward_spec, best_key=best), h, names, slots)
else:
    cid = sys.argv[2] if len(sys.argv) > 2 else "c3b_ant_rs_n500_h10_pb5_3x512"
    case = dict(cases.CASES[cid])
    if len(sys.argv) > 3:
        case["n"] = int(sys.argv[3])
    if len(sys.argv) > 4:
        case["m"] = int(sys.argv[4])
    if len(sys.argv) > 5:
        case["h"] = int(sys.argv[5])
    env, model = cases.product_model(case)
    native = model.planner_model()
    dev = native.device
    m, n, h = case["m"], case["n"], case["h"]
    od, ad = env.observation_space.shape[0], env.action_space.shape[0]
    obs0 = torch.randn((m, od), device=dev)
    a = torch.rand((h, m * n, ad), device=dev) * 2 - 1
    best = torch.zeros((m,), dtype=torch.int64, device=dev)
    print("micro-tile MLP rollout, %s hidden %s mode %s E %d, n %d, m %d, h %d" % (cid, case["hidden"], case["mode"], case["E"], n, m, h))
    run(lambda: native.plan_rs(obs0, a, m, n, h, 1.0, env.reward_spec, best_key=best), h, NAMES_MLP, MLP_SLOTS)
