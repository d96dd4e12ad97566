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
NAMES_MLP = ["x rows", "sets (all phases)", "mean+reward"]


def run(plan, h, names, nslots):
    ctx = _lib.Context.get(0)
    ctx.set_micro(2)
    for _ in range(3):
        plan()
    torch.cuda.synchronize()
    dbg = torch.zeros(h * 4 * 16 + 64, dtype=torch.int64, device="cuda")
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
        seg = np.median(np.diff(d[lo:, w, :nslots], axis=1), axis=0)
        print("  wave %d: " % w + "  ".join("%s %6.0f" % (nm, v) for nm, v in zip(names, seg)))
    if d[0, 0, 8]:
        print("  prologue (wave 0): entry -> constants in LDS %d, -> state / first operands requested %d, -> first step %d clocks"
              % (d[0, 0, 9] - d[0, 0, 8], d[0, 0, 10] - d[0, 0, 9], d[0, 0, 0] - d[0, 0, 10]))
    step = np.diff(d[:, 0, 0])
    print("  step period: median %d clocks (min %d max %d); first stamp -> last stamp of the launch %d"
          % (np.median(step), step.min(), step.max(), d[:, :, :nslots].max() - d[0, :, 0].min()))
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
    run(lambda: native.plan_rs(obs0, c0, h0, a, m, n, h, 1.0, env.reward_spec, best_key=best), h, NAMES_LSTM, 7)
else:
    cid = sys.argv[2] if len(sys.argv) > 2 else "c3b_ant_grbal_n500_h10_m5"
    case = dict(cases.CASES[cid])
    if len(sys.argv) > 3:
        case["n"] = int(sys.argv[3])
    if len(sys.argv) > 4:
        case["m"] = int(sys.argv[4])
    if len(sys.argv) > 5:
        case["h"] = int(sys.argv[5])
    import bench_configs as bc  # noqa: E402,F401
    raise SystemExit("mlp timeline: see tools/ab_micro.py mlp (stamps are read there)")
