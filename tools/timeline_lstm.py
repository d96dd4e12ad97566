#!/usr/bin/env python
"""Phase timeline of the MFMA LSTM rollout kernel (developer aid, needs a GPU).

Stamps (shader clock, the four waves of workgroup 0) per step: 0 start | 1 x-part done | 2 h-part
(all passes) done | 3 gates done | 4 output partials written | 5 past barrier | 6 end of step.
"""
import ctypes
import os
import sys

# the phase stamps exist only in a library built with -DL2A_TIMELINE (they cost the product kernel 0.6 %):
#   python tools/build_variant.py timeline -DL2A_TIMELINE
_TL = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "learning_to_adapt_amd",
                   "libl2a_hip_timeline.so")
if not os.path.exists(_TL):
    raise SystemExit("build the timeline library first: python tools/build_variant.py timeline -DL2A_TIMELINE")
os.environ["L2A_LIB_PATH"] = _TL

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402

units = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
case = dict(cases.CASES["c6_hc_rnn_rs_n500_h10_m5"], units=units)
env, model = cases.product_rnn_model(case)
native = model.planner_model()
dev = native.device
h, m = 30, 1
obs0 = torch.randn((m, 20), device=dev)
c0 = torch.randn((m, units), device=dev)
h0 = torch.tanh(torch.randn((m, units), device=dev))
a = torch.rand((h, m * n, 6), device=dev) * 2 - 1
best = torch.zeros(m, dtype=torch.int64, device=dev)
for _ in range(3):
    native.plan_rs(obs0, c0, h0, a, m, n, h, 1.0, env.reward_spec, best_key=best)
torch.cuda.synchronize()
dbg = torch.zeros((h, 4, 16), dtype=torch.int64, device=dev)
native.ctx.check(native.lib.l2a_set_debug_buffer(native.ctx.handle, ctypes.c_void_p(dbg.data_ptr())), "dbg")
native.plan_rs(obs0, c0, h0, a, m, n, h, 1.0, env.reward_spec, best_key=best)
torch.cuda.synchronize()
native.ctx.check(native.lib.l2a_set_debug_buffer(native.ctx.handle, ctypes.c_void_p(0)), "dbg")
d = dbg.cpu().numpy().astype(np.int64)
names = ["h own half", "poll+h other half", "deferred tail + x part", "gates", "out", "barrier (+S publish)"]
print("units %d, n %d: median clocks per phase over steps 2.. (s_memtime ticks; 100 MHz -> x24 for shader clocks at 2.4 GHz)" % (units, n))
for w in range(4):
    seg = np.median(np.diff(d[2:, w, :7], axis=1), axis=0)
    line = "  wave %d: " % w + "  ".join("%s %6.0f" % (nm, v) for nm, v in zip(names, seg))
    if d[2:, w, 7].any():       # unit-tile split: the wait for the partner's half of h and the barrier behind it
        line += "  | partner-h poll %6.0f  barrier %6.0f" % (np.median(d[2:, w, 7] - d[2:, w, 1]),
                                                             np.median(d[2:, w, 8] - d[2:, w, 7]))
    print(line)
step = np.diff(d[:, 0, 0])
print("  step period: median %d ticks (min %d max %d)" % (np.median(step), step.min(), step.max()))
