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
n_tiles = m * ((n + 15) // 16)
dbg_all = torch.zeros(h * 4 * 16 + 2 * n_tiles * 6, dtype=torch.int64, device=dev)
dbg = dbg_all[:h * 4 * 16].view(h, 4, 16)
native.ctx.check(native.lib.l2a_set_debug_buffer(native.ctx.handle, ctypes.c_void_p(dbg.data_ptr())), "dbg")
for _ in range(int(os.environ.get('L2A_TL_WARM', '20'))):     # clocks up; the stamps of the last launch stay
    native.plan_rs(obs0, c0, h0, a, m, n, h, 1.0, env.reward_spec, best_key=best)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
native.plan_rs(obs0, c0, h0, a, m, n, h, 1.0, env.reward_spec, best_key=best)
ev1.record()
torch.cuda.synchronize()
launch_ms = ev0.elapsed_time(ev1)
native.ctx.check(native.lib.l2a_set_debug_buffer(native.ctx.handle, ctypes.c_void_p(0)), "dbg")
d = dbg.cpu().numpy().astype(np.int64)
wg = dbg_all[h * 4 * 16:].cpu().numpy().astype(np.int64).reshape(2, n_tiles, 6)
ran_wg = wg[:, :, 1] != 0
if ran_wg.any():
    life = np.where(ran_wg, wg[:, :, 1] - wg[:, :, 0], 0)
    real = (wg[:, :, 5] - wg[:, :, 4])[ran_wg]          # s_memrealtime: constant 100 MHz
    print("shader clock over the workgroups' lifetimes (s_memtime ticks per 10 ns of s_memrealtime): median %.1f MHz; "
          "launch = first start -> last end by real time: %.4f ms"
          % (np.median(life[ran_wg] / real) * 100.0, (wg[:, :, 5][ran_wg].max() - wg[:, :, 4][ran_wg].min()) / 1e5))
    print("stamped launch by events: %.4f ms; longest workgroup lifetime %d ticks = %.3f ticks/ns if the launch were nothing else"
          % (launch_ms, life.max(), life.max() / (launch_ms * 1e6)))
    xcd = wg[:, :, 2] & 15
    t_first, t_last = wg[:, :, 0][ran_wg].min(), wg[:, :, 1][ran_wg].max()
    print("units %d, n %d: workgroup lifetimes (ticks): median %d, min %d, max %d over %d workgroups; first start -> last end %d"
          " (comparable only if the XCDs' clocks agree)" % (units, n, np.median(life[ran_wg]), life[ran_wg].min(), life[ran_wg].max(),
                                                            ran_wg.sum(), t_last - t_first))
    for g in (0, 1):
        for x in range(8):
            sel = ran_wg[g] & (xcd[g] == x)
            if sel.any():
                print("  group %d XCD %d: %3d workgroups, lifetime median %d max %d; start spread %d"
                      % (g, x, sel.sum(), np.median(life[g][sel]), life[g][sel].max(), wg[g][sel, 0].max() - wg[g][sel, 0].min()))
names = ["h own half", "poll+h other half", "deferred tail + x part", "gates", "out", "barrier (+S publish)"]
print("units %d, n %d: median clocks per phase over steps 2.. (s_memtime ticks = shader clocks)" % (units, n))
for w in range(4):
    seg = np.median(np.diff(d[2:, w, :7], axis=1), axis=0)
    line = "  wave %d: " % w + "  ".join("%s %6.0f" % (nm, v) for nm, v in zip(names, seg))
    if d[2:, w, 7].any():       # unit-tile split: the wait for the partner's half of h and the barrier behind it
        line += "  | partner-h poll %6.0f  barrier %6.0f" % (np.median(d[2:, w, 7] - d[2:, w, 1]),
                                                             np.median(d[2:, w, 8] - d[2:, w, 7]))
    print(line)
step = np.diff(d[:, 0, 0])
print("  step period: median %d ticks (min %d max %d)" % (np.median(step), step.min(), step.max()))
