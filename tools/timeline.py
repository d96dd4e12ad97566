#!/usr/bin/env python
"""Phase timeline of the MFMA rollout kernel on config 2 (developer aid, needs a GPU).

Stamps (shader clock, wave 0 of candidate tile 0, both member groups): per (step, set)
0 start | 1 layer 0 done | 2 past barrier | 3 last hidden GEMM + epilogue done |
4 output partials written | 5 past barrier | 6 partials reduced ; slot 7 of set 7 = end of step.
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

case = cases.CASES["c2_hc_rs_n2000_h30_e5"]
env, model = cases.product_model(case)
native = model.planner_model()
dev = native.device
gold = cases.load_golden("c2_hc_rs_n2000_h30_e5_s0")
obs0 = torch.from_numpy(gold["obs0"].astype(np.float32)).to(dev)
a = (torch.rand((30, 2000, 6), device=dev) * 2 - 1)
best = torch.zeros(1, dtype=torch.int64, device=dev)
h = 30
for _ in range(3):
    native.plan_rs(obs0, a, 1, 2000, h, 1.0, env.reward_spec, best_key=best)
torch.cuda.synchronize()
dbg = torch.zeros((2, h, 8, 8, 16), dtype=torch.int64, device=dev)
native.ctx.check(native.lib.l2a_set_debug_buffer(native.ctx.handle, ctypes.c_void_p(dbg.data_ptr())), "dbg")
native.plan_rs(obs0, a, 1, 2000, h, 1.0, env.reward_spec, best_key=best)
torch.cuda.synchronize()
native.ctx.check(native.lib.l2a_set_debug_buffer(native.ctx.handle, ctypes.c_void_p(0)), "dbg")
d = dbg.cpu().numpy().astype(np.int64)       # [grp, t, e, wave, slot]
names = ["L0", "bar0", "hidden", "out", "bar1", "reduce"]
for grp, members in ((0, (0, 1, 2)), (1, (3, 4))):
    print("group %d members %s" % (grp, members))
    for e in members:
        print("  set %d (median over steps 2.., clocks):" % e)
        for w in range(8):
            seg = np.diff(d[grp, 2:, e, w, :7], axis=1)
            med = np.median(seg, axis=0)
            # arrival offsets relative to wave 0 at the start of the set and at the two barriers
            off0 = np.median(d[grp, 2:, e, w, 0] - d[grp, 2:, e, 0, 0])
            off4 = np.median(d[grp, 2:, e, w, 4] - d[grp, 2:, e, 0, 4])
            sub = [np.median(d[grp, 2:, e, w, 8] - d[grp, 2:, e, w, 0]), np.median(d[grp, 2:, e, w, 9] - d[grp, 2:, e, w, 8]),
                   np.median(d[grp, 2:, e, w, 10] - d[grp, 2:, e, w, 9]), np.median(d[grp, 2:, e, w, 1] - d[grp, 2:, e, w, 10])]
            if w >= 4 and d[grp, 2:, e, w, 0].max() == 0:
                continue
            print("    wave %d: " % w + "  ".join("%s %6.0f" % (n, v) for n, v in zip(names, med)) +
                  "  | start vs w0 %+6.0f  reach bar1 vs w0 %+6.0f | L0: wait-pfL0 %5.0f  mfma %5.0f  epilogue %5.0f  lds-write %5.0f"
                  % (off0, off4, sub[0], sub[1], sub[2], sub[3]))
    step = np.diff(d[grp, :, 7, 0, 7])
    last_end = d[grp, :, 7, 0, 7] - d[grp, :, members[-1], 0, 6]
    print("  step period: median %d clk (min %d max %d); set-loop-end -> step end (exchange+reward) %d clk"
          % (np.median(step), step.min(), step.max(), np.median(last_end)))

print("\nper-step schedule, wave 0 (clocks since the step began; median over steps 2..):")
for grp in (0, 1):
    sets = [e for e in range(8) if d[grp, 5, e, 0, 0] != 0 and e != 7]
    t_end_prev = d[grp, 1:-1, 7, 0, 7]                      # end of the previous step
    line = []
    for e in sets:
        for slot, nm in ((0, "start"), (1, "L0"), (3, "gemm"), (4, "out"), (6, "red")):
            line.append("s%d.%s %6.0f" % (e, nm, np.median(d[grp, 2:, e, 0, slot] - t_end_prev)))
    line.append("step.end %6.0f" % np.median(d[grp, 2:, 7, 0, 7] - t_end_prev))
    print("  group %d: " % grp + "  ".join(line))

print("\nexchange (wave 0, clocks since the last set's reduce ended; median over steps 2..):")
for grp in (0, 1):
    sets = [e for e in range(8) if d[grp, 5, e, 0, 0] != 0 and e != 7]
    base = d[grp, 2:, sets[-1], 0, 6]
    x = d[grp, 2:, 7, 0, :]
    print("  group %d: enter %5.0f  published %5.0f  swept %5.0f (sweeps: median %d max %d)  wave0@barrier %5.0f  past barrier %5.0f  step end %5.0f"
          % (grp, np.median(x[:, 9] - base), np.median(x[:, 11] - base), np.median(x[:, 12] - base),
             np.median(x[:, 14]) + 1, x[:, 14].max() + 1, np.median(x[:, 10] - base), np.median(x[:, 13] - base),
             np.median(x[:, 7] - base)))
    w = d[grp, 2:, 7, :4, 10] - base[:, None]
    print("           waves reach the post-exchange barrier at", np.median(w, axis=0).astype(int).tolist())
