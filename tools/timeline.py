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
_TL = os.environ.get("L2A_TIMELINE_LIB") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                          "learning_to_adapt_amd", "libl2a_hip_timeline.so")
if not os.path.exists(_TL):
    raise SystemExit("build the timeline library first: python tools/build_variant.py timeline -DL2A_TIMELINE")
os.environ["L2A_LIB_PATH"] = _TL

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c2_hc_rs_n2000_h30_e5"          # any MLP case of tests/golden/cases.json
case = dict(cases.CASES[name])
for kv in sys.argv[2:]:                 # overrides, e.g. `c5_hc_cem_n4000_h30_e5 n=500` = one rank's shard of config 5 (member fan)
    k, v = kv.split("=")
    case[k] = int(v)
env, model = cases.product_model(case)
native = model.planner_model()
dev = native.device
m, n, h = case["m"], case["n"], case["h"]
od, ad = env.observation_space.shape[0], env.action_space.shape[0]
obs0 = torch.randn((m, od), device=dev) * 0.3
a = (torch.rand((h, m * n, ad), device=dev) * 2 - 1)
best = torch.zeros(m, dtype=torch.int64, device=dev)
for _ in range(3):
    native.plan_rs(obs0, a, m, n, h, 1.0, env.reward_spec, best_key=best)
torch.cuda.synchronize()
n_tiles = m * ((n + 15) // 16)
dbg_all = torch.zeros(2 * h * 8 * 8 * 16 + 2 * n_tiles * 6, dtype=torch.int64, device=dev)
dbg = dbg_all[:2 * h * 8 * 8 * 16].view(2, h, 8, 8, 16)
native.ctx.check(native.lib.l2a_set_debug_buffer(native.ctx.handle, ctypes.c_void_p(dbg.data_ptr())), "dbg")
for _ in range(int(os.environ.get('L2A_TL_WARM', '20'))):     # clocks up; the stamps of the last launch stay
    native.plan_rs(obs0, a, m, n, h, 1.0, env.reward_spec, best_key=best)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
native.plan_rs(obs0, a, m, n, h, 1.0, env.reward_spec, best_key=best)
ev1.record()
torch.cuda.synchronize()
launch_ms = ev0.elapsed_time(ev1)
native.ctx.check(native.lib.l2a_set_debug_buffer(native.ctx.handle, ctypes.c_void_p(0)), "dbg")
d = dbg.cpu().numpy().astype(np.int64)       # [grp, t, e, wave, slot]
# [grp, t, e, wave, slot]; batched flow (l2a_mfma.h): per step  L0(s0) L0(s1) .. | barrier | GEMM+out(s0) GEMM+out(s1) .. |
# barrier | reduce(s0) reduce(s1) .. | exchange.  Slots of a set: 0 L0 start, 8 operands requested, 9 MFMAs done,
# 10 bias / activation done, 1 written to LDS; 2 GEMM start, 3 GEMM + epilogue done, 4 output partials written;
# 5 reduce start, 6 reduce done.
print("case %s %s" % (name, " ".join(sys.argv[2:])))
WGREC_ONLY = "wgrec" in os.path.basename(_TL)        # built with -DL2A_WGREC: the per-workgroup record without the phase stamps
# per-workgroup record (NT = 1 shapes): start, end (own CU's clock), XCC_ID register, hardware workgroup id
wg = dbg_all[2 * h * 8 * 8 * 16:].cpu().numpy().astype(np.int64).reshape(2, n_tiles, 6)
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
    print("workgroup lifetimes (clocks): median %d, min %d, max %d over %d workgroups"
          % (np.median(life[ran_wg]), life[ran_wg].min(), life[ran_wg].max(), ran_wg.sum()))
    for g in (0, 1):
        for x in range(8):
            sel = ran_wg[g] & (xcd[g] == x)
            if sel.any():
                print("  group %d XCD %d: %3d workgroups, lifetime median %d max %d (hw ids %d..%d)"
                      % (g, x, sel.sum(), np.median(life[g][sel]), life[g][sel].max(), wg[g][sel, 3].min(), wg[g][sel, 3].max()))
    slow = np.argsort(-(life * ran_wg).ravel())[:6]
    print("  slowest: " + ", ".join("g%d tile %d xcd %d hw %d: %d" % (i // n_tiles, i % n_tiles, xcd.ravel()[i], wg.reshape(-1, 6)[i, 3],
                                                                     life.ravel()[i]) for i in slow))
if ran_wg.any():
    # s_memrealtime is one device-wide 100 MHz counter: starts and ends of all workgroups on one axis (10 ns resolution)
    st = (wg[:, :, 4] - wg[:, :, 4][ran_wg].min()) / 100.0
    en = (wg[:, :, 5] - wg[:, :, 4][ran_wg].min()) / 100.0
    q = lambda a: "min %.1f / 10%% %.1f / median %.1f / 90%% %.1f / max %.1f" % tuple(np.percentile(a, [0, 10, 50, 90, 100]))  # noqa: E731
    print("workgroup starts after the first one (us): " + q(st[ran_wg]))
    print("workgroup ends after the first start (us):  " + q(en[ran_wg]))
    print("workgroup durations (us):                   " + q((en - st)[ran_wg]))
    for g in (0, 1):
        for x in range(8):
            sel = ran_wg[g] & (xcd[g] == x)
            if sel.any():
                print("  group %d XCD %d: start median %.1f max %.1f | end median %.1f max %.1f | duration median %.1f us"
                      % (g, x, np.median(st[g][sel]), st[g][sel].max(), np.median(en[g][sel]), en[g][sel].max(),
                         np.median((en - st)[g][sel])))
if WGREC_ONLY:
    raise SystemExit(0)
t_mid = min(5, h - 1)
GROUPS = []
for grp in (0, 1):
    ran = [e for e in range(7) if d[grp, t_mid, e, 0, 0] != 0]
    if ran:
        GROUPS.append((grp, tuple(sorted(ran, key=lambda e: d[grp, t_mid, e, 0, 0]))))
for grp, members in GROUPS:
    print("group %d sets %s (median over steps 2.., clocks)" % (grp, members))
    for e in members:
        for w in range(4):
            x = d[grp, 2:, e, w, :]
            if x[:, 0].max() == 0:
                continue
            med = lambda a, b: np.median(x[:, a] - x[:, b])      # noqa: E731
            print("  set %d wave %d: L0 %5.0f (issue %4.0f mfma %5.0f epilogue %4.0f lds-write %4.0f)  inner layers %6.0f  last gemm %6.0f  out %5.0f  reduce %5.0f"
                  % (e, w, med(1, 0), med(8, 0), med(9, 8), med(10, 9), med(1, 10), med(15, 2), med(3, 15), med(4, 3), med(6, 5)))
    step = np.diff(d[grp, :, 7, 0, 7])
    print("  step period: median %d clk (min %d max %d)" % (np.median(step), step.min(), step.max()))

print("\nper-step schedule, wave 0 (clocks since the previous step ended; median over steps 2..):")
for grp, members in GROUPS:
    t_end_prev = d[grp, 1:-1, 7, 0, 7]
    line = []
    for slot, nm in ((0, "L0"), (1, "L0.end"), (2, "gemm"), (3, "gemm.end"), (4, "out.end"), (5, "red"), (6, "red.end")):
        for e in members:
            line.append("s%d.%s %6.0f" % (e, nm, np.median(d[grp, 2:, e, 0, slot] - t_end_prev)))
    x = d[grp, 2:, 7, 0, :]
    for slot, nm in ((9, "sets.done"), (11, "published"), (12, "swept"), (10, "at.barrier"), (13, "past.barrier"), (7, "step.end")):
        if slot == 12 and x[:, 12].max() == 0 and d[grp, 2:, 7, 3, 12].max() != 0:
            # early exchange (round 6): wave 3 requests the partner's records in phase C and holds them after the hand-over
            line.append("swept(wave 3) %6.0f" % np.median(d[grp, 2:, 7, 3, 12] - t_end_prev))
            continue
        line.append("%s %6.0f" % (nm, np.median(x[:, slot] - t_end_prev)))
    print("  group %d: " % grp + "  ".join(line))
    print("           sweeps: median %d max %d; waves reach the post-exchange barrier at %s"
          % (np.median(x[:, 14]) + 1, x[:, 14].max() + 1,
             np.median(d[grp, 2:, 7, :4, 10] - t_end_prev[:, None], axis=0).astype(int).tolist()))

if len(GROUPS) == 2:    # are the two workgroups' clocks comparable?  (s_memtime across CUs / XCDs)
    d0, d1 = d[0, 2:, 7, 0, 7], d[1, 2:, 7, 0, 7]
    diff = d1 - d0
    print("\ngroup 1 step end minus group 0 step end (same step, raw stamps): median %d min %d max %d"
          % (np.median(diff), diff.min(), diff.max()))
    p0, p1 = d[0, 2:, 7, 0, 11], d[1, 2:, 7, 0, 11]
    print("group 1 published minus group 0 published: median %d min %d max %d" % (np.median(p1 - p0), (p1 - p0).min(), (p1 - p0).max()))
    s0, s1 = d[0, 2:, 7, 0, 12], d[1, 2:, 7, 0, 12]
    print("group 0 swept minus group 1 published: median %d ; group 1 swept minus group 0 published: median %d"
          % (np.median(s0 - p1), np.median(s1 - p0)))
