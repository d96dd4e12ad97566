#!/usr/bin/env python
"""Phase timeline of the GrBAL adaptation launches (csrc/l2a_adapt.h; developer aid, needs a GPU and
`python tools/build_variant.py timeline --units=l2a_api,l2a_mfma_1_8,l2a_lstm_4,l2a_micro,l2a_rnn_micro -DL2A_TIMELINE`).
Workgroup (0, 0) of every launch stamps the shader clock: 0 entry | 1 operands requested | 2 MFMAs done | 3 reduced |
4 results stored; slots 6 / 7 hold the 100 MHz constant clock at entry / end, which places the launches on one time axis:
`gap` = end of the previous launch's workgroup (0, 0) -> entry of this one's (the kernel boundary as a workgroup sees it).

    python tools/timeline_adapt.py [cold]       cold: a 3x512 per-block plan (the GrBAL default) runs between two adaptations
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_TL = os.path.join(ROOT, "learning_to_adapt_amd", "libl2a_hip_timeline.so")
if not os.path.exists(_TL):
    raise SystemExit("build the timeline library first (see the docstring)")
os.environ["L2A_LIB_PATH"] = _TL

import numpy as np   # noqa: E402
import torch   # noqa: E402

sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from learning_to_adapt_amd import _lib  # noqa: E402
from learning_to_adapt_amd.dynamics.native_model import NativeModel  # noqa: E402
from learning_to_adapt_amd.envs import RewardSpec  # noqa: E402
from learning_to_adapt_amd.utils import synthetic  # noqa: E402

cold = len(sys.argv) > 1 and sys.argv[1] == "cold"
od, ad, hidden, m, rows = 41, 8, (512, 512, 512), 5, 16
dev = torch.device("cuda:0")
base = [torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32)).to(dev) for w in synthetic.make_weight_set(od, ad, list(hidden), 1000)]
nm = NativeModel(od, ad, hidden, "relu", None, m, "per_block")
rs = np.random.RandomState(0)
xd = torch.from_numpy(rs.randn(m, rows, od + ad).astype(np.float32)).to(dev)
yd = torch.from_numpy(rs.randn(m, rows, od).astype(np.float32)).to(dev)
low, high = -150 * np.ones(ad), 150 * np.ones(ad)
for e in range(m):
    nm.set_norm(e, synthetic.make_norm(od, ad, low, high, 2000))
n, h = 500, 10
obs0 = torch.randn((m, od), device=dev)
acts = (torch.rand((h, m * n, ad), device=dev) * 300 - 150)
best = torch.zeros((m,), dtype=torch.int64, device=dev)
spec = RewardSpec.ant(od, 0.02)


def step():
    nm.adapt_sgd(base, xd, yd, 0.01)
    if cold:
        nm.plan_rs(obs0, acts, m, n, h, 1.0, spec, best_key=best)


ctx = _lib.Context.get(0)
for _ in range(300):
    step()
torch.cuda.synchronize()
dbg = torch.zeros(8 * 8 * 8 + 64, dtype=torch.int64, device=dev)
ctx.check(ctx.lib.l2a_set_debug_buffer(ctx.handle, ctypes.c_void_p(dbg.data_ptr())), "dbg")
nm.adapt_sgd(base, xd, yd, 0.01)
torch.cuda.synchronize()
ctx.check(ctx.lib.l2a_set_debug_buffer(ctx.handle, ctypes.c_void_p(0)), "dbg")
d = dbg[:512].view(8, 8, 8).cpu().numpy().astype(np.int64)
names = ["fwd0 + fwd1", "fwd1", "fwd2", "fwd3 (out, dZ_L)", "fwd3 + bwd3 + upd3", "bwd2 | upd2", "bwd1 | upd1, upd0", "-"]
print("adaptation, 5 tasks x 16 rows, 3 x 512%s: workgroup (0, 0), wave 0; clocks (us by the 100 MHz clock)" % (", plan between adaptations" if cold else ", back to back"))
t_first = d[0, 0, 6]
prev_end = None
for ph in range(8):
    w = d[ph, 0]
    if not w[0]:
        continue
    us = (w[7] - w[6]) / 100.0
    clk = w[4] - w[0]
    seg = "entry->operands requested %5d | ->MFMAs done %5d | ->reduced %5d | ->stored %5d" % (w[1] - w[0], w[2] - w[1], w[3] - w[2], w[4] - w[3]) \
        if w[1] else "entry->end %6d" % clk
    gap = "" if prev_end is None else "  gap %.2f us" % ((w[6] - prev_end) / 100.0)
    print("  %-18s start %6.2f us  body %5.2f us = %6d clk (%.2f GHz)  %s%s" % (names[ph], (w[6] - t_first) / 100.0, us, clk,
                                                                              clk / max(us, 1e-9) / 1e3, seg, gap))
    prev_end = w[7]
    late = [int(d[ph, k, 0] - w[0]) for k in range(1, 8) if d[ph, k, 0]]
    if late:
        print("  %-18s other waves enter %s clocks after wave 0" % ("", late))
print("  first entry -> last end: %.2f us" % ((prev_end - t_first) / 100.0))
