#!/usr/bin/env python
"""50 adaptation steps (5 tasks x 16 transitions, 3 x 512) for `rocprofv3 --kernel-trace --stats` (developer aid)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from learning_to_adapt_amd.dynamics.native_model import NativeModel  # noqa: E402
from learning_to_adapt_amd.utils import synthetic  # noqa: E402

od, ad, hidden, m, rows = 41, 8, (512, 512, 512), 5, 16
dev = torch.device("cuda:0")
base = [torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32)).to(dev)
        for w in synthetic.make_weight_set(od, ad, list(hidden), 1000)]
nm = NativeModel(od, ad, hidden, "relu", None, m, "per_block")
rs = np.random.RandomState(0)
x = rs.randn(m, rows, od + ad).astype(np.float32)
y = rs.randn(m, rows, od).astype(np.float32)
for _ in range(50):
    nm.adapt_sgd_host(base, x, y, 0.01)
torch.cuda.synchronize()
