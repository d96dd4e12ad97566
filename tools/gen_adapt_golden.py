#!/usr/bin/env python
"""Fixtures for GrBAL's inner adaptation step (`MetaMLPDynamicsModel.adapt`, reference
`dynamics/meta_mlp_dynamics.py:321-345,96-120,409-421`).

The reference's own `adapt` needs a live TensorFlow 1.13 session (absent here - SURVEY.md 8(c)), so the expected
values are those of the line-by-line restatement `oracle/adapt.py` in FLOAT64, after it has been checked against
float64 central finite differences of the restated loss (done here before anything is written).  The fixture pins
the oracle (so it cannot drift silently) and gives the GPU tests a reference that does not depend on the package.

    python tools/gen_adapt_golden.py            # rewrites tests/golden/adapt_cases.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import adapt as oadapt                      # noqa: E402
import adapt_cases                                       # noqa: E402


def finite_difference_check(params, x, y, hid, rng, n_probe=12, eps=1e-6):
    g = oadapt.loss_gradients(params, x, y, hid, None, dtype=np.float64)
    for pi in range(len(params)):
        flat = params[pi].reshape(-1)
        for k in rng.choice(flat.size, size=min(n_probe, flat.size), replace=False):
            old = flat[k]
            flat[k] = old + eps
            lp = oadapt.pre_loss(params, x, y, hid, None, dtype=np.float64)
            flat[k] = old - eps
            lm = oadapt.pre_loss(params, x, y, hid, None, dtype=np.float64)
            flat[k] = old
            fd = (lp - lm) / (2 * eps)
            an = g[pi].reshape(-1)[k]
            assert abs(fd - an) <= 1e-6 * max(1.0, abs(an)) + 1e-9, (pi, k, fd, an)


def main():
    out = {}
    for name in adapt_cases.CASES:
        c = adapt_cases.build(name)
        params64 = [np.array(p, dtype=np.float64) for p in c["params"]]
        nn_input, delta = oadapt.build_adapt_batch(c["obs"], c["act"], c["obs_next"], c["meta_batch_size"], c["norm"])
        pre_x, pre_y = oadapt.pre_split(nn_input, delta, c["meta_batch_size"])
        rng = np.random.RandomState(0)
        for i in range(len(c["obs"])):
            finite_difference_check(params64, pre_x[i], pre_y[i], c["hidden_nonlinearity"], rng)
        sets = oadapt.adapt_sets(params64, c["obs"], c["act"], c["obs_next"], c["meta_batch_size"],
                                 c["inner_learning_rate"], c["norm"], c["hidden_nonlinearity"], None, dtype=np.float64)
        for i, s in enumerate(sets):
            for pi, (p, q) in enumerate(zip(params64, s)):
                step = q - p
                out["%s/t%d/p%d/step_sum" % (name, i, pi)] = np.float64(step.sum())
                out["%s/t%d/p%d/step_abs" % (name, i, pi)] = np.float64(np.abs(step).sum())
                out["%s/t%d/p%d/step_head" % (name, i, pi)] = step.reshape(-1)[:16].copy()
        print(name, "ok:", len(sets), "tasks")
    path = os.path.join(ROOT, "tests", "golden", "adapt_cases.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
