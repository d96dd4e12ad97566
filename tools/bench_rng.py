#!/usr/bin/env python
"""Host RNG helper on this machine: time per draw vs NumPy, by thread count (no GPU).  JSON lines.

    python tools/bench_rng.py
Shapes: config 2 (360 k doubles), config 4 (2.88 M doubles; one rank's shard of 1/8 kept), config 5 (720 k normals
per CEM iteration)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learning_to_adapt_amd.utils import fast_rng  # noqa: E402


def best(fn, reps=9):
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t)
    ts.sort()
    return round(1e3 * ts[0], 3), round(1e3 * ts[len(ts) // 2], 3)


def main():
    np.random.seed(0)
    ok = {k: fast_rng.available(k) for k in ("double", "uniform", "normal")}
    print(json.dumps({"verified": ok, "cpus": os.cpu_count(), "default_threads": fast_rng.threads()}), flush=True)
    low, high = -np.ones(6), np.ones(6)
    st = fast_rng.State.from_global()
    out = {"numpy_ms": {
        "uniform_360k": best(lambda: np.random.uniform(low, high, (60000, 6)), 5),
        "uniform_2.88M": best(lambda: np.random.uniform(low, high, (480000, 6)), 3),
        "normal_720k": best(lambda: np.random.normal(size=720000), 5)}}
    print(json.dumps(out), flush=True)
    f32_c2 = np.empty((60000, 6), dtype=np.float32)
    f32_c4 = np.empty((60000, 6), dtype=np.float32)
    c64 = np.empty((16000, 6))
    z = np.empty(720000)
    for T in (1, 2, 4, 8, 16):
        if T > (os.cpu_count() or 1):
            break
        fast_rng.set_threads(T)
        row = {"threads": T,
               "uniform_rows_c2_ms": best(lambda: st.uniform_rows(60000, low, high, 2000, 0, 2000, f32_c2, 2000, c64)),
               "uniform_rows_c4_shard_ms": best(lambda: st.uniform_rows(480000, low, high, 16000, 2000, 4000, f32_c4,
                                                                        16000, c64)),
               "normal_720k_ms": best(lambda: st.standard_normal(720000, z))}
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
