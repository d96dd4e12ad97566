#!/usr/bin/env python
"""Kernel-only timings of the main plan shapes with whatever library L2A_LIB_PATH selects (A/B of kernel variants
built by tools/build_variant.py).  One JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import cases  # noqa: E402
import bench_configs as bc  # noqa: E402


def main():
    out = {"lib": os.path.basename(os.environ.get("L2A_LIB_PATH", "libl2a_hip.so"))}
    for name in ("c2_hc_rs_n2000_h30_e5", "c3_ant_rs_n2000_h20_pb5", "c3b_ant_rs_n500_h10_pb5_3x512",
                 "c1_hc_rs_n500_h10_e1", "c4_hc_rs_n16000_h30_e5"):
        case = cases.CASES[name]
        env, model = cases.product_model(case)
        ms = min(bc.time_plan(model.planner_model(), case, env, reps=40) for _ in range(3))
        out[name.split("_")[0]] = {"ms": round(ms, 4), "frac": round(bc.flops(case, env) / ms / 1e9 / bc.PEAK, 4)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
