#!/usr/bin/env python
"""Developer aid for kernel A/B runs on ONE GPU box: build `learning_to_adapt_amd/libl2a_hip_<name>.so` in which the named
objects (default: the HalfCheetah / Ant 512-wide MFMA instance, the 256-unit LSTM instance and the micro-tile kernels) are
compiled with extra flags; every other object is reused from the regular build.  Select it at run time with L2A_LIB_PATH.

    python tools/build_variant.py timeline -DL2A_TIMELINE
    python tools/build_variant.py exp --units=l2a_micro -DL2A_SOMETHING=1
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from learning_to_adapt_amd.csrc import build as b  # noqa: E402


def main():
    name = sys.argv[1]
    units = ["l2a_api.o", "l2a_mfma_1_8.o", "l2a_mfma_fan_1_8.o", "l2a_mfma_whole_2_8.o", "l2a_lstm_4.o", "l2a_micro.o", "l2a_rnn_micro.o"]
    defs = []
    for a in sys.argv[2:]:
        if a.startswith("--units="):
            units = [u if u.endswith(".o") else u + ".o" for u in a[len("--units="):].split(",")]
        elif a != "--no-base":
            defs.append(a)
    if "--no-base" not in sys.argv:         # (developer shortcut: reuse the cached objects of the other units as they are)
        b.build()
    print(b.variant(name, units, defs))


if __name__ == "__main__":
    main()
