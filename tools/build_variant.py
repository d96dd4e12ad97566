#!/usr/bin/env python
"""Developer aid for kernel A/B runs on ONE GPU box: build `learning_to_adapt_amd/libl2a_hip_<name>.so` whose
MFMA instance (NT, TPW) = (1, 8) - the HalfCheetah / Ant 512-wide kernel - is compiled with extra -D flags; every
other object is reused from the regular build.  Select it at run time with L2A_LIB_PATH.

    python tools/build_variant.py nots -DL2A_NO_TS
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from learning_to_adapt_amd.csrc import build as b  # noqa: E402


def main():
    name, defs = sys.argv[1], [a for a in sys.argv[2:] if a != "--no-base"]
    if "--no-base" not in sys.argv:         # (developer shortcut: reuse the cached objects of the other units as they are)
        b.build()
    obj = os.path.join(b.OBJ_DIR, "l2a_mfma_1_8_%s.o" % name)
    subprocess.check_call([b._hipcc()] + b.FLAGS + b.KERNEL_FLAGS + ["-DL2A_INST_NT=1", "-DL2A_INST_TPW=8"] + defs +
                          ["-c", os.path.join(b.HERE, "l2a_mfma_inst.hip"), "-o", obj], cwd=b.HERE)
    objs = [os.path.join(b.OBJ_DIR, "l2a_api.o"), os.path.join(b.OBJ_DIR, "l2a_lstm_api.o"),
            os.path.join(b.OBJ_DIR, "l2a_comm.o"), os.path.join(b.OBJ_DIR, "l2a_cem.o")]
    objs += [obj if i == (1, 8) else os.path.join(b.OBJ_DIR, "l2a_mfma_%d_%d.o" % i) for i in b.INSTANCES]
    for u in b.LSTM_INSTANCES:
        lobj = os.path.join(b.OBJ_DIR, "l2a_lstm_%d.o" % u)
        if u == 4 and any(("L2A_TIMELINE" in d or "L2A_LSTM" in d) for d in defs):     # units 256: the recurrent timeline tool's shape
            lobj = os.path.join(b.OBJ_DIR, "l2a_lstm_%d_%s.o" % (u, name))
            subprocess.check_call([b._hipcc()] + b.FLAGS + b.LSTM_FLAGS + ["-DL2A_INST_UTW=%d" % u] + defs +
                                  ["-c", os.path.join(b.HERE, "l2a_lstm_inst.hip"), "-o", lobj], cwd=b.HERE)
        objs.append(lobj)
    out = os.path.join(b.PKG, "libl2a_hip_%s.so" % name)
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-ldl"], cwd=b.HERE)
    print(out)


if __name__ == "__main__":
    main()
