"""Build libl2a_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python learning_to_adapt_amd/csrc/build.py [--force]
"""

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.path.join(PKG, "libl2a_hip.so")
SOURCES = ["l2a_api.hip"]
HEADERS = ["l2a_kernels.h", "l2a_mfma.h", os.path.join("..", "..", "include", "l2a.h")]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (looked on PATH and in /opt/rocm/bin)")
    return exe


def up_to_date():
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    deps = [os.path.join(HERE, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force=False, verbose=True):
    if not force and up_to_date():
        if verbose:
            print("[l2a] %s is up to date" % OUT)
        return OUT
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-o", OUT] + [os.path.join(HERE, s) for s in SOURCES]
    if verbose:
        print("[l2a] " + " ".join(cmd))
    subprocess.check_call(cmd, cwd=HERE)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
