"""Build libl2a_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python learning_to_adapt_amd/csrc/build.py [--force]

The MFMA kernel template is instantiated in six translation units (one per (NT, TPW) pair), the
LSTM kernel in three (one per units / 64); all are compiled in parallel and linked with the two API units.
"""

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.path.join(PKG, "libl2a_hip.so")
RNG_OUT = os.path.join(PKG, "libl2a_rng.so")        # host-only helper (gcc), see l2a_rng.c
OBJ_DIR = os.path.join(HERE, "_obj")
HEADERS = ["l2a_host.h", "l2a_kernels.h", "l2a_valu.h", "l2a_adapt.h", "l2a_mfma.h", "l2a_mfma_launch.h", "l2a_lstm.h",
           "l2a_lstm_valu.h", "l2a_rnn_valu.h", "l2a_rnn_mfma.h", "l2a_lstm_launch.h", "l2a_micro.h", "l2a_micro_pack.h", "l2a_micro_launch.h", os.path.join("..", "..", "include", "l2a.h")]
SOURCES = ["l2a_api.hip", "l2a_mfma_inst.hip", "l2a_lstm_api.hip", "l2a_lstm_inst.hip", "l2a_micro_inst.hip", "l2a_comm.hip", "l2a_cem.hip", "l2a_rng.c"]
INSTANCES = [(1, 2), (1, 4), (1, 8), (2, 2), (2, 4), (2, 8)]
LSTM_INSTANCES = [2, 4, 8]          # UTW = units / 64
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
# Rollout kernels: MFMA accumulators in architectural VGPRs where the allocator can afford it - every epilogue reads its
# accumulators with VALU instructions, which cannot address AGPRs, and each v_accvgpr_read costs issue time between MFMAs
# (536 -> 344 of them in the HalfCheetah instance; config 2 +0.35 %, config 3 +1 %, config 3b +2 %: profiles/r03_ab_kernel_variants.jsonl).
# NT = 1 instances only: the NT = 2 instances (config 4 on one GPU) LOSE 2.5 % with it (10.44 -> 10.76 ms, same file), and the
# recurrent kernels get more accumulator moves, not fewer (70 -> 231 in the U = 256 instance).
KERNEL_FLAGS = ["-mllvm", "--amdgpu-mfma-vgpr-form"]
LSTM_FLAGS = []


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (looked on PATH and in /opt/rocm/bin)")
    return exe


def build_rng(verbose=True):
    """The host RNG helper: plain C, built with gcc.  Optional - without it parity mode uses NumPy's own loop."""
    gcc = shutil.which("gcc")
    if gcc is None:
        if verbose:
            print("[l2a] gcc not found: skipping libl2a_rng.so (parity mode falls back to np.random.random_sample)")
        return None
    # -ffp-contract=off: NumPy's baseline build has no FMA; a contracted x1*x1 + x2*x2 would change bits
    subprocess.check_call([gcc, "-O3", "-fPIC", "-shared", "-ffp-contract=off", "-pthread",
                           os.path.join(HERE, "l2a_rng.c"), "-o", RNG_OUT, "-lm"])
    return RNG_OUT


def up_to_date():
    if not os.path.exists(OUT) or (shutil.which("gcc") and not os.path.exists(RNG_OUT)):
        return False
    t = os.path.getmtime(OUT)
    deps = [os.path.join(HERE, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return all(os.path.getmtime(d) <= t for d in deps)


def _compile(job):
    src, obj, defs = job
    cmd = [_hipcc()] + FLAGS + defs + ["-c", os.path.join(HERE, src), "-o", obj]
    subprocess.check_call(cmd, cwd=HERE)
    return obj


def relink(units, verbose=True):
    """Developer shortcut: recompile the named translation units only (``l2a_micro_inst.hip`` ...) and link with the other
    cached objects."""
    table = {"l2a_api.hip": ("l2a_api.o", []), "l2a_lstm_api.hip": ("l2a_lstm_api.o", KERNEL_FLAGS), "l2a_comm.hip": ("l2a_comm.o", []),
             "l2a_cem.hip": ("l2a_cem.o", []), "l2a_micro_inst.hip": ("l2a_micro.o", [])}
    jobs = [(u, os.path.join(OBJ_DIR, table[u][0]), table[u][1]) for u in units]
    with ThreadPoolExecutor(max_workers=max(1, len(jobs))) as pool:
        list(pool.map(_compile, jobs))
    objs = [os.path.join(OBJ_DIR, o) for o in ("l2a_api.o", "l2a_lstm_api.o", "l2a_comm.o", "l2a_cem.o", "l2a_micro.o")]
    objs += [os.path.join(OBJ_DIR, "l2a_mfma_%d_%d.o" % i) for i in INSTANCES]
    objs += [os.path.join(OBJ_DIR, "l2a_lstm_%d.o" % u) for u in LSTM_INSTANCES]
    subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-ldl"], cwd=HERE)
    return OUT


def build(force=False, verbose=True, only=None):
    """``only``: optional list of (NT, TPW) pairs to (re)compile - the others reuse their cached
    objects (developer shortcut; a clean build compiles all six)."""
    if not force and only is None and up_to_date():
        if verbose:
            print("[l2a] %s is up to date" % OUT)
        return OUT
    os.makedirs(OBJ_DIR, exist_ok=True)
    jobs = [("l2a_api.hip", os.path.join(OBJ_DIR, "l2a_api.o"), []),
            # (holds the generic recurrent matrix-core kernel, l2a_rnn_mfma.h: its small accumulator tiles are read by the gate
            # arithmetic right away - in architectural VGPRs that needs no v_accvgpr moves, 590 of them otherwise)
            ("l2a_lstm_api.hip", os.path.join(OBJ_DIR, "l2a_lstm_api.o"), KERNEL_FLAGS),
            ("l2a_comm.hip", os.path.join(OBJ_DIR, "l2a_comm.o"), []),
            ("l2a_cem.hip", os.path.join(OBJ_DIR, "l2a_cem.o"), []),
            ("l2a_micro_inst.hip", os.path.join(OBJ_DIR, "l2a_micro.o"), [])]
    for utw in LSTM_INSTANCES:
        jobs.append(("l2a_lstm_inst.hip", os.path.join(OBJ_DIR, "l2a_lstm_%d.o" % utw), ["-DL2A_INST_UTW=%d" % utw] + LSTM_FLAGS))
    for nt, tpw in INSTANCES:
        obj = os.path.join(OBJ_DIR, "l2a_mfma_%d_%d.o" % (nt, tpw))
        if only is not None and (nt, tpw) not in only and os.path.exists(obj):
            continue
        jobs.append(("l2a_mfma_inst.hip", obj, ["-DL2A_INST_NT=%d" % nt, "-DL2A_INST_TPW=%d" % tpw] + (KERNEL_FLAGS if nt == 1 else [])))
    if verbose:
        print("[l2a] hipcc %s : %d translation units, %d parallel jobs"
              % (" ".join(FLAGS), len(jobs), min(len(jobs), os.cpu_count() or 1)))
    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as pool:
        list(pool.map(_compile, jobs))
    objs = [os.path.join(OBJ_DIR, "l2a_api.o"), os.path.join(OBJ_DIR, "l2a_lstm_api.o"), os.path.join(OBJ_DIR, "l2a_comm.o"),
            os.path.join(OBJ_DIR, "l2a_cem.o"), os.path.join(OBJ_DIR, "l2a_micro.o")]
    objs += [os.path.join(OBJ_DIR, "l2a_mfma_%d_%d.o" % i) for i in INSTANCES]
    objs += [os.path.join(OBJ_DIR, "l2a_lstm_%d.o" % u) for u in LSTM_INSTANCES]
    subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-ldl"], cwd=HERE)
    build_rng(verbose)
    return OUT


if __name__ == "__main__":
    only = None
    for a in sys.argv[1:]:
        if a.startswith("--units="):
            relink(a[len("--units="):].split(","))
            sys.exit(0)
        if a.startswith("--only="):
            only = [tuple(int(x) for x in part.split("x")) for part in a[len("--only="):].split(",")]
    build(force="--force" in sys.argv, only=only)
