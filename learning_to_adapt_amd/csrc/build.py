"""Build libl2a_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python learning_to_adapt_amd/csrc/build.py [--force]

The MFMA kernel template is instantiated in ten translation units (one per (NT, TPW) pair, the four
member-fan units and the whole-tiles unit), the LSTM kernel in three (one per units / 64); all are compiled in parallel and linked with the two API units.
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
           "l2a_lstm_valu.h", "l2a_rnn_valu.h", "l2a_rnn_mfma.h", "l2a_lstm_launch.h", "l2a_micro.h", "l2a_rnn_micro.h", "l2a_micro_pack.h", "l2a_micro_launch.h", "l2a_rng.h", os.path.join("..", "..", "include", "l2a.h")]
SOURCES = ["l2a_api.hip", "l2a_mfma_inst.hip", "l2a_lstm_api.hip", "l2a_lstm_inst.hip", "l2a_micro_inst.hip", "l2a_rnn_micro_inst.hip", "l2a_comm.hip", "l2a_cem.hip", "l2a_step.hip", "l2a_rng.c"]
INSTANCES = [(1, 2), (1, 4), (1, 8), (2, 2), (2, 4)]
# member-fan instances of the same template (-DL2A_INST_FAN=1): NT = 1 at every width, NT = 2 at width 512 (the fan instances
# carry no half-member code: (2, 8) keeps its registers - 456 VGPRs, no scratch - where the tile-split (2, 8) instances spilled)
FAN_INSTANCES = [(1, 2), (1, 4), (1, 8), (2, 8)]
# whole-tiles-only instances (-DL2A_INST_FAN=2: neither exchange nor half-member code): two candidate tiles per workgroup at
# width 512 - the double rounds launch_rollout puts in front of a multi-round plan
WHOLE_INSTANCES = [(2, 8), (1, 8)]
LSTM_INSTANCES = [2, 4, 8]          # UTW = units / 64
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
# Rollout kernels: MFMA accumulators in architectural VGPRs where the allocator can afford it - every epilogue reads its
# accumulators with VALU instructions, which cannot address AGPRs, and each v_accvgpr_read costs issue time between MFMAs
# (536 -> 344 of them in the HalfCheetah instance; config 2 +0.35 %, config 3 +1 %, config 3b +2 %: profiles/r03_ab_kernel_variants.jsonl).
# NT = 1 instances only: the NT = 2 instances (config 4 on one GPU) LOSE 2.5 % with it (10.44 -> 10.76 ms, same file), and the
# recurrent kernels get more accumulator moves, not fewer (70 -> 231 in the U = 256 instance).
KERNEL_FLAGS = ["-mllvm", "--amdgpu-mfma-vgpr-form"]
LSTM_FLAGS = []
MICRO_FLAGS = []


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (looked on PATH and in /opt/rocm/bin)")
    return exe


def build_rng(verbose=True):
    """The host RNG helper: plain C, built with gcc.  REQUIRED since round 5: libl2a_hip.so links against it (the controller
    step's draw-ahead chain, l2a_step.hip; found next to the library at load time through rpath $ORIGIN - an L2A_LIB_PATH
    library placed elsewhere needs its libl2a_rng.so beside it).  Both sides require l2a_rng_version() >= 8."""
    gcc = shutil.which("gcc") or shutil.which("clang") or ("/opt/rocm/lib/llvm/bin/clang" if os.path.exists("/opt/rocm/lib/llvm/bin/clang") else None)
    if gcc is None:
        if verbose:
            print("[l2a] no C compiler found: libl2a_rng.so cannot be built")
        return None
    # -ffp-contract=off: NumPy's baseline build has no FMA; a contracted x1*x1 + x2*x2 would change bits
    subprocess.check_call([gcc, "-O3", "-fPIC", "-shared", "-ffp-contract=off", "-pthread",
                           os.path.join(HERE, "l2a_rng.c"), "-o", RNG_OUT, "-lm"])
    return RNG_OUT


def _obj_fresh(obj):
    """An object is fresh when it is newer than every file its compilation read (hipcc -MD writes the list next to it)
    and than this script."""
    dep = obj[:-2] + ".d"
    if not os.path.exists(obj) or not os.path.exists(dep):
        return False
    t = os.path.getmtime(obj)
    with open(dep) as f:
        words = f.read().replace("\\\n", " ").split()
    files = [w for w in words[1:] if not w.endswith(":")]
    own = [w for w in files if not w.startswith("/opt/") and not w.startswith("/usr/")]     # toolchain headers do not change under us
    return all(os.path.exists(w) and os.path.getmtime(w) <= t for w in own + [os.path.abspath(__file__)])


def up_to_date():
    if not os.path.exists(OUT) or (shutil.which("gcc") and not os.path.exists(RNG_OUT)):
        return False
    t = os.path.getmtime(OUT)
    if os.path.isdir(OBJ_DIR) and os.listdir(OBJ_DIR):
        if not all(_obj_fresh(os.path.join(OBJ_DIR, o)) and os.path.getmtime(os.path.join(OBJ_DIR, o)) <= t for o in unit_table()):
            return False
        return os.path.getmtime(os.path.join(HERE, "l2a_rng.c")) <= os.path.getmtime(RNG_OUT) if os.path.exists(RNG_OUT) else True
    # a library that travelled without its objects (the GPU box) is judged by the sources' time stamps
    deps = [os.path.join(HERE, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return all(os.path.getmtime(d) <= t for d in deps)


def _compile(job):
    src, obj, defs = job
    cmd = [_hipcc()] + FLAGS + defs + ["-MD", "-MF", obj[:-2] + ".d", "-c", os.path.join(HERE, src), "-o", obj]
    subprocess.check_call(cmd, cwd=HERE)
    return obj


def unit_table():
    """Every translation unit of libl2a_hip.so: object name -> (source, extra flags)."""
    t = {"l2a_api.o": ("l2a_api.hip", []),
         # (holds the generic recurrent matrix-core kernel, l2a_rnn_mfma.h: its small accumulator tiles are read by the gate
         # arithmetic right away - in architectural VGPRs that needs no v_accvgpr moves, 590 of them otherwise)
         "l2a_lstm_api.o": ("l2a_lstm_api.hip", KERNEL_FLAGS),
         "l2a_comm.o": ("l2a_comm.hip", []),
         "l2a_cem.o": ("l2a_cem.hip", []),
         "l2a_step.o": ("l2a_step.hip", []),
         "l2a_micro.o": ("l2a_micro_inst.hip", MICRO_FLAGS),
         "l2a_rnn_micro.o": ("l2a_rnn_micro_inst.hip", MICRO_FLAGS)}
    for utw in LSTM_INSTANCES:
        t["l2a_lstm_%d.o" % utw] = ("l2a_lstm_inst.hip", ["-DL2A_INST_UTW=%d" % utw] + LSTM_FLAGS)
    for nt, tpw in INSTANCES:
        t["l2a_mfma_%d_%d.o" % (nt, tpw)] = ("l2a_mfma_inst.hip", ["-DL2A_INST_NT=%d" % nt, "-DL2A_INST_TPW=%d" % tpw] +
                                               (KERNEL_FLAGS if nt == 1 else []))
    for nt, tpw in FAN_INSTANCES:
        t["l2a_mfma_fan_%d_%d.o" % (nt, tpw)] = ("l2a_mfma_inst.hip", ["-DL2A_INST_NT=%d" % nt, "-DL2A_INST_TPW=%d" % tpw,
                                                                        "-DL2A_INST_FAN=1"] + KERNEL_FLAGS)
    for nt, tpw in WHOLE_INSTANCES:
        t["l2a_mfma_whole_%d_%d.o" % (nt, tpw)] = ("l2a_mfma_inst.hip", ["-DL2A_INST_NT=%d" % nt, "-DL2A_INST_TPW=%d" % tpw,
                                                                          "-DL2A_INST_FAN=2"] + KERNEL_FLAGS)
    return t


def link(objs, out):
    """libl2a_hip.so needs libl2a_rng.so (the controller step's draw-ahead chain, l2a_step.hip): found next to it at load time."""
    if not os.path.exists(RNG_OUT) or os.path.getmtime(RNG_OUT) < os.path.getmtime(os.path.join(HERE, "l2a_rng.c")):
        build_rng(verbose=False)
    if not os.path.exists(RNG_OUT):
        raise RuntimeError("libl2a_rng.so could not be built (no gcc / clang): libl2a_hip.so links against it")
    subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs +
                          ["-L" + PKG, "-ll2a_rng", "-Wl,-rpath,$ORIGIN", "-ldl"], cwd=HERE)
    return out


def relink(units, verbose=True):
    """Developer shortcut: recompile the named objects only (``l2a_micro.o`` ...) and link with the other cached objects."""
    table = unit_table()
    jobs = [(table[u][0], os.path.join(OBJ_DIR, u), table[u][1]) for u in units]
    with ThreadPoolExecutor(max_workers=max(1, len(jobs))) as pool:
        list(pool.map(_compile, jobs))
    return link([os.path.join(OBJ_DIR, o) for o in table], OUT)


def variant(name, units, defs):
    """Developer aid for A/B runs on one GPU box: ``libl2a_hip_<name>.so`` whose objects `units` are compiled with the extra
    flags `defs`; every other object is the regular build's (select the library at run time with L2A_LIB_PATH)."""
    table = unit_table()
    jobs, objs = [], []
    for o, (src, flags) in table.items():
        if o in units:
            vo = os.path.join(OBJ_DIR, o[:-2] + "_" + name + ".o")
            jobs.append((src, vo, flags + list(defs)))
            objs.append(vo)
        else:
            objs.append(os.path.join(OBJ_DIR, o))
    with ThreadPoolExecutor(max_workers=max(1, len(jobs))) as pool:
        list(pool.map(_compile, jobs))
    return link(objs, os.path.join(PKG, "libl2a_hip_%s.so" % name))


def build(force=False, verbose=True, only=None):
    """Compiles the translation units whose object is missing or older than a file it includes (all of them with
    ``force``) and links.  ``only``: optional list of (NT, TPW) pairs - the other MFMA instances are left as they are even
    when stale (developer shortcut for A/B work on one instance)."""
    if not force and only is None and up_to_date():
        if verbose:
            print("[l2a] %s is up to date" % OUT)
        return OUT
    os.makedirs(OBJ_DIR, exist_ok=True)
    table = unit_table()
    jobs = []
    for o, (src, flags) in table.items():
        obj = os.path.join(OBJ_DIR, o)
        if only is not None and o.startswith("l2a_mfma_") and os.path.exists(obj) and \
                tuple(int(x) for x in o[len("l2a_mfma_"):-2].replace("fan_", "").replace("whole_", "").split("_")) not in only:
            continue
        if not force and only is None and _obj_fresh(obj):
            continue
        jobs.append((src, obj, flags))
    if verbose:
        print("[l2a] hipcc %s : %d of %d translation units, %d parallel jobs"
              % (" ".join(FLAGS), len(jobs), len(table), min(max(len(jobs), 1), os.cpu_count() or 1)))
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as pool:
            list(pool.map(_compile, jobs))
    build_rng(verbose)
    link([os.path.join(OBJ_DIR, o) for o in table], OUT)
    return OUT


if __name__ == "__main__":
    only = None
    for a in sys.argv[1:]:
        if a.startswith("--units="):
            relink([u if u.endswith(".o") else u + ".o" for u in a[len("--units="):].split(",")])
            sys.exit(0)
        if a.startswith("--only="):
            only = [tuple(int(x) for x in part.split("x")) for part in a[len("--only="):].split(",")]
    build(force="--force" in sys.argv, only=only)
