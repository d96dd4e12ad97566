"""The C ABI from plain C: ``tests/c/abi_demo.c`` is compiled with gcc against ``include/l2a.h`` and
``libl2a_hip.so`` (no Python, no torch in that process), fed a golden case through a binary file and must
pick the reference planner's candidates."""

import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

import cases

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
ROCM = "/opt/rocm"


def _build(tmp_path):
    exe = os.path.join(str(tmp_path), "abi_demo")
    pkg = os.path.join(ROOT, "learning_to_adapt_amd")
    cmd = [shutil.which("gcc") or "gcc", "-O1", "-std=c11", "-D__HIP_PLATFORM_AMD__",
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROCM, "include"),
           os.path.join(HERE, "c", "abi_demo.c"), "-o", exe,
           "-L", pkg, "-l:libl2a_hip.so", "-L", os.path.join(ROCM, "lib"), "-lamdhip64",
           "-Wl,-rpath," + pkg, "-Wl,-rpath," + os.path.join(ROCM, "lib")]
    subprocess.check_call(cmd)
    return exe


def test_c_demo_compiles_against_the_header(tmp_path):
    """CPU box: the demo must compile and link against include/l2a.h + the built library (no run)."""
    if not os.path.exists(os.path.join(ROOT, "learning_to_adapt_amd", "libl2a_hip.so")):
        pytest.skip("library not built")
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
def test_c_demo_picks_the_reference_candidates(tmp_path):
    cid = "hc_rs_m3_n64_h5_s0"
    case, seed = cases.split_id(cid)
    gold = cases.load_golden(cid)
    env, sets, norms = cases.recipe(case)
    od, ad = env.observation_space.shape[0], env.action_space.shape[0]
    np.random.seed(seed)
    a = np.random.uniform(env.action_space.low, env.action_space.high,
                          (case["h"] * case["n"] * case["m"], ad)).reshape(case["h"], case["n"] * case["m"], ad)
    path = os.path.join(str(tmp_path), "case.bin")
    with open(path, "wb") as f:
        f.write(struct.pack("8i", od, ad, len(case["hidden"]), case["hidden"][0], case["m"], case["n"], case["h"], od - 3))
        f.write(struct.pack("2f", env.dt, 0.05))
        for p in sets[0]:
            f.write(np.ascontiguousarray(p, dtype=np.float32).tobytes())
        nm = norms[0]
        for key in ("obs", "act", "delta"):
            for j in (0, 1):
                f.write(np.ascontiguousarray(nm[key][j], dtype=np.float64).tobytes())
        f.write(np.ascontiguousarray(gold["obs0"], dtype=np.float32).tobytes())
        f.write(np.ascontiguousarray(a, dtype=np.float32).tobytes())
    exe = _build(tmp_path)

    def check(out):
        out = [ln for ln in out.decode().strip().splitlines() if ln.startswith("env ")]   # (RCCL prints a banner)
        assert len(out) == case["m"]
        for i, line in enumerate(out):
            tok = line.split()
            assert int(tok[1]) == i and int(tok[3]) == int(gold["best"][i])
            assert abs(float(tok[5]) - float(gold["returns"][i, gold["best"][i]])) <= 1e-4 * max(1.0, abs(float(tok[5])))
    check(subprocess.check_output([exe, path], timeout=120))
    # the sharded form: one rank per GPU, keys combined by l2a_allreduce_best over RCCL.  One rank here (RCCL
    # refuses two ranks on one device); with more GPUs visible the same program runs as a real multi-rank job.
    import torch
    world = max(1, min(torch.cuda.device_count(), 4))
    idfile = os.path.join(str(tmp_path), "comm.id")
    procs = [subprocess.Popen([exe, path, str(r), str(world), idfile], stdout=subprocess.PIPE) for r in range(world)]
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0
        check(out)


@pytest.mark.gpu
def test_c_demo_drives_the_controller_step(tmp_path):
    """The round-5 entry points from plain C (`abi_demo <case> ctrl <seed> <steps> <state file>`): the program seeds its OWN
    MT19937 state like ``np.random.seed`` does, then l2a_controller_create / _step x 4 / _rearm / _stats / _destroy.  Step 0
    must be the reference planner's golden plan (index, the float64 first action bit for bit); every step must be what the
    drop-in ``MPCController`` returns for the same call sequence; and the generator must be left where NumPy's is."""
    cid = "hc_rs_m3_n64_h5_s0"
    case, seed = cases.split_id(cid)
    gold = cases.load_golden(cid)
    env, sets, norms = cases.recipe(case)
    od, ad = env.observation_space.shape[0], env.action_space.shape[0]
    steps = 4
    obs32 = np.ascontiguousarray(gold["obs0"], dtype=np.float32)
    path = os.path.join(str(tmp_path), "case.bin")
    with open(path, "wb") as f:
        f.write(struct.pack("8i", od, ad, len(case["hidden"]), case["hidden"][0], case["m"], case["n"], case["h"], od - 3))
        f.write(struct.pack("2f", env.dt, 0.05))
        for p in sets[0]:
            f.write(np.ascontiguousarray(p, dtype=np.float32).tobytes())
        nm = norms[0]
        for key in ("obs", "act", "delta"):
            for j in (0, 1):
                f.write(np.ascontiguousarray(nm[key][j], dtype=np.float64).tobytes())
        f.write(obs32.tobytes())
        f.write(np.zeros((case["h"] * case["n"] * case["m"], ad), dtype=np.float32).tobytes())     # (unused in this mode)
    exe = _build(tmp_path)
    state_file = os.path.join(str(tmp_path), "mt.bin")
    out = subprocess.check_output([exe, path, "ctrl", str(seed), str(steps), state_file], timeout=120).decode()
    rows = {}
    for ln in out.splitlines():
        tok = ln.split()
        if tok and tok[0] == "step":
            rows[(int(tok[1]), int(tok[3]))] = (int(tok[5]), float(tok[7]), np.array([float.fromhex(x) for x in tok[9:9 + ad]]))
    assert len(rows) == steps * case["m"]
    stats = [ln for ln in out.splitlines() if ln.startswith("stats ")][0].split()
    assert int(stats[2]) == steps
    # step 0 = the reference planner's golden vector
    for i in range(case["m"]):
        idx, ret, act = rows[(0, i)]
        assert idx == int(gold["best"][i])
        assert np.array_equal(act, gold["chosen"][i])
        assert abs(ret - float(gold["returns"][i, idx])) <= 1e-4 * max(1.0, abs(ret))
    # every step = the drop-in controller on the same call sequence (the observations the C program cast to float64 again)
    ctrl = cases.product_controller(case)
    np.random.seed(seed)
    obs64 = obs32.astype(np.float64)
    for s in range(steps):
        want, _ = ctrl.get_actions(obs64)
        for i in range(case["m"]):
            assert rows[(s, i)][0] == int(ctrl.last_plan["best_index"][i])
            assert np.array_equal(rows[(s, i)][2], want[i])
    # ... and the C program's generator is where NumPy's global one is
    raw = np.fromfile(state_file, dtype=np.uint32)
    key, pos = raw[:624], int(raw[624])
    st = np.random.get_state()
    theirs = np.random.RandomState()
    theirs.set_state(("MT19937", key, pos))
    ours = np.random.RandomState()
    ours.set_state(st)
    assert np.array_equal(theirs.uniform(size=8), ours.uniform(size=8))
