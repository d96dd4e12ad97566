#!/usr/bin/env python
"""bench.py - controller-steps/s of the fused MPC random-shooting plan step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json config 2, the configuration `metric` is quoted on): HalfCheetah-shaped
synthetic inputs, n_candidates = 2000 per GPU, horizon 30, mean-ensemble of 5 MLPs
26->512->512->20 (fp32), discount 1, random-init weights from the seeded recipe
(learning_to_adapt_amd/utils/synthetic.py).  A "step" is one COMPLETED controller step: one call of the
drop-in `MPCController.get_actions(obs)` in bit-exact parity mode - candidates drawn from NumPy's
legacy global generator exactly as the reference draws them, cast, uploaded, rolled out by the fused
kernel, arg-max (+ one int64 MAX all-reduce over RCCL when N > 1), 8 bytes read back, the chosen
float64 action returned.  `value` = those calls per second.  At N > 1 the plan's candidates are
sharded over the ranks, 2000 per GPU (weak scaling: N = 8 is BASELINE.json's config 4, one
16000-candidate plan per step); `value` then counts 2000-candidate controller steps,
value = N * plan_steps_per_s, and `config.plan_steps_per_s` states the plans themselves.

Beside it, in `config`: the same loop with `rng="device"` (candidates drawn on the GPU) and the
kernel-only loop over candidates resident in HBM (memset + kernel [+ all-reduce]) - the three modes of
SURVEY.md H5.  Extra objects: `roofline` (fp32 MFMA roofline of the rollout kernel, HIP events around
each launch on the launch stream) and `cpu_baseline` (the NumPy oracle = CPU restatement of the reference
path, timed on this box's host cores, N = 1 only).

`config.get_actions_parity_foreign_draw_plan_steps_per_s` is the parity loop with one foreign `np.random.uniform()`
between calls (what an in-process env reset does): the draw-ahead block is stale every step and the draw is pipelined
under the rollout instead.  `config.c5_cem` is BASELINE config 5 at this N: one CEM plan step (5 x 4000 candidates, the plan's
candidates sharded over the ranks - strong scaling).  At N > 1 `config.strong_scaling` compares the sharded plan with the SAME plan on one GPU.

`python bench.py --gpus N` without a torch.distributed.run environment launches the N ranks itself.
"""

import argparse
import json
import os
import re
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

N_CAND, HORIZON, ENSEMBLE, HIDDEN = 2000, 30, 5, (512, 512)
OBS_DIM, ACT_DIM = 20, 6
MAC_PER_ROW = (OBS_DIM + ACT_DIM) * 512 + 512 * 512 + 512 * OBS_DIM          # 285 696
FLOP_PER_LAUNCH = 2.0 * MAC_PER_ROW * ENSEMBLE * N_CAND * HORIZON             # 171.4 GFLOP
PARAMS_PER_SET = (OBS_DIM + ACT_DIM) * 512 + 512 + 512 * 512 + 512 + 512 * OBS_DIM + OBS_DIM
HBM_BYTES_PER_LAUNCH = 4.0 * (N_CAND * HORIZON * ACT_DIM + ENSEMBLE * PARAMS_PER_SET + OBS_DIM) + 8
PEAK_FP32_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32 MFMA = fp32 vector peak
PEAK_HBM_GBS = 8000.0


def cpu_table():
    """SURVEY.md 8(d) 'CPU baseline beside it': the oracle on this host, median of 5 runs after one
    warm-up (3 runs for the config-4 shape), at 1 thread and at the fastest probed thread count."""
    import cases
    from oracle import make_reward
    from oracle.planner import rollout_returns
    from threadpoolctl import threadpool_limits
    ncpu = os.cpu_count() or 1
    rows = [("config 1 (n=500, h=10, E=1)", "c1_hc_rs_n500_h10_e1", {}, 5),
            ("config 2 shape with E=1", "c2_hc_rs_n2000_h30_e5", {"E": 1}, 5),
            ("config 2 (n=2000, h=30, E=5)", "c2_hc_rs_n2000_h30_e5", {}, 5),
            ("config 4 shape (n=16000, h=30, E=5), one host", "c2_hc_rs_n2000_h30_e5", {"n": 16000}, 3)]
    for label, name, over, reps in rows:
        case = dict(cases.CASES[name], **over)
        env = cases.recipe(case)[0]
        dyn = cases.oracle_dynamics(case)
        reward = make_reward("half_cheetah", env.dt)
        n, h, m = case["n"], case["h"], case["m"]
        np.random.seed(0)
        a = np.random.uniform(low=env.action_space.low, high=env.action_space.high,
                              size=(h * n * m, ACT_DIM)).reshape((h, n * m, ACT_DIM))
        obs0 = np.random.RandomState(1).randn(m, OBS_DIM)

        def once():
            t = time.perf_counter()
            rollout_returns(dyn, reward, obs0, a, n, 1.0)
            return time.perf_counter() - t
        probe = {}
        for thr in sorted(set([t for t in (1, 8, 16, 32, 64) if t <= ncpu] + [ncpu])):
            with threadpool_limits(limits=thr):
                once()
                probe[thr] = once()
        best = min((t for t in probe if t != 1), key=lambda t: probe[t], default=1)
        out = {"config": label, "host_cpus": ncpu, "probe_s": {str(k): round(v, 4) for k, v in probe.items()}}
        for tag, thr in (("threads_1", 1), ("threads_best", best)):
            with threadpool_limits(limits=thr):
                ts = sorted(once() for _ in range(reps))
            out[tag] = {"threads": thr, "s_per_step": round(ts[len(ts) // 2], 4),
                        "steps_per_s": round(1.0 / ts[len(ts) // 2], 4)}
        print(json.dumps(out), flush=True)


def _self_launch(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks ourselves (one process per
    GPU, rendezvous on 127.0.0.1) and hand their output through."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def _timed_calls(fn, steps, warmup, sync):
    for _ in range(warmup):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true",
                    help="profiling runs: only the kernel-only loop (candidates resident in HBM), so that the "
                         "kernel statistics hold the timed plan launches and nothing else; `value` is then that loop")
    ap.add_argument("--cpu-steps", type=int, default=0, help="oracle steps to time (0 = auto, ~10-30 s)")
    ap.add_argument("--cpu-table", action="store_true",
                    help="no GPU work: time the oracle (CPU restatement) on configs 1, 2 (E=1, E=5) and the "
                         "config-4 shape at 1 BLAS thread and at the fastest thread count; JSON lines")
    args = ap.parse_args()
    if args.cpu_table:
        return cpu_table()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(_self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the rollout path is HIP-only)"
    # Developer hook (tools/gpu_round.sh): L2A_BENCH_SHARE_GPU=1 runs every rank on GPU 0 over gloo so that
    # the N > 1 code path can be exercised on a one-GPU box.  Never set by the driver; numbers are meaningless.
    share = os.environ.get("L2A_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    collective_ranks = None
    backend = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            backend = "gloo"
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            backend = "nccl"
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        ones = torch.ones((1,), dtype=torch.int64, device=torch.device("cuda", local_rank))
        dist.all_reduce(ones)                               # a real collective: how many ranks answered
        collective_ranks = int(ones.cpu()[0])
        assert collective_ranks == world

    from learning_to_adapt_amd import _lib
    from learning_to_adapt_amd.dynamics.native_model import NativeModel
    from learning_to_adapt_amd.utils import fast_rng
    import cases

    n_glob = N_CAND * world
    case = dict(cases.CASES["c2_hc_rs_n2000_h30_e5"], n=n_glob)
    env, model = cases.product_model(case)       # the drop-in model (binds to this process' current GPU)
    native = NativeModel(OBS_DIM, ACT_DIM, HIDDEN, "relu", None, ENSEMBLE, "mean", device=local_rank)
    _, sets, norms = cases.recipe(case)
    for e in range(ENSEMBLE):
        native.set_weights(e, sets[e])
        native.set_norm(e, norms[e])
    dev = native.device
    spec = env.reward_spec
    gold = cases.load_golden("c2_hc_rs_n2000_h30_e5_s0")
    gold_glob = None                             # golden vector of the reference planner for THIS global plan
    if world == 1:
        gold_glob = gold
    elif n_glob == 16000:
        gold_glob = cases.load_golden("c4_hc_rs_n16000_h30_e5_s0")
    obs_np = np.array(gold["obs0"])              # (c4 uses the same observation recipe)
    obs0 = torch.from_numpy(obs_np.astype(np.float32)).to(dev)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.cpu()[0])

    # ---- kernel-only loop: candidates resident in HBM ----------------------------------------------------------
    # buffer 0 = the reference's seed-0 draw of THIS rank's shard (index check), 1..3 = device-generated
    np.random.seed(0)                        # the draw of mpc_controller.py:67-69,114 for n_glob candidates
    a0 = np.random.uniform(low=env.action_space.low, high=env.action_space.high,
                           size=(HORIZON * n_glob, ACT_DIM)).reshape((HORIZON, n_glob, ACT_DIM))
    lo = rank * N_CAND
    bufs = [torch.from_numpy(np.ascontiguousarray(a0[:, lo:lo + N_CAND, :], dtype=np.float32)).to(dev)]
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    for _ in range(3):
        bufs.append(torch.rand((HORIZON, N_CAND, ACT_DIM), generator=gen, device=dev) * 2.0 - 1.0)
    best = torch.zeros((1,), dtype=torch.int64, device=dev)
    counter = [0]

    def resident_step():
        i = counter[0]
        counter[0] += 1
        native.plan_rs(obs0, bufs[i % len(bufs)], 1, N_CAND, HORIZON, 1.0, spec, cand_offset=lo, best_key=best)
        if world > 1:
            dist.all_reduce(best, op=dist.ReduceOp.MAX)

    # correctness gate: the first plan must pick the reference planner's candidate
    resident_step()
    torch.cuda.synchronize()
    ret0, idx0 = _lib.key_decode(int(best.cpu()[0]))
    index_match = None
    if gold_glob is not None:
        index_match = bool(idx0 == int(gold_glob["best"][0]))
        assert index_match, "plan picked candidate %d, reference picked %d" % (idx0, int(gold_glob["best"][0]))
    # Clocks: after idle time the MI355X needs some tens of milliseconds of back-to-back work to reach its ~2.39 GHz (2.0 -
    # 2.15 GHz over the first launches, tools/timeline*.py), and the fractions below are quoted against the peak-clock number.
    # A short run (the driver's 20 steps are 30 ms) measured the ramp, not the kernel: round 3 read 649.7 steps/s there and
    # 665 - 682 with longer runs.  Every timed leg is therefore preceded by ~0.15 s of untimed kernel launches (not steps: the
    # W warm-up steps and the K timed steps of each leg are untouched).
    def clock_warm(ms=150.0):
        t_end = time.perf_counter() + ms * 1e-3
        while time.perf_counter() < t_end:
            for _ in range(8):
                native.plan_rs(obs0, bufs[1], 1, N_CAND, HORIZON, 1.0, spec, cand_offset=lo, best_key=best)
            torch.cuda.synchronize()

    clock_warm()
    resident_s = max_over_ranks(_timed_calls(resident_step, args.steps, args.warmup, sync))
    native.ctx.launch_status()          # raises if any launch flagged a problem

    # ---- kernel duration: HIP events around each launch on the launch stream (every rank: min / max over ranks at N > 1) -----
    k2 = min(args.steps, 100)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(k2)]
    clock_warm()
    for i in range(k2):
        evs[i][0].record()
        native.plan_rs(obs0, bufs[i % len(bufs)], 1, N_CAND, HORIZON, 1.0, spec, cand_offset=lo, best_key=best)
        evs[i][1].record()
    torch.cuda.synchronize()
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    kern_ms_ranks = None
    collective = None
    if world > 1:
        t = torch.zeros((world,), dtype=torch.float64, device=dev)
        t[rank] = kern_ms
        dist.all_reduce(t)
        kern_ms_ranks = [float(v) for v in t.cpu()]
        # the collective on its own: the step's one int64 MAX all-reduce of the keys, HIP events on the launch stream around each
        # call (the blocking form torch.distributed gives the controller: the launch stream waits for the collective), the ranks
        # aligned by a barrier in front of every call so that the number is the collective, not the ranks' skew
        k3 = min(args.steps, 50)
        cev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(k3)]
        wall = []
        for _ in range(5):
            dist.all_reduce(best, op=dist.ReduceOp.MAX)
        for i in range(k3):
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            cev[i][0].record()
            dist.all_reduce(best, op=dist.ReduceOp.MAX)
            cev[i][1].record()
            torch.cuda.synchronize()
            wall.append(1e6 * (time.perf_counter() - t0))
        dev_us = [1e3 * a.elapsed_time(b) for a, b in cev]
        c = torch.tensor([float(np.median(dev_us)), float(np.median(wall))], dtype=torch.float64, device=dev)
        dist.all_reduce(c, op=dist.ReduceOp.MAX)
        collective = {"collective_us": round(float(c.cpu()[0]), 2), "collective_host_wall_us": round(float(c.cpu()[1]), 2),
                      "what": "median over %d isolated int64 MAX all-reduces of the step's key word (max over ranks): HIP events "
                              "on the launch stream | host wall clock incl. the stream synchronisation" % k3}
        dist.barrier()

    # ---- completed controller steps through the drop-in MPCController ------------------------------------------
    e2e = {}
    e2e_index_match = None
    if not args.no_e2e:
        for mode in ("numpy", "device"):
            ctrl = cases.product_controller(case, model=model, env=env, rng=mode)
            np.random.seed(0)                   # every rank alike: the shards are slices of ONE candidate tensor
            torch.manual_seed(0)
            ctrl.get_actions(obs_np)
            first_index = int(ctrl.last_plan["best_index"][0])
            # two more untimed calls before the clocks are warmed: the second call of a controller starts the draw-ahead chain
            # (thread pool, pinned staging: ~8 ms with the GPU idle - long enough for the board to drop its clocks again, which
            # then climb back over the next ~25 calls: tools/probe_first_calls.py: 1.72 -> 1.50 ms per call)
            ctrl.get_actions(obs_np)
            ctrl.get_actions(obs_np)
            clock_warm()
            if mode == "numpy" and gold_glob is not None:
                e2e_index_match = bool(first_index == int(gold_glob["best"][0]))
                assert e2e_index_match, "get_actions picked %d, reference picked %d" % (first_index, int(gold_glob["best"][0]))
            e2e[mode] = max_over_ranks(_timed_calls(lambda: ctrl.get_actions(obs_np), args.steps, args.warmup, sync))
            if mode == "numpy":
                e2e["draw_ahead_hits"] = int(ctrl.draw_ahead_stats()["hits"])
                e2e["native_step"] = ctrl._cstep is not None
                if ctrl._cstep is not None:
                    e2e["native_step_stage_us"] = {k: round(v, 1) for k, v in ctrl._cstep.stats()["stage_us"].items()}
            if mode == "numpy":
                # the mode an in-process env sees (n_parallel = 1: samplers/vectorized_env_executor.py:45 ->
                # envs/mujoco_env.py:85-87 reset noise; samplers/utils.py rollout()): somebody else consumes the global
                # generator between two plans, so the block drawn ahead is stale and the step draws synchronously
                # (pipelined over the horizon under the rollout).  Every rank consumes alike: the shards stay slices of
                # one candidate tensor.
                def foreign_step():
                    np.random.uniform()
                    ctrl.get_actions(obs_np)
                hits0 = int(ctrl.draw_ahead_stats()["hits"])
                clock_warm()
                e2e["foreign"] = max_over_ranks(_timed_calls(foreign_step, args.steps, min(args.warmup, 5), sync))
                e2e["foreign_hits"] = int(ctrl.draw_ahead_stats()["hits"]) - hits0
            if ctrl._ahead is not None:
                ctrl._ahead.stop()
            if ctrl._cstep is not None:
                ctrl._cstep.close()
                ctrl._cstep = None

    # ---- BASELINE config 5: one CEM plan step (5 iterations x n = 4000, h = 30, E = 5), candidates sharded over the ranks ----
    # Device-RNG controller (sample / rollout of this rank's shard / all-gather of the returns / refit, five times, + pick:
    # policies/mpc_controller.py get_cem_action_device).  n = 4000 is the WHOLE plan at every N (strong scaling: 4000 / N per
    # rank - at N = 8 the 500-candidate shard the member fan of csrc/l2a_mfma.h exists for).  Bounded: <= ~0.5 s.
    c5 = None
    if not args.no_e2e:
        case5 = cases.CASES["c5_hc_cem_n4000_h30_e5"]
        ctrl5 = cases.product_controller(case5, model=model, env=env, rng="device")
        torch.manual_seed(0)                    # every rank alike: the ranks sample the same normals and keep their slice
        for _ in range(3):
            ctrl5.get_actions(obs_np)
        clock_warm()
        k5 = max(5, min(20, args.steps))
        c5_s = max_over_ranks(_timed_calls(lambda: ctrl5.get_actions(obs_np), k5, 2, sync))
        lo5, hi5 = ctrl5._shard_range(case5["n"], rank, world)
        c5 = {"workload": "BASELINE config 5: HalfCheetah CEM, %d iterations x n=%d, h=%d, ens=%d; candidates sharded over %d rank(s) "
                          "(%d per rank), rng=device" % (case5["num_cem_iters"], case5["n"], case5["h"], case5["E"], world, hi5 - lo5),
              "plan_step_ms": round(1e3 * c5_s / k5, 4), "plan_steps_per_s": round(k5 / c5_s, 3), "timed_plan_steps": k5,
              "candidates_per_rank": hi5 - lo5, "scaling": "strong (the plan is 4000 candidates at every N)",
              "best_index": int(ctrl5.last_plan["best_index"][0])}

    # ---- strong scaling of THIS plan (n_glob candidates): rank 0 alone runs the whole plan on its one GPU ------------
    strong = None
    if world > 1:
        full_ms = None
        if rank == 0:
            a_full = torch.rand((HORIZON, n_glob, ACT_DIM), generator=gen, device=dev) * 2.0 - 1.0
            k3 = max(3, min(10, args.steps))
            for _ in range(2):
                native.plan_rs(obs0, a_full, 1, n_glob, HORIZON, 1.0, spec, best_key=best)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(k3):
                native.plan_rs(obs0, a_full, 1, n_glob, HORIZON, 1.0, spec, best_key=best)
            torch.cuda.synchronize()
            full_ms = 1e3 * (time.perf_counter() - t0) / k3
            native.ctx.launch_status()
            del a_full
        dist.barrier()
        if rank == 0:
            n_ms = 1e3 * resident_s / args.steps
            strong = {"plan_candidates": n_glob, "one_gpu_kernel_only_ms": round(full_ms, 4),
                      "n_gpu_kernel_only_ms": round(n_ms, 4), "speedup": round(full_ms / n_ms, 3),
                      "efficiency": round(full_ms / n_ms / world, 4),
                      "note": "the SAME %d-candidate plan on one GPU (rank 0 alone) vs sharded over %d GPUs incl. the "
                              "all-reduce; kernel-only loops" % (n_glob, world)}

    # ---- CPU baseline: the NumPy oracle on this box's host cores (N = 1, rank 0) ---------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import make_reward
        from oracle.planner import rollout_returns
        case1 = cases.CASES["c2_hc_rs_n2000_h30_e5"]
        dyn = cases.oracle_dynamics(case1)
        reward = make_reward("half_cheetah", env.dt)
        a64 = a0.astype(np.float64)
        # NumPy/OpenBLAS with every hardware thread is slower than with a few (2000 x 512 GEMMs):
        # probe a few thread counts, keep the fastest, report the one actually used.
        try:
            from threadpoolctl import threadpool_limits
        except Exception:
            threadpool_limits = None
        ncpu = os.cpu_count() or 1
        trials = sorted(set([t for t in (8, 16, 32, 64) if t <= ncpu] + [ncpu])) if threadpool_limits else [ncpu]

        import contextlib

        def limited(thr):
            return threadpool_limits(limits=thr) if threadpool_limits else contextlib.nullcontext()

        best_t, best_dt = trials[-1], None
        for thr in trials:
            with limited(thr):
                rollout_returns(dyn, reward, gold["obs0"], a64, N_CAND, 1.0)        # warm-up at this setting
                t1 = time.perf_counter()
                rollout_returns(dyn, reward, gold["obs0"], a64, N_CAND, 1.0)
                dt = time.perf_counter() - t1
            if best_dt is None or dt < best_dt:
                best_t, best_dt = thr, dt
        k4 = args.cpu_steps or int(max(3, min(20, round(10.0 / max(best_dt, 1e-3)))))
        with limited(best_t):
            t1 = time.perf_counter()
            for _ in range(k4):
                r = rollout_returns(dyn, reward, gold["obs0"], a64, N_CAND, 1.0)
            cpu_t = (time.perf_counter() - t1) / k4
        assert int(np.argmax(r)) == int(gold["best"][0])
        cpu = {"value": round(1.0 / cpu_t, 4), "unit": "controller-steps/s", "cores": int(best_t),
               "kind": "port",
               "sample": "%d plan steps of the same workload (n=2000, h=30, ens=5) through oracle/ "
                         "(NumPy/OpenBLAS fp32 MLP, float64 host state; candidates already drawn), %.2f s each; "
                         "fastest of %s BLAS threads" % (k4, cpu_t, trials),
               "host_cpus": ncpu}

    if rank == 0:
        headline_s = e2e.get("numpy", resident_s)
        ms_per_step = 1e3 * headline_s / args.steps
        plan_steps = args.steps / headline_s
        out = {
            "metric": "controller-steps/sec (n_cand=2000, H=30, ens=5)",
            "value": round(world * plan_steps, 3),
            "unit": "controller-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "value_strong": round(plan_steps, 3),
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "HalfCheetah mb_mpc random shooting, n_candidates=2000 per GPU (one plan of %d "
                            "candidates per step), horizon=30, mean-ensemble=5 x MLP 26-512-512-20, m=1 env" % n_glob,
                "step": ("completed MPCController.get_actions call, parity mode (host MT19937 draw identical to the "
                         "reference, upload, fused rollout, arg-max%s, read-back)"
                         % (", int64 MAX all-reduce" if world > 1 else "")) if "numpy" in e2e else
                        "kernel-only: memset + fused rollout over candidates resident in HBM (--no-e2e)",
                "value_definition": "value = n_gpus x plan_steps_per_s: 2000-candidate controller steps per second (weak scaling: per-GPU "
                                    "work fixed); value_strong = plan_steps_per_s = completed plans of n_gpus x 2000 candidates per second - "
                                    "read both (at N = 1 they coincide)",
                "candidates_per_plan": n_glob,
                "plan_steps_per_s": round(plan_steps, 3),
                "kernel_only_plan_steps_per_s": round(args.steps / resident_s, 3),
                "kernel_only_ms_per_step": round(1e3 * resident_s / args.steps, 4),
                "mlp_steps_per_ms": round(world * N_CAND * HORIZON * ENSEMBLE / ms_per_step, 1),
                "collective": "int64 MAX all-reduce of 1 key per plan step" if world > 1 else "none",
                "backend": backend, "collective_ranks": collective_ranks,
                "action_index_match_vs_reference": index_match if e2e_index_match is None else
                                                   bool(index_match and e2e_index_match),
                "best_index": idx0, "best_return": round(ret0, 4),
                "host_rng_threads": fast_rng.threads(),
                "clock_warm": "0.15 s of untimed kernel launches in front of every timed leg (the board's clock ramp; since round 4 - "
                              "BENCH_r01..r03 were measured without it and read 2 - 3 % lower on short runs)",
            },
        }
        if "numpy" in e2e:
            out["config"]["get_actions_parity_plan_steps_per_s"] = round(args.steps / e2e["numpy"], 3)
            out["config"]["draw_ahead_hits"] = e2e.get("draw_ahead_hits")
            out["config"]["native_step"] = e2e.get("native_step")
            if world > 1:
                out["config"]["native_comm"] = ("the library's own RCCL communicator (l2a_allreduce_best)" if os.environ.get("L2A_NATIVE_COMM", "0") == "1"
                                                else "torch.distributed behind the C step's reduce callback")
            if e2e.get("native_step_stage_us"):
                out["config"]["native_step_stage_us"] = e2e["native_step_stage_us"]
            out["config"]["host_path_us_per_step"] = round(1e3 * (ms_per_step - 1e3 * resident_s / args.steps), 1)
        if "foreign" in e2e:
            out["config"]["get_actions_parity_foreign_draw_plan_steps_per_s"] = round(args.steps / e2e["foreign"], 3)
            out["config"]["foreign_draw_note"] = ("one np.random.uniform() consumed between calls (an in-process env reset): "
                                                  "draw-ahead hits in that loop: %d of %d" % (e2e["foreign_hits"], args.steps))
        if c5 is not None:
            out["config"]["c5_cem"] = c5
        if strong is not None:
            out["config"]["strong_scaling"] = strong
        if collective is not None:
            out["config"].update(collective)
        if kern_ms_ranks is not None:
            out["config"]["kernel_ms_per_rank"] = {"min": round(min(kern_ms_ranks), 4), "max": round(max(kern_ms_ranks), 4),
                                                   "ranks": [round(v, 4) for v in kern_ms_ranks]}
        if "device" in e2e:
            out["config"]["get_actions_device_rng_plan_steps_per_s"] = round(args.steps / e2e["device"], 3)
        if kern_ms is not None and rank == 0:
            # HBM-side bytes per launch from the committed PMC passes (FETCH_SIZE x 2 correction +
            # WRITE_SIZE, MI355X_MICROARCH.md section HBM); counters cannot be read live.
            traffic, traffic_src = None, None
            pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(pmc_path):
                with open(pmc_path) as f:
                    pm = json.load(f)
                traffic, traffic_src = pm.get("hbm_bytes_per_launch"), pm.get("source")
            # the same kernel's average in the committed rocprofv3 --kernel-trace --stats summary (same command, another run)
            rocprof_ms, rocprof_src = None, None
            for name in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
                # (rNN_kernel_stats.csv only: the summaries of other commands - defaults, config-5 shard - hold other instances)
                if re.match(r"^r\d+_kernel_stats\.csv$", name):
                    with open(os.path.join(ROOT, "profiles", name)) as f:
                        for line in f:
                            if "l2a_rollout_mfma_k<1, 8, 2, 2, false" in line:
                                cols = line.rsplit('",', 1)[-1].split(",") if line.startswith('"') else line.split(",")[1:]
                                rocprof_ms, rocprof_src = float(cols[2]) * 1e-6, "profiles/" + name
                                break
                    if rocprof_ms is not None:
                        break
            ach = FLOP_PER_LAUNCH / (kern_ms * 1e-3) / 1e12
            out["roofline"] = {
                "bound": "mfma", "kernel": "l2a_rollout_mfma_k<NT=1,TPW=8,OT=2,KG0=2,GACT=false>",
                "achieved": round(ach, 3), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / PEAK_FP32_TFLOPS, 4), "traffic": traffic,
                "kernel_ms": round(kern_ms, 4),
                "kernel_ms_rocprof_committed": None if rocprof_ms is None else round(rocprof_ms, 4),
                "kernel_ms_rocprof_committed_source": rocprof_src,
                "kernel_ms_rocprof_committed_note": "NOT from this run: the same kernel's average in the rocprofv3 --kernel-trace --stats "
                                                    "summary committed under profiles/ (same command, another box and run); kernel_ms is this run's",
                "flop_per_launch": FLOP_PER_LAUNCH,
                "hbm_algorithmic_bytes_per_launch": HBM_BYTES_PER_LAUNCH,
                "hbm_achieved_GBps": round(HBM_BYTES_PER_LAUNCH / (kern_ms * 1e-3) / 1e9, 3),
                "hbm_frac": round(HBM_BYTES_PER_LAUNCH / (kern_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 6),
                "traffic_source": traffic_src,
                "note": "dense contraction (24 kFLOP per compulsory HBM byte): bound by the fp32 matrix "
                        "rate, not HBM; hbm_* reported because BASELINE.json names the HBM roofline",
            }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
