"""N > 1 path on REAL kernels: several `gloo` ranks share GPU 0 (tile split off, as DESIGN section 6 requires of processes
that share a GPU) and run the sharded controller step end to end - launch, payload packed on the device
(`l2a_plan_payload`), the collective, the relaunch protocol - through `MPCController.get_actions` /
`RNNMPCController.get_actions`.  `tests/test_distributed_cpu.py` runs the same host paths with the launch replaced by the
oracle; here nothing is replaced.  What the driver's multi-GPU run meets is what ran here, except RCCL for gloo.

Reference semantics preserved: `np.argmax` first-max over ALL candidates (policies/mpc_controller.py:128-129)."""

import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _setup(rank, world, port):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["L2A_SPLIT"] = "0"           # ranks sharing one GPU: a tile's two workgroups may not be co-resident
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)


def _worker_golden(rank, world, port, cid, steps, out_dir):
    _setup(rank, world, port)
    try:
        import cases
        case, seed = cases.split_id(cid)
        gold = cases.load_golden(cid)
        ctrl = cases.product_controller(case)
        np.random.seed(seed)                      # every rank draws the same candidate tensor and keeps its slice
        out = {}
        for k in range(steps):                    # (later steps run through the draw-ahead chain: same state everywhere)
            actions, _ = ctrl.get_actions(gold["obs0"])
            out["actions_%d" % k] = actions
            out["best_%d" % k] = np.asarray(ctrl.last_plan["best_index"])
        out["shard"] = np.asarray(ctrl.last_plan.get("shard", (-1, -1)))
        out["rng_next"] = np.asarray(np.random.uniform())
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("cid,world", [("c4_hc_rs_n16000_h30_e5_s0", 2), ("c4_hc_rs_n16000_h30_e5_s0", 4),
                                       ("c5_hc_cem_n4000_h30_e5_s0", 2), ("c5_hc_cem_n4000_h30_e5_s0", 3),
                                       ("c3b_ant_rs_n500_h10_pb5_3x512_s0", 3), ("hc_rs_ragged_n37_h3_s0", 4)])
def test_sharded_plan_on_the_gpu_equals_the_reference_plan(cid, world, tmp_path):
    """Config 4 (16 000 candidates in 2 / 4 shards, one MAX all-reduce of the device-packed payload), config 5 (CEM: the
    returns all-gathered every iteration, uneven shards padded), the GrBAL default plan on per-block sets (micro-tile
    kernels, uneven shards), a ragged tiny plan with 9 / 10 candidates per rank: every rank must return the golden action
    of the reference planner and leave np.random where the reference leaves it."""
    steps = 3 if cid.startswith("hc_rs_ragged") else 1       # (three steps: the later ones take the blocks the C chain drew ahead, sliced per rank)
    mp.spawn(_worker_golden, args=(world, _free_port(), cid, steps, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, HERE)
    import cases
    gold = cases.load_golden(cid)
    outs = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    for o in outs:
        assert np.array_equal(o["best_0"], gold["best"])
        np.testing.assert_array_equal(o["actions_0"], gold["chosen"])
        np.testing.assert_array_equal(o["rng_next"], outs[0]["rng_next"])
        for k in range(1, steps):
            np.testing.assert_array_equal(o["best_%d" % k], outs[0]["best_%d" % k])
            np.testing.assert_array_equal(o["actions_%d" % k], outs[0]["actions_%d" % k])
    if steps == 1:
        assert float(outs[0]["rng_next"]) == float(gold["rng_next"])          # same RNG consumption as the reference
    else:
        # later steps against ONE process planning the whole case (the single-GPU path, itself pinned to the reference's goldens)
        case, seed = cases.split_id(cid)
        ctrl = cases.product_controller(case)
        np.random.seed(seed)
        for k in range(steps):
            want, _ = ctrl.get_actions(gold["obs0"])
            np.testing.assert_array_equal(outs[0]["actions_%d" % k], want)
        assert float(outs[0]["rng_next"]) == float(np.random.uniform())
    if outs[0]["shard"][0] >= 0:
        edges = [int(outs[0]["shard"][0])] + [int(o["shard"][1]) for o in outs]
        assert edges[0] == 0 and edges[-1] == cases.split_id(cid)[0]["n"]
        assert all(int(o["shard"][0]) == e for o, e in zip(outs, edges[:-1]))      # contiguous, disjoint, complete


def _worker_protocol(rank, world, port, scenario, out_dir):
    _setup(rank, world, port)
    try:
        import cases
        from learning_to_adapt_amd import _lib
        cid = "hc_rs_m3_n64_h5_s0"
        case, seed = cases.split_id(cid)
        gold = cases.load_golden(cid)
        ctrl = cases.product_controller(case)
        ctx = ctrl.dynamics_model.planner_model().ctx
        np.random.seed(seed)
        result = {"raised": np.asarray(0)}
        if scenario == "flag":
            # a flagged launch on ONE rank (what a lost tile-split partner reports): the reduced flag makes every rank
            # relaunch together, nobody contributes a stale key, the plan is still the reference's
            if rank == 1:
                ctx.check(ctx.lib.l2a_inject_status(ctx.handle, 1), "l2a_inject_status")
            actions, _ = ctrl.get_actions(gold["obs0"])
            result.update(actions=actions, best=np.asarray(ctrl.last_plan["best_index"]),
                          degraded=np.asarray(int(getattr(ctx, "split_degraded", False))))
        else:
            # a foreign consumer of np.random on ONE rank between two plans: caught on that step, on every rank
            ctrl.get_actions(gold["obs0"])
            if rank == world - 1:
                np.random.uniform()
            try:
                ctrl.get_actions(gold["obs0"])
            except _lib.L2AError as e:
                result["raised"] = np.asarray(1)
                result["message"] = np.asarray(str(e))
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **result)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_a_flagged_launch_on_one_rank_makes_every_rank_relaunch_on_the_gpu(world, tmp_path):
    mp.spawn(_worker_protocol, args=(world, _free_port(), "flag", str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, HERE)
    import cases
    gold = cases.load_golden("hc_rs_m3_n64_h5_s0")
    for r in range(world):
        o = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        assert np.array_equal(o["best"], gold["best"])
        np.testing.assert_array_equal(o["actions"], gold["chosen"])
        assert int(o["degraded"]) == 1, "rank %d did not take part in the collective switch to the unsplit geometry" % r


def test_a_rank_whose_generator_drifts_is_caught_on_that_step_on_the_gpu(tmp_path):
    world = 2
    mp.spawn(_worker_protocol, args=(world, _free_port(), "drift", str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        o = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        assert int(o["raised"]) == 1, "rank %d planned on with ranks whose candidate tensors differ" % r
        assert "identical" in str(o["message"])


def _worker_device(rank, world, port, out_dir):
    _setup(rank, world, port)
    try:
        import cases
        case = dict(cases.CASES["hc_rs_m3_n64_h5"], n=333, h=4)
        ctrl = cases.product_controller(case, rng="device")
        torch.manual_seed(1234)
        obs = np.random.RandomState(3).randn(case["m"], 20)
        out = {}
        for k in range(3):
            a, _ = ctrl.get_actions(obs)
            out["actions_%d" % k] = a
            out["best_%d" % k] = np.asarray(ctrl.last_plan["best_index"])
        out["native"] = np.asarray(int(ctrl._cstep is not None))
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_device_rng_plan_does_not_depend_on_the_world_size(world, tmp_path):
    """`rng="device"` at N > 1 (round 6: `l2a_controller_create_sharded_device`): every rank fills its slice of the SAME Philox
    stream, so the sharded plan picks the candidates - indices and fp32 first actions - that ONE process planning the whole case
    picks, step after step, and every rank returns them (the winner's action recomputed from the stream, no second collective)."""
    mp.spawn(_worker_device, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, HERE)
    import cases
    case = dict(cases.CASES["hc_rs_m3_n64_h5"], n=333, h=4)
    ctrl = cases.product_controller(case, rng="device")
    torch.manual_seed(1234)
    obs = np.random.RandomState(3).randn(case["m"], 20)
    outs = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    for k in range(3):
        want, _ = ctrl.get_actions(obs)
        for o in outs:
            assert int(o["native"]) == 1
            assert np.array_equal(o["best_%d" % k], ctrl.last_plan["best_index"])
            np.testing.assert_array_equal(o["actions_%d" % k], want)


def _worker_rnn(rank, world, port, cid, out_dir):
    _setup(rank, world, port)
    try:
        import cases
        case, seed = cases.split_id(cid)
        gold = cases.load_golden(cid)
        ctrl = cases.product_rnn_controller(case)
        ctrl.reset(dones=[True] * case["m"])
        np.random.seed(seed)
        out = {}
        for k in range(int(gold["obs"].shape[0])):
            actions, _ = ctrl.get_actions(gold["obs"][k])
            out["actions_%d" % k] = actions
            out["best_%d" % k] = np.asarray(ctrl.last_plan["best_index"])
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("cid,world", [("c6_hc_rnn_rs_n500_h10_m5_s0", 2), ("hc_rnn_rs_gru2_n48_h4_s0", 3),
                                       ("c6g_hc_rnn_rs_gru256_n500_h10_m5_s0", 2), ("ant_rnn_cem_gru2x256_n200_h4_m2_s0", 3)])
def test_sharded_recurrent_plan_on_the_gpu_equals_the_reference_plan(cid, world, tmp_path):
    """The recurrent planner (ReBAL default: LSTM 256; a GRU stack on the generic kernel; GRU 256 / a CEM plan on a 2 x 256 GRU stack on
    the generic micro-tile kernel) sharded over ranks: the payload is
    packed through the recurrent model (`NativeLSTM` had no `plan_payload` - the first sharded step raised) and
    every controller step of the golden replay is the reference's on every rank."""
    mp.spawn(_worker_rnn, args=(world, _free_port(), cid, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, HERE)
    import cases
    gold = cases.load_golden(cid)
    for r in range(world):
        o = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        for k in range(int(gold["obs"].shape[0])):
            assert np.array_equal(o["best_%d" % k], gold["best_%d" % k]), (r, k)
            np.testing.assert_array_equal(o["actions_%d" % k], gold["chosen_%d" % k])
