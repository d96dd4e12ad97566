"""The controller step as one C call (`l2a_controller_step` / `l2a_lstm_controller_step`, policies/native_step.py) against the
reference planner's golden vectors and against the step-by-step Python path it replaces: same actions, indices, returns, same
consumption of NumPy's global generator, a foreign draw between two steps makes it fall back (and re-arm) without a trace."""

import numpy as np
import pytest
import torch

import cases
from learning_to_adapt_amd import _lib

pytestmark = pytest.mark.gpu

RS_IDS = [cid for cid in cases.case_ids() if cases.split_id(cid)[0]["planner"] == "rs"]


@pytest.mark.parametrize("cid", RS_IDS)
def test_native_step_reproduces_the_reference_golden(cid):
    """The golden call itself served by the C controller: a warm-up call builds it, the generator is seeded like the
    reference run, the chain re-armed at that state (what the controller does after any synchronous draw) - the next step
    adopts the block and must return the reference's action, index and generator position."""
    case, seed = cases.split_id(cid)
    gold = cases.load_golden(cid)
    ctrl = cases.product_controller(case)
    np.random.seed(12345)
    ctrl.get_actions(gold["obs0"])                     # builds the controller; no block yet: the C step draws synchronously, arms the chain
    assert ctrl._cstep is not None, "the C controller does not apply to this case"
    np.random.seed(seed)
    ctrl._cstep.misses_in_row = 0
    ctrl._cstep.rearm()
    before = ctrl._cstep.stats()["hits"]
    actions, info = ctrl.get_actions(gold["obs0"])
    assert info == {} and ctrl._cstep.stats()["hits"] == before + 1
    assert np.random.uniform() == float(gold["rng_next"])
    assert np.array_equal(ctrl.last_plan["best_index"], gold["best"])
    np.testing.assert_array_equal(actions, gold["chosen"])
    want = gold["returns"][np.arange(case["m"]), gold["best"]]
    assert np.max(np.abs(ctrl.last_plan["best_return"] - want) / np.maximum(1.0, np.abs(want))) < 1e-4
    ctrl._cstep.close()
    ctrl._cstep = None


@pytest.mark.parametrize("name", ["c2_hc_rs_n2000_h30_e5", "c3b_ant_rs_n500_h10_pb5_3x512", "c1_hc_rs_n500_h10_e1", "hc_rs_m3_n64_h5"])
def test_native_step_equals_the_python_path(name):
    case = cases.CASES[name]
    env, model = cases.product_model(case)
    rs = np.random.RandomState(5)
    obs = [rs.randn(case["m"], env.observation_space.shape[0]) for _ in range(7)]
    outs = []
    for native in (False, True):
        ctrl = cases.product_controller(case, model=model, env=env, native_step=native)
        np.random.seed(11)
        seq = []
        for k in range(7):
            a, _ = ctrl.get_actions(obs[k])
            seq.append((a.copy(), np.array(ctrl.last_plan["best_index"]), np.array(ctrl.last_plan["best_return"])))
            if k == 2:
                np.random.normal(size=3)           # a foreign draw: the prepared block must be dropped
        outs.append((seq, np.random.uniform(), ctrl))
    (s0, t0, c0), (s1, t1, c1) = outs
    assert t0 == t1
    for (a0, i0, r0), (a1, i1, r1) in zip(s0, s1):
        assert np.array_equal(a0, a1) and np.array_equal(i0, i1) and np.array_equal(r0, r1)
    assert c0._cstep is None and c1._cstep is not None
    st = c1._cstep.stats()
    # steps 0 (nothing armed yet) and 3 (foreign draw: the block drawn ahead was stale) drew synchronously INSIDE the C step
    assert st["hits"] == 5 and st["misses"] == 1 and st["steps"] == 7 and st["sync_draws"] == 2
    assert c1.draw_ahead_stats()["hits"] == 5
    c1._cstep.close()
    c1._cstep = None
    if c0._ahead is not None:
        c0._ahead.stop()


@pytest.mark.parametrize("name", ["c6_hc_rnn_rs_n500_h10_m5", "c6g_hc_rnn_rs_gru256_n500_h10_m5", "hc_rnn_rs_gru2_n48_h4"])
def test_native_recurrent_step_equals_the_python_path(name):
    """Plan + state advance in one call: actions AND the controller's hidden state after every step are those of the Python
    path; a `reset` of one env in the middle (host arrays rewritten, device copy stale) is picked up."""
    case = dict(cases.CASES[name])
    case.pop("reset_after", None)
    env, model = cases.product_rnn_model(case)
    m = case["m"]
    rs = np.random.RandomState(6)
    obs = [rs.randn(m, env.observation_space.shape[0]) for _ in range(6)]
    outs = []
    for native in (False, True):
        ctrl = cases.product_rnn_controller(case, model=model, env=env)
        ctrl.native_step = native
        ctrl.reset(dones=[True] * m)
        np.random.seed(3)
        seq = []
        for k in range(6):
            a, _ = ctrl.get_actions(obs[k])
            c, h = ctrl._pack(ctrl._hidden_state)
            seq.append((a.copy(), np.array(ctrl.last_plan["best_index"]), np.array(c), np.array(h)))
            if k == 3:
                dones = [False] * m
                dones[0] = True
                ctrl.reset(dones=dones)
        outs.append((seq, np.random.uniform(), ctrl))
    (s0, t0, c0), (s1, t1, c1) = outs
    assert t0 == t1
    for (a0, i0, cc0, hh0), (a1, i1, cc1, hh1) in zip(s0, s1):
        assert np.array_equal(a0, a1) and np.array_equal(i0, i1)
        assert np.array_equal(cc0, cc1) and np.array_equal(hh0, hh1)
    assert c1._cstep is not None and c1._cstep.stats()["hits"] == 5 and c1._cstep.stats()["sync_draws"] == 1
    c1._cstep.close()
    c1._cstep = None
    if c0._ahead is not None:
        c0._ahead.stop()


def test_native_step_survives_a_flagged_launch():
    """A tile-split launch that loses its partner inside the C step: the step repeats it unsplit (same bits), tells the
    caller, and the context stays degraded - the action is the reference's."""
    ctx = _lib.Context.get(0)
    cid = "c2_hc_rs_n2000_h30_e5_s1"
    case, seed = cases.split_id(cid)
    gold = cases.load_golden(cid)
    ctrl = cases.product_controller(case)
    try:
        ctx.set_split(1)
        ctx.split_degraded = False
        np.random.seed(999)
        ctrl.get_actions(gold["obs0"])
        np.random.seed(seed)
        ctrl._cstep.misses_in_row = 0
        ctrl._cstep.rearm()
        ctx.set_spin_limit(1)               # one poll per launch: a split workgroup almost surely misses its partner
        actions, _ = ctrl.get_actions(gold["obs0"])
        assert np.array_equal(ctrl.last_plan["best_index"], gold["best"])
        np.testing.assert_array_equal(actions, gold["chosen"])
        st = ctrl._cstep.stats()
        assert st["hits"] >= 1
        if st["relaunches"]:
            assert ctx.split_degraded
    finally:
        ctx.set_spin_limit(0)
        ctx.set_split(1)
        ctx.split_degraded = False
        torch.cuda.synchronize()
        ctx.launch_status_value()
        if ctrl._cstep is not None:
            ctrl._cstep.close()
            ctrl._cstep = None


def test_native_step_stage_table_adds_up():
    case = cases.CASES["c1_hc_rs_n500_h10_e1"]
    ctrl = cases.product_controller(case)
    obs = np.random.RandomState(0).randn(1, 20)
    np.random.seed(0)
    for _ in range(5):
        ctrl.get_actions(obs)
    s = ctrl._cstep.stats()["stage_us"]
    parts = s["take"] + s["stage_obs"] + s["launch"] + s["kick"] + s["wait"] + s["decode"]
    assert 0 < parts <= s["call"] + 1.0 and s["call"] - parts < 20.0
    ctrl._cstep.close()
    ctrl._cstep = None


@pytest.mark.parametrize("cid", [c for c in RS_IDS if c.endswith("_s0")])
def test_the_synchronous_draw_inside_the_c_step_reproduces_the_reference_golden(cid):
    """No block waiting (the first call of a controller): the C step draws the candidates itself - the reference's draw from the
    GLOBAL generator on the helper's threads - and must leave the generator where the reference leaves it."""
    case, seed = cases.split_id(cid)
    gold = cases.load_golden(cid)
    ctrl = cases.product_controller(case)
    np.random.seed(seed)
    actions, _ = ctrl.get_actions(gold["obs0"])
    assert ctrl._cstep is not None and ctrl._cstep.stats()["sync_draws"] == 1 and ctrl._cstep.stats()["hits"] == 0
    assert np.random.uniform() == float(gold["rng_next"])
    assert np.array_equal(ctrl.last_plan["best_index"], gold["best"])
    np.testing.assert_array_equal(actions, gold["chosen"])
    ctrl._cstep.close()
    ctrl._cstep = None


def test_steps_that_keep_missing_back_off():
    """A consumer of np.random between every two steps: every step draws synchronously (bit-identical results: the golden tests),
    and after the third miss in a row the chain is re-armed only every 16th step - no block is drawn ahead just to be thrown away."""
    case = cases.CASES["c1_hc_rs_n500_h10_e1"]
    ctrl = cases.product_controller(case)
    obs = np.random.RandomState(0).randn(1, 20)
    np.random.seed(0)
    for _ in range(40):
        ctrl.get_actions(obs)
        np.random.uniform()
    st = ctrl._cstep.stats()
    assert st["sync_draws"] == 40 and st["hits"] == 0
    assert st["produced"] <= 6              # 2 early re-arms + every 16th step
    for _ in range(40):                     # the foreign consumer stops: the chain comes back within 16 steps and then hits
        ctrl.get_actions(obs)
    assert ctrl._cstep.stats()["hits"] >= 20
    ctrl._cstep.close()
    ctrl._cstep = None


# ---- rng="device" through the C controller: candidates from the library's counter-based Philox stream ---------------------------
def _philox_uniform_ref(seed, e0, count, act_dim, low, high):
    """NumPy restatement of csrc/l2a_philox.h: elements e0 .. e0 + count - 1 of the uniform stream (four per Philox4x32-10 block,
    counter (e >> 2), domain word 'unif'), value = fma(high - low, (word >> 8) / 2^24, low) in fp32, dimension = local index % act_dim."""
    M = np.uint64(0xFFFFFFFF)
    e = np.arange(e0, e0 + count, dtype=np.uint64)
    ctr = e >> np.uint64(2)
    c = [ctr & M, (ctr >> np.uint64(32)) & M, np.full(count, 0x756E6966, dtype=np.uint64), np.zeros(count, dtype=np.uint64)]
    k0, k1 = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = np.uint64(0xD2511F53) * c[0], np.uint64(0xCD9E8D57) * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ k0) & M, p1 & M, ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & M, p0 & M]
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & M, (k1 + np.uint64(0xBB67AE85)) & M
    words = np.stack(c, axis=1)[np.arange(count), (e & np.uint64(3)).astype(np.int64)]
    u = ((words >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)).astype(np.float64)
    k = (np.arange(count) % act_dim)
    lo32 = np.asarray(low, dtype=np.float32)
    rng32 = np.asarray(high, dtype=np.float32) - lo32
    # one rounding (the product of two 24-bit significands and the sum with a 24-bit addend are exact in float64 here)
    return (rng32[k].astype(np.float64) * u + lo32[k].astype(np.float64)).astype(np.float32)


@pytest.mark.parametrize("name", ["hc_rs_m3_n64_h5", "c1_hc_rs_n500_h10_e1"])
def test_device_rng_step_plans_on_the_librarys_philox_stream(name):
    """`MPCController(rng="device")` on one GPU is one C call too: a Philox kernel fills the candidate tensor in front of the plan
    and the HOST recomputes the winner's first action from the same counter-based stream.  Against a NumPy restatement of the
    stream + the oracle: the chosen index, its return, and the returned action = exactly the fp32 candidate the kernel planned on;
    step k uses block k of the stream; same torch seed -> same plan; NumPy's generator is not touched."""
    from oracle import make_reward
    from oracle.planner import rollout_returns
    case = cases.CASES[name]
    env, model = cases.product_model(case)
    dyn, reward = cases.oracle_dynamics(case), make_reward(case["env"], env.dt)
    m, n, h, ad = case["m"], case["n"], case["h"], env.action_space.shape[0]
    obs = np.random.RandomState(4).randn(m, env.observation_space.shape[0])
    per_step = (h * m * n * ad + 3) // 4 * 4
    runs = []
    for rep in range(2):
        ctrl = cases.product_controller(case, model=model, env=env, rng="device")
        torch.manual_seed(77)
        state = np.random.get_state()[1].copy()
        seq = []
        for k in range(3):
            a, _ = ctrl.get_actions(obs)
            seq.append((a.copy(), np.array(ctrl.last_plan["best_index"]), np.array(ctrl.last_plan["best_return"])))
        assert ctrl._cstep is not None and ctrl._cstep.device_rng and ctrl._cstep.stats()["steps"] == 3
        assert np.array_equal(np.random.get_state()[1], state)
        runs.append(seq)
        ctrl._cstep.close()
        ctrl._cstep = None
    for (a0, i0, r0), (a1, i1, r1) in zip(*runs):
        assert np.array_equal(a0, a1) and np.array_equal(i0, i1) and np.array_equal(r0, r1)
    for k, (a, idx, ret) in enumerate(runs[0]):
        cand = _philox_uniform_ref(77, k * per_step, h * m * n * ad, ad, env.action_space.low, env.action_space.high)
        cand = cand.reshape(h, m * n, ad).astype(np.float64)
        rets = rollout_returns(dyn, reward, obs, cand, n, case.get("discount", 1.0)).reshape(m, n)
        want = np.argmax(rets, axis=1)
        top2 = np.sort(rets, axis=1)[:, -2:]
        clear = (top2[:, 1] - top2[:, 0]) > 1e-4 * np.maximum(1.0, np.abs(top2[:, 1]))
        assert np.array_equal(idx[clear], want[clear])
        assert np.all(np.abs(ret - rets[np.arange(m), idx]) <= 1e-4 * np.maximum(1.0, np.abs(ret)))
        np.testing.assert_array_equal(a, cand[0].reshape(m, n, ad)[np.arange(m), idx])


def test_device_rng_recurrent_step_runs_through_the_c_controller():
    case = dict(cases.CASES["c6_hc_rnn_rs_n500_h10_m5"])
    ctrl = cases.product_rnn_controller(case, rng="device")
    obs = np.random.RandomState(0).randn(5, 20)
    ctrl.reset(dones=[True] * 5)
    torch.manual_seed(5)
    outs = [ctrl.get_actions(obs)[0].copy() for _ in range(3)]
    assert ctrl._cstep is not None and ctrl._cstep.device_rng and ctrl._cstep.stats()["steps"] == 3
    assert all(o.shape == (5, 6) and np.all(np.abs(o) <= 1.0) for o in outs)
    assert not np.array_equal(outs[0], outs[1])                 # another block of the stream every step
    c, h = ctrl._pack(ctrl._hidden_state)
    assert np.isfinite(c).all() and np.abs(h).max() <= 1.0 and np.abs(h).max() > 0.0       # the state was advanced
    ctrl._cstep.close()
    ctrl._cstep = None


@pytest.mark.parametrize("collective", ["rccl", "callback"])
def test_sharded_c_step_with_one_rank_reproduces_the_reference_golden(collective):
    """`l2a_controller_create_sharded` with world = 1 runs the SAME code path as a multi-rank step - slice draw, launch into a
    device key slot, payload packed on the device, the collective, the page-locked read-back, digest and flag checks - with a
    one-rank collective: the library's own RCCL communicator (`l2a_comm_init` + `l2a_allreduce_best` on m + 3 words; RCCL cannot
    run two ranks on one GPU, the multi-rank protocol is tests/test_distributed_gpu.py's with gloo behind the callback) or a
    Python callback.  Must return the reference planner's golden plan, use the draw-ahead chain and leave np.random where the
    reference leaves it."""
    import ctypes
    from learning_to_adapt_amd.policies.native_step import NativeStep
    cid = "hc_rs_m3_n64_h5_s0"
    case, seed = cases.split_id(cid)
    gold = cases.load_golden(cid)
    env, model = cases.product_model(case)
    native = model.planner_model()
    ctx = native.ctx
    seen = []
    if collective == "rccl":
        buf = ctypes.create_string_buffer(128)
        ctx.check(native.lib.l2a_comm_unique_id(buf), "l2a_comm_unique_id")
        ctx.check(native.lib.l2a_comm_init(ctx.handle, 0, 1, buf.raw), "l2a_comm_init")
        reduce = None
    else:
        def reduce(payload):
            assert payload.is_cuda and payload.dtype == torch.int64 and payload.numel() == case["m"] + 3
            seen.append(payload.cpu().numpy().copy())
    try:
        st = NativeStep(native, False, case["m"], case["n"], case["h"], env.action_space.low, env.action_space.high, 1.0,
                        env.reward_spec, shard=(0, 1, reduce))
        stream = torch.cuda.current_stream(native.device).cuda_stream
        np.random.seed(seed)
        assert st.step(gold["obs0"], stream)
        assert np.array_equal(st.idx, gold["best"])
        np.testing.assert_array_equal(st.act, gold["chosen"])
        # steps 2 and 3 adopt blocks the chain drew ahead; they must be what an unsharded controller returns
        ctrl = cases.product_controller(case, model=model, env=env)
        np.random.seed(seed)
        ctrl.get_actions(gold["obs0"])
        state_after_1 = np.random.get_state()
        outs = []
        for _ in range(2):
            a, _ = ctrl.get_actions(gold["obs0"])
            outs.append((a.copy(), np.array(ctrl.last_plan["best_index"])))
        want_next = np.random.uniform()
        np.random.set_state(state_after_1)
        st.rearm()
        for k in range(2):
            assert st.step(gold["obs0"], stream)
            np.testing.assert_array_equal(st.act, outs[k][0])
            assert np.array_equal(st.idx, outs[k][1])
        assert np.random.uniform() == want_next
        assert st.stats()["hits"] >= 1
        if collective == "callback":
            assert len(seen) == 3 and all(int(v[case["m"]]) == 0 for v in seen)       # keys | flag | digest pair
            assert all(int(v[case["m"] + 1]) + int(v[case["m"] + 2]) == 0x7FFFFFFFFFFF for v in seen)
        st.close()
        if ctrl._cstep is not None:
            ctrl._cstep.close()
            ctrl._cstep = None
    finally:
        if collective == "rccl":
            native.lib.l2a_comm_destroy(ctx.handle)
