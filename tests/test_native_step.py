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
