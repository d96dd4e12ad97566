"""Recurrent planner (ReBAL): oracle vs the golden vectors of the real
``RNNMPCController``, the LSTM restatement vs an independent implementation, host logic of the drop-in
classes on CPU (launch replaced by the oracle) and - marked ``gpu`` - parity of the fused recurrent
rollout through the C ABI."""

import pickle

import numpy as np
import pytest
import torch

import cases
import oracle_backend
from oracle import LSTMStateTuple, lstm_step_f32, make_reward, rnn_cem_plan, rnn_rs_plan

RNN_IDS = cases.rnn_case_ids()


def _flat(hidden):
    """Any hidden-state structure (LSTMStateTuple / array / list or tuple of those) -> flat (c, h) [rows, W]."""
    layers = [hidden] if (isinstance(hidden, np.ndarray) or hasattr(hidden, "_fields")) else list(hidden)
    cs = [np.asarray(st[0]) if hasattr(st, "_fields") else np.zeros_like(np.asarray(st)) for st in layers]
    hs = [np.asarray(st[1]) if hasattr(st, "_fields") else np.asarray(st) for st in layers]
    return np.concatenate(cs, axis=1), np.concatenate(hs, axis=1)


def _zero_rows(hidden, dones):
    layers = [hidden] if (isinstance(hidden, np.ndarray) or hasattr(hidden, "_fields")) else list(hidden)
    for st in layers:
        for part in (st if hasattr(st, "_fields") else [st]):
            part[dones] = 0.0


def _replay(case, gold, plan_step, reset_hook):
    """Drive ``plan_step(k, obs) -> (chosen, hidden_after)`` over the recorded controller steps."""
    resets = {int(k): v for k, v in case.get("reset_after", {}).items()}
    out = []
    for k in range(case["steps"]):
        out.append(plan_step(k, gold["obs"][k]))
        if k in resets:
            reset_hook(np.array(resets[k], dtype=bool))
    return out


# ------------------------------------------------------------------------------------------ oracle (CPU)
@pytest.mark.parametrize("cid", RNN_IDS)
def test_oracle_reproduces_reference_rnn_controller(cid):
    case, seed = cases.split_id(cid)
    gold = cases.load_golden(cid)
    env, _, _ = cases.rnn_recipe(case)
    dyn = cases.oracle_rnn_dynamics(case)
    reward = make_reward(case["env"], env.dt)
    state = {"hid": dyn.get_initial_hidden(case["m"])}
    np.random.seed(seed)

    def step(k, obs):
        args = (dyn, reward, obs, state["hid"], env.action_space.low, env.action_space.high, case["n"], case["h"],
                case.get("discount", 1.0))
        if case["planner"] == "rnn_cem":
            chosen, best, returns, hid = rnn_cem_plan(*args, num_cem_iters=case["num_cem_iters"])
        else:
            chosen, best, returns, hid = rnn_rs_plan(*args)
        state["hid"] = hid
        assert np.array_equal(returns, gold["returns_%d" % k])
        assert np.array_equal(best, gold["best_%d" % k])
        assert np.array_equal(chosen, gold["chosen_%d" % k])
        c, h = _flat(hid)
        assert np.array_equal(c, gold["hidden_c_%d" % k]) and np.array_equal(h, gold["hidden_h_%d" % k])

    def reset(dones):
        _zero_rows(state["hid"], dones)

    _replay(case, gold, step, reset)
    assert np.random.uniform() == float(gold["rng_next"])        # RNG consumption


def test_lstm_restatement_matches_torch_lstm_cell():
    """Independent check of the cell arithmetic: torch.nn.LSTMCell (gate order i, f, g, o; no forget
    bias) fed with the re-ordered TF-layout weights and the forget bias folded into b_f."""
    rs = np.random.RandomState(0)
    n_in, U, B = 26, 64, 9
    kernel = (0.2 * rs.randn(n_in + U, 4 * U)).astype(np.float32)
    bias = (0.1 * rs.randn(4 * U)).astype(np.float32)
    x = rs.randn(B, n_in).astype(np.float32)
    c0 = rs.randn(B, U).astype(np.float32)
    h0 = np.tanh(rs.randn(B, U)).astype(np.float32)
    c1, h1 = lstm_step_f32(x, c0, h0, kernel, bias)
    cell = torch.nn.LSTMCell(n_in, U)
    i, j, f, o = np.split(kernel, 4, axis=1)
    bi, bj, bf, bo = np.split(bias, 4)
    w = np.concatenate([i, f, j, o], axis=1)                      # torch order: i, f, g(=j), o
    with torch.no_grad():
        cell.weight_ih.copy_(torch.from_numpy(w[:n_in].T.copy()))
        cell.weight_hh.copy_(torch.from_numpy(w[n_in:].T.copy()))
        cell.bias_ih.copy_(torch.from_numpy(np.concatenate([bi, bf + 1.0, bj, bo])))
        cell.bias_hh.zero_()
        h_t, c_t = cell(torch.from_numpy(x), (torch.from_numpy(h0), torch.from_numpy(c0)))
    np.testing.assert_allclose(c1, c_t.numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(h1, h_t.numpy(), rtol=2e-5, atol=2e-6)
    c64, h64 = lstm_step_f32(x, c0, h0, kernel, bias, dtype=np.float64)
    np.testing.assert_allclose(c1, c64, rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------------------------------ host logic (CPU)
@pytest.mark.parametrize("cid", [c for c in RNN_IDS if cases.split_id(c)[0]["h"] < 6])      # (longer horizons: pipelined launches)
def test_rnn_controller_host_logic_matches_golden(cid):
    case, seed = cases.split_id(cid)
    gold = cases.load_golden(cid)
    ctrl = oracle_backend.install_rnn(cases.product_rnn_controller(case), case)
    ctrl.reset(dones=[True] * case["m"])
    np.random.seed(seed)

    def step(k, obs):
        chosen, info = ctrl.get_actions(obs)
        assert info == {}
        assert np.array_equal(ctrl.last_plan["best_index"], gold["best_%d" % k])
        assert np.array_equal(chosen, gold["chosen_%d" % k])
        c, h = _flat(ctrl._hidden_state)
        assert np.array_equal(c, gold["hidden_c_%d" % k]) and np.array_equal(h, gold["hidden_h_%d" % k])

    _replay(case, gold, step, lambda dones: ctrl.reset(dones=dones))
    assert np.random.uniform() == float(gold["rng_next"])


def test_rnn_model_surface_and_pickling():
    case = cases.CASES["hc_rnn_rs_u128_n40_h3"]
    env, model = cases.product_rnn_model(case)
    assert model.recurrent is True
    hid = model.get_initial_hidden(3)
    assert isinstance(hid, tuple) and hid.c.shape == (3, 128) and hid.h.dtype == np.float32 and not hid.c.any()
    twin = pickle.loads(pickle.dumps(model))
    for a, b in zip(model.get_param_values().values(), twin.get_param_values().values()):
        assert np.array_equal(a, b)
    assert list(twin.get_param_values()) == ["rnn/lstm_cell/kernel", "rnn/lstm_cell/bias", "output/kernel",
                                             "output/bias"]
    assert np.array_equal(twin.normalization["obs"][0], model.normalization["obs"][0])
    ctrl = cases.product_rnn_controller(case, model=model, env=env)
    ctrl2 = pickle.loads(pickle.dumps(ctrl))
    assert ctrl2.n_candidates == case["n"] and ctrl2.percent_elites == 0.05 and ctrl2._hidden_state is None
    with pytest.raises(NotImplementedError):
        type(model)(name="x", env=env, hidden_sizes=(64,), cell_type="ugrnn")
    # the other cells of create_rnn (core/utils.py:199-220): parameter names / shapes and hidden-state structures
    gru = type(model)(name="x", env=env, hidden_sizes=(64,), cell_type="gru", init_seed=0)
    assert list(gru.get_param_values()) == ["rnn/gru_cell/gates/kernel", "rnn/gru_cell/gates/bias",
                                            "rnn/gru_cell/candidate/kernel", "rnn/gru_cell/candidate/bias",
                                            "output/kernel", "output/bias"]
    assert gru.get_param_values()["rnn/gru_cell/gates/kernel"].shape == (26 + 64, 128)
    assert np.all(gru.get_param_values()["rnn/gru_cell/gates/bias"] == 1.0)        # GRUCell's bias initialiser
    assert isinstance(gru.get_initial_hidden(2), np.ndarray) and gru.get_initial_hidden(2).shape == (2, 64)
    stack = type(model)(name="x", env=env, hidden_sizes=(48, 32), init_seed=0)
    assert list(stack.get_param_values())[:4] == ["rnn/multi_rnn_cell/cell_0/lstm_cell/kernel",
                                                  "rnn/multi_rnn_cell/cell_0/lstm_cell/bias",
                                                  "rnn/multi_rnn_cell/cell_1/lstm_cell/kernel",
                                                  "rnn/multi_rnn_cell/cell_1/lstm_cell/bias"]
    assert stack.get_param_values()["rnn/multi_rnn_cell/cell_1/lstm_cell/kernel"].shape == (48 + 32, 128)
    hid = stack.get_initial_hidden(3)
    assert isinstance(hid, list) and len(hid) == 2 and hid[1].h.shape == (3, 32)
    c, h = stack.pack_hidden(hid)
    assert c.shape == h.shape == (3, 80)
    back = stack.unpack_hidden(c, h, as_tuple=True)
    assert isinstance(back, tuple) and back[0].c.shape == (3, 48)
    twin = pickle.loads(pickle.dumps(stack))
    assert all(np.array_equal(a, b) for a, b in zip(stack.get_param_values().values(), twin.get_param_values().values()))


def test_rnn_fit_reduces_loss_on_a_learnable_sequence():
    """Truncated-BPTT training (stock PyTorch): a linear system with memory is learnt."""
    from learning_to_adapt_amd.dynamics import RNNDynamicsModel
    from learning_to_adapt_amd.envs import SyntheticEnv
    env = SyntheticEnv("half_cheetah")
    od, ad = 20, 6
    rs = np.random.RandomState(0)
    paths, T = 12, 24
    act = rs.uniform(-1, 1, (paths, T, ad))
    obs = np.zeros((paths, T + 1, od))
    obs[:, 0] = rs.randn(paths, od)
    A = 0.1 * rs.randn(ad, od)
    for t in range(T):
        obs[:, t + 1] = 0.95 * obs[:, t] + act[:, t] @ A
    model = RNNDynamicsModel(name="dyn", env=env, hidden_sizes=(32,), learning_rate=1e-2, batch_size=4,
                             backprop_steps=8, init_seed=0)
    np.random.seed(0)
    torch.manual_seed(0)

    def valid_loss():
        from learning_to_adapt_amd.dynamics import core
        from learning_to_adapt_amd.dynamics.rnn_dynamics import lstm_forward
        nm = model.normalization
        x = np.concatenate([core.normalize(obs[:, :-1], *nm["obs"]), core.normalize(act, *nm["act"])], axis=2)
        y = core.normalize(obs[:, 1:] - obs[:, :-1], *nm["delta"])
        dev = core.training_device()
        z = torch.zeros((paths, 32), device=dev)
        with torch.no_grad():
            pred, _, _ = lstm_forward(torch.as_tensor(x, dtype=torch.float32, device=dev), z, z,
                                      [p.to(dev) for p in model._params], "tanh", None)
        return float(np.mean((pred.cpu().numpy() - y) ** 2))

    model.fit(obs[:, :-1], act, obs[:, 1:], epochs=1, valid_split_ratio=0.2)
    first = valid_loss()
    stats = model.fit(obs[:, :-1], act, obs[:, 1:], epochs=30, valid_split_ratio=0.2, compute_normalization=False)
    assert valid_loss() < 0.6 * first
    assert set(stats) == {"AvgModelEpochTime", "Epochs"}


# ------------------------------------------------------------------------------------------ GPU parity
def _tol_returns(got, want):
    """Largest error of any candidate's return, relative PER ELEMENT (floor 1.0) like ``rel_err`` of test_gpu_parity.py -
    until round 3 the recurrent tests divided by the table's largest return, a looser bar for small returns."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    return float(np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want))))


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["auto", "valu"])
@pytest.mark.parametrize("cid", RNN_IDS)
def test_gpu_rnn_controller_matches_golden(cid, kernel):
    """The drop-in controller over several consecutive steps (hidden state carried, one reset): chosen
    index bit-exact, chosen action bit-exact, hidden state and returns within fp32 tolerance."""
    from learning_to_adapt_amd import _lib
    case, seed = cases.split_id(cid)
    if kernel == "valu" and case["n"] * case["h"] * case["m"] > 6000:
        pytest.skip("VALU kernel: small cases only")
    gold = cases.load_golden(cid)
    ctrl = cases.product_rnn_controller(case)
    ctx = _lib.Context.get(0)
    ctx.set_kernel(kernel)
    try:
        ctrl.reset(dones=[True] * case["m"])
        np.random.seed(seed)

        def step(k, obs):
            hidden_before = ctrl.dynamics_model.unpack_hidden(*[np.array(a) for a in _flat(ctrl._hidden_state)])
            state_before = np.random.get_state()
            chosen, _ = ctrl.get_actions(obs)
            if case["planner"] == "rnn_cem" and "cem_trace" in ctrl.last_plan:
                # CEM feeds the returns back through a rank-based elite mask (reference mpc_controller.py:100-104): a rank swap of
                # two candidates closer than the rollout's fp32 error legitimately changes the later iterations.  Every swap must
                # be a PROVEN tie (tests/cem_ties.py: a witness pair in the reference's own returns); with one inside the mask the
                # replay stops here as an xfail - never a silent skip (tests/test_gpu_parity.py does the same for the MLP planner)
                import cem_ties
                tie_rtol = 2e-5 if case["env"].startswith("ant") else 1e-5
                kk = max(int(case["n"] * ctrl.percent_elites), 1)        # (rnn_mpc_controller.py:41: 5 % elites, refit without alpha)
                for it, tr in enumerate(ctrl.last_plan["cem_trace"]):
                    assert _tol_returns(tr["returns"], gold["cem_returns_%d" % k][it]) < tie_rtol
                    swaps, mask_flips, worst = cem_ties.assert_flips_are_ties(tr["returns"], gold["cem_returns_%d" % k][it], kk, tie_rtol)
                    if mask_flips:
                        # ... and what the product did after the tie is checked against the oracle continued from ITS statistics
                        from oracle.rnn_planner import rnn_rollout_returns
                        dyn, reward_fn = cases.oracle_rnn_dynamics(case), make_reward(case["env"], ctrl.env.dt)
                        trace = ctrl.last_plan["cem_trace"]
                        again = cem_ties.verify_tail_from_product(
                            lambda seq: rnn_rollout_returns(dyn, reward_fn, np.asarray(obs, dtype=np.float64), hidden_before, seq,
                                                            case["n"], case.get("discount", 1.0)),
                            ctrl.env.action_space.low, ctrl.env.action_space.high, case["n"], case["m"], case["h"], ctrl.alpha, kk,
                            state_before, trace, it, ctrl.last_plan["best_index"], chosen, tie_rtol)
                        pytest.xfail("rank tie inside the CEM elite mask at step %d, iteration %d: %d mask flips, largest witness gap "
                                     "%.1e relative (error bar %.0e); the %d later iterations, the chosen index and action verified "
                                     "against the oracle continued from the product's statistics (%d tied again)"
                                     % (k, it, mask_flips, worst, 2 * tie_rtol, len(trace) - it - 1, again))
            margin = gold["margin_%d" % k]
            safe = margin > 1e-4 * np.maximum(1.0, np.abs(gold["returns_%d" % k]).max(axis=1))
            assert np.array_equal(ctrl.last_plan["best_index"][safe], gold["best_%d" % k][safe])
            assert np.array_equal(chosen[safe], gold["chosen_%d" % k][safe])
            if "best_return" in ctrl.last_plan:
                want = gold["returns_%d" % k][np.arange(case["m"]), gold["best_%d" % k]]
                np.testing.assert_allclose(ctrl.last_plan["best_return"][safe], want[safe], rtol=1e-4, atol=1e-4)
            if safe.all():
                c, h = _flat(ctrl._hidden_state)
                np.testing.assert_allclose(c, gold["hidden_c_%d" % k], rtol=1e-4, atol=2e-5)
                np.testing.assert_allclose(h, gold["hidden_h_%d" % k], rtol=1e-4, atol=2e-5)

        _replay(case, gold, step, lambda dones: ctrl.reset(dones=dones))
        if case["planner"] == "rnn_rs":
            assert np.random.uniform() == float(gold["rng_next"])
    finally:
        ctx.set_kernel("auto")


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["mfma", "valu"])
@pytest.mark.parametrize("name", ["hc_rnn_rs_m2_n64_h4_reset", "hc_rnn_rs_u128_n40_h3", "ant_rnn_rs_n100_h5_m2",
                                  "hc_rnn_rs_m2_n64_h4_reset:relu", "hc_rnn_rs_u128_n40_h3:swish",
                                  "hc_rnn_rs_lstm2_n48_h4", "hc_rnn_rs_gru2_n48_h4", "hc_rnn_rs_gru1_u96_n40_h3",
                                  "arm_rnn_rs_rnn1_u80_n40_h3", "hc_rnn_rs_rnn3_n32_h3:sigmoid"])
def test_gpu_rnn_returns_table_matches_oracle(name, kernel):
    """Every candidate's return of one recurrent plan step, from a NON-zero hidden state (also with
    other cell activations than the default tanh)."""
    from learning_to_adapt_amd import _lib
    from oracle.rnn_planner import rnn_rollout_returns
    name, _, act = name.partition(":")
    case = dict(cases.CASES[name])
    if act:
        case["activation"] = act
    # "mfma" on stacks / GRU / RNN cells = the generic matrix-core kernel (l2a_rnn_mfma.h), "valu" = l2a_rnn_valu_k
    env, model = cases.product_rnn_model(case)
    dyn = cases.oracle_rnn_dynamics(case)
    reward = make_reward(case["env"], env.dt)
    rs = np.random.RandomState(5)
    m, n, h, U = case["m"], case["n"], case["h"], case["units"]
    obs0 = rs.randn(m, env.observation_space.shape[0])
    flat = LSTMStateTuple(rs.randn(m, U).astype(np.float32), np.tanh(rs.randn(m, U)).astype(np.float32))
    hid = model.unpack_hidden(flat.c, flat.h)            # the model's own structure (a stack: list of layer states)
    if case.get("cell_type", "lstm") != "lstm":
        flat = LSTMStateTuple(np.zeros_like(flat.c), flat.h)
    acts = rs.uniform(env.action_space.low, env.action_space.high, (h, m * n, env.action_space.shape[0]))
    want = rnn_rollout_returns(dyn, reward, obs0, hid, acts, n, case.get("discount", 1.0)).reshape(m, n)
    hid = flat
    ctx = _lib.Context.get(0)
    ctx.set_kernel(kernel)
    try:
        native = model.planner_model()
        dev = native.device
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)  # noqa: E731
        rets = torch.empty((m, n), dtype=torch.float32, device=dev)
        best = torch.zeros((m,), dtype=torch.int64, device=dev)
        native.plan_rs(up(obs0), up(hid.c), up(hid.h), up(acts), m, n, h, case.get("discount", 1.0),
                       env.reward_spec, returns_out=rets, best_key=best)
        got = rets.cpu().numpy()
        assert _tol_returns(got, want) < 1e-4
        keys = best.cpu().numpy()
        for i in range(m):
            ret, idx = _lib.key_decode(keys[i])
            assert idx == int(np.argmax(got[i])) and ret == got[i, idx]
    finally:
        ctx.set_kernel("auto")


@pytest.mark.gpu
@pytest.mark.parametrize("units", [128, 256, 512, 200])
def test_gpu_rnn_predict_matches_oracle(units):
    """``RNNDynamicsModel.predict``: per-row hidden states, ragged row count."""
    case = dict(cases.CASES["hc_rnn_rs_u128_n40_h3"], units=units)
    env, model = cases.product_rnn_model(case)
    dyn = cases.oracle_rnn_dynamics(case)
    rs = np.random.RandomState(2)
    rows = 37
    obs = rs.randn(rows, 20)
    act = rs.uniform(-1, 1, (rows, 6))
    hid = LSTMStateTuple(rs.randn(rows, units).astype(np.float32), np.tanh(rs.randn(rows, units)).astype(np.float32))
    want, whid = dyn.predict(obs, act, hid)
    got, ghid = model.predict(obs, act, hid)
    assert got.dtype == np.float64 and ghid.c.shape == (rows, units)
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ghid.c, whid.c, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(ghid.h, whid.h, rtol=1e-5, atol=2e-6)


@pytest.mark.gpu
def test_gpu_rnn_full_size_properties():
    """ReBAL default size (LSTM 256, n = 500, h = 10, 5 envs) and a chip-filling plan: determinism,
    sharding invariance (max of shard keys == key of the full plan, bit for bit), ragged prefix
    (the first n' candidates of a plan return the same bits whatever follows them), n = 1 / h = 1."""
    from learning_to_adapt_amd import _lib
    case = cases.CASES["c6_hc_rnn_rs_n500_h10_m5"]
    env, model = cases.product_rnn_model(case)
    native = model.planner_model()
    dev = native.device
    gen = torch.Generator(device=dev)
    gen.manual_seed(0)
    for m, n, h in ((5, 500, 10), (1, 4096, 12), (2, 1, 1)):
        obs0 = torch.randn((m, 20), generator=gen, device=dev)
        c0 = torch.randn((m, 256), generator=gen, device=dev)
        h0 = torch.tanh(torch.randn((m, 256), generator=gen, device=dev))
        a = torch.rand((h, m * n, 6), generator=gen, device=dev) * 2 - 1

        def plan(acts, n_, off=0):
            rets = torch.empty((m, n_), dtype=torch.float32, device=dev)
            best = torch.zeros((m,), dtype=torch.int64, device=dev)
            native.plan_rs(obs0, c0, h0, acts.contiguous(), m, n_, h, 0.97, env.reward_spec, cand_offset=off,
                           returns_out=rets, best_key=best)
            return rets.cpu().numpy(), best.cpu().numpy()

        r1, k1 = plan(a, n)
        r2, k2 = plan(a, n)
        assert np.array_equal(r1, r2) and np.array_equal(k1, k2)            # deterministic
        assert np.all(np.isfinite(r1))
        for i in range(m):
            ret, idx = _lib.key_decode(k1[i])
            assert idx == int(np.argmax(r1[i])) and ret == r1[i, idx]
        if n >= 64:
            a4 = a.reshape(h, m, n, 6)
            shards = 4
            keys = []
            for s in range(shards):
                lo, hi = s * n // shards, (s + 1) * n // shards
                rs_, ks = plan(a4[:, :, lo:hi].reshape(h, m * (hi - lo), 6), hi - lo, off=lo)
                assert np.array_equal(rs_, r1[:, lo:hi])
                keys.append(ks)
            assert np.array_equal(np.max(np.stack(keys), axis=0), k1)
            npre = 37
            rp, _ = plan(a4[:, :, :npre].reshape(h, m * npre, 6), npre)
            assert np.array_equal(rp, r1[:, :npre])


@pytest.mark.gpu
def test_gpu_rnn_invalid_calls_are_rejected():
    from learning_to_adapt_amd import _lib
    from learning_to_adapt_amd.dynamics.native_lstm import NativeLSTM
    from learning_to_adapt_amd.envs import RewardSpec
    nat = NativeLSTM(20, 6, 256)
    dev = nat.device
    z = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)  # noqa: E731
    best = torch.zeros((1,), dtype=torch.int64, device=dev)
    spec = RewardSpec.half_cheetah(20, 0.01)
    with pytest.raises(_lib.L2AError, match="never set"):
        nat.plan_rs(z(1, 20), z(1, 256), z(1, 256), z(2, 16, 6), 1, 16, 2, 1.0, spec, best_key=best)
    case = cases.CASES["hc_rnn_rs_m2_n64_h4_reset"]
    _, params, norm = cases.rnn_recipe(case)
    nat.set_weights(params)
    with pytest.raises(_lib.L2AError, match="normalisation"):
        nat.plan_rs(z(1, 20), z(1, 256), z(1, 256), z(2, 16, 6), 1, 16, 2, 1.0, spec, best_key=best)
    nat.set_norm(norm)
    nat.plan_rs(z(1, 20), z(1, 256), z(1, 256), z(2, 16, 6), 1, 16, 2, 1.0, spec, best_key=best)
    with pytest.raises(_lib.L2AError, match="nothing to write"):
        nat.plan_rs(z(1, 20), z(1, 256), z(1, 256), z(2, 16, 6), 1, 16, 2, 1.0, spec)
    bad = RewardSpec.make(w_vel=1.0, dt=1.0, vel_index=99)
    with pytest.raises(_lib.L2AError, match="vel_index"):
        nat.plan_rs(z(1, 20), z(1, 256), z(1, 256), z(2, 16, 6), 1, 16, 2, 1.0, bad, best_key=best)
    with pytest.raises(AssertionError):
        nat.set_weights(params[:3])
    # a width the tuned LSTM kernel is not instantiated for: the model of l2a_lstm_create refuses kernel='mfma' ...
    ctx = _lib.Context.get(0)
    lib = _lib.load()
    assert lib.l2a_lstm_mfma_eligible(20, 6, 256) == 1 and lib.l2a_lstm_mfma_eligible(20, 6, 200) == 0
    import ctypes
    handle = ctypes.c_void_p()
    ctx.check(lib.l2a_lstm_create(ctx.handle, 20, 6, 200, _lib.ACT_CODES["tanh"], _lib.ACT_CODES[None], ctypes.byref(handle)),
              "l2a_lstm_create")
    odd = NativeLSTM.__new__(NativeLSTM)
    odd.ctx, odd.lib, odd.device, odd.handle, odd._keep = ctx, lib, nat.device, handle, {}
    odd.layer_units, odd.cell_type, odd.obs_dim, odd.act_dim, odd.units = (200,), "lstm", 20, 6, 200
    odd.set_weights(synthetic_lstm(200))
    odd.set_norm(norm)
    ctx.set_kernel("mfma")
    try:
        with pytest.raises(_lib.L2AError, match="not eligible"):
            odd.plan_rs(z(1, 20), z(1, 200), z(1, 200), z(2, 16, 6), 1, 16, 2, 1.0, spec, best_key=best)
    finally:
        ctx.set_kernel("auto")
    odd.close()
    # ... while NativeLSTM (l2a_rnn_create) gives such a layer the generic matrix-core kernel: same returns as its VALU twin
    gen = NativeLSTM(20, 6, 200)
    gen.set_weights(synthetic_lstm(200))
    gen.set_norm(norm)
    a = torch.rand((2, 16, 6), device=nat.device) * 2 - 1
    out = {}
    try:
        for kernel in ("mfma", "valu"):
            ctx.set_kernel(kernel)
            rets = torch.empty((1, 16), dtype=torch.float32, device=nat.device)
            gen.plan_rs(z(1, 20) + 0.1, z(1, 200), z(1, 200) + 0.2, a, 1, 16, 2, 1.0, spec, returns_out=rets, best_key=best)
            out[kernel] = rets.cpu().numpy()
    finally:
        ctx.set_kernel("auto")
    np.testing.assert_allclose(out["mfma"], out["valu"], rtol=1e-5, atol=1e-6)


def synthetic_lstm(units):
    from learning_to_adapt_amd.utils import synthetic
    return synthetic.make_lstm_set(20, 6, units, 1000)


@pytest.mark.gpu
def test_gpu_rnn_unfused_path_matches_fused():
    """An env whose reward has no closed-form spec goes through the reference's loop shape with the LSTM
    step on the GPU (`dynamics_model.predict`); same decision as the fused kernel."""
    case = cases.CASES["hc_rnn_rs_m2_n64_h4_reset"]
    gold = cases.load_golden("hc_rnn_rs_m2_n64_h4_reset_s0")
    env, model = cases.product_rnn_model(case)
    fused = cases.product_rnn_controller(case, model=model, env=env)
    fused.reset(dones=[True, True])
    np.random.seed(0)
    a_fused, _ = fused.get_actions(gold["obs"][0])

    class OpaqueEnv(object):
        def __init__(self, inner):
            self.inner = inner
            self.action_space, self.observation_space, self.dt = inner.action_space, inner.observation_space, inner.dt

        def reward(self, obs, act, nxt):
            return self.inner.reward(obs, act, nxt)

    plain = cases.product_rnn_controller(case, model=model, env=OpaqueEnv(env))
    assert not plain._fusable()
    plain.reset(dones=[True, True])
    np.random.seed(0)
    a_plain, _ = plain.get_actions(gold["obs"][0])
    np.testing.assert_array_equal(a_plain, a_fused)
    np.testing.assert_array_equal(a_plain, gold["chosen_0"])
    np.testing.assert_allclose(plain._hidden_state.c, fused._hidden_state.c, rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("cem_mode", ["reference", "fixed"])
def test_gpu_rnn_device_cem_matches_host_loop_with_injected_normals(cem_mode):
    """Recurrent CEM with sampling / elites / refit on the GPU vs the host loop on the same normals."""
    case = dict(cases.CASES["hc_rnn_cem_n200_h5_m2"])
    gold = cases.load_golden("hc_rnn_cem_n200_h5_m2_s0")
    n, m, D = case["n"], case["m"], case["h"] * 6
    zs = [np.random.RandomState(300 + i).normal(size=(n, m, D)) for i in range(case["num_cem_iters"])]
    env, model = cases.product_rnn_model(case)
    host = cases.product_rnn_controller(case, model=model, env=env, cem_mode=cem_mode)
    host.reset(dones=[True] * m)
    it = iter(zs)
    host._cem_draw = lambda n_, m_, D_: next(it).reshape(n_ * m_, D_)      # inject the iteration's normals
    a_host, _ = host.get_actions(gold["obs"][0])
    dev = cases.product_rnn_controller(case, model=model, env=env, rng="device", cem_mode=cem_mode)
    dev.reset(dones=[True] * m)
    it2 = iter(zs)
    dev._cem_normal_device = lambda shape, device: torch.from_numpy(next(it2).astype(np.float32)).to(device)
    a_dev, _ = dev.get_actions(gold["obs"][0])
    tr = host.last_plan["cem_trace"][-1]
    np.testing.assert_allclose(dev.last_plan["cem_mean"], np.broadcast_to(tr["mean"], (m, D)), rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(dev.last_plan["cem_std"], np.broadcast_to(tr["std"], (m, D)), rtol=1e-3, atol=1e-3)
    assert np.array_equal(dev.last_plan["best_index"], host.last_plan["best_index"])
    np.testing.assert_allclose(a_dev, a_host, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(dev._hidden_state.c, host._hidden_state.c, rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("cid", ["c6_hc_rnn_rs_n500_h10_m5_s0", "ant_rnn_rs_n100_h5_m2_s0", "hc_rnn_rs_u200_n40_h3_s0"])
def test_gpu_rnn_pipelined_controller_equals_single_launch(cid):
    """Recurrent parity mode pipelined over the horizon (observation, LSTM state and returns handed from launch
    to launch) vs one launch per plan step: same RNG consumption, same actions, same return bits, same hidden
    state - over consecutive controller steps."""
    case, seed = cases.split_id(cid)
    gold = cases.load_golden(cid)
    env, model = cases.product_rnn_model(case)
    runs = []
    for chunks in (3, 1, 2):
        ctrl = cases.product_rnn_controller(case, model=model, env=env, pipeline_chunks=chunks)
        ctrl.reset(dones=[True] * case["m"])
        np.random.seed(seed)
        rec = []
        for k in range(case["steps"]):
            a, _ = ctrl.get_actions(gold["obs"][k])
            rec.append((a, ctrl.last_plan["best_index"].copy(), ctrl.last_plan["best_return"].copy(),
                        ctrl._hidden_state.c.copy()))
        runs.append(rec)
    for rec in runs[1:]:
        for got, want in zip(rec, runs[0]):
            for x, y in zip(got, want):
                assert np.array_equal(x, y)
    if case["h"] >= 6:
        assert np.array_equal(runs[0][0][1], gold["best_0"])


@pytest.mark.parametrize("name", ["hc_rnn_rs_lstm2_n48_h4", "hc_rnn_rs_gru2_n48_h4", "hc_rnn_rs_gru1_u96_n40_h3",
                                  "arm_rnn_rs_rnn1_u80_n40_h3", "hc_rnn_rs_rnn3_n32_h3"])
def test_cell_restatements_match_the_torch_forward(name):
    """The NumPy restatement of GRUCell / BasicRNNCell / MultiRNNCell (oracle/rnn_cells.py) against the package's
    independent stock-PyTorch forward pass used for training (dynamics/rnn_cells.stack_forward), two steps."""
    from learning_to_adapt_amd.dynamics import core, rnn_cells
    case = cases.CASES[name]
    env, params, norm = cases.rnn_recipe(case)
    dyn = cases.oracle_rnn_dynamics(case)
    rs = np.random.RandomState(0)
    rows = 7
    od, ad = env.observation_space.shape[0], env.action_space.shape[0]
    hid = dyn.get_initial_hidden(rows)
    obs = rs.randn(rows, od)
    tparams = [torch.from_numpy(np.asarray(p)) for p in params]
    tstate = rnn_cells.zero_state(case["cell_type"], case["hidden_sizes"], rows, "cpu")
    for _ in range(2):
        act = rs.uniform(env.action_space.low, env.action_space.high, (rows, ad))
        want, hid = dyn.predict(obs, act, hid)
        x = np.concatenate([core.normalize(obs, *norm["obs"]), core.normalize(act, *norm["act"])], axis=1)
        pred, tstate = rnn_cells.stack_forward(torch.as_tensor(x[:, None, :], dtype=torch.float32), tstate, tparams,
                                               case["hidden_sizes"], case["cell_type"], case.get("activation", "tanh"), None)
        got = obs + core.denormalize(pred[:, 0].numpy().astype(np.float64), *norm["delta"])
        np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5)
        obs = want
    c, h = _flat(hid)
    th = torch.cat([s[1] if isinstance(s, tuple) else s for s in tstate], dim=1).numpy()
    np.testing.assert_allclose(h, th, rtol=2e-5, atol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("units,m,n,h", [(256, 1, 2000, 6), (256, 2, 500, 5), (128, 1, 333, 4), (512, 3, 100, 3),
                                          (256, 5, 500, 4)])
def test_gpu_rnn_unit_tile_split_is_bit_identical(units, m, n, h):
    """The LSTM kernel's unit-tile split (two workgroups per candidate tile, halves of h and of the output layer's
    sum exchanged once per step; plans of at most CUs / 2 tiles) against the unsplit launch: returns, keys, the
    handed-over chunk state and `predict` bit for bit.  (5 x 32 tiles = the ReBAL default is too large to split:
    both policies run the same launch there.)"""
    from learning_to_adapt_amd import _lib
    case = dict(cases.CASES["hc_rnn_rs_u128_n40_h3"], units=units, m=m, n=n, h=h)
    env, model = cases.product_rnn_model(case)
    native = model.planner_model()
    dev = native.device
    ctx = _lib.Context.get(0)
    rs = np.random.RandomState(units + n)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)  # noqa: E731
    obs0, c0 = up(rs.randn(m, 20)), up(rs.randn(m, units))
    h0 = up(np.tanh(rs.randn(m, units)))
    acts = up(rs.uniform(-1, 1, (h, m * n, 6)))
    out = {}
    try:
        for policy in (0, 1):
            ctx.set_split(policy)
            rets = torch.empty((m, n), dtype=torch.float32, device=dev)
            best = torch.zeros((m,), dtype=torch.int64, device=dev)
            native.plan_rs(obs0, c0, h0, acts, m, n, h, 0.98, env.reward_spec, cand_offset=7, returns_out=rets, best_key=best)
            # the same plan in two chunks (per-candidate state handed over)
            r1 = torch.empty((m, n), dtype=torch.float32, device=dev)
            r2 = torch.empty((m, n), dtype=torch.float32, device=dev)
            st = torch.empty((m * n, 20), dtype=torch.float32, device=dev)
            cc = torch.empty((m * n, units), dtype=torch.float32, device=dev)
            hh = torch.empty((m * n, units), dtype=torch.float32, device=dev)
            k2 = torch.zeros((m,), dtype=torch.int64, device=dev)
            h1 = h // 2
            native.plan_rs_chunk(obs0, c0, h0, False, acts[:h1].contiguous(), m, n, h1, 0, 0.98, env.reward_spec, cand_offset=7,
                                 returns_out=r1, state_out=st, c_out=cc, h_out=hh)
            native.plan_rs_chunk(st, cc, hh, True, acts[h1:].contiguous(), m, n, h - h1, h1, 0.98, env.reward_spec,
                                 cand_offset=7, returns_in=r1, returns_out=r2, best_key=k2)
            rows = min(m * n, 48)
            nxt, c1, hn = native.predict(up(rs.randn(rows, 20) * 0 + 0.3), up(np.zeros((rows, 6))), cc[:rows].contiguous(),
                                         hh[:rows].contiguous())
            torch.cuda.synchronize()
            ctx.launch_status()
            out[policy] = [t.cpu().numpy() for t in (rets, best, r2, k2, st, cc, hh, nxt, c1, hn)]
    finally:
        ctx.set_split(1)
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)
    assert np.array_equal(out[0][0], out[0][2]) and np.array_equal(out[0][1], out[0][3])     # chunks == one launch


@pytest.mark.gpu
@pytest.mark.parametrize("units,m,n,h,env_name", [(256, 5, 500, 10, "hc"), (256, 1, 2000, 6, "hc"), (256, 2, 37, 3, "hc"), (256, 3, 333, 4, "hc"),
                                                  (256, 1, 5, 2, "hc"), (512, 2, 500, 3, "hc"), (256, 5, 250, 4, "hc"), (256, 64, 12, 2, "hc"),
                                                  (256, 5, 500, 5, "ant"), (256, 2, 100, 3, "arm")])
def test_gpu_rnn_micro_tiles_are_bit_identical(units, m, n, h, env_name):
    """The micro-tile kernel (csrc/l2a_micro.h: candidate tiles of FOUR on the 4x4x1 MFMA, workgroups of 4 / 8 / 12
    candidates, no exchange) against the 16-candidate kernel, split and unsplit: every return and the arg-max keys bit
    for bit, from non-zero hidden states, with a discount, ragged last tiles, more envs than a workgroup per CU allows
    three micro tiles for (falls back), through the plain and the blocking entry point."""
    from learning_to_adapt_amd import _lib
    base = {"hc": "hc_rnn_rs_u128_n40_h3"}.get(env_name, "hc_rnn_rs_u128_n40_h3")
    case = dict(cases.CASES[base], units=units, m=m, n=n, h=h)
    if env_name != "hc":
        case["env"] = {"ant": "ant", "arm": "arm_7dof"}[env_name]
    env, model = cases.product_rnn_model(case)
    native = model.planner_model()
    dev = native.device
    od, ad = env.observation_space.shape[0], env.action_space.shape[0]
    ctx = _lib.Context.get(0)
    rs = np.random.RandomState(units + n + m)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)  # noqa: E731
    obs0, c0 = up(rs.randn(m, od)), up(rs.randn(m, units))
    h0 = up(np.tanh(rs.randn(m, units)))
    lo, hi_ = env.action_space.low, env.action_space.high
    acts = up(rs.uniform(lo, hi_, (h, m * n, ad)))
    out = {}
    try:
        for micro, split in ((0, 0), (0, 1), (2, 1)):
            ctx.set_micro(micro)
            ctx.set_split(split)
            rets = torch.empty((m, n), dtype=torch.float32, device=dev)
            best = torch.zeros((m,), dtype=torch.int64, device=dev)
            native.plan_rs(obs0, c0, h0, acts, m, n, h, 0.97, env.reward_spec, cand_offset=11, returns_out=rets, best_key=best)
            keys_only = torch.zeros((m,), dtype=torch.int64, device=dev)
            native.plan_rs(obs0, c0, h0, acts, m, n, h, 0.97, env.reward_spec, cand_offset=11, best_key=keys_only)
            torch.cuda.synchronize()
            ctx.launch_status()
            out[(micro, split)] = [t.cpu().numpy() for t in (rets, best, keys_only)]
    finally:
        ctx.set_micro(1)
        ctx.set_split(1)
    ref = out[(0, 0)]
    assert np.array_equal(ref[1], ref[2])
    assert np.isfinite(ref[0]).all()
    for k in ((0, 1), (2, 1)):
        for a, b in zip(ref, out[k]):
            assert np.array_equal(a, b), k


@pytest.mark.gpu
@pytest.mark.parametrize("cell,units,m,n,h", [("lstm", (256,), 5, 500, 10), ("lstm", (256,), 1, 2000, 6),
                                              ("lstm", (128,), 3, 77, 4), ("gru", (64,), 2, 90, 3),
                                              ("lstm", (64, 32), 2, 50, 3)])
def test_gpu_blocking_recurrent_plan_matches_the_plain_launches(cell, units, m, n, h):
    """`l2a_lstm_plan_rs_sync` (observations from host-mapped staging, keys through the mailbox, the controller's state
    advanced in stream order with the winning first actions) against `l2a_lstm_plan_rs` + `l2a_lstm_advance` with the
    action picked on the host: keys and next state bit for bit - MFMA kernel (mailbox, split and unsplit plans) and the
    generic kernels (copy + synchronise), repeated calls, changing m."""
    from learning_to_adapt_amd import _lib
    case = dict(cases.CASES["hc_rnn_rs_u128_n40_h3"], units=units[0], m=m, n=n, h=h)
    if cell != "lstm" or len(units) > 1:
        case.update(cell_type=cell, hidden_sizes=list(units))
    env, model = cases.product_rnn_model(case)
    native = model.planner_model()
    dev = native.device
    U = native.units
    rs = np.random.RandomState(U + n)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)  # noqa: E731
    c0, h0 = up(rs.randn(m, U)), up(np.tanh(rs.randn(m, U)))
    for it in range(4):
        obs = rs.randn(m, 20)
        acts = up(rs.uniform(-1, 1, (h, m * n, 6)))
        best = torch.zeros((m,), dtype=torch.int64, device=dev)
        native.plan_rs(up(obs), c0, h0, acts, m, n, h, 0.97, env.reward_spec, cand_offset=11 * it, best_key=best)
        want = best.cpu().numpy()
        idx = np.array([_lib.key_decode(k)[1] for k in want]) - 11 * it
        chosen = acts[0].reshape(m, n, 6)[torch.arange(m), torch.from_numpy(idx).to(dev)]
        # the state step: `l2a_lstm_advance` with the action picked on the host - bit for bit (one LSTM layer: the small-rows
        # kernel, which the blocking launch feeds through the keys instead; other cells: one step of the rollout kernel) - and
        # `l2a_lstm_predict`'s states to fp32 rounding (the small-rows kernel sums its 4 x (in + U) / 4 products in another order)
        c_want, h_want = native.advance(up(obs), chosen.contiguous(), c0, h0)
        _, c_pred, h_pred = native.predict(up(obs), chosen.contiguous(), c0, h0)
        c1 = torch.full((m, U), float("nan"), device=dev)
        h1 = torch.full((m, U), float("nan"), device=dev)
        got = native.plan_rs_sync(obs, c0, h0, acts, m, n, h, 0.97, env.reward_spec, cand_offset=11 * it, c_next=c1, h_next=h1)
        assert got is not None and np.array_equal(got.view(np.int64), want), (it, got, want)
        assert torch.equal(c1, c_want) and torch.equal(h1, h_want), it
        assert float((c1 - c_pred).abs().max()) < 2e-6 and float((h1 - h_pred).abs().max()) < 2e-6, it
        keys_only = native.plan_rs_sync(obs, c0, h0, acts, m, n, h, 0.97, env.reward_spec, cand_offset=11 * it)
        assert np.array_equal(keys_only, got)
        c0, h0 = c1, h1
    torch.cuda.synchronize()
    _lib.Context.get(0).launch_status()


@pytest.mark.gpu
@pytest.mark.parametrize("cell,hidden,m,n,h,env_name,act", [
    ("gru", (256,), 5, 500, 10, "hc", "tanh"), ("lstm", (256, 256), 5, 500, 10, "hc", "tanh"), ("rnn", (256,), 5, 500, 10, "hc", "tanh"),
    ("gru", (256, 256), 1, 2000, 4, "hc", "tanh"), ("rnn", (256, 256, 256), 2, 37, 3, "hc", "relu"), ("lstm", (256, 256, 256), 3, 333, 3, "ant", "tanh"),
    ("gru", (256,), 64, 12, 2, "arm", "sigmoid"), ("lstm", (256, 256), 1, 5, 2, "hc", "swish"), ("gru", (256, 256, 256), 5, 250, 4, "ant", "tanh"),
    # plans beyond three micro tiles per CU: workgroups of four micro tiles, one round (875 quads on 219 workgroups) and several
    ("gru", (256, 256), 1, 3500, 3, "hc", "tanh"), ("lstm", (256, 256), 3, 1500, 3, "hc", "tanh"), ("rnn", (256,), 2, 3001, 2, "ant", "relu"),
    ("gru", (256,), 1, 16000, 2, "hc", "tanh")])
def test_gpu_rnn_generic_micro_tiles_match_the_16_candidate_kernel_and_the_oracle(cell, hidden, m, n, h, env_name, act):
    """The micro-tile form of the generic recurrent kernel (csrc/l2a_rnn_micro.h: GRU / BasicRNN / LSTM stacks of 256-unit layers on
    candidate tiles of four, workgroups of up to three of them - or, for larger plans, of four, in several rounds) against the 16-candidate kernel (l2a_rnn_mfma.h; the hidden layers sum in the same order, the output
    layer per wave and then over the waves) and against the oracle's cells (oracle/rnn_cells.py <- dynamics/core/utils.py:192-236):
    every return, the arg-max key, from non-zero hidden states, with a discount, ragged last tiles, keys-only launches."""
    from learning_to_adapt_amd import _lib
    from oracle.rnn_planner import rnn_rollout_returns
    case = dict(cases.CASES["hc_rnn_rs_gru2_n48_h4"], units=sum(hidden), m=m, n=n, h=h, cell_type=cell, hidden_sizes=list(hidden),
                activation=act)
    if env_name != "hc":
        case["env"] = {"ant": "ant", "arm": "arm_7dof"}[env_name]
    env, model = cases.product_rnn_model(case)
    dyn = cases.oracle_rnn_dynamics(case)
    reward = make_reward(case["env"], env.dt)
    native = model.planner_model()
    dev = native.device
    od, ad, U = env.observation_space.shape[0], env.action_space.shape[0], sum(hidden)
    rs = np.random.RandomState(U + n + m)
    obs0 = rs.randn(m, od)
    flat = LSTMStateTuple((rs.randn(m, U) if cell == "lstm" else np.zeros((m, U))).astype(np.float32), np.tanh(rs.randn(m, U)).astype(np.float32))
    acts = rs.uniform(env.action_space.low, env.action_space.high, (h, m * n, ad))
    want = rnn_rollout_returns(dyn, reward, obs0, model.unpack_hidden(flat.c, flat.h), acts, n, 0.97).reshape(m, n)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)  # noqa: E731
    ctx = _lib.Context.get(0)
    out = {}
    try:
        for micro in (0, 1, 2):
            ctx.set_micro(micro)
            rets = torch.empty((m, n), dtype=torch.float32, device=dev)
            best = torch.zeros((m,), dtype=torch.int64, device=dev)
            native.plan_rs(up(obs0), up(flat.c), up(flat.h), up(acts), m, n, h, 0.97, env.reward_spec, cand_offset=11, returns_out=rets, best_key=best)
            keys_only = torch.zeros((m,), dtype=torch.int64, device=dev)
            native.plan_rs(up(obs0), up(flat.c), up(flat.h), up(acts), m, n, h, 0.97, env.reward_spec, cand_offset=11, best_key=keys_only)
            torch.cuda.synchronize()
            ctx.launch_status()
            out[micro] = [t.cpu().numpy() for t in (rets, best, keys_only)]
    finally:
        ctx.set_micro(1)
    for micro in (0, 1, 2):
        got, keys, keys_only = out[micro]
        assert np.isfinite(got).all()
        assert _tol_returns(got, want) < 1e-4, micro
        assert np.array_equal(keys, keys_only)
        for i in range(m):
            ret, idx = _lib.key_decode(keys[i])
            assert idx - 11 == int(np.argmax(got[i])) and ret == got[i, idx - 11]
    assert np.array_equal(out[1][0], out[2][0])                                  # every one of these plans is eligible: policy 1 = 2
    # the two kernels differ by the output layer's summation order only: a few ulps of the LARGEST partial return (the returns
    # are sums of up to h rewards of either sign), far inside the bar against the oracle above
    assert float(np.max(np.abs(out[2][0] - out[0][0])) / max(1.0, float(np.max(np.abs(out[0][0]))))) < 2e-6
    assert _tol_returns(out[2][0], out[0][0]) < 1e-4
