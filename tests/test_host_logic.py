"""Host side of the drop-in API on a CPU-only box: with the device launch replaced by the
oracle (tests/oracle_backend.py) the package's ``MPCController`` must reproduce the reference
planner's golden vectors - same RNG consumption, same chosen action."""

import pickle

import numpy as np
import pytest

import cases
import oracle_backend


@pytest.mark.parametrize("cid", cases.case_ids(max_work=2000 * 30 * 5 + 1))
def test_controller_host_logic_matches_reference(cid):
    case, seed = cases.split_id(cid)
    gold = cases.load_golden(cid)
    ctrl = oracle_backend.install(cases.product_controller(case), case)
    np.random.seed(seed)
    actions, info = ctrl.get_actions(gold["obs0"])
    assert info == {}
    assert np.random.uniform() == float(gold["rng_next"])
    assert np.array_equal(ctrl.last_plan["best_index"], gold["best"])
    np.testing.assert_array_equal(actions, gold["chosen"])
    if case["planner"] == "cem":
        for it, tr in enumerate(ctrl.last_plan["cem_trace"]):
            np.testing.assert_allclose(np.broadcast_to(tr["mean"], gold["cem_mean"][it].shape),
                                       gold["cem_mean"][it], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(tr["std"], gold["cem_std"][it], rtol=1e-4, atol=1e-6)


def test_get_action_adds_batch_axis():
    case = cases.CASES["hc_rs_n1_h1"]
    gold = cases.load_golden("hc_rs_n1_h1_s0")
    ctrl = oracle_backend.install(cases.product_controller(case), case)
    np.random.seed(0)
    a, info = ctrl.get_action(gold["obs0"][0])
    assert a.shape == (1, 6) and info == {}
    np.testing.assert_array_equal(a, gold["chosen"])


def test_cem_fixed_mode_is_clipped_and_topk():
    case = dict(cases.CASES["hc_cem_m2_n100_h4"])
    ctrl = oracle_backend.install(cases.product_controller(case, cem_mode="fixed"), case)
    np.random.seed(0)
    obs0 = cases.load_golden("hc_cem_m2_n100_h4_s0")["obs0"]
    actions, _ = ctrl.get_actions(obs0)
    assert actions.shape == (2, 6)
    assert np.all(actions >= -1.0) and np.all(actions <= 1.0)      # reference mode may exceed the bounds
    tr = ctrl.last_plan["cem_trace"][-1]
    assert tr["mean"].shape == (2, 24) and tr["std"].shape == (2, 24)


def test_models_and_controller_pickle_roundtrip():
    case = cases.CASES["hc_rs_m2_n100_h7_e2"]
    env, model = cases.product_model(case)
    ctrl = cases.product_controller(case, model=model, env=env)
    clone = pickle.loads(pickle.dumps(ctrl))
    assert clone.n_candidates == 100 and clone.horizon == 7
    for e in range(2):
        a, b = model.get_param_values(e), clone.dynamics_model.get_param_values(e)
        assert list(a.keys()) == ["hidden_0/kernel", "hidden_0/bias", "hidden_1/kernel", "hidden_1/bias",
                                  "output/kernel", "output/bias"]
        assert all(np.array_equal(a[k], b[k]) for k in a)
    np.testing.assert_array_equal(clone.dynamics_model.normalization["obs"][0], model.normalization["obs"][0])

    case = cases.CASES["c3b_ant_rs_n500_h10_pb5_3x512"]
    env, meta = cases.product_model(case)
    clone = pickle.loads(pickle.dumps(meta))
    a, b = meta.get_param_values(), clone.get_param_values()
    assert all(np.array_equal(a[k], b[k]) for k in a)


def test_adapt_is_one_sgd_step_per_env():
    """MetaMLPDynamicsModel.adapt (reference meta_mlp_dynamics.py:321-345,409-421) against a
    NumPy finite-difference-free restatement: theta' = theta - alpha * dL/dtheta with
    L = mean((delta_norm - MLP(x_norm))**2) over the real rows."""
    import torch
    from learning_to_adapt_amd.dynamics import MetaMLPDynamicsModel
    from learning_to_adapt_amd.envs import SyntheticEnv
    from learning_to_adapt_amd.utils import synthetic
    env = SyntheticEnv("half_cheetah")
    model = MetaMLPDynamicsModel(name="m", env=env, hidden_sizes=(32, 32), inner_learning_rate=0.05, init_seed=3)
    norm = synthetic.make_norm(20, 6, env.action_space.low, env.action_space.high, 2000)
    model.set_normalization(norm)
    rs = np.random.RandomState(0)
    obs = [rs.randn(16, 20) for _ in range(3)]
    act = [rs.uniform(-1, 1, (16, 6)) for _ in range(3)]
    nxt = [o + 0.1 * rs.randn(16, 20) for o in obs]
    base = [p.clone() for p in model._params]
    model.adapt(obs, act, nxt)
    assert model.planner_blocks(3) == 3 and model.mode == "per_block"
    for i in range(3):
        x = np.concatenate([(obs[i] - norm["obs"][0]) / (norm["obs"][1] + 1e-10),
                            (act[i] - norm["act"][0]) / (norm["act"][1] + 1e-10)], axis=1)
        y = ((nxt[i] - obs[i]) - norm["delta"][0]) / (norm["delta"][1] + 1e-10)
        ps = [p.clone().double().requires_grad_(True) for p in base]
        t = torch.from_numpy(x)
        for li in range(3):
            t = t @ ps[2 * li] + ps[2 * li + 1]
            if li < 2:
                t = torch.relu(t)
        loss = torch.mean((torch.from_numpy(y) - t) ** 2)
        grads = torch.autograd.grad(loss, ps)
        for p, g, got in zip(ps, grads, model._adapted_param_values[i]):
            np.testing.assert_allclose(got.cpu().numpy(), (p - 0.05 * g).detach().numpy(), rtol=2e-4, atol=2e-6)
    model.switch_to_pre_adapt()
    assert model.mode == "single" and model._adapted_param_values is None
    assert all(torch.equal(a, b) for a, b in zip(base, model._params))


def test_fit_reduces_loss_on_a_linear_system():
    from learning_to_adapt_amd.dynamics import MLPDynamicsModel
    from learning_to_adapt_amd.envs import SyntheticEnv
    env = SyntheticEnv("half_cheetah", obs_dim=5, act_dim=2)
    rs = np.random.RandomState(0)
    A, B = 0.1 * rs.randn(5, 5), 0.1 * rs.randn(2, 5)
    obs = rs.randn(2000, 5)
    act = rs.uniform(-1, 1, (2000, 2))
    nxt = obs + obs @ A + act @ B
    model = MLPDynamicsModel(name="d", env=env, hidden_sizes=(32, 32), hidden_nonlinearity="relu",
                             batch_size=200, learning_rate=1e-2, init_seed=0)
    np.random.seed(0)
    stats = model.fit(obs, act, nxt, epochs=15)
    assert model.normalization is not None and stats["Epochs"] >= 1
    import torch
    from learning_to_adapt_amd.dynamics import core
    x = np.concatenate(model._normalize_data(obs, act), axis=1)
    pred = core.mlp_forward(torch.as_tensor(x, dtype=torch.float32), model._param_sets[0], "relu", None).numpy()
    target = core.normalize(nxt - obs, *model.normalization["delta"])
    assert np.mean((pred - target) ** 2) < 0.3      # an untrained net sits at ~1.0 (unit-variance targets)


def test_fast_rng_reproduces_numpy_global_stream():
    """`utils/fast_rng.random_sample` (vectorised MT19937 helper, csrc/l2a_rng.c) against
    `np.random.random_sample`: same doubles, same generator state afterwards (cached Gaussian included),
    for sizes that start / end inside a 624-word state block."""
    from learning_to_adapt_amd.utils import fast_rng
    if not fast_rng.available():
        pytest.skip("libl2a_rng.so not built (no gcc) - the controller then uses np.random.random_sample itself")
    for seed in range(4):
        for sizes in ([2048], [4097, 2049, 3000], [(700, 6), (1, 4096)], [312 * 7 + 1, 5000]):
            np.random.seed(seed)
            np.random.normal(size=3)                      # odd stream position + a cached Gaussian
            want = [np.random.random_sample(s) for s in sizes]
            tail_want = (np.random.normal(), np.random.uniform())
            np.random.seed(seed)
            np.random.normal(size=3)
            got = [fast_rng.random_sample(s) for s in sizes]
            tail_got = (np.random.normal(), np.random.uniform())
            for a, b in zip(want, got):
                assert a.shape == b.shape and np.array_equal(a, b)
            assert tail_want == tail_got
    np.random.seed(0)
    small = fast_rng.random_sample((3, 2))                # below the threshold: NumPy's own call
    np.random.seed(0)
    assert np.array_equal(small, np.random.random_sample((3, 2)))


def test_meta_fit_splits_accumulates_and_validates_on_held_out_paths():
    """MetaMLPDynamicsModel.fit against the reference's contract (meta_mlp_dynamics.py:167-268): paths
    are split by valid_split_ratio, both parts are APPENDED to the datasets of earlier calls (the trainer passes
    only the newest rollouts), validation runs on the held-out part, and training reduces the loss on data the
    optimiser never saw."""
    import torch
    from learning_to_adapt_amd.dynamics import MetaMLPDynamicsModel, core
    from learning_to_adapt_amd.envs import SyntheticEnv
    env = SyntheticEnv("half_cheetah")
    od, ad = 20, 6
    rs = np.random.RandomState(0)
    A = 0.05 * rs.randn(od, od)
    B = 0.1 * rs.randn(ad, od)

    def paths(n, length=40):
        obs = rs.randn(n, length + 1, od)
        act = rs.uniform(-1, 1, (n, length, ad))
        for t in range(length):
            obs[:, t + 1] = obs[:, t] + obs[:, t] @ A + act[:, t] @ B
        return obs[:, :-1], act, obs[:, 1:]

    model = MetaMLPDynamicsModel(name="dyn", env=env, hidden_sizes=(32, 32), meta_batch_size=4, batch_size=5,
                                 learning_rate=3e-3, inner_learning_rate=0.01, valid_split_ratio=0.25, init_seed=0)
    np.random.seed(0)
    o, a, n = paths(8)
    s1 = model.fit(o, a, n, epochs=3)
    assert (s1["TrainPaths"], s1["ValidPaths"]) == (6, 2)
    assert model._dataset_train["obs"].shape == (6, 40, od) and model._dataset_test["delta"].shape == (2, 40, od)
    held_out = model._dataset_test["obs"].copy()

    def plain_loss(ds):
        x = torch.as_tensor(np.concatenate([ds["obs"], ds["act"]], axis=2), dtype=torch.float32)
        y = torch.as_tensor(ds["delta"], dtype=torch.float32)
        return float(torch.mean((y - core.mlp_forward(x, model._params, "relu", None)) ** 2))
    before = plain_loss(model._dataset_test)
    o2, a2, n2 = paths(4)
    s2 = model.fit(o2, a2, n2, epochs=40, compute_normalization=False)
    assert (s2["TrainPaths"], s2["ValidPaths"]) == (9, 3)                       # 6 + 3, 2 + 1: accumulated
    assert np.array_equal(model._dataset_test["obs"][:2], held_out)             # the earlier paths are still there
    assert s2["Epochs"] >= 1
    assert plain_loss(model._dataset_test) < before                             # the held-out loss went down
    # valid_split_ratio = 0: no held-out paths, validation falls back to the train set instead of failing
    m2 = MetaMLPDynamicsModel(name="dyn", env=env, hidden_sizes=(16,), meta_batch_size=2, batch_size=4, init_seed=0)
    s3 = m2.fit(o, a, n, epochs=2, valid_split_ratio=0.0)
    assert (s3["TrainPaths"], s3["ValidPaths"]) == (8, 0)


def test_launch_geometry_routing_table():
    """`l2a_plan_geometry` runs the launcher's own decision code without a GPU: which geometry every BASELINE config and the
    boundaries of the round-6 member fan take on a 256-CU device.  (Results never depend on the geometry - the GPU suite pins the
    bit-identity; this pins WHICH launch a plan gets.)"""
    from learning_to_adapt_amd import _lib
    hc = dict(obs_dim=20, act_dim=6, hidden=[512, 512])
    ant = dict(obs_dim=41, act_dim=8, hidden=[512, 512])
    g = lambda shape, E, mode, m, n, h, **kw: _lib.plan_geometry(shape["obs_dim"], shape["act_dim"], shape["hidden"], E, mode, m, n, h, **kw)  # noqa: E731
    # config 2: 125 tiles -> tile split with the shared middle set, two XCD placement units (group A | group B), three sets side by side in LDS
    c2 = g(hc, 5, "mean", 1, 2000, 30)
    assert (c2["kernel"], c2["nt"], c2["split"], c2["fan"], c2["placement_units"], c2["sets_per_batch"]) == ("mfma16", 1, 2, False, 2, 3)
    assert c2["workgroups"] == 256 and c2["lds_bytes"] <= 160 * 1024
    # config 3: 625 tiles.  Double rounds (default): 5 x 51 double tiles in front (candidates 0 .. 1631 of every env, two tiles per
    # workgroup on the whole-tiles-only instances), the other 5 x 368 candidates behind them on micro tiles (5 x 46 workgroups of
    # eight candidates; micro tiles off: 5 x 23 tiles shared by pairs); without double rounds: 512 whole + 113 shared by pairs in
    # one launch (tail split)
    c3 = g(ant, 5, "per_block", 5, 2000, 20)
    assert (c3["front_workgroups"], c3["kernel"], c3["workgroups"], c3["micro_tiles"]) == (255, "micro", 5 * 46, 2)
    c3 = g(ant, 5, "per_block", 5, 2000, 20, micro=0)
    assert (c3["front_workgroups"], c3["nt"], c3["split"], c3["split_from"], c3["workgroups"]) == (255, 1, 2, -1, 2 * 115)
    c3 = g(ant, 5, "per_block", 5, 2000, 20, double=0)
    assert (c3["front_workgroups"], c3["split"], c3["split_from"], c3["workgroups"]) == (0, 2, 512, 512 + 2 * 113)
    # config 4's plan on ONE GPU: 1000 tiles = 500 double tiles in one launch (the rest after one double round would be nearly two
    # rounds); without double rounds 1000 whole tiles, no tail worth splitting; config 5's iteration: 250 whole tiles (under two rounds)
    c4 = g(hc, 5, "mean", 1, 16000, 30)
    assert (c4["nt"], c4["split"], c4["workgroups"], c4["front_workgroups"]) == (2, 0, 500, 0) and c4["lds_bytes"] <= 160 * 1024
    c4 = g(hc, 5, "mean", 1, 16000, 30, double=0)
    assert (c4["nt"], c4["split"], c4["workgroups"]) == (1, 0, 1000)
    assert g(hc, 5, "mean", 1, 4000, 30)["workgroups"] == 250 and g(hc, 5, "mean", 1, 4000, 30)["nt"] == 1
    # run_mb_mpc.py's default (10 envs x 2000 candidates = 1250 tiles): two double rounds in front (10 x 51 double tiles), 230 whole
    # tiles behind; double rounds need width 512 and at least two rounds of tiles
    mb = g(hc, 1, "single", 10, 2000, 20)
    assert (mb["front_workgroups"], mb["nt"], mb["split"], mb["workgroups"], mb["whole_instance"]) == (510, 1, 0, 230, True)
    # whole single tiles of ONE set per candidate run on the whole-tiles-only instances too (a full round of a single model; not
    # ensembles: config 5's iteration keeps the general instance), off with the double rounds
    assert g(hc, 1, "single", 1, 4096, 10)["whole_instance"] and not g(hc, 1, "single", 1, 4096, 10, double=0)["whole_instance"]
    assert not g(hc, 5, "mean", 1, 4000, 30)["whole_instance"] and g(hc, 5, "mean", 1, 16000, 30)["whole_instance"]
    assert g(dict(hc, hidden=[256, 256]), 1, "single", 10, 2000, 20)["front_workgroups"] == 0
    assert g(hc, 1, "single", 1, 8176, 5)["front_workgroups"] == 0 and g(hc, 1, "single", 1, 8192, 5)["nt"] == 2
    # config 5's shards: one rank of 8 -> member fan (32 tiles x 5 members, every member on its own XCDs); one rank of 4 -> the fan
    # with two tiles per workgroup; one rank of 2 -> config 2's tile split
    s8 = g(hc, 5, "mean", 1, 500, 30)
    assert (s8["split"], s8["fan"], s8["nt"], s8["placement_units"], s8["sets_per_batch"]) == (3, True, 1, 5, 1)
    s4 = g(hc, 5, "mean", 1, 1000, 30)
    assert (s4["split"], s4["fan"], s4["nt"]) == (3, True, 2) and s4["lds_bytes"] <= 160 * 1024
    assert g(hc, 5, "mean", 1, 2000, 30) == c2
    # the fan's boundaries: E x tiles <= CUs (51 tiles x 5 = 255), then E x double tiles <= CUs, then back to the tile split
    assert g(hc, 5, "mean", 1, 816, 3)["nt"] == 1 and g(hc, 5, "mean", 1, 816, 3)["workgroups"] == 255
    assert (g(hc, 5, "mean", 1, 817, 3)["fan"], g(hc, 5, "mean", 1, 817, 3)["nt"]) == (True, 2)
    assert g(hc, 5, "mean", 1, 1632, 3)["fan"] and not g(hc, 5, "mean", 1, 1633, 3)["fan"]
    assert g(hc, 5, "mean", 1, 1633, 3)["split"] == 2
    # ... and its conditions: mean ensembles of 3 .. 8 sets only, off with l2a_set_fan(0) or l2a_set_split(0), never for per-block sets
    assert not g(hc, 2, "mean", 1, 500, 30)["fan"] and g(hc, 2, "mean", 1, 500, 30)["split"] == 1
    assert g(hc, 3, "mean", 2, 300, 6)["fan"] and g(hc, 8, "mean", 1, 333, 3)["fan"] and not g(hc, 9, "mean", 1, 100, 3)["fan"]
    off = g(hc, 5, "mean", 1, 500, 30, fan=0)
    assert (off["fan"], off["split"], off["workgroups"]) == (False, 2, 64)
    assert g(hc, 5, "mean", 1, 500, 30, split=0)["split"] == 0
    assert not g(ant, 5, "per_block", 5, 100, 10, micro=0)["fan"]
    # width 256: no two-tile fan instance
    assert not g(dict(hc, hidden=[256, 256]), 5, "mean", 1, 1000, 30)["fan"]
    # the reference's own default plans take micro tiles (c1: 125 workgroups of one micro tile; run_grbal.py default: 5 x 42 of three)
    c1 = g(hc, 1, "single", 1, 500, 10)
    assert (c1["kernel"], c1["workgroups"], c1["micro_tiles"]) == ("micro", 125, 1)
    c3b = g(dict(ant, hidden=[512, 512, 512]), 5, "per_block", 5, 500, 10)
    assert (c3b["kernel"], c3b["workgroups"], c3b["micro_tiles"]) == ("micro", 210, 3)
    # shapes without a matrix-core instance run the generic kernel
    assert g(dict(hc, hidden=[200, 72]), 1, "single", 1, 80, 4)["kernel"] == "valu"
    # a smaller device: the same plan moves to two tiles per workgroup, then stops fitting the fan
    assert (g(hc, 5, "mean", 1, 500, 30, cus=128)["fan"], g(hc, 5, "mean", 1, 500, 30, cus=128)["nt"]) == (True, 2)
    assert not g(hc, 5, "mean", 1, 500, 30, cus=64)["fan"]
