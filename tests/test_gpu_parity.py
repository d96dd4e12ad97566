"""Parity of the HIP path with the reference planner (golden vectors) and the CPU oracle.

Everything here goes through the C ABI of ``libl2a_hip.so`` (via the package's drop-in
classes).  Tolerances (BASELINE.json north star): chosen candidate index bit-exact, returns
within 1e-4 relative (fp32 MLP; error measured as |got - want| / max(1, |want|)).
"""

import ctypes

import numpy as np
import pytest
import torch

import cases
import cem_ties
from learning_to_adapt_amd import _lib

pytestmark = pytest.mark.gpu

RTOL = 1e-4
# CEM: the error bar a rank swap inside the elite mask has to fit in (measured error of a config-5 rollout: 5e-7
# relative on most iterations, 6e-6 on one - profiles/r03_parity_report.txt; a swap of two candidates further apart than 2 x this
# bar is a ranking bug)
CEM_TIE_RTOL = 1e-5
CEM_TIE_RTOL_ANT = 2e-5      # 41-dimensional state, unclipped CEM samples: RS rollouts of the Ant cases measure up to 9e-6


def _cem_tie_rtol(case):
    return CEM_TIE_RTOL_ANT if case["env"].startswith("ant") else CEM_TIE_RTOL


def rel_err(got, want):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    return float(np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want))))


def _set_kernel(kind):
    _lib.Context.get(0).set_kernel(kind)


@pytest.fixture(autouse=True)
def _auto_kernel():
    _set_kernel("auto")
    yield
    _set_kernel("auto")


def _rs_actions(case, seed, env):
    from oracle.planner import sample_rs_actions
    np.random.seed(seed)
    return sample_rs_actions(env.action_space.low, env.action_space.high, case["n"], case["m"], case["h"])


def _plan_returns(native, case, env, obs0, actions, cand_offset=0, n=None):
    n = case["n"] if n is None else n
    dev = native.device
    rets = torch.full((case["m"], n), float("nan"), dtype=torch.float32, device=dev)
    best = torch.zeros((case["m"],), dtype=torch.int64, device=dev)
    native.plan_rs(torch.from_numpy(np.ascontiguousarray(obs0, dtype=np.float32)).to(dev),
                   torch.from_numpy(np.ascontiguousarray(actions, dtype=np.float32)).to(dev),
                   case["m"], n, case["h"], case.get("discount", 1.0), env.reward_spec,
                   cand_offset=cand_offset, returns_out=rets, best_key=best)
    torch.cuda.synchronize()
    return rets.cpu().numpy(), best.cpu().numpy()


# ------------------------------------------------------------------------------------------
# 1. the drop-in controller against the reference planner's golden vectors (all configs)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cid", cases.case_ids())
def test_controller_matches_reference_golden(cid):
    case, seed = cases.split_id(cid)
    gold = cases.load_golden(cid)
    ctrl = cases.product_controller(case)
    np.random.seed(seed)
    actions, info = ctrl.get_actions(gold["obs0"])
    assert info == {}
    assert np.random.uniform() == float(gold["rng_next"])            # same RNG consumption
    margin = gold["margin"] / np.maximum(1.0, np.abs(gold["returns"][np.arange(case["m"]), gold["best"]]))
    declared_tie = margin < RTOL                                       # none in the fixtures
    assert not declared_tie.any()
    if case["planner"] == "cem":
        # CEM feeds returns back through a rank-based elite mask (reference :101): two candidates
        # whose returns differ by less than the fp32 reassociation error may swap ranks and move a
        # sample in or out of the elite set, after which the iterations legitimately diverge.
        # Bit-exact equality is therefore only required when no such rank tie occurred; the
        # per-iteration parity is pinned by test_cem_iterations_teacher_forced below.
        trace = ctrl.last_plan["cem_trace"]
        k = max(int(case["n"] * 0.1), 1)
        for it, tr in enumerate(trace):
            # until a mask differs the iterations run on the reference's own mean / std (the refit is float64 NumPy on
            # identical elites), so this iteration's returns are comparable with the reference's
            tie_rtol = _cem_tie_rtol(case)
            assert rel_err(tr["returns"], gold["cem_returns"][it]) < tie_rtol
            swaps, mask_flips, worst = cem_ties.assert_flips_are_ties(tr["returns"], gold["cem_returns"][it], k, tie_rtol)
            if mask_flips:
                # PROVEN rank tie: every flipped position has a witness pair of reference returns within the error bar.
                # From here on the two runs legitimately differ; visible in the -q summary as `x`, never a silent skip.  Every
                # iteration of this very case is checked on its own by test_cem_iterations_teacher_forced, and what the product
                # did AFTER the tie - the later iterations, the index and the action it finally returned - is checked here against
                # the oracle continued from the product's own statistics (a failure in there is a FAIL, not an `x`)
                from oracle import make_reward
                from oracle.planner import rollout_returns
                dyn, reward_fn = cases.oracle_dynamics(case), make_reward(case["env"], ctrl.env.dt)
                again = cem_ties.verify_tail_from_product(
                    lambda seq: rollout_returns(dyn, reward_fn, np.asarray(gold["obs0"], dtype=np.float64), seq, case["n"],
                                                case.get("discount", 1.0)),
                    ctrl.env.action_space.low, ctrl.env.action_space.high, case["n"], case["m"], case["h"], ctrl.alpha, k, seed,
                    trace, it, ctrl.last_plan["best_index"], actions, tie_rtol)
                pytest.xfail("rank tie inside the CEM elite mask at iteration %d: %d mask flips, largest witness gap %.1e "
                             "relative (error bar %.0e); the %d later iterations, the chosen index and action verified against "
                             "the oracle continued from the product's statistics (%d of them tied again)"
                             % (it, mask_flips, worst, 2 * tie_rtol, len(trace) - it - 1, again))
    assert np.array_equal(ctrl.last_plan["best_index"], gold["best"])  # bit-exact index
    np.testing.assert_array_equal(actions, gold["chosen"])             # float64 action, bit for bit
    want_best = gold["returns"][np.arange(case["m"]), gold["best"]]
    assert rel_err(ctrl.last_plan["best_return"], want_best) < RTOL


@pytest.mark.parametrize("cid", cases.case_ids(planner="cem"))
def test_cem_iterations_teacher_forced(cid):
    """Every CEM iteration on its own: start it from the REFERENCE's mean/std of the previous
    iteration (golden vectors) with the same normal draws; returns must match to 1e-4 and the
    refitted mean/std may differ only through declared rank ties in the elite mask."""
    case, seed = cases.split_id(cid)
    gold = cases.load_golden(cid)
    ctrl = cases.product_controller(case)
    env = ctrl.env
    m, n, h = case["m"], case["n"], case["h"]
    act_dim = env.action_space.shape[0]
    k = max(int(n * 0.1), 1)
    clip_low = np.concatenate([env.action_space.low] * h)
    clip_high = np.concatenate([env.action_space.high] * h)
    np.random.seed(seed)
    mean, std = np.zeros((m, h * act_dim)), np.ones((m, h * act_dim))
    flips_total = swaps_total = 0
    for it in range(case["num_cem_iters"]):
        state = np.random.get_state()
        z = np.random.normal(size=(n, m, h * act_dim))             # the iteration's draw (:85), for the refit check below
        np.random.set_state(state)
        new_mean, new_std, returns, _ = ctrl._cem_iteration(gold["obs0"], mean, std, k, clip_low, clip_high,
                                                            0, n, 1)
        assert rel_err(returns, gold["cem_returns"][it]) < _cem_tie_rtol(case)
        # every rank swap - and with it every elite-mask flip - must be a proven tie (a witness pair of REFERENCE
        # returns within twice the error bar); a flip without a witness is a ranking bug and fails here
        swaps, flips, _ = cem_ties.assert_flips_are_ties(returns, gold["cem_returns"][it], k, _cem_tie_rtol(case))
        swaps_total += swaps
        flips_total += flips
        if flips == 0:
            np.testing.assert_allclose(np.broadcast_to(new_mean, gold["cem_mean"][it].shape),
                                       gold["cem_mean"][it], rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(new_std, gold["cem_std"][it], rtol=1e-9, atol=1e-12)
        else:
            # the elites differ by tied candidates: the refit must then be the reference's arithmetic (:101-104) on
            # the product's own returns
            clipped = np.clip(mean + z * std, clip_low, clip_high)
            want_mean, want_std = cem_ties.reference_refit(mean, clipped, returns, k, ctrl.alpha)
            np.testing.assert_allclose(np.broadcast_to(new_mean, want_mean.shape), want_mean, rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(new_std, want_std, rtol=1e-9, atol=1e-12)
        mean, std = gold["cem_mean"][it], gold["cem_std"][it]      # teacher forcing
    assert np.random.uniform() == float(gold["rng_next"])
    print("rank swaps / elite-mask flips over all iterations: %d / %d" % (swaps_total, flips_total))


# ------------------------------------------------------------------------------------------
# 2. every candidate's return, both kernels
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kernel", ["auto", "valu"])
@pytest.mark.parametrize("cid", cases.case_ids(planner="rs"))
def test_returns_table_matches_oracle(cid, kernel):
    case, seed = cases.split_id(cid)
    if kernel == "valu" and case["n"] * case["h"] * case["m"] > 2000 * 30:
        pytest.skip("VALU kernel: parity is covered by the smaller cases")
    gold = cases.load_golden(cid)
    env, model = cases.product_model(case)
    _set_kernel(kernel)
    native = model.planner_model()
    a = _rs_actions(case, seed, env)
    rets, keys = _plan_returns(native, case, env, gold["obs0"], a)
    assert not np.isnan(rets).any()
    assert rel_err(rets, gold["returns"]) < RTOL
    for i in range(case["m"]):
        ret, idx = _lib.key_decode(keys[i])
        assert idx == int(gold["best"][i]) == int(np.argmax(rets[i]))
        assert np.float32(ret) == rets[i, idx]


def test_mfma_and_valu_kernels_agree():
    case = cases.CASES["c2_hc_rs_n2000_h30_e5"]
    env, model = cases.product_model(case)
    a = _rs_actions(case, 0, env)
    obs0 = cases.load_golden("c2_hc_rs_n2000_h30_e5_s0")["obs0"]
    _set_kernel("mfma")
    r1, k1 = _plan_returns(model.planner_model(), case, env, obs0, a)
    _set_kernel("valu")
    r2, k2 = _plan_returns(model.planner_model(), case, env, obs0, a)
    assert rel_err(r1, r2) < 2e-5
    assert np.array_equal(k1 & 0x7FFFFFFF, k2 & 0x7FFFFFFF)          # same winner


SPLIT_CASES = [
    # name, overrides - every flavour of the tile split: odd ensemble (shared middle set), even
    # ensemble (whole sets), single model (the only set is shared), 3 hidden layers (inner layers
    # run in full on both workgroups), 1 hidden layer (whole sets only), two envs.
    ("c2_hc_rs_n2000_h30_e5", {}),
    ("c1_hc_rs_n500_h10_e1", {}),
    ("hc_rs_m2_n100_h7_e2", {}),
    ("hc_rs_m3_n64_h5", {}),
    ("c2_hc_rs_n2000_h30_e5", dict(hidden=[256, 256, 256], E=3, n=300, h=6)),
    ("c2_hc_rs_n2000_h30_e5", dict(hidden=[128], E=3, n=200, h=5)),
    ("c2_hc_rs_n2000_h30_e5", dict(hidden=[128], E=1, n=200, h=5)),
    ("ant_rs_n300_h6_e3", {}),
    ("c2_hc_rs_n2000_h30_e5", dict(hidden=[128, 128], E=1, n=150, h=4)),
    ("c2_hc_rs_n2000_h30_e5", dict(hidden=[128, 128], E=2, n=150, h=4, activation="tanh")),
    ("c2_hc_rs_n2000_h30_e5", dict(hidden=[256, 256, 256, 256], E=1, n=90, h=3)),
    # tail split: multi-round plans whose last round fills under half of the chip (l2a_api.hip)
    ("c3_ant_rs_n2000_h20_pb5", dict(h=3)),                 # 625 tiles = 512 whole + 113 shared (per-block sets)
    ("c2_hc_rs_n2000_h30_e5", dict(n=4800, h=2)),           # 300 tiles = 256 + 44, E = 5 (groups + shared middle set)
    ("hc_rs_m2_n100_h7_e2", dict(n=2600, h=2)),             # m = 2, 326 tiles = 256 + 70, E = 2 (whole-set split)
]


@pytest.mark.parametrize("name,over", SPLIT_CASES)
def test_tile_split_is_bit_identical_to_single_workgroup(name, over):
    """Two workgroups per candidate tile (policy 1: shared middle set; policy 2: whole sets) must
    reproduce the one-workgroup launch (policy 0) bit for bit, deterministically, and agree with
    the oracle."""
    case = dict(cases.CASES[name], **over)
    env, model = cases.product_model(case)
    native = model.planner_model()
    a = _rs_actions(case, 1, env)
    obs0 = np.random.RandomState(11).randn(case["m"], env.observation_space.shape[0])
    ctx = _lib.Context.get(0)
    out = {}
    try:
        ctx.set_fan(0)          # (small ensembles would take the member fan under policies 1 / 2: its own test below)
        for policy in (1, 2, 0):
            ctx.set_split(policy)
            r1, k1 = _plan_returns(native, case, env, obs0, a)
            r2, k2 = _plan_returns(native, case, env, obs0, a)
            ctx.launch_status()
            assert np.array_equal(r1, r2) and np.array_equal(k1, k2)
            out[policy] = (r1, k1)
    finally:
        ctx.set_split(1)
        ctx.set_fan(1)
    for policy in (1, 2):
        assert np.array_equal(out[policy][0], out[0][0]), "policy %d differs from the unsplit launch" % policy
        assert np.array_equal(out[policy][1], out[0][1])
    from oracle import make_reward
    from oracle.planner import rollout_returns
    want = rollout_returns(cases.oracle_dynamics(case), make_reward(case["env"], env.dt), obs0, a, case["n"],
                           case.get("discount", 1.0)).reshape(case["m"], case["n"])
    assert rel_err(out[1][0], want) < RTOL


FAN_CASES = [
    # name, overrides - the member fan (one workgroup per candidate tile and ensemble member, csrc/l2a_mfma.h): one rank's
    # shard of BASELINE config 5, odd / even ensembles of 3 .. 8 sets, every hidden width, one to four hidden layers, generic
    # activations, the Ant's 41 observations (no O4 tile, K0L = 1), a distance reward, several envs with ragged tiles, a
    # discount, and as many workgroups as the chip has CUs (51 tiles x 5)
    ("c5_hc_cem_n4000_h30_e5", dict(n=500, h=6)),
    ("c2_hc_rs_n2000_h30_e5", dict(n=816, h=3)),
    ("c2_hc_rs_n2000_h30_e5", dict(hidden=[256, 256, 256], E=3, n=300, h=6)),
    ("c2_hc_rs_n2000_h30_e5", dict(hidden=[128], E=3, n=200, h=5)),
    ("c2_hc_rs_n2000_h30_e5", dict(hidden=[512], E=4, n=100, h=4)),
    ("c2_hc_rs_n2000_h30_e5", dict(hidden=[128, 128], E=4, n=150, h=4, activation="tanh")),
    ("c2_hc_rs_n2000_h30_e5", dict(hidden=[512, 512, 512, 512], E=6, n=90, h=3)),
    ("c2_hc_rs_n2000_h30_e5", dict(hidden=[256, 256], E=8, n=333, h=3)),
    ("c2_hc_rs_n2000_h30_e5", dict(hidden=[256, 256], E=7, n=37, h=9, discount=0.9)),
    ("ant_rs_n300_h6_e3", {}),
    ("ant_rs_n300_h6_e3", dict(E=5, n=500, h=3)),
    ("arm_cem_n160_h5_e3", {}),
    ("hc_rs_m3_n64_h5", dict(E=3, mode="mean", n=70)),
    # two candidate tiles per workgroup (width 512): one of FOUR ranks' shard of config 5, the largest plan that fits (51 double
    # tiles x 5), two envs with a ragged last double tile, generic activations on the Ant's shapes
    ("c5_hc_cem_n4000_h30_e5", dict(n=1000, h=5)),
    ("c2_hc_rs_n2000_h30_e5", dict(n=1632, h=2)),
    ("hc_rs_m3_n64_h5", dict(E=3, mode="mean", m=2, n=909, h=3)),
    ("ant_rs_n300_h6_e3", dict(n=1100, m=1, h=3, activation="tanh")),
]


@pytest.mark.parametrize("name,over", FAN_CASES)
def test_member_fan_is_bit_identical(name, over):
    """E workgroups per candidate tile, one weight set each, the members' terms swapped once per horizon step and added in
    the unsplit launch's order: every return and the arg-max keys bit for bit against the tile split and the unsplit launch,
    deterministically, with a candidate offset, with and without a returns table; then against the oracle."""
    case = dict(cases.CASES[name], **over)
    env, model = cases.product_model(case)
    native = model.planner_model()
    a = _rs_actions(case, 4, env)
    obs0 = np.random.RandomState(14).randn(case["m"], env.observation_space.shape[0])
    ctx = _lib.Context.get(0)
    dev = native.device
    out = {}
    try:
        ctx.set_micro(0)
        for fan, split in ((1, 1), (0, 1), (0, 0), (1, 2)):
            ctx.set_fan(fan)
            ctx.set_split(split)
            r, k = _plan_returns(native, case, env, obs0, a, cand_offset=7)
            r2, k2 = _plan_returns(native, case, env, obs0, a, cand_offset=7)
            assert np.array_equal(r, r2) and np.array_equal(k, k2)
            keys_only = torch.zeros((case["m"],), dtype=torch.int64, device=dev)
            native.plan_rs(torch.from_numpy(np.ascontiguousarray(obs0, dtype=np.float32)).to(dev),
                           torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev),
                           case["m"], case["n"], case["h"], case.get("discount", 1.0), env.reward_spec,
                           cand_offset=7, best_key=keys_only)
            torch.cuda.synchronize()
            ctx.launch_status()
            out[(fan, split)] = (r, k, keys_only.cpu().numpy())
    finally:
        ctx.set_micro(1)
        ctx.set_split(1)
        ctx.set_fan(1)
    ref = out[(0, 0)]
    assert np.array_equal(ref[1], ref[2])
    assert np.isfinite(ref[0]).all()
    for key in ((1, 1), (0, 1), (1, 2)):
        for x, y in zip(ref, out[key]):
            assert np.array_equal(x, y), key
    from oracle import make_reward
    from oracle.planner import rollout_returns
    want = rollout_returns(cases.oracle_dynamics(case), make_reward(case["env"], env.dt), obs0, a, case["n"],
                           case.get("discount", 1.0)).reshape(case["m"], case["n"])
    assert rel_err(out[(1, 1)][0], want) < RTOL


DOUBLE_CASES = [
    # name, overrides - plans of at least two rounds of 16-candidate tiles at width 512 (csrc/l2a_api.hip "Double rounds"):
    # BASELINE config 3 (5 x 51 double tiles in front, 5 x 23 tiles shared by pairs behind), config 4's shape on one GPU (all on
    # double tiles), run_mb_mpc.py's default (two double rounds + a whole round), exactly two rounds (no rest), a ragged last
    # tile in the rest and in the front launch, three hidden layers, generic activations, the distance reward, a discount
    ("c3_ant_rs_n2000_h20_pb5", dict(h=3)),
    ("c4_hc_rs_n16000_h30_e5", dict(h=2)),
    ("c2_hc_rs_n2000_h30_e5", dict(E=1, mode="single", m=10, n=2000, h=3)),
    ("c1_hc_rs_n500_h10_e1", dict(n=8192, h=2)),
    ("c1_hc_rs_n500_h10_e1", dict(n=8999, h=2, discount=0.9)),
    ("c1_hc_rs_n500_h10_e1", dict(n=16370, h=2)),
    ("c3b_ant_rs_n500_h10_pb5_3x512", dict(n=2050, h=2)),
    ("ant_rs_n300_h6_e3", dict(n=9000, m=1, h=2, activation="tanh")),
    ("arm_rs_n256_h8", dict(n=4100, m=2, h=2)),
    # whole SINGLE tiles of one set per candidate on the whole-tiles-only instances: a full round of a single model, per-block
    # sets whose plan is too large for micro tiles
    ("c1_hc_rs_n500_h10_e1", dict(n=4096, h=3)),
    ("c3_ant_rs_n2000_h20_pb5", dict(n=800, h=3)),
]


@pytest.mark.parametrize("name,over", DOUBLE_CASES)
def test_double_rounds_are_bit_identical(name, over):
    """A multi-round plan cut in two launches - double tiles on the whole-tiles-only instances in front, the rest of every env's
    candidates behind - against the one-launch geometries (whole rounds + tail split; no split at all): every return and the
    arg-max keys bit for bit, with a candidate offset, with and without a returns table, through the blocking mailbox entry, the
    rest on micro tiles (where they apply) and on 16-candidate tiles; then against the oracle."""
    case = dict(cases.CASES[name], **over)
    env, model = cases.product_model(case)
    native = model.planner_model()
    a = _rs_actions(case, 5, env)
    obs0 = np.random.RandomState(15).randn(case["m"], env.observation_space.shape[0])
    ctx = _lib.Context.get(0)
    dev = native.device
    geo = _lib.plan_geometry(env.observation_space.shape[0], env.action_space.shape[0], case["hidden"], case.get("E", 1),
                             case.get("mode", "single"), case["m"], case["n"], case["h"])
    assert geo["front_workgroups"] > 0 or geo["nt"] == 2 or geo["whole_instance"], geo     # the case does take the whole-tiles-only instances
    a_dev = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    out = {}
    try:
        for dbl, split, micro in ((1, 1, 1), (1, 1, 0), (0, 1, 1), (0, 0, 1)):   # (the rest behind a double round: micro tiles where they apply | 16-candidate tiles)
            ctx.set_double_rounds(dbl)
            ctx.set_split(split)
            ctx.set_micro(micro)
            r, k = _plan_returns(native, case, env, obs0, a, cand_offset=11)
            keys_only = torch.zeros((case["m"],), dtype=torch.int64, device=dev)
            native.plan_rs(torch.from_numpy(np.ascontiguousarray(obs0, dtype=np.float32)).to(dev), a_dev,
                           case["m"], case["n"], case["h"], case.get("discount", 1.0), env.reward_spec,
                           cand_offset=11, best_key=keys_only)
            torch.cuda.synchronize()
            mailed = []
            for _ in range(3):      # both ring slots of the mailbox, the tile counter re-armed by the last tile of the SECOND launch
                got = native.plan_rs_sync(obs0, a_dev, case["m"], case["n"], case["h"], case.get("discount", 1.0), env.reward_spec,
                                          cand_offset=11)
                assert got is not None
                mailed.append(got.view(np.int64).copy())
            ctx.launch_status()
            assert all(np.array_equal(x, mailed[0]) for x in mailed)
            out[(dbl, split, micro)] = (r, k, keys_only.cpu().numpy(), mailed[0])
    finally:
        ctx.set_split(1)
        ctx.set_double_rounds(1)
        ctx.set_micro(1)
    ref = out[(0, 0, 1)]
    assert np.array_equal(ref[1], ref[2]) and np.array_equal(ref[1], ref[3])
    assert np.isfinite(ref[0]).all()
    for key in ((1, 1, 1), (1, 1, 0), (0, 1, 1)):
        for x, y in zip(ref, out[key]):
            assert np.array_equal(x, y), key
    from oracle import make_reward
    from oracle.planner import rollout_returns
    want = rollout_returns(cases.oracle_dynamics(case), make_reward(case["env"], env.dt), obs0, a, case["n"],
                           case.get("discount", 1.0)).reshape(case["m"], case["n"])
    assert rel_err(out[(1, 1, 1)][0], want) < RTOL


MICRO_CASES = [
    # name, overrides - the micro-tile kernel (csrc/l2a_micro.h) on every code path it has: per-block sets on the reference's
    # default GrBAL plan, a single model, mean ensembles odd / even (groups A | B), the O4 quarter sums (HalfCheetah, two or
    # more hidden layers) and their absence (one hidden layer, the Ant's 41 observations), hidden width 256 (one 64-unit
    # tile per wave, the wave's two chunks one after the other), generic activations, a distance reward, ragged micro tiles,
    # more envs than a workgroup per CU leaves three micro tiles for (falls back to the 16-candidate kernel), a discount
    ("c3b_ant_rs_n500_h10_pb5_3x512", dict(h=4)),
    ("c1_hc_rs_n500_h10_e1", {}),
    ("c2_hc_rs_n2000_h30_e5", dict(n=700, h=5)),
    ("c2_hc_rs_n2000_h30_e5", dict(n=300, h=4, E=2)),
    ("c2_hc_rs_n2000_h30_e5", dict(n=333, h=3, E=3, hidden=[256, 256])),
    ("c2_hc_rs_n2000_h30_e5", dict(n=90, h=3, E=4, hidden=[512, 512, 512, 512])),
    ("c2_hc_rs_n2000_h30_e5", dict(n=200, h=5, E=3, hidden=[512])),
    ("c2_hc_rs_n2000_h30_e5", dict(n=150, h=4, E=2, hidden=[256, 256], activation="tanh")),
    ("hc_rs_ragged_n37_h3", {}),
    ("hc_rs_n1_h1", {}),
    ("hc_rs_m3_n64_h5", {}),
    ("hc_rs_discount", {}),
    ("hc_rs_tanh_256", {}),
    ("arm_rs_n256_h8", {}),
    ("ant_rs_n300_h6_e3", {}),
    ("ant_rs_4x256_e2", {}),
    ("arm_cem_n160_h5_e3", {}),
    ("hc_rs_m3_n64_h5", dict(m=64, n=12, h=2)),
    ("c3_ant_rs_n2000_h20_pb5", dict(n=600, h=3)),
]


@pytest.mark.parametrize("name,over", MICRO_CASES)
def test_micro_tiles_are_bit_identical(name, over):
    """The micro-tile kernel (candidate tiles of FOUR on the 4x4x1 MFMA, workgroups of 4 / 8 / 12 candidates, no exchange
    between workgroups) against the 16-candidate kernel, split and unsplit: every return and the arg-max keys bit for bit,
    with a candidate offset, through the plain launch with and without a returns table; then against the oracle."""
    case = dict(cases.CASES[name], **over)
    if case["mode"] == "per_block":
        case["E"] = max(case["E"], case["m"])
    env, model = cases.product_model(case)
    native = model.planner_model()
    a = _rs_actions(case, 3, env)
    obs0 = np.random.RandomState(13).randn(case["m"], env.observation_space.shape[0])
    ctx = _lib.Context.get(0)
    dev = native.device
    out = {}
    try:
        for micro, split in ((0, 0), (0, 1), (2, 1)):
            ctx.set_micro(micro)
            ctx.set_split(split)
            r, k = _plan_returns(native, case, env, obs0, a, cand_offset=11)
            keys_only = torch.zeros((case["m"],), dtype=torch.int64, device=dev)
            native.plan_rs(torch.from_numpy(np.ascontiguousarray(obs0, dtype=np.float32)).to(dev),
                           torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev),
                           case["m"], case["n"], case["h"], case.get("discount", 1.0), env.reward_spec,
                           cand_offset=11, best_key=keys_only)
            torch.cuda.synchronize()
            ctx.launch_status()
            out[(micro, split)] = (r, k, keys_only.cpu().numpy())
    finally:
        ctx.set_micro(1)
        ctx.set_split(1)
    ref = out[(0, 0)]
    assert np.array_equal(ref[1], ref[2])
    assert np.isfinite(ref[0]).all()
    for key in ((0, 1), (2, 1)):
        for x, y in zip(ref, out[key]):
            assert np.array_equal(x, y), key
    from oracle import make_reward
    from oracle.planner import rollout_returns
    want = rollout_returns(cases.oracle_dynamics(case), make_reward(case["env"], env.dt), obs0, a, case["n"],
                           case.get("discount", 1.0)).reshape(case["m"], case["n"])
    assert rel_err(out[(2, 1)][0], want) < RTOL


BATCH_CASES = [
    ("c2_hc_rs_n2000_h30_e5", {}),                                      # split: 2 full sets + the shared one per workgroup
    ("c2_hc_rs_n2000_h30_e5", dict(n=4000, h=4)),                       # unsplit, E = 5: batches straddle the A | B groups
    ("c2_hc_rs_n2000_h30_e5", dict(n=4800, h=2)),                       # tail split: whole tiles and pairs in one launch
    ("hc_rs_m2_n100_h7_e2", {}),                                        # E = 2, whole-set split (one set per workgroup)
    ("c2_hc_rs_n2000_h30_e5", dict(hidden=[128, 128], E=4, n=150, h=4, activation="tanh")),
    ("c2_hc_rs_n2000_h30_e5", dict(hidden=[256, 256], E=7, n=3000, h=3)),   # E = 7 unsplit: 4 + 3
    ("ant_rs_n300_h6_e3", {}),
    ("c4_hc_rs_n16000_h30_e5", dict(h=2)),                              # NT = 2: only one set fits
]


@pytest.mark.parametrize("name,over", BATCH_CASES)
def test_set_batching_is_bit_identical(name, over):
    """Layer 0 of several sets back to back, then their GEMMs, then their reduces (l2a_set_batch) is a reordering of
    independent work: every batch size must give the bits of the one-set-at-a-time launch, under every split policy."""
    case = dict(cases.CASES[name], **over)
    env, model = cases.product_model(case)
    native = model.planner_model()
    a = _rs_actions(case, 2, env)
    obs0 = np.random.RandomState(12).randn(case["m"], env.observation_space.shape[0])
    ctx = _lib.Context.get(0)
    out = {}
    try:
        ctx.set_fan(0)          # (the member fan runs one set per workgroup: nothing to batch)
        for policy in (1, 0):
            ctx.set_split(policy)
            for sets in (1, 0, 2, 3, 4):
                ctx.set_batch(sets)
                out[(policy, sets, 1)] = _plan_returns(native, case, env, obs0, a)
                ctx.launch_status()
            # placement of the split pairs (l2a_set_xcd_align: padded grid, spare workgroups return at once)
            ctx.set_xcd_align(0)
            out[(policy, 4, 0)] = _plan_returns(native, case, env, obs0, a)
            ctx.launch_status()
            ctx.set_xcd_align(1)
    finally:
        ctx.set_split(1)
        ctx.set_batch(0)
        ctx.set_xcd_align(1)
        ctx.set_fan(1)
    ref = out[(0, 1, 1)]
    for key, (r, k) in out.items():
        assert np.array_equal(r, ref[0]) and np.array_equal(k, ref[1]), "split %d, batch %d, xcd_align %d differs" % key


def test_split_survives_stale_tags_of_short_launches():
    """h = 1 and h = 2 launches leave low tags in the exchange buffer; later launches must not
    mistake them for fresh data."""
    case = cases.CASES["c2_hc_rs_n2000_h30_e5"]
    env, model = cases.product_model(case)
    native = model.planner_model()
    a = _rs_actions(case, 1, env)
    obs0 = cases.load_golden("c2_hc_rs_n2000_h30_e5_s1")["obs0"]
    ctx = _lib.Context.get(0)
    for hh in (1, 2, 3, 1, 30, 2):
        c = dict(case, h=hh)
        r1, _ = _plan_returns(native, c, env, obs0, a[:hh])
        r2, _ = _plan_returns(native, c, env, obs0, a[:hh])
        assert np.array_equal(r1, r2)
    ctx.launch_status()


def test_mfma_kernel_refuses_ineligible_shape():
    case = cases.CASES["hc_rs_odd_hidden"]
    env, model = cases.product_model(case)
    _set_kernel("mfma")
    with pytest.raises(_lib.L2AError):
        _plan_returns(model.planner_model(), case, env, np.zeros((1, 20)), np.zeros((case["h"], case["n"], 6)))


# ------------------------------------------------------------------------------------------
# 3. predict (the reference's per-step API) against the oracle
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["c1_hc_rs_n500_h10_e1", "c2_hc_rs_n2000_h30_e5", "c3_ant_rs_n2000_h20_pb5",
                                  "hc_rs_odd_hidden", "hc_rs_tanh_256", "arm_rs_n256_h8"])
def test_predict_matches_oracle(name):
    case = cases.CASES[name]
    env, model = cases.product_model(case)
    dyn = cases.oracle_dynamics(case)
    rs = np.random.RandomState(7)
    rows = 5 * 37 if case["mode"] == "per_block" else 333
    obs = rs.randn(rows, env.observation_space.shape[0])
    act = rs.uniform(env.action_space.low, env.action_space.high, size=(rows, env.action_space.shape[0]))
    got = model.predict(obs, act)
    want = dyn.predict(obs, act)
    assert got.shape == want.shape and got.dtype == np.float64
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5)


def test_predict_asserts_like_the_reference():
    case = cases.CASES["c1_hc_rs_n500_h10_e1"]
    env, model = cases.product_model(case)
    with pytest.raises(AssertionError):
        model.predict(np.zeros((4, 20)), np.zeros((5, 6)))
    with pytest.raises(AssertionError):
        model.predict(np.zeros((4, 19)), np.zeros((4, 6)))


# ------------------------------------------------------------------------------------------
# 4. size-independent properties at BASELINE.json's full sizes
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,n", [("c2_hc_rs_n2000_h30_e5", 2000), ("c4_hc_rs_n16000_h30_e5", 16000)])
def test_full_size_properties(name, n):
    case = dict(cases.CASES[name])
    env, model = cases.product_model(case)
    native = model.planner_model()
    obs0 = cases.load_golden(name + "_s0")["obs0"]
    a = _rs_actions(case, 0, env)
    rets, keys = _plan_returns(native, case, env, obs0, a)
    _, idx = _lib.key_decode(keys[0])
    assert idx == int(np.argmax(rets[0]))

    # (a) determinism: a second launch is bit-identical
    rets2, keys2 = _plan_returns(native, case, env, obs0, a)
    assert np.array_equal(rets, rets2) and np.array_equal(keys, keys2)

    # (b) candidates are independent: permuting them permutes the returns, bit for bit
    perm = np.random.RandomState(3).permutation(n)
    rets_p, keys_p = _plan_returns(native, case, env, obs0, a[:, perm, :])
    assert np.array_equal(rets_p[0], rets[0][perm])
    assert perm[_lib.key_decode(keys_p[0])[1]] == idx

    # (c) sharding (what N ranks do): shards with global offsets, max of the keys == full plan
    best = 0
    for s in range(8):
        lo, hi = (s * n) // 8, ((s + 1) * n) // 8
        sub = dict(case)
        r_s, k_s = _plan_returns(native, sub, env, obs0, a[:, lo:hi, :], cand_offset=lo, n=hi - lo)
        assert np.array_equal(r_s[0], rets[0][lo:hi])
        best = max(best, int(k_s[0]))
    assert best == int(keys[0])

    # (d) ragged prefix: dropping the tail (n not a multiple of the tile) leaves the rest unchanged
    cut = n - 7
    r_c, _ = _plan_returns(native, case, env, obs0, a[:, :cut, :], n=cut)
    assert np.array_equal(r_c[0], rets[0][:cut])


def test_discount_is_linear_in_the_rewards():
    """sum_t g^t r_t with g = 1 equals the h = k prefix sums' telescoping: returns(h) - returns(h-1)
    is the last reward; with discount g the same difference is scaled by g^(h-1)."""
    case = dict(cases.CASES["hc_rs_discount"])
    env, model = cases.product_model(case)
    native = model.planner_model()
    obs0 = cases.load_golden("hc_rs_discount_s0")["obs0"]
    a = _rs_actions(case, 0, env)
    h = case["h"]

    def run(hh, g):
        c = dict(case, h=hh, discount=g)
        return _plan_returns(native, c, env, obs0, a[:hh])[0][0].astype(np.float64)

    last_undiscounted = run(h, 1.0) - run(h - 1, 1.0)
    last_discounted = run(h, 0.9) - run(h - 1, 0.9)
    np.testing.assert_allclose(last_discounted, 0.9 ** (h - 1) * last_undiscounted, rtol=1e-3, atol=2e-4)


# ------------------------------------------------------------------------------------------
# 5. API behaviour
# ------------------------------------------------------------------------------------------
def test_unfused_path_agrees_with_fused_choice():
    case = cases.CASES["hc_rs_m3_n64_h5"]
    gold = cases.load_golden("hc_rs_m3_n64_h5_s0")
    env, model = cases.product_model(case)

    class NoSpecEnv(object):            # a custom env: reward() only, nothing to fuse
        def __init__(self, inner):
            self._inner = inner
            self.dt = inner.dt
            self.observation_space = inner.observation_space
            self.action_space = inner.action_space

        def reward(self, obs, act, nxt):
            return self._inner.reward(obs, act, nxt)

    ctrl = cases.product_controller(case, model=model, env=NoSpecEnv(env))
    assert not ctrl._fusable()
    np.random.seed(0)
    actions, _ = ctrl.get_actions(gold["obs0"])
    np.testing.assert_array_equal(actions, gold["chosen"])


def test_reward_model_path_agrees_with_fused_choice():
    """`use_reward_model=True` (reference `mpc_controller.py:122-124`): rewards come from
    `reward_model.predict(obs, act, next_obs)` on the host, dynamics from the GPU `predict`."""
    from learning_to_adapt_amd.policies import MPCController
    case = cases.CASES["hc_rs_m3_n64_h5"]
    gold = cases.load_golden("hc_rs_m3_n64_h5_s0")
    env, model = cases.product_model(case)

    class RewardModel(object):
        calls = 0

        def predict(self, obs, act, nxt):
            RewardModel.calls += 1
            return env.reward(obs, act, nxt)

    ctrl = MPCController(name="policy", env=env, dynamics_model=model, reward_model=RewardModel(),
                         use_reward_model=True, n_candidates=case["n"], horizon=case["h"])
    assert not ctrl._fusable()
    np.random.seed(0)
    actions, _ = ctrl.get_actions(gold["obs0"])
    assert RewardModel.calls == case["h"]
    np.testing.assert_array_equal(actions, gold["chosen"])


def test_device_rng_mode_plans_within_bounds():
    case = cases.CASES["c1_hc_rs_n500_h10_e1"]
    gold = cases.load_golden("c1_hc_rs_n500_h10_e1_s0")
    ctrl = cases.product_controller(case, rng="device")
    torch.manual_seed(0)
    state = np.random.get_state()[1].copy()
    actions, _ = ctrl.get_actions(gold["obs0"])
    assert np.array_equal(np.random.get_state()[1], state)       # host RNG untouched
    assert actions.shape == (1, 6) and np.all(np.abs(actions) <= 1.0)
    # random shooting with 500 candidates lands in the same return range as the reference's draw
    assert abs(float(ctrl.last_plan["best_return"][0]) - float(gold["returns"].max())) < 10.0


@pytest.mark.parametrize("cem_mode", ["fixed", "reference"])
@pytest.mark.parametrize("name", ["hc_cem_n400_h10", "hc_cem_m2_n100_h4"])
def test_device_cem_matches_host_loop_with_injected_normals(name, cem_mode):
    """Sampling / clip / elites / refit on the GPU vs the host loop of the same mode on the same normals
    ('reference' = the loop the golden vectors pin against the reference's get_cem_action)."""
    case = dict(cases.CASES[name])
    obs0 = cases.load_golden(name + "_s0")["obs0"]
    n, m, D = case["n"], case["m"], case["h"] * 6
    zs = [np.random.RandomState(100 + i).normal(size=(n, m, D)) for i in range(case["num_cem_iters"])]
    host = cases.product_controller(case, cem_mode=cem_mode, pipeline_chunks=1)   # whole-iteration draws are injected
    it = iter(zs)
    host._cem_draw = lambda n_, m_, D_: next(it).reshape(n_ * m_, D_)      # inject the iteration's normals
    a_host, _ = host.get_actions(obs0)
    dev = cases.product_controller(case, rng="device", cem_mode=cem_mode)
    it2 = iter(zs)
    dev._cem_normal_device = lambda shape, device: torch.from_numpy(next(it2).astype(np.float32)).to(device)
    a_dev, _ = dev.get_actions(obs0)
    tr = host.last_plan["cem_trace"][-1]
    np.testing.assert_allclose(dev.last_plan["cem_mean"], np.broadcast_to(tr["mean"], (m, D)), rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(dev.last_plan["cem_std"], np.broadcast_to(tr["std"], (m, D)), rtol=1e-3, atol=1e-3)
    assert np.array_equal(dev.last_plan["best_index"], host.last_plan["best_index"])
    np.testing.assert_allclose(a_dev, a_host, rtol=1e-5, atol=1e-6)
    # and the plain device-RNG run: in bounds, deterministic under torch.manual_seed
    torch.manual_seed(3)
    c1 = cases.product_controller(case, rng="device", cem_mode=cem_mode)
    r1, _ = c1.get_actions(obs0)
    torch.manual_seed(3)
    c2 = cases.product_controller(case, rng="device", cem_mode=cem_mode)
    r2, _ = c2.get_actions(obs0)
    assert np.array_equal(r1, r2)
    if cem_mode == "fixed":                      # the reference returns the UNCLIPPED first action (:92,106)
        assert np.all(np.abs(r1) <= 1.0)


def test_plan_before_weights_is_an_error_not_garbage():
    from learning_to_adapt_amd.dynamics.native_model import NativeModel
    from learning_to_adapt_amd.envs import RewardSpec
    nm = NativeModel(20, 6, (512, 512), "relu", None, 1, "single")
    dev = nm.device
    with pytest.raises(_lib.L2AError, match="never set"):
        nm.plan_rs(torch.zeros((1, 20), device=dev), torch.zeros((2, 16, 6), device=dev), 1, 16, 2, 1.0,
                   RewardSpec.half_cheetah(20, 0.01), best_key=torch.zeros(1, dtype=torch.int64, device=dev))
    nm.close()


def test_grbal_step_adapt_then_plan_matches_oracle():
    """The GrBAL controller step of samplers/sampler.py:81-91: switch_to_pre_adapt, adapt on the last
    16 transitions of each env (one SGD step per env, on the device), then plan with env i <-> adapted
    set i.  The oracle plans with the same adapted sets pulled back to the host."""
    from learning_to_adapt_amd.dynamics import MetaMLPDynamicsModel
    from learning_to_adapt_amd.envs import SyntheticEnv
    from learning_to_adapt_amd.policies import MPCController
    from learning_to_adapt_amd.utils import synthetic
    from oracle import OracleMLPDynamics, make_reward, rs_plan
    env = SyntheticEnv("ant")
    od, ad = 41, 8
    model = MetaMLPDynamicsModel(name="dyn", env=env, hidden_sizes=(512, 512, 512), inner_learning_rate=0.01,
                                 meta_batch_size=5, init_seed=0)
    model.set_params(synthetic.make_weight_set(od, ad, [512, 512, 512], 1000))
    norm = synthetic.make_norm(od, ad, env.action_space.low, env.action_space.high, 2000)
    model.set_normalization(norm)
    rs = np.random.RandomState(4)
    m = 5
    obs = [rs.randn(16, od) for _ in range(m)]
    act = [rs.uniform(-150, 150, (16, ad)) for _ in range(m)]
    nxt = [o + 0.2 * rs.randn(16, od) for o in obs]
    ctrl = MPCController(name="p", env=env, dynamics_model=model, n_candidates=300, horizon=6)
    obs0 = rs.randn(m, od)
    for _ in range(2):                       # twice: exercises switch_to_pre_adapt + re-upload
        model.switch_to_pre_adapt()
        model.adapt(obs, act, nxt)
        np.random.seed(9)
        got, _ = ctrl.get_actions(obs0)
    sets = [[q.detach().cpu().numpy() for q in ps] for ps in model._adapted_param_values]
    base = [q.numpy() for q in model._prev_params]
    assert max(float(np.abs(sets[i][0] - base[0]).max()) for i in range(m)) > 1e-5      # adapt did something
    dyn = OracleMLPDynamics(od, ad, sets, norm, mode="per_block")
    np.random.seed(9)
    want, best, returns, _ = rs_plan(dyn, make_reward("ant", env.dt), obs0, env.action_space.low,
                                     env.action_space.high, 300, 6)
    assert np.array_equal(ctrl.last_plan["best_index"], best)
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("hidden,act,rows", [((512, 512, 512), "relu", 16), ((256, 256), "tanh", 7),
                                             ((128,), "sigmoid", 16), ((200, 72), "relu", 3)])
def test_device_adapt_matches_autograd(hidden, act, rows):
    """`l2a_model_adapt_sgd` (two kernels writing the adapted sets in place) against the stock PyTorch
    autograd inner step on the same inputs: every adapted parameter, and the packed copies the planner
    reads (checked through a plan)."""
    from learning_to_adapt_amd.dynamics import MetaMLPDynamicsModel
    from learning_to_adapt_amd.envs import SyntheticEnv
    from learning_to_adapt_amd.policies import MPCController
    from learning_to_adapt_amd.utils import synthetic
    env = SyntheticEnv("ant")
    od, ad, m = 41, 8, 3
    norm = synthetic.make_norm(od, ad, env.action_space.low, env.action_space.high, 2000)
    rs = np.random.RandomState(rows)
    obs = [rs.randn(rows, od) for _ in range(m)]
    act_ = [rs.uniform(-150, 150, (rows, ad)) for _ in range(m)]
    nxt = [o + 0.3 * rs.randn(rows, od) for o in obs]
    models = []
    for native in (True, False):
        model = MetaMLPDynamicsModel(name="dyn", env=env, hidden_sizes=hidden, hidden_nonlinearity=act,
                                     inner_learning_rate=0.05, meta_batch_size=m, init_seed=0)
        model.set_params(synthetic.make_weight_set(od, ad, list(hidden), 1000))
        model.set_normalization(norm)
        model.use_native_adapt = native
        model.adapt(obs, act_, nxt)
        models.append(model)
    fused, stock = models
    assert type(fused._adapted_param_values).__name__ == "_ResidentSets" and isinstance(stock._adapted_param_values, list)
    base = [p.numpy() for p in fused._prev_params]
    for i in range(m):
        for b, got, want in zip(base, fused._adapted_param_values[i], stock._adapted_param_values[i]):
            got, want = got.cpu().numpy(), want.cpu().numpy()
            step = np.abs(want - b).max()
            assert np.abs(got - want).max() <= 2e-5 * max(step, 1e-3) + 1e-7, (i, got.shape)
    assert max(np.abs(q.cpu().numpy() - b).max() for q, b in zip(fused._adapted_param_values[0], base)) > 1e-5
    obs0 = rs.randn(m, od)
    picks = []
    for model in models:
        ctrl = MPCController(name="p", env=env, dynamics_model=model, n_candidates=120, horizon=4)
        np.random.seed(3)
        a, _ = ctrl.get_actions(obs0)
        picks.append((a, ctrl.last_plan["best_index"], ctrl.last_plan["best_return"]))
    assert np.array_equal(picks[0][1], picks[1][1]) and np.array_equal(picks[0][0], picks[1][0])
    np.testing.assert_allclose(picks[0][2], picks[1][2], rtol=1e-4, atol=1e-4)


def test_host_staged_and_raw_adapt_are_bit_identical_to_the_device_pointer_entry():
    """`l2a_model_adapt_sgd_host` (batches copied into host-mapped staging that the kernels read directly, two slots)
    against `l2a_model_adapt_sgd` on device tensors: every adapted parameter bit for bit - over repeated steps with
    fresh data (both staging slots, arrays overwritten right after the call), a changed learning rate, fewer tasks /
    rows."""
    from learning_to_adapt_amd.dynamics.native_model import NativeModel
    from learning_to_adapt_amd.utils import synthetic
    od, ad, hidden, m = 41, 8, (512, 512, 512), 5
    dev = torch.device("cuda:0")
    base = [torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32)).to(dev)
            for w in synthetic.make_weight_set(od, ad, list(hidden), 1000)]
    plain = NativeModel(od, ad, hidden, "relu", None, m, "per_block")
    staged = NativeModel(od, ad, hidden, "relu", None, m, "per_block")
    raw = NativeModel(od, ad, hidden, "relu", None, m, "per_block")
    rs = np.random.RandomState(0)
    norm = {"obs": (10.0 * rs.randn(od), 0.01 + rs.rand(od)), "act": (rs.randn(ad), 50.0 + 100.0 * rs.rand(ad)),
            "delta": (0.1 * rs.randn(od), 1e-3 + 0.1 * rs.rand(od))}
    for step, (mm, rows, lr) in enumerate([(5, 16, 0.01), (5, 16, 0.01), (5, 16, 0.01), (5, 16, 0.02), (3, 9, 0.02),
                                           (1, 1, 0.5), (5, 16, 0.01)]):
        # un-normalised transitions; the host normalisation of MetaMLPDynamicsModel.adapt (float64, then the cast)
        ob = norm["obs"][0] + norm["obs"][1] * rs.randn(mm, rows, od)
        ac = rs.uniform(-150, 150, (mm, rows, ad))
        nx = ob + norm["delta"][0] + norm["delta"][1] * rs.randn(mm, rows, od)
        nz = lambda v, k: (v - norm[k][0]) / (norm[k][1] + 1e-10)  # noqa: E731
        x = np.concatenate([nz(ob, "obs"), nz(ac, "act")], axis=2).astype(np.float32)
        y = nz(nx - ob, "delta").astype(np.float32)
        plain.adapt_sgd(base, torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev), lr)
        staged.adapt_sgd_host(base, x, y, lr)
        raw.adapt_sgd_raw(base, ob, ac, nx, norm, lr)       # normalised on the device, same arithmetic
        x[:] = 0.0      # the arrays may be reused as soon as the call returns
        y[:] = 0.0
        ob[:] = 0.0
        for e in range(mm):
            for a, b, c in zip(plain.get_weights(e), staged.get_weights(e), raw.get_weights(e)):
                assert torch.equal(a, b) and torch.equal(a, c), (step, e, tuple(a.shape))
        assert not torch.equal(plain.get_weights(0)[0], base[0])
    plain.close()
    staged.close()
    raw.close()


@pytest.mark.parametrize("kernel", ["auto", "valu"])
@pytest.mark.parametrize("name,over", [("c2_hc_rs_n2000_h30_e5", dict(h=9)), ("hc_rs_m3_n64_h5", dict(h=7)),
                                        ("c3_ant_rs_n2000_h20_pb5", dict(n=300, h=6)),
                                        ("hc_rs_discount", dict(h=10)), ("c2_hc_rs_n2000_h30_e5", dict(n=4800, h=4))])
def test_horizon_chunks_are_bit_identical_to_one_launch(name, over, kernel):
    """`l2a_plan_rs_chunk` chains (state + returns handed from launch to launch) against one `l2a_plan_rs`:
    returns and arg-max key bit for bit, for uneven chunkings, both kernels, every split flavour."""
    case = dict(cases.CASES[name], **over)
    if kernel == "valu" and case["n"] * case["m"] * case["h"] > 20000:
        pytest.skip("VALU kernel: small cases only")
    env, model = cases.product_model(case)
    native = model.planner_model()
    dev = native.device
    m, n, h = case["m"], case["n"], case["h"]
    a = _rs_actions(case, 2, env)
    obs0 = np.random.RandomState(12).randn(m, env.observation_space.shape[0])
    ctx = _lib.Context.get(0)
    ctx.set_kernel(kernel)
    try:
        want_r, want_k = _plan_returns(native, case, env, obs0, a)
        a_dev = torch.from_numpy(a.astype(np.float32)).to(dev)
        obs_dev = torch.from_numpy(obs0.astype(np.float32)).to(dev)
        disc = case.get("discount", 1.0)
        for bounds in ([0, h // 3, h], [0, 1, 2, h], [0, h - 1, h]):
            rets = [torch.empty((m, n), dtype=torch.float32, device=dev) for _ in (0, 1)]
            state = [torch.empty((m * n, native.obs_dim), dtype=torch.float32, device=dev) for _ in (0, 1)]
            best = torch.zeros((m,), dtype=torch.int64, device=dev)
            K = len(bounds) - 1
            for c in range(K):
                t0, t1 = bounds[c], bounds[c + 1]
                last = c == K - 1
                native.plan_rs_chunk(obs_dev if c == 0 else state[(c + 1) % 2], c > 0, a_dev[t0:t1].contiguous(), m, n,
                                     t1 - t0, t0, disc, env.reward_spec,
                                     returns_in=rets[(c + 1) % 2] if c > 0 else None, returns_out=rets[c % 2],
                                     state_out=None if last else state[c % 2], best_key=best if last else None)
            got_r, got_k = rets[(K - 1) % 2].cpu().numpy(), best.cpu().numpy()
            ctx.launch_status()
            assert np.array_equal(got_r, want_r), bounds
            assert np.array_equal(got_k, want_k), bounds
    finally:
        ctx.set_kernel("auto")


@pytest.mark.parametrize("cid", ["c2_hc_rs_n2000_h30_e5_s0", "c3_ant_rs_n2000_h20_pb5_s0", "hc_rs_discount_s0"])
def test_pipelined_controller_equals_single_launch_controller(cid):
    """Parity mode with the horizon pipeline (draw chunk k + 1 while chunk k rolls out) vs one launch: same RNG
    consumption, same action, same return bits; and the golden vector."""
    case, seed = cases.split_id(cid)
    gold = cases.load_golden(cid)
    env, model = cases.product_model(case)
    out = []
    for chunks in (3, 1, 5):
        ctrl = cases.product_controller(case, model=model, env=env, pipeline_chunks=chunks)
        np.random.seed(seed)
        a, _ = ctrl.get_actions(gold["obs0"])
        assert np.random.uniform() == float(gold["rng_next"])
        out.append((a, ctrl.last_plan["best_index"].copy(), ctrl.last_plan["best_return"].copy()))
    for o in out[1:]:
        assert np.array_equal(o[0], out[0][0]) and np.array_equal(o[1], out[0][1]) and np.array_equal(o[2], out[0][2])
    np.testing.assert_array_equal(out[0][0], gold["chosen"])


def test_weight_upload_round_trips_and_strided_equals_per_set():
    """`l2a_model_get_weights` returns what `l2a_model_set_weights` stored; the strided batch upload of stacked
    sets plans exactly like per-set uploads (same packed copies); bad strides are rejected."""
    from learning_to_adapt_amd.dynamics.native_model import NativeModel
    case = cases.CASES["c3b_ant_rs_n500_h10_pb5_3x512"]
    env, sets, norms = cases.recipe(case)
    od, ad = env.observation_space.shape[0], env.action_space.shape[0]
    a = NativeModel(od, ad, case["hidden"], "relu", None, case["E"], "per_block")
    b = NativeModel(od, ad, case["hidden"], "relu", None, case["E"], "per_block")
    for e in range(case["E"]):
        a.set_weights(e, sets[e])
        a.set_norm(e, norms[e])
        b.set_norm(e, norms[e])
    stacked = [torch.from_numpy(np.stack([sets[e][i] for e in range(case["E"])])).to(b.device) for i in range(len(sets[0]))]
    b.set_weights_stacked(0, stacked)
    for e in range(case["E"]):
        for got_a, got_b, want in zip(a.get_weights(e), b.get_weights(e), sets[e]):
            assert np.array_equal(got_a.cpu().numpy(), want) and np.array_equal(got_b.cpu().numpy(), want)
    rs = np.random.RandomState(0)
    m, n, h = case["m"], 100, 3
    obs0 = torch.from_numpy(rs.randn(m, od).astype(np.float32)).to(a.device)
    acts = torch.from_numpy(rs.uniform(-150, 150, (h, m * n, ad)).astype(np.float32)).to(a.device)
    outs = []
    for nat in (a, b):
        rets = torch.empty((m, n), dtype=torch.float32, device=nat.device)
        best = torch.zeros((m,), dtype=torch.int64, device=nat.device)
        nat.plan_rs(obs0, acts, m, n, h, 1.0, env.reward_spec, returns_out=rets, best_key=best)
        outs.append((rets.cpu().numpy(), best.cpu().numpy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    ptrs = (ctypes.c_void_p * len(stacked))(*[t.data_ptr() for t in stacked])
    bad = (ctypes.c_longlong * len(stacked))(*([1] * len(stacked)))
    rc = b.lib.l2a_model_set_weights_strided(b.handle, 0, case["E"], ptrs, bad, None)
    assert rc < 0 and b"stride" in b.lib.l2a_last_error(b.ctx.handle)
    rc = b.lib.l2a_model_set_weights_strided(b.handle, 3, case["E"], ptrs, bad, None)
    assert rc < 0 and b"range" in b.lib.l2a_last_error(b.ctx.handle)


def test_pipelined_plan_of_candidate_shards_combines_to_the_full_plan():
    """The N > 1 flavour of the pipelined parity mode on one GPU: every "rank" draws the full candidate stream
    chunk by chunk and rolls out only its shard (global indices through cand_offset); the MAX of the shard keys
    is the key of the unsharded single-launch plan, bit for bit."""
    case = cases.CASES["hc_rs_m3_n64_h5"]
    case = dict(case, h=8)
    env, model = cases.product_model(case)
    obs0 = np.random.RandomState(3).randn(case["m"], env.observation_space.shape[0])
    full = cases.product_controller(case, model=model, env=env, pipeline_chunks=1)
    np.random.seed(5)
    a_full, _ = full.get_actions(obs0)
    want_idx, want_ret = full.last_plan["best_index"], full.last_plan["best_return"]
    n, m, h = case["n"], case["m"], case["h"]
    keys = []
    for rank in range(3):
        ctrl = cases.product_controller(case, model=model, env=env, pipeline_chunks=4)
        lo, hi = ctrl._shard_range(n, rank, 3)
        np.random.seed(5)
        best, cand_a = ctrl._plan_pipelined(obs0, n, m, h, lo, hi, 3)
        keys.append(best.cpu().numpy().copy())
        assert cand_a.shape == (m, n, env.action_space.shape[0])
    key = np.max(np.stack(keys), axis=0)
    for i in range(m):
        ret, idx = _lib.key_decode(key[i])
        assert idx == int(want_idx[i]) and np.float32(ret) == np.float32(want_ret[i])
    np.testing.assert_array_equal(cand_a[np.arange(m), want_idx], a_full)


def test_invalid_plans_are_rejected():
    case = cases.CASES["c1_hc_rs_n500_h10_e1"]
    env, model = cases.product_model(case)
    native = model.planner_model()
    dev = native.device
    obs0 = torch.zeros((1, 20), device=dev)
    key = torch.zeros(1, dtype=torch.int64, device=dev)
    acts = torch.zeros((2, 16, 6), device=dev)
    from learning_to_adapt_amd.envs import RewardSpec
    ok = RewardSpec.half_cheetah(20, 0.01)
    for kwargs in (dict(m=1, n=0, h=2), dict(m=0, n=16, h=2), dict(m=1, n=16, h=0)):
        rc = native.lib.l2a_plan_rs(native.handle, ctypes.c_void_p(obs0.data_ptr()), ctypes.c_void_p(acts.data_ptr()),
                                    kwargs["m"], kwargs["n"], kwargs["h"], 1.0, ctypes.byref(ok), 0, None,
                                    ctypes.c_void_p(key.data_ptr()), None)
        assert rc == -1 and b"must be >= 1" in native.lib.l2a_last_error(native.ctx.handle)
    bad = RewardSpec.make(w_vel=1.0, dt=0.01, vel_index=25)
    rc = native.lib.l2a_plan_rs(native.handle, ctypes.c_void_p(obs0.data_ptr()), ctypes.c_void_p(acts.data_ptr()),
                                1, 16, 2, 1.0, ctypes.byref(bad), 0, None, ctypes.c_void_p(key.data_ptr()), None)
    assert rc == -1 and b"vel_index" in native.lib.l2a_last_error(native.ctx.handle)
    rc = native.lib.l2a_plan_rs(native.handle, None, ctypes.c_void_p(acts.data_ptr()), 1, 16, 2, 1.0,
                                ctypes.byref(ok), 0, None, ctypes.c_void_p(key.data_ptr()), None)
    assert rc == -1


def test_device_info_reports_gfx950():
    info = _lib.Context.get(0).info()
    assert info["arch"].startswith("gfx950") and info["compute_units"] >= 200


# ------------------------------------------------------------------------------------------
# round 2: degrade instead of raising, draw-ahead on the device
# ------------------------------------------------------------------------------------------
@pytest.fixture
def _fresh_split_state():
    ctx = _lib.Context.get(0)
    yield ctx
    ctx.set_spin_limit(0)
    ctx.set_split(1)
    ctx.split_degraded = False
    ctx.launch_status_value()


@pytest.mark.parametrize("cid", ["c2_hc_rs_n2000_h30_e5_s0", "c5_hc_cem_n4000_h30_e5_s0", "c1_hc_rs_n500_h10_e1_s0"])
def test_flagged_launch_is_relaunched_unsplit_not_raised(cid, _fresh_split_state):
    """A status word set by a launch (injected on the host here, deterministically) must make the controller switch
    the context to the unsplit geometry and relaunch - same action as the reference, no exception."""
    ctx = _fresh_split_state
    case, seed = cases.split_id(cid)
    gold = cases.load_golden(cid)
    ctrl = cases.product_controller(case)
    ctrl.dynamics_model.planner_model()            # create the context's model before injecting
    np.random.seed(seed)
    ctx.check(ctx.lib.l2a_inject_status(ctx.handle, 1), "l2a_inject_status")
    actions, _ = ctrl.get_actions(gold["obs0"])
    assert ctx.split_degraded is True
    assert np.random.uniform() == float(gold["rng_next"])
    if case["planner"] != "cem":
        assert np.array_equal(ctrl.last_plan["best_index"], gold["best"])
        np.testing.assert_array_equal(actions, gold["chosen"])
    # a second flagged launch with the split already off is a real error
    ctx.check(ctx.lib.l2a_inject_status(ctx.handle, 1), "l2a_inject_status")
    with pytest.raises(_lib.L2AError):
        ctrl.get_actions(gold["obs0"])


def test_plan_payload_packs_keys_flag_and_digest_on_the_device(_fresh_split_state):
    """``l2a_plan_payload`` (what the ranks of a sharded plan all-reduce): keys copied, the launch status word read ON
    THE DEVICE in stream order behind the launch, the digest pair summing to the mask."""
    ctx = _fresh_split_state
    case = cases.CASES["hc_rs_m3_n64_h5"]
    env, model = cases.product_model(case)
    native = model.planner_model()
    dev = native.device
    m = case["m"]
    a = _rs_actions(case, 0, env)
    best = torch.zeros((m,), dtype=torch.int64, device=dev)
    native.plan_rs(torch.from_numpy(np.random.RandomState(1).randn(m, 20).astype(np.float32)).to(dev),
                   torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev), m, case["n"], case["h"], 1.0,
                   env.reward_spec, best_key=best)
    payload = torch.full((m + 3,), -1, dtype=torch.int64, device=dev)
    mask = 0x7FFFFFFFFFFF
    digest = 0x123456789ABC
    native.plan_payload(best, m, digest | (1 << 60), payload)          # bits above the mask are dropped
    got = payload.cpu().numpy()
    assert np.array_equal(got[:m], best.cpu().numpy()) and got[m] == 0
    assert got[m + 1] == digest and got[m + 1] + got[m + 2] == mask
    ctx.check(ctx.lib.l2a_inject_status(ctx.handle, 1), "l2a_inject_status")        # as if the launch had been flagged
    native.plan_payload(best, m, digest, payload)
    assert int(payload.cpu()[m]) == 1
    assert ctx.launch_status_value() == 1                                           # (read and cleared)
    # MAX over ranks is what turns the pairs into an agreement test: two different digests cannot sum to the mask
    other = np.array([digest + 1, mask - (digest + 1)])
    merged = np.maximum(got[m + 1:], other)
    assert merged[0] + merged[1] != mask


def test_exchange_timeout_in_the_kernel_degrades(_fresh_split_state):
    """The kernel's own time-out path: with one poll allowed per launch a split workgroup almost surely misses its
    partner at some horizon step and flags the launch; whether or not it does, the controller must return the
    reference's action (relaunching unsplit when flagged), and the launch must not take long."""
    import time
    ctx = _fresh_split_state
    cid = "c2_hc_rs_n2000_h30_e5_s1"
    case, seed = cases.split_id(cid)
    gold = cases.load_golden(cid)
    ctrl = cases.product_controller(case, draw_ahead=False)
    ctrl.dynamics_model.planner_model()
    ctx.set_spin_limit(1)
    np.random.seed(seed)
    t = time.perf_counter()
    actions, _ = ctrl.get_actions(gold["obs0"])
    assert time.perf_counter() - t < 5.0
    assert np.array_equal(ctrl.last_plan["best_index"], gold["best"])
    np.testing.assert_array_equal(actions, gold["chosen"])
    predicted = ctrl.dynamics_model.predict(np.repeat(gold["obs0"], 64, axis=0), np.zeros((64, 6)))
    ctx.set_spin_limit(0)
    ctx.set_split(0)
    want = ctrl.dynamics_model.predict(np.repeat(gold["obs0"], 64, axis=0), np.zeros((64, 6)))
    np.testing.assert_array_equal(predicted, want)


@pytest.mark.parametrize("name", ["c2_hc_rs_n2000_h30_e5", "c3b_ant_rs_n500_h10_pb5_3x512", "hc_cem_m2_n100_h4"])
def test_draw_ahead_on_the_device_changes_nothing(name):
    """Consecutive controller steps with the draw-ahead chain (candidates of step k + 1 drawn and uploaded on a side
    stream while step k runs) against the same steps without it: actions, indices, returns and the generator state
    afterwards are identical; the chain really was used."""
    case = cases.CASES[name]
    env, model = cases.product_model(case)
    rs = np.random.RandomState(5)
    obs = [rs.randn(case["m"], env.observation_space.shape[0]) for _ in range(4)]
    outs = []
    for ahead in (False, True):
        ctrl = cases.product_controller(case, model=model, env=env, draw_ahead=ahead)
        np.random.seed(11)
        seq = []
        for k in range(4):
            a, _ = ctrl.get_actions(obs[k])
            seq.append((a.copy(), np.array(ctrl.last_plan["best_index"]), np.array(ctrl.last_plan["best_return"])))
            if k == 1:
                np.random.normal(size=3)           # a foreign draw: the prepared block must be dropped
        outs.append((seq, np.random.uniform(), ctrl))
    (s0, t0, _), (s1, t1, c1) = outs
    assert t0 == t1
    for (a0, i0, r0), (a1, i1, r1) in zip(s0, s1):
        assert np.array_equal(a0, a1) and np.array_equal(i0, i1) and np.array_equal(r0, r1)
    assert c1.draw_ahead_stats()["hits"] >= 2           # (the C controller's chain for random shooting, the Python chain for CEM)
    for c in (c1,):
        if c._cstep is not None:
            c._cstep.close()
            c._cstep = None


@pytest.mark.parametrize("kernel", ["auto", "valu"])
@pytest.mark.parametrize("name,over", [("c2_hc_rs_n2000_h30_e5", dict(h=6)), ("hc_rs_m3_n64_h5", {}),
                                        ("c3_ant_rs_n2000_h20_pb5", dict(n=333, h=4)), ("c1_hc_rs_n500_h10_e1", {})])
def test_blocking_plan_publishes_the_same_keys(name, over, kernel):
    """`l2a_plan_rs_sync` (observations read from host-mapped memory, keys published to the host-mapped mailbox by
    the last candidate tile, key slot zeroed by the previous launch) against `l2a_plan_rs` + read-back: identical
    keys over a run of back-to-back calls with changing inputs (both ring slots, counter re-arming), also when
    classic launches are interleaved and for the VALU kernel (which takes the copy-back path inside)."""
    case = dict(cases.CASES[name], **over)
    if kernel == "valu" and case["n"] * case["m"] * case["h"] > 20000:
        pytest.skip("VALU kernel: small cases only")
    _set_kernel(kernel)
    env, model = cases.product_model(case)
    native = model.planner_model()
    dev = native.device
    m, n, h = case["m"], case["n"], case["h"]
    rs = np.random.RandomState(4)
    od = env.observation_space.shape[0]
    best = torch.zeros((m,), dtype=torch.int64, device=dev)
    # a second model with ONE env planning on the same context in between: its launches zero only one entry of
    # the other key slot (the slots are per context, shared by every model on it)
    one = cases.CASES["hc_rs_n1_h1"]
    env1, model1 = cases.product_model(one)
    native1 = model1.planner_model()
    a1 = torch.from_numpy(np.ascontiguousarray(_rs_actions(one, 1, env1), dtype=np.float32)).to(dev)
    obs1 = rs.randn(1, env1.observation_space.shape[0])
    for it in range(7):
        for _ in range(it % 3):
            assert native1.plan_rs_sync(obs1, a1, 1, one["n"], one["h"], 1.0, env1.reward_spec) is not None
        a = _rs_actions(case, 10 + it, env)
        a_dev = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
        obs0 = rs.randn(m, od)
        native.plan_rs(torch.from_numpy(obs0.astype(np.float32)).to(dev), a_dev, m, n, h, case.get("discount", 1.0),
                       env.reward_spec, cand_offset=3 * it, best_key=best)
        want = best.cpu().numpy()
        got = native.plan_rs_sync(obs0, a_dev, m, n, h, case.get("discount", 1.0), env.reward_spec, cand_offset=3 * it)
        assert got is not None and np.array_equal(got.view(np.int64), want), (it, got, want)
        if it % 3 == 2:     # two blocking launches in a row: the second one's key slot was zeroed by the first
            got2 = native.plan_rs_sync(obs0, a_dev, m, n, h, case.get("discount", 1.0), env.reward_spec,
                                       cand_offset=3 * it)
            assert np.array_equal(got2, got)
    _lib.Context.get(0).launch_status()


def _philox_normal_ref(seed, index):
    """Python restatement of csrc/l2a_cem.hip's generator: Philox4x32-10 (Salmon et al. 2011) on counter
    (index_lo, index_hi, 0x4c32614d, 0) under key (seed_lo, seed_hi), Box-Muller on the first two words."""
    M = 0xFFFFFFFF
    c = [index & M, (index >> 32) & M, 0x4C32614D, 0]
    k0, k1 = seed & M, (seed >> 32) & M
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c[0], 0xCD9E8D57 * c[2]
        c = [(p1 >> 32) ^ c[1] ^ k0, p1 & M, (p0 >> 32) ^ c[3] ^ k1, p0 & M]
        k0, k1 = (k0 + 0x9E3779B9) & M, (k1 + 0xBB67AE85) & M
    u1 = np.float32(np.float32((c[0] >> 8) + 1.0) * np.float32(1.0 / 16777216.0))
    u2 = np.float32(np.float32(c[1] >> 8) * np.float32(1.0 / 16777216.0))
    return float(np.sqrt(-2.0 * np.log(float(u1))) * np.cos(6.28318530717958647692 * float(u2)))


@pytest.mark.parametrize("reference", [True, False])
def test_cem_kernels_match_numpy(reference):
    """`l2a_cem_sample` / `l2a_cem_refit` / `l2a_cem_pick` on their own against NumPy: samples, clip and the rollout's candidate tensor (both
    row readings, a candidate shard), the library's Philox normals (values of single counters, moments, determinism),
    elite rows and refit - including exact ties in the returns, where the stable descending order decides."""
    from learning_to_adapt_amd.dynamics.native_model import _ptr, _stream_ptr
    ctx = _lib.Context.get(0)
    lib = ctx.lib
    dev = torch.device("cuda:0")
    n, m, h, ad, k, alpha = 333, 3, 5, 4, 33, 0.1
    D = h * ad
    rs = np.random.RandomState(7)
    z = rs.randn(n, m, D).astype(np.float32)
    mean = (0.3 * rs.randn(m, D)).astype(np.float32)
    std = (0.5 + rs.rand(m, D)).astype(np.float32)
    low, high = (-0.8 * np.ones(ad)).astype(np.float32), (0.9 * np.ones(ad)).astype(np.float32)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    lo, hi = 100, 250
    a_clip = torch.empty((n, m, D), device=dev)
    a_raw = torch.empty((n, m, D), device=dev)
    seq = torch.full((h, m * (hi - lo), ad), float("nan"), device=dev)
    mean_d, std_d, z_d, low_d, high_d = up(mean), up(std), up(z), up(low), up(high)     # (kept alive: launches are async)
    ctx.check(lib.l2a_cem_sample(ctx.handle, _ptr(z_d), 0, 0, _ptr(mean_d), _ptr(std_d), _ptr(low_d), _ptr(high_d),
                                 n, m, h, ad, 1 if reference else 0, lo, hi, _ptr(a_clip), _ptr(a_raw), _ptr(seq),
                                 _stream_ptr(dev)), "l2a_cem_sample")
    a_want = mean[None] + z * std[None]
    c_want = np.clip(a_want, np.tile(low, h), np.tile(high, h))
    np.testing.assert_allclose(a_raw.cpu().numpy(), a_want, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(a_clip.cpu().numpy(), c_want, rtol=1e-6, atol=1e-6)
    if reference:       # the reference reads its candidate-major rows as [m, n, D] (:92-96), unclipped
        full = np.transpose(a_want.reshape(n * m, h, ad), (1, 0, 2)).reshape(h, m, n, ad)
    else:
        full = np.transpose(c_want.transpose(1, 0, 2).reshape(m * n, h, ad), (1, 0, 2)).reshape(h, m, n, ad)
    np.testing.assert_allclose(seq.cpu().numpy(), full[:, :, lo:hi, :].reshape(h, m * (hi - lo), ad), rtol=1e-6, atol=1e-6)

    # the library's own normals: counter -> value, moments, determinism, disjoint offsets
    seed = 0x1234567887654321
    g1 = torch.empty((n, m, D), device=dev)
    g2 = torch.empty((n, m, D), device=dev)
    zero, one = torch.zeros((m, D), device=dev), torch.ones((m, D), device=dev)
    wide_lo, wide_hi = up(np.full(ad, -1e9, dtype=np.float32)), up(np.full(ad, 1e9, dtype=np.float32))
    for buf, off in ((g1, 1000), (g2, 1000 + n * m * D)):
        ctx.check(lib.l2a_cem_sample(ctx.handle, None, ctypes.c_ulonglong(seed), ctypes.c_ulonglong(off), _ptr(zero), _ptr(one),
                                     _ptr(wide_lo), _ptr(wide_hi), n, m, h, ad, 1, 0, 0, _ptr(buf), None, None,
                                     _stream_ptr(dev)), "l2a_cem_sample")
    v1, v2 = g1.cpu().numpy().reshape(-1), g2.cpu().numpy().reshape(-1)
    for e in (0, 1, 2, 63, 64, 4999, len(v1) - 1):
        assert abs(v1[e] - _philox_normal_ref(seed, 1000 + e)) < 2e-5, e
    assert np.isfinite(v1).all() and abs(v1.mean()) < 0.03 and abs(v1.std() - 1.0) < 0.03
    assert abs(np.corrcoef(v1[:-1], v1[1:])[0, 1]) < 0.03 and abs(np.corrcoef(v1, v2)[0, 1]) < 0.03
    g3 = torch.empty((n, m, D), device=dev)
    ctx.check(lib.l2a_cem_sample(ctx.handle, None, ctypes.c_ulonglong(seed), ctypes.c_ulonglong(1000), _ptr(zero), _ptr(one),
                                 _ptr(wide_lo), _ptr(wide_hi), n, m, h, ad, 1, 0, 0, _ptr(g3), None, None,
                                 _stream_ptr(dev)), "l2a_cem_sample")
    assert torch.equal(g1, g3)

    # refit: returns with exact ties
    rets = rs.randn(m, n).astype(np.float32)
    rets[:, 5] = rets[:, 17]
    rets[0, 40:44] = rets[0, 2]
    rets[1, [7, 90, 200]] = np.nan                 # diverged rollouts: they sort last, like np.argsort(-returns) has them
    rets[2, 11] = -np.inf
    rows = torch.empty((m * k,), dtype=torch.int32, device=dev)
    rets_d = up(rets)
    ctx.check(lib.l2a_cem_refit(ctx.handle, _ptr(rets_d), _ptr(a_clip), n, m, D, k, 1 if reference else 0, alpha,
                                _ptr(rows), _ptr(mean_d), _ptr(std_d), _stream_ptr(dev)), "l2a_cem_refit")
    a_st = a_clip.cpu().numpy().astype(np.float64)
    order = np.argsort(-np.where(np.isnan(rets), -np.inf, rets).astype(np.float64), axis=1, kind="stable")
    if reference:
        mask = (order < k).T                                            # mpc_controller.py:101
        elites = a_st[mask]
        want_mean = mean * alpha + (1 - alpha) * elites.mean(axis=0)
        want_std = np.broadcast_to(elites.std(axis=0), (m, D))
        got_rows = np.sort(rows.cpu().numpy().reshape(m, k), axis=1)
        want_rows = np.stack([np.sort(np.nonzero(mask[:, i])[0] * m + i) for i in range(m)])
    else:
        elites = np.stack([a_st[order[i, :k], i] for i in range(m)])    # [m, k, D]
        want_mean = mean * alpha + (1 - alpha) * elites.mean(axis=1)
        want_std = elites.std(axis=1)
        got_rows = rows.cpu().numpy().reshape(m, k)
        want_rows = order[:, :k] * m + np.arange(m)[:, None]
    assert np.array_equal(got_rows, want_rows)
    np.testing.assert_allclose(mean_d.cpu().numpy(), want_mean, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(std_d.cpu().numpy(), want_std, rtol=2e-5, atol=2e-6)

    # l2a_cem_pick: arg-max (first maximum - env 0 has an exact tie at its top), that candidate's first action in the reading of
    # the mode, its return, and the final mean / std, all in one buffer
    rets2 = rets.copy()
    rets2[0, 50] = rets2[0, 300] = np.nanmax(rets2[0]) + 1.0
    rets2_d = up(rets2)
    packed = torch.full((m * (ad + 2) + 2 * m * D,), float("nan"), device=dev)
    cand = a_raw if reference else a_clip
    ctx.check(lib.l2a_cem_pick(ctx.handle, _ptr(rets2_d), _ptr(cand), _ptr(mean_d), _ptr(std_d), n, m, D, ad, 1 if reference else 0,
                               _ptr(packed), _stream_ptr(dev)), "l2a_cem_pick")
    host = packed.cpu().numpy()
    head = host[:m * (ad + 2)].reshape(m, ad + 2)
    # np.argmax's order, as the reference's `np.argmax(returns, axis=1)` (:128-129): env 1's first NaN wins and its NaN return
    # is what comes back (a diverged plan must not look finite)
    want_idx = np.array([int(np.argmax(rets2[i])) for i in range(m)])
    assert want_idx[0] == 50 and want_idx[1] == 7
    assert np.array_equal(head[:, ad + 1].copy().view(np.int32), want_idx)
    assert np.array_equal(head[:, ad], rets2[np.arange(m), want_idx], equal_nan=True) and np.isnan(head[1, ad])
    cand_h = cand.cpu().numpy()
    view = cand_h.reshape(m, n, D) if reference else cand_h.transpose(1, 0, 2)       # (:92-96 / candidate j of env i = row j * m + i)
    assert np.array_equal(head[:, :ad], view[np.arange(m), want_idx, :ad])
    assert np.array_equal(host[m * (ad + 2):m * (ad + 2) + m * D].reshape(m, D), mean_d.cpu().numpy())
    assert np.array_equal(host[m * (ad + 2) + m * D:].reshape(m, D), std_d.cpu().numpy())
