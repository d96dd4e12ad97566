"""Randomised shape sweep on the GPU: model shapes, ensemble modes, plan sizes, discounts and reward
kinds drawn from a seeded generator; every candidate's return is compared with the oracle (rel 1e-4)
and the arg-max key with the returns the kernel itself wrote (bit exact).  Covers the corners between
the hand-picked golden cases: obs/act dims that straddle 16-feature tiles, every MFMA-eligible hidden
width and depth, odd ensembles under all tile-split policies, ragged candidate counts."""

import os

import numpy as np
import pytest
import torch

from learning_to_adapt_amd import _lib
from learning_to_adapt_amd.dynamics.native_lstm import NativeLSTM
from learning_to_adapt_amd.dynamics.native_model import NativeModel
from learning_to_adapt_amd.envs import RewardSpec
from learning_to_adapt_amd.utils import synthetic
from oracle import LSTMStateTuple, OracleLSTMDynamics, OracleMLPDynamics
from oracle.planner import rollout_returns
from oracle.rnn_planner import rnn_rollout_returns

pytestmark = pytest.mark.gpu

# more seeds for a one-off sweep: L2A_RANDOM_SEEDS=100 python -m pytest tests/test_gpu_random_shapes.py -m gpu
_EXTRA = int(os.environ.get("L2A_RANDOM_SEEDS", "0"))


def _reward(rs, obs_dim, act_dim):
    kind = rs.randint(3)
    if kind == 0:       # velocity + control cost (half-cheetah form), random index / coefficients
        spec = RewardSpec.make(w_vel=float(rs.uniform(0.5, 2.0)), dt=float(rs.choice([0.01, 0.02, 0.05])),
                               ctrl_coef=float(rs.uniform(0.0, 0.1)), vel_index=int(rs.randint(obs_dim)))
    elif kind == 1:     # velocity + alive bonus (ant form)
        spec = RewardSpec.make(w_vel=1.0, dt=0.02, alive=0.05, vel_index=obs_dim - 1)
    else:               # distance + control cost (arm form)
        spec = RewardSpec.make(dist_coef=1.0, ctrl_coef=0.005, dist_index=int(rs.randint(max(obs_dim - 2, 1))))
    return spec


def _norm(rs, obs_dim, act_dim, low, high):
    return synthetic.make_norm(obs_dim, act_dim, low, high, int(rs.randint(1 << 30)))


def _check(got, want, keys, n, offset):
    scale = max(1.0, float(np.max(np.abs(want))))
    assert float(np.max(np.abs(got - want))) / scale < 1e-4
    for i in range(got.shape[0]):
        ret, idx = _lib.key_decode(keys[i])
        assert idx - offset == int(np.argmax(got[i])) and ret == got[i, idx - offset]


@pytest.mark.parametrize("seed", range(max(12, _EXTRA)))
def test_random_mlp_shapes_match_oracle(seed):
    rs = np.random.RandomState(100 + seed)
    obs_dim = int(rs.choice([3, 15, 16, 17, 20, 31, 33, 41, 48, 64]))
    act_dim = int(rs.choice([1, 6, 8, 12, 16]))
    if obs_dim + act_dim > 80:
        act_dim = 80 - obs_dim
    width = int(rs.choice([128, 256, 512, 96]))
    depth = int(rs.choice([1, 2, 3]))
    hidden = [width] * depth
    mode = str(rs.choice(["single", "mean", "per_block"]))
    E = 1 if mode == "single" else int(rs.choice([2, 3, 5]))
    m = E if mode == "per_block" else int(rs.choice([1, 2, 3]))
    n = int(rs.choice([1, 15, 16, 17, 47, 100, 260]))
    h = int(rs.choice([1, 2, 5, 9]))
    discount = float(rs.choice([1.0, 0.9]))
    act = str(rs.choice(["relu", "relu", "tanh"]))
    low, high = -np.ones(act_dim) * 2.0, np.ones(act_dim) * 2.0
    sets = [synthetic.make_weight_set(obs_dim, act_dim, hidden, int(rs.randint(1 << 30))) for _ in range(E)]
    norms = [_norm(rs, obs_dim, act_dim, low, high) for _ in range(E)]
    if mode == "per_block":
        norms = [norms[0]] * E
    spec = _reward(rs, obs_dim, act_dim)
    dyn = OracleMLPDynamics(obs_dim, act_dim, sets, norms, mode=mode, hidden_nonlinearity=act)
    obs0 = rs.randn(m, obs_dim)
    acts = rs.uniform(low, high, (h, m * n, act_dim))
    want = rollout_returns(dyn, spec.evaluate, obs0, acts, n, discount).reshape(m, n)

    native = NativeModel(obs_dim, act_dim, hidden, act, None, E, mode)
    for e in range(E):
        native.set_weights(e, sets[e])
        native.set_norm(e, norms[e])
    dev = native.device
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)  # noqa: E731
    offset = int(rs.choice([0, 1000]))
    results = []
    ctx = _lib.Context.get(0)
    try:
        # (split 1 with the member fan - what small mean ensembles take by default - and without it: the tile split)
        for split, fan in ((1, 1), (0, 1), (2, 1), (1, 0)):
            ctx.set_split(split)
            ctx.set_fan(fan)
            rets = torch.empty((m, n), dtype=torch.float32, device=dev)
            best = torch.zeros((m,), dtype=torch.int64, device=dev)
            native.plan_rs(up(obs0), up(acts), m, n, h, discount, spec, cand_offset=offset, returns_out=rets,
                           best_key=best)
            got, keys = rets.cpu().numpy(), best.cpu().numpy()
            ctx.launch_status()
            _check(got, want, keys, n, offset)
            results.append(got)
    finally:
        ctx.set_split(1)
        ctx.set_fan(1)
    assert all(np.array_equal(results[0], r) for r in results[1:])
    native.close()


@pytest.mark.parametrize("seed", range(max(8, _EXTRA)))
def test_random_lstm_shapes_match_oracle(seed):
    rs = np.random.RandomState(200 + seed)
    obs_dim = int(rs.choice([3, 16, 17, 20, 33, 41, 64]))
    act_dim = int(rs.choice([1, 6, 8, 16]))
    units = int(rs.choice([128, 256, 512, 72]))
    m = int(rs.choice([1, 2, 4]))
    n = int(rs.choice([1, 16, 17, 50, 130]))
    h = int(rs.choice([1, 2, 4, 7]))
    discount = float(rs.choice([1.0, 0.95]))
    act = str(rs.choice(["tanh", "tanh", "relu"]))
    low, high = -np.ones(act_dim), np.ones(act_dim)
    params = synthetic.make_lstm_set(obs_dim, act_dim, units, int(rs.randint(1 << 30)))
    norm = _norm(rs, obs_dim, act_dim, low, high)
    spec = _reward(rs, obs_dim, act_dim)
    dyn = OracleLSTMDynamics(obs_dim, act_dim, params, norm, hidden_nonlinearity=act)
    obs0 = rs.randn(m, obs_dim)
    hid = LSTMStateTuple(rs.randn(m, units).astype(np.float32), np.tanh(rs.randn(m, units)).astype(np.float32))
    acts = rs.uniform(low, high, (h, m * n, act_dim))
    want = rnn_rollout_returns(dyn, spec.evaluate, obs0, hid, acts, n, discount).reshape(m, n)

    native = NativeLSTM(obs_dim, act_dim, units, act, None)
    native.set_weights(params)
    native.set_norm(norm)
    dev = native.device
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)  # noqa: E731
    rets = torch.empty((m, n), dtype=torch.float32, device=dev)
    best = torch.zeros((m,), dtype=torch.int64, device=dev)
    native.plan_rs(up(obs0), up(hid.c), up(hid.h), up(acts), m, n, h, discount, spec, cand_offset=7,
                   returns_out=rets, best_key=best)
    _check(rets.cpu().numpy(), want, best.cpu().numpy(), n, 7)
    native.close()


@pytest.mark.parametrize("seed", range(max(10, _EXTRA)))
def test_random_recurrent_stacks_matrix_core_matches_valu(seed):
    """GRU / BasicRNN / LSTM stacks of odd widths (units that are no multiple of 16 or of 4, inputs that straddle k-groups,
    action vectors wider than the prefetch registers): the matrix-core kernel (l2a_rnn_mfma.h) against the VALU kernel
    (l2a_rnn_valu.h) - every return, the arg-max keys' consistency, and the states `predict` writes."""
    rs = np.random.RandomState(900 + seed)
    cell = str(rs.choice(["gru", "lstm", "rnn"]))
    n_layers = int(rs.choice([1, 2, 3])) if cell != "lstm" else int(rs.choice([2, 3]))
    sizes = [int(rs.choice([5, 16, 23, 40, 57, 64, 96, 130])) for _ in range(n_layers)]
    obs_dim = int(rs.choice([3, 16, 17, 20, 33]))
    act_dim = int(rs.choice([1, 6, 8, 16, 40]))
    m = int(rs.choice([1, 2, 3]))
    n = int(rs.choice([1, 16, 17, 50]))
    h = int(rs.choice([1, 2, 5]))
    discount = float(rs.choice([1.0, 0.9]))
    act = str(rs.choice(["tanh", "tanh", "relu", "sigmoid"]))
    low, high = -np.ones(act_dim), np.ones(act_dim)
    params = synthetic.make_rnn_stack_set(obs_dim, act_dim, sizes, cell, int(rs.randint(1 << 30)))
    norm = _norm(rs, obs_dim, act_dim, low, high)
    spec = _reward(rs, obs_dim, act_dim)
    U = sum(sizes)
    native = NativeLSTM(obs_dim, act_dim, sizes, act, None, cell_type=cell)
    native.set_weights(params)
    native.set_norm(norm)
    dev = native.device
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)  # noqa: E731
    obs0, acts = up(rs.randn(m, obs_dim)), up(rs.uniform(low, high, (h, m * n, act_dim)))
    c0 = up(rs.randn(m, U) * (1.0 if cell == "lstm" else 0.0))
    h0 = up(np.tanh(rs.randn(m, U)))
    ctx = _lib.Context.get(0)
    got = {}
    try:
        for kernel in ("mfma", "valu"):
            ctx.set_kernel(kernel)
            rets = torch.empty((m, n), dtype=torch.float32, device=dev)
            best = torch.zeros((m,), dtype=torch.int64, device=dev)
            native.plan_rs(obs0, c0, h0, acts, m, n, h, discount, spec, cand_offset=3, returns_out=rets, best_key=best)
            nxt = native.predict(obs0, acts[0, ::n][:m].contiguous(), c0, h0)
            got[kernel] = (rets.cpu().numpy(), best.cpu().numpy(), [x.cpu().numpy() for x in nxt])
    finally:
        ctx.set_kernel("auto")
    (r_m, k_m, s_m), (r_v, k_v, s_v) = got["mfma"], got["valu"]
    _check(r_m, r_v, k_m, n, 3)
    for a, b in zip(s_m, s_v):
        np.testing.assert_allclose(a, b, rtol=2e-5, atol=2e-6)
    native.close()


@pytest.mark.parametrize("seed", range(max(12, _EXTRA)))
def test_random_recurrent_micro_tile_shapes_match_valu(seed):
    """The micro-tile form of the generic recurrent kernel (csrc/l2a_rnn_micro.h; layers of 256 - GRU / BasicRNN also 512 - units)
    against the VALU kernel on random input shapes: observation / action widths that end anywhere in a 16-feature group - an ODD
    number of input groups takes the zero-padded one of the eight-deep operand ring -, 1 - 3 layers, every cell type and reward
    family, plans of one micro tile up to several rounds of four-tile workgroups, ragged last tiles, discount."""
    rs = np.random.RandomState(1700 + seed)
    cell = str(rs.choice(["gru", "lstm", "rnn"]))
    U = 512 if (cell != "lstm" and rs.rand() < 0.25) else 256
    n_layers = 1 if U == 512 else (int(rs.choice([1, 2, 3])) if cell != "lstm" else int(rs.choice([2, 3])))
    sizes = [U] * n_layers
    obs_dim = int(rs.choice([3, 11, 16, 17, 20, 33, 41, 48, 64]))
    act_dim = int(rs.choice([1, 2, 6, 8, 13, 16]))
    if obs_dim + act_dim > 80:
        act_dim = 80 - obs_dim
    m = int(rs.choice([1, 2, 5]))
    n = int(rs.choice([1, 5, 37, 500, 1300, 4100]))
    h = int(rs.choice([1, 2, 4]))
    act = str(rs.choice(["tanh", "tanh", "relu", "sigmoid"]))
    low, high = -np.ones(act_dim), np.ones(act_dim)
    params = synthetic.make_rnn_stack_set(obs_dim, act_dim, sizes, cell, int(rs.randint(1 << 30)))
    norm = _norm(rs, obs_dim, act_dim, low, high)
    spec = _reward(rs, obs_dim, act_dim)
    W = sum(sizes)
    native = NativeLSTM(obs_dim, act_dim, sizes, act, None, cell_type=cell)
    native.set_weights(params)
    native.set_norm(norm)
    dev = native.device
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)  # noqa: E731
    obs0, acts = up(rs.randn(m, obs_dim)), up(rs.uniform(low, high, (h, m * n, act_dim)))
    c0 = up(rs.randn(m, W) * (1.0 if cell == "lstm" else 0.0))
    h0 = up(np.tanh(rs.randn(m, W)))
    ctx = _lib.Context.get(0)
    got = {}
    try:
        # ("auto" with policy 0: a stack whose 16-candidate matrix-core kernel does not fit the LDS runs the VALU kernel; the
        #  micro-tile kernel still takes it under policies 1 / 2 - hence tolerances here, not bit identity)
        for name, kernel, micro in (("micro", "auto", 2), ("tiles16", "auto", 0), ("valu", "valu", 0)):
            if name == "valu" and m * n * h > 12000:
                continue
            ctx.set_kernel(kernel)
            ctx.set_micro(micro)
            rets = torch.empty((m, n), dtype=torch.float32, device=dev)
            best = torch.zeros((m,), dtype=torch.int64, device=dev)
            native.plan_rs(obs0, c0, h0, acts, m, n, h, 0.95, spec, cand_offset=3, returns_out=rets, best_key=best)
            torch.cuda.synchronize()
            ctx.launch_status()
            got[name] = (rets.cpu().numpy(), best.cpu().numpy())
    finally:
        ctx.set_kernel("auto")
        ctx.set_micro(1)
    _check(got["micro"][0], got["tiles16"][0], got["micro"][1], n, 3)
    if "valu" in got:
        _check(got["micro"][0], got["valu"][0], got["micro"][1], n, 3)
    native.close()


@pytest.mark.parametrize("m,n", [(129, 17), (200, 20), (160, 19), (256, 18)])
def test_recurrent_micro_tiles_with_more_envs_than_half_the_cus(m, n):
    """with more envs than CUs / 2 and five micro tiles per env, forcing workgroups of FOUR micro tiles left a
    NEGATIVE count of full workgroups (quads = 5: W = 2, 5 - 2 * 3 = -1) - workgroup 0 started at candidate -4, wrote in front
    of its env's returns and offered negative indices to the arg-max.  The dispatch now takes the natural ceil(quads / W);
    sentinels in front of and behind the returns prove nothing is written out of bounds."""
    rs = np.random.RandomState(7 * m + n)
    obs_dim, act_dim, h = 20, 6, 2
    low, high = -np.ones(act_dim), np.ones(act_dim)
    params = synthetic.make_rnn_stack_set(obs_dim, act_dim, [256], "gru", 99)
    norm = _norm(rs, obs_dim, act_dim, low, high)
    spec = _reward(rs, obs_dim, act_dim)
    native = NativeLSTM(obs_dim, act_dim, [256], "tanh", None, cell_type="gru")
    native.set_weights(params)
    native.set_norm(norm)
    dev = native.device
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)  # noqa: E731
    obs0, acts = up(rs.randn(m, obs_dim)), up(rs.uniform(low, high, (h, m * n, act_dim)))
    c0, h0 = up(np.zeros((m, 256))), up(np.tanh(rs.randn(m, 256)))
    ctx = _lib.Context.get(0)
    got = {}
    guard = 64
    try:
        for name, kernel, micro in (("micro", "auto", 2), ("tiles16", "auto", 0), ("valu", "valu", 0)):
            ctx.set_kernel(kernel)
            ctx.set_micro(micro)
            flat = torch.full((m * n + 2 * guard,), 12345.0, dtype=torch.float32, device=dev)
            rets = flat[guard:guard + m * n].view(m, n)
            best = torch.zeros((m,), dtype=torch.int64, device=dev)
            native.plan_rs(obs0, c0, h0, acts, m, n, h, 0.95, spec, cand_offset=3, returns_out=rets, best_key=best)
            torch.cuda.synchronize()
            ctx.launch_status()
            host = flat.cpu().numpy()
            assert np.all(host[:guard] == 12345.0) and np.all(host[guard + m * n:] == 12345.0), name
            got[name] = (host[guard:guard + m * n].reshape(m, n).copy(), best.cpu().numpy())
    finally:
        ctx.set_kernel("auto")
        ctx.set_micro(1)
    _check(got["micro"][0], got["tiles16"][0], got["micro"][1], n, 3)
    _check(got["micro"][0], got["valu"][0], got["micro"][1], n, 3)
    native.close()


@pytest.mark.parametrize("seed", range(max(10, _EXTRA)))
def test_random_adaptation_shapes_match_autograd_and_their_packed_copies_match_the_packer(seed):
    """The GrBAL adaptation launches (csrc/l2a_adapt.h; round 5: every layer's update rides in the backward launch that follows its
    dZ, layer 0's columns are updated by the last backward workgroups) on random shapes - observation / action widths anywhere
    in a tile, 1 - 3 hidden layers of widths on and off the tile sizes, 1 - 16 rows, 1 - 5 tasks, every nonlinearity - against
    a float64 autograd step; and the three copies the update writes (reference layout, MFMA fragment order, micro-tile order)
    against the library's own packer: a second model loaded with the adapted raw weights through `set_weights` must plan
    bit for bit like the adapted one, on the 16-candidate and on the micro-tile kernel."""
    rs = np.random.RandomState(4200 + seed)
    od = int(rs.choice([3, 11, 16, 17, 20, 33, 41, 48, 64]))
    ad = int(rs.choice([1, 2, 6, 8, 13, 16]))
    depth = int(rs.choice([1, 2, 3]))
    same = rs.rand() < 0.6          # equal widths of 128 / 256 / 512: the matrix-core (and micro-tile) planner kernels take the model
    hidden = [int(rs.choice([128, 256, 512]))] * depth if same else [int(rs.choice([64, 72, 128, 200, 256])) for _ in range(depth)]
    act = str(rs.choice(["relu", "tanh", "sigmoid"]))
    m, rows = int(rs.choice([1, 2, 3, 5])), int(rs.choice([1, 3, 7, 16]))
    lr = float(rs.choice([0.01, 0.1]))
    dev = torch.device("cuda:0")
    low, high = -np.ones(ad), np.ones(ad)
    base_np = synthetic.make_weight_set(od, ad, hidden, int(rs.randint(1 << 30)))
    base = [torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32)).to(dev) for w in base_np]
    x = rs.randn(m, rows, od + ad).astype(np.float32)
    y = rs.randn(m, rows, od).astype(np.float32)
    adapted = NativeModel(od, ad, hidden, act, None, m, "per_block")
    adapted.adapt_sgd(base, torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev), lr)
    got = [[w.cpu().numpy() for w in adapted.get_weights(e)] for e in range(m)]
    # float64 autograd step per task (meta_mlp_dynamics.py:409-421, loss :118)
    fn = {"relu": torch.relu, "tanh": torch.tanh, "sigmoid": torch.sigmoid}[act]
    for e in range(m):
        params = [torch.tensor(np.asarray(w, dtype=np.float64), requires_grad=True) for w in base_np]
        t = torch.from_numpy(x[e].astype(np.float64))
        for li in range(depth + 1):
            t = t @ params[2 * li] + params[2 * li + 1]
            if li < depth:
                t = fn(t)
        loss = torch.mean((torch.from_numpy(y[e].astype(np.float64)) - t) ** 2)
        grads = torch.autograd.grad(loss, params)
        for w0, g, w1 in zip(params, grads, got[e]):
            want = (w0 - lr * g).detach().numpy()
            step = float(np.abs(lr * g.numpy()).max())
            assert np.abs(w1 - want).max() <= 3e-5 * max(step, 1e-3) + 2e-7, (seed, e, w1.shape)
    # the packed copies: adapted model vs a model PACKED from the adapted raw weights
    packed = NativeModel(od, ad, hidden, act, None, m, "per_block")
    norm = _norm(rs, od, ad, low, high)
    for e in range(m):
        packed.set_weights(e, adapted.get_weights(e))
        adapted.set_norm(e, norm)
        packed.set_norm(e, norm)
    spec = _reward(rs, od, ad)
    n, h = int(rs.choice([5, 37, 130])), 3
    obs0 = torch.from_numpy(rs.randn(m, od).astype(np.float32)).to(dev)
    acts = torch.from_numpy(rs.uniform(low, high, (h, m * n, ad)).astype(np.float32)).to(dev)
    ctx = _lib.Context.get(0)
    try:
        for micro in (0, 2):
            ctx.set_micro(micro)
            outs = []
            for nm in (adapted, packed):
                rets = torch.empty((m, n), dtype=torch.float32, device=dev)
                best = torch.zeros((m,), dtype=torch.int64, device=dev)
                nm.plan_rs(obs0, acts, m, n, h, 0.97, spec, returns_out=rets, best_key=best)
                torch.cuda.synchronize()
                ctx.launch_status()
                outs.append((rets.cpu().numpy(), best.cpu().numpy()))
            assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]), (seed, micro)
    finally:
        ctx.set_micro(1)
    adapted.close()
    packed.close()
