"""``MPCController`` - drop-in for ``learning_to_adapt/policies/mpc_controller.py:6-135``.

Same constructor, ``get_action`` / ``get_actions`` / ``reset`` / ``vectorized`` surface and the
same consumption of NumPy's legacy global RNG (``np.random.uniform`` for random shooting,
``:67-69``; ``np.random.normal`` for CEM, ``:85``) as the reference, so that with a fixed seed
the chosen action matches the reference's bit for bit (modulo declared near-ties).

What changed underneath: the horizon loop (``:116-127`` / ``:92-99``) - ``dynamics_model.predict``
+ ``env.reward`` + return accumulation, h sequential host<->device round trips in the
reference - is ONE launch of the fused HIP rollout kernel (``l2a_plan_rs``), and the arg-max
(``:128-129``) comes back as 8 bytes per env.  With ``torch.distributed`` initialised the
candidates are sharded over the ranks (one process per GPU) and the per-rank best keys are
combined by a single max all-reduce (RCCL over xGMI); CEM all-gathers the returns instead,
because its elite rule (``:101``) needs every candidate's rank.

Extra keyword arguments (all optional, defaults reproduce the reference):
``rng`` (``'numpy'`` = parity mode, host MT19937 as in the reference; ``'device'`` = candidates
drawn on the GPU with torch's Philox generator - statistically equivalent, not bit-identical),
``cem_mode`` (``'reference'`` keeps the reference's three CEM quirks, SURVEY.md section 3.3;
``'fixed'`` = clipped rollouts, true top-k elites, env-consistent row order),
``shard_candidates`` (use torch.distributed when initialised), ``pipeline_chunks`` (parity mode: the
plan step is cut into this many horizon chunks so that the host draws / uploads chunk k + 1 while the
GPU rolls out chunk k - same RNG stream, bit-identical result; 1 = one launch).
"""

import numpy as np
import torch

from .. import _lib
from ..envs.reward_spec import reward_spec_for_env
from ..utils import fast_rng
from ..utils.serializable import Serializable
from .policy import Policy, innermost_env


class MPCController(Policy, Serializable):
    def __init__(
            self,
            name,
            env,
            dynamics_model,
            reward_model=None,
            discount=1,
            use_cem=False,
            n_candidates=1024,
            horizon=10,
            num_cem_iters=8,
            percent_elites=0.1,
            use_reward_model=False,
            alpha=0.1,
            rng="numpy",
            cem_mode="reference",
            shard_candidates=True,
            pipeline_chunks=5,
    ):
        self.dynamics_model = dynamics_model
        self.reward_model = reward_model
        self.discount = discount
        self.n_candidates = n_candidates
        self.horizon = horizon
        self.use_cem = use_cem
        self.num_cem_iters = num_cem_iters
        self.percent_elites = percent_elites
        self.env = env
        self.use_reward_model = use_reward_model
        self.alpha = alpha
        assert rng in ("numpy", "device")
        assert cem_mode in ("reference", "fixed")
        self.rng = rng
        self.cem_mode = cem_mode
        self.shard_candidates = shard_candidates
        self.pipeline_chunks = int(pipeline_chunks)

        self.unwrapped_env = innermost_env(env)

        # make sure that env has reward function (reference :39)
        assert hasattr(self.unwrapped_env, 'reward'), "env must have a reward function"

        Serializable.quick_init(self, locals())
        super(MPCController, self).__init__(env=env)

        self._reward_spec = None if use_reward_model else reward_spec_for_env(env)
        self._bufs = {}
        self.last_plan = None       # diagnostics of the latest fused plan (returns, keys, ...)

    @property
    def vectorized(self):
        return True

    # ------------------------------------------------------------------ reference API
    def get_action(self, observation):
        if observation.ndim == 1:
            observation = observation[None]
        if self.use_cem:
            action = self.get_cem_action(observation)
        else:
            action = self.get_rs_action(observation)
        return action, dict()

    def get_actions(self, observations):
        if self.use_cem:
            actions = self.get_cem_action(observations)
        else:
            actions = self.get_rs_action(observations)
        return actions, dict()

    _fast_uniform = None        # class-wide: does the vectorised draw reproduce np.random.uniform bit for bit here?

    def get_random_action(self, n):
        """``np.random.uniform(low, high, (n, act_dim))`` (reference ``:67-69``) - same values, same
        consumption of the global MT19937 stream.  NumPy's legacy ``uniform`` with array bounds walks a
        broadcast iterator per element (measured 2.6x the cost of the raw doubles); ``low + (high - low) *
        random_sample()`` is the same arithmetic (``random_uniform``: ``lower + range * next_double``, two
        roundings) vectorised.  Verified once per process against the real call on a saved RNG state; if
        the platform's NumPy ever differed, the reference call is used."""
        low, high = self.action_space.low, self.action_space.high
        shape = (n,) + low.shape
        cls = MPCController
        if cls._fast_uniform is None:
            state = np.random.get_state()
            want = np.random.uniform(low=low, high=high, size=(257,) + low.shape)
            np.random.set_state(state)
            got = np.random.random_sample((257,) + low.shape)
            got *= (high - low)
            got += low
            np.random.set_state(state)
            cls._fast_uniform = bool(np.array_equal(want, got))
        if not cls._fast_uniform:
            return np.random.uniform(low=low, high=high, size=shape)
        u = fast_rng.random_sample(shape)            # same stream, vectorised generator (utils/fast_rng.py)
        # full-size (contiguous) scale / offset arrays: broadcasting a length-act_dim vector over the last
        # axis makes NumPy run act_dim-element inner loops, slower than the draw itself
        key = ("uniform_affine", shape)
        aff = self._bufs.get(key)
        if aff is None:
            aff = (np.ascontiguousarray(np.broadcast_to(high - low, shape)),
                   np.ascontiguousarray(np.broadcast_to(low, shape)))
            self._bufs[key] = aff
        u *= aff[0]
        u += aff[1]
        return u

    def get_params_internal(self, **tags):
        return []

    def reset(self, dones=None):
        pass

    # ------------------------------------------------------------------ sharding helpers
    def _dist(self):
        """(rank, world) when candidates are sharded over torch.distributed ranks."""
        if self.shard_candidates and torch.distributed.is_available() and torch.distributed.is_initialized():
            return torch.distributed.get_rank(), torch.distributed.get_world_size()
        return 0, 1

    @staticmethod
    def _shard_range(n, rank, world):
        return (rank * n) // world, ((rank + 1) * n) // world

    def _fusable(self):
        return (self._reward_spec is not None) and hasattr(self.dynamics_model, "planner_model")

    def _buf(self, key, shape, dtype, device):
        t = self._bufs.get(key)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype or t.device != device:
            t = torch.empty(shape, dtype=dtype, device=device)
            self._bufs[key] = t
        return t

    # ------------------------------------------------------------------ fused rollout of a candidate batch
    def _device(self):
        return self.dynamics_model.planner_model().device

    def _upload(self, actions_local):
        """Host ``[h, m * n_local, act_dim]`` (any float dtype) -> fp32 device tensor: cast straight into a
        pinned staging buffer, one asynchronous H2D copy per plan step (the caller reads the plan's result
        back before it can call again, so the staging buffer is never overwritten in flight)."""
        dev = self._device()
        shape = tuple(actions_local.shape)
        pin = self._bufs.get("a_pin")
        if pin is None or tuple(pin.shape) != shape:
            pin = torch.empty(shape, dtype=torch.float32, pin_memory=True)
            self._bufs["a_pin"] = pin
        np.copyto(pin.numpy(), actions_local, casting="same_kind")
        a_dev = self._buf("a_up", shape, torch.float32, dev)
        a_dev.copy_(pin, non_blocking=True)
        return a_dev

    def _check_status(self):
        """After a device->host read-back (= stream sync): raise if a launch flagged a problem."""
        self.dynamics_model.planner_model().ctx.launch_status()

    def _upload_obs(self, observations):
        """``[m, obs_dim]`` host observations -> the fp32 device buffer the rollout reads."""
        native = self.dynamics_model.planner_model()
        obs0 = self._buf("obs0", (len(observations), native.obs_dim), torch.float32, native.device)
        obs0.copy_(torch.from_numpy(np.ascontiguousarray(observations, dtype=np.float32)), non_blocking=False)
        return obs0

    def _rollout(self, observations, actions_local, n_local, cand_offset, want_returns, obs_dev=None):
        """Launch the fused kernel on this rank's shard.

        ``actions_local``: fp32 CUDA tensor ``[h, m * n_local, act_dim]`` (row = env * n_local + j).
        Returns ``(best_key int64 CUDA [m], returns fp32 CUDA [m, n_local] or None)``.
        """
        native = self.dynamics_model.planner_model()
        m = len(observations)
        blocks = self.dynamics_model.planner_blocks(m)
        if blocks != 1 and blocks != m:
            raise _lib.L2AError("the dynamics model holds %d adapted weight sets but %d observations "
                                "were passed" % (blocks, m))
        dev = native.device
        obs0 = obs_dev if obs_dev is not None else self._upload_obs(observations)
        best = self._buf("best", (m,), torch.int64, dev)
        rets = self._buf("rets", (m, n_local), torch.float32, dev) if want_returns else None
        native.plan_rs(obs0, actions_local, m, n_local, self.horizon, self.discount, self._reward_spec,
                       cand_offset=cand_offset, returns_out=rets, best_key=best)
        return best, rets

    # ------------------------------------------------------------------ random shooting (reference :108-129)
    def get_rs_action(self, observations):
        n = self.n_candidates
        m = len(observations)
        h = self.horizon

        if not self._fusable():
            return self._get_rs_action_unfused(observations)

        rank, world = self._dist()
        lo, hi = self._shard_range(n, rank, world)
        n_local = hi - lo
        act_dim = self.action_space.shape[0]

        pipelined = self.rng == "numpy" and self._can_pipeline(h, n_local)
        if pipelined:
            best, cand_a = self._plan_pipelined(observations, n, m, h, lo, hi, world)
        elif self.rng == "numpy":
            # identical draw and layout to the reference (:114): [h, n*m, act_dim], row = i*n + j
            a = self.get_random_action(h * n * m).reshape((h, n * m, -1))
            cand_a = a[0].reshape((m, n, -1))
            a_loc = a.reshape(h, m, n, act_dim)[:, :, lo:hi, :]
            a_dev = self._upload(a_loc if world > 1 else a).view(h, m * n_local, act_dim) if n_local > 0 else \
                self._upload(np.zeros((h, 0, act_dim), dtype=np.float32))
        else:
            dev = self._device()
            a_dev = self._buf("a_dev", (h, m * n_local, act_dim), torch.float32, dev)
            gen = self._bufs.get("rs_gen")
            if gen is None or gen.device != dev:
                # every rank must draw DIFFERENT candidates for its shard
                gen = torch.Generator(device=dev)
                gen.manual_seed((int(torch.initial_seed()) + 7919 * rank) & 0x7FFFFFFF)
                self._bufs["rs_gen"] = gen
            lo_np, hi_np = np.asarray(self.action_space.low), np.asarray(self.action_space.high)
            if np.all(lo_np == lo_np[0]) and np.all(hi_np == hi_np[0]):
                a_dev.uniform_(float(lo_np[0]), float(hi_np[0]), generator=gen)     # one launch (the usual +-bound box)
            else:
                low = torch.as_tensor(lo_np, dtype=torch.float32, device=dev)
                high = torch.as_tensor(hi_np, dtype=torch.float32, device=dev)
                a_dev.uniform_(0.0, 1.0, generator=gen)
                a_dev.mul_(high - low).add_(low)
            cand_a = None

        if pipelined:
            pass                            # already planned, chunk by chunk
        elif n_local > 0:
            best, rets = self._rollout(observations, a_dev, n_local, lo, want_returns=False)
        else:   # more ranks than candidates: this rank contributes the neutral key
            best = torch.zeros((m,), dtype=torch.int64, device=a_dev.device)
        if world > 1:
            torch.distributed.all_reduce(best, op=torch.distributed.ReduceOp.MAX)
        keys = best.cpu().numpy()
        self._check_status()
        idx = np.empty(m, dtype=np.int64)
        best_ret = np.empty(m, dtype=np.float32)
        for i in range(m):
            best_ret[i], idx[i] = _lib.key_decode(keys[i])
        self.last_plan = dict(best_index=idx, best_return=best_ret, n_local=n_local, shard=(lo, hi))

        if cand_a is not None:
            return cand_a[range(m), idx]
        # device RNG: the winning first action lives on the rank that owns the candidate
        if world == 1:
            rows = torch.from_numpy(np.arange(m, dtype=np.int64) * n_local + (idx - lo)).to(a_dev.device)
            return a_dev[0].index_select(0, rows).cpu().numpy().astype(np.float64)
        first = a_dev[0].reshape(m, n_local, act_dim)
        gidx = torch.from_numpy(idx).to(a_dev.device)
        own = ((gidx >= lo) & (gidx < hi)).to(torch.float32).unsqueeze(1)
        loc = (gidx - lo).clamp(0, max(n_local - 1, 0))
        out = first[torch.arange(m, device=a_dev.device), loc] * own if n_local > 0 else \
            torch.zeros((m, act_dim), dtype=torch.float32, device=a_dev.device)
        torch.distributed.all_reduce(out, op=torch.distributed.ReduceOp.SUM)
        return out.cpu().numpy().astype(np.float64)

    # ------------------------------------------------------------------ parity mode, pipelined over the horizon
    def _can_pipeline(self, h, n_local):
        return (self.pipeline_chunks > 1 and h >= 6 and n_local > 0
                and hasattr(self.dynamics_model.planner_model(), "plan_rs_chunk"))

    def _plan_pipelined(self, observations, n, m, h, lo, hi, world):
        """The reference draws its ``h*n*m`` candidate rows horizon-major (``:114``), so the first rows of the
        stream are the first horizon steps: draw chunk k + 1 on the host while the GPU rolls out chunk k
        (``l2a_plan_rs_chunk`` carries per-candidate state and returns between launches).  Same RNG
        consumption, bit-identical returns and arg-max as the single launch."""
        native = self.dynamics_model.planner_model()
        dev = native.device
        blocks = self.dynamics_model.planner_blocks(m)
        if blocks != 1 and blocks != m:
            raise _lib.L2AError("the dynamics model holds %d adapted weight sets but %d observations "
                                "were passed" % (blocks, m))
        act_dim = self.action_space.shape[0]
        n_local = hi - lo
        K = max(2, min(self.pipeline_chunks, h // 2))      # at least two horizon steps per launch
        bounds = [(h * c) // K for c in range(K + 1)]
        obs0 = self._upload_obs(observations)
        best = self._buf("best", (m,), torch.int64, dev)
        cand_a = None
        for c in range(K):
            t0, hc = bounds[c], bounds[c + 1] - bounds[c]
            a = self.get_random_action(hc * n * m).reshape((hc, n * m, -1))
            if c == 0:
                cand_a = a[0].reshape((m, n, -1))
            src = a if world == 1 else a.reshape(hc, m, n, act_dim)[:, :, lo:hi, :]
            shape = (hc, m * n_local, act_dim)
            pin = self._bufs.get(("pipe_pin", c))
            if pin is None or tuple(pin.shape) != shape:
                pin = torch.empty(shape, dtype=torch.float32, pin_memory=True)
                self._bufs[("pipe_pin", c)] = pin
            np.copyto(pin.numpy().reshape(src.shape), src, casting="same_kind")
            a_dev = self._buf(("pipe_dev", c), shape, torch.float32, dev)
            a_dev.copy_(pin, non_blocking=True)
            self._launch_chunk(native, c, c == K - 1, obs0, a_dev, m, n_local, hc, t0, lo, best)
        return best, cand_a

    def _launch_chunk(self, native, c, last, obs0, a_dev, m, n_local, hc, t0, lo, best):
        """Chunk ``c`` of a pipelined plan: state and returns ping-pong between two buffer pairs."""
        dev = native.device
        rets = [self._buf("pipe_ret%d" % i, (m, n_local), torch.float32, dev) for i in (0, 1)]
        state = [self._buf("pipe_state%d" % i, (m * n_local, native.obs_dim), torch.float32, dev) for i in (0, 1)]
        native.plan_rs_chunk(obs0 if c == 0 else state[(c + 1) % 2], c > 0, a_dev, m, n_local, hc, t0,
                             self.discount, self._reward_spec, cand_offset=lo,
                             returns_in=rets[(c + 1) % 2] if c > 0 else None, returns_out=rets[c % 2],
                             state_out=None if last else state[c % 2], best_key=best if last else None)

    def _get_rs_action_unfused(self, observations):
        """No closed-form reward available (custom env reward or ``use_reward_model``): keep the
        reference's loop shape; the MLP still runs on the GPU through ``dynamics_model.predict``."""
        n = self.n_candidates
        m = len(observations)
        h = self.horizon
        returns = np.zeros((n * m,))
        a = self.get_random_action(h * n * m).reshape((h, n * m, -1))
        cand_a = a[0].reshape((m, n, -1))
        observation = np.repeat(observations, n, axis=0)
        for t in range(h):
            next_observation = self.dynamics_model.predict(observation, a[t])
            if self.use_reward_model:
                assert self.reward_model is not None
                rewards = self.reward_model.predict(observation, a[t], next_observation)
            else:
                rewards = self.unwrapped_env.reward(observation, a[t], next_observation)
            returns += self.discount ** t * rewards
            observation = next_observation
        returns = returns.reshape(m, n)
        return cand_a[range(m), np.argmax(returns, axis=1)]

    def _cem_iteration(self, observations, mean, std, num_elites, clip_low, clip_high, lo, hi, world):
        """One CEM iteration (reference ``:85-104``): sample, roll out on the GPU, refit mean/std.
        Consumes ``n * m * h * act_dim`` normals of the global NumPy stream."""
        n = self.n_candidates
        m = len(observations)
        h = self.horizon
        act_dim = self.action_space.shape[0]
        n_local = hi - lo
        reference = (self.cem_mode == "reference")
        if self._can_pipeline_cem(m, world, n):
            a, a_stacked, returns = self._cem_rollout_pipelined(observations, mean, std, clip_low, clip_high,
                                                                reference)
            cand_a = a.reshape((n, h, act_dim))[:, 0, :].reshape((1, n, act_dim)) if reference else \
                a_stacked.reshape((n, h, act_dim))[:, 0, :].reshape((1, n, act_dim))
            return self._cem_refit(mean, a_stacked, returns, num_elites, reference) + (returns, cand_a)
        z = np.random.normal(size=(n, m, h * act_dim))
        a = mean + z * std
        a_stacked = np.clip(a, clip_low, clip_high)
        if reference:
            # reference quirks: rollouts use the UNCLIPPED samples, and the flat row order is
            # candidate-major (row = j*m + i) while observations are env-major (row // n).
            seq = np.transpose(a.reshape((n * m, h, act_dim)), (1, 0, 2))          # [h, n*m, act]
        else:
            seq = np.transpose(a_stacked.transpose(1, 0, 2).reshape((m * n, h, act_dim)), (1, 0, 2))
        cand_a = seq[0].reshape((m, n, -1))
        seq_loc = seq.reshape(h, m, n, act_dim)[:, :, lo:hi, :].astype(np.float32)
        a_dev = self._upload(seq_loc.reshape(h, m * n_local, act_dim))
        _, rets = self._rollout(observations, a_dev, n_local, lo, want_returns=True)
        if world > 1:
            if any(self._shard_range(n, r, world)[1] - self._shard_range(n, r, world)[0] != n_local
                   for r in range(world)):
                raise _lib.L2AError("CEM sharding needs n_candidates divisible by the world size")
            parts = [torch.empty_like(rets) for _ in range(world)]
            torch.distributed.all_gather(parts, rets)
            rets = torch.cat(parts, dim=1)
        returns = rets.cpu().numpy().astype(np.float64).reshape(m, n)
        self._check_status()
        mean, std = self._cem_refit(mean, a_stacked, returns, num_elites, reference)
        return mean, std, returns, cand_a

    def _cem_refit(self, mean, a_stacked, returns, num_elites, reference):
        """Elite statistics of one iteration (reference ``:101-104``)."""
        m = returns.shape[0]
        if reference:
            elites_idx = ((-returns).argsort(axis=-1) < num_elites).T                # :101
            elites = a_stacked[elites_idx]
            mean = mean * self.alpha + (1 - self.alpha) * np.mean(elites, axis=0)
            std = np.std(elites, axis=0)
        else:
            order = np.argsort(-returns, axis=1)[:, :num_elites]                      # [m, k]
            elites = np.stack([a_stacked[order[i], i] for i in range(m)], axis=0)     # [m, k, h*act]
            mean = mean * self.alpha + (1 - self.alpha) * np.mean(elites, axis=1)
            std = np.std(elites, axis=1)
        return mean, std

    def _can_pipeline_cem(self, m, world, n):
        """Single env, single rank: rows are candidates, the stream is candidate-major - chunks of candidates can be
        drawn while earlier chunks roll out."""
        return (self.pipeline_chunks > 1 and m == 1 and world == 1 and n >= 64 * self.pipeline_chunks
                and hasattr(self.dynamics_model.planner_model(), "plan_rs_chunk"))

    def _cem_rollout_pipelined(self, observations, mean, std, clip_low, clip_high, reference):
        """One CEM iteration's sampling + rollout for m == 1, pipelined over candidate chunks: ``np.random.normal``
        consumes its stream candidate-major (``:85``; chunked draws continue the legacy generator exactly, cached
        Gaussian included), so chunk k + 1 is drawn and clipped on the host while the GPU rolls out chunk k into
        its slice of the returns.  Returns ``(a [n, 1, D], a_stacked, returns [1, n] float64)``."""
        native = self.dynamics_model.planner_model()
        dev = native.device
        n, h = self.n_candidates, self.horizon
        act_dim = self.action_space.shape[0]
        D = h * act_dim
        K = self.pipeline_chunks
        bounds = [(n * c) // K for c in range(K + 1)]
        obs0 = self._upload_obs(observations)
        rets = self._buf("rets", (1, n), torch.float32, dev)
        a = np.empty((n, 1, D))
        a_stacked = np.empty((n, 1, D))
        for c in range(K):
            j0, j1 = bounds[c], bounds[c + 1]
            nc = j1 - j0
            z = np.random.normal(size=(nc, 1, D))
            a[j0:j1] = mean + z * std
            np.clip(a[j0:j1], clip_low, clip_high, out=a_stacked[j0:j1])
            src = (a if reference else a_stacked)[j0:j1].reshape(nc, h, act_dim)
            shape = (h, nc, act_dim)
            pin = self._bufs.get(("cem_pin", c))
            if pin is None or tuple(pin.shape) != shape:
                pin = torch.empty(shape, dtype=torch.float32, pin_memory=True)
                self._bufs[("cem_pin", c)] = pin
            np.copyto(pin.numpy(), np.transpose(src, (1, 0, 2)), casting="same_kind")
            a_dev = self._buf(("cem_dev", c), shape, torch.float32, dev)
            a_dev.copy_(pin, non_blocking=True)
            native.plan_rs(obs0, a_dev, 1, nc, h, self.discount, self._reward_spec, cand_offset=j0,
                           returns_out=rets[0, j0:j1])
        returns = rets.cpu().numpy().astype(np.float64).reshape(1, n)
        self._check_status()
        return a, a_stacked, returns

    # ------------------------------------------------------------------ CEM (reference :71-106)
    def _cem_normal_device(self, shape, device):
        """Standard normals for on-device CEM.  Every rank must draw the SAME numbers (each one
        keeps all samples and rolls out only its shard), hence a private generator with a fixed
        seed sequence instead of torch's global one."""
        gen = self._bufs.get("cem_gen")
        if gen is None or gen.device != device:
            gen = torch.Generator(device=device)
            gen.manual_seed(int(torch.initial_seed()) & 0x7FFFFFFF)
            self._bufs["cem_gen"] = gen
        return torch.randn(shape, generator=gen, device=device, dtype=torch.float32)

    def get_cem_action_device(self, observations):
        """CEM with sampling, clipping, elite selection and refit on the GPU (SURVEY.md section 8(f)
        rank 2): the five host synchronisations and 5 x n*m*h*act_dim host normals per plan step
        of the reference loop (``:84-104``) disappear; only the chosen action comes back.
        ``cem_mode='reference'`` keeps the reference's semantics on the device - rollouts on the
        UNCLIPPED samples, candidate-major rows (row = j*m + i read as env row // n), the rank-mask
        "elites" of ``:101`` pooled over the envs; ``cem_mode='fixed'`` uses clipped rollouts,
        env-major rows and true top-k elites per env.  Numbers come from torch's Philox generator, so
        both are validated against the host loops with injected normals, not bit-for-bit against NumPy."""
        reference = (self.cem_mode == "reference")
        n, m, h = self.n_candidates, len(observations), self.horizon
        act_dim = self.action_space.shape[0]
        D = h * act_dim
        dev = self._device()
        num_elites = max(int(n * self.percent_elites), 1)
        low = torch.as_tensor(np.concatenate([self.action_space.low] * h), dtype=torch.float32, device=dev)
        high = torch.as_tensor(np.concatenate([self.action_space.high] * h), dtype=torch.float32, device=dev)
        mean = torch.zeros((m, D), dtype=torch.float32, device=dev)
        std = torch.ones((m, D), dtype=torch.float32, device=dev)
        rank, world = self._dist()
        lo, hi = self._shard_range(n, rank, world)
        n_local = hi - lo
        if world > 1 and n % world != 0:
            raise _lib.L2AError("CEM sharding needs n_candidates divisible by the world size")
        rets = None
        cand = None
        obs_dev = self._upload_obs(observations)             # once per plan step, not once per CEM iteration
        for _ in range(self.num_cem_iters):
            z = self._cem_normal_device((n, m, D), dev)
            a = mean + z * std                                                                # [n, m, D]
            a_clip = torch.clamp(a, low, high)
            if reference:
                cand = a.reshape(m, n, D)                  # the reference's reading of the same memory (:92-96)
            else:
                cand = a_clip.permute(1, 0, 2).contiguous()                                       # [m, n, D]
            seq = cand[:, lo:hi, :].reshape(m * n_local, h, act_dim).permute(1, 0, 2).contiguous()
            _, r_loc = self._rollout(observations, seq, n_local, lo, want_returns=True, obs_dev=obs_dev)
            if world > 1:
                parts = [torch.empty_like(r_loc) for _ in range(world)]
                torch.distributed.all_gather(parts, r_loc)
                rets = torch.cat(parts, dim=1)
            else:
                rets = r_loc
            if reference:
                # :101-104: positions of the descending argsort whose VALUE is < num_elites, used as a
                # mask over candidates, pooled over envs; mean / std broadcast back to every env
                mask = (torch.argsort(rets, dim=1, descending=True, stable=True) < num_elites).t()   # [n, m]
                # masked moments instead of `a_clip[mask]`: boolean indexing needs the element count on the
                # host (a sync per iteration); the mask always selects exactly m * num_elites samples
                w = mask.to(torch.float32).unsqueeze(-1)                                      # [n, m, 1]
                cnt = float(m * num_elites)
                e_mean = (a_clip * w).sum(dim=(0, 1)) / cnt                                   # [D]
                e_var = (((a_clip - e_mean) ** 2) * w).sum(dim=(0, 1)) / cnt
                mean = mean * self.alpha + (1 - self.alpha) * e_mean
                std = torch.sqrt(e_var).expand(m, D)
            else:
                top = torch.topk(rets, num_elites, dim=1).indices                             # [m, k]
                elites = torch.gather(cand, 1, top.unsqueeze(-1).expand(m, num_elites, D))    # [m, k, D]
                mean = mean * self.alpha + (1 - self.alpha) * elites.mean(dim=1)
                std = elites.std(dim=1, unbiased=False)
        idx = torch.argmax(rets, dim=1)                                                       # [m]
        first = cand[torch.arange(m, device=dev), idx, :act_dim]
        out = first.cpu().numpy().astype(np.float64)
        self._check_status()
        self.last_plan = dict(best_index=idx.cpu().numpy(), best_return=rets.max(dim=1).values.cpu().numpy(),
                              cem_mean=mean.cpu().numpy(), cem_std=std.cpu().numpy())
        return out

    def get_cem_action(self, observations):
        if not self._fusable():
            raise _lib.L2AError("CEM planning needs a fusable closed-form reward (env.reward_spec)")
        if self.rng == "device":
            return self.get_cem_action_device(observations)
        n = self.n_candidates
        m = len(observations)
        h = self.horizon
        act_dim = self.action_space.shape[0]
        num_elites = max(int(self.n_candidates * self.percent_elites), 1)
        mean = np.zeros((m, h * act_dim))
        std = np.ones((m, h * act_dim))
        clip_low = np.concatenate([self.action_space.low] * h)
        clip_high = np.concatenate([self.action_space.high] * h)

        rank, world = self._dist()
        lo, hi = self._shard_range(n, rank, world)
        trace = []
        cand_a = None
        returns = None

        for it in range(self.num_cem_iters):
            mean, std, returns, cand_a = self._cem_iteration(observations, mean, std, num_elites,
                                                             clip_low, clip_high, lo, hi, world)
            trace.append(dict(mean=np.array(mean), std=np.array(std), returns=returns))

        idx = np.argmax(returns, axis=1)
        self.last_plan = dict(best_index=idx, best_return=returns[range(m), idx], cem_trace=trace)
        return cand_a[range(m), idx]
