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
drawn on the GPU - the library's counter-based Philox stream under ``torch.initial_seed()`` on one GPU, torch's generator
when sharded - statistically equivalent, not bit-identical),
``cem_mode`` (``'reference'`` keeps the reference's three CEM quirks, SURVEY.md section 3.3;
``'fixed'`` = clipped rollouts, true top-k elites, env-consistent row order),
``shard_candidates`` (use torch.distributed when initialised), ``pipeline_chunks`` (parity mode: the
plan step is cut into this many horizon chunks so that the host draws / uploads chunk k + 1 while the
GPU rolls out chunk k - same RNG stream, bit-identical result; 1 = one launch), ``draw_ahead`` (parity
mode: the NEXT controller step's candidates are drawn on a private copy of the generator state while the
GPU runs the current plan and adopted only if the global generator is still in exactly that state -
``policies/draw_ahead.py``; numbers, order and the state left behind are the reference's), ``native_step`` (parity-mode
random shooting on one GPU: the whole step - adopt the block drawn ahead by a C thread, launch, wait, decode, gather - is ONE
C call, ``l2a_controller_step``, ``policies/native_step.py``; a step that finds no valid block falls back to the path above).

A tile-split launch whose exchange partner was not co-resident (another process on the GPU) flags a status
word instead of hanging; the controller then switches the context to the unsplit geometry (bit-identical
results) and relaunches the plan - it never raises mid-rollout for that.
"""

import os

import numpy as np
import torch

from .. import _lib
from ..envs.reward_spec import reward_spec_for_env
from ..utils import fast_rng
from ..utils.serializable import Serializable
from .draw_ahead import DrawAhead
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
            draw_ahead=True,
            native_step=True,
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
        self.draw_ahead = bool(draw_ahead)
        self.native_step = bool(native_step)

        self.unwrapped_env = innermost_env(env)

        # make sure that env has reward function (reference :39)
        assert hasattr(self.unwrapped_env, 'reward'), "env must have a reward function"

        Serializable.quick_init(self, locals())
        super(MPCController, self).__init__(env=env)

        self._reward_spec = None if use_reward_model else reward_spec_for_env(env)
        self._bufs = {}
        self._ahead = None          # DrawAhead chain (parity mode), created on first use
        self._cstep = None          # NativeStep (l2a_controller): the whole parity-mode step in one C call
        self._cstep_no = None       # request key the C controller was found ineligible for
        self._cem_first_chunk = 4   # horizon steps of the first chunk of a pipelined CEM rollout (0: equal chunks)
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

    DIGEST_MASK = 0x7FFFFFFFFFFF          # include/l2a.h: L2A_DIGEST_MASK

    def _rank_digest(self):
        """Fingerprint of what this rank's candidates were drawn from.  Sharded planning lets every rank draw the SAME
        candidate tensor and keep its slice (parity mode), or the same CEM normals (device mode).  Ranks seeded
        differently (the common ``seed + rank``) - or one rank whose generator was consumed by something else between
        two plans (an in-process env reset, a logger) - would combine keys and returns that refer to different
        actions, silently.  Parity mode: the global MT19937 state as this step's draw left it (~1 us); device mode:
        ``torch.initial_seed()``.  Travels with EVERY collective of a plan step (two more words), so a disagreement is
        caught on the step it happens, on all ranks at once."""
        if self.rng == "numpy":
            return fast_rng.global_digest(with_gauss=bool(self.use_cem)) & self.DIGEST_MASK
        return int(torch.initial_seed()) & self.DIGEST_MASK

    def _digest_error(self):
        what = "np.random global state" if self.rng == "numpy" else "torch.initial_seed()"
        return _lib.L2AError("candidate sharding needs identical %s on every rank (seed all ranks alike and keep other "
                             "consumers of the generator off the planning process; the shards themselves are disjoint)"
                             % what)

    def _agree(self, flag, world):
        """Host-side facts every rank must share before a collective whose inputs depend on them (CEM: the all-gather
        of the returns): MAX all-reduce of ``[flag, digest, MASK - digest]``.  Returns ``any rank's flag``; raises - on
        every rank - when the digests differ."""
        if world == 1:
            return bool(flag)
        d = self._rank_digest()
        t = torch.tensor([1 if flag else 0, d, self.DIGEST_MASK - d], dtype=torch.int64, device=self._device())
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        v = t.cpu().numpy()
        if int(v[1]) + int(v[2]) != self.DIGEST_MASK:
            raise self._digest_error()
        return bool(int(v[0]))

    @staticmethod
    def _all_gather(mine, world):
        """``all_gather`` of equally shaped tensors.  RCCL gathers device tensors; gloo implements the collective for host
        tensors only, so a GPU tensor goes through the host there."""
        if mine.is_cuda and torch.distributed.get_backend() == "gloo":
            mine = mine.cpu()
        parts = [torch.empty_like(mine) for _ in range(world)]
        torch.distributed.all_gather(parts, mine)
        return parts

    def _force_unsplit(self):
        """Every rank of a sharded plan switches to the unsplit launch geometry (some rank's launch lost its tile-split
        partner)."""
        self.dynamics_model.planner_model().ctx.force_unsplit()

    def _pack_payload(self, best, m):
        """``[keys (m), launch flag, digest, MASK - digest]`` as an int64 tensor on the planning device - packed ON the
        device behind the launch (``l2a_plan_payload`` reads the status word there), so nothing on the host waits
        between the launch and the collective."""
        dev = best.device
        payload = self._buf("payload", (m + 3,), torch.int64, dev)
        self.dynamics_model.planner_model().plan_payload(best, m, self._rank_digest(), payload)
        return payload

    def _fusable(self):
        return (self._reward_spec is not None) and hasattr(self.dynamics_model, "planner_model")

    def _buf(self, key, shape, dtype, device):
        t = self._bufs.get(key)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype or t.device != device:
            t = torch.empty(shape, dtype=dtype, device=device)
            self._bufs[key] = t
        return t

    def _pinned(self, key, shape, dev=None):
        """fp32 host staging tensor (page-locked when the planner runs on a GPU)."""
        t = self._bufs.get(key)
        if t is None or tuple(t.shape) != tuple(shape):
            dev = dev if dev is not None else self._device()
            t = torch.empty(tuple(shape), dtype=torch.float32, pin_memory=(dev.type == "cuda"))
            self._bufs[key] = t
        return t

    def _host64(self, key, shape):
        a = self._bufs.get(key)
        if a is None or a.shape != tuple(shape):
            a = np.empty(tuple(shape), dtype=np.float64)
            self._bufs[key] = a
        return a

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

    def _to_device(self, pin, key):
        """Staging tensor -> device tensor ``key`` on the current stream (asynchronous)."""
        dev = self._device()
        if dev.type != "cuda":
            return pin
        a_dev = self._buf(key, tuple(pin.shape), torch.float32, dev)
        a_dev.copy_(pin, non_blocking=True)
        return a_dev

    def _to_device_side(self, pin, key, dev):
        """As ``_to_device`` but on a private copy stream (the draw-ahead worker's upload must not queue behind
        the rollout that is running on the main stream).  Returns ``(tensor, None)``: the copy has completed.  Runs on
        the worker thread: it must not touch the dynamics model (``dev`` is captured by the caller)."""
        if dev.type != "cuda":
            return pin, None
        side = self._bufs.get("side_stream")
        if side is None:
            side = torch.cuda.Stream(device=dev)
            self._bufs["side_stream"] = side
        a_dev = self._buf(key, tuple(pin.shape), torch.float32, dev)
        with torch.cuda.stream(side):
            a_dev.copy_(pin, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(side)
        # The worker has a whole plan's duration to spare: it waits for its copy here (GIL released), so that the
        # consumer neither has to make its stream wait for the event (2-45 us on the step's critical path, measured)
        # nor launches behind an unfinished copy.
        ev.synchronize()
        return a_dev, None

    def _sync(self):
        dev = self._device()
        if dev.type == "cuda":
            torch.cuda.current_stream(dev).synchronize()

    def _check_status(self):
        """After a device->host read-back (= stream sync).  ``False``: a launch since the last check was invalid
        (its tile-split partner was not co-resident); the context has been switched to the unsplit launch
        geometry and the caller must relaunch.  Raises only if launches fail with the split already off."""
        return self.dynamics_model.planner_model().ctx.check_or_degrade()

    def _status_flag(self):
        """After a stream sync: was a launch since the last read flagged invalid?  Reads and clears the status word
        WITHOUT switching the geometry (the ranks of a sharded plan decide that together, ``_agree``)."""
        return self.dynamics_model.planner_model().ctx.launch_status_value() != 0

    def _check_blocks(self, m):
        blocks = self.dynamics_model.planner_blocks(m)
        if blocks != 1 and blocks != m:
            raise _lib.L2AError("the dynamics model holds %d adapted weight sets but %d observations "
                                "were passed" % (blocks, m))

    def _upload_obs(self, observations):
        """``[m, obs_dim]`` host observations -> the fp32 device buffer the rollout reads."""
        native = self.dynamics_model.planner_model()
        shape = (len(observations), native.obs_dim)
        obs0 = self._buf("obs0", shape, torch.float32, native.device)
        # through page-locked staging: an asynchronous copy instead of a pageable one that synchronises the stream
        # (every caller reads the plan's result back before it uploads the next observation)
        pin = self._pinned("obs0_pin", shape, native.device)
        np.copyto(pin.numpy(), np.asarray(observations).reshape(shape), casting="same_kind")
        obs0.copy_(pin, non_blocking=True)
        return obs0

    def _rollout(self, observations, actions_local, n_local, cand_offset, want_returns, obs_dev=None):
        """Launch the fused kernel on this rank's shard.

        ``actions_local``: fp32 CUDA tensor ``[h, m * n_local, act_dim]`` (row = env * n_local + j).
        Returns ``(best_key int64 CUDA [m], returns fp32 CUDA [m, n_local] or None)``.
        """
        native = self.dynamics_model.planner_model()
        m = len(observations)
        self._check_blocks(m)
        dev = native.device
        obs0 = obs_dev if obs_dev is not None else self._upload_obs(observations)
        best = self._buf("best", (m,), torch.int64, dev)
        rets = self._buf("rets", (m, n_local), torch.float32, dev) if want_returns else None
        native.plan_rs(obs0, actions_local, m, n_local, self.horizon, self.discount, self._reward_spec,
                       cand_offset=cand_offset, returns_out=rets, best_key=best)
        return best, rets

    def _plan_keys(self, observations, a_dev, n_local, lo, world):
        """Roll out ``a_dev`` and return the arg-max keys: as a NumPy uint64 array when the blocking single-GPU
        launch is available (``l2a_plan_rs_sync``: observations staged in host-mapped memory, keys published to a
        host-mapped mailbox by the last tile - no copies, no stream synchronisation; a launch flagged invalid is
        repeated unsplit right here), else as the device tensor ``_rollout`` fills (sharded plans, recurrent
        model, CPU test harness)."""
        m = len(observations)
        stock = getattr(self._rollout, "__func__", None) is MPCController._rollout      # not overridden / replaced
        native = self.dynamics_model.planner_model() if (stock and world == 1 and n_local > 0) else None
        if native is not None and getattr(native, "sync_max_envs", 0) >= m:
            self._check_blocks(m)
            for _ in range(2):
                keys = native.plan_rs_sync(observations, a_dev, m, n_local, self.horizon, self.discount,
                                           self._reward_spec, cand_offset=lo)
                if keys is not None:
                    return keys.view(np.int64)
            raise _lib.L2AError("rollout launch failed twice")
        if n_local > 0:
            return self._rollout(observations, a_dev, n_local, lo, want_returns=False)[0]
        return torch.zeros((m,), dtype=torch.int64, device=self._device())       # more ranks than candidates

    # ------------------------------------------------------------------ parity-mode draws
    def _draw_rows(self, rows, n, lo, hi, out_f32, rows64=0, out_f64=None):
        """``rows`` rows of the reference's draw (``get_random_action``, ``:67-69``) from the GLOBAL generator:
        candidates ``lo <= row % n < hi`` of every block of n rows as fp32 into ``out_f32`` (this rank's shard, the
        layout ``l2a_plan_rs`` reads), the first ``rows64`` rows as float64 into ``out_f64``.  One threaded pass of
        ``csrc/l2a_rng.c`` when the helper has proven itself on this machine, NumPy otherwise."""
        low, high = self.action_space.low, self.action_space.high
        if fast_rng.available("uniform"):
            st = fast_rng.State.from_global()
            if st is not None:
                st.uniform_rows(rows, low, high, n, lo, hi, out_f32, rows64, out_f64)
                st.to_global()
                return
        a = self.get_random_action(rows)
        act_dim = a.shape[-1]
        if out_f64 is not None and rows64:
            out_f64.reshape(-1, act_dim)[:rows64] = a[:rows64]
        if out_f32 is not None and hi > lo:
            np.copyto(out_f32.reshape(-1, hi - lo, act_dim), a.reshape(-1, n, act_dim)[:, lo:hi, :], casting="same_kind")

    def _use_draw_ahead(self, kind="uniform"):
        return self.draw_ahead and self.rng == "numpy" and fast_rng.available(kind)

    def _ahead_chain(self):
        if self._ahead is None:
            self._ahead = DrawAhead(depth=1)
        return self._ahead

    def _rs_producer(self, n, m, h, lo, hi):
        """Producer of one controller step's candidates for the draw-ahead chain (runs on its worker thread)."""
        low = np.array(self.action_space.low, dtype=np.float64)
        high = np.array(self.action_space.high, dtype=np.float64)
        act_dim = low.shape[0]
        n_local = hi - lo
        dev = self._device()

        def produce(state, slot):
            c64 = self._host64(("ahead_c64", slot), (m * n, act_dim))
            if n_local > 0:
                pin = self._pinned(("ahead_pin", slot), (h, m * n_local, act_dim), dev)
                state.uniform_rows(h * n * m, low, high, n, lo, hi, pin.numpy(), n * m, c64)
                a_dev, ev = self._to_device_side(pin, ("ahead_dev", slot), dev)
            else:
                state.uniform_rows(h * n * m, low, high, n, lo, hi, None, n * m, c64)
                a_dev, ev = None, None
            return dict(a_dev=a_dev, event=ev, cand_a=c64.reshape(m, n, act_dim))
        return produce

    # ------------------------------------------------------------------ the whole step in one C call
    def _native_step_stock(self):
        """The launch path is the product's own (a test harness that replaces it must not be bypassed)."""
        return (getattr(self._rollout, "__func__", None) is MPCController._rollout
                and getattr(self._plan_keys, "__func__", None) is MPCController._plan_keys)

    def _native_step_state(self, native, m):
        """Recurrent planners: ``((c0, h0, c_next, h_next) device pointers, what to keep)``; None here."""
        return None, None

    def _native_step_done(self, keep):
        pass

    @staticmethod
    def _reduce_payload(payload):
        """The sharded C step's one collective through torch.distributed: int64 MAX all-reduce of ``[keys, flag, digest pair]``
        in place (``backend="nccl"`` is RCCL over xGMI; gloo reduces host tensors only, so the words go through the host)."""
        if torch.distributed.get_backend() == "gloo":
            host = payload.cpu()
            torch.distributed.all_reduce(host, op=torch.distributed.ReduceOp.MAX)
            payload.copy_(host)
        else:
            torch.distributed.all_reduce(payload, op=torch.distributed.ReduceOp.MAX)

    @staticmethod
    def _native_comm(native, rank, world):
        """``L2A_NATIVE_COMM=1``: the library's OWN RCCL communicator (``l2a_comm_init``) carries the step's collective
        (``l2a_allreduce_best`` on m + 3 words) instead of torch.distributed; its id travels through torch.distributed once."""
        import ctypes
        ctx = native.ctx
        if getattr(ctx, "native_comm", None) == (rank, world):
            return True
        ids = [None]
        if rank == 0:
            buf = ctypes.create_string_buffer(128)
            ctx.check(native.lib.l2a_comm_unique_id(buf), "l2a_comm_unique_id")
            ids[0] = buf.raw
        torch.distributed.broadcast_object_list(ids, src=0)
        ctx.check(native.lib.l2a_comm_init(ctx.handle, int(rank), int(world), ids[0]), "l2a_comm_init")
        ctx.native_comm = (rank, world)
        return True

    def _native_step_build(self, native, m, key, rank=0, world=1):
        from .native_step import NativeStep
        if self._cstep is not None:
            self._cstep.close()
            self._cstep = None
        if os.environ.get("L2A_NATIVE_STEP", "1") == "0" or not hasattr(native.lib, "l2a_controller_step"):
            return None
        if getattr(native, "sync_max_envs", 0) < m or m * native.obs_dim > 4096 or native.act_dim > 16:
            return None
        device = self.rng == "device"
        if not device and (not (fast_rng.available("uniform") and fast_rng.available("direct")) or fast_rng._global_addr() is None):
            return None
        shard = None
        if world > 1:
            # the sharded step in one C call (parity mode, MLP models): launch, payload, the ONE collective, read-back - the ~80 us
            # of Python glue per step of `_rs_parity_plan` / `_combine_keys` gone.  The collective: torch.distributed behind a
            # callback (default), or the library's own RCCL communicator (L2A_NATIVE_COMM=1).
            # (rng="device": every rank fills its slice of the SAME Philox stream - the plan does not depend on the world size -
            #  and recomputes the winner's action from the stream: no second collective)
            if hasattr(native, "units") or not hasattr(native.lib, "l2a_controller_create_sharded_device"):
                return None
            own = os.environ.get("L2A_NATIVE_COMM", "0") == "1" and torch.distributed.get_backend() != "gloo"
            if own:
                self._native_comm(native, rank, world)
            else:
                # dry run of the collective the callback will issue every step (every rank builds its controller in the same
                # step, so the call is symmetric): a backend that cannot MAX-reduce int64 words on this device keeps the Python path
                try:
                    self._reduce_payload(torch.zeros((m + 3,), dtype=torch.int64, device=native.device))
                except Exception:
                    return None
            shard = (rank, world, None if own else self._reduce_payload)
        st = NativeStep(native, hasattr(native, "units"), m, self.n_candidates, self.horizon, self.action_space.low,
                        self.action_space.high, self.discount, self._reward_spec,
                        device_seed=(int(torch.initial_seed()) if device else None), shard=shard)
        st.key = key
        self._cstep = st
        return st

    def _native_rs_step(self, observations, m):
        """Parity-mode random shooting on one GPU through ``l2a_controller_step`` (``policies/native_step.py``).  Returns the
        actions, or None when the C controller does not apply (sharded plan, test harness, no helper library, a forked child):
        the caller then takes the ordinary path.  A step that finds no valid block of candidates (first call, a foreign draw
        from ``np.random`` since the last step) draws them itself inside the C call - the reference's draw from the global
        generator - and re-arms the chain behind it."""
        self._cstep_missed = False
        if self.use_cem or not self._native_step_stock():
            if self._cstep is not None:     # built earlier, bypassed now (a harness replaced the launch path): no idle C chain
                self._cstep.close()
                self._cstep = None
            return None
        rank, world = self._dist()
        native = self.dynamics_model.planner_model()
        # (device mode: the library's counter-based stream (seed, steps so far) restarts whenever torch's seed VALUE changes, like
        #  the device CEM's.  Calling torch.manual_seed(s) again with the same s does NOT rewind it - build a new controller, or
        #  seed with another value in between, to replay a run.  Since round 6 a sharded MLP plan draws its slices from the SAME
        #  Philox stream (`l2a_controller_create_sharded_device`): device-mode plans no longer depend on the world size; the
        #  recurrent planner's sharded device path still uses torch's generator - ADVICE r5)
        # (parity mode: the address of the global generator's state - the C controller caches it; a generator object that was
        #  replaced, np.random.set_bit_generator, must not leave it reading and writing the old, possibly freed one: ADVICE r5)
        key = (os.getpid(), id(native), native.handle.value, world, m, self.n_candidates, self.horizon, float(self.discount),
               int(torch.initial_seed()) if self.rng == "device" else fast_rng._global_addr())
        st = self._cstep
        if st is None or st.key != key:
            if self._cstep_no == key:
                return None
            st = self._native_step_build(native, m, key, rank, world)
            if st is None:
                self._cstep_no = key
                return None
            self._cstep_no = None
        self._check_blocks(m)
        state, keep = self._native_step_state(native, m)
        if not st.step(observations, torch.cuda.current_stream(native.device).cuda_stream, state):
            self._cstep_missed = True       # (a forked child): this call's fallback plan re-arms the C chain instead of a Python one
            return None
        self._native_step_done(keep)
        lo, hi = self._shard_range(self.n_candidates, rank, world)
        self.last_plan = dict(best_index=st.idx.copy(), best_return=st.ret.copy(), n_local=hi - lo, shard=(lo, hi))
        return st.act.copy()

    def draw_ahead_stats(self):
        """``dict(hits=, misses=)`` of the candidate blocks drawn ahead of time (C controller and Python chain together)."""
        hits = misses = 0
        if self._cstep is not None:
            s = self._cstep.stats()
            hits, misses = hits + s["hits"], misses + s["misses"]
        if self._ahead is not None:
            hits, misses = hits + self._ahead.hits, misses + self._ahead.misses
        return dict(hits=hits, misses=misses)

    # ------------------------------------------------------------------ random shooting (reference :108-129)
    def get_rs_action(self, observations):
        n = self.n_candidates
        m = len(observations)
        h = self.horizon

        if not self._fusable():
            return self._get_rs_action_unfused(observations)

        if self.native_step and (self.draw_ahead or self.rng == "device"):
            out = self._native_rs_step(observations, m)
            if out is not None:
                return out

        rank, world = self._dist()
        lo, hi = self._shard_range(n, rank, world)
        n_local = hi - lo
        act_dim = self.action_space.shape[0]
        dev = self._device()
        a_dev = None

        if self.rng == "numpy":
            best, cand_a, relaunch = self._rs_parity_plan(observations, n, m, h, lo, hi, world)
        else:
            cand_a = None
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

            def relaunch():
                return self._plan_keys(observations, a_dev, n_local, lo, world)
            best = relaunch()

        keys = self._combine_keys(best, relaunch, world)
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

    def _combine_keys(self, best, relaunch, world):
        """Read the arg-max keys back.  Sharded plan: ONE max all-reduce of ``[keys, launch flag, digest pair]`` enqueued
        right behind the launch (no host synchronisation, no status read before the collective), then ONE device-to-host
        copy of ``m + 3`` words.  A set flag means some rank's launch lost its tile-split partner: ALL ranks switch to
        the unsplit geometry (bit-identical results) and repeat their launch and the collective together, so no rank is
        ever out of step and a stale key never decides anything."""
        if isinstance(best, np.ndarray):        # blocking launch: keys already on the host, status already handled
            return best
        if world > 1:
            m = best.numel()
            for attempt in (0, 1):
                payload = self._pack_payload(best, m)
                torch.distributed.all_reduce(payload, op=torch.distributed.ReduceOp.MAX)
                host = payload.cpu().numpy()
                if int(host[m + 1]) + int(host[m + 2]) != self.DIGEST_MASK:
                    raise self._digest_error()
                if int(host[m]) == 0:
                    return host[:m].copy()
                if attempt == 1:
                    raise _lib.L2AError("rollout launch failed twice")
                self._force_unsplit()           # every rank: the reduced flag is the same everywhere
                best = relaunch()
        keys = best.cpu().numpy()
        if self._check_status() is False:
            keys = relaunch().cpu().numpy()
            if self._check_status() is False:
                raise _lib.L2AError("rollout launch failed twice")
        return keys

    def _rs_parity_plan(self, observations, n, m, h, lo, hi, world):
        """Parity mode: candidates from NumPy's global generator exactly as the reference draws them (``:114``:
        ``[h, n*m, act_dim]``, row = i*n + j).  Returns ``(best_key, cand_a float64 [m, n, act], relaunch)``.

        Three ways to the same bits, fastest first: (1) the block the draw-ahead chain prepared while the
        previous plan ran (already in HBM); (2) the draw pipelined over horizon chunks under the rollout;
        (3) one draw, one upload, one launch."""
        n_local = hi - lo
        act_dim = self.action_space.shape[0]
        dev = self._device()
        # the C controller's chain serves the fallback only when the native step was tried in THIS call and could not serve it;
        # a controller that is bypassed for good keeps the Python chain (a silent loss of the draw-ahead otherwise)
        cstep = self._cstep if (self._cstep is not None and self._cstep_no is None and getattr(self, "_cstep_missed", False)) else None
        ahead = self._use_draw_ahead("uniform") and cstep is None      # one chain at a time: the C controller's, if it applies
        sig = ("rs", n, m, h, lo, hi)
        chain = self._ahead_chain() if ahead else None
        blk = chain.take(sig) if ahead else None

        def kick():     # the next controller step's candidates are drawn while this plan runs on the GPU
            if cstep is not None:
                cstep.rearm()
            elif ahead and not chain.active_for(sig):
                chain.start(sig, self._rs_producer(n, m, h, lo, hi), depth=1, words_only=True)

        if blk is not None:
            cand_a, a_dev = blk["cand_a"], blk["a_dev"]
            if blk["event"] is not None:
                torch.cuda.current_stream(dev).wait_event(blk["event"])

            def relaunch():
                return self._plan_keys(observations, a_dev, n_local, lo, world)
            kick()
            best = relaunch()
        elif n_local > 0 and self._can_pipeline(h, n_local):
            best, cand_a = self._plan_pipelined(observations, n, m, h, lo, hi, world)

            def relaunch():
                return self._plan_pipelined(observations, n, m, h, lo, hi, world, redraw=False)[0]
            kick()
        else:
            c64 = self._host64("rs_c64", (m * n, act_dim))
            if n_local > 0:
                pin = self._pinned("rs_pin", (h, m * n_local, act_dim))
                self._draw_rows(h * n * m, n, lo, hi, pin.numpy(), n * m, c64)
                a_dev = self._to_device(pin, "rs_dev")
            else:
                self._draw_rows(h * n * m, n, lo, hi, None, n * m, c64)
                a_dev = None
            cand_a = c64.reshape(m, n, act_dim)

            def relaunch():
                return self._plan_keys(observations, a_dev, n_local, lo, world)
            kick()
            best = relaunch()
        return best, cand_a, relaunch

    # ------------------------------------------------------------------ parity mode, pipelined over the horizon
    def _can_pipeline(self, h, n_local):
        return (self.pipeline_chunks > 1 and h >= 6 and n_local > 0
                and hasattr(self.dynamics_model.planner_model(), "plan_rs_chunk"))

    def _plan_pipelined(self, observations, n, m, h, lo, hi, world, redraw=True):
        """The reference draws its ``h*n*m`` candidate rows horizon-major (``:114``), so the first rows of the
        stream are the first horizon steps: draw chunk k + 1 on the host while the GPU rolls out chunk k
        (``l2a_plan_rs_chunk`` carries per-candidate state and returns between launches).  Same RNG
        consumption, bit-identical returns and arg-max as the single launch.  ``redraw=False`` relaunches the
        chunks already in HBM (after a launch was flagged invalid)."""
        native = self.dynamics_model.planner_model()
        dev = native.device
        self._check_blocks(m)
        act_dim = self.action_space.shape[0]
        n_local = hi - lo
        K = max(2, min(self.pipeline_chunks, h // 2))      # at least two horizon steps per launch
        bounds = [(h * c) // K for c in range(K + 1)]
        obs0 = self._upload_obs(observations)
        best = self._buf("best", (m,), torch.int64, dev)
        c64 = self._host64("pipe_c64", (m * n, act_dim))
        for c in range(K):
            t0, hc = bounds[c], bounds[c + 1] - bounds[c]
            shape = (hc, m * n_local, act_dim)
            if redraw:
                pin = self._pinned(("pipe_pin", c), shape)
                self._draw_rows(hc * n * m, n, lo, hi, pin.numpy(), n * m if c == 0 else 0, c64 if c == 0 else None)
                a_dev = self._to_device(pin, ("pipe_dev", c))
            else:
                a_dev = self._bufs[("pipe_dev", c)]
            self._launch_chunk(native, c, c == K - 1, obs0, a_dev, m, n_local, hc, t0, lo, best)
        return best, c64.reshape(m, n, act_dim)

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

    def _cem_draw(self, n, m, D):
        """The iteration's standard normals ``np.random.normal(size=(n, m, D))`` (``:85``) as a ``[n * m, D]``
        float64 array (flat row g = j * m + i): from the draw-ahead chain when it is valid (the z of upcoming
        iterations - and of the next controller step - do not depend on mean / std, so they are drawn while
        the GPU rolls out), else synchronously from the global generator (threaded helper or NumPy)."""
        sig = ("cem", n, m, D)
        if self._use_draw_ahead("normal"):
            chain = self._ahead_chain()
            z = chain.take(sig)
            if z is None:
                z = self._host64("cem_z_sync", (n * m, D))
                st = fast_rng.State.from_global()
                if st is not None:
                    st.standard_normal(n * m * D, out=z.reshape(-1))
                    st.to_global()
                else:
                    z[...] = np.random.normal(size=(n * m, D))

                def produce(state, slot):
                    buf = self._host64(("cem_z", slot), (n * m, D))
                    state.standard_normal(n * m * D, out=buf.reshape(-1))
                    return buf
                chain.start(sig, produce, depth=max(1, int(self.num_cem_iters)))
            return z
        return fast_rng.standard_normal((n * m, D))

    def _cem_iteration(self, observations, mean, std, num_elites, clip_low, clip_high, lo, hi, world):
        """One CEM iteration (reference ``:85-104``): sample, roll out on the GPU, refit mean/std.
        Consumes ``n * m * h * act_dim`` normals of the global NumPy stream."""
        n = self.n_candidates
        m = len(observations)
        h = self.horizon
        act_dim = self.action_space.shape[0]
        D = h * act_dim
        n_local = hi - lo
        reference = (self.cem_mode == "reference")
        z = self._cem_draw(n, m, D)                                         # [n*m, D], row g = j*m + i
        mean2 = np.ascontiguousarray(np.broadcast_to(mean, (m, D)), dtype=np.float64)
        std2 = np.ascontiguousarray(np.broadcast_to(std, (m, D)), dtype=np.float64)
        a = self._host64("cem_a", (n * m, D))                               # :86 (unclipped)
        a_st = self._host64("cem_clip", (n * m, D))                         # :87
        low, high = self.action_space.low, self.action_space.high
        fused = fast_rng.available("double")        # helper library present and trusted
        if fused and n_local > 0 and self._can_pipeline_cem(m, world, n):
            returns = self._cem_rollout_pipelined(observations, z, a, a_st, mean2, std2, reference)
        else:
            if fused and n_local > 0 and D <= 4096:
                pin = self._pinned("cem_pin_all", (h, m * n_local, act_dim))
                fast_rng.cem_samples(z, 0, h, act_dim, mean2, std2, low, high, a, a_st, pin.numpy(), n, lo, hi,
                                     env_major=not reference, use_clipped=not reference)
                a_dev = self._to_device(pin, "cem_dev_all")
            else:
                a3 = mean2 + z.reshape(n, m, D) * std2
                a[...] = a3.reshape(n * m, D)
                a_st[...] = np.clip(a3, clip_low, clip_high).reshape(n * m, D)
                if reference:
                    # reference quirks: rollouts use the UNCLIPPED samples, and the flat row order is
                    # candidate-major (row = j*m + i) while observations are env-major (row // n).
                    seq = np.transpose(a.reshape((n * m, h, act_dim)), (1, 0, 2))          # [h, n*m, act]
                else:
                    seq = np.transpose(a_st.reshape(n, m, D).transpose(1, 0, 2).reshape((m * n, h, act_dim)), (1, 0, 2))
                seq_loc = seq.reshape(h, m, n, act_dim)[:, :, lo:hi, :].astype(np.float32)
                a_dev = self._upload(seq_loc.reshape(h, m * n_local, act_dim))
            if n_local > 0:
                _, rets = self._rollout(observations, a_dev, n_local, lo, want_returns=True)
            else:               # more ranks than candidates: an empty shard still joins the collective
                rets = torch.zeros((m, 0), dtype=torch.float32, device=self._device())
            if world > 1:
                # every rank must gather VALID returns drawn from the SAME normals: the ranks agree on (any launch
                # flagged?, generator digests equal?) first - one more 3-word collective per iteration beside a 2.8 ms
                # rollout - and repeat the rollout unsplit TOGETHER when any of them lost a tile-split partner
                bad = False
                if n_local > 0:
                    self._sync()
                    bad = self._status_flag()
                if self._agree(bad, world):
                    self._force_unsplit()
                    bad = False
                    if n_local > 0:
                        _, rets = self._rollout(observations, a_dev, n_local, lo, want_returns=True)
                        self._sync()
                        bad = self._status_flag()
                    if self._agree(bad, world):
                        raise _lib.L2AError("rollout launch failed twice")
                # all-gather needs equal shapes: shards are padded to the widest one (they differ by at most one
                # candidate) and cut back to their own width afterwards
                widths = [self._shard_range(n, r, world)[1] - self._shard_range(n, r, world)[0] for r in range(world)]
                wmax = max(widths)
                mine = rets if n_local == wmax else torch.cat(
                    [rets, torch.zeros((m, wmax - n_local), dtype=rets.dtype, device=rets.device)], dim=1)
                parts = self._all_gather(mine.contiguous(), world)
                rets = torch.cat([part[:, :w] for part, w in zip(parts, widths)], dim=1)
            returns = rets.cpu().numpy().astype(np.float64).reshape(m, n)
            if world == 1 and self._check_status() is False:
                _, rets = self._rollout(observations, a_dev, n_local, lo, want_returns=True)
                returns = rets.cpu().numpy().astype(np.float64).reshape(m, n)
                if self._check_status() is False:
                    raise _lib.L2AError("rollout launch failed twice")
        if reference:
            cand_a = a[:, :act_dim].reshape(m, n, act_dim)                  # = seq[0].reshape(m, n, -1)
        else:
            cand_a = a_st.reshape(n, m, D)[:, :, :act_dim].transpose(1, 0, 2)
        mean, std = self._cem_refit(mean, a_st.reshape(n, m, D), returns, num_elites, reference)
        return mean, std, returns, np.array(cand_a)

    def _cem_refit(self, mean, a_stacked, returns, num_elites, reference):
        """Elite statistics of one iteration (reference ``:101-104``)."""
        m = returns.shape[0]
        if reference:
            elites_idx = ((-returns).argsort(axis=-1) < num_elites).T                # :101
            # np.mean / np.std of `a_stacked[elites_idx]` (:102-104) without the gather and the temporaries - the same row-after-row
            # float64 sums, verified against NumPy on this machine (csrc/l2a_rng.c: l2a_cem_elite_stats); NumPy itself otherwise
            n_, m_, D_ = a_stacked.shape
            st = fast_rng.elite_stats(a_stacked.reshape(n_ * m_, D_), elites_idx.reshape(-1)) if a_stacked.flags.c_contiguous else None
            if st is not None:
                mu, std = st
            else:
                elites = a_stacked[elites_idx]
                mu, std = np.mean(elites, axis=0), np.std(elites, axis=0)
            mean = mean * self.alpha + (1 - self.alpha) * mu
        else:
            order = np.argsort(-returns, axis=1)[:, :num_elites]                      # [m, k]
            elites = np.stack([a_stacked[order[i], i] for i in range(m)], axis=0)     # [m, k, h*act]
            mean = mean * self.alpha + (1 - self.alpha) * np.mean(elites, axis=1)
            std = np.std(elites, axis=1)
        return mean, std

    def _cem_chunks(self, h):
        """Horizon chunks of a pipelined CEM iteration: two or three (measured on config 5: 1 chunk 16.8 ms per plan
        step, 2 or 3 chunks 16.35, 5 chunks 16.5; cutting the candidates instead: 18.1)."""
        return max(1, min(self.pipeline_chunks, 3, h // 4))

    def _can_pipeline_cem(self, m, world, n):
        """Single env, single rank: the host prepares horizon chunk k + 1 (``a = mean + z * std``, clip, cast,
        transpose of its steps) while the GPU rolls out chunk k."""
        return (m == 1 and world == 1 and self._cem_chunks(self.horizon) > 1
                and hasattr(self.dynamics_model.planner_model(), "plan_rs_chunk"))

    def _cem_rollout_pipelined(self, observations, z, a, a_st, mean2, std2, reference, relaunch=False):
        """One CEM iteration's rollout for m == 1, pipelined ALONG THE HORIZON: ``a = mean + z * std``, the clip, the
        fp32 cast and the ``[h, rows, act]`` transposition are elementwise in the horizon step, so the steps of chunk
        k + 1 are prepared on the host (``l2a_cem_samples_steps``, threaded) and uploaded while the GPU rolls out chunk k
        of ALL candidates (``l2a_plan_rs_chunk`` hands state and returns from launch to launch, bit-identical to one
        launch).  Every launch keeps the full candidate count - the chip stays as full as with a single launch
        (cutting the candidates instead made two half-empty launches of one full one: 2 x 1.52 ms against 2.77 ms for
        config 5's 4000).  Fills ``a`` / ``a_st`` ``[n, D]``; returns ``returns [1, n]`` float64."""
        native = self.dynamics_model.planner_model()
        dev = native.device
        n, h = self.n_candidates, self.horizon
        act_dim = self.action_space.shape[0]
        self._check_blocks(1)
        K = self._cem_chunks(h)
        # a short first chunk - its samples are the only ones the GPU has to wait for - but long enough for the next
        # chunk's samples to be ready when it ends (host: ~15-30 us per step of 4000 candidates, GPU: ~90 us)
        h0 = max(2, min(self._cem_first_chunk, h // K)) if self._cem_first_chunk else h // K
        bounds = [0] + [h0 + ((h - h0) * c) // (K - 1) for c in range(K)]
        obs0 = self._upload_obs(observations)
        low, high = self.action_space.low, self.action_space.high
        pin = self._pinned("cem_pin_h", (h, n, act_dim))
        a_dev = self._buf("cem_dev_h", (h, n, act_dim), torch.float32, dev)
        rets = [self._buf("cem_ret%d" % i, (1, n), torch.float32, dev) for i in (0, 1)]
        state = [self._buf("cem_state%d" % i, (n, native.obs_dim), torch.float32, dev) for i in (0, 1)]
        for c in range(K):
            t0, t1 = bounds[c], bounds[c + 1]
            if not relaunch:
                fast_rng.cem_samples(z, 0, h, act_dim, mean2, std2, low, high, a, a_st, pin.numpy(), n, 0, n,
                                     env_major=False, use_clipped=not reference, steps=(t0, t1))
                side = self._bufs.get("side_stream")
                if side is None:
                    side = self._bufs["side_stream"] = torch.cuda.Stream(device=dev)
                with torch.cuda.stream(side):           # the upload of chunk k + 1 overlaps the rollout of chunk k
                    a_dev[t0:t1].copy_(pin[t0:t1], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(side)
                torch.cuda.current_stream(dev).wait_event(ev)
            last = (c == K - 1)
            src, dst = (c + 1) % 2, c % 2
            native.plan_rs_chunk(obs0 if c == 0 else state[src], c > 0, a_dev[t0:t1], 1, n, t1 - t0, t0, self.discount,
                                 self._reward_spec, cand_offset=0, returns_in=rets[src] if c > 0 else None,
                                 returns_out=rets[dst], state_out=None if last else state[dst])
        returns = rets[(K - 1) % 2].cpu().numpy().astype(np.float64).reshape(1, n)
        if self._check_status() is False:
            if relaunch:
                raise _lib.L2AError("rollout launch failed twice")
            return self._cem_rollout_pipelined(observations, z, a, a_st, mean2, std2, reference, relaunch=True)
        return returns

    # ------------------------------------------------------------------ CEM (reference :71-106)
    def _cem_normal_device(self, shape, device):
        """Hook for tests: return the iteration's standard normals ``[n, m, D]`` as a CUDA fp32 tensor to inject them;
        ``None`` (the default) lets ``l2a_cem_sample`` draw them on the device (Philox4x32-10 + Box-Muller - every
        rank generates the same numbers: each one keeps all samples and rolls out only its shard)."""
        return None

    def get_cem_action_device(self, observations, retry=False):
        """CEM with sampling, clipping, elite selection and refit on the GPU: the five
        host synchronisations and 5 x n*m*h*act_dim host normals per plan step of the reference loop (``:84-104``)
        disappear; only the chosen action comes back.  Per iteration three hand-written launches around the fused
        rollout (``csrc/l2a_cem.hip``: ``l2a_cem_sample`` - draw, ``a = mean + z * std``, clip, the rollout's
        candidate tensor in one pass; ``l2a_cem_refit`` - elite rows by rank counting instead of a sort, then their
        mean / std).  ``cem_mode='reference'`` keeps the reference's semantics on the device - rollouts on the
        UNCLIPPED samples, candidate-major rows (row = j*m + i read as env row // n), the rank-mask "elites" of
        ``:101`` pooled over the envs; ``cem_mode='fixed'`` uses clipped rollouts, env-major rows and true top-k
        elites per env.  The numbers are the library's own Philox stream, so both modes are validated against the
        host loops with injected normals, not bit-for-bit against NumPy."""
        import ctypes
        from ..dynamics.native_model import _ptr, _stream_ptr
        reference = (self.cem_mode == "reference")
        n, m, h = self.n_candidates, len(observations), self.horizon
        act_dim = self.action_space.shape[0]
        D = h * act_dim
        dev = self._device()
        native = self.dynamics_model.planner_model()
        ctx, lib = native.ctx, native.lib
        num_elites = max(int(n * self.percent_elites), 1)
        low = self._bufs.get("cem_low")
        if low is None or low.device != dev:
            low = self._bufs["cem_low"] = torch.as_tensor(np.asarray(self.action_space.low), dtype=torch.float32, device=dev)
            self._bufs["cem_high"] = torch.as_tensor(np.asarray(self.action_space.high), dtype=torch.float32, device=dev)
        high = self._bufs["cem_high"]
        mean = self._buf("cem_mean", (m, D), torch.float32, dev).zero_()
        std = self._buf("cem_std", (m, D), torch.float32, dev).fill_(1.0)
        rank, world = self._dist()
        lo, hi = self._shard_range(n, rank, world)
        n_local = hi - lo
        widths = [self._shard_range(n, r, world)[1] - self._shard_range(n, r, world)[0] for r in range(world)]
        wmax = max(widths)
        a_clip = self._buf("cem_a_clip", (n, m, D), torch.float32, dev)
        a_raw = self._buf("cem_a_raw", (n, m, D), torch.float32, dev) if reference else None
        seq = self._buf("cem_seq", (h, m * max(n_local, 1), act_dim), torch.float32, dev)
        elite_rows = self._buf("cem_rows", (m * num_elites,), torch.int32, dev)
        rets = None
        obs_dev = self._upload_obs(observations)             # once per plan step, not once per CEM iteration
        # counter-based stream: (seed, calls so far) - identical on every rank; restarts when torch's seed VALUE changes (seeding
        # again with the same value does not rewind it)
        seed = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
        if self._bufs.get("cem_seed") != seed:
            self._bufs["cem_seed"], self._bufs["cem_calls"] = seed, 0
        call0 = self._bufs["cem_calls"]
        for it in range(self.num_cem_iters):
            z = self._cem_normal_device((n, m, D), dev)
            if z is not None:
                z = z.to(device=dev, dtype=torch.float32).contiguous()
            offset = (call0 + it) * n * m * D
            ctx.check(lib.l2a_cem_sample(ctx.handle, _ptr(z), ctypes.c_ulonglong(seed), ctypes.c_ulonglong(offset),
                                         _ptr(mean), _ptr(std), _ptr(low), _ptr(high), n, m, h, act_dim,
                                         1 if reference else 0, lo, hi, _ptr(a_clip), _ptr(a_raw),
                                         _ptr(seq) if n_local > 0 else None, _stream_ptr(dev)), "l2a_cem_sample")
            if n_local > 0:
                _, r_loc = self._rollout(observations, seq, n_local, lo, want_returns=True, obs_dev=obs_dev)
            else:
                r_loc = torch.zeros((m, 0), dtype=torch.float32, device=dev)
            if world > 1:       # shards differ by at most one candidate: pad to the widest for the all-gather
                mine = r_loc if n_local == wmax else torch.cat(
                    [r_loc, torch.zeros((m, wmax - n_local), dtype=r_loc.dtype, device=dev)], dim=1)
                parts = self._all_gather(mine.contiguous(), world)
                rets = torch.cat([part[:, :w] for part, w in zip(parts, widths)], dim=1).to(dev).contiguous()
            else:
                rets = r_loc
            ctx.check(lib.l2a_cem_refit(ctx.handle, _ptr(rets), _ptr(a_clip), n, m, D, num_elites,
                                        1 if reference else 0, float(self.alpha), _ptr(elite_rows), _ptr(mean),
                                        _ptr(std), _stream_ptr(dev)), "l2a_cem_refit")
        # candidates as the rollout saw them: the reference reads the sample memory as [m, n, D] (:92-96) and returns
        # the UNCLIPPED first action (:106); the fixed mode's candidate (i, j) is clipped sample row j * m + i.  ONE launch packs
        # arg-max, that candidate's first action, its return and the final mean / std (`l2a_cem_pick`), one copy brings them back
        # (until round 4: five stock launches and five copies, ~1 % of a config-5 plan step)
        width = act_dim + 2
        packed = self._buf("cem_packed", (m * width + 2 * m * D,), torch.float32, dev)
        ctx.check(lib.l2a_cem_pick(ctx.handle, _ptr(rets), _ptr(a_raw if reference else a_clip), _ptr(mean), _ptr(std), n, m, D,
                                   act_dim, 1 if reference else 0, _ptr(packed), _stream_ptr(dev)), "l2a_cem_pick")
        host = packed.cpu().numpy()
        head = host[:m * width].reshape(m, width)
        out = head[:, :act_dim].astype(np.float64)
        best_return = head[:, act_dim].copy()
        best_index = head[:, act_dim + 1].copy().view(np.int32).astype(np.int64)
        mean_h = host[m * width:m * width + m * D].reshape(m, D).copy()
        std_h = host[m * width + m * D:].reshape(m, D).copy()
        if world > 1:
            # every rank gathered every rank's returns: all of them replay when any launch lost its tile-split partner;
            # the same collective carries the seed digests (ranks seeded differently sampled different normals)
            if self._agree(self._status_flag(), world):
                bad = True
                if not retry:
                    self._force_unsplit()
            else:
                bad = False
        else:
            bad = self._check_status() is False     # a launch lost its tile-split partner: the context is unsplit now
        if bad:
            if retry:
                raise _lib.L2AError("rollout launch failed twice")
            return self.get_cem_action_device(observations, retry=True)      # same counters: the same normals again
        self._bufs["cem_calls"] = call0 + self.num_cem_iters
        self.last_plan = dict(best_index=best_index, best_return=best_return, cem_mean=mean_h, cem_std=std_h)
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
