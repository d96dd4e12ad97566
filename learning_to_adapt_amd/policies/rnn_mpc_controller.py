"""``RNNMPCController`` - drop-in for ``learning_to_adapt/policies/rnn_mpc_controller.py:7-195`` (ReBAL).

Same constructor (``run_scripts/run_rebal.py:34-43``; note ``percent_elites=0.05`` and no ``alpha``
here, ``:19``), same ``get_action`` / ``get_actions`` / ``reset(dones)`` / ``repeat_hidden`` surface
and the same consumption of NumPy's global RNG as the reference.  The controller owns the model's
hidden state (one LSTM state row per env): ``reset`` zeroes the rows of finished envs
(``:136-163``), ``get_actions`` plans from it and then advances it with the chosen actions
(``:57-65``).

Underneath, the horizon loop with its per-step ``dynamics_model.predict(obs, a[t], hidden)``
(``:121-131`` / ``:97-104``) is ONE launch of the fused recurrent rollout (``l2a_lstm_plan_rs``):
the env's state is broadcast to its candidates inside the kernel (``repeat_hidden`` never
materialises), cell states stay in registers.  Candidate sharding over ``torch.distributed`` ranks,
the device RNG mode and CEM work as in ``MPCController`` (whose helpers are reused).
"""

import numpy as np
import torch

from ..utils.serializable import Serializable
from .mpc_controller import MPCController


class RNNMPCController(MPCController):
    _hid_host = None        # LSTMStateTuple of NumPy arrays (the reference's `_hidden_state`)
    _hid_dev = None         # (c, h) CUDA tensors
    _hid_stale = None       # which copy is out of date: None | "host" | "dev"
    _hid_next = None        # (c, h) already advanced with the chosen actions by the blocking plan launch
    _hid_flip = 0

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
            percent_elites=0.05,
            use_reward_model=False,
            rng="numpy",
            cem_mode="reference",
            shard_candidates=True,
            pipeline_chunks=3,
    ):
        Serializable.quick_init(self, locals())
        # alpha = 0 makes the shared CEM update `mean * alpha + (1 - alpha) * mean(elites)` the
        # reference's plain `np.mean(elites)` (:107), bit for bit.
        MPCController.__init__(self, name=name, env=env, dynamics_model=dynamics_model,
                               reward_model=reward_model, discount=discount, use_cem=use_cem,
                               n_candidates=n_candidates, horizon=horizon, num_cem_iters=num_cem_iters,
                               percent_elites=percent_elites, use_reward_model=use_reward_model, alpha=0.0,
                               rng=rng, cem_mode=cem_mode, shard_candidates=shard_candidates,
                               pipeline_chunks=pipeline_chunks)
        self._hidden_state = None

    # ------------------------------------------------------------------ hidden state: host view + device copy
    # The reference keeps `_hidden_state` as NumPy arrays (:30, :136-163) and so does this class for
    # everyone who looks at it.  The planner however consumes and produces the state on the GPU, so a
    # device copy is carried between controller steps and the host arrays are refreshed only when somebody
    # reads them (`reset` of a finished env, tests, pickling): no per-step round trip of 2 x [m, units].
    @property
    def _hidden_state(self):
        if self._hid_stale == "host":
            c, h = self._hid_dev
            self._hid_host = self._unpack(c.cpu().numpy(), h.cpu().numpy())
            self._hid_stale = None
        return self._hid_host

    @_hidden_state.setter
    def _hidden_state(self, value):
        self._hid_host = value
        self._hid_dev = None
        self._hid_stale = None if value is None else "dev"

    # The model converts between the reference's hidden-state structure (LSTMStateTuple / array / list of those,
    # rnn_dynamics.py:273-293) and the flat (c, h) [rows, sum(units)] arrays the kernels take; a model without the
    # helpers is a single-layer LSTM whose state is the (c, h) pair itself.
    def _pack(self, hidden):
        if hasattr(self.dynamics_model, "pack_hidden"):
            return self.dynamics_model.pack_hidden(hidden)
        return hidden[0], hidden[1]

    def _unpack(self, c, h):
        if hasattr(self.dynamics_model, "unpack_hidden"):
            return self.dynamics_model.unpack_hidden(c, h)
        return type(self.dynamics_model.get_initial_hidden(1))(c, h)

    def _device_hidden(self, device):
        """(c, h) CUDA tensors [m, sum(units)] of the current hidden state."""
        if self._hid_dev is None or self._hid_stale == "dev":
            c, h = self._pack(self._hid_host)
            up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)  # noqa: E731
            self._hid_dev = (up(c), up(h))
            self._hid_stale = None
        return self._hid_dev

    # ------------------------------------------------------------------ reference API
    def get_action(self, observation):
        if observation.ndim == 1:
            observation = observation[None]
        action = self.get_actions(observation)[0]
        return action, dict()

    def get_actions(self, observations):
        if self._hid_host is None and self._hid_dev is None:
            self.reset(dones=[True] * len(observations))
        if self.use_cem:
            actions = self.get_cem_action(observations)
        else:
            actions = self.get_rs_action(observations)
        # advance the controller's own hidden state with the chosen actions (:63)
        self._advance_hidden(observations, actions)
        return actions, dict()

    def _advance_hidden(self, observations, actions):
        if self._hid_next is not None:          # the blocking plan launch has done it in stream order (`_plan_keys`)
            self._hid_dev, self._hid_next = self._hid_next, None
            self._hid_stale = "host"
            return
        if not self._fusable():
            _, self._hidden_state = self.dynamics_model.predict(np.array(observations), actions, self._hidden_state)
            return
        native = self.dynamics_model.planner_model()
        dev = native.device
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)  # noqa: E731
        c, h = self._device_hidden(dev)
        if len(observations) <= 64 and hasattr(native, "advance"):
            c_out, h_out = native.advance(up(observations), up(actions), c, h)      # the kernel the blocking plan uses for this
        else:
            _, c_out, h_out = native.predict(up(observations), up(actions), c, h)
        self._hid_dev = (c_out, h_out)
        self._hid_stale = "host"

    def reset(self, dones=None):
        if dones is None:
            dones = [True]
        dones = np.asarray(dones, dtype=bool)
        if self._hid_host is None and self._hid_dev is None:
            self._hidden_state = self.dynamics_model.get_initial_hidden(batch_size=len(dones))
        if not dones.any():
            return                                              # nothing to zero: no host round trip
        # rows of finished envs go back to the cell's zero state (:139-163) - in the flat view, for every layer and
        # both parts at once (the reference's own loop only handles LSTM layers and stacks)
        zc, zh = self._pack(self.dynamics_model.get_initial_hidden(batch_size=1))
        c, h = self._pack(self._hidden_state)                   # refreshes the host arrays if needed
        c, h = np.array(c), np.array(h)
        c[dones] = zc
        h[dones] = zh
        self._hidden_state = self._unpack(c, h)                 # device copy is stale now

    def repeat_hidden(self, hidden, n):
        """``:165-187``: every row n times (env-major).  Only the unfused path materialises this."""
        if isinstance(hidden, (list, tuple)):
            rep = [self.repeat_hidden(part, n) for part in hidden]
            return type(hidden)(*rep) if hasattr(hidden, "_fields") else rep
        return np.repeat(hidden, n, axis=0)

    def __getstate__(self):
        state = dict()
        state["init_args"] = Serializable.__getstate__(self)                # reference :189-192
        return state

    def __setstate__(self, state):
        Serializable.__setstate__(self, state["init_args"] if "init_args" in state else state)

    # ------------------------------------------------------------------ fused rollout with the env's LSTM state
    def _plan_keys(self, observations, a_dev, n_local, lo, world):
        """One GPU, random shooting: the blocking launch (``l2a_lstm_plan_rs_sync``) - observations staged in
        host-mapped memory, keys through the mailbox - which also moves the controller's hidden state on with the
        winning first actions, in stream order behind the plan (the value ``_advance_hidden`` would upload is the
        fp32 candidate the kernel gathers).  Everything else goes through ``_rollout`` as before."""
        m = len(observations)
        self._hid_next = None
        stock = getattr(self._rollout, "__func__", None) is RNNMPCController._rollout
        native = self.dynamics_model.planner_model() if (stock and world == 1 and n_local > 0 and not self.use_cem) else None
        if native is None or not hasattr(native, "plan_rs_sync") or getattr(native, "sync_max_envs", 0) < m:
            return MPCController._plan_keys(self, observations, a_dev, n_local, lo, world)
        dev = native.device
        c0, h0 = self._device_hidden(dev)
        assert tuple(c0.shape) == (m, native.units), "hidden state holds %d rows, %d observations were passed" % (
            c0.shape[0], m)
        self._hid_flip ^= 1
        c1 = self._buf("adv_c%d" % self._hid_flip, (m, native.units), torch.float32, dev)
        h1 = self._buf("adv_h%d" % self._hid_flip, (m, native.units), torch.float32, dev)
        if c1.data_ptr() == c0.data_ptr() or h1.data_ptr() == h0.data_ptr():        # never write over the inputs
            c1, h1 = torch.empty_like(c0), torch.empty_like(h0)
        for _ in range(2):
            keys = native.plan_rs_sync(observations, c0, h0, a_dev, m, n_local, self.horizon, self.discount,
                                       self._reward_spec, cand_offset=lo, c_next=c1, h_next=h1)
            if keys is not None:
                self._hid_next = (c1, h1)
                return keys.view(np.int64)
        raise RuntimeError("recurrent rollout launch failed twice")

    # ---- the whole step in one C call (l2a_lstm_controller_step): the plan AND the state advance
    def _native_step_stock(self):
        return (getattr(self._rollout, "__func__", None) is RNNMPCController._rollout
                and getattr(self._plan_keys, "__func__", None) is RNNMPCController._plan_keys)

    def _native_step_state(self, native, m):
        dev = native.device
        self._hid_next = None
        c0, h0 = self._device_hidden(dev)
        assert tuple(c0.shape) == (m, native.units), "hidden state holds %d rows, %d observations were passed" % (
            c0.shape[0], m)
        self._hid_flip ^= 1
        c1 = self._buf("adv_c%d" % self._hid_flip, (m, native.units), torch.float32, dev)
        h1 = self._buf("adv_h%d" % self._hid_flip, (m, native.units), torch.float32, dev)
        if c1.data_ptr() == c0.data_ptr() or h1.data_ptr() == h0.data_ptr():        # never write over the inputs
            c1, h1 = torch.empty_like(c0), torch.empty_like(h0)
        return (c0.data_ptr(), h0.data_ptr(), c1.data_ptr(), h1.data_ptr()), (c1, h1)

    def _native_step_done(self, keep):
        self._hid_next = keep          # `_advance_hidden` adopts it: the C step advanced the state behind the plan

    def _rollout(self, observations, actions_local, n_local, cand_offset, want_returns, obs_dev=None):
        native = self.dynamics_model.planner_model()
        m = len(observations)
        dev = native.device
        c0, h0 = self._device_hidden(dev)
        assert tuple(c0.shape) == (m, native.units), "hidden state holds %d rows, %d observations were passed" % (
            c0.shape[0], m)
        obs0 = obs_dev if obs_dev is not None else self._upload_obs(observations)
        best = self._buf("best", (m,), torch.int64, dev)
        rets = self._buf("rets", (m, n_local), torch.float32, dev) if want_returns else None
        native.plan_rs(obs0, c0, h0, actions_local, m, n_local, self.horizon, self.discount, self._reward_spec,
                       cand_offset=cand_offset, returns_out=rets, best_key=best)
        return best, rets

    def _launch_chunk(self, native, c, last, obs0, a_dev, m, n_local, hc, t0, lo, best):
        """Recurrent chunk: observation, LSTM state and returns ping-pong between two buffer sets."""
        dev = native.device
        rows, U = m * n_local, native.units
        rets = [self._buf("pipe_ret%d" % i, (m, n_local), torch.float32, dev) for i in (0, 1)]
        state = [self._buf("pipe_state%d" % i, (rows, native.obs_dim), torch.float32, dev) for i in (0, 1)]
        cs = [self._buf("pipe_c%d" % i, (rows, U), torch.float32, dev) for i in (0, 1)]
        hs = [self._buf("pipe_h%d" % i, (rows, U), torch.float32, dev) for i in (0, 1)]
        c0, h0 = self._device_hidden(dev)
        src, dst = (c + 1) % 2, c % 2
        native.plan_rs_chunk(obs0 if c == 0 else state[src], c0 if c == 0 else cs[src], h0 if c == 0 else hs[src],
                             c > 0, a_dev, m, n_local, hc, t0, self.discount, self._reward_spec, cand_offset=lo,
                             returns_in=rets[src] if c > 0 else None, returns_out=rets[dst],
                             state_out=None if last else state[dst], c_out=None if last else cs[dst],
                             h_out=None if last else hs[dst], best_key=best if last else None)

    def _can_pipeline_cem(self, m, world, n):
        return False            # the candidate-chunked CEM rollout is wired for the feed-forward launch only

    def _get_rs_action_unfused(self, observations):
        """Custom env reward / reward model: the reference's loop shape (:112-134); the LSTM step
        still runs on the GPU through ``dynamics_model.predict``."""
        n, m, h = self.n_candidates, len(observations), self.horizon
        returns = np.zeros((n * m,))
        a = self.get_random_action(h * n * m).reshape((h, n * m, -1))
        cand_a = a[0].reshape((m, n, -1))
        observation = np.repeat(observations, n, axis=0)
        hidden_state = self.repeat_hidden(self._hidden_state, n)
        for t in range(h):
            next_observation, hidden_state = self.dynamics_model.predict(observation, a[t], hidden_state)
            if self.use_reward_model:
                assert self.reward_model is not None
                rewards = self.reward_model.predict(observation, a[t], next_observation)
            else:
                rewards = self.unwrapped_env.reward(observation, a[t], next_observation)
            returns += self.discount ** t * rewards
            observation = next_observation
        returns = returns.reshape(m, n)
        return cand_a[range(m), np.argmax(returns, axis=1)]
