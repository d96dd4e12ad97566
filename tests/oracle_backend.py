"""Test-only stand-in for the HIP launch: lets the HOST logic of ``MPCController`` (RNG
consumption, candidate sharding, key decode, CEM bookkeeping, collectives) run on a CPU-only
box by computing one shard's returns with the oracle.  Never used by the product."""

import ctypes

import numpy as np
import torch

from learning_to_adapt_amd import _lib
from oracle import make_reward
from oracle.planner import rollout_returns

import cases


def _install_status_hooks(controller):
    """Launch-status hooks of the sharded paths: ``controller.harness_flags`` is a list of booleans handed out one per
    launch-status read (empty = every launch fine); ``controller.harness_unsplit`` counts the collective "switch to the
    unsplit geometry" decisions this rank took part in."""
    controller.harness_flags = []
    controller.harness_unsplit = 0

    def flag():
        return bool(controller.harness_flags.pop(0)) if controller.harness_flags else False

    def force_unsplit():
        controller.harness_unsplit += 1

    def pack_payload(best, m):
        # what `l2a_plan_payload` packs on the device: [keys, this rank's launch flag, digest, MASK - digest]
        payload = torch.empty((m + 3,), dtype=torch.int64)
        d = controller._rank_digest()
        payload[:m] = best
        payload[m] = 1 if flag() else 0
        payload[m + 1] = d
        payload[m + 2] = controller.DIGEST_MASK - d
        return payload

    controller._pack_payload = pack_payload
    controller._status_flag = flag
    controller._force_unsplit = force_unsplit


def install(controller, case):
    lib = _lib.load()
    env, _, _ = cases.recipe(case)
    dyn = cases.oracle_dynamics(case)
    reward = make_reward(case["env"], env.dt)
    cpu = torch.device("cpu")

    def _device():
        return cpu

    def _upload(actions_local):
        return torch.from_numpy(np.ascontiguousarray(actions_local, dtype=np.float32))

    def _rollout(observations, actions_local, n_local, cand_offset, want_returns, obs_dev=None):
        m = len(observations)
        acts = actions_local.numpy().astype(np.float64)
        rets = rollout_returns(dyn, reward, observations, acts, n_local, controller.discount)
        rets = rets.reshape(m, n_local).astype(np.float32)
        keys = np.zeros(m, dtype=np.int64)
        for i in range(m):
            best = max(lib.l2a_key_encode(ctypes.c_float(float(rets[i, j])), cand_offset + j)
                       for j in range(n_local))
            keys[i] = best
        return torch.from_numpy(keys), (torch.from_numpy(rets) if want_returns else None)

    controller._check_status = lambda: None
    _install_status_hooks(controller)
    controller.pipeline_chunks = 1            # the chunked launches bypass `_rollout`; the harness replaces that
    controller._upload_obs = lambda observations: None
    controller._device = _device
    controller._upload = _upload
    controller._rollout = _rollout
    return controller


def install_rnn(controller, case):
    """Recurrent twin of ``install``: the fused recurrent rollout AND the model's ``predict`` (used by
    ``get_actions`` to advance the hidden state) are computed by the oracle."""
    from oracle import LSTMStateTuple
    from oracle.rnn_planner import rnn_rollout_returns
    lib = _lib.load()
    env, _, _ = cases.rnn_recipe(case)
    dyn = cases.oracle_rnn_dynamics(case)
    reward = make_reward(case["env"], env.dt)
    cpu = torch.device("cpu")

    def _rollout(observations, actions_local, n_local, cand_offset, want_returns, obs_dev=None):
        m = len(observations)
        acts = actions_local.numpy().astype(np.float64)
        rets = rnn_rollout_returns(dyn, reward, observations, controller._hidden_state, acts, n_local,
                                   controller.discount)
        rets = rets.reshape(m, n_local).astype(np.float32)
        keys = np.zeros(m, dtype=np.int64)
        for i in range(m):
            keys[i] = max(lib.l2a_key_encode(ctypes.c_float(float(rets[i, j])), cand_offset + j)
                          for j in range(n_local))
        return torch.from_numpy(keys), (torch.from_numpy(rets) if want_returns else None)

    def predict(obs, act, hidden):
        return dyn.predict(obs, act, hidden)

    controller._check_status = lambda: None
    _install_status_hooks(controller)
    controller._upload_obs = lambda observations: None
    controller._device = lambda: cpu
    controller._upload = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    def _advance_hidden(observations, actions):
        _, controller._hidden_state = predict(np.array(observations), actions, controller._hidden_state)

    controller._rollout = _rollout
    controller._advance_hidden = _advance_hidden
    controller.dynamics_model.predict = predict
    return controller
