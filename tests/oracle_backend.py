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

    def _rollout(observations, actions_local, n_local, cand_offset, want_returns):
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
    controller._device = _device
    controller._upload = _upload
    controller._rollout = _rollout
    return controller
