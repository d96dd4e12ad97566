"""Python handle on an ``l2a_lstm`` (C ABI: ``include/l2a.h``, recurrent planner section).

Like ``native_model.NativeModel``: no numerics of its own, PyTorch-ROCm tensors are storage, the
fused kernels run on torch's current HIP stream, ``L2AError`` on any failure, no CPU fallback.
"""

import ctypes

import numpy as np
import torch

from .. import _lib
from ..envs.reward_spec import RewardSpec
from .native_model import _dvec, _ptr, _stream_ptr


class NativeLSTM(object):
    def __init__(self, obs_dim, act_dim, units, cell_act="tanh", output_act=None, device=None, cell_type="lstm"):
        """``units``: int (one layer) or a sequence (stacked cells, ``l2a_rnn_create``).  ``self.units`` is the
        state width ``sum(units)`` - the row length of every c / h tensor handed to the launches."""
        if not torch.cuda.is_available():
            raise _lib.L2AError("no MI355X visible to PyTorch-ROCm: the rollout path is HIP-only "
                                "(there is no CPU fallback)")
        if device is None:                  # one process per GPU: the process' current device (torch.cuda.set_device)
            device = torch.cuda.current_device()
        self.ctx = _lib.Context.get(device)
        self.lib = self.ctx.lib
        self.device = torch.device("cuda", device)
        self.layer_units = tuple(int(u) for u in (units if isinstance(units, (tuple, list)) else (units,)))
        self.cell_type = cell_type
        self.obs_dim, self.act_dim, self.units = int(obs_dim), int(act_dim), int(sum(self.layer_units))
        if cell_act not in _lib.ACT_CODES or output_act not in _lib.ACT_CODES:
            raise _lib.L2AError("nonlinearity %r / %r is not supported by the HIP kernels" % (cell_act, output_act))
        if cell_type not in _lib.CELL_CODES:
            raise _lib.L2AError("cell type %r is not supported by the HIP kernels" % (cell_type,))
        handle = ctypes.c_void_p()
        arr = (ctypes.c_int * len(self.layer_units))(*self.layer_units)
        rc = self.lib.l2a_rnn_create(self.ctx.handle, self.obs_dim, self.act_dim, len(self.layer_units), arr,
                                     _lib.CELL_CODES[cell_type], _lib.ACT_CODES[cell_act], _lib.ACT_CODES[output_act],
                                     ctypes.byref(handle))
        self.ctx.check(rc, "l2a_rnn_create")
        self.handle = handle
        self._keep = {}

    def close(self):
        if getattr(self, "handle", None):
            self.lib.l2a_lstm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_weights(self, params):
        """``params``: the model's variables in ``get_params()`` order (``dynamics/rnn_cells.param_spec``); for one
        LSTM layer [kernel [in + U, 4U], bias [4U], output kernel [U, obs_dim], output bias]."""
        from . import rnn_cells
        shapes = [shape for _, shape in rnn_cells.param_spec(self.obs_dim, self.act_dim, self.layer_units, self.cell_type)]
        assert len(params) == len(shapes), "expected %d parameter arrays" % len(shapes)
        dev = []
        for p, shp in zip(params, shapes):
            t = torch.as_tensor(p) if not torch.is_tensor(p) else p
            t = t.detach().to(device=self.device, dtype=torch.float32).contiguous()
            assert tuple(t.shape) == shp, "LSTM parameter has shape %s, expected %s" % (tuple(t.shape), shp)
            dev.append(t)
        ptrs = (ctypes.c_void_p * len(dev))(*[t.data_ptr() for t in dev])
        self.ctx.check(self.lib.l2a_lstm_set_weights(self.handle, ptrs, _stream_ptr(self.device)), "l2a_lstm_set_weights")
        self._keep["w"] = dev

    def set_norm(self, norm):
        if norm is None:
            null = ctypes.POINTER(ctypes.c_double)()
            rc = self.lib.l2a_lstm_set_norm(self.handle, null, null, null, null, null, null, _stream_ptr(self.device))
        else:
            keep, args = [], []
            for key in ("obs", "act", "delta"):
                for j in (0, 1):
                    arr, p = _dvec(norm[key][j])
                    expect = self.act_dim if key == "act" else self.obs_dim
                    assert arr.shape == (expect,), "normalization[%r] has shape %s" % (key, arr.shape)
                    keep.append(arr)
                    args.append(p)
            rc = self.lib.l2a_lstm_set_norm(self.handle, *args, _stream_ptr(self.device))
        self.ctx.check(rc, "l2a_lstm_set_norm")

    def plan_rs(self, obs0, c0, h0, actions, m, n, h, discount, reward, cand_offset=0, returns_out=None,
                best_key=None):
        for t in (obs0, c0, h0, actions):
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        assert obs0.numel() == m * self.obs_dim and actions.numel() == h * m * n * self.act_dim
        assert c0.numel() == m * self.units and h0.numel() == m * self.units
        if returns_out is not None:
            assert returns_out.is_cuda and returns_out.dtype == torch.float32 and returns_out.numel() == m * n
        if best_key is not None:
            assert best_key.is_cuda and best_key.dtype == torch.int64 and best_key.numel() == m
        assert isinstance(reward, RewardSpec)
        rc = self.lib.l2a_lstm_plan_rs(self.handle, _ptr(obs0), _ptr(c0), _ptr(h0), _ptr(actions), int(m), int(n),
                                       int(h), float(discount), ctypes.byref(reward), int(cand_offset),
                                       _ptr(returns_out), _ptr(best_key), _stream_ptr(self.device))
        self.ctx.check(rc, "l2a_lstm_plan_rs")

    sync_max_envs = 64

    def plan_rs_sync(self, obs_host, c0, h0, actions, m, n, h, discount, reward, cand_offset=0, c_next=None, h_next=None):
        """Blocking plan step (``l2a_lstm_plan_rs_sync``): ``obs_host`` is a HOST array ``[m, obs_dim]``; returns the
        arg-max keys as a NumPy uint64 array ``[m]``.  With ``c_next`` / ``h_next`` (CUDA ``[m, units]``, not aliasing
        ``c0`` / ``h0``) the state is also advanced with every env's winning first action, in stream order behind
        the plan.  ``None``: the launch was flagged invalid (unit-tile split partner missing) - the context has been
        switched to the unsplit geometry and the caller repeats the call (which rewrites ``c_next`` / ``h_next``)."""
        for t in (c0, h0, actions):
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        assert actions.numel() == h * m * n * self.act_dim
        assert c0.numel() == m * self.units and h0.numel() == m * self.units
        if c_next is not None:
            for t in (c_next, h_next):
                assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == m * self.units
        obs = np.ascontiguousarray(obs_host, dtype=np.float32)
        assert obs.size == m * self.obs_dim
        keys = np.empty(m, dtype=np.uint64)
        rc = self.lib.l2a_lstm_plan_rs_sync(self.handle, ctypes.c_void_p(obs.ctypes.data), _ptr(c0), _ptr(h0), _ptr(actions),
                                            int(m), int(n), int(h), float(discount), ctypes.byref(reward),
                                            int(cand_offset), ctypes.c_void_p(keys.ctypes.data), _ptr(c_next),
                                            _ptr(h_next), _stream_ptr(self.device))
        if rc == _lib.L2A_ESPLIT:
            if getattr(self.ctx, "split_degraded", False):
                raise _lib.L2AError("recurrent rollout launch was flagged invalid with the tile split disabled")
            self.ctx.set_split(0)
            self.ctx.split_degraded = True
            return None
        self.ctx.check(rc, "l2a_lstm_plan_rs_sync")
        return keys

    def plan_payload(self, best_key, m, digest, payload):
        """As ``NativeModel.plan_payload``: the sharded recurrent planner packs its collective's payload the same way."""
        self.ctx.plan_payload(best_key, m, digest, payload, _stream_ptr(self.device))

    def plan_rs_chunk(self, state, c, h, per_row, actions, m, n, h_chunk, t0, discount, reward, cand_offset=0,
                      returns_in=None, returns_out=None, state_out=None, c_out=None, h_out=None, best_key=None):
        """Horizon steps ``t0 .. t0 + h_chunk - 1`` of a recurrent plan (``l2a_lstm_plan_rs_chunk``)."""
        for t in (state, c, h, actions):
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        rows = m * n if per_row else m
        assert state.numel() == rows * self.obs_dim and c.numel() == rows * self.units and h.numel() == rows * self.units
        assert actions.numel() == h_chunk * m * n * self.act_dim and returns_out.numel() == m * n
        assert isinstance(reward, RewardSpec)
        rc = self.lib.l2a_lstm_plan_rs_chunk(self.handle, _ptr(state), _ptr(c), _ptr(h), 1 if per_row else 0, _ptr(actions),
                                             int(m), int(n), int(h_chunk), int(t0), float(discount), ctypes.byref(reward),
                                             int(cand_offset), _ptr(returns_in), _ptr(returns_out), _ptr(state_out),
                                             _ptr(c_out), _ptr(h_out), _ptr(best_key), _stream_ptr(self.device))
        self.ctx.check(rc, "l2a_lstm_plan_rs_chunk")

    def predict(self, obs, act, c, h):
        """One step for independent rows.  Returns ``(next_obs, c_out, h_out)`` CUDA tensors."""
        for t in (obs, act, c, h):
            assert t.is_cuda and t.dtype == torch.float32
        rows = obs.shape[0]
        assert c.shape == (rows, self.units) and h.shape == (rows, self.units)
        nxt = torch.empty((rows, self.obs_dim), dtype=torch.float32, device=self.device)
        c_out = torch.empty((rows, self.units), dtype=torch.float32, device=self.device)
        h_out = torch.empty((rows, self.units), dtype=torch.float32, device=self.device)
        rc = self.lib.l2a_lstm_predict(self.handle, _ptr(obs.contiguous()), _ptr(act.contiguous()),
                                       _ptr(c.contiguous()), _ptr(h.contiguous()), int(rows), _ptr(nxt),
                                       _ptr(c_out), _ptr(h_out), _stream_ptr(self.device))
        self.ctx.check(rc, "l2a_lstm_predict")
        return nxt, c_out, h_out

    def advance(self, obs, act, c, h, c_out=None, h_out=None):
        """The controller's own state step (``l2a_lstm_advance``): ``(c_out, h_out)`` CUDA tensors from the chosen actions - no
        predicted observation.  The kernel the blocking plan launch enqueues behind its plan, so every controller path moves the
        state with the same arithmetic."""
        for t in (obs, act, c, h):
            assert t.is_cuda and t.dtype == torch.float32
        rows = obs.shape[0]
        assert c.shape == (rows, self.units) and h.shape == (rows, self.units)
        if c_out is None:
            c_out = torch.empty((rows, self.units), dtype=torch.float32, device=self.device)
            h_out = torch.empty((rows, self.units), dtype=torch.float32, device=self.device)
        rc = self.lib.l2a_lstm_advance(self.handle, _ptr(obs.contiguous()), _ptr(act.contiguous()), _ptr(c.contiguous()),
                                       _ptr(h.contiguous()), int(rows), _ptr(c_out), _ptr(h_out), _stream_ptr(self.device))
        self.ctx.check(rc, "l2a_lstm_advance")
        return c_out, h_out
