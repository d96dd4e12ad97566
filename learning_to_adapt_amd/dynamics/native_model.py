"""Python handle on an ``l2a_model`` (C ABI: ``include/l2a.h``).

Holds no numerics of its own: it hands device pointers of PyTorch-ROCm tensors (storage only)
to ``libl2a_hip.so`` and launches the fused rollout / predict kernels on torch's current HIP
stream.  Raises ``L2AError`` on any failure - there is no CPU fallback.
"""

import ctypes

import numpy as np
import torch

from .. import _lib
from ..envs.reward_spec import RewardSpec


def _stream_ptr(device=None):
    """torch's current HIP stream on `device` (the model's GPU, not necessarily the process' current one)."""
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _dvec(x):
    x = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
    return x, x.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


class NativeModel(object):
    def __init__(self, obs_dim, act_dim, hidden_sizes, hidden_act, output_act, n_sets, mode, device=None):
        if not torch.cuda.is_available():
            raise _lib.L2AError("no MI355X visible to PyTorch-ROCm: the rollout path is HIP-only "
                                "(there is no CPU fallback)")
        if device is None:                  # one process per GPU: the process' current device (torch.cuda.set_device)
            device = torch.cuda.current_device()
        self.ctx = _lib.Context.get(device)
        self.lib = self.ctx.lib
        self.device = torch.device("cuda", device)
        self.obs_dim, self.act_dim = int(obs_dim), int(act_dim)
        self.hidden_sizes = tuple(int(h) for h in hidden_sizes)
        self.n_sets, self.mode = int(n_sets), mode
        if hidden_act not in _lib.ACT_CODES or output_act not in _lib.ACT_CODES:
            raise _lib.L2AError("nonlinearity %r / %r is not supported by the HIP kernels"
                                % (hidden_act, output_act))
        hid = (ctypes.c_int * len(self.hidden_sizes))(*self.hidden_sizes)
        handle = ctypes.c_void_p()
        rc = self.lib.l2a_model_create(self.ctx.handle, self.obs_dim, self.act_dim, len(self.hidden_sizes),
                                       hid, _lib.ACT_CODES[hidden_act], _lib.ACT_CODES[output_act],
                                       self.n_sets, _lib.MODE_CODES[mode], ctypes.byref(handle))
        self.ctx.check(rc, "l2a_model_create")
        self.handle = handle
        self._keep = {}     # source tensors kept alive until the stream has consumed them

    def close(self):
        if getattr(self, "handle", None):
            self.lib.l2a_model_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- parameters ---------------------------------------------------------------------
    def set_weights(self, e, params):
        """``params``: [W0, b0, ..., Wout, bout] (torch tensors or arrays, kernels ``[in, out]``)."""
        dev = []
        for p in params:
            t = torch.as_tensor(p) if not torch.is_tensor(p) else p
            dev.append(t.detach().to(device=self.device, dtype=torch.float32).contiguous())
        sizes = (self.obs_dim + self.act_dim,) + self.hidden_sizes + (self.obs_dim,)
        assert len(dev) == 2 * (len(sizes) - 1), "expected %d parameter arrays" % (2 * (len(sizes) - 1))
        for li in range(len(sizes) - 1):
            assert tuple(dev[2 * li].shape) == (sizes[li], sizes[li + 1]), \
                "kernel %d has shape %s, expected %s" % (li, tuple(dev[2 * li].shape), (sizes[li], sizes[li + 1]))
            assert tuple(dev[2 * li + 1].shape) == (sizes[li + 1],)
        ptrs = (ctypes.c_void_p * len(dev))(*[t.data_ptr() for t in dev])
        rc = self.lib.l2a_model_set_weights(self.handle, int(e), ptrs, _stream_ptr(self.device))
        self.ctx.check(rc, "l2a_model_set_weights")
        self._keep[("w", e)] = dev

    def set_weights_stacked(self, first_set, stacked):
        """``stacked``: [W0, b0, ..., Wout, bout] with a leading set axis (``[count, in, out]`` / ``[count, out]``
        CUDA tensors) - all ``count`` sets go up in one strided call (``l2a_model_set_weights_strided``)."""
        sizes = (self.obs_dim + self.act_dim,) + self.hidden_sizes + (self.obs_dim,)
        assert len(stacked) == 2 * (len(sizes) - 1)
        count = int(stacked[0].shape[0])
        dev = []
        for i, t in enumerate(stacked):
            t = t.detach().to(device=self.device, dtype=torch.float32).contiguous()
            li = i // 2
            expect = (count, sizes[li], sizes[li + 1]) if i % 2 == 0 else (count, sizes[li + 1])
            assert tuple(t.shape) == expect, "stacked parameter %d has shape %s, expected %s" % (i, tuple(t.shape), expect)
            dev.append(t)
        ptrs = (ctypes.c_void_p * len(dev))(*[t.data_ptr() for t in dev])
        strides = (ctypes.c_longlong * len(dev))(*[int(t.stride(0)) for t in dev])
        rc = self.lib.l2a_model_set_weights_strided(self.handle, int(first_set), count, ptrs, strides, _stream_ptr(self.device))
        self.ctx.check(rc, "l2a_model_set_weights_strided")
        self._keep[("w_stacked", first_set)] = dev

    def adapt_sgd(self, base_params, x, y, lr):
        """One SGD step per task on the device, written straight into weight sets ``0 .. m-1``
        (``l2a_model_adapt_sgd``).  ``base_params``: pre-update [W0, b0, ...] CUDA tensors; ``x`` /
        ``y``: fp32 CUDA ``[m, rows, in_dim]`` / ``[m, rows, obs_dim]`` (normalised)."""
        m, rows = int(x.shape[0]), int(x.shape[1])
        assert x.is_cuda and y.is_cuda and x.dtype == torch.float32 and y.dtype == torch.float32
        assert tuple(x.shape) == (m, rows, self.obs_dim + self.act_dim) and tuple(y.shape) == (m, rows, self.obs_dim)
        base = [t.detach().to(device=self.device, dtype=torch.float32).contiguous() for t in base_params]
        ptrs = (ctypes.c_void_p * len(base))(*[t.data_ptr() for t in base])
        x, y = x.contiguous(), y.contiguous()
        rc = self.lib.l2a_model_adapt_sgd(self.handle, ptrs, _ptr(x), _ptr(y), m, rows, float(lr), _stream_ptr(self.device))
        self.ctx.check(rc, "l2a_model_adapt_sgd")
        self._keep["adapt"] = (base, x, y)

    def adapt_sgd_host(self, base_params, x, y, lr):
        """``adapt_sgd`` with the batches as HOST float32 arrays ``[m, rows, in_dim]`` / ``[m, rows, obs_dim]``
        (``l2a_model_adapt_sgd_host``: host-mapped staging, the launches replayed as one hipGraph)."""
        m, rows = int(x.shape[0]), int(x.shape[1])
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.ascontiguousarray(y, dtype=np.float32)
        assert x.shape == (m, rows, self.obs_dim + self.act_dim) and y.shape == (m, rows, self.obs_dim)
        base = self._keep.get("adapt_base")
        if base is None or base[0] is not base_params:       # same list object = same tensors: keep the graph valid
            dev = [t.detach().to(device=self.device, dtype=torch.float32).contiguous() for t in base_params]
            base = (base_params, dev, (ctypes.c_void_p * len(dev))(*[t.data_ptr() for t in dev]))
            self._keep["adapt_base"] = base
        rc = self.lib.l2a_model_adapt_sgd_host(self.handle, base[2], ctypes.c_void_p(x.ctypes.data),
                                               ctypes.c_void_p(y.ctypes.data), m, rows, float(lr),
                                               _stream_ptr(self.device))
        self.ctx.check(rc, "l2a_model_adapt_sgd_host")

    raw_adapt_max_inputs = 128

    def adapt_sgd_raw(self, base_params, obs, act, obs_next, normalization, lr):
        """``adapt_sgd`` from the un-normalised float64 transitions ``[m, rows, dim]`` and ``normalization`` (the
        model's dict of (mean, std) pairs): the device normalises exactly as the host would
        (``l2a_model_adapt_sgd_raw``).  This wrapper sits in front of the GrBAL step's first launch: arrays and
        pointers it has seen before (the caller's stacking buffers, the normalisation vectors) are not converted again."""
        raw = self._keep.get("adapt_raw")
        if raw is None or raw[0] is not obs or raw[1] is not act or raw[2] is not obs_next:
            obs_c = np.ascontiguousarray(obs, dtype=np.float64)
            act_c = np.ascontiguousarray(act, dtype=np.float64)
            next_c = np.ascontiguousarray(obs_next, dtype=np.float64)
            m, rows = int(obs_c.shape[0]), int(obs_c.shape[1])
            assert obs_c.shape == (m, rows, self.obs_dim) and act_c.shape == (m, rows, self.act_dim) and next_c.shape == obs_c.shape
            # (identity of the CALLER's arrays is only remembered when they needed no conversion: the library copies out of
            #  them during the call, so the same buffers with new contents are fine)
            same = obs_c is obs and act_c is act and next_c is obs_next
            raw = (obs if same else None, act, obs_next, obs_c, act_c, next_c, m, rows,
                   ctypes.c_void_p(obs_c.ctypes.data), ctypes.c_void_p(act_c.ctypes.data), ctypes.c_void_p(next_c.ctypes.data))
            self._keep["adapt_raw"] = raw
        m, rows = raw[6], raw[7]
        base = self._keep.get("adapt_base")
        if base is None or base[0] is not base_params:
            dev = [t.detach().to(device=self.device, dtype=torch.float32).contiguous() for t in base_params]
            base = (base_params, dev, (ctypes.c_void_p * len(dev))(*[t.data_ptr() for t in dev]))
            self._keep["adapt_base"] = base
        nv = self._keep.get("adapt_norm")
        if nv is None or nv[0] is not normalization:
            vecs = [np.ascontiguousarray(normalization[k][j], dtype=np.float64) for k in ("obs", "act", "delta") for j in (0, 1)]
            assert vecs[0].shape == (self.obs_dim,) and vecs[2].shape == (self.act_dim,) and vecs[4].shape == (self.obs_dim,)
            nv = (normalization, vecs, [ctypes.c_void_p(v.ctypes.data) for v in vecs])
            self._keep["adapt_norm"] = nv
        p = nv[2]
        rc = self.lib.l2a_model_adapt_sgd_raw(self.handle, base[2], raw[8], raw[9], raw[10],
                                              p[0], p[1], p[2], p[3], p[4], p[5], m, rows, float(lr),
                                              _stream_ptr(self.device))
        self.ctx.check(rc, "l2a_model_adapt_sgd_raw")

    def get_weights(self, e):
        """Weight set ``e`` as fresh CUDA tensors in the reference's order and layout."""
        sizes = (self.obs_dim + self.act_dim,) + self.hidden_sizes + (self.obs_dim,)
        out = []
        for li in range(len(sizes) - 1):
            out.append(torch.empty((sizes[li], sizes[li + 1]), dtype=torch.float32, device=self.device))
            out.append(torch.empty((sizes[li + 1],), dtype=torch.float32, device=self.device))
        ptrs = (ctypes.c_void_p * len(out))(*[t.data_ptr() for t in out])
        rc = self.lib.l2a_model_get_weights(self.handle, int(e), ptrs, _stream_ptr(self.device))
        self.ctx.check(rc, "l2a_model_get_weights")
        return out

    def set_norm(self, e, norm):
        """``norm``: the reference's ``normalization`` dict (``'obs'/'act'/'delta' -> (mean, std)``)
        or ``None`` for identity."""
        if norm is None:
            null = ctypes.POINTER(ctypes.c_double)()
            rc = self.lib.l2a_model_set_norm(self.handle, int(e), null, null, null, null, null, null,
                                             _stream_ptr(self.device))
        else:
            keep, args = [], []
            for key in ("obs", "act", "delta"):
                for j in (0, 1):
                    arr, p = _dvec(norm[key][j])
                    expect = self.act_dim if key == "act" else self.obs_dim
                    assert arr.shape == (expect,), "normalization[%r] has shape %s" % (key, arr.shape)
                    keep.append(arr)
                    args.append(p)
            rc = self.lib.l2a_model_set_norm(self.handle, int(e), *args, _stream_ptr(self.device))
        self.ctx.check(rc, "l2a_model_set_norm")

    # ---- launches -----------------------------------------------------------------------
    def plan_rs(self, obs0, actions, m, n, h, discount, reward, cand_offset=0, returns_out=None,
                best_key=None):
        """All tensor arguments are fp32 / int64 CUDA tensors; see ``l2a_plan_rs`` in include/l2a.h."""
        assert obs0.is_cuda and actions.is_cuda and obs0.dtype == torch.float32 and actions.dtype == torch.float32
        assert obs0.is_contiguous() and actions.is_contiguous()
        assert obs0.numel() == m * self.obs_dim and actions.numel() == h * m * n * self.act_dim
        if returns_out is not None:
            assert returns_out.is_cuda and returns_out.dtype == torch.float32 and returns_out.numel() == m * n
        if best_key is not None:
            assert best_key.is_cuda and best_key.dtype == torch.int64 and best_key.numel() == m
        assert isinstance(reward, RewardSpec)
        rc = self.lib.l2a_plan_rs(self.handle, _ptr(obs0), _ptr(actions), int(m), int(n), int(h),
                                  float(discount), ctypes.byref(reward), int(cand_offset),
                                  _ptr(returns_out), _ptr(best_key), _stream_ptr(self.device))
        self.ctx.check(rc, "l2a_plan_rs")

    def plan_rs_sync(self, obs_host, actions, m, n, h, discount, reward, cand_offset=0, returns_out=None):
        """Blocking plan step (``l2a_plan_rs_sync``): ``obs_host`` is a HOST array ``[m, obs_dim]``; returns the
        arg-max keys as a NumPy uint64 array ``[m]``, or ``None`` when the launch was flagged invalid (tile-split
        partner missing) - the context has then been switched to the unsplit geometry and the caller repeats the
        call.  ctypes releases the GIL for the duration, so other Python threads run while the GPU plans."""
        assert actions.is_cuda and actions.dtype == torch.float32 and actions.is_contiguous()
        assert actions.numel() == h * m * n * self.act_dim
        obs = np.ascontiguousarray(obs_host, dtype=np.float32)
        assert obs.size == m * self.obs_dim
        keys = np.empty(m, dtype=np.uint64)
        rc = self.lib.l2a_plan_rs_sync(self.handle, ctypes.c_void_p(obs.ctypes.data), _ptr(actions), int(m), int(n),
                                       int(h), float(discount), ctypes.byref(reward), int(cand_offset),
                                       _ptr(returns_out), ctypes.c_void_p(keys.ctypes.data), _stream_ptr(self.device))
        if rc == _lib.L2A_ESPLIT:
            if getattr(self.ctx, "split_degraded", False):
                raise _lib.L2AError("rollout launch was flagged invalid with the tile split disabled")
            self.ctx.set_split(0)
            self.ctx.split_degraded = True
            return None
        self.ctx.check(rc, "l2a_plan_rs_sync")
        return keys

    sync_max_envs = 64

    def plan_payload(self, best_key, m, digest, payload):
        """Pack what a sharded plan all-reduces (``l2a_plan_payload``): keys, this rank's launch flag, the digest pair -
        on the device, in stream order behind the launch, no host synchronisation."""
        self.ctx.plan_payload(best_key, m, digest, payload, _stream_ptr(self.device))

    def plan_rs_chunk(self, state, state_per_row, actions, m, n, h_chunk, t0, discount, reward, cand_offset=0,
                      returns_in=None, returns_out=None, state_out=None, best_key=None):
        """Horizon steps ``t0 .. t0 + h_chunk - 1`` of a plan (``l2a_plan_rs_chunk``); all tensors fp32 CUDA."""
        assert state.is_cuda and actions.is_cuda and state.is_contiguous() and actions.is_contiguous()
        assert actions.numel() == h_chunk * m * n * self.act_dim
        assert state.numel() == (m * n if state_per_row else m) * self.obs_dim
        assert returns_out is not None and returns_out.numel() == m * n
        assert isinstance(reward, RewardSpec)
        rc = self.lib.l2a_plan_rs_chunk(self.handle, _ptr(state), 1 if state_per_row else 0, _ptr(actions), int(m), int(n),
                                        int(h_chunk), int(t0), float(discount), ctypes.byref(reward), int(cand_offset),
                                        _ptr(returns_in), _ptr(returns_out), _ptr(state_out), _ptr(best_key),
                                        _stream_ptr(self.device))
        self.ctx.check(rc, "l2a_plan_rs_chunk")

    def predict(self, obs, act, n_blocks=1, out=None):
        assert obs.is_cuda and act.is_cuda and obs.dtype == torch.float32 and act.dtype == torch.float32
        rows = obs.shape[0]
        if out is None:
            out = torch.empty((rows, self.obs_dim), dtype=torch.float32, device=self.device)
        rc = self.lib.l2a_predict(self.handle, _ptr(obs.contiguous()), _ptr(act.contiguous()), int(rows),
                                  int(n_blocks), _ptr(out), _stream_ptr(self.device))
        self.ctx.check(rc, "l2a_predict")
        return out
