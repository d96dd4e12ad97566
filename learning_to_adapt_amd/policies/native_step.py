"""Python handle on an ``l2a_controller`` (``include/l2a.h``: ``l2a_controller_*``) - the whole parity-mode controller step
of ``MPCController.get_actions`` / ``RNNMPCController.get_actions`` in ONE C call.

The reference's step (``policies/mpc_controller.py:59-69,108-129``) draws ``h*n*m`` candidate rows from NumPy's global
generator, rolls them out and returns the best candidate's first action.  The C controller keeps that contract - same numbers,
same generator state afterwards - with the draw done ahead of time by a C thread (``csrc/l2a_rng.c``, adopted only when the
global generator is still in the state the block started from), and the launch, the wait for the mailbox word, the key decode
and the gather of the float64 action inside ``l2a_controller_step``.  What is left on the Python side of a step: one copy of
the observations into a preallocated array, one ``ctypes`` call, two small array copies.
"""

import ctypes
import os

import numpy as np

from .. import _lib
from ..utils import fast_rng


class _DevWords(object):
    """``__cuda_array_interface__`` view of ``words`` int64 words at a raw device address (the C controller's payload buffer)."""

    def __init__(self, ptr, words):
        self.__cuda_array_interface__ = {"shape": (int(words),), "typestr": "<i8", "data": (int(ptr), False), "version": 2}


def _alias_int64(ptr, words, device):
    import torch
    return torch.as_tensor(_DevWords(ptr, words), device=device)


class NativeStep(object):
    def __init__(self, native, recurrent, m, n, h, low, high, discount, reward, device_seed=None, shard=None):
        """``device_seed``: None = parity mode (NumPy's global generator, candidates drawn ahead by a C thread); an integer =
        ``rng="device"``: the candidates come from the library's counter-based Philox stream under that seed, drawn on the GPU.
        ``shard``: None, or ``(rank, world, reduce)`` - the sharded step (``l2a_controller_create_sharded``, MLP models, parity
        mode): ``reduce`` = None runs the step's one collective over the library's own RCCL communicator (``l2a_comm_init``),
        else a callable ``reduce(payload)`` that MAX-all-reduces the int64 CUDA tensor it is handed in place (torch.distributed)."""
        lib = native.lib
        self.lib, self.ctx, self.native, self.recurrent = lib, native.ctx, native, bool(recurrent)
        self.m, self.n, self.h = int(m), int(n), int(h)
        low = np.ascontiguousarray(low, dtype=np.float64)
        high = np.ascontiguousarray(high, dtype=np.float64)
        handle = ctypes.c_void_p()
        self.device_rng = device_seed is not None
        self.reduce_error = None
        if self.device_rng and shard is not None and not self.recurrent:
            self.addr, self.lock = None, None
            rank, world, reduce = shard
            cb = self._make_reduce_cb(lib, native, reduce)
            rc = lib.l2a_controller_create_sharded_device(native.handle, self.m, self.n, self.h, low.ctypes.data, high.ctypes.data,
                                                          float(discount), ctypes.byref(reward),
                                                          ctypes.c_ulonglong(int(device_seed) & 0xFFFFFFFFFFFFFFFF), int(rank), int(world),
                                                          cb, None, ctypes.byref(handle))
        elif self.device_rng:
            self.addr, self.lock = None, None
            create = lib.l2a_lstm_controller_create_device if self.recurrent else lib.l2a_controller_create_device
            rc = create(native.handle, self.m, self.n, self.h, low.ctypes.data, high.ctypes.data, float(discount),
                        ctypes.byref(reward), ctypes.c_ulonglong(int(device_seed) & 0xFFFFFFFFFFFFFFFF), ctypes.byref(handle))
        else:
            self.addr = fast_rng._global_addr()
            if self.addr is None:
                raise _lib.L2AError("np.random's global generator is not the legacy MT19937")
            self.lock = fast_rng._global_lock()
            if shard is not None and not self.recurrent:
                rank, world, reduce = shard
                cb = self._make_reduce_cb(lib, native, reduce)
                rc = lib.l2a_controller_create_sharded(native.handle, self.m, self.n, self.h, low.ctypes.data, high.ctypes.data,
                                                       float(discount), ctypes.byref(reward), self.addr, fast_rng.threads(),
                                                       int(rank), int(world), cb, None, ctypes.byref(handle))
            else:
                create = lib.l2a_lstm_controller_create if self.recurrent else lib.l2a_controller_create
                rc = create(native.handle, self.m, self.n, self.h, low.ctypes.data, high.ctypes.data, float(discount),
                            ctypes.byref(reward), self.addr, fast_rng.threads(), ctypes.byref(handle))
        self.ctx.check(rc, "l2a_controller_create")
        self.handle = handle
        self.pid = os.getpid()      # a forked child must not tear down the parent's HIP objects (it drops the handle instead)
        self.obs = np.empty((self.m, native.obs_dim), dtype=np.float64)
        self.act = np.empty((self.m, native.act_dim), dtype=np.float64)
        self.idx = np.empty((self.m,), dtype=np.int64)
        self.ret = np.empty((self.m,), dtype=np.float32)
        self._p = (self.obs.ctypes.data, self.act.ctypes.data, self.idx.ctypes.data, self.ret.ctypes.data)
        self.misses_in_row = 0
        self.cooldown = 0
        self._stats = (ctypes.c_double * 16)()

    def _make_reduce_cb(self, lib, native, reduce):
        """ctypes callback around ``reduce(tensor)`` (None: the library's own RCCL communicator is used).  The collective runs on
        a tensor torch allocated itself (what every backend is used to); the library's words are copied in and out on the same
        stream (two tiny copies) - the alias of the library's buffer never reaches the process group."""
        self._reduce_cb = None
        if reduce is None:
            return None
        self._payload = None

        def _cb(arg, ptr, words, stream, _reduce=reduce):
            try:
                if self._payload is None or self._payload[0] != (ptr, words):
                    import torch
                    alias = _alias_int64(ptr, words, native.device)
                    self._payload = ((ptr, words), alias, torch.empty_like(alias))
                _, alias, own = self._payload
                own.copy_(alias)
                _reduce(own)
                alias.copy_(own)
                return 0
            except Exception as exc:          # an exception must not unwind through the C frame
                self.reduce_error = exc
                return -1
        self._reduce_cb = lib.REDUCE_FN(_cb)    # (kept alive with the controller)
        return ctypes.cast(self._reduce_cb, ctypes.c_void_p)

    def step(self, observations, stream, state=None):
        """One controller step.  Returns True (``self.act`` / ``self.idx`` / ``self.ret`` hold the result; when no valid block of
        candidates was waiting the C step drew them itself from the global generator), or False when the controller cannot serve
        the call (a forked child) - nothing has been consumed or launched then.  ``state`` (recurrent): ``(c0, h0, c_next,
        h_next)`` device pointers."""
        np.copyto(self.obs, observations, casting="same_kind")
        p = self._p
        if self.device_rng:
            rc = self._call(p, state, stream)
        else:
            # The generator's own lock - no other thread draws between the state compare and the adoption - held around the
            # FIRST half of the step only (take / draw, launch, producer kick: everything that touches the generator,
            # include/l2a.h "threading contract"); the wait for the GPU runs without it, so a thread that draws from
            # np.random meanwhile (an env reset) is not stalled for the length of a plan.
            with self.lock:
                rc = self._begin(p, state, stream)
            if rc == _lib.L2A_OK:
                rc = self.lib.l2a_controller_finish(self.handle, p[1], p[2], p[3])
        if rc == _lib.L2A_OK or rc == _lib.L2A_STEP_DREW:   # (DREW: no valid block was waiting, the step drew synchronously itself)
            self.misses_in_row = 0
            return True
        if rc == _lib.L2A_STEP_MISS:
            return False
        if rc == _lib.L2A_STEP_UNSPLIT:        # the C side has switched the context to the unsplit geometry (same bits)
            self.ctx.split_degraded = True
            self.misses_in_row = 0
            return True
        if getattr(self, "reduce_error", None) is not None:
            exc, self.reduce_error = self.reduce_error, None
            raise exc
        self.ctx.check(rc, "l2a_controller_step")

    def _begin(self, p, state, stream):
        if self.recurrent:
            return self.lib.l2a_lstm_controller_begin(self.handle, p[0], state[0], state[1], state[2], state[3], stream)
        return self.lib.l2a_controller_begin(self.handle, p[0], stream)

    def _call(self, p, state, stream):
        if self.recurrent:
            return self.lib.l2a_lstm_controller_step(self.handle, p[0], state[0], state[1], state[2], state[3], p[1], p[2], p[3], stream)
        return self.lib.l2a_controller_step(self.handle, p[0], p[1], p[2], p[3], stream)

    def rearm(self):
        """After a synchronous draw: the chain restarts at the current global state.  Backs off while steps keep missing
        (a consumer of ``np.random`` runs between the controller's steps: every block drawn ahead would be thrown away)."""
        if self.device_rng:
            return
        self.misses_in_row += 1
        if self.misses_in_row > 2:
            self.cooldown += 1
            if self.cooldown % 16 != 0:
                return
        with self.lock:
            self.ctx.check(self.lib.l2a_controller_rearm(self.handle), "l2a_controller_rearm")

    def stats(self):
        self.ctx.check(self.lib.l2a_controller_stats(self.handle, self._stats, 16), "l2a_controller_stats")
        v = list(self._stats)
        return dict(stage_us=dict(take=v[0], stage_obs=v[1], launch=v[2], kick=v[3], wait=v[4], decode=v[5], call=v[6]),
                    steps=int(v[7]), relaunches=int(v[8]), sync_draws=int(v[15]), hits=int(v[9]), misses=int(v[10]), produced=int(v[11]),
                    producer_us_per_block=v[12], consumer_wait_us_per_take=v[13], armed=bool(v[14]))

    def actions_ptr(self):
        return self.lib.l2a_controller_actions(self.handle)

    def close(self):
        if getattr(self, "handle", None):
            if getattr(self, "pid", None) == os.getpid():
                self.lib.l2a_controller_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
