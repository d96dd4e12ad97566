"""``np.random.random_sample`` of the legacy global generator, ~4x faster, same stream.

Parity mode must consume NumPy's global MT19937 stream exactly as the reference does
(``policies/mpc_controller.py:67-69``); its draw is the host-side cost of a controller step (1.6 M doubles per
GrBAL step).  ``csrc/l2a_rng.c`` restates the generator with vectorisable loops; this module moves the state out
of ``np.random.get_state()``, lets the helper fill the array and puts the advanced state back (the cached
Gaussian of ``np.random.normal`` is preserved).  Trust is earned at run time: the first call compares the helper
with ``np.random.random_sample`` on a saved state, across a state-block boundary; on any difference, or when
``libl2a_rng.so`` is missing (no gcc at build time), NumPy's own call is used.
"""

import ctypes
import os

import numpy as np

_LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libl2a_rng.so")
_state = {"lib": None, "ok": None}


def _load():
    if _state["lib"] is None and os.path.exists(_LIB_PATH):
        try:
            lib = ctypes.CDLL(_LIB_PATH)
            lib.l2a_mt19937_fill_double.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_void_p,
                                                    ctypes.c_longlong]
            lib.l2a_mt19937_fill_double.restype = ctypes.c_int
            _state["lib"] = lib
        except OSError:
            _state["lib"] = None
    return _state["lib"]


def _fill(lib, n):
    st = np.random.get_state()
    if st[0] != "MT19937":
        return None
    key = np.array(st[1], dtype=np.uint32, copy=True)
    pos = ctypes.c_int(int(st[2]))
    out = np.empty(n, dtype=np.float64)
    if lib.l2a_mt19937_fill_double(key.ctypes.data, ctypes.byref(pos), out.ctypes.data, n) != 0:
        return None
    np.random.set_state(("MT19937", key, pos.value, st[3], st[4]))
    return out


def available():
    """True when the helper is loaded and has reproduced NumPy's stream on this machine."""
    if _state["ok"] is None:
        lib = _load()
        ok = False
        if lib is not None:
            saved = np.random.get_state()
            try:
                want = np.random.random_sample(1500)            # crosses at least two 624-word blocks
                after_want = np.random.get_state()
                np.random.set_state(saved)
                got = _fill(lib, 1500)
                after_got = np.random.get_state()
                ok = (got is not None and np.array_equal(want, got) and after_want[2] == after_got[2]
                      and np.array_equal(after_want[1], after_got[1]))
            finally:
                np.random.set_state(saved)
        _state["ok"] = bool(ok)
    return _state["ok"]


def random_sample(shape):
    """Drop-in for ``np.random.random_sample(shape)`` (legacy global generator)."""
    shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
    n = int(np.prod(shape)) if shape else 1
    if n >= 2048 and available():
        out = _fill(_state["lib"], n)
        if out is not None:
            return out.reshape(shape)
    return np.random.random_sample(shape)
