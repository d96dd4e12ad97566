"""NumPy's legacy global generator (``np.random.random_sample`` / ``uniform`` / ``normal``), faster, same stream.

Parity mode must consume NumPy's global MT19937 stream exactly as the reference does
(``policies/mpc_controller.py:67-69`` uniform, ``:85`` normal); the draw is the host-side cost of a controller step
(360 k doubles per config-2 step, 2.88 M per config-4 step on every rank, 5 x 720 k normals per config-5 step).
``csrc/l2a_rng.c`` restates the generator with vectorisable loops, a data-parallel form of the legacy Gaussian's
rejection loop and a fork-join thread pool whose threads produce disjoint slices of the same stream; this module
moves the state out of ``np.random.get_state()``, lets the helper fill the arrays and puts the advanced state back
(position and cached Gaussian included).

Trust is earned at run time, per entry point: the first use compares the helper with NumPy's own call on a saved
state (across state-block boundaries, odd counts, a cached Gaussian); on any difference, or when ``libl2a_rng.so``
is missing (no gcc at build time), NumPy's own call is used.

A ``State`` is a private copy of the generator state: the planner uses it to draw the NEXT controller step's
candidates ahead of time without touching the global generator (``MPCController`` adopts the result only if the
global state is still the one the copy was taken from).
"""

import ctypes
import os

import numpy as np

_LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libl2a_rng.so")
_state = {"lib": None, "ok": {}, "threads": None}
_dp = ctypes.POINTER(ctypes.c_double)


def threads():
    """Worker threads of the helper: ``L2A_RNG_THREADS``, else half of the CPUs this process may run on divided
    by the ranks of a torch.distributed.run launch (``LOCAL_WORLD_SIZE``), at most 8."""
    if _state["threads"] is None:
        env = os.environ.get("L2A_RNG_THREADS")
        if env:
            t = max(1, int(env))
        else:
            try:
                ncpu = len(os.sched_getaffinity(0))
            except AttributeError:
                ncpu = os.cpu_count() or 1
            local = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
            t = max(1, min(8, ncpu // (2 * local)))
        _state["threads"] = t
    return _state["threads"]


def set_threads(n):
    _state["threads"] = max(1, int(n))


def _load():
    if _state["lib"] is None and os.path.exists(_LIB_PATH):
        try:
            lib = ctypes.CDLL(_LIB_PATH)
            c = ctypes
            vp, ll, i32 = c.c_void_p, c.c_longlong, c.c_int
            ip = c.POINTER(i32)
            if lib.l2a_rng_version() < 8:
                return None
            lib.l2a_mt19937_fill_double_mt.argtypes = [vp, ip, vp, ll, i32]
            lib.l2a_mt19937_fill_double_mt.restype = i32
            lib.l2a_mt19937_skip.argtypes = [vp, ip, ll]
            lib.l2a_mt19937_skip.restype = i32
            lib.l2a_mt19937_skip_mode.argtypes = [vp, ip, ll, i32]
            lib.l2a_mt19937_skip_mode.restype = i32
            lib.l2a_mt19937_uniform_rows.argtypes = [vp, ip, ll, i32, vp, vp, ll, ll, ll, vp, ll, vp, i32]
            lib.l2a_mt19937_uniform_rows.restype = i32
            lib.l2a_mt19937_fill_gauss.argtypes = [vp, ip, ip, _dp, vp, ll, i32]
            lib.l2a_mt19937_fill_gauss.restype = i32
            lib.l2a_cem_samples.argtypes = [vp, ll, ll, i32, i32, vp, vp, i32, vp, vp, vp, vp, vp, ll, ll, ll, i32, i32,
                                            i32]
            lib.l2a_cem_samples.restype = i32
            lib.l2a_cem_samples_steps.argtypes = [vp, ll, ll, i32, i32, vp, vp, i32, vp, vp, vp, vp, vp, ll, ll, ll, i32,
                                                  i32, i32, i32, i32]
            lib.l2a_cem_samples_steps.restype = i32
            lib.l2a_mt19937_state_equal.argtypes = [vp, vp, i32]
            lib.l2a_mt19937_state_equal.restype = i32
            lib.l2a_mt19937_state_store.argtypes = [vp, vp, i32]
            lib.l2a_mt19937_state_store.restype = None
            lib.l2a_mt19937_state_load.argtypes = [vp, vp, ip]
            lib.l2a_mt19937_state_load.restype = None
            lib.l2a_mt19937_state_digest.argtypes = [vp]
            lib.l2a_mt19937_state_digest.restype = c.c_ulonglong
            lib.l2a_cem_elite_stats.argtypes = [vp, vp, ll, i32, vp, vp]
            lib.l2a_cem_elite_stats.restype = ll
            lib.l2a_ahead_create.argtypes = [ll, i32, vp, vp, ll, ll, ll, ll, vp, vp, vp, vp, i32, vp, vp]
            lib.l2a_ahead_create.restype = vp
            lib.l2a_ahead_destroy.argtypes = [vp]
            lib.l2a_ahead_destroy.restype = None
            lib.l2a_ahead_arm.argtypes = [vp, vp]
            lib.l2a_ahead_arm.restype = i32
            lib.l2a_ahead_take.argtypes = [vp, vp]
            lib.l2a_ahead_take.restype = i32
            lib.l2a_ahead_next.argtypes = [vp]
            lib.l2a_ahead_next.restype = i32
            lib.l2a_ahead_stats.argtypes = [vp, _dp]
            lib.l2a_ahead_stats.restype = None
            _state["lib"] = lib
        except (OSError, AttributeError):
            _state["lib"] = None
    return _state["lib"]


def _global_addr():
    """Address of the global legacy generator's `mt19937_state` (key[624], pos), or None."""
    try:
        bg = np.random.mtrand._rand._bit_generator
        if type(bg).__name__ != "MT19937":
            return None
        return int(bg.ctypes.state_address)
    except Exception:
        return None


def _global_lock():
    """The bit generator's own lock: every NumPy draw holds it, so the 624 words + position are never read or written
    half way through another thread's draw (``get_state`` / ``set_state``, which the direct path replaces, hold it too)."""
    return np.random.mtrand._rand._bit_generator.lock


_FNV_OFFSET, _FNV_PRIME, _M64 = 1469598103934665603, 1099511628211, (1 << 64) - 1


def _fnv1a_words(h, words):
    for w in words:
        h = ((h ^ int(w)) * _FNV_PRIME) & _M64
    return h


def global_digest(with_gauss=False):
    """Fingerprint (< 2**47) of the global legacy generator's state - what the ranks of a sharded plan compare every
    step (``MPCController._combine_keys``): 64-bit FNV-1a over (key[624], pos), ~1 us through the helper library
    (``l2a_mt19937_state_digest``) and the SAME function in Python without it, so that ranks whose helper availability
    differs (a build or self-test failure on one node) still agree when their states do.  ``with_gauss``:
    the cached second value of the legacy polar Gaussian is folded in as well - planners that draw normals (CEM in parity
    mode) would otherwise combine returns of different samples when only that cache differs."""
    h = None
    st = None
    addr = _global_addr() if available("direct") else None
    with _global_lock():            # ONE consistent view: key, position and the Gaussian cache read under the generator's own lock
        if addr is not None:        # (every NumPy draw holds it; get_state itself does not take it)
            h = int(_state["lib"].l2a_mt19937_state_digest(addr))
        if h is None or with_gauss:
            st = np.random.get_state()      # copies the 624 words: CEM in parity mode only (one call per plan step beside ~15 ms)
    if h is None:
        # no helper library: the same FNV-1a in Python, a sequential 625-word loop (~100 us per step - the price of planning
        # sharded without libl2a_rng.so, which then also draws with NumPy's own loop)
        h = _fnv1a_words(_FNV_OFFSET, list(np.asarray(st[1], dtype=np.uint32)) + [int(st[2]) & 0xFFFFFFFF])
    if with_gauss:
        g = np.float64(st[4]).view(np.uint64)
        h = _fnv1a_words(h, [int(st[3]) & 0xFFFFFFFF, int(g) & 0xFFFFFFFF, int(g) >> 32])
    return h & 0x7FFFFFFFFFFF


class State(object):
    """A copy of a legacy MT19937 state: ``key`` (624 words), ``pos``, ``has_gauss``, ``gauss``."""

    __slots__ = ("key", "pos", "has_gauss", "gauss")

    def __init__(self, key, pos, has_gauss, gauss):
        self.key = np.array(key, dtype=np.uint32, copy=True)
        self.pos = ctypes.c_int(int(pos))
        self.has_gauss = ctypes.c_int(int(has_gauss))
        self.gauss = ctypes.c_double(float(gauss))

    @classmethod
    def from_global(cls):
        st = np.random.get_state()
        if st[0] != "MT19937":
            return None
        return cls(st[1], st[2], st[3], st[4])

    def copy(self):
        return State(self.key, self.pos.value, self.has_gauss.value, self.gauss.value)

    def to_global(self):
        np.random.set_state(("MT19937", self.key, self.pos.value, self.has_gauss.value, self.gauss.value))

    def same_as_global(self):
        """Is the global generator exactly in this state (nobody drew from it since the copy was taken)?"""
        st = np.random.get_state()
        return (st[0] == "MT19937" and st[2] == self.pos.value and st[3] == self.has_gauss.value
                and (not st[3] or st[4] == self.gauss.value) and np.array_equal(st[1], self.key))

    # ---- word-level access (key + pos only; the cached Gaussian is neither compared nor written): enough for the
    #      uniform stream, which never touches it - and ~1 us instead of get_state() / set_state() ----------------
    def same_words_as_global(self):
        if available("direct"):
            addr = _global_addr()
            if addr is not None:
                with _global_lock():
                    return bool(_state["lib"].l2a_mt19937_state_equal(addr, self.key.ctypes.data, self.pos.value))
        st = np.random.get_state()
        return st[0] == "MT19937" and st[2] == self.pos.value and np.array_equal(st[1], self.key)

    def words_to_global(self):
        if available("direct"):
            addr = _global_addr()
            if addr is not None:
                with _global_lock():
                    _state["lib"].l2a_mt19937_state_store(addr, self.key.ctypes.data, self.pos.value)
                return
        st = np.random.get_state()
        np.random.set_state(("MT19937", self.key, self.pos.value, st[3], st[4]))

    # ---- draws on this state (advance it) ------------------------------------------------------------------
    def random_sample(self, n, out=None):
        lib = _state["lib"]
        if out is None:
            out = np.empty(int(n), dtype=np.float64)
        rc = lib.l2a_mt19937_fill_double_mt(self.key.ctypes.data, ctypes.byref(self.pos), out.ctypes.data, int(n),
                                            threads())
        if rc != 0:
            raise RuntimeError("l2a_mt19937_fill_double_mt failed (%d)" % rc)
        return out

    def skip_doubles(self, n, jump=None):
        """Advance by ``n`` doubles.  ``jump``: None = the library's choice (polynomial jump-ahead for long
        distances, block regeneration otherwise), True / False = force one of them (tests)."""
        lib = _state["lib"]
        if jump is None:
            rc = lib.l2a_mt19937_skip(self.key.ctypes.data, ctypes.byref(self.pos), 2 * int(n))
        else:
            rc = lib.l2a_mt19937_skip_mode(self.key.ctypes.data, ctypes.byref(self.pos), 2 * int(n), 1 if jump else 0)
        if rc != 0:
            raise RuntimeError("l2a_mt19937_skip failed (%d)" % rc)

    def uniform_rows(self, rows, low, high, period, sel_lo, sel_hi, out_f32, rows64=0, out_f64=None):
        """``rows`` rows of ``len(low)`` uniforms ``low + (high - low) * u`` (the reference's
        ``get_random_action``); rows whose ``row % period`` lies in ``[sel_lo, sel_hi)`` go to ``out_f32``
        (fp32, compact), the first ``rows64`` rows also to ``out_f64``."""
        low = np.ascontiguousarray(low, dtype=np.float64)
        high = np.ascontiguousarray(high, dtype=np.float64)
        rc = _state["lib"].l2a_mt19937_uniform_rows(
            self.key.ctypes.data, ctypes.byref(self.pos), int(rows), int(low.shape[0]), low.ctypes.data,
            high.ctypes.data, int(period), int(sel_lo), int(sel_hi),
            out_f32.ctypes.data if out_f32 is not None else None, int(rows64),
            out_f64.ctypes.data if out_f64 is not None else None, threads())
        if rc != 0:
            raise RuntimeError("l2a_mt19937_uniform_rows failed (%d)" % rc)

    def standard_normal(self, n, out=None):
        if out is None:
            out = np.empty(int(n), dtype=np.float64)
        rc = _state["lib"].l2a_mt19937_fill_gauss(self.key.ctypes.data, ctypes.byref(self.pos),
                                                  ctypes.byref(self.has_gauss), ctypes.byref(self.gauss),
                                                  out.ctypes.data, int(n), threads())
        if rc != 0:
            raise RuntimeError("l2a_mt19937_fill_gauss failed (%d)" % rc)
        return out


def _verify(kind):
    """Reproduce NumPy's own call with the helper on a saved state; restores the global state."""
    lib = _load()
    if lib is None:
        return False
    saved = np.random.get_state()
    ok = False
    try:
        if kind == "double":
            want = np.random.random_sample(70001)               # > 65536: the threaded path; odd block offsets
            after = State.from_global()
            np.random.set_state(saved)
            st = State.from_global()
            got = st.random_sample(70001)
            ok = st is not None and np.array_equal(want, got) and _same(st, after)
        elif kind == "uniform":
            low, high = np.array([-1.0, -0.5, -150.0]), np.array([1.0, 2.5, 150.0])
            rows = 30011
            want = np.random.uniform(low=low, high=high, size=(rows, 3))
            after = State.from_global()
            np.random.set_state(saved)
            st = State.from_global()
            f32 = np.empty((rows // 7 + 7, 3), dtype=np.float32)
            f64 = np.empty((100, 3))
            # candidates 2..2 of every block of 7 rows
            st.uniform_rows(rows, low, high, 7, 2, 3, f32, 100, f64)
            sel = want[2::7].astype(np.float32)
            ok = (np.array_equal(f64, want[:100]) and np.array_equal(f32[:len(sel)], sel) and _same(st, after))
        elif kind == "direct":
            addr = _global_addr()
            if addr is not None:
                np.random.random_sample(701)                     # somewhere inside a block
                ref = np.random.get_state()
                key = np.empty(624, dtype=np.uint32)
                pos = ctypes.c_int(-1)
                lib.l2a_mt19937_state_load(addr, key.ctypes.data, ctypes.byref(pos))
                ok = np.array_equal(key, ref[1]) and pos.value == ref[2]
                st = State(ref[1], ref[2], ref[3], ref[4])
                ok = ok and bool(lib.l2a_mt19937_state_equal(addr, st.key.ctypes.data, st.pos.value))
                want = np.random.random_sample(1300)             # advance the real generator ...
                after = np.random.get_state()
                np.random.set_state(ref)
                got = st.random_sample(1300)                     # ... and the copy, then store the copy's words
                lib.l2a_mt19937_state_store(addr, st.key.ctypes.data, st.pos.value)
                now = np.random.get_state()
                ok = (ok and np.array_equal(want, got) and np.array_equal(now[1], after[1]) and now[2] == after[2])
                if ok:                                           # and the generator really continues from there
                    nxt = np.random.random_sample(5)
                    np.random.set_state(after)
                    ok = np.array_equal(nxt, np.random.random_sample(5))
        elif kind == "elite":
            rs = np.random.RandomState(12345)
            ok = True
            for rows, D, frac in ((4000, 180, 0.1), (333, 7, 0.5), (64, 2, 1.0), (50, 33, 0.02), (1000, 60, 0.05)):
                a = rs.randn(rows, D) * 10.0 ** rs.randint(-3, 4, size=(1, D))
                mask = rs.rand(rows) < frac
                mask[rs.randint(rows)] = True
                mu, sd = elite_stats(a, mask, _trusted=True)
                el = a[mask]
                ok = ok and np.array_equal(mu, np.mean(el, axis=0)) and np.array_equal(sd, np.std(el, axis=0))
        elif kind == "normal":
            np.random.normal()                                   # leave a cached Gaussian behind
            start = np.random.get_state()
            want = np.concatenate([np.random.normal(size=50001), np.random.normal(size=(3, 7)).ravel(),
                                   np.random.normal(size=40000)])
            after = State.from_global()
            np.random.set_state(start)
            st = State.from_global()
            got = np.concatenate([st.standard_normal(50001), st.standard_normal(21), st.standard_normal(40000)])
            ok = np.array_equal(want, got) and _same(st, after)
    except Exception:
        ok = False
    finally:
        np.random.set_state(saved)
    return bool(ok)


def _same(a, b):
    return (a.pos.value == b.pos.value and a.has_gauss.value == b.has_gauss.value
            and a.gauss.value == b.gauss.value and np.array_equal(a.key, b.key))


def available(kind="double"):
    """True when the helper is loaded and has reproduced NumPy's ``kind`` stream on this machine."""
    ok = _state["ok"].get(kind)
    if ok is None:
        ok = _verify(kind)
        _state["ok"][kind] = ok
    return ok


def random_sample(shape):
    """Drop-in for ``np.random.random_sample(shape)`` (legacy global generator)."""
    shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
    n = int(np.prod(shape)) if shape else 1
    if n >= 2048 and available("double"):
        st = State.from_global()
        if st is not None:
            out = st.random_sample(n)
            st.to_global()
            return out.reshape(shape)
    return np.random.random_sample(shape)


def standard_normal(shape):
    """Drop-in for ``np.random.normal(size=shape)`` (loc 0, scale 1) of the legacy global generator."""
    shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
    n = int(np.prod(shape)) if shape else 1
    if n >= 2048 and available("normal"):
        st = State.from_global()
        if st is not None:
            out = st.standard_normal(n)
            st.to_global()
            return out.reshape(shape)
    return np.random.normal(size=shape)


def elite_stats(a, mask, _trusted=False):
    """``(np.mean(a[mask], axis=0), np.std(a[mask], axis=0))`` for ``a [rows, D]`` float64 and a boolean ``mask [rows]`` - bit for
    bit (NumPy reduces the leading axis row after row; ``l2a_cem_elite_stats`` does the same in two passes without the gather and
    the temporaries), or ``None`` when the helper is missing / has not reproduced NumPy on this machine."""
    if not _trusted and not available("elite"):
        return None
    a = np.ascontiguousarray(a, dtype=np.float64)
    mask = np.ascontiguousarray(mask, dtype=np.bool_)
    rows, D = a.shape
    assert mask.shape == (rows,)
    if D < 2:       # one column: the reduced axis is the contiguous one and NumPy sums it PAIRWISE - leave that to NumPy
        return None
    mu, sd = np.empty(D), np.empty(D)
    cnt = _state["lib"].l2a_cem_elite_stats(a.ctypes.data, mask.ctypes.data, rows, D, mu.ctypes.data, sd.ctypes.data)
    if cnt <= 0:
        return None
    return mu, sd


def cem_samples(z, row_base, h, act_dim, mean, std, low, high, a_out, clip_out, seq_f32, n, sel_lo, sel_hi,
                env_major, use_clipped, steps=None, nthreads=None):
    """One CEM iteration's ``a = mean + z * std``, ``clip`` and the fp32 ``[h, m * nsel, act_dim]`` tensor the
    rollout reads, in one threaded pass (``l2a_cem_samples``).  ``steps=(t0, t1)``: only those horizon steps (their
    columns of ``a_out`` / ``clip_out``, their slices of ``seq_f32``).  Returns False when the helper is unavailable."""
    lib = _load()
    if lib is None:
        return False
    rows = int(z.shape[0])
    m = int(mean.shape[0])
    assert z.flags.c_contiguous and a_out.flags.c_contiguous and mean.flags.c_contiguous and std.flags.c_contiguous
    assert z.dtype == np.float64 and mean.dtype == np.float64 and std.dtype == np.float64
    low = np.ascontiguousarray(low, dtype=np.float64)
    high = np.ascontiguousarray(high, dtype=np.float64)
    t0, t1 = (0, int(h)) if steps is None else (int(steps[0]), int(steps[1]))
    rc = lib.l2a_cem_samples_steps(z.ctypes.data, rows, int(row_base), int(h), int(act_dim), mean.ctypes.data,
                                   std.ctypes.data, m, low.ctypes.data, high.ctypes.data, a_out.ctypes.data,
                                   clip_out.ctypes.data if clip_out is not None else None,
                                   seq_f32.ctypes.data if seq_f32 is not None else None, int(n), int(sel_lo), int(sel_hi),
                                   1 if env_major else 0, 1 if use_clipped else 0, t0, t1,
                                   threads() if nthreads is None else int(nthreads))
    if rc != 0:
        raise RuntimeError("l2a_cem_samples_steps failed (%d)" % rc)
    return True


class AheadChain(object):
    """The C draw-ahead chain of ``csrc/l2a_rng.c`` (``l2a_ahead_*``) on host buffers - the random-shooting draw of the NEXT
    controller step produced by a C thread while the caller is busy, adopted only when the global generator is still in the
    state the block started from.  ``libl2a_hip.so``'s controller step (``l2a_controller_step``) drives the same chain with
    page-locked buffers and an upload callback; this wrapper serves the tests and hosts without a GPU."""

    def __init__(self, rows, low, high, period, sel_lo, sel_hi, rows64, nthreads=None):
        lib = _load()
        if lib is None or not available("uniform") or not available("direct"):
            raise RuntimeError("libl2a_rng.so is not available / not trusted on this machine")
        self.lib = lib
        low = np.ascontiguousarray(low, dtype=np.float64)
        high = np.ascontiguousarray(high, dtype=np.float64)
        act_dim = int(low.shape[0])
        nsel = int(sel_hi) - int(sel_lo)
        blocks = (int(rows) + int(period) - 1) // int(period)
        self.f32 = [np.zeros((blocks * max(nsel, 1), act_dim), dtype=np.float32) for _ in (0, 1)]
        self.f64 = [np.zeros((max(int(rows64), 1), act_dim), dtype=np.float64) for _ in (0, 1)]
        self.addr = _global_addr()
        self.handle = lib.l2a_ahead_create(int(rows), act_dim, low.ctypes.data, high.ctypes.data, int(period), int(sel_lo),
                                           int(sel_hi), int(rows64), self.f32[0].ctypes.data, self.f32[1].ctypes.data,
                                           self.f64[0].ctypes.data, self.f64[1].ctypes.data,
                                           threads() if nthreads is None else int(nthreads), None, None)
        if not self.handle:
            raise RuntimeError("l2a_ahead_create refused the request")

    def arm(self):
        with _global_lock():
            return self.lib.l2a_ahead_arm(self.handle, self.addr)

    def take(self):
        with _global_lock():
            return self.lib.l2a_ahead_take(self.handle, self.addr)

    def next(self):
        return self.lib.l2a_ahead_next(self.handle)

    def stats(self):
        out = (ctypes.c_double * 6)()
        self.lib.l2a_ahead_stats(self.handle, out)
        return dict(hits=int(out[0]), misses=int(out[1]), produced=int(out[2]), produce_us=out[3], wait_us=out[4],
                    armed=bool(out[5]))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.l2a_ahead_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
