"""The C-ABI library: loads on a CPU-only box, exports every symbol ``include/l2a.h`` declares,
and its GPU-free helpers (key packing, weight packing, eligibility) behave."""

import ctypes
import os
import re

import numpy as np

from learning_to_adapt_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "l2a.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(l2a_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _declared_functions()
    assert len(declared) >= 14
    for name in declared:
        assert hasattr(lib, name), "libl2a_hip.so does not export %s" % name
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared


def test_reward_struct_matches_header():
    from learning_to_adapt_amd.envs import RewardSpec
    text = open(os.path.join(ROOT, "include", "l2a.h")).read()
    body = re.search(r"typedef struct l2a_reward \{(.*?)\} l2a_reward;", text, flags=re.S).group(1)
    fields = re.findall(r"(float|int)\s+([a-z_]+);", body)
    assert [f[1] for f in fields] == [f[0] for f in RewardSpec._fields_]
    assert ctypes.sizeof(RewardSpec) == 4 * len(fields)


def test_init_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        return
    lib = _lib.load()
    handle = ctypes.c_void_p()
    rc = lib.l2a_init(0, ctypes.byref(handle))
    assert rc < 0 and not handle.value
    assert len(lib.l2a_last_error(None)) > 0


def test_key_roundtrip_and_order():
    lib = _lib.load()
    rs = np.random.RandomState(0)
    vals = np.concatenate([rs.randn(200).astype(np.float32) * 50, np.float32([0.0, -0.0, np.inf, -np.inf, 1e-38, -1e-38])])
    idxs = rs.randint(0, 2 ** 31 - 1, size=vals.shape[0])
    keys = [lib.l2a_key_encode(ctypes.c_float(float(v)), int(i)) for v, i in zip(vals, idxs)]
    for k, v, i in zip(keys, vals, idxs):
        assert k < 2 ** 63                      # signed max == unsigned max (int64 all-reduce MAX)
        r, j = _lib.key_decode(k)
        assert j == i and (np.float32(r) == v)
    order = np.argsort(keys)
    assert np.all(np.diff(vals[order]) >= 0)    # key order == return order
    # ties on the return -> the LOWER index wins, as np.argmax
    a = lib.l2a_key_encode(ctypes.c_float(1.5), 7)
    b = lib.l2a_key_encode(ctypes.c_float(1.5), 9)
    assert a > b
    # NaN sorts above everything (np.argmax picks the first NaN)
    assert lib.l2a_key_encode(ctypes.c_float(float("nan")), 3) > lib.l2a_key_encode(ctypes.c_float(float("inf")), 0)
    assert lib.l2a_key_encode(ctypes.c_float(-1e30), 0) > 0    # 0 is the "no candidate" sentinel


def test_pack_layer_is_a_padded_permutation():
    lib = _lib.load()
    rs = np.random.RandomState(1)
    for k_in, n_out in ((26, 512), (512, 20), (49, 128), (16, 16), (1, 1)):
        w = rs.randn(k_in, n_out).astype(np.float32)
        total = lib.l2a_packed_layer_floats(k_in, n_out)
        assert total == ((n_out + 15) // 16) * ((k_in + 15) // 16) * 256
        out = np.full(total, np.nan, dtype=np.float32)
        rc = lib.l2a_pack_layer_host(w.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), k_in, n_out,
                                     out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
        assert rc == 0 and not np.isnan(out).any()
        KG = (k_in + 15) // 16
        pk = out.reshape(-1, KG, 64, 4)                       # [c, g, lane, ii]
        lane = np.arange(64)
        for c in range(pk.shape[0]):
            for g in range(KG):
                for ii in range(4):
                    k = 16 * g + 4 * (lane >> 4) + ii
                    u = 16 * c + (lane & 15)
                    ok = (k < k_in) & (u < n_out)
                    want = np.where(ok, w[np.minimum(k, k_in - 1), np.minimum(u, n_out - 1)], 0.0)
                    assert np.array_equal(pk[c, g, :, ii], want.astype(np.float32))
        assert np.isclose(np.abs(out).sum(), np.abs(w).sum(), rtol=1e-5)   # every weight exactly once


def test_mfma_eligibility():
    lib = _lib.load()

    def ok(obs, act, hidden):
        arr = (ctypes.c_int * len(hidden))(*hidden)
        return bool(lib.l2a_mfma_eligible(obs, act, len(hidden), arr))

    assert ok(20, 6, [512, 512]) and ok(41, 8, [512, 512, 512]) and ok(23, 7, [128]) and ok(64, 16, [256] * 8)
    assert not ok(20, 6, [200, 72]) and not ok(20, 6, [512, 256]) and not ok(65, 6, [512]) and not ok(20, 17, [512])
