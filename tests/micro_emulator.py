"""NumPy emulation of ONE horizon step of ``learning_to_adapt_amd/csrc/l2a_micro.h`` (``l2a_mlp_micro_body``) for one
micro tile of four candidates - developer documentation that runs: it consumes the micro-tile kernels' own weight layout
(wave-stream order, produced by the library's host packer ``l2a_micro_pack_layer_host``) and mirrors the kernel's index
algebra: records of four chain positions, ``v_mfma_f32_4x4x1_16b_f32`` as sixteen 4 x 4 outer products (one fma per
output), D registers = four consecutive chain positions of the next layer's row, the wave's two chunks of the canonical
output-layer reduce, the O4 quarter sums.  ``tests/test_micro_emulation.py`` compares it BIT FOR BIT with the lane-level
emulation of the 16-candidate kernel (``tests/mfma_emulator.py``): the claim the GPU tests then confirm on the chip."""

import ctypes

import numpy as np

F32 = np.float32
LANE = np.arange(64)
BLK = LANE >> 2           # block of the 4x4x1 MFMA = output slot group
J = LANE & 3              # candidate of the micro tile


def chain_k(p):
    """chain position <-> feature inside a 16-feature k-group (l2a_chain_k: swaps the two 2-bit fields, an involution)"""
    p = np.asarray(p)
    return (p & ~15) | ((p & 3) << 2) | ((p >> 2) & 3)


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F32)


def mfma_4x4x1(a, b, acc):
    """lane 4 blk + i supplies A[i], lane 4 blk + j supplies B[j]; lane (blk, j) holds D[i][j] in register i"""
    out = np.array(acc, dtype=F32)
    for i in range(4):
        out[:, i] = fma(a[4 * BLK + i], b, out[:, i])
    return out


class MicroSet(object):
    """One weight set in the micro-tile layout + its constants as the kernel caches them in LDS."""

    def __init__(self, lib, params, norm, obs_dim, act_dim):
        self.n_hidden = len(params) // 2 - 1
        self.H = params[0].shape[1]
        self.obs_dim, self.act_dim = obs_dim, act_dim
        in_dim = obs_dim + act_dim
        self.KG0 = (in_dim + 15) // 16
        self.KG0E = (self.KG0 + 1) & ~1
        self.OT = (obs_dim + 15) // 16
        self.o4 = (self.OT == 2 and self.KG0 == 2 and self.n_hidden > 1 and obs_dim - 16 <= 4)
        n = lib.l2a_micro_layout_floats(obs_dim, act_dim, self.n_hidden, self.H)
        assert n > 0, "no micro-tile instance for this shape"
        self.nrec = 4 * self.KG0E + (self.n_hidden - 1) * (self.H // 4) + 16
        assert n == (self.H // 64) * self.nrec * 256
        buf = np.zeros(n, dtype=F32)
        fp = ctypes.POINTER(ctypes.c_float)
        for l in range(self.n_hidden + 1):
            w = np.ascontiguousarray(params[2 * l], dtype=F32)
            rc = lib.l2a_micro_pack_layer_host(w.ctypes.data_as(fp), obs_dim, act_dim, self.n_hidden, self.H, l,
                                               buf.ctypes.data_as(fp))
            assert rc == 0
        self.rec = buf.reshape(self.H // 64, self.nrec, 64, 4)          # [stream T][record][lane][e]
        # every weight must have landed in exactly one place
        nz = sum(int(np.count_nonzero(np.asarray(params[2 * l]))) for l in range(self.n_hidden + 1))
        assert int(np.count_nonzero(buf)) == nz
        # hidden biases in SLOT order (slot u of its 64-unit tile is unit chain_k(u)), output constants padded to 64
        self.bias = [np.asarray(params[2 * l + 1], dtype=F32)[chain_k(np.arange(self.H))] for l in range(self.n_hidden)]
        eps = 1e-10
        self.in_mu = np.zeros(16 * self.KG0E, dtype=F32)
        self.in_iv = np.zeros(16 * self.KG0E, dtype=F32)
        self.in_mu[:in_dim] = np.concatenate([norm["obs"][0], norm["act"][0]]).astype(F32)
        self.in_iv[:in_dim] = (1.0 / (np.concatenate([norm["obs"][1], norm["act"][1]]) + eps)).astype(F32)
        self.out_mu = np.zeros(64, dtype=F32)
        self.out_sd = np.zeros(64, dtype=F32)
        self.b_out = np.zeros(64, dtype=F32)
        self.out_mu[:obs_dim] = norm["delta"][0].astype(F32)
        self.out_sd[:obs_dim] = (norm["delta"][1] + eps).astype(F32)
        self.b_out[:obs_dim] = np.asarray(params[2 * self.n_hidden + 1], dtype=F32)


def _phase(ms, stream, rec0, nrec, brow):
    """acc[lane][i] over `nrec` records of `stream` starting at record `rec0`; brow: [4 candidates][>= 4 nrec] activations in
    chain order.  One 4x4x1 MFMA per chain position, in order - the k-ordered fma chain of the 16-candidate kernels."""
    acc = np.zeros((64, 4), dtype=F32)
    for r in range(nrec):
        a = ms.rec[stream, rec0 + r]            # [lane][e]
        for e in range(4):
            acc = mfma_4x4x1(a[:, e], brow[J, 4 * r + e], acc)
    return acc


def _act(v, kind):
    if kind == "relu":
        return np.maximum(v, F32(0))
    if kind in (None, "identity"):
        return v
    if kind == "tanh":
        return np.tanh(v).astype(F32)
    raise ValueError(kind)


def micro_step(sets, mode, env, state, actions_row, hidden_act="relu"):
    """New states [4, obs_dim] of the four candidates of one micro tile after one step.  ``state``: [4, obs_dim] fp32,
    ``actions_row``: [4, act_dim] fp32.  ``sets``: MicroSet per member (mean) / per env (per_block)."""
    ms0 = sets[0]
    H, KG0E, n_hidden, obs_dim, act_dim = ms0.H, ms0.KG0E, ms0.n_hidden, ms0.obs_dim, ms0.act_dim
    UW = H // 256
    e_loop = len(sets) if mode == "mean" else 1
    e_half = (e_loop + 1) >> 1
    # state in the kernel's registers: lane (blk, j) holds dims 4 blk .. 4 blk + 3 of candidate j
    st = np.zeros((64, 4), dtype=F32)
    for i in range(4):
        dim = 4 * BLK + i
        st[:, i] = np.where(dim < obs_dim, state[J, np.minimum(dim, obs_dim - 1)], F32(0))
    dsum = np.zeros((64, 4), dtype=F32)
    dgrp = np.zeros((64, 4), dtype=F32)
    for i_set in range(e_loop):
        ms = sets[env] if mode == "per_block" else sets[i_set]
        # input rows in chain order: x[chain_k(k)] = ((s + 0) - mu[k]) * iv[k]
        x = np.zeros((4, 16 * KG0E), dtype=F32)
        for k in range(obs_dim):
            x[:, chain_k(k)] = ((state[:, k] + F32(0)) - ms.in_mu[k]) * ms.in_iv[k]
        for ka in range(act_dim):
            k = obs_dim + ka
            x[:, chain_k(k)] = ((F32(0) + actions_row[:, ka]) - ms.in_mu[k]) * ms.in_iv[k]
        x = x.astype(F32)
        rows = x
        rec0 = 0
        for l in range(n_hidden):
            nrec = 4 * KG0E if l == 0 else H // 4
            new = np.zeros((4, H), dtype=F32)
            for T in range(H // 64):
                acc = _phase(ms, T, rec0, nrec, rows)
                v = _act(acc + ms.bias[l][64 * T + 4 * BLK[:, None] + np.arange(4)[None, :]], hidden_act)
                # D registers of lane (blk, j) = slots 4 blk .. 4 blk + 3 = four consecutive chain positions of row j
                for i in range(4):
                    new[J, 64 * T + 4 * BLK + i] = v[:, i]
            rows = new
            rec0 += nrec
        # output layer: wave w owns the hidden units of chunks 2 w, 2 w + 1 = streams [w UW, (w + 1) UW)
        part = []
        for w in range(4):
            chunks = []
            if UW == 2:
                for tl in range(2):
                    T = 2 * w + tl
                    chunks.append(_phase(ms, T, rec0, 16, rows[:, 64 * T:64 * T + 64]))
            else:
                for sq in range(2):
                    chunks.append(_phase(ms, w, rec0 + 8 * sq, 8, rows[:, 64 * w + 32 * sq:64 * w + 32 * sq + 32]))
            part.append((chunks[0] + chunks[1]).astype(F32))
        s = ((part[0] + part[1]).astype(F32) + (part[2] + part[3]).astype(F32)).astype(F32)
        if ms.o4:
            # blocks 4 .. 7 hold dims 16 .. 19 summed over the quarters of the hidden units: (Q0 + Q1) + (Q2 + Q3) -> block 4
            t = s.copy()
            for i in range(4):
                q = [s[16 + 4 * qq + J, i] for qq in range(4)]                 # lane of block 4 + qq, candidate j
                tot = ((q[0] + q[1]).astype(F32) + (q[2] + q[3]).astype(F32)).astype(F32)
                t[(BLK >= 4) & (BLK < 8), i] = tot[(BLK >= 4) & (BLK < 8)]
            s = t
        slot = 4 * BLK[:, None] + np.arange(4)[None, :]
        s = (s + ms.b_out[slot]).astype(F32)                                    # (identity output layer)
        if i_set == e_half:
            dsum, dgrp = dgrp, np.zeros_like(dgrp)
        dgrp = (dgrp + fma(s, ms.out_sd[slot], ms.out_mu[slot])).astype(F32)
    d = (dsum + dgrp).astype(F32)
    if e_loop > 1:
        d = (d / F32(e_loop)).astype(F32)       # (the kernel's Markstein sequence is the correctly rounded quotient)
    nx = (st + d).astype(F32)
    out = np.zeros((4, obs_dim), dtype=F32)
    for i in range(4):
        dim = 4 * BLK + i
        ok = dim < obs_dim
        out[J[ok], dim[ok]] = nx[ok, i]
    return out
