"""Lane-level NumPy emulation of ``learning_to_adapt_amd/csrc/l2a_mfma.h`` (one workgroup).

Purpose: validate, WITHOUT a GPU, the index algebra the MFMA kernel rests on - the packed
weight layout (produced by the library's own host packer, ``l2a_pack_layer_host``), the
``v_mfma_f32_16x16x4_f32`` operand/result lane maps, the D-fragment == next-layer B-fragment
chaining, the input-fragment assembly from state + actions, the K-split of the output layer
and the reward lane ownership.  It mirrors the kernel statement by statement (same names);
waves run one after the other between barriers, so it cannot see races - those are argued in
the kernel comments.

Lane maps (cdna_hip_programming.md section 3): for ``mfma_f32_16x16x4f32(a, b, c)`` lane ``l``
supplies ``A[i = l & 15][k = l >> 4]`` and ``B[k = l >> 4][j = l & 15]`` and holds
``D[row = 4 (l >> 4) + reg][col = l & 15]`` in its 4 result registers.
"""

import numpy as np

NW = 4          # waves per workgroup (L2A_NW in csrc/l2a_mfma.h)
LANE = np.arange(64)
JC = LANE & 15
QQ = LANE >> 4
F32 = np.float32


def mfma_16x16x4(a, b, c):
    """a, b: [64] fp32 (one VGPR each); c: [64, 4] fp32 accumulator.  Returns new accumulator."""
    A = np.zeros((16, 4), dtype=F32)
    B = np.zeros((4, 16), dtype=F32)
    A[LANE & 15, LANE >> 4] = a
    B[LANE >> 4, LANE & 15] = b
    out = np.array(c, dtype=F32)
    for reg in range(4):
        rowi = 4 * QQ + reg
        acc = out[:, reg]
        for k in range(4):      # k-ordered fma chain, one rounding per product-add
            acc = (A[rowi, k].astype(np.float64) * B[k, JC].astype(np.float64) + acc.astype(np.float64)).astype(F32)
        out[:, reg] = acc
    return out


def mfma_4x4x1(a, b, c):
    """``v_mfma_f32_4x4x1_16b_f32``: sixteen 4x4 outer products, block ``l >> 2``: lane ``4 blk + i`` supplies ``A[i]``,
    lane ``4 blk + j`` supplies ``B[j]`` and holds ``D[i][j]`` in result register ``i``."""
    out = np.array(c, dtype=F32)
    blk = LANE >> 2
    for i in range(4):
        out[:, i] = (a[4 * blk + i].astype(np.float64) * b.astype(np.float64) + out[:, i].astype(np.float64)).astype(F32)
    return out


def fma(a, b, c):
    """One rounding: what the compiler's contraction of ``a * b + c`` (and an explicit ``fmaf``) gives on the device."""
    return (np.asarray(a, dtype=np.float64) * np.asarray(b, dtype=np.float64) + np.asarray(c, dtype=np.float64)).astype(F32)


def act4(v, kind):
    if kind == "relu":
        return np.maximum(v, F32(0))
    if kind in (None, "identity"):
        return v
    if kind == "tanh":
        return np.tanh(v).astype(F32)
    if kind == "sigmoid":
        return (F32(1) / (F32(1) + np.exp(-v))).astype(F32)
    if kind == "swish":
        return (v / (F32(1) + np.exp(-v))).astype(F32)
    raise ValueError(kind)


class PackedSet(object):
    """One weight set in the device layout: packed layers (via the library's host packer),
    biases, padded normalisation vectors."""

    def __init__(self, lib, params, norm, obs_dim, act_dim):
        import ctypes
        self.n_hidden = len(params) // 2 - 1
        self.H = params[0].shape[1]
        in_dim = obs_dim + act_dim
        self.KG0 = (in_dim + 15) // 16
        self.OT = (obs_dim + 15) // 16
        self.HT = self.H // 16

        def pack(w):
            w = np.ascontiguousarray(w, dtype=F32)
            n = lib.l2a_packed_layer_floats(w.shape[0], w.shape[1])
            out = np.empty(n, dtype=F32)
            rc = lib.l2a_pack_layer_host(w.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), w.shape[0],
                                         w.shape[1], out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
            assert rc == 0
            return out.reshape(-1, 64, 4)     # [(c * KG + g), lane, ii]

        self.w0 = pack(params[0])
        self.b0 = np.asarray(params[1], dtype=F32)
        self.wmid = [pack(params[2 * l]) for l in range(1, self.n_hidden)]
        self.bmid = [np.asarray(params[2 * l + 1], dtype=F32) for l in range(1, self.n_hidden)]
        self.wout = pack(params[2 * self.n_hidden])
        bout = np.zeros(16 * self.OT, dtype=F32)
        bout[:obs_dim] = params[2 * self.n_hidden + 1]
        self.bout = bout
        eps = 1e-10
        self.in_mu = np.zeros(16 * self.KG0, dtype=F32)
        self.in_iv = np.zeros(16 * self.KG0, dtype=F32)
        mu = np.concatenate([norm["obs"][0], norm["act"][0]])
        sd = np.concatenate([norm["obs"][1], norm["act"][1]])
        self.in_mu[:in_dim] = mu.astype(F32)
        self.in_iv[:in_dim] = (1.0 / (sd + eps)).astype(F32)
        self.out_mu = np.zeros(16 * self.OT, dtype=F32)
        self.out_sd = np.zeros(16 * self.OT, dtype=F32)
        self.out_mu[:obs_dim] = norm["delta"][0].astype(F32)
        self.out_sd[:obs_dim] = (norm["delta"][1] + eps).astype(F32)


def rollout_workgroup(sets, mode, obs0_env, actions, env, tb, n, m, obs_dim, act_dim, discount,
                      reward, hidden_act="relu", output_act=None, NT=1):
    """Emulate workgroup (env, tb).  ``actions``: fp32 [h, m*n, act_dim].  ``reward``: dict with the
    l2a_reward fields.  Returns (ret[NT, 16], valid[NT, 16], final_state[NT, 16, obs_dim])."""
    ps0 = sets[0]
    HT, KG0, OT = ps0.HT, ps0.KG0, ps0.OT
    TPW = HT // NW
    h = actions.shape[0]
    R = m * n
    e_loop = len(sets) if mode == "mean" else 1

    cand = np.zeros((NT, 64), dtype=np.int64)
    valid = np.zeros((NT, 64), dtype=bool)
    row = np.zeros((NT, 64), dtype=np.int64)
    for nt in range(NT):
        cand[nt] = tb * (16 * NT) + nt * 16 + JC
        valid[nt] = cand[nt] < n
        row[nt] = env * n + np.where(valid[nt], cand[nt], n - 1)

    # every wave keeps its own copy of the state; emulate all 8 and assert they stay identical
    st = np.zeros((NW, NT, OT, 64, 4), dtype=F32)
    for nt in range(NT):
        for c in range(OT):
            for ii in range(4):
                dim = 16 * c + 4 * QQ + ii
                v = obs0_env[np.minimum(dim, obs_dim - 1)].astype(F32)
                st[:, nt, c, :, ii] = np.where(dim < obs_dim, v, F32(0))

    ga0 = obs_dim >> 4

    def load_actions(t):
        dst = np.zeros((NT, 2, 64, 4), dtype=F32)
        for nt in range(NT):
            for s in range(2):
                for ii in range(4):
                    ka = 16 * (ga0 + s) + 4 * QQ + ii - obs_dim
                    ok = (ka >= 0) & (ka < act_dim)
                    v = actions[t, row[nt], np.where(ok, ka, 0)]
                    dst[nt, s, :, ii] = np.where(ok, v, F32(0))
        return dst

    # O4 (csrc/l2a_mfma.h, l2a_out_phase): the last obs tile has at most four live units -> computed with the 4x4x1 MFMA,
    # every lane accumulating over ITS quarter of the hidden units; the quarters are added after the chunk reduce
    O4 = (OT == 2 and KG0 == 2 and ps0.n_hidden > 1 and 1 <= obs_dim - 16 <= 4)
    LANE4 = (LANE & 48) | (LANE & 3)
    ret = np.zeros((NW, NT, 64), dtype=F32)
    sa = max(NT * HT, 2 * NW * NT * OT)
    lds = [np.zeros((sa, 64, 4), dtype=F32), np.zeros((sa, 64, 4), dtype=F32)]
    cur = 0
    disc_pow = 1.0

    for t in range(h):
        av = load_actions(t)
        asq = np.zeros((NT, 64), dtype=F32)
        for nt in range(NT):
            s = np.zeros(64, dtype=F32)
            for ii in range(4):
                s = fma(av[nt, 0, :, ii], av[nt, 0, :, ii], s)
                s = fma(av[nt, 1, :, ii], av[nt, 1, :, ii], s)
            asq[nt] = s
        dsum = np.zeros((NW, NT, OT, 64, 4), dtype=F32)      # finished group
        dgrp = np.zeros((NW, NT, OT, 64, 4), dtype=F32)      # group being summed (A then B)
        e_half = (e_loop + 1) >> 1

        for e in range(e_loop):
            if e == e_half:
                dsum, dgrp = dgrp, np.zeros_like(dgrp)
            ps = sets[env] if mode == "per_block" else sets[e]
            hcur, hoth = lds[cur], lds[cur ^ 1]
            # ---- layer 0 ----
            for wave in range(NW):
                c0 = wave * TPW
                acc = np.zeros((NT, TPW, 64, 4), dtype=F32)
                for g in range(KG0):
                    mu = ps.in_mu[16 * g + 4 * QQ[:, None] + np.arange(4)[None, :]]
                    iv = ps.in_iv[16 * g + 4 * QQ[:, None] + np.arange(4)[None, :]]
                    for nt in range(NT):
                        sv = st[wave, nt, g] if g < OT else np.zeros((64, 4), dtype=F32)
                        aa = np.zeros((64, 4), dtype=F32)
                        if g == ga0:
                            aa = av[nt, 0]
                        if g == ga0 + 1:
                            aa = av[nt, 1]
                        k = 16 * g + 4 * QQ[:, None] + np.arange(4)[None, :]
                        v = np.where(k < obs_dim, sv, aa)
                        x = ((v - mu) * iv).astype(F32)
                        for ii in range(4):
                            for tt in range(TPW):
                                a = ps.w0[(c0 + tt) * KG0 + g, :, ii]
                                acc[nt, tt] = mfma_16x16x4(a, x[:, ii], acc[nt, tt])
                for tt in range(TPW):
                    bias = ps.b0[16 * (c0 + tt) + 4 * QQ[:, None] + np.arange(4)[None, :]]
                    for nt in range(NT):
                        hcur[nt * HT + c0 + tt] = act4((acc[nt, tt] + bias).astype(F32), hidden_act)
            # barrier
            # ---- hidden layers ----
            for l in range(1, ps.n_hidden):
                wl = ps.wmid[l - 1]
                bl = ps.bmid[l - 1]
                for wave in range(NW):
                    c0 = wave * TPW
                    acc = np.zeros((NT, TPW, 64, 4), dtype=F32)
                    for g in range(HT):
                        for ii in range(4):
                            for nt in range(NT):
                                b = hcur[nt * HT + g][:, ii]
                                for tt in range(TPW):
                                    a = wl[(c0 + tt) * HT + g, :, ii]
                                    acc[nt, tt] = mfma_16x16x4(a, b, acc[nt, tt])
                    for tt in range(TPW):
                        bias = bl[16 * (c0 + tt) + 4 * QQ[:, None] + np.arange(4)[None, :]]
                        for nt in range(NT):
                            hoth[nt * HT + c0 + tt] = act4((acc[nt, tt] + bias).astype(F32), hidden_act)
                # barrier + swap
                cur ^= 1
                hcur, hoth = lds[cur], lds[cur ^ 1]
            # ---- output layer: K cut into 2 * NW chunks of TPW / 2 k-groups, summed as a balanced tree (canonical order) ----
            CS = TPW // 2
            for wave in range(NW):
                for ch in range(2):
                    acc = np.zeros((NT, OT, 64, 4), dtype=F32)
                    for t2 in range(CS):
                        g = wave * TPW + ch * CS + t2
                        for ii in range(4):
                            for nt in range(NT):
                                b = hcur[nt * HT + g][:, ii]
                                for c in range(OT):
                                    if O4 and c == OT - 1:
                                        acc[nt, c] = mfma_4x4x1(ps.wout[c * HT + g, LANE4, ii], b, acc[nt, c])
                                    else:
                                        a = ps.wout[c * HT + g, :, ii]
                                        acc[nt, c] = mfma_16x16x4(a, b, acc[nt, c])
                    for nt in range(NT):
                        for c in range(OT):
                            # one partial per wave: chunk 2 wave + chunk 2 wave + 1 (first level of the canonical tree)
                            k = (wave * NT + nt) * OT + c
                            hoth[k] = acc[nt, c] if ch == 0 else (np.array(hoth[k]) + acc[nt, c]).astype(F32)
            # barrier
            for wave in range(NW):
                for c in range(OT):
                    bias = ps.bout[16 * c + 4 * QQ[:, None] + np.arange(4)[None, :]]
                    omu = ps.out_mu[16 * c + 4 * QQ[:, None] + np.arange(4)[None, :]]
                    osd = ps.out_sd[16 * c + 4 * QQ[:, None] + np.arange(4)[None, :]]
                    for nt in range(NT):
                        part = [np.array(hoth[(w * NT + nt) * OT + c]) for w in range(NW)]
                        s = ((part[0] + part[1]).astype(F32) + (part[2] + part[3]).astype(F32)).astype(F32)
                        if O4 and c == OT - 1:
                            s = (s + s[LANE ^ 16]).astype(F32)
                            s = (s + s[LANE ^ 32]).astype(F32)
                        s = act4((s + bias).astype(F32), output_act)
                        dgrp[wave, nt, c] = (dgrp[wave, nt, c] + fma(s, osd, omu)).astype(F32)
            # (no barrier; `cur` is NOT flipped: the next layer 0 writes the region it just read)

        dsum = (dsum + dgrp).astype(F32)                     # group A + group B
        disc_t = F32(disc_pow)
        disc_pow *= float(F32(discount))
        for wave in range(NW):
            for nt in range(NT):
                plin = fma(-F32(reward["ctrl_coef"]), asq[nt], np.where(QQ == 0, F32(reward["alive"]), F32(0)))
                psq = np.zeros(64, dtype=F32)
                for c in range(OT):
                    d = dsum[wave, nt, c]
                    if e_loop > 1:
                        d = (d / F32(e_loop)).astype(F32)       # (the kernel's Markstein sequence is the correctly rounded quotient)
                    nx = (st[wave, nt, c] + d).astype(F32)
                    for ii in range(4):
                        dim = 16 * c + 4 * QQ + ii
                        plin = np.where(dim == reward["vel_index"],
                                        fma((F32(reward["w_vel"]) * d[:, ii]).astype(F32), F32(reward["inv_dt"]), plin), plin).astype(F32)
                        in_dist = (reward["dist_coef"] != 0.0) & (dim >= reward["dist_index"]) & \
                                  (dim < reward["dist_index"] + 3) & (dim < obs_dim)
                        psq = (psq + np.where(in_dist, nx[:, ii] * nx[:, ii], F32(0))).astype(F32)
                    st[wave, nt, c] = nx
                plin = (plin + plin[LANE ^ 16]).astype(F32)
                plin = (plin + plin[LANE ^ 32]).astype(F32)
                psq = (psq + psq[LANE ^ 16]).astype(F32)
                psq = (psq + psq[LANE ^ 32]).astype(F32)
                r = plin
                if reward["dist_coef"] != 0.0:
                    r = fma(-F32(reward["dist_coef"]), np.sqrt(psq), r)
                ret[wave, nt] = fma(disc_t, r, ret[wave, nt])

    for wave in range(1, NW):
        assert np.array_equal(st[wave], st[0]), "waves diverged"
        assert np.array_equal(ret[wave], ret[0])

    out_ret = np.zeros((NT, 16), dtype=F32)
    out_valid = np.zeros((NT, 16), dtype=bool)
    out_state = np.zeros((NT, 16, obs_dim), dtype=F32)
    for nt in range(NT):
        out_ret[nt] = ret[0, nt, :16]          # lanes with qq == 0
        out_valid[nt] = valid[nt, :16]
        for c in range(OT):
            for ii in range(4):
                for lane in range(64):
                    dim = 16 * c + 4 * (lane >> 4) + ii
                    if dim < obs_dim:
                        out_state[nt, lane & 15, dim] = st[0, nt, c, lane, ii]
    return out_ret, out_valid, out_state
