// l2a_rnn_micro.h - MICRO TILES for the stacked recurrent cells (GRU, BasicRNN, LSTM stacks: everything `create_rnn`,
// reference dynamics/core/utils.py:192-236, builds besides run_rebal.py's single LSTM layer, which has l2a_lstm_micro_k).
//
// The 16-candidate kernel of these models (l2a_rnn_mfma.h) keeps one workgroup per 16 candidates: the reference's default
// plan (run_rebal.py:77-78: 5 x 500 candidates = 160 tiles) leaves 96 of 256 CUs idle, and every product call of a wave
// starts its operand stream from nothing.  Here a workgroup owns MT = 1 .. 3 candidate tiles of FOUR on
// v_mfma_f32_4x4x1_16b_f32 (l2a_micro.h: operands, chain order, launch geometry), wave w owns the 64-unit tiles
// [w UW, (w + 1) UW) of EVERY layer (UW = U / 256; all layers U units wide), all gates of a unit in one wave:
//
//   per layer   product 0 = [input | old h] x the cell's first kernel: 4 (LSTM: i, j, f, o), 2 (GRU: r, u) or 1 (BasicRNN) gate
//               tiles per unit tile, accumulators in registers, gate arithmetic register local;
//               GRU: r * h -> LDS, barrier, product 1 = [input | r * h] x candidate kernel, u and the old h stay in registers;
//               the new h goes to the LDS rows of the other step parity (the layer above reads it as its input after one
//               barrier, this layer as its old h in the next step); an LSTM layer's c lives in LDS rows only its owner touches;
//   operands    A: per layer and product [64-unit tile][gate][k-group of four chain positions][lane][4] (l2a_rnn_micro_pack_k),
//               the input part first, then the recurrent part - the K order of l2a_rnn_mfma.h, so the hidden layers sum in the
//               same order as the 16-candidate kernel; three k-groups are requested ahead, the first three of the NEXT product
//               right behind a product's last MFMA, so that they travel under the gate arithmetic and the barrier;
//               B: ds_read_b128 of the candidate's row (four chain positions, the same in all sixteen blocks);
//   output      every wave multiplies its own units of the top layer's new h (no barrier in front), the partials meet in LDS,
//               (p0 + p1) + (p2 + p3); reduce, reward, state, return and the next step's input rows by wave c for micro tile c
//               (l2a_lstm_micro_body's tail).
#pragma once

#include <type_traits>

#include "l2a_micro.h"

template <int MT, int UW, int CELL, int MTM>
__device__ __forceinline__ void l2a_rnn_micro_body(const L2ALstmParams& p, const int env, const int cand0, char* smem) {
    constexpr int U = 256 * UW;
    constexpr int HR = l2a_rnn_micro_row(U);
    constexpr int XR = HR;
    constexpr bool LSTM = CELL == L2A_CELL_LSTM, GRU = CELL == L2A_CELL_GRU;
    constexpr int G0 = LSTM ? 4 : (GRU ? 2 : 1);
    constexpr int NTLM = G0 * UW;           // accumulator tiles of a wave in product 0 (product 1 of a GRU layer: UW)
    constexpr int NBIAS = (LSTM ? 4 : (GRU ? 3 : 1)) * U;
    constexpr int NR = 4 * MTM;             // candidate rows the LDS arrays are laid out for (MTM = the launch's largest workgroup: 3 or 4 micro tiles)
    constexpr int LROWS = (CELL == L2A_CELL_RNN ? 2 : 3) * NR * HR;
    constexpr int HIT = U / 16;             // loop iterations (four k-groups = sixteen features each) of a recurrent part
    constexpr int NGO = 16 * UW;            // output-layer k-groups of this wave's units
    const int KG0 = p.KG0, L = p.n_layers;
    const int KGX0 = l2a_rnn_micro_kgx(p.in_dim);      // layer 0's input iterations (sixteen features each), padded to even with zeros
    float* xs = reinterpret_cast<float*>(smem);                     // [12][XR]
    float* lrows = xs + NR * XR;
    f32x4* pbuf = reinterpret_cast<f32x4*>(lrows + L * LROWS);      // [4 waves][MTM][64]
    float* c_in_mu = reinterpret_cast<float*>(pbuf + 4 * MTM * 64);
    float* c_in_iv = c_in_mu + 16 * KG0;
    float* c_out_mu = c_in_iv + 16 * KG0;   // [64] each, zero past the observation
    float* c_out_sd = c_out_mu + 64;
    float* c_bo = c_out_sd + 64;
    float* c_gb = c_bo + 64;                // per layer: biases in slot order [gate][U] (GRU: r, u, candidate)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = lane >> 2, j = lane & 3;
    const int qq = b & 3;
    const int obs_dim = p.obs_dim, act_dim = p.act_dim;
    const int R = p.m * p.n;

    for (int i = tid; i < 32 * KG0 + 192 + L * NBIAS; i += 256) {
        float v;
        if (i < 32 * KG0) v = p.wblk[p.nm_off + i];
        else if (i < 32 * KG0 + 192) {
            const int a = (i - 32 * KG0) >> 6, d = (i - 32 * KG0) & 63;
            v = (d >= obs_dim) ? 0.0f : (a == 0 ? p.wblk[p.nm_off + 32 * KG0 + d]
                                                : (a == 1 ? p.wblk[p.nm_off + 32 * KG0 + 16 * p.OT + d] : p.wblk[p.raw_bo + d]));
        } else {
            const int o = i - (32 * KG0 + 192), l = o / NBIAS, r = o - l * NBIAS, q = r / U, s = r - q * U;
            // slot s of tile s / 64 is unit chain_k(s); a GRU layer's candidate bias is its second bias vector
            v = (GRU && q == 2) ? p.wblk[p.layer_b[l][1] + l2a_chain_k(s)] : p.wblk[p.layer_b[l][0] + q * U + l2a_chain_k(s)];
        }
        c_in_mu[i] = v;
    }
    for (int i = tid; i < NR * XR + L * LROWS; i += 256) xs[i] = 0.0f;   // padding (and rows of absent micro tiles) stay zero
    __syncthreads();

    int cand[MT], row[MT];
    bool valid[MT];
#pragma unroll
    for (int c = 0; c < MT; ++c) {
        cand[c] = cand0 + 4 * c + j;
        valid[c] = cand[c] < p.n;
        row[c] = env * p.n + (valid[c] ? cand[c] : p.n - 1);
    }
    const int ct = wave < MT ? wave : MT - 1;           // the micro tile whose per-candidate work this wave does (l2a_lstm_micro_body)
    const int cand_t = cand0 + 4 * ct + j;
    const bool valid_t = cand_t < p.n;
    const int row_t = env * p.n + (valid_t ? cand_t : p.n - 1);
    f32x4 st;
    {
        const float* orow = p.obs0 + (long long)env * obs_dim;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int dim = 4 * b + i;
            const float v = orow[dim < obs_dim ? dim : obs_dim - 1];
            st[i] = (dim < obs_dim) ? v : 0.0f;
        }
    }
    // hidden state: this wave's units of every layer -> the rows of step parity 0 (c: the layer's third array)
    for (int l = 0; l < L; ++l) {
        float* hl = lrows + l * LROWS;
#pragma unroll
        for (int c = 0; c < MT; ++c) {
            const long long hrow = (p.hid_per_row ? (long long)row[c] : (long long)env) * p.units + (long long)l * U;
#pragma unroll
            for (int uw = 0; uw < UW; ++uw) {
                const int tile = wave * UW + uw;
                f32x4 hv, cv;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int unit = 64 * tile + l2a_chain_k(4 * b + i);
                    hv[i] = p.h0[hrow + unit];
                    cv[i] = LSTM ? p.c0[hrow + unit] : 0.0f;
                }
                *reinterpret_cast<f32x4*>(hl + (4 * c + j) * HR + 64 * tile + 4 * b) = hv;
                if (LSTM) *reinterpret_cast<f32x4*>(hl + 2 * NR * HR + (4 * c + j) * HR + 64 * tile + 4 * b) = cv;
            }
        }
    }

    // raw actions / input rows: l2a_lstm_micro_body (lanes b < 4 play the 16-candidate kernels' quarter role)
    const int ga0 = obs_dim >> 4;
    f32x4 av_next[2];
    int aoff[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const int ka = 16 * (ga0 + s) + 4 * qq + ii - obs_dim;
            const bool in = (b < 4) && (ka >= 0) && (ka < act_dim);
            aoff[s][ii] = in ? (row_t * act_dim + ka) * 4 : 0x7ffffff0;
        }
    const long long a_step = (long long)R * act_dim;
    auto load_actions = [&](int t, f32x4 (&dst)[2]) {
        const __amdgpu_buffer_rsrc_t ars = l2a_rsrc(p.actions + (long long)t * a_step, a_step * 4);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
                dst[s][ii] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ars, aoff[s][ii], 0, 0));
    };
    f32x4 av[2];
    float asq;
    int xo_s[4], xo_a[2][4];
    f32x4 mu_s, iv_s, mu_a[2], iv_a[2];
    {
        mu_s = *reinterpret_cast<const f32x4*>(c_in_mu + 4 * b);
        iv_s = *reinterpret_cast<const f32x4*>(c_in_iv + 4 * b);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = 4 * b + i;
            xo_s[i] = (k < obs_dim) ? l2a_chain_k(k) : 96 + (lane & 7);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            mu_a[s2] = *reinterpret_cast<const f32x4*>(c_in_mu + 16 * (ga0 + s2) + 4 * qq);
            iv_a[s2] = *reinterpret_cast<const f32x4*>(c_in_iv + 16 * (ga0 + s2) + 4 * qq);
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int k = 16 * (ga0 + s2) + 4 * qq + ii;
                xo_a[s2][ii] = (b < 4 && k >= obs_dim && k < obs_dim + act_dim) ? l2a_chain_k(k) : 96 + (lane & 7);
            }
        }
    }
    auto write_x = [&]() {
        float* xr = xs + (4 * ct + j) * XR;
#pragma unroll
        for (int i = 0; i < 4; ++i) xr[xo_s[i]] = (st[i] - mu_s[i]) * iv_s[i];
        float s = 0.0f;
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            s = fmaf(av[0][ii], av[0][ii], s);
            s = fmaf(av[1][ii], av[1][ii], s);
        }
        asq = s;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) xr[xo_a[s2][ii]] = (av[s2][ii] - mu_a[s2][ii]) * iv_a[s2][ii];
    };
    load_actions(0, av_next);
    av[0] = av_next[0]; av[1] = av_next[1];
    load_actions(p.h > 1 ? 1 : 0, av_next);
    write_x();

    float ret = p.ret_in ? p.ret_in[(long long)env * p.n + (valid_t ? cand_t : p.n - 1)] : 0.0f;
    double disc_pow = p.disc0;

    // ---- operand streams ---------------------------------------------------------------------------------------------
    const __amdgpu_buffer_rsrc_t rsO = l2a_rsrc(p.wblk + p.pk_mo, (long long)(U / 4) * 1024);
    const int voffO = lane * 16 + wave * NGO * 1024;
    // weight ring: RD k-groups of a product's NTL tiles, RD - 1 requested ahead: eight deep.  With three k-groups ahead the products
    // with one or two gate tiles had 3 - 6 KiB per wave in flight against the ~8 KiB that 43 B/clock per CU times the L2's latency asks
    // for (81 - 88 % of their matrix time, timeline r04); an LSTM layer's four tiles had 12 KiB and still gained 1.2 - 1.8 % from 28
    // (a stack streams 3.3 MB per step through a 4 MB L2: some k-groups come from further away)
    constexpr int RD0 = (NTLM >= 4 && !LSTM) ? 4 : 8;   // product 0 (512-unit GRU: four tiles of two gates - eight deep measured 0.3 % slower)
    constexpr int RD1 = 8;                              // a GRU layer's candidate product
    constexpr int RING = RD0 * NTLM;
    f32x4 ring[RING];
    f32x4 rb[2][MT];                                    // activation ring: two k-groups
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using NT0 = std::integral_constant<int, NTLM>; using NT1 = std::integral_constant<int, UW>;
    using RDT0 = std::integral_constant<int, RD0>; using RDT1 = std::integral_constant<int, RD1>;
    // a product's stream: resource over its packed array, this wave's tile offsets (tile tl = gate q of its unit tile uw: q UW + uw)
    struct Stream { __amdgpu_buffer_rsrc_t rs; int voff[NTLM]; };
    auto stream_of = [&](int l, auto prod_tag, auto ntl_tag) {
        constexpr int PROD = decltype(prod_tag)::value, NTL = decltype(ntl_tag)::value, G = NTL / UW;
        const int nkg = (l == 0 ? 4 * KGX0 : U / 4) + U / 4;
        Stream s;
        s.rs = l2a_rsrc(p.wblk + p.layer_mk[l][PROD], (long long)(U / 64) * G * nkg * 1024);
#pragma unroll
        for (int tl = 0; tl < NTLM; ++tl) {
            const int q = tl / UW, uw = tl - q * UW;
            s.voff[tl] = lane * 16 + (((wave * UW + uw) * G + q) * nkg) * 1024;
        }
        return s;
    };
    // the first RD - 1 k-groups of a product, requested in this order (pinned: the loops' s_waitcnt counts hold on every path)
    // (lo_tag .. hi_tag: the requests [lo, hi) of the (RD - 1) NTL, for an epilogue that issues them in portions between its chunks)
    auto prefetch_part = [&](const Stream& s, auto ntl_tag, auto lo_tag, auto hi_tag) {
        constexpr int NTL = decltype(ntl_tag)::value, LO = decltype(lo_tag)::value, HI = decltype(hi_tag)::value;
        __builtin_amdgcn_sched_barrier(0);
        l2a_static_for<LO, HI>([&](auto kv) {
            constexpr int k = decltype(kv)::value, g = k / NTL, tl = k % NTL;
            ring[g * NTL + tl] = l2a_ldw(s.rs, s.voff[tl] + (g % 4) * 1024, (g / 4) * 4096);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    auto prefetch = [&](const Stream& s, auto ntl_tag, auto rd_tag) {
        constexpr int NTL = decltype(ntl_tag)::value, RD = decltype(rd_tag)::value;
        static_assert(RD * NTL <= RING, "ring too small");
        prefetch_part(s, ntl_tag, I0(), std::integral_constant<int, (RD - 1) * NTL>());
    };
    // acc[tl][c] += the product over [nx iterations of sixteen input features from bx | HIT iterations of the recurrent part from bh]
    // (bx, bh: this lane's candidate row j of micro tile 0; the other micro tiles lie 4 HR floats apart); a loop trip = RD k-groups
    auto gemm = [&](const Stream& s, auto ntl_tag, auto rd_tag, const float* bx, int nx, const float* bh, f32x4 (&acc)[NTLM][MT]) {
        constexpr int NTL = decltype(ntl_tag)::value, RD = decltype(rd_tag)::value, IT = RD / 4;
        auto issue_b = [&](const float* bp, auto off_tag, auto slot_tag) {
            constexpr int sl = decltype(slot_tag)::value, OFF = decltype(off_tag)::value;
#pragma unroll
            for (int c = 0; c < MT; ++c) rb[sl][c] = *reinterpret_cast<const f32x4*>(bp + 4 * c * HR + OFF);
        };
        auto body = [&](auto last_tag, const float* bp, const float* bn, int s0) {
            constexpr bool LAST = decltype(last_tag)::value;
            l2a_static_for<0, RD>([&](auto iv) {
                constexpr int I = decltype(iv)::value;
                constexpr int GA = RD - 1 + I;          // the k-group requested now, counted from the trip's first
                if constexpr (I == 0 || !LAST) {
#pragma unroll
                    for (int tl = 0; tl < NTL; ++tl)
                        ring[(GA % RD) * NTL + tl] = l2a_ldw(s.rs, s.voff[tl] + (GA % 4) * 1024, s0 + (GA / 4) * 4096);
                }
                if constexpr (I < RD - 1) issue_b(bp, std::integral_constant<int, 4 * (I + 1)>(), std::integral_constant<int, (I + 1) & 1>());
                else if constexpr (!LAST) issue_b(bn, I0(), I0());
                // (the LAST-requested operands first: one s_waitcnt per kind and k-group, l2a_lstm_micro_body)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int tl = NTL - 1; tl >= 0; --tl)
#pragma unroll
                        for (int c = MT - 1; c >= 0; --c) acc[tl][c] = L2A_MFMA4(ring[I * NTL + tl][e], rb[I & 1][c][e], acc[tl][c]);
                constexpr int NA = (I == 0 || !LAST) ? NTL : 0;
                constexpr int NB = (I < RD - 1 || !LAST) ? MT : 0;
                l2a_micro_hint<NA, NB, 4 * NTL * MT>();
            });
        };
        issue_b(bx, I0(), I0());
        int s0 = 0;
        const int tx = nx / IT;
#pragma unroll 1
        for (int it = 0; it < tx; ++it, s0 += 1024 * RD)
            body(std::false_type(), bx + 4 * RD * it, (it + 1 < tx) ? bx + 4 * RD * (it + 1) : bh, s0);
#pragma unroll 1
        for (int it = 0; it < HIT / IT - 1; ++it, s0 += 1024 * RD) body(std::false_type(), bh + 4 * RD * it, bh + 4 * RD * (it + 1), s0);
        body(std::true_type(), bh + 4 * RD * (HIT / IT - 1), bh, s0);
        __builtin_amdgcn_sched_barrier(0);
    };
    struct ActTanh { __device__ __forceinline__ float operator()(float x) const { return l2a_fast_tanh(x); } };
    struct ActAny { int kind; __device__ __forceinline__ float operator()(float x) const { return l2a_act1(x, kind); } };
    // (the nonlinearity is chosen once per epilogue, not per element: l2a_lstm_micro_body)
    auto with_act = [&](auto&& f) { if (p.cell_act == L2A_ACT_TANH) f(ActTanh()); else f(ActAny{p.cell_act}); };
    f32x4 pfo[2][NGO / 2];                              // output-layer operands of this wave's units (requested under the last layer's gates)

    prefetch(stream_of(0, I0(), NT0()), NT0(), RDT0());
    __syncthreads();        // every wave's share of h(0) and every micro tile's input rows are in LDS

    for (int t = 0; t < p.h; ++t) {
        const int cur = t & 1;
        const float* xin = xs + j * HR;
        int nx = KGX0;
        L2A_MTS(0)
        // one layer; LASTL: the top layer - the next product is the next step's first, and the output layer's operands follow it
        auto layer = [&](const int l, auto last_tag) {
            constexpr bool LASTL = decltype(last_tag)::value;
            float* hl = lrows + l * LROWS;
            const float* hc = hl + cur * NR * HR;
            float* hn = hl + (cur ^ 1) * NR * HR;
            float* aux = hl + 2 * NR * HR;              // LSTM: c; GRU: r * h (this step's, every unit: the candidate product's B)
            const float* gb = c_gb + l * NBIAS;
            const int ln = LASTL ? 0 : l + 1;           // whose product 0 comes next
            // What the wave multiplies next is requested INSIDE the gate arithmetic, a portion in front of each of its UW MT chunks
            // (thirty 1 KiB requests in a row wait for queue space - the CU's four waves share one 64 B/clock path - with the VALU
            // idle, timeline r04): the next layer's first k-groups, or - top layer - the output layer's operands; the next STEP's
            // first k-groups follow the output product (they have the whole tail to arrive).
            constexpr int NLOAD = LASTL ? NGO : (RD0 - 1) * NTLM;
            constexpr int NPART = UW * MT;
            const Stream sn = stream_of(ln, I0(), NT0());
            auto next_operands = [&](auto part_tag) {
                constexpr int PART = decltype(part_tag)::value;
                constexpr int LO = NLOAD * PART / NPART, HI = NLOAD * (PART + 1) / NPART;
                if constexpr (LASTL) {
                    __builtin_amdgcn_sched_barrier(0);
                    l2a_static_for<LO, HI>([&](auto kv) {
                        constexpr int k = decltype(kv)::value;
                        pfo[k / (NGO / 2)][k % (NGO / 2)] = l2a_ldw(rsO, voffO + k * 1024, 0);
                    });
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    prefetch_part(sn, NT0(), std::integral_constant<int, LO>(), std::integral_constant<int, HI>());
                }
            };
            f32x4 acc[NTLM][MT];
#pragma unroll
            for (int tl = 0; tl < NTLM; ++tl)
#pragma unroll
                for (int c = 0; c < MT; ++c) acc[tl][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
            f32x4 hv[UW][MT];                           // GRU: the old h of this lane's units
            if constexpr (GRU) {
#pragma unroll
                for (int uw = 0; uw < UW; ++uw)
#pragma unroll
                    for (int c = 0; c < MT; ++c)
                        hv[uw][c] = *reinterpret_cast<const f32x4*>(hc + (4 * c + j) * HR + 64 * (wave * UW + uw) + 4 * b);
            }
            gemm(stream_of(l, I0(), NT0()), NT0(), RDT0(), xin, nx, hc + j * HR, acc);
            L2A_MTS(6 + 3 * l)

            if constexpr (GRU) {
                const Stream s1 = stream_of(l, I1(), NT1());
                prefetch(s1, NT1(), RDT1());
                f32x4 ug[UW][MT];
#pragma unroll
                for (int uw = 0; uw < UW; ++uw) {
                    const int tile = wave * UW + uw;
                    const f32x4 br = *reinterpret_cast<const f32x4*>(gb + 64 * tile + 4 * b);
                    const f32x4 bu = *reinterpret_cast<const f32x4*>(gb + U + 64 * tile + 4 * b);
#pragma unroll
                    for (int c = 0; c < MT; ++c) {
                        f32x4 rh;
#pragma unroll
                        for (int ii = 0; ii < 4; ++ii) {
                            rh[ii] = l2a_fast_sigmoid(acc[uw][c][ii] + br[ii]) * hv[uw][c][ii];
                            ug[uw][c][ii] = l2a_fast_sigmoid(acc[UW + uw][c][ii] + bu[ii]);
                        }
                        *reinterpret_cast<f32x4*>(aux + (4 * c + j) * HR + 64 * tile + 4 * b) = rh;
                        acc[uw][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    }
                }
                __syncthreads();                        // every unit's r * h before the candidate product
                L2A_MTS(15)
                gemm(s1, NT1(), RDT1(), xin, nx, aux + j * HR, acc);
                L2A_MTS(7 + 3 * l)
                with_act([&](auto act) {
                    l2a_static_for<0, UW>([&](auto uwv) {
                        constexpr int uw = decltype(uwv)::value;
                        const int tile = wave * UW + uw;
                        const f32x4 bc = *reinterpret_cast<const f32x4*>(gb + 2 * U + 64 * tile + 4 * b);
                        l2a_static_for<0, MT>([&](auto cv_) {
                            constexpr int c = decltype(cv_)::value;
                            next_operands(std::integral_constant<int, uw * MT + c>());
                            f32x4 hnew;
#pragma unroll
                            for (int ii = 0; ii < 4; ++ii) {
                                const float cnd = act(acc[uw][c][ii] + bc[ii]);
                                hnew[ii] = ug[uw][c][ii] * hv[uw][c][ii] + (1.0f - ug[uw][c][ii]) * cnd;
                            }
                            *reinterpret_cast<f32x4*>(hn + (4 * c + j) * HR + 64 * tile + 4 * b) = hnew;
                        });
                    });
                });
            } else {
                with_act([&](auto act) {
                    l2a_static_for<0, UW>([&](auto uwv) {
                        constexpr int uw = decltype(uwv)::value;
                        const int tile = wave * UW + uw;
                        f32x4 bias[G0];
#pragma unroll
                        for (int q = 0; q < G0; ++q) bias[q] = *reinterpret_cast<const f32x4*>(gb + q * U + 64 * tile + 4 * b);
                        l2a_static_for<0, MT>([&](auto cv_) {
                            constexpr int c = decltype(cv_)::value;
                            next_operands(std::integral_constant<int, uw * MT + c>());
                            float* at_h = hn + (4 * c + j) * HR + 64 * tile + 4 * b;
                            f32x4 hnew;
                            if constexpr (LSTM) {
                                float* at_c = aux + (4 * c + j) * HR + 64 * tile + 4 * b;
                                f32x4 cv = *reinterpret_cast<const f32x4*>(at_c);
#pragma unroll
                                for (int ii = 0; ii < 4; ++ii) {
                                    const float ig = l2a_fast_sigmoid(acc[uw][c][ii] + bias[0][ii]);
                                    const float jg = act(acc[UW + uw][c][ii] + bias[1][ii]);
                                    const float fg = l2a_fast_sigmoid(acc[2 * UW + uw][c][ii] + bias[2][ii] + 1.0f);
                                    const float og = l2a_fast_sigmoid(acc[3 * UW + uw][c][ii] + bias[3][ii]);
                                    const float cn = fg * cv[ii] + ig * jg;
                                    cv[ii] = cn;
                                    hnew[ii] = og * act(cn);
                                }
                                *reinterpret_cast<f32x4*>(at_c) = cv;
                            } else {
#pragma unroll
                                for (int ii = 0; ii < 4; ++ii) hnew[ii] = act(acc[uw][c][ii] + bias[0][ii]);
                            }
                            *reinterpret_cast<f32x4*>(at_h) = hnew;
                        });
                    });
                });
            }
            L2A_MTS(8 + 3 * l)
            if constexpr (!LASTL) {
                __syncthreads();                        // the layer above multiplies every unit's new h
                xin = hn + j * HR;
                nx = HIT;
            }
        };
#pragma unroll 1
        for (int l = 0; l + 1 < L; ++l) layer(l, std::false_type());
        layer(L - 1, std::true_type());
        L2A_MTS(1)

        // ---- output layer over this wave's units of the top layer's new h (its own writes: no barrier) -------------------
        f32x4* pb = pbuf;
        {
            const float* hb = lrows + (L - 1) * LROWS + (cur ^ 1) * NR * HR + j * HR + 64 * UW * wave;
            f32x4 oacc[2][MT];
#pragma unroll
            for (int ch = 0; ch < 2; ++ch)
#pragma unroll
                for (int c = 0; c < MT; ++c) oacc[ch][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ch = 0; ch < 2; ++ch)
#pragma unroll
                for (int g0 = 0; g0 < NGO / 2; g0 += 8) {
                    f32x4 hb4[8][MT];
#pragma unroll
                    for (int g = 0; g < 8; ++g)
#pragma unroll
                        for (int c = 0; c < MT; ++c)
                            hb4[g][c] = *reinterpret_cast<const f32x4*>(hb + 4 * c * HR + 4 * (ch * (NGO / 2) + g0 + g));
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int g = 0; g < 8; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
#pragma unroll
                            for (int c = 0; c < MT; ++c) oacc[ch][c] = L2A_MFMA4(pfo[ch][g0 + g][e], hb4[g][c][e], oacc[ch][c]);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
            for (int c = 0; c < MT; ++c) pb[(wave * MT + c) * 64 + lane] = oacc[0][c] + oacc[1][c];
        }
        // the next step's first k-groups (they have the reduce, the input rows and two barriers to arrive), then ...
        prefetch(stream_of(0, I0(), NT0()), NT0(), RDT0());
        // the coming steps' actions: requested here, behind the last wait of the output operands and two barriers ahead of the
        // next product's first wait - a read from HBM in front of the operand ring would stall the ring (returns are in order)
        av[0] = av_next[0]; av[1] = av_next[1];
        load_actions((t + 2 < p.h) ? t + 2 : p.h - 1, av_next);
        L2A_MTS(2)
        __syncthreads();
        L2A_MTS(3)

        // ---- reduce, output activation, denormalisation, reward, state update (this wave's micro tile): l2a_lstm_micro_body ---
        const float disc_t = (float)disc_pow;
        disc_pow *= p.discount;
        {
            const f32x4 bias = *reinterpret_cast<const f32x4*>(c_bo + 4 * b);
            const f32x4 omu = *reinterpret_cast<const f32x4*>(c_out_mu + 4 * b);
            const f32x4 osd = *reinterpret_cast<const f32x4*>(c_out_sd + 4 * b);
            f32x4 part[4];
#pragma unroll
            for (int w = 0; w < 4; ++w) part[w] = pb[(w * MT + ct) * 64 + lane];
            __builtin_amdgcn_sched_barrier(0);
            f32x4 s = (part[0] + part[1]) + (part[2] + part[3]);
            s = l2a_act4(s + bias, p.output_act);
            const f32x4 d = s * osd + omu;
            const f32x4 nx4 = st + d;
            float plin = ((qq == 0) ? p.rw.alive : 0.0f) - p.rw.ctrl_coef * asq;
            float psq = 0.0f;
            const int vi = p.rw.vel_index;
            const float dsel = (vi & 2) ? ((vi & 1) ? d[3] : d[2]) : ((vi & 1) ? d[1] : d[0]);
            const float dvel = l2a_from_row(dsel, vi >> 4);
            if (qq == ((vi >> 2) & 3)) plin += p.rw.w_vel * dvel * p.rw.inv_dt;
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int dim = 4 * b + ii;
                const bool in_dist = (p.rw.dist_coef != 0.0f) && (dim >= p.rw.dist_index) &&
                                     (dim < p.rw.dist_index + 3) && (dim < obs_dim);
                psq += in_dist ? nx4[ii] * nx4[ii] : 0.0f;
            }
            st = nx4;
            plin = l2a_row_quarter_sum(plin);
            float r = plin;
            if (p.rw.dist_coef != 0.0f) {
                psq = l2a_sum_xor32(l2a_sum_xor16(psq));
                psq = l2a_row_quarter_sum(psq);
                r -= p.rw.dist_coef * sqrtf(psq);
            }
            ret = fmaf(disc_t, r, ret);
        }
        L2A_MTS(4)
        write_x();
        __syncthreads();        // every micro tile's input rows are written (and the partials are through)
        L2A_MTS(5)
    }

    // ---- results: lanes of block 0 of wave c hold the returns of the candidates cand0 + 4 c + j; the keys meet in LDS ------
    {
        unsigned long long key = 0ull;
        if (valid_t && b == 0 && wave < MT) {
            if (p.returns_out) p.returns_out[(long long)env * p.n + cand_t] = ret;
            key = l2a_key_pack(ret, p.cand_offset + cand_t);
        }
        if (p.best_key) {
#pragma unroll
            for (int off = 2; off >= 1; off >>= 1) {
                const unsigned int hi = __shfl_xor((unsigned int)(key >> 32), off);
                const unsigned int lo = __shfl_xor((unsigned int)(key & 0xffffffffu), off);
                const unsigned long long other = ((unsigned long long)hi << 32) | lo;
                key = (other > key) ? other : key;
            }
            unsigned long long* kbuf = reinterpret_cast<unsigned long long*>(pbuf);
            if (lane == 0) kbuf[wave] = (wave < MT) ? key : 0ull;
            __syncthreads();
            if (tid == 0) {
#pragma unroll
                for (int w = 1; w < 4; ++w) key = (kbuf[w] > key) ? kbuf[w] : key;
                if (key != 0ull) atomicMax(p.best_key + env, key);
                l2a_publish_result(p, (int)gridDim.x);
            }
        }
    }
}

// Workgroup -> (env, first candidate, micro tiles): l2a_lstm_micro_k's dealing (host: l2a_lstm_api.hip).  MTM = 3: plans of at most
// three micro tiles per CU; MTM = 4: larger plans, ceil(quads / 4) workgroups of four (and three) micro tiles per env, as many
// rounds as that takes - the same loop at 16 candidates a workgroup, in place of l2a_rnn_mfma_k's product calls.
template <int UW, int CELL, int MTM>
__global__ void __launch_bounds__(256) l2a_rnn_micro_k(const L2ALstmParams p) {
    extern __shared__ __attribute__((aligned(16))) char l2a_smem[];
    const int env = (int)blockIdx.x / p.mc_w;
    const int idx = (int)blockIdx.x - env * p.mc_w;
    const int mt = idx < p.mc_r ? p.mc_hi : p.mc_hi - 1;
    const int q0 = idx < p.mc_r ? idx * p.mc_hi : p.mc_r * p.mc_hi + (idx - p.mc_r) * (p.mc_hi - 1);
    if constexpr (MTM == 4) {
        if (mt == 4) { l2a_rnn_micro_body<4, UW, CELL, MTM>(p, env, 4 * q0, l2a_smem); return; }
    }
    if (mt == 3) l2a_rnn_micro_body<3, UW, CELL, MTM>(p, env, 4 * q0, l2a_smem);
    else if (mt == 2) l2a_rnn_micro_body<2, UW, CELL, MTM>(p, env, 4 * q0, l2a_smem);
    else l2a_rnn_micro_body<1, UW, CELL, MTM>(p, env, 4 * q0, l2a_smem);
}
