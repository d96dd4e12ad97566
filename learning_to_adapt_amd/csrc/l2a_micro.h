// l2a_micro.h - MICRO TILES: candidate tiles of FOUR on v_mfma_f32_4x4x1_16b_f32 (gfx950 / CDNA4 only).
//
// Why.  v_mfma_f32_16x16x4_f32 fixes the candidate tile of l2a_mfma.h / l2a_lstm.h at N = 16, so a plan of T tiles keeps
// T of the 256 CUs busy and a plan between 128 and 256 tiles (the reference's own defaults: run_grbal.py:84-85 and
// run_rebal.py:77-78 give 5 x 500 candidates = 160 tiles) leaves a third of the chip idle - a tile cannot be cut without a
// per-step exchange between workgroups.  v_mfma_f32_4x4x1_16b_f32 multiplies sixteen independent 4 x 4 blocks, one k
// each, in two passes: with the sixteen blocks = 64 output units and the four columns = four candidates (replicated over
// the blocks) it does 256 MACs in 8 clocks - the 16x16x4 form's 32 MAC/clock - for FOUR candidates.  A workgroup can
// then own 4, 8 or 12 candidates (MT = 1 .. 3 micro tiles) at full matrix rate, no exchange: 2 500 candidates become 256
// workgroups of 12 and 8 instead of 160 of 16.  What it costs: the weight stream per candidate grows (every workgroup
// still streams every weight once per step: 64 B/clock of L1 bandwidth bound a 4-candidate workgroup at ~45 % of its
// matrix time, an 8-candidate one at ~80 %), and the MFMAs are four times shorter, so the loop has to be free of address
// arithmetic (tools/probes/microtile2.hip, profiles/r04_probe_microtile2_*.jsonl: 64 % of the matrix time with the
// compiler's schedule, 86 % with the loop below at 12 candidates).
//
// Operands.  Lane L = 4 b + j of a wave (block b, column j):
//   A  lane L supplies the weight of output SLOT L of a 64-unit tile for one k: a 16-byte load per lane = four consecutive
//      k of the chain (below); a wave-level load = 1 KiB contiguous (packed at set_weights, l2a_*_micro_pack_k);
//   B  lane L supplies the activation of candidate j for the same k - the same value in all sixteen blocks: one
//      ds_read_b128 of the candidate's LDS row (four consecutive chain positions, broadcast);
//   D  lane L, register i = slot 4 b + i, candidate j.
// Same bits as the 16-candidate kernels.  v_mfma_f32_16x16x4_f32 is a k-ordered fma chain (guide section 3), and with
// the packed layout of l2a_kernels.h the chain of a 16-feature k-group runs k = 16 g + 4 kk + ii for ii = 0..3 (MFMA),
// kk = 0..3 (inside one).  A 4x4x1 MFMA is ONE fma per output, so issuing them in that order reproduces the chain bit
// for bit: "chain position" p = 16 g + 4 ii + kk <-> feature k = 16 g + 4 kk + ii (l2a_chain_k, an involution).  LDS rows
// hold activations in chain order, the packed A arrays hold k in chain order, and output slot s of a 64-unit tile is
// unit l2a_chain_k(s) - so that a lane's D registers (slots 4 b .. 4 b + 3) are four consecutive chain positions of the
// next layer's row: one ds_write_b128.  Reduction trees, gate arithmetic, reward and key are the 16-candidate kernels'.
#pragma once

#include <type_traits>

#include "l2a_lstm.h"
#include "l2a_micro_pack.h"
#include "l2a_micro_launch.h"

template <int I, int N, class F>
__device__ __forceinline__ void l2a_static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>()); l2a_static_for<I + 1, N>(f); }
}

// issue-order hint of one k-group step (NM MFMAs): its NB LDS reads first, one per two MFMAs - they are consumed at the top of the
// NEXT step and need their ~100 clocks - then its NA weight loads (consumed three steps later) spread over the rest
template <int NA, int NB, int NM>
__device__ __forceinline__ void l2a_micro_hint() {
    constexpr int LEAD = (2 * NB + 2 <= NM) ? 2 : 1;
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    l2a_static_for<0, NB>([&](auto) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, LEAD, 0);
    });
    constexpr int REST = NM - 1 - LEAD * NB;
    constexpr int PER = REST / (NA + 1) > 0 ? REST / (NA + 1) : 1;
    l2a_static_for<0, NA>([&](auto) {
        __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    });
    __builtin_amdgcn_sched_group_barrier(0x008, REST - PER * NA > 0 ? REST - PER * NA : 0, 0);
}

// Lane moves at VALU rate (no LDS crossbar round trip like ds_bpermute: the reduce tail of a step was 1.8k clocks of
// dependent __shfl's, timeline r04).  DPP row_shl:n - lane l of a 16-lane row reads lane l + n of its row; only lanes whose
// source exists are used (block 0 of a row for n = 4, 8).
template <int N>
__device__ __forceinline__ float l2a_row_shl(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x100 + N, 0xf, 0xf, true));
}
// (x0 + x1) + (x2 + x3) over the four 4-lane blocks of a 16-lane row, valid in the row's block 0
__device__ __forceinline__ float l2a_row_quarter_sum(float x) {
    const float t = x + l2a_row_shl<4>(x);
    return t + l2a_row_shl<8>(t);
}
// the value the same lane position holds in 16-lane row `r` (wave uniform), valid in row 0 (l2a_sum_xor16 / 32: the swap
// instructions return {own, other})
__device__ __forceinline__ float l2a_from_row(float x, int r) {
    if (r & 1) {
        const unsigned int u = __float_as_uint(x);
        x = __uint_as_float(__builtin_amdgcn_permlane16_swap(u, u, false, false)[1]);
    }
    if (r & 2) {
        const unsigned int u = __float_as_uint(x);
        x = __uint_as_float(__builtin_amdgcn_permlane32_swap(u, u, false, false)[1]);
    }
    return x;
}

// Phase timeline for tools/timeline_micro.py (builds with -DL2A_TIMELINE only): the waves of workgroup 0 stamp the shader
// clock, dbg[(t * 4 + wave) * 16 + slot].
#ifndef L2A_TIMELINE
#define L2A_MTS(slot)
#define L2A_MTS_AT(step, slot)
#else
#define L2A_MTS_AT(step, slot) { const int t = (step); L2A_MTS(slot) }
#define L2A_MTS(slot)                                                                       \
    if (p.dbg && blockIdx.x == 0) {                                                         \
        unsigned long long ts_;                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_) : : "memory");       \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        if (lane == 0) p.dbg[((long long)t * 4 + wave) * 16 + (slot)] = ts_;                \
    }
#endif

// ------------------------------------------------------------------------------------------------------------------
// Recurrent planner (one LSTM layer of 256 / 512 units: l2a_lstm.h is the 16-candidate kernel, same arithmetic)
// ------------------------------------------------------------------------------------------------------------------

template <int MT, int UW>
__device__ __forceinline__ void l2a_lstm_micro_body(const L2ALstmParams& p, const int env, const int cand0, char* smem) {
    constexpr int U = 256 * UW;
    constexpr int ROWF = l2a_micro_row(U);
    constexpr int NTL = 4 * UW;             // accumulator tiles of a wave: gate q of its 64-unit tile uw -> q * UW + uw
    constexpr int HG = U / 8;               // k-groups per half of h
    constexpr int HI = HG / 4;              // loop iterations (four k-groups each) per half
    constexpr int NGO = 16 * UW;            // output-layer k-groups of this wave's units
    const int KG0 = p.KG0;
    const int NIT = 2 * HI + KG0;
    const int NG = 4 * NIT;
    float* rows = reinterpret_cast<float*>(smem);                   // [2][4 MT][ROWF]
    f32x4* pbuf = reinterpret_cast<f32x4*>(rows + 2 * 12 * ROWF);   // [2][4 waves][MT][64]
    float* c_in_mu = reinterpret_cast<float*>(pbuf + 2 * 4 * 3 * 64);
    float* c_in_iv = c_in_mu + 16 * KG0;
    float* c_out_mu = c_in_iv + 16 * KG0;   // [64] each, zero past the observation
    float* c_out_sd = c_out_mu + 64;
    float* c_bo = c_out_sd + 64;
    float* c_gb = c_bo + 64;                // gate bias in slot order [q][U]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = lane >> 2, j = lane & 3;
    const int qq = b & 3;                   // the lane's role in the 16-candidate kernels' quarter sums (dims 16 c + 4 qq + ii)
    const int obs_dim = p.obs_dim, act_dim = p.act_dim;
    const int R = p.m * p.n;
    L2A_MTS_AT(0, 8)

    for (int i = tid; i < 32 * KG0 + 192 + 4 * U; i += 256) {
        float v;
        if (i < 32 * KG0) v = p.wblk[p.nm_off + i];
        else if (i < 32 * KG0 + 192) {
            const int a = (i - 32 * KG0) >> 6, d = (i - 32 * KG0) & 63;
            v = (d >= obs_dim) ? 0.0f : (a == 0 ? p.wblk[p.nm_off + 32 * KG0 + d]
                                                : (a == 1 ? p.wblk[p.nm_off + 32 * KG0 + 16 * p.OT + d] : p.wblk[p.raw_bo + d]));
        } else {
            const int o = i - (32 * KG0 + 192), q = o / U, s = o - q * U;
            v = p.wblk[p.raw_bk + q * U + l2a_chain_k(s)];         // slot s of tile s / 64 is unit chain_k(s)
        }
        c_in_mu[i] = v;
    }
    for (int i = tid; i < 2 * 12 * ROWF; i += 256) rows[i] = 0.0f;  // padding (and rows of absent micro tiles) stay zero
    __syncthreads();
    L2A_MTS_AT(0, 9)

    int cand[MT], row[MT];
    bool valid[MT];
#pragma unroll
    for (int c = 0; c < MT; ++c) {
        cand[c] = cand0 + 4 * c + j;
        valid[c] = cand[c] < p.n;
        row[c] = env * p.n + (valid[c] ? cand[c] : p.n - 1);
    }
    // Everything per candidate outside the gate GEMM - the output-layer reduce, reward, state, return, the next step's actions
    // and input rows - is done for micro tile c by wave c alone (a wave beyond the micro tiles repeats the last one's work, same
    // values to the same places: no divergence); one barrier per step hands the input rows to the other waves.
    const int ct = wave < MT ? wave : MT - 1;
    const int cand_t = cand0 + 4 * ct + j;
    const bool valid_t = cand_t < p.n;
    const int row_t = env * p.n + (valid_t ? cand_t : p.n - 1);
    // state: dims 4 b .. 4 b + 3 of candidate j of this wave's micro tile
    f32x4 st, creg[UW][MT];
    {
        const float* orow = p.obs0 + (long long)env * obs_dim;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int dim = 4 * b + i;
            const float v = orow[dim < obs_dim ? dim : obs_dim - 1];
            st[i] = (dim < obs_dim) ? v : 0.0f;
        }
    }
#pragma unroll
    for (int c = 0; c < MT; ++c) {
        const long long hrow = (p.hid_per_row ? (long long)row[c] : (long long)env) * U;
#pragma unroll
        for (int uw = 0; uw < UW; ++uw) {
            const int tile = wave * UW + uw;
            f32x4 hv;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int unit = 64 * tile + l2a_chain_k(4 * b + i);
                creg[uw][c][i] = p.c0[hrow + unit];
                hv[i] = p.h0[hrow + unit];
            }
            *reinterpret_cast<f32x4*>(rows + (4 * c + j) * ROWF + 64 * tile + 4 * b) = hv;
        }
    }

    // raw actions of the lanes that play the 16-candidate kernel's quarter role (b < 4: qq = b): bounds-checked buffer
    // loads from a per-step descriptor, slots without an action read 0.0 (l2a_mfma.h)
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
    // normalised inputs of the coming step -> the x section of `dst` rows (chain order), the rows of this wave's micro tile.
    // Branch-free: a lane's normalisation constants and the row offsets of its values are loop invariants (registers); a slot
    // that holds no feature of this lane goes to the row's padding (floats U + 80 .. U + 87, never read).  (With a guard per
    // element the compiler serialised read - wait - compute - write behind a branch each: 5.4k clocks per step, timeline r04.)
    f32x4 av[2];
    float asq;
    int xo_s[4], xo_a[2][4];
    f32x4 mu_s, iv_s, mu_a[2], iv_a[2];
    {
        mu_s = *reinterpret_cast<const f32x4*>(c_in_mu + 4 * b);        // (past the inputs: other constants, unused)
        iv_s = *reinterpret_cast<const f32x4*>(c_in_iv + 4 * b);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = 4 * b + i;
            xo_s[i] = U + ((k < obs_dim) ? l2a_chain_k(k) : 80 + (lane & 7));
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            mu_a[s2] = *reinterpret_cast<const f32x4*>(c_in_mu + 16 * (ga0 + s2) + 4 * qq);
            iv_a[s2] = *reinterpret_cast<const f32x4*>(c_in_iv + 16 * (ga0 + s2) + 4 * qq);
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int k = 16 * (ga0 + s2) + 4 * qq + ii;
                xo_a[s2][ii] = U + ((b < 4 && k >= obs_dim && k < obs_dim + act_dim) ? l2a_chain_k(k) : 80 + (lane & 7));
            }
        }
    }
    auto write_x = [&](float* dst) {
        float* xr = dst + (4 * ct + j) * ROWF;
#pragma unroll
        for (int i = 0; i < 4; ++i) xr[xo_s[i]] = ((st[i] + 0.0f) - mu_s[i]) * iv_s[i];
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
            for (int ii = 0; ii < 4; ++ii) xr[xo_a[s2][ii]] = ((0.0f + av[s2][ii]) - mu_a[s2][ii]) * iv_a[s2][ii];
    };
    load_actions(0, av_next);
    av[0] = av_next[0]; av[1] = av_next[1];
    load_actions(p.h > 1 ? 1 : 0, av_next);
    write_x(rows);

    float ret = p.ret_in ? p.ret_in[(long long)env * p.n + (valid_t ? cand_t : p.n - 1)] : 0.0f;
    double disc_pow = p.disc0;

    // ---- operand streams ---------------------------------------------------------------------------------------------
    const __amdgpu_buffer_rsrc_t rsA = l2a_rsrc(p.wblk + p.pk_mg, l2a_lstm_micro_gate_floats(U, KG0) * 4);
    const __amdgpu_buffer_rsrc_t rsO = l2a_rsrc(p.wblk + p.pk_mo, (long long)(U / 4) * 1024);
    int voffA[NTL];
#pragma unroll
    for (int tl = 0; tl < NTL; ++tl) {
        const int q = tl / UW, uw = tl - q * UW;
        voffA[tl] = lane * 16 + (((wave * UW + uw) * 4 + q) * NG) * 1024;
    }
    const int voffO = lane * 16 + wave * NGO * 1024;
    const int own = (wave * UW * 64) / (U / 2);         // which half of h holds this wave's units
    f32x4 ra[4][NTL];                                   // weight ring: four k-groups, three requested ahead; lives across steps
    f32x4 rb[2][MT];                                    // activation ring: two k-groups
    auto issue_a = [&](int soff, auto imm_tag, auto slot_tag) {
        constexpr int s = decltype(slot_tag)::value, IMM = decltype(imm_tag)::value;
#pragma unroll
        for (int tl = 0; tl < NTL; ++tl) ra[s][tl] = l2a_ldw(rsA, voffA[tl] + IMM, soff);
    };
    auto issue_b = [&](const float* bp, auto off_tag, auto slot_tag) {
        constexpr int s = decltype(slot_tag)::value, OFF = decltype(off_tag)::value;
#pragma unroll
        for (int c = 0; c < MT; ++c) rb[s][c] = *reinterpret_cast<const f32x4*>(bp + 4 * c * ROWF + OFF);
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    // (in this order, pinned - see l2a_mlp_micro_body: the loops' s_waitcnt counts have to hold on the path from here too)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tl = 0; tl < NTL; ++tl) { ra[0][tl] = l2a_ldw(rsA, voffA[tl], 0); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
    for (int tl = 0; tl < NTL; ++tl) { ra[1][tl] = l2a_ldw(rsA, voffA[tl] + 1024, 0); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
    for (int tl = 0; tl < NTL; ++tl) { ra[2][tl] = l2a_ldw(rsA, voffA[tl] + 2048, 0); __builtin_amdgcn_sched_barrier(0); }
    __syncthreads();        // every wave's share of h(0) and every micro tile's input rows are in the rows
    L2A_MTS_AT(0, 10)

    for (int t = 0; t < p.h; ++t) {
        float* rows_c = rows + (t & 1) * 12 * ROWF;
        float* rows_n = rows + ((t + 1) & 1) * 12 * ROWF;
        f32x4* pb = pbuf + (t & 1) * (4 * 3 * 64);
        const float* hb_own = rows_c + j * ROWF + own * (U / 2);
        const float* hb_oth = rows_c + j * ROWF + (1 - own) * (U / 2);
        const float* xb = rows_c + j * ROWF + U;

        L2A_MTS(0)
        f32x4 acc[NTL][MT];
#pragma unroll
        for (int tl = 0; tl < NTL; ++tl)
#pragma unroll
            for (int c = 0; c < MT; ++c) acc[tl][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // ---- gate GEMM: K order own half of h, other half, x (l2a_lstm.h) ------------------------------------------
        issue_b(hb_own, I0(), I0());
        int itg = 0;
        auto segment = [&](const float* bbase, int nit, const float* bnext) {
#pragma unroll 1
            for (int it = 0; it < nit; ++it, ++itg) {
                const int s0 = itg * 4096;
                const int s1 = (itg + 1 == NIT) ? 0 : s0 + 4096;       // past the last k-group: the NEXT step's first ones
                const float* bp = bbase + 16 * it;
                const float* bn = (it + 1 < nit) ? bp + 16 : bnext;
                l2a_static_for<0, 4>([&](auto iv) {
                    constexpr int I = decltype(iv)::value;
                    if constexpr (I == 0) issue_a(s0, std::integral_constant<int, 3072>(), I3());
                    else issue_a(s1, std::integral_constant<int, (I - 1) * 1024>(), std::integral_constant<int, I - 1>());
                    if constexpr (I < 3) issue_b(bp, std::integral_constant<int, 4 * (I + 1)>(), std::integral_constant<int, (I + 1) & 1>());
                    else issue_b(bn, I0(), I0());
                    // (the LAST-requested operands first: one s_waitcnt per kind and step instead of one per operand - with MFMAs of
                    // 8 clocks every instruction between them is paid in matrix time; each accumulator's own chain order is e = 0 .. 3)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int tl = NTL - 1; tl >= 0; --tl)
#pragma unroll
                            for (int c = MT - 1; c >= 0; --c) acc[tl][c] = L2A_MFMA4(ra[I][tl][e], rb[I & 1][c][e], acc[tl][c]);
                    l2a_micro_hint<NTL, MT, 4 * NTL * MT>();
                });
            }
        };
        segment(hb_own, HI, hb_oth);
        segment(hb_oth, HI, xb);
        segment(xb, KG0, xb);
        __builtin_amdgcn_sched_barrier(0);
        L2A_MTS(1)

        // output-layer operands of this wave's first chunk: in flight under the gate arithmetic
        f32x4 pfo[2][NGO / 2];
#pragma unroll
        for (int g = 0; g < NGO / 2; ++g) pfo[0][g] = l2a_ldw(rsO, voffO + g * 1024, 0);

        // ---- gate arithmetic (register local) -> c, h (LDS rows of the next step, this wave's units) ------------------
#pragma unroll
        for (int uw = 0; uw < UW; ++uw) {
            const int tile = wave * UW + uw;
            f32x4 bias[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bias[q] = *reinterpret_cast<const f32x4*>(c_gb + q * U + 64 * tile + 4 * b);
#pragma unroll
            for (int c = 0; c < MT; ++c) {
                f32x4 z[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) z[q] = acc[q * UW + uw][c] + bias[q];
                f32x4 cn, hn;
                if (p.cell_act == L2A_ACT_TANH) {
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) {
                        const float ig = l2a_fast_sigmoid(z[0][ii]);
                        const float jg = l2a_fast_tanh(z[1][ii]);
                        const float fg = l2a_fast_sigmoid(z[2][ii] + 1.0f);
                        const float og = l2a_fast_sigmoid(z[3][ii]);
                        cn[ii] = fmaf(fg, creg[uw][c][ii], ig * jg);
                        hn[ii] = og * l2a_fast_tanh(cn[ii]);
                    }
                } else {
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) {
                        const float ig = l2a_fast_sigmoid(z[0][ii]);
                        const float jg = l2a_act1(z[1][ii], p.cell_act);
                        const float fg = l2a_fast_sigmoid(z[2][ii] + 1.0f);
                        const float og = l2a_fast_sigmoid(z[3][ii]);
                        cn[ii] = fmaf(fg, creg[uw][c][ii], ig * jg);
                        hn[ii] = og * l2a_act1(cn[ii], p.cell_act);
                    }
                }
                creg[uw][c] = cn;
                *reinterpret_cast<f32x4*>(rows_n + (4 * c + j) * ROWF + 64 * tile + 4 * b) = hn;
            }
        }

        L2A_MTS(2)
        // ---- output layer over this wave's units (its two chunks of the canonical tree), B = its own new h -----------
        {
#pragma unroll
            for (int g = 0; g < NGO / 2; ++g) pfo[1][g] = l2a_ldw(rsO, voffO + (NGO / 2 + g) * 1024, 0);
            const float* hb = rows_n + j * ROWF + 64 * UW * wave;
            f32x4 oacc[2][MT];
#pragma unroll
            for (int ch = 0; ch < 2; ++ch)
#pragma unroll
                for (int c = 0; c < MT; ++c) oacc[ch][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
            // (the activations of eight k-groups at a time, all reads issued before the first MFMA: read where they are used,
            // every k-group paid an exposed LDS round trip - 4.1k clocks for 1.5k of MFMA issue, timeline r04)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch)
#pragma unroll
                for (int g0 = 0; g0 < NGO / 2; g0 += 8) {
                    f32x4 hb4[8][MT];
#pragma unroll
                    for (int g = 0; g < 8; ++g)
#pragma unroll
                        for (int c = 0; c < MT; ++c)
                            hb4[g][c] = *reinterpret_cast<const f32x4*>(hb + 4 * c * ROWF + 4 * (ch * (NGO / 2) + g0 + g));
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
        // the coming steps' actions: requested HERE, behind the last wait of the output operands and two barriers ahead of the next
        // step's second k-group iteration: memory returns in order, so a read that comes from HBM (the actions of a fresh plan)
        // requested after the reduce would sit in the queue in front of that iteration's weight requests (with the actions warm
        // in the L2, as in the launch-loop benchmarks, the two placements measure the same: profiles/r04_ab_action_loads.txt)
        av[0] = av_next[0]; av[1] = av_next[1];
        load_actions((t + 2 < p.h) ? t + 2 : p.h - 1, av_next);
        L2A_MTS(3)
        __syncthreads();
        L2A_MTS(4)

        // ---- canonical reduce, output activation, denormalisation, reward, state update (this wave's micro tile) ---------
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
            {
                f32x4 s = (part[0] + part[1]) + (part[2] + part[3]);
                s = l2a_act4(s + bias, p.output_act);
                const f32x4 d = s * osd + omu;
                const f32x4 nx = st + d;
                // reward in the 16-candidate kernel's order: quarter partials r_qq (lanes b = qq < 4), (r0 + r1) + (r2 + r3)
                float plin = ((qq == 0) ? p.rw.alive : 0.0f) - p.rw.ctrl_coef * asq;
                float psq = 0.0f;
                const int vi = p.rw.vel_index;
                const float dsel = (vi & 2) ? ((vi & 1) ? d[3] : d[2]) : ((vi & 1) ? d[1] : d[0]);
                const float dvel = l2a_from_row(dsel, vi >> 4);         // block (vi >> 2) & 3 of row 0 <- the lane that holds dim vel_index
                if (qq == ((vi >> 2) & 3)) plin += p.rw.w_vel * dvel * p.rw.inv_dt;
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    const int dim = 4 * b + ii;
                    const bool in_dist = (p.rw.dist_coef != 0.0f) && (dim >= p.rw.dist_index) &&
                                         (dim < p.rw.dist_index + 3) && (dim < obs_dim);
                    psq += in_dist ? nx[ii] * nx[ii] : 0.0f;
                }
                st = nx;
                plin = l2a_row_quarter_sum(plin);
                float r = plin;
                if (p.rw.dist_coef != 0.0f) {
                    psq = l2a_sum_xor32(l2a_sum_xor16(psq));                    // over the obs tiles of a quarter (one is non-zero)
                    psq = l2a_row_quarter_sum(psq);
                    r -= p.rw.dist_coef * sqrtf(psq);
                }
                ret = fmaf(disc_t, r, ret);
            }
        }
        // the next step's inputs
        L2A_MTS(5)
        write_x(rows_n);
        __syncthreads();        // every micro tile's input rows are written
        L2A_MTS(6)
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
            unsigned long long* kbuf = reinterpret_cast<unsigned long long*>(pbuf);     // (the partials are through)
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

// Workgroup -> (env, first candidate, micro tiles): every env's ceil(n / 4) micro tiles are dealt to p.mc_w workgroups,
// the first p.mc_r of them take p.mc_hi micro tiles, the others one fewer (host: l2a_lstm_api.hip).
template <int UW>
__global__ void __launch_bounds__(256) l2a_lstm_micro_k(const L2ALstmParams p) {
    extern __shared__ __attribute__((aligned(16))) char l2a_smem[];
    const int env = (int)blockIdx.x / p.mc_w;
    const int idx = (int)blockIdx.x - env * p.mc_w;
    const int mt = idx < p.mc_r ? p.mc_hi : p.mc_hi - 1;
    const int q0 = idx < p.mc_r ? idx * p.mc_hi : p.mc_r * p.mc_hi + (idx - p.mc_r) * (p.mc_hi - 1);
    if (mt == 3) l2a_lstm_micro_body<3, UW>(p, env, 4 * q0, l2a_smem);
    else if (mt == 2) l2a_lstm_micro_body<2, UW>(p, env, 4 * q0, l2a_smem);
    else l2a_lstm_micro_body<1, UW>(p, env, 4 * q0, l2a_smem);
}


// ------------------------------------------------------------------------------------------------------------------
// MLP planner (l2a_mfma.h is the 16-candidate kernel: same arithmetic in the same order, so the same bits)
//
// Workgroup = 4 waves, MT micro tiles (4 MT candidates) of one env for the whole horizon; wave w owns the 64-unit tiles
// [w UW, (w + 1) UW) of every hidden layer (UW = H / 256) - i.e. the hidden units of chunks 2 w and 2 w + 1 of the
// canonical output-layer reduce - and, in the output layer, those two chunks.  Per (step, set): layer 0 (B = the set's
// normalised inputs, written to LDS once per step for all sets) -> rows A | barrier | hidden layers (rows A -> rows B ->
// rows A ...) with a barrier between two of them | output layer over the wave's OWN columns of the last rows (its own
// writes: no barrier) -> one partial per wave | barrier | the four partials are reduced (p0 + p1) + (p2 + p3), the set joins
// its ensemble group.  Everything per candidate - this reduce, the ensemble mean, reward, state, return, the next step's
// actions and input rows - is done for micro tile c by wave c alone (a wave beyond the micro tiles repeats the last one's
// work, same values to the same places: no divergence); one barrier per step hands the input rows to the other waves.  The weights of a (step, set) are ONE linear stream per 64-unit tile (l2a_micro_pack.h),
// so the operand ring (three records requested ahead) runs through every phase boundary and into the next set with a
// record counter and nothing else.
// ------------------------------------------------------------------------------------------------------------------
template <int MT, int UW, bool GACT>
__device__ __forceinline__ void l2a_mlp_micro_body(const L2AKParams& p, const int env, const int cand0, char* smem) {
    constexpr int H = 256 * UW;
    constexpr int ROWF = l2a_micro_row(H);
    constexpr int XROWF = 104;              // input rows: 96 floats (six k-groups) + 8 (rows 8 banks apart: conflict-free 16-byte accesses)
    constexpr int RD = 8;                   // operand ring: eight records, seven requested ahead (~1.3k clocks of matrix work at 12
                                            // candidates: a workgroup that shares its XCD's L2 with another env's weights misses it)
    constexpr int HI = H / 32;              // loop iterations (eight records each) of a hidden layer
    constexpr int CIT = H / 256;            // ... of one output-layer chunk (H / 32 records)
    constexpr int NSEQ = 2 / UW;            // the wave's two chunks: side by side as two streams (UW == 2) or one after the other
    const int KG0 = p.KG0, n_hidden = p.n_hidden;
    const int KG0E = l2a_mlp_micro_kg0e(KG0);
    const int NREC = p.m_nrec;
    const bool per_block = (p.mode == L2A_MODE_PER_BLOCK);
    const int e_loop = (p.mode == L2A_MODE_MEAN) ? p.n_sets : 1;
    const int e_half = (e_loop + 1) >> 1;               // group A = [0, e_half), B = [e_half, e_loop)
    const int CST = l2a_mlp_micro_cst(H, KG0, n_hidden);
    const int CST_B = 32 * KG0 + 192;                   // [in_mu 16 KG0][in_iv 16 KG0][out_mu 64][out_sd 64][b_out 64][hidden biases, slot order]

    float* rows = reinterpret_cast<float*>(smem);                   // [2][12][ROWF]
    f32x4* pbuf = reinterpret_cast<f32x4*>(rows + 2 * 12 * ROWF);   // [2][4 waves][3][64]
    float* xs = reinterpret_cast<float*>(pbuf + 2 * 4 * 3 * 64);    // [e_loop][12][XROWF]
    float* cst = xs + e_loop * 12 * XROWF;                          // [e_loop][CST]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = lane >> 2, j = lane & 3;
    const int qq = b & 3;
    const int obs_dim = p.obs_dim, act_dim = p.act_dim;
    const int R = p.m * p.n;
    auto set_of = [&](int i) { return per_block ? env : i; };       // weight set of the i-th member this workgroup runs
    L2A_MTS_AT(0, 8)

    for (int i = tid; i < e_loop * CST; i += 256) {
        const int sl = i / CST, o = i - sl * CST;
        const float* src = p.wblk + (long long)set_of(sl) * p.set_stride;
        float v;
        if (o < 32 * KG0) v = src[p.nm_off + o];
        else if (o < CST_B) {
            const int a = (o - 32 * KG0) >> 6, d = (o - 32 * KG0) & 63;
            v = (d >= obs_dim) ? 0.0f : (a == 0 ? src[p.nm_off + 32 * KG0 + d]
                                                : (a == 1 ? src[p.nm_off + 32 * KG0 + 16 * p.OT + d] : src[p.raw_b[n_hidden] + d]));
        } else {
            const int l = (o - CST_B) / H, u = (o - CST_B) - l * H;
            v = src[p.raw_b[l] + l2a_chain_k(u)];                   // slot u of its 64-unit tile is unit chain_k(u)
        }
        cst[i] = v;
    }
    for (int i = tid; i < e_loop * 12 * XROWF; i += 256) xs[i] = 0.0f;     // input padding stays zero
    __syncthreads();
    L2A_MTS_AT(0, 9)

    // this wave's micro tile
    const int ct = wave < MT ? wave : MT - 1;
    const int cand = cand0 + 4 * ct + j;
    const bool valid = cand < p.c_hi;          // (a launch covers candidates [c_lo, c_hi) of every env)
    const int row = env * p.n + (valid ? cand : p.c_hi - 1);
    // state: dims 4 b .. 4 b + 3 of candidate j
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

    // raw actions of the lanes that play the 16-candidate kernel's quarter role (b < 4: qq = b), as in l2a_mfma.h
    const int ga0 = obs_dim >> 4;
    f32x4 av_next[2];
    int aoff[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const int ka = 16 * (ga0 + s) + 4 * qq + ii - obs_dim;
            const bool in = (b < 4) && (ka >= 0) && (ka < act_dim);
            aoff[s][ii] = in ? (row * act_dim + ka) * 4 : 0x7ffffff0;
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
    // normalised inputs of the coming step, every set's -> xs (chain order), the rows of this wave's micro tile.
    // Branch-free: the row offsets of a lane's values are loop invariants, a slot that holds no feature of this lane goes to
    // the row's padding (floats 96 .. 103, never read); a set's constants come as 16-byte reads, all issued before the arithmetic.
    f32x4 av[2];
    float asq;
    int xo_s[4], xo_a[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = 4 * b + i;
        xo_s[i] = (k < obs_dim) ? l2a_chain_k(k) : 96 + (lane & 7);
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const int k = 16 * (ga0 + s2) + 4 * qq + ii;
            xo_a[s2][ii] = (b < 4 && k >= obs_dim && k < obs_dim + act_dim) ? l2a_chain_k(k) : 96 + (lane & 7);
        }
    auto write_x = [&]() {
        {
            float s = 0.0f;
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                s = fmaf(av[0][ii], av[0][ii], s);
                s = fmaf(av[1][ii], av[1][ii], s);
            }
            asq = s;
        }
        for (int sl = 0; sl < e_loop; ++sl) {
            const float* mu = cst + sl * CST;
            const float* iv = mu + 16 * KG0;
            const f32x4 mu_s = *reinterpret_cast<const f32x4*>(mu + 4 * b);     // (past the inputs: other constants, unused)
            const f32x4 iv_s = *reinterpret_cast<const f32x4*>(iv + 4 * b);
            f32x4 mu_a[2], iv_a[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                mu_a[s2] = *reinterpret_cast<const f32x4*>(mu + 16 * (ga0 + s2) + 4 * qq);
                iv_a[s2] = *reinterpret_cast<const f32x4*>(iv + 16 * (ga0 + s2) + 4 * qq);
            }
            __builtin_amdgcn_sched_barrier(0);
            float* xr = xs + (sl * 12 + 4 * ct + j) * XROWF;
#pragma unroll
            for (int i = 0; i < 4; ++i) xr[xo_s[i]] = ((st[i] + 0.0f) - mu_s[i]) * iv_s[i];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) xr[xo_a[s2][ii]] = ((0.0f + av[s2][ii]) - mu_a[s2][ii]) * iv_a[s2][ii];
        }
    };
    load_actions(0, av_next);
    av[0] = av_next[0]; av[1] = av_next[1];
    load_actions(p.h > 1 ? 1 : 0, av_next);
    write_x();

    float ret = p.ret_in ? p.ret_in[(long long)env * p.n + (valid ? cand : p.c_hi - 1)] : 0.0f;
    double disc_pow = p.disc0;
    const float e_count = (float)e_loop;
    const float e_inv = 1.0f / e_count;

    // ---- operand stream ----------------------------------------------------------------------------------------------
    const __amdgpu_buffer_rsrc_t rsA = l2a_rsrc(p.wblk + p.pk_m, p.m_bytes);
    int voffA[UW];
#pragma unroll
    for (int tl = 0; tl < UW; ++tl) voffA[tl] = lane * 16 + (wave * UW + tl) * NREC * 1024;
    const int set_bytes = (int)(p.set_stride * 4);
    int sbase = set_of(0) * set_bytes;                  // byte offset of the running set's streams
    int snext = set_of(e_loop > 1 ? 1 : 0) * set_bytes; // ... of the set after it
    int rec = 0;                                        // records of the running set consumed so far (multiple of 4)
    f32x4 ra[RD][UW];                                   // weight ring; lives across phases, sets, steps
    auto issue_a = [&](int soff, auto imm_tag, auto slot_tag) {
        constexpr int s = decltype(slot_tag)::value, IMM = decltype(imm_tag)::value;
#pragma unroll
        for (int tl = 0; tl < UW; ++tl) ra[s][tl] = l2a_ldw(rsA, voffA[tl] + IMM, soff);
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    // (in THIS order, pinned: the loops wait for a ring slot by counting the loads issued after it - s_waitcnt vmcnt - and the
    // compiler inserts ONE count per wait that has to hold on every path into the loop; left to itself the scheduler issued
    // the first loads backwards, record 0 last, and every loop iteration then drained the whole ring: 1.7x the time, r04)
    __builtin_amdgcn_sched_barrier(0);
    l2a_static_for<0, RD - 1>([&](auto rv) {
        constexpr int r = decltype(rv)::value;
#pragma unroll
        for (int tl = 0; tl < UW; ++tl) {
            ra[r][tl] = l2a_ldw(rsA, voffA[tl] + (r & 3) * 1024, sbase + (r >> 2) * 4096);
            __builtin_amdgcn_sched_barrier(0);
        }
    });

    // One phase = `nit` iterations of eight records for each of the wave's UW streams: acc[tl][c] += A(stream tl) x B.
    // B = four chain positions of a candidate's LDS row per record: NB == 1: the same row section for every stream (a layer:
    // the streams are output tiles), NB == UW: stream tl reads 64 floats further (output layer: the streams are chunks of K).
    // CS = floats between the rows of two micro tiles.  Activation ring: four records, two requested ahead (a step of 8
    // candidates is 128 clocks of matrix work - less than an LDS round trip).
    auto run_phase = [&](const float* b0, const int nit, auto cs_tag, auto nb_tag, f32x4 (&acc)[UW][MT]) {
        constexpr int CS = decltype(cs_tag)::value, NB = decltype(nb_tag)::value;
        f32x4 rb[4][NB][MT];
        auto issue_b = [&](const float* bp, auto off_tag, auto slot_tag) {
            constexpr int s = decltype(slot_tag)::value, OFF = decltype(off_tag)::value;
#pragma unroll
            for (int tb = 0; tb < NB; ++tb)
#pragma unroll
                for (int c = 0; c < MT; ++c) rb[s][tb][c] = *reinterpret_cast<const f32x4*>(bp + 64 * tb + CS * c + OFF);
        };
#pragma unroll
        for (int tl = 0; tl < UW; ++tl)
#pragma unroll
            for (int c = 0; c < MT; ++c) acc[tl][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        issue_b(b0, I0(), I0());
        issue_b(b0, std::integral_constant<int, 4>(), I1());
#pragma unroll 1
        for (int it = 0; it < nit; ++it) {
            const int s0b = sbase + rec * 1024 + 4096;              // records 4 .. 7 of this iteration
            rec += 8;
            const int s1a = (rec == NREC) ? snext : s0b + 4096;     // past the set's last record: the next set's first ones
            const int s1b = s1a + 4096;
            const float* bp = b0 + 32 * it;
            l2a_static_for<0, 8>([&](auto iv) {
                constexpr int I = decltype(iv)::value;
                // record 8 it + I + 7 -> the slot that was consumed a step ago
                if constexpr (I == 0) issue_a(s0b, std::integral_constant<int, 3072>(), std::integral_constant<int, 7>());
                else if constexpr (I <= 4) issue_a(s1a, std::integral_constant<int, (I - 1) * 1024>(), std::integral_constant<int, I - 1>());
                else issue_a(s1b, std::integral_constant<int, (I - 5) * 1024>(), std::integral_constant<int, I - 1>());
                // activations two steps ahead (past the phase's last record: a harmless read of what follows in LDS)
                issue_b(bp, std::integral_constant<int, 4 * (I + 2)>(), std::integral_constant<int, (I + 2) & 3>());
                // (the LAST-requested operands first: one s_waitcnt per kind and step - see l2a_lstm_micro_body)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int tl = UW - 1; tl >= 0; --tl)
#pragma unroll
                        for (int c = MT - 1; c >= 0; --c)
                            acc[tl][c] = L2A_MFMA4(ra[I][tl][e], rb[I & 3][NB == 1 ? 0 : tl][c][e], acc[tl][c]);
                l2a_micro_hint<UW, NB * MT, 4 * UW * MT>();
            });
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // bias, activation, write-out of a layer's tiles (this wave's columns of `dst`, every micro tile's rows)
    auto epilogue = [&](const f32x4 (&acc)[UW][MT], const float* bl, float* dst) {
#pragma unroll
        for (int tl = 0; tl < UW; ++tl) {
            const int col = 64 * (wave * UW + tl) + 4 * b;
            const f32x4 bias = *reinterpret_cast<const f32x4*>(bl + col);
#pragma unroll
            for (int c = 0; c < MT; ++c)
                *reinterpret_cast<f32x4*>(dst + (4 * c + j) * ROWF + col) = l2a_actv<GACT>(acc[tl][c] + bias, p.hidden_act, p.hid_floor);
        }
    };
    using CSX = std::integral_constant<int, 4 * XROWF>;
    using CSR = std::integral_constant<int, 4 * ROWF>;
    int pp = 0;                                         // parity of the partials buffer
    __syncthreads();                                    // every micro tile's input rows are written
    L2A_MTS_AT(0, 10)

    for (int t = 0; t < p.h; ++t) {
        L2A_MTS(0)
        f32x4 dsum = (f32x4){0.f, 0.f, 0.f, 0.f}, dgrp = (f32x4){0.f, 0.f, 0.f, 0.f};

        for (int i = 0; i < e_loop; ++i) {
            const float* cs = cst + i * CST;
            f32x4 acc[UW][MT];
            // ---- layer 0 -> rows A ---------------------------------------------------------------------------------
            run_phase(xs + (i * 12 + j) * XROWF, KG0E >> 1, CSX(), I1(), acc);
            if (i == 0) { L2A_MTS(3) }
            epilogue(acc, cs + CST_B, rows);
            if (i == 0) { L2A_MTS(4) }
            // ---- hidden layers ----------------------------------------------------------------------------------------
            for (int l = 1; l < n_hidden; ++l) {
                __syncthreads();
                run_phase(rows + ((l - 1) & 1) * 12 * ROWF + j * ROWF, HI, CSR(), I1(), acc);
                epilogue(acc, cs + CST_B + l * H, rows + (l & 1) * 12 * ROWF);
            }
            // ---- output layer over this wave's own columns of the last rows (its two chunks of the canonical tree) -------
            if (i == 0) { L2A_MTS(5) }
            const float* hl = rows + ((n_hidden - 1) & 1) * 12 * ROWF + j * ROWF + 64 * UW * wave;
            f32x4 oacc[NSEQ][UW][MT];
#pragma unroll
            for (int sq = 0; sq < NSEQ; ++sq) run_phase(hl + 32 * sq, CIT, CSR(), std::integral_constant<int, UW>(), oacc[sq]);
            f32x4* pb = pbuf + pp * (4 * 3 * 64);
            pp ^= 1;
#pragma unroll
            for (int c = 0; c < MT; ++c)
                pb[(wave * MT + c) * 64 + lane] = (UW == 2) ? oacc[0][0][c] + oacc[0][UW - 1][c] : oacc[0][0][c] + oacc[NSEQ - 1][0][c];
            // the set's stream is through: the ring already holds the next set's first records
            rec = 0;
            sbase = snext;
            snext = set_of((i + 2 < e_loop) ? i + 2 : (i + 2 - e_loop < e_loop ? i + 2 - e_loop : 0)) * set_bytes;
            // (the coming steps' actions: behind the last set's output product, two barriers ahead of the next step - l2a_lstm_micro_body)
            if (i == e_loop - 1) {
                av[0] = av_next[0]; av[1] = av_next[1];
                load_actions((t + 2 < p.h) ? t + 2 : p.h - 1, av_next);
            }
            if (i == 0) { L2A_MTS(6) }
            __syncthreads();
            if (i == 0) { L2A_MTS(7) }

            // ---- canonical reduce, output activation, denormalisation; the set joins its ensemble group ------------------
            if (i == e_half) { dsum = dgrp; dgrp = (f32x4){0.f, 0.f, 0.f, 0.f}; }      // group A complete: park it, start group B
            {
                const f32x4 omu = *reinterpret_cast<const f32x4*>(cs + 32 * KG0 + 4 * b);
                const f32x4 osd = *reinterpret_cast<const f32x4*>(cs + 32 * KG0 + 64 + 4 * b);
                const f32x4 bias = *reinterpret_cast<const f32x4*>(cs + 32 * KG0 + 128 + 4 * b);
                f32x4 part[4];
#pragma unroll
                for (int w = 0; w < 4; ++w) part[w] = pb[(w * MT + ct) * 64 + lane];
                __builtin_amdgcn_sched_barrier(0);
                {
                    f32x4 s = (part[0] + part[1]) + (part[2] + part[3]);
                    if (p.m_o4) {
                        // dims 16 .. 19: blocks 4 .. 7 hold the sums over the four quarters of the hidden units -> (Q0 + Q1) + (Q2 + Q3)
                        // (valid in block 4, the row's first: the lanes that hold dims 16 .. 19 of the state)
                        f32x4 t2;
#pragma unroll
                        for (int ii = 0; ii < 4; ++ii) t2[ii] = l2a_row_quarter_sum(s[ii]);
                        if (b >= 4 && b < 8) s = t2;
                    }
                    s = l2a_actv<GACT>(s + bias, p.output_act, p.out_floor);
                    dgrp += s * osd + omu;
                }
            }
        }
        L2A_MTS(11)

        // ---- group A + group B, ensemble mean, reward, state update (this wave's micro tile) ----------------------------------
        const float disc_t = (float)disc_pow;
        disc_pow *= p.discount;
        {
            f32x4 d = dsum + dgrp;
            if (e_loop > 1) {
                // d / E, correctly rounded, +-inf kept (Markstein; l2a_mfma.h)
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    const float q = d[ii] * e_inv;
                    const float qc = fmaf(fmaf(-q, e_count, d[ii]), e_inv, q);
                    d[ii] = (fabsf(q) < INFINITY) ? qc : q;
                }
            }
            const f32x4 nx = st + d;
            // reward in the 16-candidate kernel's order: quarter partials r_qq (lanes b = qq < 4), (r0 + r1) + (r2 + r3)
            float plin = ((qq == 0) ? p.rw.alive : 0.0f) - p.rw.ctrl_coef * asq;
            float psq = 0.0f;
            const int vi = p.rw.vel_index;
            const float dsel = (vi & 2) ? ((vi & 1) ? d[3] : d[2]) : ((vi & 1) ? d[1] : d[0]);
            const float dvel = l2a_from_row(dsel, vi >> 4);         // block (vi >> 2) & 3 of row 0 <- the lane that holds dim vel_index
            if (qq == ((vi >> 2) & 3)) plin += p.rw.w_vel * dvel * p.rw.inv_dt;
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int dim = 4 * b + ii;
                const bool in_dist = (p.rw.dist_coef != 0.0f) && (dim >= p.rw.dist_index) &&
                                     (dim < p.rw.dist_index + 3) && (dim < obs_dim);
                psq += in_dist ? nx[ii] * nx[ii] : 0.0f;
            }
            st = nx;
            plin = l2a_row_quarter_sum(plin);
            float r = plin;
            if (p.rw.dist_coef != 0.0f) {
                psq = l2a_sum_xor32(l2a_sum_xor16(psq));                    // over the obs tiles of a quarter (one is non-zero)
                psq = l2a_row_quarter_sum(psq);
                r -= p.rw.dist_coef * sqrtf(psq);
            }
            ret = fmaf(disc_t, r, ret);
        }
        // the next step's inputs
        L2A_MTS(12)
        write_x();
        __syncthreads();                                // every micro tile's input rows are written
        L2A_MTS(13)
    }

    // ---- results: lanes of block 0 of wave c hold the returns of the candidates cand0 + 4 c + j; the keys meet in LDS ------
    {
        unsigned long long key = 0ull;
        if (valid && b == 0 && wave < MT) {
            if (p.returns_out) p.returns_out[(long long)env * p.n + cand] = ret;
            key = l2a_key_pack(ret, p.cand_offset + cand);
        }
        if (p.best_key) {
#pragma unroll
            for (int off = 2; off >= 1; off >>= 1) {
                const unsigned int hi = __shfl_xor((unsigned int)(key >> 32), off);
                const unsigned int lo = __shfl_xor((unsigned int)(key & 0xffffffffu), off);
                const unsigned long long other = ((unsigned long long)hi << 32) | lo;
                key = (other > key) ? other : key;
            }
            unsigned long long* kbuf = reinterpret_cast<unsigned long long*>(pbuf);     // (the partials are through)
            if (lane == 0) kbuf[wave] = (wave < MT) ? key : 0ull;
            __syncthreads();
            if (tid == 0) {
#pragma unroll
                for (int w = 1; w < 4; ++w) key = (kbuf[w] > key) ? kbuf[w] : key;
                if (key != 0ull) atomicMax(p.best_key + env, key);
                l2a_publish_result(p, p.done_total > 0 ? p.done_total : (int)gridDim.x);
            }
        }
    }
}

// Workgroup -> (env, first candidate, micro tiles) as in l2a_lstm_micro_k; logical ids are contiguous per XCD
// (l2a_logical_wg), so that the workgroups of one env - one weight set in per-block mode - share an XCD's L2.
template <int UW, bool GACT>
__global__ void __launch_bounds__(256) l2a_mlp_micro_k(const L2AKParams p) {
    extern __shared__ __attribute__((aligned(16))) char l2a_smem[];
    const int bid = l2a_logical_wg((int)blockIdx.x, (int)gridDim.x);
    const int env = bid / p.mc_w;
    const int idx = bid - env * p.mc_w;
    const int mt = idx < p.mc_r ? p.mc_hi : p.mc_hi - 1;
    const int q0 = idx < p.mc_r ? idx * p.mc_hi : p.mc_r * p.mc_hi + (idx - p.mc_r) * (p.mc_hi - 1);
#ifdef L2A_TIMELINE
    unsigned long long wg_r0_;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wg_r0_) : : "memory");
#endif
    if (mt == 3) l2a_mlp_micro_body<3, UW, GACT>(p, env, p.c_lo + 4 * q0, l2a_smem);
    else if (mt == 2) l2a_mlp_micro_body<2, UW, GACT>(p, env, p.c_lo + 4 * q0, l2a_smem);
    else l2a_mlp_micro_body<1, UW, GACT>(p, env, p.c_lo + 4 * q0, l2a_smem);
#ifdef L2A_TIMELINE
    if (p.dbg && threadIdx.x == 0) {        // per-workgroup record behind the phase stamps: lifetime (100 MHz real time), XCD, size
        unsigned long long wg_r1_;
        unsigned int xcc_;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wg_r1_) : : "memory");
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));
        unsigned long long* r = p.dbg + (long long)p.h * 4 * 16 + 64 + (long long)blockIdx.x * 4;
        r[0] = wg_r0_; r[1] = wg_r1_; r[2] = xcc_ & 15; r[3] = (unsigned long long)mt | ((unsigned long long)env << 8);
    }
#endif
}
