// l2a_lstm.h - fused rollout of the RECURRENT planner (gfx950 / CDNA4 only).
//
// One launch = the loop body of `RNNMPCController.get_rs_action`
// (reference policies/rnn_mpc_controller.py:112-134): every candidate action sequence is rolled
// through a single-layer LSTM dynamics model (`RNNDynamicsModel.predict`,
// dynamics/rnn_dynamics.py:233-252; cell = tf.nn.rnn_cell.LSTMCell, dynamics/core/utils.py:192-236)
// over the whole horizon, with the env's hidden state repeated to its candidates
// (`repeat_hidden`, :165-187), the closed-form reward accumulated and the arg-max reduced.
//
// Per (candidate, step):
//     z = [norm(obs) | norm(act) | h] @ kernel + bias          kernel [in + U, 4 U], gates i j f o
//     c = sigmoid(f + 1) * c + sigmoid(i) * act(j);   h = sigmoid(o) * act(c)
//     obs += denorm(h @ Wout + bout)
//
// Two kernels:
//  * l2a_lstm_mfma_k  - fp32 MFMA path for U = 128 / 256 / 512 (same fragment scheme as
//                       l2a_mfma.h: D = W^T-tile x activations, the D fragment of the gate GEMM is the
//                       B fragment of the next step's h k-groups and of the output layer).
//                       Workgroup = 4 waves, NT tiles of 16 candidates.  Wave w owns unit tiles
//                       [w UTW, (w+1) UTW) for ALL FOUR gates (16 UTW accumulator registers), so the
//                       gate arithmetic is register-local and the cell state c never leaves registers;
//                       h travels through a double-buffered LDS region; ONE barrier per step.
//  * l2a_lstm_valu_k  - generic fp32 VALU path (any U); baseline + fallback.
//
// Unit-tile split (template parameter SPLIT, plans of at most CUs / 2 candidate tiles): workgroup g of a tile owns
// the lower / upper half of the unit tiles - every wave UTW / 2 of them, all four gates - i.e. half of the gate GEMM's
// output columns over the FULL K range (x, then all of h: the same per-tile summation order as the unsplit kernel,
// so both give the same bits).  Per step the two workgroups swap (a) their halves of the new h - each wave publishes
// its own tiles right after the gate arithmetic, so they travel under the output layer - and (b) the output layer's
// half sums S_g = (c0+c1)+(c2+c3) over their own unit chunks; total = S_0 + S_1 is the unsplit kernel's canonical
// reduce.  Same self-validating {tag, v, tag, v} granules, bounded spin and status word as l2a_mfma.h.
#pragma once

#include "l2a_kernels.h"
#include "l2a_mfma.h"

#define L2A_RNN_MAX_LAYERS 4

struct L2ALstmParams {
    // ---- model -------------------------------------------------------------------------
    const float* wblk;
    long long raw_wk, raw_bk;   // TF kernel [in + U, 4U] row-major, bias [4U]
    long long raw_wo, raw_bo;   // output layer [U, obs_dim], [obs_dim]
    long long pk_wg;            // packed gate matrix [4 UT tiles][KG0 + UT k-groups][64][4]
    long long pk_wout;          // packed output layer [OT][UT][64][4]
    long long pk_bout;          // output bias padded to 16 OT
    long long nm_off;           // [in_mu 16 KG0][in_inv 16 KG0][out_mu 16 OT][out_sd 16 OT]
    long long pk_mg, pk_mo;     // micro-tile kernel (l2a_micro.h): gate matrix / output layer in its fragment order
    int mc_w, mc_r, mc_hi;      // micro-tile launch: workgroups per env, how many of them take mc_hi micro tiles (the others mc_hi - 1)
    int obs_dim, act_dim, in_dim, units;
    int cell_act, output_act;
    int KG0, OT;
    // generic stacks (l2a_rnn_create; l2a_rnn_valu.h): `units` is then the state width sum(layer_units)
    int n_layers, cell_type;
    int layer_units[L2A_RNN_MAX_LAYERS];
    long long layer_w[L2A_RNN_MAX_LAYERS][2], layer_b[L2A_RNN_MAX_LAYERS][2];   // offsets of the layer's kernels / biases
    long long layer_pk[L2A_RNN_MAX_LAYERS][2];  // ... of their copies in MFMA fragment order (l2a_rnn_mfma.h), pk_wout alike
    long long layer_mk[L2A_RNN_MAX_LAYERS][2];  // ... and in the micro-tile kernel's order (l2a_rnn_micro.h; its output layer: pk_mo)
    // ---- launch ------------------------------------------------------------------------
    const float* obs0;          // [m, obs_dim] (or [R, obs_dim] when obs_per_row)
    const float* c0;            // [m, U] (or [R, U] when hid_per_row)
    const float* h0;
    const float* actions;       // [h, m*n, act_dim]
    float* returns_out;         // [m, n] or null
    unsigned long long* best_key;
    float* state_out;           // [m*n, obs_dim] or null
    float* c_out;               // [m*n, U] or null
    float* h_out;
    const float* ret_in;        // [m, n] returns of earlier horizon chunks, or null (= 0)
    double disc0;               // discount ** (first horizon step of this launch)
    int obs_per_row, hid_per_row;
    int m, n, h;
    int tiles_per_env;
    int cand_offset;
    double discount;
    l2a_reward rw;
    unsigned long long* dbg;
    // ---- unit-tile split (SPLIT instances): two workgroups share a candidate tile, see l2a_lstm_mfma_k ----
    int split;                  // 0 | 1
    unsigned int xtag;          // per-launch tag base (launch nonce << 12); tag = xtag + t + 1
    unsigned long long* xbuf;   // exchange granules [pair][group][slot][UT / 2 + OT][2][64] x 16 B
    unsigned int* status;       // host-visible word; bit 0 set = exchange timed out
    unsigned int spin_limit;    // polls one wave may spend waiting for its partner, per launch
    // ---- result mailbox (l2a_lstm_plan_rs_sync; same protocol as L2AKParams, l2a_kernels.h) ----
    unsigned int* done_ctr;
    unsigned long long* mail_keys;
    unsigned long long* mail_seq_ptr;   // null = no mailbox
    unsigned long long mail_seq;
    unsigned long long* next_keys;
};

// Gate-matrix tile order: tile T = w * (4 UTW) + q * UTW + uu  <->  gate q (i, j, f, o) of unit tile
// u = w * UTW + uu.  k order: the KG0 input k-groups (in_dim padded to 16 KG0), then the UT k-groups of h.
__host__ __device__ inline void l2a_lstm_pack_decode(long long idx, int KG0, int UT, int in_dim, int* k_tf, int* col_tf) {
    const int KG = KG0 + UT;
    const int UTW = UT / L2A_NW;
    const int U = 16 * UT;
    const int ii = (int)(idx & 3);
    const int lane = (int)((idx >> 2) & 63);
    const long long rest = idx >> 8;
    const int g = (int)(rest % KG);
    const int T = (int)(rest / KG);
    const int w = T / (4 * UTW);
    const int q = (T - w * 4 * UTW) / UTW;
    const int uu = T - w * 4 * UTW - q * UTW;
    const int k = 16 * g + 4 * (lane >> 4) + ii;
    if (k < 16 * KG0) *k_tf = (k < in_dim) ? k : -1;
    else *k_tf = in_dim + (k - 16 * KG0);
    *col_tf = q * U + 16 * (w * UTW + uu) + (lane & 15);
}

// Phase timeline for tools/timeline_lstm.py: the waves of workgroup 0 stamp the shader clock,
// dbg[((t * 4 + wave) * 16 + slot].  One uniform branch per stamp, pinned by scheduling barriers.
#ifndef L2A_TIMELINE
#define L2A_LTS(slot)
#else
#define L2A_LTS(slot)                                                                       \
    if (p.dbg && bid == 0) {                                                                \
        unsigned long long ts_;                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_) : : "memory");       \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        if (lane == 0) p.dbg[((long long)t * 4 + wave) * 16 + (slot)] = ts_;                \
    }
#endif

__device__ __forceinline__ float l2a_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// Gate transcendentals of the MFMA kernel: one v_exp_f32 + one v_rcp_f32 each (both ~1 ulp), i.e.
// ~1e-7 ABSOLUTE error on values in [-1, 1] - the libm forms cost ~900 cycles per (unit, candidate)
// and made the gate phase 24 % of a step (tools/timeline_lstm.py).  Saturate correctly: exp2 -> 0 / inf.
__device__ __forceinline__ float l2a_fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float l2a_fast_tanh(float x) {
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177792681f * x)), 1.0f);
}

// ------------------------------------------------------------------------------------------
// h-part of the gate GEMM for TW of this wave's gate tiles: acc += Wg[:, h k-groups] x h.
// Same 4-buffer / distance-2 software pipeline as l2a_hidden_gemm (l2a_mfma.h); aA / aB arrive
// preloaded with h k-groups 0 / 1; `tail2` / `tail3` run in the last two refill slots (they fetch the
// next phase's first operands instead of reloading the final k-group).
// ------------------------------------------------------------------------------------------
template <int NT, int TW, int HT, class F2, class F3>
__device__ __forceinline__ void l2a_lstm_gemm(__amdgpu_buffer_rsrc_t rs, const int (&voff)[TW], int sbase, const f32x4* hin,
                                              f32x4 (&aA)[TW], f32x4 (&aB)[TW], f32x4 (&acc)[NT][TW], int lane,
                                              F2 tail2, F3 tail3) {
    static_assert(HT % 4 == 0, "the k-group pipeline is unrolled by 4");
    f32x4 aC[TW], aD[TW], bA[NT], bB[NT], bC[NT], bD[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        bA[nt] = hin[(nt * HT + 0) * 64 + lane];
        bB[nt] = hin[(nt * HT + 1) * 64 + lane];
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { L2A_OPAQUE(bA[nt]); L2A_OPAQUE(bB[nt]); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int g = 0; g < HT - 4; g += 4) {
        const int soff = sbase + (g + 2) * 1024;     // sbase: byte offset of this call's first k-group (wave-uniform)
        L2A_STAGE(aA, bA, aC, bC, soff, 0, g + 2)
        L2A_STAGE(aB, bB, aD, bD, soff, 1024, g + 3)
        L2A_STAGE(aC, bC, aA, bA, soff, 2048, g + 4)
        L2A_STAGE(aD, bD, aB, bB, soff, 3072, g + 5)
    }
    L2A_STAGE(aA, bA, aC, bC, sbase + (HT - 2) * 1024, 0, HT - 2)
    L2A_STAGE(aB, bB, aD, bD, sbase + (HT - 2) * 1024, 1024, HT - 1)
    {
        tail2();
        L2A_STAGE_MFMA(aC, bC)
        __builtin_amdgcn_sched_barrier(0);
    }
    {
        tail3();
        L2A_STAGE_MFMA(aD, bD)
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int NT, int UTW, int OT, int KG0, bool SPLIT>
__global__ void __launch_bounds__(64 * L2A_NW) l2a_lstm_mfma_k(const L2ALstmParams p) {
    static_assert(UTW % 2 == 0, "a wave's 4 UTW gate tiles are processed in passes of 8");
    static_assert(!SPLIT || NT == 1, "the unit-tile split shares single candidate tiles");
    constexpr int UT = L2A_NW * UTW;            // unit tiles = h k-groups
    constexpr int U = 16 * UT;
    constexpr int KG = KG0 + UT;                // k-groups of the gate matrix
    constexpr int UTWS = SPLIT ? UTW / 2 : UTW; // unit tiles this wave owns
    constexpr int GTW = 4 * UTWS;               // gate tiles per wave
    constexpr int TW = GTW < 8 ? GTW : 8;       // gate tiles per pass
    constexpr int NP = GTW / TW;                // passes
    constexpr int HT = UT;                      // (name used by the L2A_STAGE macros: LDS stride)
    // LDS: h fragments, double buffered; output-layer chunk partials, double buffered; constants
    constexpr int HB = NT * UT * 64;            // f32x4 per h buffer
    constexpr int PB = L2A_NW * NT * OT * 64;           // one output-layer partial per wave (the canonical tree: l2a_mfma.h)
    constexpr int CST_BOUT = 32 * KG0 + 32 * OT;
    constexpr int CST_BG = CST_BOUT + 16 * OT;  // gate bias [4 U], TF order
    extern __shared__ __attribute__((aligned(16))) char l2a_smem[];
    f32x4* hbuf = reinterpret_cast<f32x4*>(l2a_smem);
    f32x4* pbuf = hbuf + 2 * HB;
    float* nrm = reinterpret_cast<float*>(pbuf + 2 * PB);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int jc = lane & 15;
    const int qq = lane >> 4;
    const int bid = l2a_logical_wg(blockIdx.x, gridDim.x);
    const int n_tiles = p.m * p.tiles_per_env;
    const int grp = SPLIT ? bid / n_tiles : 0;              // which of the two workgroups of a tile
    const int pairid = SPLIT ? bid - grp * n_tiles : bid;   // the candidate tile
    const int u0 = SPLIT ? grp * (UT / 2) + wave * UTWS : wave * UTW;   // first unit tile of this wave
    const int env = pairid / p.tiles_per_env;
    const int tb = pairid - env * p.tiles_per_env;
    const int R = p.m * p.n;
    const int obs_dim = p.obs_dim, act_dim = p.act_dim;
#ifdef L2A_TIMELINE
    unsigned long long wg_t0_, wg_r0_;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(wg_t0_), "=s"(wg_r0_) : : "memory");
#endif

    int cand[NT], row[NT];
    bool valid[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        cand[nt] = tb * (16 * NT) + nt * 16 + jc;
        valid[nt] = cand[nt] < p.n;
        row[nt] = env * p.n + (valid[nt] ? cand[nt] : p.n - 1);
    }

    for (int i = tid; i < CST_BG + 4 * U; i += 64 * L2A_NW) {
        float v;
        if (i < CST_BOUT) v = p.wblk[p.nm_off + i];
        else if (i < CST_BG) v = p.wblk[p.pk_bout + (i - CST_BOUT)];
        else v = p.wblk[p.raw_bk + (i - CST_BG)];
        nrm[i] = v;
    }

    // ---- state, cell state (registers) and h (LDS buffer 0) ----------------------------------
    f32x4 st[NT][OT], creg[NT][UTWS];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float* orow = p.obs0 + (p.obs_per_row ? (long long)row[nt] : (long long)env) * obs_dim;
#pragma unroll
        for (int c = 0; c < OT; ++c)
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int dim = 16 * c + 4 * qq + ii;
                const float v = orow[dim < obs_dim ? dim : obs_dim - 1];
                st[nt][c][ii] = (dim < obs_dim) ? v : 0.0f;
            }
        const long long hrow = (p.hid_per_row ? (long long)row[nt] : (long long)env) * U;
#pragma unroll
        for (int uu = 0; uu < UTWS; ++uu)
            creg[nt][uu] = *reinterpret_cast<const f32x4*>(p.c0 + hrow + 16 * (u0 + uu) + 4 * qq);
#pragma unroll
        for (int uu = 0; uu < UTW; ++uu)        // the whole h goes to LDS (split: both halves, no exchange needed yet)
            hbuf[(nt * UT + wave * UTW + uu) * 64 + lane] =
                *reinterpret_cast<const f32x4*>(p.h0 + hrow + 16 * (wave * UTW + uu) + 4 * qq);
    }
    __syncthreads();

    const int ga0 = obs_dim >> 4;
    f32x4 av_next[NT][2];
    // (raw buffer loads from a per-step descriptor; slots that hold no action point past the slab and read 0.0 - no lane
    // mask, no select: see l2a_mfma.h)
    int aoff[NT][2][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int ka = 16 * (ga0 + s) + 4 * qq + ii - obs_dim;
                const bool in = (ka >= 0) && (ka < act_dim);
                aoff[nt][s][ii] = in ? (row[nt] * act_dim + ka) * 4 : 0x7ffffff0;
            }
    const long long a_step = (long long)R * act_dim;
    auto load_actions = [&](int t, f32x4 (&dst)[NT][2]) {
        const __amdgpu_buffer_rsrc_t ars = l2a_rsrc(p.actions + (long long)t * a_step, a_step * 4);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int ii = 0; ii < 4; ++ii)
                    dst[nt][s][ii] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ars, aoff[nt][s][ii], 0, 0));
    };
    load_actions(0, av_next);

    float ret[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) ret[nt] = p.ret_in ? p.ret_in[(long long)env * p.n + (valid[nt] ? cand[nt] : p.n - 1)] : 0.0f;
    double disc_pow = p.disc0;

    const long long wg_bytes = (long long)4 * UT * KG * 1024, wo_bytes = (long long)OT * UT * 1024;
    const __amdgpu_buffer_rsrc_t rs = l2a_rsrc(p.wblk + p.pk_wg, wg_bytes);
    const __amdgpu_buffer_rsrc_t rs_out = l2a_rsrc(p.wblk + p.pk_wout, wo_bytes);
    // byte offset of this lane in gate tile (pass, tt): x k-groups start at +0, h k-group k at + (KG0 + k) KiB (added
    // as a wave-uniform scalar offset: one offset array serves both parts)
    int voffx[NP][TW];
#pragma unroll
    for (int ps = 0; ps < NP; ++ps)
#pragma unroll
        for (int tt = 0; tt < TW; ++tt) {
            // local gate tile lt = q * UTWS + uu <-> gate q of unit tile u = u0 + uu, stored as packed tile
            // (u / UTW) * 4 UTW + q * UTW + u % UTW (l2a_lstm_pack_decode); unsplit: wave * GTW + lt
            const int lt = ps * TW + tt, q = lt / UTWS, u = u0 + (lt - q * UTWS);
            voffx[ps][tt] = lane * 16 + ((u / UTW) * (4 * UTW) + q * UTW + (u % UTW)) * KG * 1024;
        }
    f32x4 pfA[TW], pfB[TW];         // first two k-groups of the upcoming half-GEMM
    f32x4 pfO[UTWS][OT];            // output-layer A fragments of this wave's unit tiles
    f32x4 hreg[NT][UTWS];
    // exchange (SPLIT): regions of 2 KiB = 2 granules x 64 lanes x 16 B; [0, UT / 2) = the group's h tiles, then OT
    // tiles of the output-layer half sum
    constexpr int XREG = (UT / 2 + OT) * 2048;
    const __amdgpu_buffer_rsrc_t xrs = l2a_rsrc(p.xbuf, SPLIT ? (long long)n_tiles * 4 * XREG : 16);
    unsigned int spin_left = p.spin_limit;

    // K order of a gate tile (the same in both launch geometries, so both give the same bits): the h k-groups of
    // the HALF that contains the tile's own units, then the other half's, then the x k-groups.  A wave's unit
    // tiles lie in one half (UT / 2 = 2 UTW), so the order is wave-uniform: own half at k-group kb_own.
    constexpr int HH = UT / 2;
    const int kb_own = (u0 >= HH) ? HH : 0, kb_oth = HH - kb_own;
    const int sb_own = (KG0 + kb_own) * 1024, sb_oth = (KG0 + kb_oth) * 1024;      // scalar byte offsets of the halves
#pragma unroll
    for (int tt = 0; tt < TW; ++tt) {       // first operands of (step 0, pass 0, own half)
        pfA[tt] = l2a_ldw(rs, voffx[0][tt], sb_own);
        pfB[tt] = l2a_ldw(rs, voffx[0][tt], sb_own + 1024);
    }

    // The reduce / reward / state update of step t is deferred into iteration t + 1, behind that step's h-part
    // (which needs h, not the state): in the SPLIT geometry the partner's half sum then has a whole GEMM to arrive.
    float asq_prev[NT], disc_prev = 0.0f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) asq_prev[nt] = 0.0f;

    unsigned int xtag = 0;
    auto xbase = [&](int g, int slot) { return ((pairid * 2 + g) * 2 + slot) * XREG + lane * 16; };
    auto xput = [&](int r, int slot, const f32x4& v) {     // region r of this group: one f32x4 per lane
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            u32x4 g;
            g.x = xtag; g.y = __float_as_uint(v[2 * hh]);
            g.z = xtag; g.w = __float_as_uint(v[2 * hh + 1]);
            __builtin_amdgcn_raw_buffer_store_b128(g, xrs, xbase(grp, slot) + (r * 2 + hh) * 1024, 0, L2A_SC1);
        }
    };
    auto xget = [&](int r, int slot, unsigned int tag, f32x4& v) {
        bool ok = true;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const u32x4 g = __builtin_amdgcn_raw_buffer_load_b128(xrs, xbase(grp ^ 1, slot) + (r * 2 + hh) * 1024, 0, L2A_SC1);
            v[2 * hh] = __uint_as_float(g.y);
            v[2 * hh + 1] = __uint_as_float(g.w);
            ok = ok && (g.x == tag) && (g.z == tag);
        }
        return ok;
    };
    auto spin = [&](bool ok) {      // true: stop polling (all granules valid, or the launch's budget is spent)
        if (__all(ok)) return true;
        if (spin_left == 0) {       // partner never arrived: flag it, do not hang
            if (lane == 0) __hip_atomic_fetch_or(p.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return true;
        }
        --spin_left;
        __builtin_amdgcn_s_sleep(4);
        return false;
    };

    // Deferred tail of step `tp` (its chunk partials are in pbuf[tp & 1]): canonical reduce, output activation,
    // denormalisation, reward, state update.  SPLIT: S_0 + S_1 with the partner's half sum from the exchange.
    f32x4 ps_[OT];                  // SPLIT: the partner's half sum of the step being finished
#pragma unroll
    for (int c = 0; c < OT; ++c) ps_[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto finish_step = [&](int tp, bool have_partner_sum) {
        const f32x4* pbp = pbuf + (tp & 1) * PB;
        f32x4 xsum[OT];
        if (SPLIT) {
            const unsigned int tag = p.xtag + (unsigned int)(tp + 1);
            while (!have_partner_sum) {
                bool ok = true;
#pragma unroll
                for (int c = 0; c < OT; ++c) ok = xget(HH + c, tp & 1, tag, ps_[c]) && ok;
                if (spin(ok)) break;
            }
#pragma unroll
            for (int c = 0; c < OT; ++c) {
                f32x4 part[L2A_NW];
#pragma unroll
                for (int ch = 0; ch < L2A_NW; ++ch) part[ch] = pbp[(ch * OT + c) * 64 + lane];
                __builtin_amdgcn_sched_barrier(0);
                const f32x4 sv = (part[0] + part[1]) + (part[2] + part[3]);
                xsum[c] = (grp == 0) ? sv + ps_[c] : ps_[c] + sv;          // S_0 + S_1
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float plin = ((qq == 0) ? p.rw.alive : 0.0f) - p.rw.ctrl_coef * asq_prev[nt];
            float psq = 0.0f;
#pragma unroll
            for (int c = 0; c < OT; ++c) {
                const f32x4 bias = *reinterpret_cast<const f32x4*>(nrm + CST_BOUT + 16 * c + 4 * qq);
                const f32x4 omu = *reinterpret_cast<const f32x4*>(nrm + 32 * KG0 + 16 * c + 4 * qq);
                const f32x4 osd = *reinterpret_cast<const f32x4*>(nrm + 32 * KG0 + 16 * OT + 16 * c + 4 * qq);
                f32x4 s;
                if (SPLIT) {
                    s = xsum[c];
                } else {
                    // all LDS reads of this obs tile first (left alone the scheduler serialises read-wait-add)
                    f32x4 part[L2A_NW];
#pragma unroll
                    for (int ch = 0; ch < L2A_NW; ++ch) part[ch] = pbp[((ch * NT + nt) * OT + c) * 64 + lane];
                    __builtin_amdgcn_sched_barrier(0);
                    static_assert(L2A_NW == 4, "the canonical tree is written out for four waves");
                    s = (part[0] + part[1]) + (part[2] + part[3]);      // wave w wrote c_2w + c_2w+1: the whole balanced tree
                }
                s = l2a_act4(s + bias, p.output_act);
                const f32x4 d = s * osd + omu;
                const f32x4 nx = st[nt][c] + d;
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    const int dim = 16 * c + 4 * qq + ii;
                    if (dim == p.rw.vel_index) plin += p.rw.w_vel * d[ii] * p.rw.inv_dt;
                    const bool in_dist = (p.rw.dist_coef != 0.0f) && (dim >= p.rw.dist_index) &&
                                         (dim < p.rw.dist_index + 3) && (dim < obs_dim);
                    psq += in_dist ? nx[ii] * nx[ii] : 0.0f;
                }
                st[nt][c] = nx;
            }
            plin = l2a_sum_xor32(l2a_sum_xor16(plin));      // lane-swap sums (l2a_mfma.h): same order as the xor shuffles
            psq = l2a_sum_xor32(l2a_sum_xor16(psq));
            float r = plin;
            if (p.rw.dist_coef != 0.0f) r -= p.rw.dist_coef * sqrtf(psq);
            ret[nt] = fmaf(disc_prev, r, ret[nt]);
        }
    };

    for (int t = 0; t < p.h; ++t) {
        f32x4* hcur = hbuf + (t & 1) * HB;
        f32x4* hnext = hbuf + ((t + 1) & 1) * HB;
        f32x4* pb = pbuf + (t & 1) * PB;
        xtag = p.xtag + (unsigned int)(t + 1);
        f32x4 av[NT][2];
        float asq[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            av[nt][0] = av_next[nt][0];
            av[nt][1] = av_next[nt][1];
            float s = 0.0f;
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                s = fmaf(av[nt][0][ii], av[nt][0][ii], s);
                s = fmaf(av[nt][1][ii], av[nt][1][ii], s);
            }
            asq[nt] = s;
        }
        load_actions((t + 1 < p.h) ? t + 1 : t, av_next);
        L2A_LTS(0)

        f32x4 acc[NP][NT][TW];
#pragma unroll
        for (int ps = 0; ps < NP; ++ps)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int tt = 0; tt < TW; ++tt) acc[ps][nt][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // ---- h part, own half of the k-groups (this workgroup's own units: already in LDS) ---------------------
        // SPLIT: the first sweep for the partner's granules is ISSUED in the second-to-last refill slot of this half
        // (vmcnt retires in order: any earlier and the weight pipeline would wait for the ~2.3k-cycle sc1 round trip),
        // so that it travels under the last two stages' MFMAs instead of in front of an idle matrix pipe (only the other half's
        // k-group-1 operands queue behind it, and those are first used after the sweep has been checked).
        constexpr bool EARLY = true;
        constexpr int NXG = SPLIT ? 2 * (UTWS + OT) : 1;
        u32x4 xg[NXG];
        auto xraw = [&](int r, int slot, int hh) {
            return __builtin_amdgcn_raw_buffer_load_b128(xrs, xbase(grp ^ 1, slot) + (r * 2 + hh) * 1024, 0, L2A_SC1);
        };
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            // the tail fetches the first operands of the next half-GEMM: next pass / first pass of the other half
            const int pn = (ps + 1 < NP) ? ps + 1 : 0;
            const int sn = (ps + 1 < NP) ? sb_own : sb_oth;
            l2a_lstm_gemm<NT, TW, HH>(rs, voffx[ps], sb_own, hcur + kb_own * 64, pfA, pfB, acc[ps], lane,
                [&]() {
#pragma unroll
                    for (int tt = 0; tt < TW; ++tt) pfA[tt] = l2a_ldw(rs, voffx[pn][tt], sn);
                    if (SPLIT && EARLY && ps == NP - 1 && t > 0) {
#pragma unroll
                        for (int uu = 0; uu < UTWS; ++uu)
#pragma unroll
                            for (int hh = 0; hh < 2; ++hh) xg[uu * 2 + hh] = xraw(wave * UTWS + uu, (t - 1) & 1, hh);
#pragma unroll
                        for (int c = 0; c < OT; ++c)
#pragma unroll
                            for (int hh = 0; hh < 2; ++hh) xg[(UTWS + c) * 2 + hh] = xraw(HH + c, (t - 1) & 1, hh);
                    }
                },
                [&]() {     // (these operands are first used after the sweep has been checked)
#pragma unroll
                    for (int tt = 0; tt < TW; ++tt) pfB[tt] = l2a_ldw(rs, voffx[pn][tt], sn + 1024);
                });
        }
        L2A_LTS(1)
        if (SPLIT && t > 0) {
            // the partner's half of h(t) - published right after ITS gate phase of step t - 1, a GEMM ago - and, in
            // the same sweep (an sc1 load round trip costs ~2.3k cycles whatever it fetches), its half sum of
            // step t - 1, which the deferred tail below needs
            const unsigned int tag = p.xtag + (unsigned int)t;
            f32x4 ph[UTWS];
            bool ok = EARLY;
#pragma unroll
            for (int q = 0; EARLY && q < UTWS + OT; ++q) {
                f32x4 v;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const u32x4 g = xg[q * 2 + hh];
                    v[2 * hh] = __uint_as_float(g.y);
                    v[2 * hh + 1] = __uint_as_float(g.w);
                    ok = ok && (g.x == tag) && (g.z == tag);
                }
                if (q < UTWS) ph[q < UTWS ? q : 0] = v; else ps_[q >= UTWS ? q - UTWS : 0] = v;
            }
            while (!spin(ok)) {
                ok = true;
#pragma unroll
                for (int uu = 0; uu < UTWS; ++uu) ok = xget(wave * UTWS + uu, (t - 1) & 1, tag, ph[uu]) && ok;
#pragma unroll
                for (int c = 0; c < OT; ++c) ok = xget(HH + c, (t - 1) & 1, tag, ps_[c]) && ok;
            }
#pragma unroll
            for (int uu = 0; uu < UTWS; ++uu) hcur[((grp ^ 1) * HH + wave * UTWS + uu) * 64 + lane] = ph[uu];
            L2A_LTS(7)
            __syncthreads();
            L2A_LTS(8)
        }
        // ---- h part, the other half ------------------------------------------------------------------------------
        f32x4 pfX[TW];              // x k-group 0 of pass 0
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            if (ps + 1 < NP) {
                const int pn = (ps + 1 < NP) ? ps + 1 : ps;
                l2a_lstm_gemm<NT, TW, HH>(rs, voffx[ps], sb_oth, hcur + kb_oth * 64, pfA, pfB, acc[ps], lane,
                    [&]() {
#pragma unroll
                        for (int tt = 0; tt < TW; ++tt) pfA[tt] = l2a_ldw(rs, voffx[pn][tt], sb_oth);
                    },
                    [&]() {
#pragma unroll
                        for (int tt = 0; tt < TW; ++tt) pfB[tt] = l2a_ldw(rs, voffx[pn][tt], sb_oth + 1024);
                    });
            } else {
                // the last tail fetches x k-group 0 of pass 0: it lands under the deferred tail of step t - 1
                l2a_lstm_gemm<NT, TW, HH>(rs, voffx[ps], sb_oth, hcur + kb_oth * 64, pfA, pfB, acc[ps], lane,
                    [&]() {
#pragma unroll
                        for (int tt = 0; tt < TW; tt += 2) pfX[tt] = l2a_ldw(rs, voffx[0][tt], 0);
                    },
                    [&]() {
#pragma unroll
                        for (int tt = 1; tt < TW; tt += 2) pfX[tt] = l2a_ldw(rs, voffx[0][tt], 0);
                    });
            }
        }
        L2A_LTS(2)

        // ---- deferred tail of step t - 1 -> state(t); then the x part: B = normalised [obs | act] -----------------
        if (t > 0) finish_step(t - 1, true);
        const float disc_t = (float)disc_pow;
        disc_pow *= p.discount;
        f32x4 x[KG0][NT];
#pragma unroll
        for (int g = 0; g < KG0; ++g) {
            const f32x4 mu = *reinterpret_cast<const f32x4*>(nrm + 16 * g + 4 * qq);
            const f32x4 iv = *reinterpret_cast<const f32x4*>(nrm + 16 * KG0 + 16 * g + 4 * qq);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4 sv = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (g < OT) sv = st[nt][g < OT ? g : 0];
                f32x4 aa = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (g == ga0) aa = av[nt][0];
                if (g == ga0 + 1) aa = av[nt][1];
                // state OR action OR padding per slot, each fragment exactly zero outside its own range: the sum is the select
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) x[g][nt][ii] = ((sv[ii] + aa[ii]) - mu[ii]) * iv[ii];
            }
        }
        {
            // pass by pass, pinned: every accumulator is live here, so the x operands must not be hoisted - k-group 0
            // of the NEXT pass and k-groups >= 1 of this pass are fetched under this pass's first 4 TW MFMAs
            f32x4 a0[TW];
#pragma unroll
            for (int tt = 0; tt < TW; ++tt) a0[tt] = pfX[tt];
#pragma unroll
            for (int ps = 0; ps < NP; ++ps) {
                f32x4 ar[KG0 > 1 ? KG0 - 1 : 1][TW], n0[TW];
#pragma unroll
                for (int g = 1; g < KG0; ++g)
#pragma unroll
                    for (int tt = 0; tt < TW; ++tt) ar[g - 1][tt] = l2a_ldw(rs, voffx[ps][tt] + g * 1024, 0);
                if (ps + 1 < NP) {
#pragma unroll
                    for (int tt = 0; tt < TW; ++tt) n0[tt] = l2a_ldw(rs, voffx[(ps + 1 < NP) ? ps + 1 : ps][tt], 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < KG0; ++g)
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                            for (int tt = 0; tt < TW; ++tt)
                                acc[ps][nt][tt] = L2A_MFMA((g == 0 ? a0[tt] : ar[g > 0 ? g - 1 : 0][tt])[ii], x[g][nt][ii],
                                                           acc[ps][nt][tt]);
                if (ps + 1 < NP) {
#pragma unroll
                    for (int tt = 0; tt < TW; ++tt) a0[tt] = n0[tt];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        L2A_LTS(3)
#pragma unroll
        for (int uu = 0; uu < UTWS; ++uu)       // output-layer fragments of this wave's unit tiles: land under the gates
#pragma unroll
            for (int c = 0; c < OT; ++c) pfO[uu][c] = l2a_ldw(rs_out, lane * 16 + (u0 + uu) * 1024, c * UT * 1024);

        // ---- gate arithmetic (register local) -> c, h ------------------------------------------
#pragma unroll
        for (int uu = 0; uu < UTWS; ++uu) {
            f32x4 bias[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                bias[q] = *reinterpret_cast<const f32x4*>(nrm + CST_BG + q * U + 16 * (u0 + uu) + 4 * qq);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4 z[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int lt = q * UTWS + uu;
                    z[q] = acc[lt / TW][nt][lt % TW] + bias[q];
                }
                f32x4 cn, hn;
                if (p.cell_act == L2A_ACT_TANH) {       // wave-uniform; the reference default (rnn_dynamics.py:20)
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) {
                        const float ig = l2a_fast_sigmoid(z[0][ii]);
                        const float jg = l2a_fast_tanh(z[1][ii]);
                        const float fg = l2a_fast_sigmoid(z[2][ii] + 1.0f);     // forget_bias = 1
                        const float og = l2a_fast_sigmoid(z[3][ii]);
                        cn[ii] = fmaf(fg, creg[nt][uu][ii], ig * jg);
                        hn[ii] = og * l2a_fast_tanh(cn[ii]);
                    }
                } else {
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) {
                        const float ig = l2a_fast_sigmoid(z[0][ii]);
                        const float jg = l2a_act1(z[1][ii], p.cell_act);
                        const float fg = l2a_fast_sigmoid(z[2][ii] + 1.0f);
                        const float og = l2a_fast_sigmoid(z[3][ii]);
                        cn[ii] = fmaf(fg, creg[nt][uu][ii], ig * jg);
                        hn[ii] = og * l2a_act1(cn[ii], p.cell_act);
                    }
                }
                creg[nt][uu] = cn;
                hreg[nt][uu] = hn;
                hnext[(nt * UT + u0 + uu) * 64 + lane] = hn;
                if (SPLIT) xput(wave * UTWS + uu, t & 1, hn);      // this half of h(t + 1) travels under the next GEMM
            }
        }

        L2A_LTS(4)
        // ---- output layer: this wave's unit tiles are its k-groups; chunk partials -> LDS -------
        {
            // chunks of CS unit tiles, each its own MFMA chain: a wave owns 2 of the 8 (unsplit) or 1 of its
            // group's 4 (SPLIT: chunk index grp * 4 + wave in the canonical order)
            constexpr int CS = UTW / 2;
            constexpr int NCH = SPLIT ? 1 : 2;
            f32x4 oacc[NCH][NT][OT];
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int c = 0; c < OT; ++c) oacc[ch][nt][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t2 = 0; t2 < CS; ++t2)
#pragma unroll
                for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                            for (int c = 0; c < OT; ++c)
                                oacc[ch][nt][c] = L2A_MFMA(pfO[ch * CS + t2][c][ii], hreg[nt][ch * CS + t2][ii], oacc[ch][nt][c]);
            // one partial per wave; an unsplit wave adds its two chunks here (first level of the canonical tree)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int c = 0; c < OT; ++c)
                    pb[((wave * NT + nt) * OT + c) * 64 + lane] = (NCH == 2) ? oacc[0][nt][c] + oacc[NCH - 1][nt][c] : oacc[0][nt][c];
        }
        L2A_LTS(5)
#pragma unroll
        for (int tt = 0; tt < TW; ++tt) {       // next step's first operands (pass 0, own half), in flight across the barrier
            pfA[tt] = l2a_ldw(rs, voffx[0][tt], sb_own);
            pfB[tt] = l2a_ldw(rs, voffx[0][tt], sb_own + 1024);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) asq_prev[nt] = asq[nt];
        disc_prev = disc_t;
        __syncthreads();
        if (SPLIT && wave == 0) {       // this group's half sum S_g of step t: (c0+c1)+(c2+c3) over its own chunks
#pragma unroll
            for (int c = 0; c < OT; ++c) {
                f32x4 part[L2A_NW];
#pragma unroll
                for (int ch = 0; ch < L2A_NW; ++ch) part[ch] = pb[(ch * OT + c) * 64 + lane];
                __builtin_amdgcn_sched_barrier(0);
                const f32x4 sv = (part[0] + part[1]) + (part[2] + part[3]);
                xput(HH + c, t & 1, sv);
            }
        }
        L2A_LTS(6)
    }
    finish_step(p.h - 1, false);

    // ---- results ----------------------------------------------------------------------------
    if (p.c_out || p.h_out) {       // every wave writes its own unit tiles (predict)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            if (valid[nt]) {
                const long long orow = ((long long)env * p.n + cand[nt]) * U;
#pragma unroll
                for (int uu = 0; uu < UTWS; ++uu) {
                    if (p.c_out) *reinterpret_cast<f32x4*>(p.c_out + orow + 16 * (u0 + uu) + 4 * qq) = creg[nt][uu];
                    if (p.h_out) *reinterpret_cast<f32x4*>(p.h_out + orow + 16 * (u0 + uu) + 4 * qq) = hreg[nt][uu];
                }
            }
    }
    if (wave == 0 && grp == 0) {
        unsigned long long key = 0ull;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (valid[nt] && qq == 0) {
                if (p.returns_out) p.returns_out[(long long)env * p.n + cand[nt]] = ret[nt];
                const unsigned long long k = l2a_key_pack(ret[nt], p.cand_offset + cand[nt]);
                key = (k > key) ? k : key;
            }
            if (p.state_out && valid[nt]) {
                float* srow = p.state_out + ((long long)env * p.n + cand[nt]) * obs_dim;
#pragma unroll
                for (int c = 0; c < OT; ++c)
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) {
                        const int dim = 16 * c + 4 * qq + ii;
                        if (dim < obs_dim) srow[dim] = st[nt][c][ii];
                    }
            }
        }
        if (p.best_key) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const unsigned int hi = __shfl_xor((unsigned int)(key >> 32), off);
                const unsigned int lo = __shfl_xor((unsigned int)(key & 0xffffffffu), off);
                const unsigned long long other = ((unsigned long long)hi << 32) | lo;
                key = (other > key) ? other : key;
            }
            if (lane == 0) {
                if (key != 0ull) atomicMax(p.best_key + env, key);
                l2a_publish_result(p, p.m * p.tiles_per_env);
            }
        }
    }
#ifdef L2A_TIMELINE
    if (p.dbg && wave == 0 && lane == 0) {      // per-workgroup record behind the phase stamps: lifetime and placement
        unsigned long long wg_t1_, wg_r1_;
        unsigned int xcc_;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(wg_t1_), "=s"(wg_r1_) : : "memory");
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));
        unsigned long long* r = p.dbg + (long long)p.h * 4 * 16 + (long long)(grp * n_tiles + pairid) * 6;
        r[0] = wg_t0_; r[1] = wg_t1_; r[2] = xcc_; r[3] = blockIdx.x; r[4] = wg_r0_; r[5] = wg_r1_;   // [4, 5]: 100 MHz real time
    }
#endif
}
