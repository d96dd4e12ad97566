// l2a_rnn_mfma.h - matrix-core rollout of the STACKED recurrent cells of l2a_rnn_valu.h (GRU, BasicRNN, LSTM stacks:
// everything `create_rnn`, reference dynamics/core/utils.py:192-236, builds besides run_rebal.py's single LSTM layer of 128 /
// 256 / 512 units, which has its own tuned kernel in l2a_lstm.h; a single LSTM layer of any other width runs here too).
// Included by l2a_lstm_api.hip only.
//
// Same semantics, same LDS state and the same thread roles for the element-wise parts as l2a_rnn_valu_k (cell arithmetic:
// see that header); what changes is who multiplies.  Every [x | h] K product runs on v_mfma_f32_16x16x4_f32:
//
//   workgroup = 4 waves = one tile of 16 candidates; wave w owns every fourth BLOCK of unit tiles of a layer (16 units a
//   tile; a block = four accumulator tiles: one unit tile x four LSTM gates, two x two GRU gates, four x one), all gates
//   of a unit in one wave, so the gate arithmetic is register local;
//   B operand  = the layer's input / old h as they lie in LDS, row-major [16 candidates][K]: MFMA ii of k-group g multiplies
//                k = 16 g + 4 (lane >> 4) + ii, i.e. a lane's four MFMAs of a k-group read ONE ds_read_b128 (rows padded
//                to a multiple of 16 with zeros + 4 floats of skew against bank conflicts);
//   A operand  = a copy of the reference's TF-layout kernel [K, G U] in fragment order (l2a_rnn_pack_k at set_weights:
//                [unit tile][gate][k-group][lane][4], zero where a row or unit is padding) - one 16-byte buffer load per
//                gate and k-group; reading the TF layout in place (four dword loads of 64-byte half lines each) ran at
//                half the speed: every workgroup streams all weights once per step, and the L1 moves whole 128-byte lines;
//   D          = lane (candidate lane & 15, units 16 T + 4 (lane >> 4) + ii): bias, gates, new h / c written back to the
//                same row-major LDS rows (padding units masked to zero, so they never feed a later product).
//
// The output layer is one more such product (wave c owns obs tile c); normalisation, reward, state update, returns and
// arg-max are the code of l2a_rnn_valu_k.  fp32 MFMA sums in another order than the VALU loop: results agree with
// l2a_rnn_valu_k to rounding (tests/test_rnn.py compares them), not bit for bit.
#pragma once

#include <type_traits>

#include "l2a_lstm.h"
#include "l2a_lstm_valu.h"

// cell nonlinearity: tanh (the reference default, rnn_dynamics.py:20) as one v_exp_f32 + one v_rcp_f32 like the gates
// (l2a_lstm.h: the libm forms made the gate arithmetic a quarter of a step); wave-uniform switch
__device__ __forceinline__ float l2a_rnn_act(float x, int kind) { return kind == L2A_ACT_TANH ? l2a_fast_tanh(x) : l2a_act1(x, kind); }

#define L2A_RNN_SKEW 4      // floats added to every LDS row (rows of 16 k floats would all start in bank 0)

__host__ __device__ inline int l2a_rnn_row(int k) { return 16 * ((k + 15) / 16) + L2A_RNN_SKEW; }

// LDS floats of the kernel for a model (host: launch; device: carve)
// (gates = 4 LSTM, 3 GRU - two gates + candidate -, 1 BasicRNN: bias floats per unit)
__host__ __device__ inline long long l2a_rnn_mfma_lds_floats(int in_dim, int obs_dim, int n_layers, const int* units, int gates) {
    long long f = (long long)L2A_LVT * l2a_rnn_row(in_dim);
    long long nb = 0;
    for (int l = 0; l < n_layers; ++l) { f += 3LL * L2A_LVT * l2a_rnn_row(units[l]); nb += (long long)gates * units[l]; }
    f += 2LL * L2A_LVT * obs_dim + L2A_LVT;                 // state, delta, returns
    f += (long long)L2A_LVT * (in_dim - obs_dim);           // this step's raw actions (reward)
    return f + 2LL * in_dim + 3LL * obs_dim + nb;           // constants: input / output normalisation, output bias, cell biases
}

// Fragment order of a [kin + U, G U] kernel (TF layout: input rows, then recurrent rows; gate q in columns [q U, (q + 1) U)):
// float index (((T * G + q) * KG + g) * 64 + lane) * 4 + ii, KG = KGx + UT k-groups (KGx = ceil(kin / 16) input groups, then
// UT = ceil(U / 16) recurrent groups), = W[row][q U + 16 T + (lane & 15)] with row = 16 g + 4 (lane >> 4) + ii in its part,
// 0 where the row or the unit is padding.  The output layer is the same with U -> obs_dim, G = 1 and no recurrent part.
// The array is padded with zero unit tiles to a multiple of FOUR: the rollout computes blocks of up to four unit tiles per
// product call and addresses a tile through the SGPR offset of its buffer loads, which the hardware's range check does
// not cover - a block that reaches past the last unit tile reads these zeros, not what follows the allocation.
__host__ __device__ inline long long l2a_rnn_pack_floats(int kin, int U, int G, bool recurrent) {
    const int UT = (U + 15) / 16, UTP = (UT + 3) & ~3;
    return (long long)UTP * G * ((kin + 15) / 16 + (recurrent ? UT : 0)) * 256;
}

__global__ void l2a_rnn_pack_k(const float* w, int kin, int U, int G, int recurrent, long long total, float* dst) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int UT = (U + 15) / 16, KGx = (kin + 15) / 16, KG = KGx + (recurrent ? UT : 0);
    const int ii = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
    const long long rest = idx >> 8;
    const int g = (int)(rest % KG);
    const int tq = (int)(rest / KG);
    const int T = tq / G, q = tq - T * G;
    const int unit = 16 * T + (lane & 15);
    int row;
    if (g < KGx) { row = 16 * g + 4 * (lane >> 4) + ii; if (row >= kin) row = -1; }
    else { row = 16 * (g - KGx) + 4 * (lane >> 4) + ii; row = (row < U) ? kin + row : -1; }
    dst[idx] = (row >= 0 && unit < U) ? w[(long long)row * G * U + q * U + unit] : 0.0f;
}

// acc[q] += sum over the KGx input k-groups (B = bx) and the KGh recurrent ones (B = bh) for unit tile T: one 16 x 16 output
// tile per gate.  P: raw-buffer resource over the packed kernel; bx / bh = this lane's B row + 4 (lane >> 4).  The A operands
// of NB - 1 k-groups are in flight ahead of the MFMAs (register ring, all indices static: the compiler tracks the loads).
#define L2A_RNN_NB 4
template <int G>
__device__ __forceinline__ void l2a_rnn_gemm(__amdgpu_buffer_rsrc_t P, int T, int KGx, const float* bx, int KGh, const float* bh,
                                             int lane, f32x4 (&acc)[G]) {
    constexpr int NB = L2A_RNN_NB;
    static_assert(NB == 4, "the stage lists below are written out for a ring of four");
    const int total = KGx + KGh;
    const int last = total - 1;
    const int tile0 = T * G * total * 1024;             // bytes; gate q of this unit tile starts q * total * 1024 further on
    f32x4 a[NB][G], bq[NB];
    // (a request past the last k-group re-reads the last one: nobody consumes it, and the loop body stays free of branches -
    // a conditional load splits the block and the compiler then waits for vmcnt(0) in front of every MFMA group).  The B
    // fragment travels with its A operands: an LDS read issued right in front of its MFMAs costs its ~130 clocks each time
    auto issue = [&](int g, auto slot_tag) {
        constexpr int slot = decltype(slot_tag)::value;
        const int gc = g < last ? g : last;
#pragma unroll
        for (int q = 0; q < G; ++q) a[slot][q] = l2a_ldw(P, lane * 16, tile0 + (q * total + gc) * 1024);
        const float* bp = (gc < KGx) ? bx + 16 * gc : bh + 16 * (gc - KGx);
        bq[slot] = *reinterpret_cast<const f32x4*>(bp);
    };
    // fewer than four gates: a second accumulator per gate takes every other MFMA, so that four independent chains are in
    // the matrix pipe either way (a dependent MFMA waits for its predecessor's eight passes)
    constexpr bool TWO = G <= 2;
    f32x4 acc2[G];
#pragma unroll
    for (int q = 0; q < G; ++q) acc2[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto mfma = [&](int g, auto slot_tag) {
        constexpr int slot = decltype(slot_tag)::value;
        (void)g;
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
            for (int q = 0; q < G; ++q) {
                if (TWO && (ii & 1)) acc2[q] = L2A_MFMA(a[slot][q][ii], bq[slot][ii], acc2[q]);
                else acc[q] = L2A_MFMA(a[slot][q][ii], bq[slot][ii], acc[q]);
            }
    };
    using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>; using S3 = std::integral_constant<int, 3>;
    issue(0, S0()); issue(1, S1()); issue(2, S2());
    const int n4 = total & ~3;
    for (int g0 = 0; g0 < n4; g0 += NB) {
        issue(g0 + 3, S3()); mfma(g0, S0());
        issue(g0 + 4, S0()); mfma(g0 + 1, S1());
        issue(g0 + 5, S1()); mfma(g0 + 2, S2());
        issue(g0 + 6, S2()); mfma(g0 + 3, S3());
    }
    const int r = total - n4;                           // 0 .. 3 k-groups left, already requested
    if (r > 0) mfma(n4, S0());
    if (r > 1) mfma(n4 + 1, S1());
    if (r > 2) mfma(n4 + 2, S2());
    if (TWO) {
#pragma unroll
        for (int q = 0; q < G; ++q) acc[q] += acc2[q];
    }
}

__global__ void __launch_bounds__(256) l2a_rnn_mfma_k(const L2ALstmParams p) {
    extern __shared__ __attribute__((aligned(16))) char l2a_smem[];
    const int in_dim = p.in_dim, obs_dim = p.obs_dim, act_dim = p.act_dim, L = p.n_layers, SW = p.units;
    const int SX = l2a_rnn_row(in_dim);
    float* xs = reinterpret_cast<float*>(l2a_smem);     // [16][SX]
    float* lbase = xs + L2A_LVT * SX;                   // per layer: h [2][16][SP], scratch / c [16][SP]
    long long lfl = 0;
    for (int l = 0; l < L; ++l) lfl += 3LL * L2A_LVT * l2a_rnn_row(p.layer_units[l]);
    float* ss = lbase + lfl;                            // [16][obs_dim] state
    float* ds = ss + L2A_LVT * obs_dim;                 // [16][obs_dim] delta
    float* rs_ = ds + L2A_LVT * obs_dim;                // [16] returns
    float* araw = rs_ + L2A_LVT;                        // [16][act_dim] this step's raw actions
    float* c_in_mu = araw + L2A_LVT * act_dim;          // constants, read once from the model block
    float* c_in_iv = c_in_mu + in_dim;
    float* c_out_mu = c_in_iv + in_dim;
    float* c_out_sd = c_out_mu + obs_dim;
    float* c_bo = c_out_sd + obs_dim;
    float* c_bias = c_bo + obs_dim;                     // per layer: the kernels' biases as they lie in the block
    const int tid = threadIdx.x;
    const int j = tid & 15, s = tid >> 4;               // element-wise role: candidate j, slice s (as l2a_rnn_valu_k)
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int jc = lane & 15, qq = lane >> 4;           // matrix role: candidate jc (B, D), column jc (A), quarter qq
    const int env = blockIdx.x / p.tiles_per_env;
    const int tb = blockIdx.x - env * p.tiles_per_env;
    const int cand = tb * L2A_LVT + j;
    const bool valid = cand < p.n;
    const int row = env * p.n + (valid ? cand : p.n - 1);
    const int R = p.m * p.n;
    const bool lstm = (p.cell_type == L2A_CELL_LSTM), gru = (p.cell_type == L2A_CELL_GRU);
    {
        const float* in_mu = p.wblk + p.nm_off;
        const float* in_iv = in_mu + 16 * p.KG0;
        const float* out_mu = in_iv + 16 * p.KG0;
        const float* out_sd = out_mu + 16 * p.OT;
        for (int k = tid; k < in_dim; k += 256) { c_in_mu[k] = in_mu[k]; c_in_iv[k] = in_iv[k]; }
        for (int d = tid; d < obs_dim; d += 256) { c_out_mu[d] = out_mu[d]; c_out_sd[d] = out_sd[d]; c_bo[d] = p.wblk[p.raw_bo + d]; }
        float* cb = c_bias;
        for (int l = 0; l < L; ++l) {
            const int U = p.layer_units[l], n0 = (lstm ? 4 : (gru ? 2 : 1)) * U;
            for (int i = tid; i < n0; i += 256) cb[i] = p.wblk[p.layer_b[l][0] + i];
            if (gru) for (int i = tid; i < U; i += 256) cb[n0 + i] = p.wblk[p.layer_b[l][1] + i];
            cb += n0 + (gru ? U : 0);
        }
    }

    for (long long i = tid; i < L2A_LVT * SX + lfl; i += 256) xs[i] = 0.0f;     // padding stays zero for the whole launch
    __syncthreads();
    const long long hrow = (p.hid_per_row ? (long long)row : (long long)env) * SW;
    {
        int off = 0;
        float* lp = lbase;
        for (int l = 0; l < L; ++l) {
            const int U = p.layer_units[l], SP = l2a_rnn_row(U);
            for (int u = s; u < U; u += 16) {
                lp[j * SP + u] = p.h0[hrow + off + u];
                lp[2 * L2A_LVT * SP + j * SP + u] = lstm ? p.c0[hrow + off + u] : 0.0f;
            }
            off += U;
            lp += 3 * L2A_LVT * SP;
        }
    }
    const float* orow = p.obs0 + (p.obs_per_row ? (long long)row : (long long)env) * obs_dim;
    for (int d = s; d < obs_dim; d += 16) ss[j * obs_dim + d] = orow[d];
    if (s == 0) rs_[j] = p.ret_in ? p.ret_in[row] : 0.0f;
    __syncthreads();

    // this thread's action elements (k = s, s + 16) of the NEXT step: requested a step ahead, so that the read from HBM
    // travels under the products instead of in front of them (wider action vectors fall back to reading in place)
    const bool apf = act_dim <= 32;
    float apre[2];
    auto fetch_actions = [&](int tt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k = s + 16 * i;
            apre[i] = (apf && k < act_dim && tt < p.h) ? p.actions[((long long)tt * R + row) * act_dim + k] : 0.0f;
        }
    };
    fetch_actions(0);
    double disc_pow = p.disc0;
    for (int t = 0; t < p.h; ++t) {
        const int cur = t & 1, nxt = cur ^ 1;
        const float* arow = p.actions + ((long long)t * R + row) * act_dim;
        for (int k = s; k < obs_dim; k += 16) xs[j * SX + k] = (ss[j * obs_dim + k] - c_in_mu[k]) * c_in_iv[k];
        if (apf) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int k = s + 16 * i;
                if (k < act_dim) {
                    araw[j * act_dim + k] = apre[i];
                    xs[j * SX + obs_dim + k] = (apre[i] - c_in_mu[obs_dim + k]) * c_in_iv[obs_dim + k];
                }
            }
            fetch_actions(t + 1);
        } else {
            for (int k = s; k < act_dim; k += 16) {
                const float v = arow[k];
                araw[j * act_dim + k] = v;
                xs[j * SX + obs_dim + k] = (v - c_in_mu[obs_dim + k]) * c_in_iv[obs_dim + k];
            }
        }
        __syncthreads();
        const float* xin = xs;          // the current layer's input rows [16][xstride] (zero padded to 16 k-groups)
        int xstride = SX, kin = in_dim;
        float* lp = lbase;
        const float* cb = c_bias;
        for (int l = 0; l < L; ++l) {
            const int U = p.layer_units[l], SP = l2a_rnn_row(U), UT = (U + 15) >> 4, KGx = (kin + 15) >> 4;
            const float* hc = lp + cur * L2A_LVT * SP;
            float* hn = lp + nxt * L2A_LVT * SP;
            float* cl = lp + 2 * L2A_LVT * SP;
            const int G0 = lstm ? 4 : (gru ? 2 : 1);
            const float* b0 = cb;
            const __amdgpu_buffer_rsrc_t W0 = l2a_rsrc(p.wblk + p.layer_pk[l][0], l2a_rnn_pack_floats(kin, U, G0, true) * 4);
            const float* bx = xin + jc * xstride + 4 * qq;
            const float* bh = hc + jc * SP + 4 * qq;
            // A product call always fills four accumulator tiles - the matrix pipe wants four independent chains, and every
            // call pays ~1.7k clocks until its first operands arrive: the four gates of ONE unit tile of an LSTM layer, the
            // 2 x 2 gate tiles of TWO unit tiles of a GRU layer, FOUR unit tiles where a product has a single gate (BasicRNN,
            // the GRU candidate) - as long as the layer is wide enough to give every wave such a block.  In the packed array
            // those tiles are consecutive ([unit tile][gate]), so a block of unit tiles is "unit tile b of a kernel with more
            // gates"; wave w owns blocks w, w + 4, ...  Tiles past the last unit tile read the zero tiles the packed array is
            // padded with (l2a_rnn_pack_floats) and are skipped when the results go back to LDS.
            if (lstm) {
                for (int T = wave; T < UT; T += 4) {
                    const int u0 = 16 * T + 4 * qq;             // this lane's four units of the D tile
                    const int at = jc * SP + u0;                 // ... and their place in a [16][SP] LDS array
                    f32x4 acc[4], bias[4];          // (the biases and c are requested before the product, not after it)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int ii = 0; ii < 4; ++ii) bias[q][ii] = b0[q * U + (u0 + ii < U ? u0 + ii : 0)];
                    }
                    f32x4 cv = *reinterpret_cast<const f32x4*>(cl + at);
                    l2a_rnn_gemm<4>(W0, T, KGx, bx, UT, bh, lane, acc);
                    f32x4 hnew;
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) {
                        const bool live = u0 + ii < U;
                        const float ig = l2a_fast_sigmoid(acc[0][ii] + bias[0][ii]);
                        const float jg = l2a_rnn_act(acc[1][ii] + bias[1][ii], p.cell_act);
                        const float fg = l2a_fast_sigmoid(acc[2][ii] + bias[2][ii] + 1.0f);
                        const float og = l2a_fast_sigmoid(acc[3][ii] + bias[3][ii]);
                        const float cn = live ? fg * cv[ii] + ig * jg : 0.0f;
                        cv[ii] = cn;
                        hnew[ii] = live ? og * l2a_rnn_act(cn, p.cell_act) : 0.0f;
                    }
                    *reinterpret_cast<f32x4*>(cl + at) = cv;
                    *reinterpret_cast<f32x4*>(hn + at) = hnew;
                }
            } else {
                // (blocks only as far as every wave still gets one: a narrow layer keeps one unit tile per call)
                // products with ONE gate (BasicRNN, GRU candidate): TB unit tiles per call; epi(T, acc) finishes unit tile T
                auto prod1 = [&](auto tb_tag, __amdgpu_buffer_rsrc_t W, const float* bsrc, auto&& epi) {
                    constexpr int TB = decltype(tb_tag)::value;
                    for (int b = wave; TB * b < UT; b += 4) {
                        f32x4 acc[TB];
#pragma unroll
                        for (int q = 0; q < TB; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
                        l2a_rnn_gemm<TB>(W, b, KGx, bx, UT, bsrc, lane, acc);
#pragma unroll
                        for (int tb = 0; tb < TB; ++tb)
                            if (TB * b + tb < UT) epi(TB * b + tb, acc[tb]);
                    }
                };
                auto run1 = [&](__amdgpu_buffer_rsrc_t W, const float* bsrc, auto&& epi) {
                    if (UT >= 16) prod1(std::integral_constant<int, 4>(), W, bsrc, epi);
                    else if (UT >= 8) prod1(std::integral_constant<int, 2>(), W, bsrc, epi);
                    else prod1(std::integral_constant<int, 1>(), W, bsrc, epi);
                };
                if (gru) {
                    // reset / update gates: TB unit tiles x 2 gates per call
                    auto prod2 = [&](auto tb_tag) {
                        constexpr int TB = decltype(tb_tag)::value;
                        for (int b = wave; TB * b < UT; b += 4) {
                            f32x4 acc[2 * TB];
#pragma unroll
                            for (int q = 0; q < 2 * TB; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
                            l2a_rnn_gemm<2 * TB>(W0, b, KGx, bx, UT, bh, lane, acc);
#pragma unroll
                            for (int tb = 0; tb < TB; ++tb) {
                                const int T = TB * b + tb;
                                if (T < UT) {
                                    const int u0 = 16 * T + 4 * qq, at = jc * SP + u0;
                                    const f32x4 hv = *reinterpret_cast<const f32x4*>(hc + at);
                                    f32x4 rh, ug;
#pragma unroll
                                    for (int ii = 0; ii < 4; ++ii) {
                                        const bool live = u0 + ii < U;
                                        const int ub = live ? u0 + ii : 0;
                                        rh[ii] = live ? l2a_fast_sigmoid(acc[2 * tb][ii] + b0[ub]) * hv[ii] : 0.0f;       // r * h
                                        ug[ii] = live ? l2a_fast_sigmoid(acc[2 * tb + 1][ii] + b0[U + ub]) : 0.0f;         // u, parked in the new-h slot
                                    }
                                    *reinterpret_cast<f32x4*>(cl + at) = rh;
                                    *reinterpret_cast<f32x4*>(hn + at) = ug;
                                }
                            }
                        }
                    };
                    if (UT >= 8) prod2(std::integral_constant<int, 2>());
                    else prod2(std::integral_constant<int, 1>());
                    __syncthreads();            // every unit's r * h before the candidate product
                    const float* b1 = cb + 2 * U;
                    const __amdgpu_buffer_rsrc_t W1 = l2a_rsrc(p.wblk + p.layer_pk[l][1], l2a_rnn_pack_floats(kin, U, 1, true) * 4);
                    run1(W1, cl + jc * SP + 4 * qq, [&](int T, const f32x4& z) {
                        const int u0 = 16 * T + 4 * qq, at = jc * SP + u0;
                        const f32x4 hv = *reinterpret_cast<const f32x4*>(hc + at);
                        f32x4 ug = *reinterpret_cast<const f32x4*>(hn + at);
#pragma unroll
                        for (int ii = 0; ii < 4; ++ii) {
                            const bool live = u0 + ii < U;
                            const float cnd = l2a_rnn_act(z[ii] + b1[live ? u0 + ii : 0], p.cell_act);
                            ug[ii] = live ? ug[ii] * hv[ii] + (1.0f - ug[ii]) * cnd : 0.0f;
                        }
                        *reinterpret_cast<f32x4*>(hn + at) = ug;
                    });
                } else {
                    run1(W0, bh, [&](int T, const f32x4& z) {
                        const int u0 = 16 * T + 4 * qq, at = jc * SP + u0;
                        f32x4 hnew;
#pragma unroll
                        for (int ii = 0; ii < 4; ++ii) {
                            const bool live = u0 + ii < U;
                            hnew[ii] = live ? l2a_rnn_act(z[ii] + b0[live ? u0 + ii : 0], p.cell_act) : 0.0f;
                        }
                        *reinterpret_cast<f32x4*>(hn + at) = hnew;
                    });
                }
            }
            __syncthreads();
            xin = hn;               // the layer above reads this layer's new h
            xstride = SP;
            kin = U;
            lp += 3 * L2A_LVT * SP;
            cb += G0 * U + (gru ? U : 0);
        }
        {   // output layer: wave c owns obs tile c; K = the top layer's new h
            const int OTn = (obs_dim + 15) >> 4, UTt = (kin + 15) >> 4;
            const __amdgpu_buffer_rsrc_t WO = l2a_rsrc(p.wblk + p.pk_wout, l2a_rnn_pack_floats(kin, obs_dim, 1, false) * 4);
            const float* bt = xin + jc * xstride + 4 * qq;
            for (int c = wave; c < OTn; c += 4) {
                f32x4 acc[1];
                acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
                l2a_rnn_gemm<1>(WO, c, UTt, bt, 0, bt, lane, acc);
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    const int d = 16 * c + 4 * qq + ii;
                    if (d < obs_dim) {
                        const float o = l2a_act1(acc[0][ii] + c_bo[d], p.output_act);
                        ds[jc * obs_dim + d] = o * c_out_sd[d] + c_out_mu[d];
                    }
                }
            }
        }
        __syncthreads();
        if (s == 0) {
            float asq = 0.0f;
            for (int k = 0; k < act_dim; ++k) asq = fmaf(araw[j * act_dim + k], araw[j * act_dim + k], asq);
            float r = p.rw.alive - p.rw.ctrl_coef * asq;
            if (p.rw.w_vel != 0.0f) r += p.rw.w_vel * ds[j * obs_dim + p.rw.vel_index] * p.rw.inv_dt;
            if (p.rw.dist_coef != 0.0f) {
                float sq = 0.0f;
                for (int d = p.rw.dist_index; d < p.rw.dist_index + 3 && d < obs_dim; ++d) {
                    const float nx = ss[j * obs_dim + d] + ds[j * obs_dim + d];
                    sq = fmaf(nx, nx, sq);
                }
                r -= p.rw.dist_coef * sqrtf(sq);
            }
            rs_[j] = fmaf((float)disc_pow, r, rs_[j]);
        }
        disc_pow *= p.discount;
        __syncthreads();
        for (int d = s; d < obs_dim; d += 16) ss[j * obs_dim + d] += ds[j * obs_dim + d];
        __syncthreads();
    }

    if (valid) {
        const long long orow_o = (long long)env * p.n + cand;
        const int fin = p.h & 1;
        int off = 0;
        const float* lp = lbase;
        for (int l = 0; l < L; ++l) {
            const int U = p.layer_units[l], SP = l2a_rnn_row(U);
            const float* hl = lp + fin * L2A_LVT * SP + j * SP;
            const float* cl = lp + 2 * L2A_LVT * SP + j * SP;
            if (p.h_out) for (int u = s; u < U; u += 16) p.h_out[orow_o * SW + off + u] = hl[u];
            if (p.c_out) for (int u = s; u < U; u += 16) p.c_out[orow_o * SW + off + u] = lstm ? cl[u] : 0.0f;
            off += U;
            lp += 3 * L2A_LVT * SP;
        }
        if (p.state_out) for (int d = s; d < obs_dim; d += 16) p.state_out[orow_o * obs_dim + d] = ss[j * obs_dim + d];
        if (s == 0 && p.returns_out) p.returns_out[orow_o] = rs_[j];
    }
    if (p.best_key && tid < 16) {
        unsigned long long key = valid ? l2a_key_pack(rs_[j], p.cand_offset + cand) : 0ull;
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) {
            const unsigned int hi = __shfl_xor((unsigned int)(key >> 32), off);
            const unsigned int lo = __shfl_xor((unsigned int)(key & 0xffffffffu), off);
            const unsigned long long other = ((unsigned long long)hi << 32) | lo;
            key = (other > key) ? other : key;
        }
        if (tid == 0) {
            if (key != 0ull) atomicMax(p.best_key + env, key);
            l2a_publish_result(p, p.m * p.tiles_per_env);       // blocking plans: the last tile fills the host-mapped mailbox
        }
    }
}
