// l2a_rnn_mfma.h - matrix-core rollout of the STACKED recurrent cells of l2a_rnn_valu.h (GRU, BasicRNN, LSTM stacks:
// everything `create_rnn`, reference dynamics/core/utils.py:192-236, builds besides run_rebal.py's single LSTM layer, which
// has its own tuned kernel in l2a_lstm.h).  Included by l2a_lstm_api.hip only.
//
// Same semantics, same LDS state and the same thread roles for the element-wise parts as l2a_rnn_valu_k (cell arithmetic:
// see that header); what changes is who multiplies.  Every [x | h] K product runs on v_mfma_f32_16x16x4_f32:
//
//   workgroup = 4 waves = one tile of 16 candidates; wave w owns the unit tiles T = w, w + 4, ... of every layer
//   (16 units each, all gates of a unit in one wave, so the gate arithmetic is register local);
//   B operand  = the layer's input / old h as they lie in LDS, row-major [16 candidates][K]: MFMA ii of k-group g multiplies
//                k = 16 g + 4 (lane >> 4) + ii, i.e. a lane's four MFMAs of a k-group read ONE ds_read_b128 (rows padded
//                to a multiple of 16 with zeros + 4 floats of skew against bank conflicts);
//   A operand  = the reference's TF-layout kernel [K, G U] read in place (dword buffer loads, 16 consecutive columns per
//                quarter wave; rows past the matrix return 0, rows of an input's zero padding meet a zero B) - no packed
//                copy: these models are small and not the run scripts' default;
//   D          = lane (candidate lane & 15, units 16 T + 4 (lane >> 4) + ii): bias, gates, new h / c written back to the
//                same row-major LDS rows (padding units masked to zero, so they never feed a later product).
//
// The output layer is one more such product (wave c owns obs tile c); normalisation, reward, state update, returns and
// arg-max are the code of l2a_rnn_valu_k.  fp32 MFMA sums in another order than the VALU loop: results agree with
// l2a_rnn_valu_k to rounding (tests/test_rnn.py compares them), not bit for bit.
#pragma once

#include "l2a_lstm.h"
#include "l2a_lstm_valu.h"

#define L2A_RNN_SKEW 4      // floats added to every LDS row (rows of 16 k floats would all start in bank 0)

__host__ __device__ inline int l2a_rnn_row(int k) { return 16 * ((k + 15) / 16) + L2A_RNN_SKEW; }

// LDS floats of the kernel for a model (host: launch; device: carve)
__host__ __device__ inline long long l2a_rnn_mfma_lds_floats(int in_dim, int obs_dim, int n_layers, const int* units) {
    long long f = (long long)L2A_LVT * l2a_rnn_row(in_dim);
    for (int l = 0; l < n_layers; ++l) f += 3LL * L2A_LVT * l2a_rnn_row(units[l]);
    return f + 2LL * L2A_LVT * obs_dim + L2A_LVT;
}

// acc[q] += W[row0 + k][q * gate_stride + col] * B[k]   for k in [0, 16 kgroups)   (one 16 x 16 output tile per gate)
// W: raw-buffer resource over the whole [rows, ncols] matrix; `colb` = byte offset of this lane's column of gate 0;
// `bl` = this lane's B row + 4 (lane >> 4).
template <int G>
__device__ __forceinline__ void l2a_rnn_gemm(__amdgpu_buffer_rsrc_t W, int ncols, int gate_stride, int colb, int row0, int kgroups,
                                             const float* bl, int qq, f32x4 (&acc)[G]) {
    const int rowb = (row0 + 4 * qq) * ncols * 4 + colb;
#pragma unroll 2
    for (int g = 0; g < kgroups; ++g) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(bl + 16 * g);
        float a[G][4];
#pragma unroll
        for (int q = 0; q < G; ++q)
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
                a[q][ii] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    W, rowb + (16 * g + ii) * ncols * 4 + q * gate_stride * 4, 0, 0));
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
            for (int q = 0; q < G; ++q) acc[q] = L2A_MFMA(a[q][ii], b[ii], acc[q]);
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
    const float* in_mu = p.wblk + p.nm_off;
    const float* in_iv = in_mu + 16 * p.KG0;
    const float* out_mu = in_iv + 16 * p.KG0;
    const float* out_sd = out_mu + 16 * p.OT;
    const float* bo = p.wblk + p.raw_bo;
    const bool lstm = (p.cell_type == L2A_CELL_LSTM), gru = (p.cell_type == L2A_CELL_GRU);

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

    double disc_pow = p.disc0;
    for (int t = 0; t < p.h; ++t) {
        const int cur = t & 1, nxt = cur ^ 1;
        const float* arow = p.actions + ((long long)t * R + row) * act_dim;
        for (int k = s; k < in_dim; k += 16) {
            const float v = (k < obs_dim) ? ss[j * obs_dim + k] : arow[k - obs_dim];
            xs[j * SX + k] = (v - in_mu[k]) * in_iv[k];
        }
        __syncthreads();
        const float* xin = xs;          // the current layer's input rows [16][xstride] (zero padded to 16 k-groups)
        int xstride = SX, kin = in_dim;
        float* lp = lbase;
        for (int l = 0; l < L; ++l) {
            const int U = p.layer_units[l], SP = l2a_rnn_row(U), UT = (U + 15) >> 4, KGx = (kin + 15) >> 4;
            const float* hc = lp + cur * L2A_LVT * SP;
            float* hn = lp + nxt * L2A_LVT * SP;
            float* cl = lp + 2 * L2A_LVT * SP;
            const int G0 = lstm ? 4 : (gru ? 2 : 1);
            const float* w0 = p.wblk + p.layer_w[l][0];
            const float* b0 = p.wblk + p.layer_b[l][0];
            const __amdgpu_buffer_rsrc_t W0 = l2a_rsrc(w0, (long long)(kin + U) * G0 * U * 4);
            const float* bx = xin + jc * xstride + 4 * qq;
            const float* bh = hc + jc * SP + 4 * qq;
            for (int T = wave; T < UT; T += 4) {
                const int u0 = 16 * T + 4 * qq;                 // this lane's four units of the D tile
                const int colb = (16 * T + jc) * 4;              // this lane's column of the A tile (gate 0)
                const int at = jc * SP + u0;                     // ... and their place in a [16][SP] LDS array
                f32x4 bias[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) bias[q][ii] = (q < G0 && u0 + ii < U) ? b0[q * U + u0 + ii] : 0.0f;
                f32x4 hnew;
                if (lstm) {
                    f32x4 acc[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    l2a_rnn_gemm<4>(W0, 4 * U, U, colb, 0, KGx, bx, qq, acc);
                    l2a_rnn_gemm<4>(W0, 4 * U, U, colb, kin, UT, bh, qq, acc);
                    f32x4 cv = *reinterpret_cast<const f32x4*>(cl + at);
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) {
                        const float ig = l2a_sigmoid(acc[0][ii] + bias[0][ii]);
                        const float jg = l2a_act1(acc[1][ii] + bias[1][ii], p.cell_act);
                        const float fg = l2a_sigmoid(acc[2][ii] + bias[2][ii] + 1.0f);
                        const float og = l2a_sigmoid(acc[3][ii] + bias[3][ii]);
                        const float cn = (u0 + ii < U) ? fg * cv[ii] + ig * jg : 0.0f;
                        cv[ii] = cn;
                        hnew[ii] = (u0 + ii < U) ? og * l2a_act1(cn, p.cell_act) : 0.0f;
                    }
                    *reinterpret_cast<f32x4*>(cl + at) = cv;
                } else if (gru) {
                    f32x4 acc[2];
                    acc[0] = acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    l2a_rnn_gemm<2>(W0, 2 * U, U, colb, 0, KGx, bx, qq, acc);
                    l2a_rnn_gemm<2>(W0, 2 * U, U, colb, kin, UT, bh, qq, acc);
                    const f32x4 hv = *reinterpret_cast<const f32x4*>(hc + at);
                    f32x4 rh;
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) {
                        const bool live = u0 + ii < U;
                        rh[ii] = live ? l2a_sigmoid(acc[0][ii] + bias[0][ii]) * hv[ii] : 0.0f;       // r * h
                        hnew[ii] = live ? l2a_sigmoid(acc[1][ii] + bias[1][ii]) : 0.0f;              // u, parked in the new-h slot
                    }
                    *reinterpret_cast<f32x4*>(cl + at) = rh;
                } else {
                    f32x4 acc[1];
                    acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    l2a_rnn_gemm<1>(W0, U, U, colb, 0, KGx, bx, qq, acc);
                    l2a_rnn_gemm<1>(W0, U, U, colb, kin, UT, bh, qq, acc);
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) hnew[ii] = (u0 + ii < U) ? l2a_act1(acc[0][ii] + bias[0][ii], p.cell_act) : 0.0f;
                }
                *reinterpret_cast<f32x4*>(hn + at) = hnew;
            }
            if (gru) {
                __syncthreads();            // every unit's r * h before the candidate product
                const float* w1 = p.wblk + p.layer_w[l][1];
                const float* b1 = p.wblk + p.layer_b[l][1];
                const __amdgpu_buffer_rsrc_t W1 = l2a_rsrc(w1, (long long)(kin + U) * U * 4);
                const float* br = cl + jc * SP + 4 * qq;
                for (int T = wave; T < UT; T += 4) {
                    const int u0 = 16 * T + 4 * qq, colb = (16 * T + jc) * 4, at = jc * SP + u0;
                    f32x4 acc[1];
                    acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    l2a_rnn_gemm<1>(W1, U, U, colb, 0, KGx, bx, qq, acc);
                    l2a_rnn_gemm<1>(W1, U, U, colb, kin, UT, br, qq, acc);
                    const f32x4 hv = *reinterpret_cast<const f32x4*>(hc + at);
                    f32x4 ug = *reinterpret_cast<const f32x4*>(hn + at);
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) {
                        const float cnd = l2a_act1(acc[0][ii] + ((u0 + ii < U) ? b1[u0 + ii] : 0.0f), p.cell_act);
                        ug[ii] = (u0 + ii < U) ? ug[ii] * hv[ii] + (1.0f - ug[ii]) * cnd : 0.0f;
                    }
                    *reinterpret_cast<f32x4*>(hn + at) = ug;
                }
            }
            __syncthreads();
            xin = hn;               // the layer above reads this layer's new h
            xstride = SP;
            kin = U;
            lp += 3 * L2A_LVT * SP;
        }
        {   // output layer: wave c owns obs tile c; K = the top layer's new h
            const int OTn = (obs_dim + 15) >> 4, UTt = (kin + 15) >> 4;
            const __amdgpu_buffer_rsrc_t WO = l2a_rsrc(p.wblk + p.raw_wo, (long long)kin * obs_dim * 4);
            const float* bt = xin + jc * xstride + 4 * qq;
            for (int c = wave; c < OTn; c += 4) {
                f32x4 acc[1];
                acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
                // columns past obs_dim read the next row's weights: their D rows are never stored
                l2a_rnn_gemm<1>(WO, obs_dim, 0, (16 * c + jc) * 4, 0, UTt, bt, qq, acc);
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    const int d = 16 * c + 4 * qq + ii;
                    if (d < obs_dim) {
                        const float o = l2a_act1(acc[0][ii] + bo[d], p.output_act);
                        ds[jc * obs_dim + d] = o * out_sd[d] + out_mu[d];
                    }
                }
            }
        }
        __syncthreads();
        if (s == 0) {
            float asq = 0.0f;
            for (int k = 0; k < act_dim; ++k) asq = fmaf(arow[k], arow[k], asq);
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
        if (tid == 0 && key != 0ull) atomicMax(p.best_key + env, key);
    }
}
