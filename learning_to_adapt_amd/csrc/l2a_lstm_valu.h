// l2a_lstm_valu.h - the non-template kernels of the recurrent planner: weight re-packing for the
// MFMA kernel and the generic VALU rollout.  Included by l2a_lstm_api.hip ONLY (one definition per
// library).
#pragma once

#include "l2a_lstm.h"

__global__ void l2a_lstm_pack_k(const float* __restrict__ wk, int KG0, int UT, int in_dim, long long total,
                                float* __restrict__ out) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int k, col;
    l2a_lstm_pack_decode(idx, KG0, UT, in_dim, &k, &col);
    out[idx] = (k >= 0) ? wk[(long long)k * (64 * UT) + col] : 0.0f;
}

// Output layer [U, obs_dim] -> MFMA fragment order (same index function as l2a_pack_layer_k, l2a_valu.h).
__global__ void l2a_lstm_pack_out_k(const float* __restrict__ w, int k_in, int n_out, int KG, long long total,
                                    float* __restrict__ out) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int k, u;
    l2a_pack_decode(idx, KG, &k, &u);
    out[idx] = (k < k_in && u < n_out) ? w[(long long)k * n_out + u] : 0.0f;
}

// ------------------------------------------------------------------------------------------
// Generic VALU kernel: workgroup = 256 threads = 16 candidates x 16 unit slices; thread (j, s)
// computes the gates of units s, s + 16, ... of candidate j.  x, h, c and the deltas live in LDS.
// Reads the raw TF-layout weights (no packing).  Baseline + fallback for any U.
// ------------------------------------------------------------------------------------------
#define L2A_LVT 16
__global__ void __launch_bounds__(256) l2a_lstm_valu_k(const L2ALstmParams p) {
    extern __shared__ __attribute__((aligned(16))) char l2a_smem[];
    const int U = p.units, in_dim = p.in_dim, obs_dim = p.obs_dim, act_dim = p.act_dim;
    float* xs = reinterpret_cast<float*>(l2a_smem);     // [16][in_dim]
    float* hs = xs + L2A_LVT * in_dim;                  // [2][16][U]
    float* cs = hs + 2 * L2A_LVT * U;                   // [16][U]
    float* ss = cs + L2A_LVT * U;                       // [16][obs_dim] state
    float* ds = ss + L2A_LVT * obs_dim;                 // [16][obs_dim] delta
    float* rs_ = ds + L2A_LVT * obs_dim;                // [16] returns
    const int tid = threadIdx.x;
    const int j = tid & 15, s = tid >> 4;
    const int env = blockIdx.x / p.tiles_per_env;
    const int tb = blockIdx.x - env * p.tiles_per_env;
    const int cand = tb * L2A_LVT + j;
    const bool valid = cand < p.n;
    const int row = env * p.n + (valid ? cand : p.n - 1);
    const int R = p.m * p.n;
    const float* wk = p.wblk + p.raw_wk;
    const float* bk = p.wblk + p.raw_bk;
    const float* wo = p.wblk + p.raw_wo;
    const float* bo = p.wblk + p.raw_bo;
    const float* in_mu = p.wblk + p.nm_off;
    const float* in_iv = in_mu + 16 * p.KG0;
    const float* out_mu = in_iv + 16 * p.KG0;
    const float* out_sd = out_mu + 16 * p.OT;

    const long long hrow = (p.hid_per_row ? (long long)row : (long long)env) * U;
    for (int u = s; u < U; u += 16) {
        hs[j * U + u] = p.h0[hrow + u];
        cs[j * U + u] = p.c0[hrow + u];
    }
    const float* orow = p.obs0 + (p.obs_per_row ? (long long)row : (long long)env) * obs_dim;
    for (int d = s; d < obs_dim; d += 16) ss[j * obs_dim + d] = orow[d];
    if (s == 0) rs_[j] = p.ret_in ? p.ret_in[row] : 0.0f;
    __syncthreads();

    double disc_pow = p.disc0;
    for (int t = 0; t < p.h; ++t) {
        float* hc = hs + (t & 1) * L2A_LVT * U;
        float* hn = hs + ((t + 1) & 1) * L2A_LVT * U;
        const float* arow = p.actions + ((long long)t * R + row) * act_dim;
        for (int k = s; k < in_dim; k += 16) {
            const float v = (k < obs_dim) ? ss[j * obs_dim + k] : arow[k - obs_dim];
            xs[j * in_dim + k] = (v - in_mu[k]) * in_iv[k];
        }
        __syncthreads();
        for (int u = s; u < U; u += 16) {
            float z[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) z[q] = 0.0f;
            for (int k = 0; k < in_dim; ++k) {
                const float xv = xs[j * in_dim + k];
                const float* wr = wk + (long long)k * 4 * U + u;
#pragma unroll
                for (int q = 0; q < 4; ++q) z[q] = fmaf(xv, wr[q * U], z[q]);
            }
            for (int k = 0; k < U; ++k) {
                const float hv = hc[j * U + k];
                const float* wr = wk + (long long)(in_dim + k) * 4 * U + u;
#pragma unroll
                for (int q = 0; q < 4; ++q) z[q] = fmaf(hv, wr[q * U], z[q]);
            }
            const float ig = l2a_sigmoid(z[0] + bk[u]);
            const float jg = l2a_act1(z[1] + bk[U + u], p.cell_act);
            const float fg = l2a_sigmoid(z[2] + bk[2 * U + u] + 1.0f);
            const float og = l2a_sigmoid(z[3] + bk[3 * U + u]);
            const float cn = fg * cs[j * U + u] + ig * jg;
            cs[j * U + u] = cn;
            hn[j * U + u] = og * l2a_act1(cn, p.cell_act);
        }
        __syncthreads();
        for (int d = s; d < obs_dim; d += 16) {
            float acc = 0.0f;
            for (int k = 0; k < U; ++k) acc = fmaf(hn[j * U + k], wo[(long long)k * obs_dim + d], acc);
            acc = l2a_act1(acc + bo[d], p.output_act);
            ds[j * obs_dim + d] = acc * out_sd[d] + out_mu[d];
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

    const float* hl = hs + (p.h & 1) * L2A_LVT * U;
    if (valid) {
        const long long orow_o = (long long)env * p.n + cand;
        if (p.c_out) for (int u = s; u < U; u += 16) p.c_out[orow_o * U + u] = cs[j * U + u];
        if (p.h_out) for (int u = s; u < U; u += 16) p.h_out[orow_o * U + u] = hl[j * U + u];
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

// ------------------------------------------------------------------------------------------
// The controller's own state step (policies/rnn_mpc_controller.py:63: `_, self._hidden_state = dynamics_model.predict(obs,
// chosen actions, hidden)`) for the few rows a controller has (one per env): c', h' only - the predicted observation is thrown
// away by the caller - and the kernel's 4 U columns cut over U / 16 workgroups, each streaming its 64 columns of all K = in + U
// rows (72 KB at U = 256).  Until round 5 this was a one-tile launch of the 16-candidate rollout kernel (ONE workgroup streaming
// the whole 1.1 MB gate matrix and the output layer: 24 us) behind a gather launch (5 us) - 29 us of a 210 us ReBAL step.
// The first actions come either as [m, act_dim] rows (`act`) or are gathered here from the plan's candidate tensor through the
// arg-max keys (`best_key`; l2a_gather_best_k's clamped index).  Thread (column c = tid & 63 -> gate c >> 4, unit u0 + (c & 15);
// k quarter tid >> 6); rows in chunks of eight; the quarters' partial sums meet in LDS in the order 0 .. 3.
// ------------------------------------------------------------------------------------------
struct L2ALstmAdvParams {
    const float* wblk;
    long long raw_wk, raw_bk, nm_off;
    int obs_dim, act_dim, in_dim, units, KG0, cell_act;
    const float* obs;                       // [m, obs_dim] (device or host-mapped)
    const float* act;                       // [m, act_dim] or null
    const unsigned long long* best_key;     // gather form: keys [m], candidate tensor `actions` (step 0: [m * n, act_dim])
    const float* actions;
    int n, cand_offset;
    const float* c0;
    const float* h0;
    float* c1;
    float* h1;
    int m;
};

#define L2A_ADV_ROWS 8
__global__ void __launch_bounds__(256) l2a_lstm_advance_k(const L2ALstmAdvParams p) {
    extern __shared__ __attribute__((aligned(16))) char l2a_smem[];
    const int U = p.units, K = p.in_dim + U;
    float* xs = reinterpret_cast<float*>(l2a_smem);             // [K][8]
    float* ps = xs + K * L2A_ADV_ROWS;                          // [4][64][8]
    const int tid = threadIdx.x;
    const int col = tid & 63, kp = tid >> 6;
    const int g = col >> 4, j = col & 15;
    const int u0 = blockIdx.x * 16;
    const float* wk = p.wblk + p.raw_wk + (long long)g * U + u0 + j;
    const float* bk = p.wblk + p.raw_bk;
    const float* in_mu = p.wblk + p.nm_off;
    const float* in_iv = in_mu + 16 * p.KG0;
    const int Kq = (K + 3) / 4;
    const int k0 = kp * Kq, k1 = (k0 + Kq < K) ? k0 + Kq : K;
    for (int r0 = 0; r0 < p.m; r0 += L2A_ADV_ROWS) {
        for (int i = tid; i < K * L2A_ADV_ROWS; i += 256) {
            const int k = i >> 3, r = i & 7;
            const int row = r0 + r;
            float v = 0.0f;
            if (row < p.m) {
                if (k < p.obs_dim) {
                    v = (p.obs[(long long)row * p.obs_dim + k] - in_mu[k]) * in_iv[k];
                } else if (k < p.in_dim) {
                    float a;
                    if (p.act) {
                        a = p.act[(long long)row * p.act_dim + (k - p.obs_dim)];
                    } else {
                        const unsigned int low = (unsigned int)(p.best_key[row] & 0x7fffffffull);
                        int idx = (int)(0x7fffffffu - low) - p.cand_offset;
                        idx = idx < 0 ? 0 : (idx >= p.n ? p.n - 1 : idx);
                        a = p.actions[((long long)row * p.n + idx) * p.act_dim + (k - p.obs_dim)];
                    }
                    v = (a - in_mu[k]) * in_iv[k];
                } else {
                    v = p.h0[(long long)row * U + (k - p.in_dim)];
                }
            }
            xs[i] = v;
        }
        __syncthreads();
        float acc[L2A_ADV_ROWS];
#pragma unroll
        for (int r = 0; r < L2A_ADV_ROWS; ++r) acc[r] = 0.0f;
        for (int k = k0; k < k1; k += 8) {
            float w[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) w[q] = (k + q < k1) ? wk[(long long)(k + q) * 4 * U] : 0.0f;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (k + q < k1) {
                    const f32x4 xa = *reinterpret_cast<const f32x4*>(xs + (k + q) * L2A_ADV_ROWS);
                    const f32x4 xb = *reinterpret_cast<const f32x4*>(xs + (k + q) * L2A_ADV_ROWS + 4);
                    acc[0] = fmaf(xa[0], w[q], acc[0]); acc[1] = fmaf(xa[1], w[q], acc[1]);
                    acc[2] = fmaf(xa[2], w[q], acc[2]); acc[3] = fmaf(xa[3], w[q], acc[3]);
                    acc[4] = fmaf(xb[0], w[q], acc[4]); acc[5] = fmaf(xb[1], w[q], acc[5]);
                    acc[6] = fmaf(xb[2], w[q], acc[6]); acc[7] = fmaf(xb[3], w[q], acc[7]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < L2A_ADV_ROWS; ++r) ps[(kp * 64 + col) * L2A_ADV_ROWS + r] = acc[r];
        __syncthreads();
        if (tid < 16 * L2A_ADV_ROWS) {
            const int r = tid >> 4, jj = tid & 15;
            const int row = r0 + r, u = u0 + jj;
            if (row < p.m) {
                float z[4];
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {
                    float s = ps[(0 * 64 + gg * 16 + jj) * L2A_ADV_ROWS + r];
#pragma unroll
                    for (int q = 1; q < 4; ++q) s += ps[(q * 64 + gg * 16 + jj) * L2A_ADV_ROWS + r];
                    z[gg] = s;
                }
                // tensorflow==1.13.1 LSTMCell, gate order i, j, f, o, forget bias 1 (l2a_lstm_valu_k above; dynamics/core/utils.py:192-213)
                const float ig = l2a_sigmoid(z[0] + bk[u]);
                const float jg = l2a_act1(z[1] + bk[U + u], p.cell_act);
                const float fg = l2a_sigmoid(z[2] + bk[2 * U + u] + 1.0f);
                const float og = l2a_sigmoid(z[3] + bk[3 * U + u]);
                const float cn = fg * p.c0[(long long)row * U + u] + ig * jg;
                p.c1[(long long)row * U + u] = cn;
                p.h1[(long long)row * U + u] = og * l2a_act1(cn, p.cell_act);
            }
        }
        __syncthreads();
    }
}
