// l2a_rnn_valu.h - generic VALU rollout of STACKED recurrent cells: the other configurations `create_rnn`
// (reference dynamics/core/utils.py:192-236) can build besides the single-layer LSTM of run_rebal.py -
// `cell_type` in {'lstm', 'gru', 'rnn'} and `len(hidden_sizes) > 1` (tf.nn.rnn_cell.MultiRNNCell).  Included by
// l2a_lstm_api.hip only.
//
// Cell arithmetic (tensorflow==1.13.1, tensorflow/python/ops/rnn_cell_impl.py; x = the layer's input, i.e. the
// normalised [obs | act] for layer 0 and the new h of the layer below otherwise):
//   LSTMCell      z = [x | h] K + b (K [in + U, 4U], gates i j f o);  c = sig(f + 1) c + sig(i) act(j);  h = sig(o) act(c)
//   GRUCell       [r | u] = sig([x | h] Kg + bg) (Kg [in + U, 2U]);  cand = act([x | r * h] Kc + bc) (Kc [in + U, U]);
//                 h = u h + (1 - u) cand
//   BasicRNNCell  h = act([x | h] K + b)                                       ('rnn'; see include/l2a.h)
// then `obs += denorm(act_out(h_top Wout + bout))`, reward, return, arg-max exactly as l2a_lstm_valu_k.
//
// Workgroup = 256 threads = 16 candidates x 16 unit slices; thread (j, s) owns units s, s + 16, ... of candidate j
// in every layer.  LDS per layer: h double-buffered [2][16][U] (a layer's new h needs ALL of its old h) and one
// [16][U] scratch: the cell state c (LSTM) or r * h (GRU; the update gate u waits in the new-h slot of its unit).
// State I/O: c / h rows are the layers' states concatenated, [rows, sum(U_l)].
#pragma once

#include "l2a_lstm.h"
#include "l2a_lstm_valu.h"

__global__ void __launch_bounds__(256) l2a_rnn_valu_k(const L2ALstmParams p) {
    extern __shared__ __attribute__((aligned(16))) char l2a_smem[];
    const int in_dim = p.in_dim, obs_dim = p.obs_dim, act_dim = p.act_dim, L = p.n_layers, SW = p.units;
    float* xs = reinterpret_cast<float*>(l2a_smem);     // [16][in_dim]
    float* hs = xs + L2A_LVT * in_dim;                  // per layer [2][16][U_l], layers back to back
    float* cs = hs + 2 * L2A_LVT * SW;                  // per layer [16][U_l]
    float* ss = cs + L2A_LVT * SW;                      // [16][obs_dim] state
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
    const float* in_mu = p.wblk + p.nm_off;
    const float* in_iv = in_mu + 16 * p.KG0;
    const float* out_mu = in_iv + 16 * p.KG0;
    const float* out_sd = out_mu + 16 * p.OT;
    const float* wo = p.wblk + p.raw_wo;
    const float* bo = p.wblk + p.raw_bo;
    const bool lstm = (p.cell_type == L2A_CELL_LSTM), gru = (p.cell_type == L2A_CELL_GRU);

    const long long hrow = (p.hid_per_row ? (long long)row : (long long)env) * SW;
    {
        int off = 0;
        for (int l = 0; l < L; ++l) {
            const int U = p.layer_units[l];
            for (int u = s; u < U; u += 16) {
                hs[2 * L2A_LVT * off + j * U + u] = p.h0[hrow + off + u];
                cs[L2A_LVT * off + j * U + u] = lstm ? p.c0[hrow + off + u] : 0.0f;
            }
            off += U;
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
            xs[j * in_dim + k] = (v - in_mu[k]) * in_iv[k];
        }
        __syncthreads();
        const float* xin = xs + j * in_dim;     // this candidate's input row of the current layer
        int kin = in_dim, off = 0;
        for (int l = 0; l < L; ++l) {
            const int U = p.layer_units[l];
            float* hl = hs + 2 * L2A_LVT * off;
            const float* hc = hl + cur * L2A_LVT * U + j * U;
            float* hn = hl + nxt * L2A_LVT * U + j * U;
            float* cl = cs + L2A_LVT * off + j * U;
            const float* w0 = p.wblk + p.layer_w[l][0];
            const float* b0 = p.wblk + p.layer_b[l][0];
            if (lstm) {
                for (int u = s; u < U; u += 16) {
                    float z[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                    for (int k = 0; k < kin; ++k) {
                        const float xv = xin[k];
                        const float* wr = w0 + (long long)k * 4 * U + u;
#pragma unroll
                        for (int q = 0; q < 4; ++q) z[q] = fmaf(xv, wr[q * U], z[q]);
                    }
                    for (int k = 0; k < U; ++k) {
                        const float hv = hc[k];
                        const float* wr = w0 + (long long)(kin + k) * 4 * U + u;
#pragma unroll
                        for (int q = 0; q < 4; ++q) z[q] = fmaf(hv, wr[q * U], z[q]);
                    }
                    const float ig = l2a_sigmoid(z[0] + b0[u]);
                    const float jg = l2a_act1(z[1] + b0[U + u], p.cell_act);
                    const float fg = l2a_sigmoid(z[2] + b0[2 * U + u] + 1.0f);
                    const float og = l2a_sigmoid(z[3] + b0[3 * U + u]);
                    const float cn = fg * cl[u] + ig * jg;
                    cl[u] = cn;
                    hn[u] = og * l2a_act1(cn, p.cell_act);
                }
            } else if (gru) {
                for (int u = s; u < U; u += 16) {          // pass 1: reset / update gates
                    float zr = 0.0f, zu = 0.0f;
                    for (int k = 0; k < kin; ++k) {
                        const float* wr = w0 + (long long)k * 2 * U + u;
                        zr = fmaf(xin[k], wr[0], zr);
                        zu = fmaf(xin[k], wr[U], zu);
                    }
                    for (int k = 0; k < U; ++k) {
                        const float* wr = w0 + (long long)(kin + k) * 2 * U + u;
                        zr = fmaf(hc[k], wr[0], zr);
                        zu = fmaf(hc[k], wr[U], zu);
                    }
                    cl[u] = l2a_sigmoid(zr + b0[u]) * hc[u];           // r * h
                    hn[u] = l2a_sigmoid(zu + b0[U + u]);               // u, parked in this unit's new-h slot
                }
                __syncthreads();
                const float* w1 = p.wblk + p.layer_w[l][1];
                const float* b1 = p.wblk + p.layer_b[l][1];
                for (int u = s; u < U; u += 16) {          // pass 2: candidate, new h
                    float zc = 0.0f;
                    for (int k = 0; k < kin; ++k) zc = fmaf(xin[k], w1[(long long)k * U + u], zc);
                    for (int k = 0; k < U; ++k) zc = fmaf(cl[k], w1[(long long)(kin + k) * U + u], zc);
                    const float cnd = l2a_act1(zc + b1[u], p.cell_act);
                    const float ug = hn[u];
                    hn[u] = ug * hc[u] + (1.0f - ug) * cnd;
                }
            } else {
                for (int u = s; u < U; u += 16) {
                    float z = 0.0f;
                    for (int k = 0; k < kin; ++k) z = fmaf(xin[k], w0[(long long)k * U + u], z);
                    for (int k = 0; k < U; ++k) z = fmaf(hc[k], w0[(long long)(kin + k) * U + u], z);
                    hn[u] = l2a_act1(z + b0[u], p.cell_act);
                }
            }
            __syncthreads();
            xin = hn;           // the layer above reads this layer's new h
            kin = U;
            off += U;
        }
        for (int d = s; d < obs_dim; d += 16) {
            float acc = 0.0f;
            for (int k = 0; k < kin; ++k) acc = fmaf(xin[k], wo[(long long)k * obs_dim + d], acc);
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

    if (valid) {
        const long long orow_o = (long long)env * p.n + cand;
        const int fin = p.h & 1;
        int off = 0;
        for (int l = 0; l < L; ++l) {
            const int U = p.layer_units[l];
            const float* hl = hs + 2 * L2A_LVT * off + fin * L2A_LVT * U + j * U;
            const float* cl = cs + L2A_LVT * off + j * U;
            if (p.h_out) for (int u = s; u < U; u += 16) p.h_out[orow_o * SW + off + u] = hl[u];
            if (p.c_out) for (int u = s; u < U; u += 16) p.c_out[orow_o * SW + off + u] = lstm ? cl[u] : 0.0f;
            off += U;
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
