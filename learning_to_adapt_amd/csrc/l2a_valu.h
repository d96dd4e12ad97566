// l2a_valu.h - non-template kernels (weight packing, generic VALU rollout); included by l2a_api.hip only.
#pragma once

#include "l2a_kernels.h"

// blockIdx.y = weight set: source / destination advance by w_stride / out_stride floats per set.
__global__ void l2a_pack_layer_k(const float* __restrict__ w, long long w_stride, int k_in, int n_out, int KG,
                                 long long total, float* __restrict__ out, long long out_stride) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    w += (long long)blockIdx.y * w_stride;
    out += (long long)blockIdx.y * out_stride;
    int k, u;
    l2a_pack_decode(idx, KG, &k, &u);
    out[idx] = (k < k_in && u < n_out) ? w[(long long)k * n_out + u] : 0.0f;
}


// ------------------------------------------------------------------------------------------
// generic VALU rollout kernel: 256 threads own 16 candidates; activations live in LDS as
// [feature][16 candidates], thread u computes output units u, u+256, ... for all 16
// candidates, reading each weight exactly once per (workgroup, step, set), coalesced.
// ------------------------------------------------------------------------------------------
#define L2A_VT 16

__device__ __forceinline__ void l2a_valu_dense(const float* __restrict__ W, const float* __restrict__ b,
                                               int k_in, int n_out, const float* hin, float* hout,
                                               int act, int tid) {
    for (int u = tid; u < n_out; u += 256) {
        float acc[L2A_VT];
#pragma unroll
        for (int c = 0; c < L2A_VT; ++c) acc[c] = 0.0f;
        for (int k = 0; k < k_in; ++k) {
            const float w = W[(long long)k * n_out + u];
            const float4* hv = reinterpret_cast<const float4*>(hin + k * L2A_VT);
            const float4 h0 = hv[0], h1 = hv[1], h2 = hv[2], h3 = hv[3];
            acc[0] = fmaf(w, h0.x, acc[0]);   acc[1] = fmaf(w, h0.y, acc[1]);
            acc[2] = fmaf(w, h0.z, acc[2]);   acc[3] = fmaf(w, h0.w, acc[3]);
            acc[4] = fmaf(w, h1.x, acc[4]);   acc[5] = fmaf(w, h1.y, acc[5]);
            acc[6] = fmaf(w, h1.z, acc[6]);   acc[7] = fmaf(w, h1.w, acc[7]);
            acc[8] = fmaf(w, h2.x, acc[8]);   acc[9] = fmaf(w, h2.y, acc[9]);
            acc[10] = fmaf(w, h2.z, acc[10]); acc[11] = fmaf(w, h2.w, acc[11]);
            acc[12] = fmaf(w, h3.x, acc[12]); acc[13] = fmaf(w, h3.y, acc[13]);
            acc[14] = fmaf(w, h3.z, acc[14]); acc[15] = fmaf(w, h3.w, acc[15]);
        }
        const float bias = b[u];
#pragma unroll
        for (int c = 0; c < L2A_VT; ++c) hout[u * L2A_VT + c] = l2a_act1(acc[c] + bias, act);
    }
}

__global__ void __launch_bounds__(256) l2a_rollout_valu_k(const L2AKParams p) {
    extern __shared__ __attribute__((aligned(16))) char l2a_smem[];
    float* xin = reinterpret_cast<float*>(l2a_smem);     // [in_dim][16]
    float* hA = xin + p.in_dim * L2A_VT;                 // [hmax][16]
    float* hB = hA + p.hmax * L2A_VT;                    // [hmax][16]
    float* dl = hB + p.hmax * L2A_VT;                    // [obs_dim][16]  one set's delta-hat
    float* st = dl + p.obs_dim * L2A_VT;                 // [obs_dim][16]
    float* ds = st + p.obs_dim * L2A_VT;                 // [obs_dim][16]  sum of deltas
    float* av = ds + p.obs_dim * L2A_VT;                 // [act_dim][16]
    float* rets = av + p.act_dim * L2A_VT;               // [16]

    const int tid = threadIdx.x;
    const int bid = l2a_logical_wg(blockIdx.x, gridDim.x);
    const int env = bid / p.tiles_per_env;
    const int tb = bid - env * p.tiles_per_env;
    const int R = p.m * p.n;
    const int obs_dim = p.obs_dim, act_dim = p.act_dim, in_dim = p.in_dim;
    const bool per_block = (p.mode == L2A_MODE_PER_BLOCK);
    const int e_loop = (p.mode == L2A_MODE_MEAN) ? p.n_sets : 1;
    const int KG0 = p.KG0, OT = p.OT;

    auto cand_of = [&](int c) { return tb * L2A_VT + c; };
    auto row_of = [&](int c) { const int j = cand_of(c); return env * p.n + (j < p.n ? j : p.n - 1); };

    for (int i = tid; i < obs_dim * L2A_VT; i += 256) {
        const int d = i / L2A_VT, c = i - d * L2A_VT;
        const long long orow = p.obs_per_row ? (long long)row_of(c) : (long long)env;
        st[i] = p.obs0[orow * obs_dim + d];
        ds[i] = 0.0f;
    }
    if (tid < L2A_VT) rets[tid] = p.ret_in ? p.ret_in[row_of(tid)] : 0.0f;
    __syncthreads();

    double disc_pow = p.disc0;
    for (int t = 0; t < p.h; ++t) {
        for (int i = tid; i < act_dim * L2A_VT; i += 256) {
            const int k = i / L2A_VT, c = i - k * L2A_VT;
            av[i] = p.actions[((long long)t * R + row_of(c)) * act_dim + k];
        }
        __syncthreads();
        for (int e = 0; e < e_loop; ++e) {
            const int ws = per_block ? env : e;
            const float* wb = p.wblk + (long long)ws * p.set_stride;
            const float* in_mu = wb + p.nm_off;
            const float* in_iv = in_mu + 16 * KG0;
            const float* out_mu = in_mu + 32 * KG0;
            const float* out_sd = out_mu + 16 * OT;
            for (int i = tid; i < in_dim * L2A_VT; i += 256) {
                const int k = i / L2A_VT, c = i - k * L2A_VT;
                const float v = (k < obs_dim) ? st[k * L2A_VT + c] : av[(k - obs_dim) * L2A_VT + c];
                xin[i] = (v - in_mu[k]) * in_iv[k];
            }
            __syncthreads();
            const float* hin = xin;
            float* hout = hA;
            int k_in = in_dim;
            for (int l = 0; l < p.n_hidden; ++l) {
                l2a_valu_dense(wb + p.raw_w[l], wb + p.raw_b[l], k_in, p.hidden[l], hin, hout,
                               p.hidden_act, tid);
                __syncthreads();
                k_in = p.hidden[l];
                hin = hout;
                hout = (hout == hA) ? hB : hA;
            }
            l2a_valu_dense(wb + p.raw_w[p.n_hidden], wb + p.raw_b[p.n_hidden], k_in, obs_dim, hin, dl,
                           p.output_act, tid);
            __syncthreads();
            for (int i = tid; i < obs_dim * L2A_VT; i += 256) {
                const int d = i / L2A_VT;
                ds[i] += dl[i] * out_sd[d] + out_mu[d];
            }
            __syncthreads();
        }
        const float disc_t = (float)disc_pow;
        disc_pow *= p.discount;
        if (tid < L2A_VT) {
            const int c = tid;
            float asq = 0.0f;
            for (int k = 0; k < act_dim; ++k) asq = fmaf(av[k * L2A_VT + c], av[k * L2A_VT + c], asq);
            float r = p.rw.alive - p.rw.ctrl_coef * asq;
            if (p.rw.w_vel != 0.0f) {
                float d = ds[p.rw.vel_index * L2A_VT + c];
                if (e_loop > 1) d = d / (float)e_loop;
                r += p.rw.w_vel * d * p.rw.inv_dt;
            }
            if (p.rw.dist_coef != 0.0f) {
                float sq = 0.0f;
                for (int d3 = 0; d3 < 3; ++d3) {
                    const int dim = p.rw.dist_index + d3;
                    if (dim < obs_dim) {
                        float d = ds[dim * L2A_VT + c];
                        if (e_loop > 1) d = d / (float)e_loop;
                        const float nx = st[dim * L2A_VT + c] + d;
                        sq = fmaf(nx, nx, sq);
                    }
                }
                r -= p.rw.dist_coef * sqrtf(sq);
            }
            rets[c] = fmaf(disc_t, r, rets[c]);
        }
        __syncthreads();
        for (int i = tid; i < obs_dim * L2A_VT; i += 256) {
            float d = ds[i];
            if (e_loop > 1) d = d / (float)e_loop;
            st[i] += d;
            ds[i] = 0.0f;
        }
        __syncthreads();
    }

    if (tid < 64) {
        unsigned long long key = 0ull;
        if (tid < L2A_VT && cand_of(tid) < p.n) {
            if (p.returns_out) p.returns_out[(long long)env * p.n + cand_of(tid)] = rets[tid];
            key = l2a_key_pack(rets[tid], p.cand_offset + cand_of(tid));
        }
        if (p.best_key) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const unsigned int hi = __shfl_xor((unsigned int)(key >> 32), off);
                const unsigned int lo = __shfl_xor((unsigned int)(key & 0xffffffffu), off);
                const unsigned long long other = ((unsigned long long)hi << 32) | lo;
                key = (other > key) ? other : key;
            }
            if (tid == 0 && key != 0ull) atomicMax(p.best_key + env, key);
        }
    }
    if (p.state_out) {
        for (int i = tid; i < obs_dim * L2A_VT; i += 256) {
            const int d = i / L2A_VT, c = i - d * L2A_VT;
            if (cand_of(c) < p.n) p.state_out[((long long)env * p.n + cand_of(c)) * obs_dim + d] = st[i];
        }
    }
}
