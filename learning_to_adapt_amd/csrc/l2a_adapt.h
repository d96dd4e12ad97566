// l2a_adapt.h - GrBAL's inner adaptation step on the GPU (included by l2a_api.hip only).
//
// Reference: `MetaMLPDynamicsModel.adapt` (dynamics/meta_mlp_dynamics.py:321-345) runs, for every task i,
// one SGD step theta_i = theta - lr * grad MSE(f_theta(x_i), y_i) (`_adapt_sym`, :409-421; loss :118) on
// `adapt_batch_size` (16, run_scripts/run_grbal.py) normalised transitions, pulls the m adapted parameter
// sets to the host and feeds them back on every later `sess.run`.  Here the step is two kernels that read the
// base parameters once and write the adapted sets straight into the per-block model the planner launches on -
// raw layout AND MFMA fragment order - so nothing is re-uploaded or re-packed (2 L + 1 + L launches, L layers):
//
//   l2a_adapt_prep_k / fwd_k / bwd_k : the forward and backward pass, one small launch per layer, every launch
//                        spread over (64-unit slice, task) workgroups - the first version ran one workgroup
//                        per task through all layers and was latency bound at ~1 ms, no better than the 45 stock
//                        PyTorch launches it replaced.  dZ_L = 2 (y_hat - y) / (rows * obs_dim),
//                        dZ_l = (W_l dZ_{l+1}) * act'(A_l); layer inputs A_l and the dZ_l ([dim][16 rows]) live
//                        in a scratch buffer
//   l2a_adapt_update_k : per layer, one thread per weight: g = sum_r A_l[k][r] dZ_{l+1}[u][r],
//                        theta' = theta - lr g -> raw kernel, packed kernel; biases alike
//
// Rows beyond `rows` (padding up to 16) carry dZ = 0 and therefore no gradient.
#pragma once

#include "l2a_kernels.h"
#include "l2a_valu.h"

#define L2A_AR 16       // rows per task held in registers / LDS (adapt_batch_size <= 16)

struct L2AAdaptParams {
    const float* w[L2A_MAX_LAYERS];     // base parameters (device), reference layout
    const float* b[L2A_MAX_LAYERS];
    int dims[L2A_MAX_LAYERS + 1];       // in_dim, hidden..., obs_dim
    int n_layers;                       // n_hidden + 1
    int hidden_act;
    int rows;
    const float* x;                     // [m, rows, in_dim]  normalised [obs | act]
    const float* y;                     // [m, rows, obs_dim] normalised deltas
    float* scratch;                     // per task: A_0 .. A_{L-1}, dZ_1 .. dZ_L, each [dim][16]
    long long scratch_stride;           // floats per task
    long long a_off[L2A_MAX_LAYERS];    // offset of A_l in a task's scratch
    long long z_off[L2A_MAX_LAYERS + 1];// offset of dZ_l (l = 1 .. L)
    int hmax;
};

__device__ __forceinline__ float l2a_act_grad_from_output(float o, int kind) {
    switch (kind) {
        case L2A_ACT_RELU: return o > 0.0f ? 1.0f : 0.0f;
        case L2A_ACT_TANH: return 1.0f - o * o;
        case L2A_ACT_SIGMOID: return o * (1.0f - o);
        default: return 1.0f;
    }
}

#define L2A_ACC16(acc, w, v0, v1, v2, v3)                                                   \
    acc[0] = fmaf(w, v0.x, acc[0]);   acc[1] = fmaf(w, v0.y, acc[1]);                        \
    acc[2] = fmaf(w, v0.z, acc[2]);   acc[3] = fmaf(w, v0.w, acc[3]);                        \
    acc[4] = fmaf(w, v1.x, acc[4]);   acc[5] = fmaf(w, v1.y, acc[5]);                        \
    acc[6] = fmaf(w, v1.z, acc[6]);   acc[7] = fmaf(w, v1.w, acc[7]);                        \
    acc[8] = fmaf(w, v2.x, acc[8]);   acc[9] = fmaf(w, v2.y, acc[9]);                        \
    acc[10] = fmaf(w, v2.z, acc[10]); acc[11] = fmaf(w, v2.w, acc[11]);                      \
    acc[12] = fmaf(w, v3.x, acc[12]); acc[13] = fmaf(w, v3.y, acc[13]);                      \
    acc[14] = fmaf(w, v3.z, acc[14]); acc[15] = fmaf(w, v3.w, acc[15]);

// A_0 = x^T, zero padded to 16 rows.  grid (ceil(in_dim * 16 / 256), m).
__global__ void l2a_adapt_prep_k(const L2AAdaptParams p) {
    const int task = blockIdx.y, in_dim = p.dims[0];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= in_dim * L2A_AR) return;
    const int k = i / L2A_AR, r = i - k * L2A_AR;
    p.scratch[(long long)task * p.scratch_stride + p.a_off[0] + i] =
        (r < p.rows) ? p.x[((long long)task * p.rows + r) * in_dim + k] : 0.0f;
}

// Forward through layer l for 64 output units of one task: grid (ceil(n_out / 64), m), 4 waves, wave w sums
// its quarter of the k range (weights coalesced over the units), partials meet in LDS.  The last layer
// writes dZ_L = 2 (y_hat - y) / (rows * obs_dim) instead of its output.
__global__ void __launch_bounds__(256) l2a_adapt_fwd_k(const L2AAdaptParams p, int l) {
    __shared__ float red[3][64][L2A_AR + 1];
    const int lane = threadIdx.x & 63, ks = threadIdx.x >> 6;
    const int task = blockIdx.y;
    const int k_in = p.dims[l], n_out = p.dims[l + 1];
    const int u = blockIdx.x * 64 + lane;
    const bool live = u < n_out;
    float* sc = p.scratch + (long long)task * p.scratch_stride;
    const float* A = sc + p.a_off[l];
    const float* W = p.w[l];
    const int chunk = (k_in + 3) / 4;
    const int k0 = ks * chunk, k1 = (k0 + chunk < k_in) ? k0 + chunk : k_in;
    float acc[L2A_AR];
#pragma unroll
    for (int r = 0; r < L2A_AR; ++r) acc[r] = 0.0f;
    for (int k = k0; k < k1; ++k) {
        const float w = live ? W[(long long)k * n_out + u] : 0.0f;
        const float4* a = reinterpret_cast<const float4*>(A + k * L2A_AR);
        const float4 a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
        L2A_ACC16(acc, w, a0, a1, a2, a3)
    }
    if (ks > 0) {
#pragma unroll
        for (int r = 0; r < L2A_AR; ++r) red[ks - 1][lane][r] = acc[r];
    }
    __syncthreads();
    if (ks != 0 || !live) return;
    const float bias = p.b[l][u];
    const bool last = (l == p.n_layers - 1);
    const float scale = 2.0f / (float)(p.rows * n_out);
    float* dst = sc + (last ? p.z_off[l + 1] : p.a_off[l + 1]) + u * L2A_AR;
#pragma unroll
    for (int r = 0; r < L2A_AR; ++r) {
        float v = ((acc[r] + red[0][lane][r]) + red[1][lane][r]) + red[2][lane][r] + bias;
        if (last) v = (r < p.rows) ? scale * (v - p.y[((long long)task * p.rows + r) * n_out + u]) : 0.0f;
        else v = l2a_act1(v, p.hidden_act);
        dst[r] = v;
    }
}

// dZ_l = (W_l dZ_{l+1}) * act'(A_l) for 64 input units k of layer l (1 <= l < L): grid (ceil(k_in / 64), m),
// wave w sums its quarter of the u range.  Each lane walks its own weight row (16 consecutive u share a
// cache line, so the row is fetched once).
__global__ void __launch_bounds__(256) l2a_adapt_bwd_k(const L2AAdaptParams p, int l) {
    __shared__ float red[3][64][L2A_AR + 1];
    const int lane = threadIdx.x & 63, us = threadIdx.x >> 6;
    const int task = blockIdx.y;
    const int k_in = p.dims[l], n_out = p.dims[l + 1];
    const int k = blockIdx.x * 64 + lane;
    const bool live = k < k_in;
    float* sc = p.scratch + (long long)task * p.scratch_stride;
    const float* Z = sc + p.z_off[l + 1];
    const float* wrow = p.w[l] + (long long)(live ? k : 0) * n_out;
    const int chunk = (n_out + 3) / 4;
    const int u0 = us * chunk, u1 = (u0 + chunk < n_out) ? u0 + chunk : n_out;
    float acc[L2A_AR];
#pragma unroll
    for (int r = 0; r < L2A_AR; ++r) acc[r] = 0.0f;
    for (int u = u0; u < u1; ++u) {
        const float w = wrow[u];
        const float4* z = reinterpret_cast<const float4*>(Z + u * L2A_AR);
        const float4 z0 = z[0], z1 = z[1], z2 = z[2], z3 = z[3];
        L2A_ACC16(acc, w, z0, z1, z2, z3)
    }
    if (us > 0) {
#pragma unroll
        for (int r = 0; r < L2A_AR; ++r) red[us - 1][lane][r] = acc[r];
    }
    __syncthreads();
    if (us != 0 || !live) return;
    const float* A = sc + p.a_off[l] + k * L2A_AR;
    float* dst = sc + p.z_off[l] + k * L2A_AR;
#pragma unroll
    for (int r = 0; r < L2A_AR; ++r) {
        const float v = ((acc[r] + red[0][lane][r]) + red[1][lane][r]) + red[2][lane][r];
        dst[r] = v * l2a_act_grad_from_output(A[r], p.hidden_act);
    }
}

// Layer l of every task (blockIdx.y): theta' = theta - lr * A_l^T dZ_{l+1}; one thread per kernel element,
// the first n_out threads of the grid also do the bias.  Writes the raw layout and (when pk != null) the
// MFMA fragment order (inverse of l2a_pack_decode); `pk_bias` = padded output-bias copy (last layer only).
__global__ void l2a_adapt_update_k(const L2AAdaptParams p, int l, float lr, float* __restrict__ blk,
                                   long long set_stride, long long raw_w, long long raw_b, long long pk,
                                   int has_pk, long long pk_bias, int has_pk_bias) {
    const int k_in = p.dims[l], n_out = p.dims[l + 1];
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)k_in * n_out) return;
    const int task = blockIdx.y;
    const int k = (int)(idx / n_out), u = (int)(idx - (long long)k * n_out);
    const float* sc = p.scratch + (long long)task * p.scratch_stride;
    const float4* a = reinterpret_cast<const float4*>(sc + p.a_off[l] + k * L2A_AR);
    const float4* z = reinterpret_cast<const float4*>(sc + p.z_off[l + 1] + u * L2A_AR);
    float g = 0.0f, gb = 0.0f;
#pragma unroll
    for (int q = 0; q < L2A_AR / 4; ++q) {
        const float4 av = a[q], zv = z[q];
        g = fmaf(av.x, zv.x, g); g = fmaf(av.y, zv.y, g); g = fmaf(av.z, zv.z, g); g = fmaf(av.w, zv.w, g);
        gb += (zv.x + zv.y) + (zv.z + zv.w);
    }
    float* dst = blk + (long long)task * set_stride;
    const float wn = p.w[l][idx] - lr * g;
    dst[raw_w + idx] = wn;
    if (has_pk) {
        const int KG = (k_in + 15) / 16;
        const long long pidx = ((((long long)(u >> 4) * KG + (k >> 4)) * 64 + ((u & 15) + 16 * ((k & 15) >> 2))) << 2) + (k & 3);
        dst[pk + pidx] = wn;
    }
    if (k == 0) {
        const float bn = p.b[l][u] - lr * gb;
        dst[raw_b + u] = bn;
        if (has_pk_bias) dst[pk_bias + u] = bn;
    }
}
