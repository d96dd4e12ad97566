// l2a_adapt.h - GrBAL's inner adaptation step on the GPU (included by l2a_api.hip only).
//
// Reference: `MetaMLPDynamicsModel.adapt` (dynamics/meta_mlp_dynamics.py:321-345) runs, for every task i,
// one SGD step theta_i = theta - lr * grad MSE(f_theta(x_i), y_i) (`_adapt_sym`, :409-421; loss :118) on
// `adapt_batch_size` (16, run_scripts/run_grbal.py) normalised transitions, pulls the m adapted parameter
// sets to the host and feeds them back on every later `sess.run`.  Here the step reads the base parameters once and
// writes the adapted sets straight into the per-block model the planner launches on - raw layout AND MFMA fragment
// order - so nothing is re-uploaded or re-packed.  2 L - 1 launches for L layers (round 5; 2 L with the update as a launch of
// its own until then):
//
//   l2a_adapt_fwd0_k, l2a_adapt_fwd_k (L - 1), l2a_adapt_bwd_k (L - 1) : the forward and backward pass, one launch per
//                        layer, every launch spread over (64-unit slice, task) workgroups of 8 waves that each take an
//                        eighth of the reduction - the first version ran one workgroup per task through all layers and
//                        was latency bound at ~1 ms, no better than the 45 stock PyTorch launches it replaced.  Layers
//                        >= 1 run on the matrix core (16x16x4 fp32: units x rows x k), both operands coalesced vector
//                        loads, 8 k-steps of operands in flight before the first MFMA; layer 0 (K = obs + act) is a
//                        lane-per-unit FMA loop over the batch staged in LDS (it reads x itself: no transpose pass).
//                        dZ_L = 2 (y_hat - y) / (rows * obs_dim), dZ_l = (W_l dZ_{l+1}) * act'(A_l); layer inputs A_l
//                        and the dZ_l ([dim][16 rows]) live in a scratch buffer.  History (rocprofv3 dispatch trace,
//                        tools/adapt_trace.sh): unbatched lane-per-unit loops, one exposed round trip per k: 58 us per
//                        512 x 512 layer; batched loads + scalar loads for the activation rows: 19 us (the SGPR budget
//                        keeps 5 of 8 row loads in flight); matrix core: see DESIGN.md 4.3
//   the update             : blocks of (256 output units, 16 input rows, task): each thread keeps its unit's dZ row in
//                        registers and walks the input rows, g = sum_r A_l[k][r] dZ_{l+1}[u][r], theta' = theta - lr g ->
//                        raw kernel, packed kernels; biases alike.  Layer l's blocks ride in the launch that computes dZ_l
//                        (l2a_adapt_bwdu_k: both operands were complete before it started); layer 0's columns are updated
//                        by the last backward launch's own workgroups
//
// Rows beyond `rows` (padding up to 16) carry dZ = 0 and therefore no gradient.
#pragma once

#include "l2a_micro_pack.h"

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
    long long y_off;                    // raw mode: the normalised target deltas [obs_dim][16] (written by the first launch)
    int hmax;
    // raw mode (l2a_model_adapt_sgd_raw): the batches arrive un-normalised as float64 and are normalised here exactly as
    // the reference does on the host - (v - mean) / (std + 1e-10) in float64, then the cast to fp32 (mlp_dynamics.py:265-266,
    // meta_mlp_dynamics.py:321-345) - x / y are then null
    const double* raw_obs;              // [m, rows, obs_dim]
    const double* raw_act;              // [m, rows, act_dim]
    const double* raw_next;             // [m, rows, obs_dim]
    const double* raw_norm;             // mean_obs[od] std_obs[od] mean_act[ad] std_act[ad] mean_delta[od] std_delta[od]
    int obs_dim, act_dim;
    unsigned long long* dbg;            // timeline builds (tools/timeline_adapt.py): [phase 8][wave 8][slot 8] shader clocks of workgroup (0, 0)
};

// Phase timeline (builds with -DL2A_TIMELINE only): the waves of workgroup (0, 0) stamp the shader clock (slot 7: the 100 MHz
// constant clock, s_memrealtime, at the same instant as slot 0 / the last slot - converts clocks to microseconds).
#ifndef L2A_TIMELINE
#define L2A_ATS(phase, slot)
#define L2A_ATS_REAL(phase, slot)
#else
#define L2A_ATS(phase, slot)                                                                            \
    if (p.dbg && blockIdx.x == 0 && blockIdx.y == 0) {                                                  \
        unsigned long long ts_;                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_) : : "memory");                  \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        if ((threadIdx.x & 63) == 0) p.dbg[(((phase) * 8) + (threadIdx.x >> 6)) * 8 + (slot)] = ts_;    \
    }
#define L2A_ATS_REAL(phase, slot)                                                                       \
    if (p.dbg && blockIdx.x == 0 && blockIdx.y == 0) {                                                  \
        unsigned long long ts_;                                                                         \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_) : : "memory");              \
        if ((threadIdx.x & 63) == 0) p.dbg[(((phase) * 8) + (threadIdx.x >> 6)) * 8 + (slot)] = ts_;    \
    }
#endif

__device__ __forceinline__ float l2a_act_grad_from_output(float o, int kind) {
    switch (kind) {
        case L2A_ACT_RELU: return o > 0.0f ? 1.0f : 0.0f;
        case L2A_ACT_TANH: return 1.0f - o * o;
        case L2A_ACT_SIGMOID: return o * (1.0f - o);
        default: return 1.0f;
    }
}

#define L2A_AW 8        // waves per forward / backward workgroup (each takes 1 / 8 of the reduction)
#define L2A_SB 16       // MFMA k-steps (of 4) whose operands are fetched before the first MFMA of a batch: a wave's whole share of
                        // K = 512 (round 4; 8 until then - two exposed round trips per forward launch, 13.5 -> 10.2 us)
#define L2A_XS_MAX 128  // widest input layer whose batch is staged through LDS in the first forward launch

// The 8 waves' accumulator tiles (4 tiles x f32x4 per lane: D[unit or k][row]) meet in LDS; then EVERY thread sums two
// outputs - local unit / k index ul = tid >> 3, rows 2 (tid & 7) + {0, 1} - over the waves in the fixed order 0 .. 7 (what
// wave 0 alone did until round 5: 112 dependent LDS reads and the whole epilogue on one wave, 4 - 6k of a launch's ~20k
// clocks).  VEC: a tile's M index m holds local unit 4 m + t (weight rows read 16 bytes per lane), else 16 t + m.
template <bool VEC>
__device__ __forceinline__ void l2a_adapt_reduce2(const f32x4 (&acc)[4], float* red, int w, int lane, float (&out)[2]) {
    const int i16 = lane & 15, q = lane >> 4;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ul = VEC ? 4 * (4 * q + i) + t : 16 * t + 4 * q + i;
            red[(w * 64 + ul) * (L2A_AR + 1) + i16] = acc[t][i];
        }
    __syncthreads();
    const int ul = threadIdx.x >> 3, r0 = (threadIdx.x & 7) * 2;
    float v0 = red[ul * (L2A_AR + 1) + r0], v1 = red[ul * (L2A_AR + 1) + r0 + 1];
#pragma unroll
    for (int ww = 1; ww < L2A_AW; ++ww) {
        v0 += red[(ww * 64 + ul) * (L2A_AR + 1) + r0];
        v1 += red[(ww * 64 + ul) * (L2A_AR + 1) + r0 + 1];
    }
    out[0] = v0; out[1] = v1;
}

// First forward launch (layer 0, K = obs_dim + act_dim): 64 output units of one task per workgroup, lane = unit,
// wave w sums its eighth of the k range.  A_0[k][r] = x[task][r][k] (zero beyond `rows`); x may live in host-mapped
// memory (l2a_model_adapt_sgd_host): one coalesced pass brings it into LDS - a single bus round trip instead of one
// per element - and workgroup 0 of the task publishes A_0 for the update pass and, in raw mode, the normalised target
// deltas (so that the last forward launch does not cross the bus again).  grid (ceil(n_out / 64), m).
__global__ void __launch_bounds__(64 * L2A_AW) l2a_adapt_fwd0_k(const L2AAdaptParams p) {
    __shared__ float red[L2A_AW * 64 * (L2A_AR + 1)];
    __shared__ float xs[L2A_XS_MAX * L2A_AR];
    const int lane = threadIdx.x & 63, ks = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    L2A_ATS(0, 0) L2A_ATS_REAL(0, 6)
    const int task = blockIdx.y;
    const int k_in = p.dims[0], n_out = p.dims[1];
    const int u = blockIdx.x * 64 + lane;
    const bool live = u < n_out;
    float* sc = p.scratch + (long long)task * p.scratch_stride;
    const float* W = p.w[0] + (live ? u : 0);           // dead lanes read unit 0's column and never store
    // this thread's share of the epilogue (as in the other launches): unit blockIdx.x * 64 + (tid >> 3), rows 2 (tid & 7) + {0, 1};
    // its bias is requested now, used after the reduction
    const int ue = blockIdx.x * 64 + (threadIdx.x >> 3), r0 = (threadIdx.x & 7) * 2;
    const float bias = (ue < n_out) ? p.b[0][ue] : 0.0f;
    const int chunk = (k_in + L2A_AW - 1) / L2A_AW;
    const int k0 = ks * chunk, k1 = (k0 + chunk < k_in) ? k0 + chunk : k_in;
    const float* x = p.x + (long long)task * p.rows * k_in;
    const bool staged = k_in <= L2A_XS_MAX;             // (raw mode is only launched for staged widths)
    constexpr int CH = (L2A_XS_MAX + L2A_AW - 1) / L2A_AW;
    float wv[CH];
    if (staged) {
        // (a wave's weights - at most 16 rows of a staged width - requested together and BEFORE the batch crosses the bus)
#pragma unroll
        for (int j = 0; j < CH; ++j) wv[j] = (k0 + j < k1) ? W[(long long)(k0 + j) * n_out] : 0.0f;
        const int od = p.obs_dim, ad = p.act_dim;
        if (p.raw_obs) {
            // raw mode: the batch and the normalisation vectors sit in HOST-mapped staging - every load is a bus round trip
            // (~2.5 us).  ALL of a thread's loads are issued before the first is used: one round trip per workgroup, where the
            // loops below (one element at a time, x then y) took up to four in a row (15 us for this launch, round 5: ~9)
            constexpr int NX = (L2A_XS_MAX * L2A_AR + 64 * L2A_AW - 1) / (64 * L2A_AW);      // 4 elements of x per thread at most
            double xr[NX], xm[NX], xd[NX], yn[NX], yo[NX], ym[NX], yd[NX];
            const int n_y = p.dims[p.n_layers];
            const bool do_y = blockIdx.x == 0;
#pragma unroll
            for (int q = 0; q < NX; ++q) {
                const int i = threadIdx.x + q * 64 * L2A_AW;
                const int r = i / k_in, kk = i - r * k_in;
                xr[q] = 0.0; xm[q] = 0.0; xd[q] = 1.0;
                if (i < k_in * L2A_AR && r < p.rows) {
                    const bool is_obs = kk < od;
                    xr[q] = is_obs ? p.raw_obs[((long long)task * p.rows + r) * od + kk] : p.raw_act[((long long)task * p.rows + r) * ad + (kk - od)];
                    xm[q] = is_obs ? p.raw_norm[kk] : p.raw_norm[2 * od + (kk - od)];
                    xd[q] = is_obs ? p.raw_norm[od + kk] : p.raw_norm[2 * od + ad + (kk - od)];
                }
                const int ry = i / n_y, uy = i - ry * n_y;
                yn[q] = 0.0; yo[q] = 0.0; ym[q] = 0.0; yd[q] = 1.0;
                if (do_y && i < n_y * L2A_AR && ry < p.rows) {
                    const long long e = ((long long)task * p.rows + ry) * n_y + uy;
                    yn[q] = p.raw_next[e]; yo[q] = p.raw_obs[e];
                    ym[q] = p.raw_norm[2 * od + 2 * ad + uy]; yd[q] = p.raw_norm[3 * od + 2 * ad + uy];
                }
            }
#pragma unroll
            for (int q = 0; q < NX; ++q) {
                const int i = threadIdx.x + q * 64 * L2A_AW;
                const int r = i / k_in, kk = i - r * k_in;
                // (v - mean) / (std + 1e-10) in float64, then the cast: the host's arithmetic (mlp_dynamics.py:265-266)
                if (i < k_in * L2A_AR) xs[kk * L2A_AR + r] = (r < p.rows) ? (float)((xr[q] - xm[q]) / (xd[q] + 1e-10)) : 0.0f;
                const int ry = i / n_y, uy = i - ry * n_y;
                if (do_y && i < n_y * L2A_AR)
                    sc[p.y_off + uy * L2A_AR + ry] = (ry < p.rows) ? (float)(((yn[q] - yo[q]) - ym[q]) / (yd[q] + 1e-10)) : 0.0f;
            }
        } else {
            for (int i = threadIdx.x; i < k_in * L2A_AR; i += blockDim.x) {      // x in its own order: coalesced
                const int r = i / k_in, kk = i - r * k_in;
                xs[kk * L2A_AR + r] = (r < p.rows) ? x[i] : 0.0f;
            }
        }
        __syncthreads();
        L2A_ATS(0, 1)
        if (blockIdx.x == 0)
            for (int i = threadIdx.x; i < k_in * L2A_AR; i += blockDim.x) sc[p.a_off[0] + i] = xs[i];
    }
    float acc[L2A_AR];
#pragma unroll
    for (int r = 0; r < L2A_AR; ++r) acc[r] = 0.0f;
    if (staged) {
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            if (k0 + j < k1) {
#pragma unroll
                for (int r = 0; r < L2A_AR; ++r) acc[r] = fmaf(wv[j], xs[(k0 + j) * L2A_AR + r], acc[r]);
            }
        }
    } else {
        for (int k = k0; k < k1; ++k) {
            const float w = W[(long long)k * n_out];
#pragma unroll
            for (int r = 0; r < L2A_AR; ++r) {
                const float a = (r < p.rows) ? x[r * k_in + k] : 0.0f;
                if (blockIdx.x == 0 && lane == r) sc[p.a_off[0] + k * L2A_AR + r] = a;
                acc[r] = fmaf(w, a, acc[r]);
            }
        }
    }
    L2A_ATS(0, 2)
    // the waves' partial chains meet in LDS; every thread sums two outputs over the waves in the fixed order 0 .. 7
#pragma unroll
    for (int r = 0; r < L2A_AR; ++r) red[(ks * 64 + lane) * (L2A_AR + 1) + r] = acc[r];
    __syncthreads();
    L2A_ATS(0, 3)
    if (ue < n_out) {
        const int ul = threadIdx.x >> 3;
        float v0 = red[ul * (L2A_AR + 1) + r0], v1 = red[ul * (L2A_AR + 1) + r0 + 1];
#pragma unroll
        for (int w = 1; w < L2A_AW; ++w) {
            v0 += red[(w * 64 + ul) * (L2A_AR + 1) + r0];
            v1 += red[(w * 64 + ul) * (L2A_AR + 1) + r0 + 1];
        }
        *reinterpret_cast<float2*>(sc + p.a_off[1] + ue * L2A_AR + r0) =
            make_float2(l2a_act1(v0 + bias, p.hidden_act), l2a_act1(v1 + bias, p.hidden_act));
    }
    L2A_ATS(0, 4) L2A_ATS_REAL(0, 7)
}

// Forward through layer l >= 1 for 64 output units of one task on the matrix core: D[unit][row] += W^T[unit][k] A_l[k][row]
// as v_mfma_f32_16x16x4_f32 (4 unit tiles per wave; the activation rows are one coalesced 256-byte load per k-step).  The 8
// waves split the k-steps; a wave's whole share of K is requested before its first MFMA; partial tiles meet in LDS in a fixed
// order.  VEC (layer widths that are multiples of four, 16-byte aligned kernels: every hidden layer): lane (i, q) reads
// W[k + q][u0 + 4 i .. 4 i + 3] with ONE 16-byte load and tile t takes element t - unit u0 + 4 i + t sits at M index i of
// tile t (round 5; until then four dword loads per k-step, 640 load instructions per workgroup: 4.3 - 4.8k clocks before the
// first MFMA could issue); otherwise (the 41-wide output layer) lane (i, q) reads W[k + q][u0 + 16 t + i] per tile.
// The last layer writes dZ_L = 2 (y_hat - y) / (rows * obs_dim) instead of its output.  grid (ceil(n_out / 64), m).
// `A`: the layer's input rows [k][16] (global scratch, or LDS when the previous layer was computed by this workgroup); `u0`:
// first output unit; `dst`: where outputs [u][16] go (global scratch or LDS), indexed by u - dst_u0.
template <bool VEC>
__device__ __forceinline__ void l2a_adapt_fwd_body(const L2AAdaptParams& p, const int l, float* red, const float* A, const int astride,
                                                   const int u0, float* dst_base, const int dst_u0) {
    const int lane = threadIdx.x & 63, ks = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16 = lane & 15, q = lane >> 4;
    const int task = blockIdx.y;
    const int k_in = p.dims[l], n_out = p.dims[l + 1];
    float* sc = p.scratch + (long long)task * p.scratch_stride;
    const float* W = p.w[l];
    const bool last = (l == p.n_layers - 1);
    // what the epilogue of this thread needs - requested first, consumed last
    const int ul = threadIdx.x >> 3, r0 = (threadIdx.x & 7) * 2;
    const int ue = u0 + ul;
    const bool ue_ok = ue < n_out;
    const float bias_e = ue_ok ? p.b[l][ue] : 0.0f;
    float y_e[2] = {0.0f, 0.0f};
    if (last && ue_ok) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int r = r0 + c;
            if (r < p.rows) y_e[c] = p.raw_obs ? sc[p.y_off + ue * L2A_AR + r] : p.y[((long long)task * p.rows + r) * n_out + ue];
        }
    }
    int ucol[4];
    bool uok[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int u = VEC ? u0 + 4 * i16 + t : u0 + 16 * t + i16;
        uok[t] = u < n_out;
        ucol[t] = uok[t] ? u : 0;
    }
    const int steps = (k_in + 3) / 4, per = (steps + L2A_AW - 1) / L2A_AW;
    const int s0 = ks * per < steps ? ks * per : steps, s1 = (s0 + per < steps) ? s0 + per : steps;
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int s = s0; s < s1; s += L2A_SB) {
        f32x4 a[L2A_SB];
        float b[L2A_SB];
#pragma unroll
        for (int j = 0; j < L2A_SB; ++j) {
            const int k = 4 * (s + j) + q;
            const bool ok = (s + j < s1) && (k < k_in);
            const int kk = ok ? k : 0;
            b[j] = A[kk * astride + i16];
            if (VEC) {
                a[j] = *reinterpret_cast<const f32x4*>(W + (long long)kk * n_out + ucol[0]);
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) a[j][t] = W[(long long)kk * n_out + ucol[t]];
            }
            if (!ok) b[j] = 0.0f;           // a zero row annihilates whatever the clamped weight loads returned
        }
        __builtin_amdgcn_sched_barrier(0);  // every load of the batch in flight before the first MFMA
        L2A_ATS(l, 1)
#pragma unroll
        for (int j = 0; j < L2A_SB; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = L2A_MFMA(uok[t] ? a[j][t] : 0.0f, b[j], acc[t]);
    }
    L2A_ATS(l, 2)
    float v[2];
    l2a_adapt_reduce2<VEC>(acc, red, ks, lane, v);
    L2A_ATS(l, 3)
    if (ue_ok) {
        const float scale = 2.0f / (float)(p.rows * n_out);
        float o[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float z = v[c] + bias_e;
            o[c] = last ? ((r0 + c < p.rows) ? scale * (z - y_e[c]) : 0.0f) : l2a_act1(z, p.hidden_act);
        }
        float* dst = dst_base + (ue - dst_u0) * L2A_AR + r0;
        *reinterpret_cast<float2*>(dst) = make_float2(o[0], o[1]);
    }
    L2A_ATS(l, 4)
}

__device__ __forceinline__ bool l2a_adapt_vec_ok(const L2AAdaptParams& p, int l) {
    return (p.dims[l + 1] & 3) == 0 && (reinterpret_cast<unsigned long long>(p.w[l]) & 15ull) == 0;
}

__global__ void __launch_bounds__(64 * L2A_AW) l2a_adapt_fwd_k(const L2AAdaptParams p, int l) {
    __shared__ float red[L2A_AW * 64 * (L2A_AR + 1)];
    L2A_ATS(l, 0) L2A_ATS_REAL(l, 6)
    float* sc = p.scratch + (long long)blockIdx.y * p.scratch_stride;
    const float* A = sc + p.a_off[l];
    float* dst = sc + ((l == p.n_layers - 1) ? p.z_off[l + 1] : p.a_off[l + 1]);
    if (l2a_adapt_vec_ok(p, l)) l2a_adapt_fwd_body<true>(p, l, red, A, L2A_AR, blockIdx.x * 64, dst, 0);
    else l2a_adapt_fwd_body<false>(p, l, red, A, L2A_AR, blockIdx.x * 64, dst, 0);
    L2A_ATS_REAL(l, 7)
}

// Where the adapted sets go: the per-block model's weight block (raw reference layout + MFMA fragment order), and
// how an update block index maps to (256-unit block, 16-row chunk) of its layer.
#define L2A_UK 16       // input rows per update block
struct L2AAdaptDst {
    float* blk;
    long long set_stride;
    long long raw_w[L2A_MAX_LAYERS], raw_b[L2A_MAX_LAYERS], pk[L2A_MAX_LAYERS];
    long long pk_bout;
    int has_pk;
    int has_mk, mk_H, mk_KG0, mk_o4;        // the micro-tile kernel's copy (l2a_micro_pack.h), at offset mk
    long long mk;
    float lr;
};

__device__ __forceinline__ int l2a_adapt_update_blocks(const L2AAdaptParams& p, int l) {
    return ((p.dims[l + 1] + 255) / 256) * ((p.dims[l] + L2A_UK - 1) / L2A_UK);
}

// g = sum_r A[k][r] dZ[u][r] as ONE sequential fp32 FMA chain over the rows (every form of the update uses this order)
__device__ __forceinline__ float l2a_adapt_grad(const float4 a0, const float4 a1, const float4 a2, const float4 a3, const float4 z0,
                                                const float4 z1, const float4 z2, const float4 z3) {
    float g = 0.0f;
    g = fmaf(a0.x, z0.x, g); g = fmaf(a0.y, z0.y, g); g = fmaf(a0.z, z0.z, g); g = fmaf(a0.w, z0.w, g);
    g = fmaf(a1.x, z1.x, g); g = fmaf(a1.y, z1.y, g); g = fmaf(a1.z, z1.z, g); g = fmaf(a1.w, z1.w, g);
    g = fmaf(a2.x, z2.x, g); g = fmaf(a2.y, z2.y, g); g = fmaf(a2.z, z2.z, g); g = fmaf(a2.w, z2.w, g);
    g = fmaf(a3.x, z3.x, g); g = fmaf(a3.y, z3.y, g); g = fmaf(a3.z, z3.z, g); g = fmaf(a3.w, z3.w, g);
    return g;
}
__device__ __forceinline__ float l2a_adapt_bias_grad(const float4 z0, const float4 z1, const float4 z2, const float4 z3) {
    float gb = 0.0f;
    gb += (z0.x + z0.y) + (z0.z + z0.w);
    gb += (z1.x + z1.y) + (z1.z + z1.w);
    gb += (z2.x + z2.y) + (z2.z + z2.w);
    gb += (z3.x + z3.y) + (z3.z + z3.w);
    return gb;
}

// One new weight into the adapted set's three copies (raw layout, MFMA fragment order, micro-tile order), element by element
__device__ __forceinline__ void l2a_adapt_put(const L2AAdaptParams& p, const L2AAdaptDst& d, float* dst, int l, int k, int u, float wn) {
    const int k_in = p.dims[l], n_out = p.dims[l + 1];
    const int KG = (k_in + 15) / 16;
    dst[d.raw_w[l] + (long long)k * n_out + u] = wn;
    if (d.has_pk) {
        const long long pidx = ((((long long)(u >> 4) * KG + (k >> 4)) * 64 + ((u & 15) + 16 * ((k & 15) >> 2))) << 2) + (k & 3);
        dst[d.pk[l] + pidx] = wn;
    }
    if (d.has_mk) dst[d.mk + l2a_mlp_micro_index(d.mk_H, d.mk_KG0, p.n_layers - 1, d.mk_o4, l, k, u)] = wn;
}

// theta' = theta - lr * A_l^T dZ_{l+1} for update block `b` of layer l (256 threads: `tid`).  A thread owns one output unit u
// (its dZ row stays in registers) and walks 16 input rows k; the row-0 blocks also do the biases.  Writes the raw layout, the
// MFMA fragment order (has_pk; inverse of l2a_pack_decode) and the micro-tile order (has_mk); the last layer's bias also goes
// to the padded output-bias copy.
__device__ __forceinline__ void l2a_adapt_update_block(const L2AAdaptParams& p, const L2AAdaptDst& d, const int l, const int b,
                                                       const int task, const int tid) {
    const int k_in = p.dims[l], n_out = p.dims[l + 1];
    const int ublocks = (n_out + 255) / 256;
    const int kc = b / ublocks, ub = b - kc * ublocks;
    const int u = ub * 256 + tid;
    if (u >= n_out) return;
    const float* sc = p.scratch + (long long)task * p.scratch_stride;
    const float4* zp = reinterpret_cast<const float4*>(sc + p.z_off[l + 1] + u * L2A_AR);
    const float4 z0 = zp[0], z1 = zp[1], z2 = zp[2], z3 = zp[3];
    float* dst = d.blk + (long long)task * d.set_stride;
    const int KG = (k_in + 15) / 16;
    const int k0 = kc * L2A_UK, k1 = (k0 + L2A_UK < k_in) ? k0 + L2A_UK : k_in;
    const float* wsrc = p.w[l] + u;
    auto grad = [&](int k) {
        const float4* a = reinterpret_cast<const float4*>(sc + p.a_off[l] + k * L2A_AR);
        return l2a_adapt_grad(a[0], a[1], a[2], a[3], z0, z1, z2, z3);
    };
    // where the micro-tile copy keeps four rows of a 16-row block side by side: rows 16 a + c + 4 j, j = 0 .. 3 (chain order), unless
    // the O4 output slots scatter them over lanes
    const bool mk_vec = d.has_mk && !(l == p.n_layers - 1 && d.mk_o4 && u >= 16);
    if (k1 - k0 == L2A_UK) {
        // a whole block of sixteen rows: the new weights in registers, then 16-byte stores into the two packed copies (the MFMA
        // fragment order keeps rows 4 a .. 4 a + 3 of a unit side by side)
        float wn[L2A_UK];
#pragma unroll
        for (int kk = 0; kk < L2A_UK; ++kk) wn[kk] = wsrc[(long long)(k0 + kk) * n_out] - d.lr * grad(k0 + kk);
#pragma unroll
        for (int kk = 0; kk < L2A_UK; ++kk) dst[d.raw_w[l] + (long long)(k0 + kk) * n_out + u] = wn[kk];
        if (d.has_pk) {
#pragma unroll
            for (int a4 = 0; a4 < 4; ++a4) {
                const long long pidx = (((long long)(u >> 4) * KG + (k0 >> 4)) * 64 + ((u & 15) + 16 * a4)) << 2;
                *reinterpret_cast<f32x4*>(dst + d.pk[l] + pidx) = (f32x4){wn[4 * a4], wn[4 * a4 + 1], wn[4 * a4 + 2], wn[4 * a4 + 3]};
            }
        }
        if (mk_vec) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                *reinterpret_cast<f32x4*>(dst + d.mk + l2a_mlp_micro_index(d.mk_H, d.mk_KG0, p.n_layers - 1, d.mk_o4, l, k0 + c, u)) =
                    (f32x4){wn[c], wn[c + 4], wn[c + 8], wn[c + 12]};
        } else if (d.has_mk) {
#pragma unroll
            for (int kk = 0; kk < L2A_UK; ++kk)
                dst[d.mk + l2a_mlp_micro_index(d.mk_H, d.mk_KG0, p.n_layers - 1, d.mk_o4, l, k0 + kk, u)] = wn[kk];
        }
    } else {
#pragma unroll 4
        for (int k = k0; k < k1; ++k) l2a_adapt_put(p, d, dst, l, k, u, wsrc[(long long)k * n_out] - d.lr * grad(k));
    }
    if (kc == 0) {
        const float bn = p.b[l][u] - d.lr * l2a_adapt_bias_grad(z0, z1, z2, z3);
        dst[d.raw_b[l] + u] = bn;
        if (d.has_pk && l == p.n_layers - 1) dst[d.pk_bout + u] = bn;
    }
}

// dZ_l = (W_l dZ_{l+1}) * act'(A_l) for the 64 input units k0 .. k0 + 63 of layer l (1 <= l < L) on the matrix core:
// D[k][row] += W[k][u] dZ[u][row]; lane (i, q) supplies W[k0 + 16 t + i][u + q] (hidden layers: 16 weight rows x 16 bytes per
// load, a row's line reused by the next 7 u-steps) and dZ_{l+1}[u + q][i] from `Z` (global scratch, or LDS when this workgroup
// computed it); the waves split the u-steps.  Every thread ends up with two outputs (k0 + (tid >> 3), rows 2 (tid & 7) + {0, 1}),
// stores them to the task's scratch and returns them.
__device__ __forceinline__ void l2a_adapt_bwd_body(const L2AAdaptParams& p, const int l, float* red, const float* Z, const int k0,
                                                   const int ph, float (&zout)[2]) {
    (void)ph;
    const int lane = threadIdx.x & 63, us = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16 = lane & 15, q = lane >> 4;
    const int task = blockIdx.y;
    const int k_in = p.dims[l], n_out = p.dims[l + 1];
    float* sc = p.scratch + (long long)task * p.scratch_stride;
    // what this thread's epilogue needs (the activations whose derivative scales its two outputs) - requested first
    const int kl = threadIdx.x >> 3, r0 = (threadIdx.x & 7) * 2;
    const int ke = k0 + kl;
    const bool ke_ok = ke < k_in;
    const float2 a_e = ke_ok ? *reinterpret_cast<const float2*>(sc + p.a_off[l] + ke * L2A_AR + r0) : make_float2(0.f, 0.f);
    const float* wrow[4];
    bool kok[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int k = k0 + 16 * t + i16;
        kok[t] = k < k_in;
        wrow[t] = p.w[l] + (long long)(kok[t] ? k : 0) * n_out;
    }
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if ((n_out & 15) == 0 && (reinterpret_cast<unsigned long long>(p.w[l]) & 15ull) == 0) {
        // Hidden layers (n_out a multiple of 16, rows 16-byte aligned): the weight rows are read 16 BYTES per lane - lane (i, q) takes W[k_i][16 G + 4 q + 0..3]
        // for a group G of sixteen u, and MFMA j of the group contracts u = 16 G + 4 q' + j over the quarters q' (dZ read to match) -
        // a quarter of the load instructions and of the cache lines each touches
        constexpr int GB = 4;                           // groups (of sixteen u) in flight: one batch covers a wave's share of 512
        const int groups = n_out >> 4, perg = (groups + L2A_AW - 1) / L2A_AW;
        const int g0 = us * perg < groups ? us * perg : groups, g1 = (g0 + perg < groups) ? g0 + perg : groups;
        for (int g = g0; g < g1; g += GB) {
            f32x4 a4[GB][4];
            float b[GB][4];
#pragma unroll
            for (int jg = 0; jg < GB; ++jg) {
                const bool ok = g + jg < g1;
                const int u4 = 16 * (ok ? g + jg : g0) + 4 * q;
#pragma unroll
                for (int t = 0; t < 4; ++t) a4[jg][t] = *reinterpret_cast<const f32x4*>(wrow[t] + u4);
#pragma unroll
                for (int j = 0; j < 4; ++j) b[jg][j] = ok ? Z[(u4 + j) * L2A_AR + i16] : 0.0f;
            }
            __builtin_amdgcn_sched_barrier(0);          // every load of the batch in flight before the first MFMA
            L2A_ATS(ph, 1)
#pragma unroll
            for (int jg = 0; jg < GB; ++jg)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t] = L2A_MFMA(kok[t] ? a4[jg][t][j] : 0.0f, b[jg][j], acc[t]);
        }
    } else {
        const int steps = (n_out + 3) / 4, per = (steps + L2A_AW - 1) / L2A_AW;
        const int s0 = us * per < steps ? us * per : steps, s1 = (s0 + per < steps) ? s0 + per : steps;
        constexpr int SBB = 8;                          // (narrow output layer: eight u-steps a batch measured faster than sixteen)
        for (int s = s0; s < s1; s += SBB) {
            float a[SBB][4], b[SBB];
#pragma unroll
            for (int j = 0; j < SBB; ++j) {
                const int u = 4 * (s + j) + q;
                const bool ok = (s + j < s1) && (u < n_out);
                const int uu = ok ? u : 0;
                b[j] = Z[uu * L2A_AR + i16];
#pragma unroll
                for (int t = 0; t < 4; ++t) a[j][t] = wrow[t][uu];
                if (!ok) b[j] = 0.0f;
            }
            __builtin_amdgcn_sched_barrier(0);
            L2A_ATS(ph, 1)
#pragma unroll
            for (int j = 0; j < SBB; ++j)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = L2A_MFMA(kok[t] ? a[j][t] : 0.0f, b[j], acc[t]);
        }
    }
    L2A_ATS(ph, 2)
    float v[2];
    l2a_adapt_reduce2<false>(acc, red, us, lane, v);
    L2A_ATS(ph, 3)
    zout[0] = v[0] * l2a_act_grad_from_output(a_e.x, p.hidden_act);
    zout[1] = v[1] * l2a_act_grad_from_output(a_e.y, p.hidden_act);
    if (ke_ok) *reinterpret_cast<float2*>(sc + p.z_off[l] + ke * L2A_AR + r0) = make_float2(zout[0], zout[1]);
    L2A_ATS(ph, 4)
}

// Layer 0's update, columns k0 .. k0 + 63, by the workgroup that has just computed their dZ_1 rows (`zout`: the thread's two
// values): through LDS (the partial sums in `red` are done with), then thread (unit = tid & 63, part = tid >> 6) walks the
// input rows part, part + 8, ...
__device__ __forceinline__ void l2a_adapt_update0_cols(const L2AAdaptParams& p, const L2AAdaptDst& d, float* red, const int k0,
                                                       const float (&zout)[2]) {
    const int task = blockIdx.y;
    const int n1 = p.dims[1], k_in0 = p.dims[0];
    const float* sc = p.scratch + (long long)task * p.scratch_stride;
    const int kl = threadIdx.x >> 3, r0 = (threadIdx.x & 7) * 2;
    __syncthreads();                        // every thread has read its partial sums
    float* zs = red;                        // [64][16]
    zs[kl * L2A_AR + r0] = zout[0];
    zs[kl * L2A_AR + r0 + 1] = zout[1];
    __syncthreads();
    const int ul = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int u = k0 + ul;
    if (u >= n1) return;
    const float4* zp = reinterpret_cast<const float4*>(zs + ul * L2A_AR);
    const float4 z0 = zp[0], z1 = zp[1], z2 = zp[2], z3 = zp[3];
    float* dst = d.blk + (long long)task * d.set_stride;
    for (int k = part; k < k_in0; k += L2A_AW) {
        const float4* a = reinterpret_cast<const float4*>(sc + p.a_off[0] + k * L2A_AR);
        const float g = l2a_adapt_grad(a[0], a[1], a[2], a[3], z0, z1, z2, z3);
        l2a_adapt_put(p, d, dst, 0, k, u, p.w[0][(long long)k * n1 + u] - d.lr * g);
    }
    if (part == 0) dst[d.raw_b[0] + u] = p.b[0][u] - d.lr * l2a_adapt_bias_grad(z0, z1, z2, z3);
}

// Backward through layer l (1 <= l < L) and, beside it, the update of that layer - one launch, grid (S + ceil(B / 2), m):
//   workgroups x < S = ceil(dims[l] / 64):  dZ_l for 64 input units of layer l (l2a_adapt_bwd_body).  The workgroups of the LAST
//       backward launch (l = 1) go on to update their 64 columns of W_0 and b_0 from the dZ_1 they have just computed (a 49 x 64
//       block each) - no launch is left behind the backward pass.
//   workgroups x >= S:  two update blocks of layer l each (theta' = theta - lr A_l^T dZ_{l+1}: both operands were complete
//       before this launch started), on the CUs the 8 x m backward workgroups leave idle.
// Until round 5 the update of all layers was a launch of its own behind the backward pass (15 of 85 us).
__global__ void __launch_bounds__(64 * L2A_AW) l2a_adapt_bwdu_k(const L2AAdaptParams p, const L2AAdaptDst d, int l) {
    __shared__ float red[L2A_AW * 64 * (L2A_AR + 1)];
    const int S = (p.dims[l] + 63) / 64;
    const int task = blockIdx.y;
    if ((int)blockIdx.x >= S) {
        const int b = 2 * ((int)blockIdx.x - S) + (int)(threadIdx.x >> 8);
        if (b < l2a_adapt_update_blocks(p, l)) l2a_adapt_update_block(p, d, l, b, task, (int)(threadIdx.x & 255));
        return;
    }
    const int ph = 2 * p.n_layers - 1 - l;      // (timeline slot of this launch)
    L2A_ATS(ph, 0) L2A_ATS_REAL(ph, 6)
    const float* Z = p.scratch + (long long)task * p.scratch_stride + p.z_off[l + 1];
    float zout[2];
    l2a_adapt_bwd_body(p, l, red, Z, blockIdx.x * 64, ph, zout);
    if (l == 1) l2a_adapt_update0_cols(p, d, red, blockIdx.x * 64, zout);
    L2A_ATS_REAL(ph, 7)
}
