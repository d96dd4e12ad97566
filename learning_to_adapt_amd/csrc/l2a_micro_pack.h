// l2a_micro_pack.h - weight layouts of the micro-tile kernels (l2a_micro.h): chain order, packing kernels, sizes.
// Included by the API translation units (set_weights) and by l2a_micro.h.
#pragma once

#include <hip/hip_runtime.h>

// chain position <-> feature inside a 16-feature k-group (swaps the two 2-bit fields: its own inverse)
__host__ __device__ inline int l2a_chain_k(int p) { return (p & ~15) | ((p & 3) << 2) | ((p >> 2) & 3); }


#define L2A_MICRO_MAXKG0 5                      // input k-groups a row reserves room for (L2A_KG0MAX)
// LDS row of a candidate: [U activations | 16 KG0 inputs | pad]; U + 88 = 24 mod 32 floats: the four rows of a micro
// tile start 24 banks apart - conflict-free 16-byte reads (4 rows x 16 B, broadcast over the blocks) and writes
__host__ __device__ constexpr int l2a_micro_row(int U) { return U + 16 * L2A_MICRO_MAXKG0 + 8; }


// Packed gate matrix: [64-unit tile T][gate q][k-group gi][lane][4], a tile's k-groups in ITS OWN K order (l2a_lstm.h:
// the h k-groups of the half that holds the tile's units, the other half's, then the x k-groups), four chain positions a
// group; + 3 KiB of slack (the operand ring requests three groups past the end of the last tile).
__host__ __device__ inline long long l2a_lstm_micro_gate_floats(int U, int KG0) {
    return (long long)(U / 64) * 4 * (U / 4 + 4 * KG0) * 256 + 1024;
}
static __global__ void l2a_lstm_micro_pack_k(const float* __restrict__ wk, int in_dim, int U, int KG0, long long total, float* dst) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int NG = U / 4 + 4 * KG0, HG = U / 8;
    const int e = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
    const long long rest = idx >> 8;
    const int gi = (int)(rest % NG);
    const int tq = (int)(rest / NG);
    const int q = tq & 3, T = tq >> 2;
    if (T >= U / 64) { dst[idx] = 0.0f; return; }
    const int unit = 64 * T + l2a_chain_k(lane);
    const int own = (64 * T) / (U / 2);
    int k_tf;
    if (gi < HG) k_tf = in_dim + own * (U / 2) + l2a_chain_k(4 * gi + e);
    else if (gi < 2 * HG) k_tf = in_dim + (1 - own) * (U / 2) + l2a_chain_k(4 * (gi - HG) + e);
    else { const int k = l2a_chain_k(4 * (gi - 2 * HG) + e); k_tf = (k < in_dim) ? k : -1; }
    dst[idx] = (k_tf >= 0) ? wk[(long long)k_tf * 4 * U + q * U + unit] : 0.0f;
}
// Packed output layer: [k-group of 4 chain positions over all U][lane = obs dim][4]
static __global__ void l2a_lstm_micro_pack_out_k(const float* __restrict__ wo, int U, int obs_dim, long long total, float* dst) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int e = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
    const int gi = (int)(idx >> 8);
    dst[idx] = (lane < obs_dim) ? wo[(long long)l2a_chain_k(4 * gi + e) * obs_dim + lane] : 0.0f;
}


// LDS bytes of the kernel (sized for MT = 3 whatever the workgroup runs; the host asks for at least half a CU's LDS, so
// that no two workgroups share a CU)
__host__ __device__ inline int l2a_lstm_micro_smem(int U, int KG0) {
    return (2 * 12 * l2a_micro_row(U) + 32 * KG0 + 192 + 4 * U) * 4 + 2 * 4 * 3 * 64 * 16;
}

