// l2a_micro_pack.h - weight layouts of the micro-tile kernels (l2a_micro.h): chain order, packing kernels, sizes.
// Included by the API translation units (set_weights) and by l2a_micro.h.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/l2a.h"

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
#ifdef L2A_PACK_KERNELS     // compiled once per library: defined by the API units that launch them (l2a_api.hip, l2a_lstm_api.hip)
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
#endif
// Packed output layer: [k-group of 4 chain positions over all U][lane = obs dim][4]
#ifdef L2A_PACK_KERNELS     // compiled once per library: defined by the API units that launch them (l2a_api.hip, l2a_lstm_api.hip)
static __global__ void l2a_lstm_micro_pack_out_k(const float* __restrict__ wo, int U, int obs_dim, long long total, float* dst) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int e = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
    const int gi = (int)(idx >> 8);
    dst[idx] = (lane < obs_dim) ? wo[(long long)l2a_chain_k(4 * gi + e) * obs_dim + lane] : 0.0f;
}
#endif


// Generic recurrent cells (l2a_rnn_micro.h): one product of a layer, TF kernel w [kin + U, G U] (input rows, then recurrent rows;
// gate q in columns [q U, (q + 1) U)) -> [64-unit tile T][gate q][k-group gi][lane][4]: float [lane][e] of record (T, q, gi) =
// W[row][q U + 64 T + chain_k(lane)], row = the feature at chain position 4 gi + e of the input part (gi < 4 KGx, KGx =
// ceil(kin / 16) rounded up to EVEN - the operand ring of the products with one or two gate tiles is eight k-groups deep and
// runs in trips of eight; zero past kin) or of the recurrent part; + 1 KiB of zeros.
__host__ __device__ inline int l2a_rnn_micro_kgx(int kin) { return (((kin + 15) / 16) + 1) & ~1; }
__host__ __device__ inline long long l2a_rnn_micro_floats(int kin, int U, int G) {
    return (long long)(U / 64) * G * (4 * l2a_rnn_micro_kgx(kin) + U / 4) * 256 + 256;
}
#ifdef L2A_PACK_KERNELS     // compiled once per library: defined by the API units that launch them (l2a_api.hip, l2a_lstm_api.hip)
static __global__ void l2a_rnn_micro_pack_k(const float* __restrict__ w, int kin, int U, int G, long long total, float* dst) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int NX = 4 * l2a_rnn_micro_kgx(kin), NKG = NX + U / 4;
    const int e = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
    const long long rest = idx >> 8;
    const int gi = (int)(rest % NKG);
    const int tq = (int)(rest / NKG);
    const int q = tq % G, T = tq / G;
    if (T >= U / 64) { dst[idx] = 0.0f; return; }
    const int unit = 64 * T + l2a_chain_k(lane);
    int row;
    if (gi < NX) { const int k = l2a_chain_k(4 * gi + e); row = (k < kin) ? k : -1; }
    else row = kin + l2a_chain_k(4 * (gi - NX) + e);
    dst[idx] = (row >= 0) ? w[(long long)row * G * U + q * U + unit] : 0.0f;
}
#endif


// LDS row of a layer's h / c / r * h: U + 24 = 24 mod 32 floats - the four rows of a micro tile start 24 banks apart
__host__ __device__ constexpr int l2a_rnn_micro_row(int U) { return U + 24; }

__host__ __device__ inline int l2a_rnn_micro_bias(int cell_type, int U) {
    return (cell_type == L2A_CELL_LSTM ? 4 : (cell_type == L2A_CELL_GRU ? 3 : 1)) * U;
}
// LDS bytes: x rows [4 mtm][row] (the layers' row stride: every B read of the loop has an immediate offset; 16 KG0 <= 80 features in
// chain order, dump slots 96 .. 103) | per layer h [2][4 mtm][row], c or r * h [4 mtm][row] (not for BasicRNN) |
// output partials [4][mtm][64] f32x4 | constants; mtm = 3 or 4, the micro tiles of the launch's largest workgroup
__host__ __device__ inline long long l2a_rnn_micro_smem(int cell_type, int n_layers, int U, int KG0, int mtm) {
    const long long per_layer = (long long)(cell_type == L2A_CELL_RNN ? 2 : 3) * 4 * mtm * l2a_rnn_micro_row(U);
    return (4LL * mtm * l2a_rnn_micro_row(U) + n_layers * per_layer + 32 * KG0 + 192 + (long long)n_layers * l2a_rnn_micro_bias(cell_type, U)) * 4 +
           4 * mtm * 64 * 16;
}

// LDS bytes of the kernel (sized for MT = 3 whatever the workgroup runs; the host asks for at least half a CU's LDS, so
// that no two workgroups share a CU)
__host__ __device__ inline int l2a_lstm_micro_smem(int U, int KG0) {
    return (2 * 12 * l2a_micro_row(U) + 32 * KG0 + 192 + 4 * U) * 4 + 2 * 4 * 3 * 64 * 16;
}


// ------------------------------------------------------------------------------------------------------------------
// MLP rollout (l2a_mlp_micro_k): per weight set one array in WAVE-STREAM order.  Stream T (= 64-unit tile of the hidden
// layers; wave T / UW runs it, UW = H / 256 streams per wave) is the sequence of 1 KiB records [64 lanes][4] that the wave
// consumes for one (horizon step, set), in consumption order:
//   [layer 0: 4 KG0e records, KG0e = KG0 rounded up to even] [hidden layer 1 .. n_hidden - 1: H / 4 records each] [output layer: 16 records]
// so that the operand ring runs straight through the phase boundaries with nothing but a record counter (and, at the end of
// a set, the next set's base).  A record holds four consecutive CHAIN positions (l2a_chain_k) of the layer's K order:
//   layer l < n_hidden : float [lane][e] = W_l[k = chain_k(4 r + e)][unit = 64 T + chain_k(lane)]
//   output layer       : chain position p = 64 T + 4 r + e of the hidden units (the wave's own: its two chunks of the canonical
//                        reduce, l2a_mfma.h), float [lane = slot][e] = W_out[k = chain_k(p)][dim(slot)]; slot = obs dim, except
//                        with O4 (the 16-candidate kernel's 4x4x1 form of an obs tile with at most four live units, HalfCheetah:
//                        dims 16 .. 19 are summed per QUARTER of the hidden units - chain positions p with p mod 4 = q - and the
//                        quarters added afterwards): slot 16 + 4 q + i carries dim 16 + i for the positions of quarter q, zero
//                        for the others.
// ------------------------------------------------------------------------------------------------------------------
// (layer 0 padded to an EVEN number of 16-feature k-groups - zero records - so that every phase is a whole number of
// eight-record loop iterations: the operand ring is eight records deep)
__host__ __device__ inline int l2a_mlp_micro_kg0e(int KG0) { return (KG0 + 1) & ~1; }
__host__ __device__ inline int l2a_mlp_micro_nrec(int H, int KG0, int n_hidden) {
    return 4 * l2a_mlp_micro_kg0e(KG0) + (n_hidden - 1) * (H / 4) + 16;
}
__host__ __device__ inline long long l2a_mlp_micro_floats(int H, int KG0, int n_hidden) {
    return (long long)(H / 64) * l2a_mlp_micro_nrec(H, KG0, n_hidden) * 256;
}
// Where W_l[k][u] lives (l == n_hidden: the output layer, u = obs dim); every weight has exactly one place, the rest of the
// array stays zero (the model block is zeroed at creation).
__host__ __device__ inline long long l2a_mlp_micro_index(int H, int KG0, int n_hidden, int o4, int l, int k, int u) {
    const int nrec = l2a_mlp_micro_nrec(H, KG0, n_hidden);
    const int p = l2a_chain_k(k);
    int T, rec, lane;
    if (l < n_hidden) {
        T = u >> 6;
        lane = l2a_chain_k(u & 63);
        rec = (l == 0 ? 0 : 4 * l2a_mlp_micro_kg0e(KG0) + (l - 1) * (H / 4)) + (p >> 2);
    } else {
        T = p >> 6;
        rec = 4 * l2a_mlp_micro_kg0e(KG0) + (n_hidden - 1) * (H / 4) + ((p & 63) >> 2);
        lane = (o4 && u >= 16) ? 16 + 4 * (p & 3) + (u - 16) : u;
    }
    return ((((long long)T * nrec + rec) * 64 + lane) << 2) + (p & 3);
}
// one layer of `count` sets: grid (ceil(k_in n_out / 256), count); w = the layer's [k_in, n_out] kernel of the first set
#ifdef L2A_PACK_KERNELS     // compiled once per library: defined by the API units that launch them (l2a_api.hip, l2a_lstm_api.hip)
static __global__ void l2a_mlp_micro_pack_k(const float* __restrict__ w, long long w_stride, int k_in, int n_out, int l, int H, int KG0,
                                            int n_hidden, int o4, float* dst, long long dst_stride) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)k_in * n_out) return;
    const int k = (int)(idx / n_out), u = (int)(idx - (long long)k * n_out);
    dst[(long long)blockIdx.y * dst_stride + l2a_mlp_micro_index(H, KG0, n_hidden, o4, l, k, u)] = w[(long long)blockIdx.y * w_stride + idx];
}
#endif
// LDS bytes: activation rows [2][12][H + 88] | output partials [2][4 waves][3][64] f32x4 | inputs [sets][12][104] | constants per set
__host__ __device__ inline int l2a_mlp_micro_cst(int H, int KG0, int n_hidden) { return 32 * KG0 + 192 + n_hidden * H; }
__host__ __device__ inline int l2a_mlp_micro_smem(int H, int KG0, int n_hidden, int sets) {
    return (2 * 12 * l2a_micro_row(H) + sets * 12 * 104 + sets * l2a_mlp_micro_cst(H, KG0, n_hidden)) * 4 + 2 * 4 * 3 * 64 * 16;
}
