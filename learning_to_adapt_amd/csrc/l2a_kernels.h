// l2a_kernels.h - device side of libl2a_hip.so (gfx950 / CDNA4 only).
//
// One launch = one MPC plan step: every candidate action sequence is rolled through the
// learned MLP dynamics over the whole planning horizon, the closed-form reward is
// accumulated and the per-env arg-max is reduced - i.e. the loop body of
// `MPCController.get_rs_action` (reference policies/mpc_controller.py:116-129) with
// `dynamics_model.predict` (dynamics/mlp_dynamics.py:204-222) and `env.reward`
// (envs/half_cheetah_env.py:58-65 etc.) fused into a single kernel.  No intermediate
// trajectory ever reaches HBM.
//
// Two kernels:
//  * l2a_rollout_mfma_k  - fp32 MFMA (v_mfma_f32_16x16x4_f32) path for hidden widths
//                          128/256/512.  See DESIGN.md section 3 for the tiling.
//  * l2a_rollout_valu_k  - generic fp32 VALU path (any layer sizes); baseline + fallback.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/l2a.h"

#define L2A_MAX_LAYERS 9   // hidden layers <= 8, plus the output layer
#define L2A_OTMAX 4        // MFMA path: obs_dim <= 64
#define L2A_KG0MAX 5       // MFMA path: obs_dim + act_dim <= 80

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Cache policy of the tile splits' exchange granules (aux operand of the raw buffer builtins on gfx950: 1 = sc0, 2 = nt,
// 16 = sc1): sc1 alone = write-through stores / miss-always loads at device scope.  (sc1 + nt cost 1.5 %, sc0 + sc1 changed
// nothing: profiles/r03_ab_kernel_variants.jsonl.)
#define L2A_SC1 16

// Everything a rollout launch needs, passed by value (kernarg segment).
struct L2AKParams {
    // ---- model (constant between launches) -------------------------------------------
    const float* wblk;          // base of the weight-set blocks
    long long set_stride;       // floats per weight set
    long long raw_w[L2A_MAX_LAYERS];  // offsets (floats) of the row-major [in,out] kernels
    long long raw_b[L2A_MAX_LAYERS];  // offsets of the biases
    long long pk_w0;            // packed (MFMA fragment order) layer 0
    long long pk_wmid;          // packed hidden->hidden layers, pk_wmid_stride apart
    long long pk_wmid_stride;
    long long pk_wout;          // packed output layer
    long long pk_bout;          // output bias padded to OT*16
    long long nm_off;           // [in_mu KG0*16][in_inv KG0*16][out_mu OT*16][out_sd OT*16]
    int hidden[L2A_MAX_LAYERS];
    int n_hidden;
    int obs_dim, act_dim, in_dim;
    int hidden_act, output_act;
    int mode, n_sets;
    int KG0, OT;                // ceil(in_dim/16), ceil(obs_dim/16)
    int sa_elems;               // MFMA: f32x4 elements per LDS activation region
    int cst_set;                // MFMA: floats of per-set constants cached in LDS (norm + biases)
    int lb;                     // MFMA: sets per batch (layer 0 of lb sets, one barrier, their GEMMs, one barrier, reduces)
    int cst_off;                // MFMA: start of the constants in LDS, in f32x4 elements from the base
    int n_cst;                  // MFMA: constant slots laid out (>= the sets any workgroup of the launch runs)
    float hid_floor, out_floor; // MFMA fast activations: relu = max(x, 0), identity = max(x, -inf)
    // micro-tile kernel (l2a_micro.h): its copy of a set's weights in wave-stream order (l2a_micro_pack.h), records per stream,
    // whether dims 16 .. 19 are summed per quarter of the hidden units like the 16-candidate O4 instance, bytes the stream
    // descriptor spans (all sets), and the launch geometry: workgroups per env, how many of them take mc_hi micro tiles
    long long pk_m;
    int m_nrec, m_o4;
    long long m_bytes;
    int mc_w, mc_r, mc_hi;
    int hmax;                   // VALU: widest hidden layer
    // ---- launch ------------------------------------------------------------------------
    const float* obs0;          // [m, obs_dim] (or [R, obs_dim] when obs_per_row)
    const float* actions;       // [h, m*n, act_dim]
    float* returns_out;         // [m, n] or null
    unsigned long long* best_key;  // [m] or null
    float* state_out;           // [m*n, obs_dim] or null (final state; used by predict and chunked plans)
    const float* ret_in;        // [m, n] returns accumulated by earlier horizon chunks, or null (= 0)
    double disc0;               // discount ** (first horizon step of this launch); 1.0 for a whole plan
    int obs_per_row;
    int m, n, h;
    int tiles_per_env;
    int c_lo, c_hi;             // MFMA: this launch covers candidates [c_lo, c_hi) of every env ([0, n) unless a plan is cut in two:
                                // double rounds on the whole-tile instances in front, the rest behind - l2a_api.hip); rows and
                                // results stay indexed by the plan's n
    int done_total;             // MFMA: workgroup-tiles of ALL launches of the plan (0 = this launch's own): the mailbox count
    int cand_offset;
    double discount;            // float64 like the reference's `self.discount ** t` (:126)
    l2a_reward rw;
    // ---- member split (MFMA kernel, mean mode): two workgroups share one candidate tile ----
    int split;                  // 0: one workgroup runs all members; 1: group A | group B; 2: + shared set;
                                // 3: member fan - n_sets workgroups per tile, one set each (FAN instances only)
    // XCD placement of a single-round launch (0 units = the contiguous remap of l2a_logical_wg).  A "unit" is a set of
    // workgroups that stream the same weights: the two ensemble groups of a split mean-mode launch, the environments of a
    // per-block launch.  The first pl_r units own pl_f + 1 XCDs each, the others pl_f; the grid is 8 ceil(pl_w / pl_f)
    // workgroups and hardware workgroup i (XCD i % 8, slot i / 8) either finds its place in its XCD's unit or returns.
    int pl_units, pl_f, pl_r, pl_w;     // units, XCDs per unit (floor), units with one XCD more, workgroups per unit
    int split_from;             // -1: `split` applies to every tile; >= 0: tail split - hardware workgroups
                                // [0, split_from) run whole tiles, the rest are pairs sharing tiles split_from ..
    unsigned int xtag;          // per-launch tag base (launch nonce << 12); tag = xtag + t + 1
    unsigned long long* xbuf;   // exchange granules [pair][group][slot][region][NT*OT*2][64] (fan: [pair][member][slot][NT*OT*2][64])
    unsigned int* status;       // host-visible word; bit 0 set = exchange timed out
    unsigned int spin_limit;    // polls one workgroup may spend waiting for its partner, per launch
    unsigned long long* dbg;    // optional phase timeline (tools/timeline.py); null in production
    // ---- result mailbox (l2a_plan_rs_sync): the last candidate tile of the launch publishes the keys to
    //      host-mapped memory, so the host learns the result without a copy or a stream synchronisation ----
    unsigned int* done_ctr;             // device: tiles that have contributed their key (reset by the last one)
    unsigned long long* mail_keys;      // host-mapped [m]
    unsigned long long* mail_seq_ptr;   // host-mapped; null = no mailbox
    unsigned long long mail_seq;        // value to publish
    unsigned long long* next_keys;      // device key slot of the NEXT launch: zeroed here (saves its memset)
};

// ------------------------------------------------------------------------------------------
// weight packing: MFMA fragment order
//
// A layer y[u] = sum_k x[k] * W[k][u] is computed transposed on the matrix core:
//   D[i][j] = sum_kk A[i][kk] * B[kk][j]   (v_mfma_f32_16x16x4_f32; i = output unit within a
//   16-unit tile, j = candidate within a 16-candidate tile, kk = 0..3)
// Lane l of a wave supplies A[i = l & 15][kk = l >> 4].  Four consecutive MFMAs (a "k-group"
// of 16 input features) take their A values from one 16-byte load per lane:
//   packed[((c * KG + g) * 64 + l) * 4 + ii] = W[16 g + 4 (l >> 4) + ii][16 c + (l & 15)]
// (c = output tile, g = k-group, ii = which of the 4 MFMAs; zero outside the real matrix).
// The k order inside a group is permuted (feature 16g + 4 kk + ii goes to MFMA ii, slot kk)
// so that the D fragment of one layer (lane (j, qq) holds units 16c + 4qq + 0..3) is,
// register for register, the B fragment of the next layer's k-group g = c.
// ------------------------------------------------------------------------------------------
__host__ __device__ inline void l2a_pack_decode(long long idx, int KG, int* k, int* u) {
    const int ii = (int)(idx & 3);
    const int lane = (int)((idx >> 2) & 63);
    const long long rest = idx >> 8;
    const int g = (int)(rest % KG);
    const int c = (int)(rest / KG);
    *k = 16 * g + 4 * (lane >> 4) + ii;
    *u = 16 * c + (lane & 15);
}

// ------------------------------------------------------------------------------------------
// shared device helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float l2a_act1(float x, int kind) {
    switch (kind) {
        case L2A_ACT_RELU: return fmaxf(x, 0.0f);
        case L2A_ACT_TANH: return tanhf(x);
        case L2A_ACT_SIGMOID: return 1.0f / (1.0f + expf(-x));
        case L2A_ACT_SWISH: return x / (1.0f + expf(-x));
        default: return x;
    }
}

__device__ __forceinline__ f32x4 l2a_act4(f32x4 v, int kind) {
    if (kind == L2A_ACT_RELU) {
        v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f);
        v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f);
        return v;
    }
    if (kind == L2A_ACT_IDENTITY) return v;
    v.x = l2a_act1(v.x, kind); v.y = l2a_act1(v.y, kind);
    v.z = l2a_act1(v.z, kind); v.w = l2a_act1(v.w, kind);
    return v;
}

// (orderable_u32(ret) << 31) | (0x7fffffff - index); NaN sorts above +inf like np.argmax.
__host__ __device__ inline unsigned long long l2a_key_pack(float ret, int index) {
    unsigned int u;
#if defined(__HIP_DEVICE_COMPILE__)
    u = __float_as_uint(ret);
#else
    union { float f; unsigned int u; } cv; cv.f = ret; u = cv.u;
#endif
    unsigned int ord;
    if (ret != ret) ord = 0xffffffffu;
    else ord = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)ord << 31) | (unsigned long long)(0x7fffffffu - (unsigned int)index);
}

// XCD-aware, bijective remap of the hardware workgroup id: workgroups are observed to land on
// XCD (id % 8); give each XCD a contiguous range of logical ids so that workgroups sharing a
// weight set (per-block mode) share one XCD's L2.  Speed only, never correctness.
__device__ __forceinline__ int l2a_logical_wg(int hw, int nwg) {
    const int xcd = hw & 7, idx = hw >> 3;
    const int qn = nwg >> 3, rn = nwg & 7;
    const int start = (xcd < rn) ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn;
    return start + idx;
}

#define L2A_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// Called by ONE thread per candidate tile after its atomicMax into best_key.  The tile that arrives last copies
// the finished keys into the host-mapped mailbox, zeroes the next launch's key slot, rearms the counter and
// finally publishes the launch's sequence number (system-scope release): the host polls that word.
template <class Params>
__device__ __forceinline__ void l2a_publish_result(const Params& p, int n_tiles) {
    if (!p.mail_seq_ptr) return;
    __threadfence();
    const unsigned int prev = __hip_atomic_fetch_add(p.done_ctr, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (prev + 1u != (unsigned int)n_tiles) return;
    for (int i = 0; i < p.m; ++i) {
        const unsigned long long k = __hip_atomic_load(p.best_key + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p.mail_keys + i, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(p.next_keys + i, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __hip_atomic_store(p.done_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence_system();
    __hip_atomic_store(p.mail_seq_ptr, p.mail_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

