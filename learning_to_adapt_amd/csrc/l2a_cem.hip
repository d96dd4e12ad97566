// l2a_cem.hip - the cross-entropy-method planner's per-iteration work around the fused rollout, on the device
// (include/l2a.h: l2a_cem_sample, l2a_cem_refit).  gfx950 only.
//
// Reference: `MPCController.get_cem_action` (policies/mpc_controller.py:71-106).  Per iteration the reference draws
// n * m * h * act_dim standard normals on the host (:85), forms a = mean + z * std (:86), clips (:87), rolls the
// UNCLIPPED samples out reading its candidate-major rows as if they were env-major (:92-96), and refits mean / std
// to the "elites" its rank mask `(-returns).argsort() < num_elites` selects (:101-104): boolean POSITIONS p for which
// the candidate ranked p-th has an index below num_elites, pooled over the envs.  `reference` = 1 keeps those
// semantics; 0 = the fixed reading (clipped rollouts, env-major rows, true top-k per env).
//
//   l2a_cem_sample_k : one thread per sample element: z (given, or Philox4x32-10 + Box-Muller from (seed, offset)),
//                      a = mean + z * std, clip, both stored [n, m, D]; the rollout's candidate tensor
//                      [h, m * n_local, act_dim] is written in the same pass (transposed, this rank's shard only)
//   l2a_cem_rank_k   : elite rows WITHOUT a sort (one wave per rank: ballot + popcount over the env's returns in LDS).  reference: position p is an elite of env i iff p = rank_i(j) for one of
//                      the first k candidates j - k rank computations of n comparisons each; fixed: candidate c is
//                      an elite iff rank_i(c) < k - n of them.  Stable descending order (ties: lower index first), like
//                      the stable argsort it replaces
//   l2a_cem_stats_k  : mean / biased std of the elite rows per dimension (two passes, 8 row slices per dimension), pooled
//                      over the envs (reference) or per env, and the update mean = alpha * mean + (1 - alpha) * elite_mean
//
// Three launches per iteration instead of the ~20 stock tensor-library launches they replace (randn, mul, add, clamp,
// permute / contiguous, argsort, compare, transpose, masked sums, topk / gather, mean, std).

#include "l2a_host.h"

#include <cstring>
#include <string>

namespace {

// ---- Philox4x32-10 (Salmon et al., SC'11): counter-based, so every element's normal is a pure function of
//      (seed, offset + element index) - all ranks of a sharded plan generate the same numbers without talking ----------
__device__ __forceinline__ void philox_round(unsigned int (&c)[4], unsigned int k0, unsigned int k1) {
    const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
    const unsigned int hi0 = (unsigned int)(p0 >> 32), lo0 = (unsigned int)p0;
    const unsigned int hi1 = (unsigned int)(p1 >> 32), lo1 = (unsigned int)p1;
    c[0] = hi1 ^ c[1] ^ k0; c[1] = lo1; c[2] = hi0 ^ c[3] ^ k1; c[3] = lo0;
}

__device__ __forceinline__ float philox_normal(unsigned long long seed, unsigned long long index) {
    unsigned int c[4] = {(unsigned int)index, (unsigned int)(index >> 32), 0x4c32614du, 0u};
    unsigned int k0 = (unsigned int)seed, k1 = (unsigned int)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    // Box-Muller on two uniforms in (0, 1]
    const float u1 = ((float)(c[0] >> 8) + 1.0f) * (1.0f / 16777216.0f);
    const float u2 = (float)(c[1] >> 8) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

struct CemSampleParams {
    const float* z;             // [n, m, D] or null (generate)
    unsigned long long seed, offset;
    const float* mean;          // [m, D]
    const float* std;
    const float* low;           // [act_dim]
    const float* high;
    int n, m, h, act_dim, reference, lo, hi;
    float* a_clip;              // [n, m, D]
    float* a_raw;               // [n, m, D] or null
    float* seq;                 // [h, m * (hi - lo), act_dim] or null
};

__global__ void __launch_bounds__(256) l2a_cem_sample_k(const CemSampleParams p) {
    const int D = p.h * p.act_dim;
    const long long total = (long long)p.n * p.m * D;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const long long g = e / D;                  // flat sample row j * m + i (the draw's [n, m, D] order, :85)
    const int d = (int)(e - g * D);
    const int i = (int)(g % p.m);
    const float z = p.z ? p.z[e] : philox_normal(p.seed, p.offset + (unsigned long long)e);
    const float a = p.mean[i * D + d] + z * p.std[i * D + d];                      // :86
    const int k = d % p.act_dim, t = d / p.act_dim;
    const float c = fminf(fmaxf(a, p.low[k]), p.high[k]);                           // :87
    p.a_clip[e] = c;
    if (p.a_raw) p.a_raw[e] = a;
    if (p.seq) {
        // plan row of this sample: the reference feeds its candidate-major rows as if they were env-major (:92-96)
        const long long prow = p.reference ? g : (long long)i * p.n + g / p.m;
        const long long cand = prow % p.n, blk = prow / p.n;
        if (cand >= p.lo && cand < p.hi) {
            const long long nsel = p.hi - p.lo;
            p.seq[((long long)t * (p.m * nsel) + blk * nsel + (cand - p.lo)) * p.act_dim + k] = p.reference ? a : c;
        }
    }
}

// Stable descending rank of candidate j among env i's returns = the number of candidates that sort before it.  One WAVE
// per j: lane l looks at candidates l, l + 64, ...; a ballot + popcount per 64 candidates (a thread-per-j loop over LDS was
// one exposed LDS round trip per candidate: 106 us for 400 ranks of 4000).
__device__ __forceinline__ int cem_rank(const float* r, int n, int j, int lane) {
    const float v = r[j];
    int rank = 0;
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int c = c0 + lane;
        const float w = (c < n) ? r[c] : -__builtin_inff();
        const bool before = (c < n) && (w > v || (w == v && c < j));
        rank += __popcll(__ballot(before));
    }
    return rank;
}

// elite_rows[i * k + q]: flat sample rows (into [n * m, D]) of env i's elites.  grid (blocks, m); the waves of an env's
// workgroups take the j's round robin (j < k in the reference's reading, j < n in the fixed one); the env's returns are
// staged in LDS.
__global__ void __launch_bounds__(256) l2a_cem_rank_k(const float* returns, int n, int m, int k, int reference,
                                                      int* elite_rows) {
    extern __shared__ float rs[];
    const int i = blockIdx.y;
    for (int c = threadIdx.x; c < n; c += 256) {
        const float x = returns[(long long)i * n + c];
        rs[c] = (x != x) ? -__builtin_inff() : x;      // NaN returns sort last (np.argsort of -returns puts them there too);
    }                                                  // with ties broken by index the ranks stay a permutation
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
    const int work = reference ? k : n;
    for (int j = wave; j < work; j += n_waves) {
        const int rank = cem_rank(rs, n, j, lane);
        if (lane != 0) continue;
        if (reference) elite_rows[i * k + j] = rank * m + i;         // position rank_i(j) of env i is an elite (:101)
        else if (rank < k) elite_rows[i * k + rank] = j * m + i;      // candidate j is among env i's top k
    }
}

// Elite statistics per dimension: group = all envs pooled (reference; the result is broadcast to every env, :103-104) or
// one env (fixed).  Workgroup = 32 dimensions x 8 row slices: the elite row ids are staged in LDS, every thread sums its
// slice of the rows (independent loads, 4 in flight), the slices meet in LDS in a fixed order; two passes (mean, then
// the squared deviations).  grid (ceil(D / 32), groups).
#define L2A_CEM_MAXROWS 8192
__global__ void __launch_bounds__(256) l2a_cem_stats_k(const float* a_clip, const int* elite_rows, int m, int D, int k,
                                                       int reference, float alpha, float* mean, float* std) {
    __shared__ int rows[L2A_CEM_MAXROWS];
    __shared__ float part[8][33];
    const int dl = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int d = blockIdx.x * 32 + dl;
    const int grp = blockIdx.y;
    const int cnt = reference ? m * k : k;
    const int* src = reference ? elite_rows : elite_rows + grp * k;
    for (int q = threadIdx.x; q < cnt; q += 256) rows[q] = src[q];
    __syncthreads();
    const bool live = d < D;
    const float* col = a_clip + (live ? d : 0);
    float mu = 0.0f, sd = 0.0f;
    for (int pass = 0; pass < 2; ++pass) {
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
        int q = sl;
        for (; q + 24 < cnt; q += 32) {
            const float x0 = col[(long long)rows[q] * D], x1 = col[(long long)rows[q + 8] * D];
            const float x2 = col[(long long)rows[q + 16] * D], x3 = col[(long long)rows[q + 24] * D];
            if (pass == 0) { s0 += x0; s1 += x1; s2 += x2; s3 += x3; }
            else { s0 = fmaf(x0 - mu, x0 - mu, s0); s1 = fmaf(x1 - mu, x1 - mu, s1);
                   s2 = fmaf(x2 - mu, x2 - mu, s2); s3 = fmaf(x3 - mu, x3 - mu, s3); }
        }
        for (; q < cnt; q += 8) {
            const float x0 = col[(long long)rows[q] * D];
            if (pass == 0) s0 += x0; else s0 = fmaf(x0 - mu, x0 - mu, s0);
        }
        part[sl][dl] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        float tot = 0.0f;
#pragma unroll
        for (int w = 0; w < 8; ++w) tot += part[w][dl];
        __syncthreads();
        if (pass == 0) mu = tot / (float)cnt; else sd = sqrtf(tot / (float)cnt);
    }
    if (!live || sl != 0) return;
    if (reference) {
        for (int i = 0; i < m; ++i) {
            mean[i * D + d] = mean[i * D + d] * alpha + (1.0f - alpha) * mu;
            std[i * D + d] = sd;
        }
    } else {
        mean[grp * D + d] = mean[grp * D + d] * alpha + (1.0f - alpha) * mu;
        std[grp * D + d] = sd;
    }
}

// np.argmax's order (the reference's `np.argmax(returns, axis=1)`, mpc_controller.py:128-129): a NaN is the maximum, the first
// NaN wins; otherwise the largest value, ties to the lowest index.  A diverged rollout therefore surfaces as a NaN best return
// exactly as it does in the reference, instead of a finite-looking action.
__device__ __forceinline__ bool l2a_argmax_better(float x, int c, float best, int idx) {
    const bool xn = x != x, bn = best != best;
    if (xn != bn) return xn;
    if (xn) return c < idx;
    return x > best || (x == best && c < idx);
}

// The plan's result in ONE buffer (one read-back instead of five launches and five copies): per env the arg-max of its
// returns (np.argmax's order, NaN included), the first action of that candidate as the rollout saw it, its return; behind them
// the final mean / std.
// out: [m][act_dim + 2] floats (action | return | index as the bit pattern of an int32), then mean [m, D], std [m, D].
// grid (m + ceil(2 m D / 256)): block i < m is env i.
__global__ void __launch_bounds__(256) l2a_cem_pick_k(const float* returns, const float* cand, const float* mean, const float* std,
                                                      int n, int m, int D, int act_dim, int reference, float* out) {
    const int W = act_dim + 2;
    if ((int)blockIdx.x >= m) {
        const long long e = (long long)(blockIdx.x - m) * 256 + threadIdx.x;
        const long long md = (long long)m * D;
        if (e < 2 * md) out[(long long)m * W + e] = (e < md) ? mean[e] : std[e - md];
        return;
    }
    const int i = blockIdx.x;
    __shared__ float bv[256];
    __shared__ int bi[256];
    float best = -__builtin_inff();
    int idx = 0x7fffffff;
    for (int c = threadIdx.x; c < n; c += 256) {
        const float x = returns[(long long)i * n + c];
        if (l2a_argmax_better(x, c, best, idx)) { best = x; idx = c; }
    }
    bv[threadIdx.x] = best; bi[threadIdx.x] = idx;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) {
            const float x = bv[threadIdx.x + s]; const int c = bi[threadIdx.x + s];
            if (l2a_argmax_better(x, c, bv[threadIdx.x], bi[threadIdx.x])) { bv[threadIdx.x] = x; bi[threadIdx.x] = c; }
        }
        __syncthreads();
    }
    int j = bi[0];
    if (j < 0 || j >= n) j = 0;                         // (all returns NaN / -inf)
    // reference: the sample memory read as [m, n, D] (:92-96), row i * n + j; fixed: candidate j of env i is sample row j * m + i
    const long long row = reference ? (long long)i * n + j : (long long)j * m + i;
    if ((int)threadIdx.x < act_dim) out[i * W + threadIdx.x] = cand[row * D + threadIdx.x];
    if (threadIdx.x == 0) { out[i * W + act_dim] = returns[(long long)i * n + j]; out[i * W + act_dim + 1] = __int_as_float(j); }
}

}  // namespace

extern "C" {

int l2a_cem_sample(l2a_ctx* ctx, const float* z, unsigned long long seed, unsigned long long offset, const float* mean,
                   const float* std, const float* low, const float* high, int n, int m, int h, int act_dim,
                   int reference, int lo, int hi, float* a_clip, float* a_raw, float* seq, void* stream_v) {
    if (!ctx) return L2A_EINVAL;
    if (!mean || !std || !low || !high || !a_clip) return l2a_fail(ctx, L2A_EINVAL, "l2a_cem_sample: null pointer");
    if (n < 1 || m < 1 || h < 1 || act_dim < 1 || lo < 0 || hi < lo || hi > n)
        return l2a_fail(ctx, L2A_EINVAL, "l2a_cem_sample: bad n / m / h / act_dim / shard");
    const long long total = (long long)n * m * h * act_dim;
    if (total > 0x7fffffffLL * 256) return l2a_fail(ctx, L2A_EINVAL, "l2a_cem_sample: too many samples");
    l2a_device_guard guard(ctx->device);
    CemSampleParams p;
    std::memset(&p, 0, sizeof(p));
    p.z = z; p.seed = seed; p.offset = offset; p.mean = mean; p.std = std; p.low = low; p.high = high;
    p.n = n; p.m = m; p.h = h; p.act_dim = act_dim; p.reference = reference ? 1 : 0; p.lo = lo; p.hi = hi;
    p.a_clip = a_clip; p.a_raw = a_raw; p.seq = (hi > lo) ? seq : nullptr;
    hipLaunchKernelGGL(l2a_cem_sample_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream_v), p);
    L2A_HIP(ctx, hipGetLastError());
    return L2A_OK;
}

int l2a_cem_refit(l2a_ctx* ctx, const float* returns, const float* a_clip, int n, int m, int D, int num_elites,
                  int reference, float alpha, int* elite_rows, float* mean, float* std, void* stream_v) {
    if (!ctx) return L2A_EINVAL;
    if (!returns || !a_clip || !elite_rows || !mean || !std) return l2a_fail(ctx, L2A_EINVAL, "l2a_cem_refit: null pointer");
    if (n < 1 || m < 1 || D < 1 || num_elites < 1 || num_elites > n)
        return l2a_fail(ctx, L2A_EINVAL, "l2a_cem_refit: bad n / m / D / num_elites");
    if ((long long)(reference ? m : 1) * num_elites > L2A_CEM_MAXROWS)
        return l2a_fail(ctx, L2A_EINVAL, "l2a_cem_refit: more than 8192 elite rows per statistics group");
    if ((size_t)n * sizeof(float) > (size_t)ctx->lds_per_block)
        return l2a_fail(ctx, L2A_EINVAL, "l2a_cem_refit: more candidates than an env's returns fit in LDS");
    l2a_device_guard guard(ctx->device);
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    const int smem = n * (int)sizeof(float);
    if (smem > 48 * 1024)
        L2A_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(l2a_cem_rank_k),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    const int work = reference ? num_elites : n;
    const int blocks = (work + 3) / 4 < 256 ? (work + 3) / 4 : 256;      // one wave per rank, at most 1024 waves per env
    hipLaunchKernelGGL(l2a_cem_rank_k, dim3((unsigned)blocks, (unsigned)m), dim3(256), smem, stream, returns,
                       n, m, num_elites, reference ? 1 : 0, elite_rows);
    hipLaunchKernelGGL(l2a_cem_stats_k, dim3((unsigned)((D + 31) / 32), (unsigned)(reference ? 1 : m)), dim3(256), 0,
                       stream, a_clip, elite_rows, m, D, num_elites, reference ? 1 : 0, alpha, mean, std);
    L2A_HIP(ctx, hipGetLastError());
    return L2A_OK;
}

int l2a_cem_pick(l2a_ctx* ctx, const float* returns, const float* cand, const float* mean, const float* std, int n, int m, int D,
                 int act_dim, int reference, float* out, void* stream_v) {
    if (!ctx) return L2A_EINVAL;
    if (!returns || !cand || !mean || !std || !out) return l2a_fail(ctx, L2A_EINVAL, "l2a_cem_pick: null pointer");
    if (n < 1 || m < 1 || D < 1 || act_dim < 1 || act_dim > D || act_dim > 256)
        return l2a_fail(ctx, L2A_EINVAL, "l2a_cem_pick: bad n / m / D / act_dim");
    l2a_device_guard guard(ctx->device);
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    const long long copy_blocks = (2LL * m * D + 255) / 256;
    hipLaunchKernelGGL(l2a_cem_pick_k, dim3((unsigned)(m + copy_blocks)), dim3(256), 0, stream, returns, cand, mean, std, n, m, D,
                       act_dim, reference ? 1 : 0, out);
    L2A_HIP(ctx, hipGetLastError());
    return L2A_OK;
}

}  // extern "C"
