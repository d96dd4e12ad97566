// stationary.hip - what does the weight-stationary cluster form of the small recurrent plans cost?  (round 6)
//
// Today (csrc/l2a_micro.h, l2a_lstm_micro_k) every workgroup of the run_rebal.py default plan (LSTM 256, 5 envs x 500 candidates,
// h = 10) owns 12 candidates and streams the whole gate matrix (1.18 MB) from L2 every horizon step.  The alternative: a CLUSTER
// of 16 CUs holds the gate matrix once - CU k keeps the 64 gate columns of units 16 k .. 16 k + 15 in registers (72 VGPRs per
// wave: 18 k-groups of its 16 columns) - and runs ALL of the cluster's 160 candidates (10 tiles of 16) through them; per step
// the new hidden state h [160 candidates x 256 units] must then be ALL-GATHERED over the cluster's CUs: every lane publishes its
// value as an 8-byte {tag, value} granule (cdna_hip_programming.md G16 recipe R2, write-through sc1 store), and every CU reads
// and validates all 160 x 256 of them (327 KB of granules) before its next gate GEMM.
//
// Three kernels, each one horizon-step loop of the c6 shape on 256 CUs (16 clusters x 16 CUs x 4 waves):
//   gemm   the gate GEMM alone: weights stationary in registers, B fragments (the x | h inputs of a candidate tile) read from LDS,
//          cell arithmetic, no exchange                                   -> the matrix-bound floor of a step
//   xchg   the all-gather alone: publish 10 tiles x 256 lanes of granules, fetch + validate the cluster's 10 x 16 x 64 x 4 granules
//          (wave w takes k-groups w, w + 4, ..), a workgroup barrier, the next step's values depend on what was read
//   both   the real thing: per candidate tile fetch -> LDS (two tiles ahead) | barrier | 72 MFMAs per wave | cell | publish
// placement 0: the 16 CUs of a cluster on ONE XCD (workgroup id % 8 = XCD); 1: on all eight (two CUs each).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/stationary tools/probes/stationary.hip && tools/probes/stationary
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define SC1 16

constexpr int KG = 18;          // k-groups: 2 (26 inputs, padded) + 16 (256 hidden units)
constexpr int KGH = 16;         // k-groups that are hidden state = published by the cluster's CUs
constexpr int TILE_BYTES = KGH * 64 * 4 * 8;                // granules of one candidate tile: [k-group][lane][4] x 8 B = 32 KiB
constexpr int MAXC = 32;        // clusters at most (cluster size 8)
constexpr long long CLUSTER_BYTES = 10LL * TILE_BYTES;      // per step parity (at most 10 tiles per cluster)

struct Params {
    unsigned long long* gran;   // [parity 2][cluster][tile][k-group 16][lane 64][4] granules
    float* out;                 // [workgroup] something that depends on everything (keeps the compiler honest)
    unsigned long long* clk;    // [workgroup][2] s_memtime start / end
    unsigned int* fail;         // polls that gave up
    int steps, placement, mode; // mode: 0 gemm, 2 both
    unsigned int tag0;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* base, long long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// cell arithmetic of one (unit, candidate): the four gate pre-activations are the lane's four accumulator registers (fast
// transcendentals like the product's recurrent kernels)
__device__ __forceinline__ float sigm(float x) { return __frcp_rn(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) { return 2.0f * __frcp_rn(1.0f + __expf(-2.0f * x)) - 1.0f; }
__device__ __forceinline__ float cell(f32x4 g, float& c) {
    c = c * sigm(g.z + 1.0f) + sigm(g.x) * tanh_fast(g.y);
    return tanh_fast(c) * sigm(g.w);
}

// CS = CUs per cluster (16: every wave keeps ONE tile of 16 gate columns = 4 units x 4 gates, 72 VGPRs of weights, and the
// cluster runs 160 candidates; 8: TWO column tiles per wave, 144 VGPRs, 80 candidates - half the all-gather per CU).
// Every iteration of a step runs two independent accumulator chains (CS 16: two candidate tiles; CS 8: two column tiles).
template <int MODE, int CS>
__global__ void __launch_bounds__(256) stat_k(const Params p) {
    constexpr int CT = 16 / CS;             // column tiles per wave
    constexpr int NTL = 10 * CS / 16;       // candidate tiles per cluster
    constexpr int TB = 2 / CT;              // candidate tiles per iteration
    constexpr int NIT = NTL / TB;           // iterations per step (5)
    constexpr int NF = TB * (KGH / 4);      // fetches (k-group x tile) of one wave per iteration
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* bl = reinterpret_cast<f32x4*>(smem);                 // B fragments: [slot 2][tile TB][k-group 18][lane 64] f32x4
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int jc = lane & 15, qq = lane >> 4;
    int cluster, member;
    if (p.placement == 0) { const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3; cluster = xcd * (32 / CS) + idx / CS; member = idx % CS; }
    else { cluster = blockIdx.x / CS; member = blockIdx.x % CS; }
    unsigned long long t_start;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_start) : : "memory");

    f32x4 w[CT][KG];            // this wave's gate columns, all 18 k-groups: stationary in registers
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            const float s = 0.02f * (float)((lane * 7 + g * 13 + member * 3 + wave + 5 * c) % 17 - 8);
            w[c][g] = (f32x4){s, -s * 0.5f, s * 0.25f, 0.01f};
        }
    float cst[NTL][CT], hval[NTL][CT];
#pragma unroll
    for (int ct = 0; ct < NTL; ++ct)
#pragma unroll
        for (int c = 0; c < CT; ++c) { cst[ct][c] = 0.0f; hval[ct][c] = 0.01f * (float)(jc + ct + c); }
    // the input k-groups (x = [obs | act], 2 k-groups) are produced locally in the probe (the product: a second, smaller exchange)
    for (int i = tid; i < 2 * TB * KG * 64; i += 256) bl[i] = (f32x4){0.01f, 0.02f, -0.01f, 0.005f};
    __syncthreads();

    const __amdgpu_buffer_rsrc_t xrs = rsrc(p.gran, 2LL * MAXC * CLUSTER_BYTES);
    unsigned int spin_left = 1u << 20;

    for (int t = 0; t < p.steps; ++t) {
        const unsigned int tag = p.tag0 + (unsigned int)t + 1u;
        const long long base_w = ((long long)(t & 1) * MAXC + cluster) * CLUSTER_BYTES;             // written this step
        const long long base_r = ((long long)((t + 1) & 1) * MAXC + cluster) * CLUSTER_BYTES;       // written by step t - 1
        const bool live = (MODE != 0) && t > 0;
        u32x4 fa[NF], fb[NF];
        auto foff = [&](int f, int it) {    // fetch f of this wave in iteration `it`: tile it * TB + f / 4, k-group wave + 4 (f % 4)
            return (int)(base_r + (long long)(it * TB + f / 4) * TILE_BYTES + (((wave + 4 * (f % 4)) * 64 + lane) * 32));
        };
        auto issue = [&](int it) {          // all loads of the iteration in flight together
            if (!live) return;
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                fa[f] = __builtin_amdgcn_raw_buffer_load_b128(xrs, foff(f, it), 0, SC1);
                fb[f] = __builtin_amdgcn_raw_buffer_load_b128(xrs, foff(f, it) + 16, 0, SC1);
            }
        };
        auto commit = [&](int it) {         // validate (re-fetch what is not there yet), write the values to LDS slot it % 2
            if (!live) return;
            const unsigned int want = tag - 1u;
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                while (true) {
                    const bool ok = fa[f].x == want && fa[f].z == want && fb[f].x == want && fb[f].z == want;
                    if (__all(ok)) break;
                    if (spin_left == 0) { if (lane == 0) atomicAdd(p.fail, 1u); break; }
                    --spin_left;
                    __builtin_amdgcn_s_sleep(2);
                    fa[f] = __builtin_amdgcn_raw_buffer_load_b128(xrs, foff(f, it), 0, SC1);
                    fb[f] = __builtin_amdgcn_raw_buffer_load_b128(xrs, foff(f, it) + 16, 0, SC1);
                }
                bl[(((it & 1) * TB + f / 4) * KG + 2 + wave + 4 * (f % 4)) * 64 + lane] =
                    (f32x4){__uint_as_float(fa[f].y), __uint_as_float(fa[f].w), __uint_as_float(fb[f].y), __uint_as_float(fb[f].w)};
            }
        };
        auto publish = [&](int ct, int c, float h) {
            if (MODE == 0) return;
            // lane (jc, qq) of wave `wave` holds unit 16 (member CT + c) + 4 wave + qq of candidate jc: the B fragment of hidden
            // k-group member CT + c wants it in lane (jc, wave), element qq
            const int off = (int)(base_w + (long long)ct * TILE_BYTES + ((((member * CT + c) * 64 + (wave * 16 + jc)) * 4 + qq) * 8));
            typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
            u32x2 g; g.x = tag; g.y = __float_as_uint(h);
            __builtin_amdgcn_raw_buffer_store_b64(g, xrs, off, 0, SC1);
        };
        issue(0);
        commit(0);
        for (int it = 0; it < NIT; ++it) {
            if (it + 1 < NIT) issue(it + 1);
            __syncthreads();            // iteration it's fragments are in LDS (committed by all four waves)
            const f32x4* b = bl + (it & 1) * TB * KG * 64;
            f32x4 acc[2];
            acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
            f32x4 bv[2], bn[2];
#pragma unroll
            for (int x = 0; x < TB; ++x) bv[x] = b[(x * KG + 0) * 64 + lane];
#pragma unroll
            for (int g = 0; g < KG; ++g) {
                if (g + 1 < KG) {
#pragma unroll
                    for (int x = 0; x < TB; ++x) bn[x] = b[(x * KG + g + 1) * 64 + lane];
                }
#pragma unroll
                for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                    for (int a = 0; a < 2; ++a)     // chain a: (column tile, candidate tile) = CS 16: (0, a); CS 8: (a, 0)
                        acc[a] = MFMA(w[CT == 2 ? a : 0][g][ii], bv[TB == 2 ? a : 0][ii], acc[a]);
#pragma unroll
                for (int x = 0; x < TB; ++x) bv[x] = bn[x];
            }
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int ct = it * TB + (TB == 2 ? a : 0), c = (CT == 2 ? a : 0);
                float cc = cst[ct][c];
                const float h = cell(acc[a], cc);
                cst[ct][c] = cc;
                hval[ct][c] = h;
                publish(ct, c, h);
            }
            if (it + 1 < NIT) commit(it + 1);
        }
        __syncthreads();                // slot 0 is refilled by the next step's first commit
    }
    float s = 0.0f;
#pragma unroll
    for (int ct = 0; ct < NTL; ++ct)
#pragma unroll
        for (int c = 0; c < CT; ++c) s += hval[ct][c] + cst[ct][c];
    unsigned long long t_end;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_end) : : "memory");
    if (tid == 0) { p.clk[2 * blockIdx.x] = t_start; p.clk[2 * blockIdx.x + 1] = t_end; }
    if (lane == 0) atomicAdd(p.out + blockIdx.x, s);
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 10;
    Params p;
    const size_t gbytes = 2 * (size_t)MAXC * (size_t)CLUSTER_BYTES;
    (void)hipMalloc(&p.gran, gbytes); (void)hipMemset(p.gran, 0, gbytes);
    (void)hipMalloc(&p.out, 256 * 4); (void)hipMemset(p.out, 0, 256 * 4);
    (void)hipMalloc(&p.clk, 256 * 16); (void)hipMalloc(&p.fail, 4); (void)hipMemset(p.fail, 0, 4);
    p.steps = steps;
    unsigned int nonce = 0;
    const int smem = 2 * 2 * KG * 64 * 16;
    for (int cs : {16, 8})
        for (int placement = 0; placement < 2; ++placement)
            for (int mode : {0, 2}) {
                if (mode == 0 && placement == 1) continue;
                p.placement = placement; p.mode = mode;
                auto launch = [&]() {
                    p.tag0 = (++nonce) << 12;
                    if (cs == 16 && mode == 0) hipLaunchKernelGGL((stat_k<0, 16>), dim3(256), dim3(256), smem, 0, p);
                    else if (cs == 16) hipLaunchKernelGGL((stat_k<2, 16>), dim3(256), dim3(256), smem, 0, p);
                    else if (mode == 0) hipLaunchKernelGGL((stat_k<0, 8>), dim3(256), dim3(256), smem, 0, p);
                    else hipLaunchKernelGGL((stat_k<2, 8>), dim3(256), dim3(256), smem, 0, p);
                };
                for (int i = 0; i < 300; ++i) launch();           // clocks up
                (void)hipDeviceSynchronize();
                hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
                const int reps = 50;
                (void)hipEventRecord(e0);
                for (int i = 0; i < reps; ++i) launch();
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
                ms /= reps;
                std::vector<unsigned long long> clk(512);
                (void)hipMemcpy(clk.data(), p.clk, 512 * 8, hipMemcpyDeviceToHost);
                unsigned long long worst = 0;
                for (int i = 0; i < 256; ++i) if (clk[2 * i + 1] - clk[2 * i] > worst) worst = clk[2 * i + 1] - clk[2 * i];
                unsigned int fails = 0; (void)hipMemcpy(&fails, p.fail, 4, hipMemcpyDeviceToHost);
                hipError_t err = hipGetLastError();
                printf("{\"probe\": \"stationary cluster, LSTM 256 gate GEMM + cell at the ReBAL plan size (2560 candidate slots on 256 CUs)\", "
                       "\"cluster_cus\": %d, \"kernel\": \"%s\", \"placement\": \"%s\", \"steps\": %d, \"launch_ms\": %.4f, "
                       "\"clk_per_step_longest_wg\": %.0f, \"gave_up_polls\": %u, \"hip\": \"%s\"}\n",
                       cs, mode == 0 ? "gemm only (no exchange)" : "gemm + all-gather of h", placement == 0 ? "cluster on one XCD" : "cluster over eight XCDs",
                       steps, ms, (double)worst / steps, fails, hipGetErrorString(err));
                fflush(stdout);
            }
    return 0;
}
