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

constexpr int CS = 16;          // CUs per cluster
constexpr int NT = 10;          // candidate tiles per cluster (160 candidates)
constexpr int KG = 18;          // k-groups: 2 (26 inputs, padded) + 16 (256 hidden units)
constexpr int KGH = 16;         // k-groups that are hidden state = published by the cluster's CUs (k-group 2 + k <- CU k)
constexpr int TILE_BYTES = KGH * 64 * 4 * 8;                // granules of one candidate tile: [k-group][lane][4] x 8 B = 32 KiB
constexpr long long CLUSTER_BYTES = (long long)NT * TILE_BYTES;    // per step parity

struct Params {
    unsigned long long* gran;   // [parity 2][cluster][tile][k-group 16][lane 64][4] granules
    float* out;                 // [workgroup] something that depends on everything (keeps the compiler honest)
    unsigned long long* clk;    // [workgroup][2] s_memtime start / end
    unsigned int* fail;         // polls that gave up
    int steps, placement, mode; // mode: 0 gemm, 1 xchg, 2 both
    unsigned int tag0;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* base, long long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// cell arithmetic of one (unit, candidate): the four gate pre-activations are the lane's four accumulator registers
__device__ __forceinline__ float cell(f32x4 g, float& c) {
    const float i = 1.0f / (1.0f + __expf(-g.x)), j = tanhf(g.y), f = 1.0f / (1.0f + __expf(-(g.z + 1.0f))), o = 1.0f / (1.0f + __expf(-g.w));
    c = c * f + i * j;
    return tanhf(c) * o;
}

template <int MODE>
__global__ void __launch_bounds__(256) stat_k(const Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* bl = reinterpret_cast<f32x4*>(smem);                 // B fragments: [slot 3][k-group 18][lane 64] f32x4 = 3 x 18 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int jc = lane & 15, qq = lane >> 4;
    int cluster, member;
    if (p.placement == 0) { const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3; cluster = xcd * 2 + idx / CS; member = idx % CS; }
    else { cluster = blockIdx.x / CS; member = blockIdx.x % CS; }
    unsigned long long t_start;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_start) : : "memory");

    // this wave's 16 gate columns (4 units x 4 gates), all 18 k-groups: stationary in registers
    f32x4 w[KG];
#pragma unroll
    for (int g = 0; g < KG; ++g) {
        const float s = 0.02f * (float)((lane * 7 + g * 13 + member * 3 + wave) % 17 - 8);
        w[g] = (f32x4){s, -s * 0.5f, s * 0.25f, 0.01f};
    }
    float c[NT], hval[NT];
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) { c[ct] = 0.0f; hval[ct] = 0.01f * (float)(jc + ct); }
    // the input k-groups (x = [obs | act], 2 k-groups) are produced locally in the probe (the product: second, smaller exchange)
    for (int i = tid; i < 3 * KG * 64; i += 256) bl[i] = (f32x4){0.01f, 0.02f, -0.01f, 0.005f};
    __syncthreads();

    const __amdgpu_buffer_rsrc_t xrs = rsrc(p.gran, 2 * 16 * CLUSTER_BYTES);
    unsigned int spin_left = 1u << 20;
    float dep = 0.0f;

    for (int t = 0; t < p.steps; ++t) {
        const unsigned int tag = p.tag0 + (unsigned int)t + 1u;
        const long long base_w = ((long long)(t & 1) * 16 + cluster) * CLUSTER_BYTES;             // written this step
        const long long base_r = ((long long)((t + 1) & 1) * 16 + cluster) * CLUSTER_BYTES;       // written by step t - 1
        // fetch the hidden-state k-groups of tile ct (published in step t - 1) into LDS slot: wave w takes k-groups w, w + 4, ..
        auto fetch = [&](int ct, int slot) {
            if (MODE == 0 || t == 0) return;
#pragma unroll
            for (int k = 0; k < KGH / 4; ++k) {
                const int g = wave + 4 * k;
                const int off = (int)(base_r + (long long)ct * TILE_BYTES + (g * 64 + lane) * 32);
                f32x4 v;
                while (true) {
                    const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, SC1);
                    const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(xrs, off + 16, 0, SC1);
                    const unsigned int want = tag - 1u;
                    const bool ok = a.x == want && a.z == want && b.x == want && b.z == want;
                    v = (f32x4){__uint_as_float(a.y), __uint_as_float(a.w), __uint_as_float(b.y), __uint_as_float(b.w)};
                    if (__all(ok)) break;
                    if (spin_left == 0) { if (lane == 0) atomicAdd(p.fail, 1u); break; }
                    --spin_left;
                    __builtin_amdgcn_s_sleep(2);
                }
                bl[(slot * KG + 2 + g) * 64 + lane] = v;
            }
        };
        auto publish = [&](int ct, float h) {
            if (MODE == 0) return;
            // lane (jc, qq) of wave `wave` holds unit 16 member + 4 wave + qq of candidate jc: the B fragment of k-group `member`
            // wants it in lane (jc, wave), element qq
            const int off = (int)(base_w + (long long)ct * TILE_BYTES + ((member * 64 + (wave * 16 + jc)) * 4 + qq) * 8);
            const unsigned long long g = ((unsigned long long)__float_as_uint(h) << 32) | tag;   // little endian: {tag, value}
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned int, g), xrs, off, 0, SC1);
        };
        if (MODE == 1) {
            // ---- the all-gather alone: publish everything, then fetch everything --------------------------------------------
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) publish(ct, hval[ct] + dep);
            // (read what THIS step published: parity t)
            for (int ct = 0; ct < NT; ++ct) {
#pragma unroll
                for (int k = 0; k < KGH / 4; ++k) {
                    const int g = wave + 4 * k;
                    const int off = (int)(base_w + (long long)ct * TILE_BYTES + (g * 64 + lane) * 32);
                    while (true) {
                        const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, SC1);
                        const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(xrs, off + 16, 0, SC1);
                        const bool ok = a.x == tag && a.z == tag && b.x == tag && b.z == tag;
                        dep += 1e-9f * (__uint_as_float(a.y) + __uint_as_float(b.w));
                        if (__all(ok)) break;
                        if (spin_left == 0) { if (lane == 0) atomicAdd(p.fail, 1u); break; }
                        --spin_left;
                        __builtin_amdgcn_s_sleep(2);
                    }
                }
            }
            __syncthreads();
            continue;
        }
        // ---- gemm / both: per candidate tile  fetch (two ahead) | barrier | MFMAs | cell | publish ---------------------------
        fetch(0, 0);
        fetch(1, 1);
        for (int ct = 0; ct < NT; ++ct) {
            if (ct + 2 < NT) fetch(ct + 2, (ct + 2) % 3);
            __syncthreads();            // tile ct's fragments are in LDS (fetched two iterations ago by all four waves)
            const f32x4* b = bl + (ct % 3) * KG * 64;
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < KG; ++g) {
                const f32x4 bv = b[g * 64 + lane];
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) acc = MFMA(w[g][ii], bv[ii], acc);
            }
            float cc = c[ct];
            const float h = cell(acc, cc);
            c[ct] = cc;
            hval[ct] = h;
            publish(ct, h);
        }
        __syncthreads();                // slots are refilled by the next step's fetches
    }
    float s = dep;
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) s += hval[ct] + c[ct];
    unsigned long long t_end;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_end) : : "memory");
    if (tid == 0) { p.clk[2 * blockIdx.x] = t_start; p.clk[2 * blockIdx.x + 1] = t_end; }
    if (lane == 0) atomicAdd(p.out + blockIdx.x, s);
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 10;
    Params p;
    const size_t gbytes = 2 * 16 * (size_t)CLUSTER_BYTES;
    (void)hipMalloc(&p.gran, gbytes); (void)hipMemset(p.gran, 0, gbytes);
    (void)hipMalloc(&p.out, 256 * 4); (void)hipMemset(p.out, 0, 256 * 4);
    (void)hipMalloc(&p.clk, 256 * 16); (void)hipMalloc(&p.fail, 4); (void)hipMemset(p.fail, 0, 4);
    p.steps = steps;
    unsigned int nonce = 0;
    const int smem = 3 * KG * 64 * 16;
    const char* names[3] = {"gemm", "xchg", "both"};
    for (int placement = 0; placement < 2; ++placement)
        for (int mode = 0; mode < 3; ++mode) {
            if (mode == 0 && placement == 1) continue;
            p.placement = placement; p.mode = mode;
            auto launch = [&]() {
                p.tag0 = (++nonce) << 12;
                if (mode == 0) hipLaunchKernelGGL(stat_k<0>, dim3(256), dim3(256), smem, 0, p);
                else if (mode == 1) hipLaunchKernelGGL(stat_k<1>, dim3(256), dim3(256), smem, 0, p);
                else hipLaunchKernelGGL(stat_k<2>, dim3(256), dim3(256), smem, 0, p);
            };
            for (int i = 0; i < 200; ++i) launch();           // clocks up
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
            printf("{\"probe\": \"stationary cluster, LSTM 256 at the ReBAL plan size (16 clusters x 16 CUs x 160 candidates)\", \"kernel\": \"%s\", "
                   "\"placement\": \"%s\", \"steps\": %d, \"launch_ms\": %.4f, \"clk_per_step_longest_wg\": %.0f, \"gave_up_polls\": %u, \"hip\": \"%s\"}\n",
                   names[mode], placement == 0 ? "cluster on one XCD" : "cluster over eight XCDs", steps, ms,
                   (double)worst / steps, fails, hipGetErrorString(err));
            fflush(stdout);
        }
    return 0;
}
