// microtile.hip - probe for DESIGN.md section 9 item 3: can a workgroup own FEWER than 16 candidates at full fp32 matrix rate?
//
// v_mfma_f32_16x16x4_f32 fixes the candidate tile at N = 16; v_mfma_f32_4x4x1_16b_f32 multiplies 64 units x 4 candidates x
// one k (sixteen 4x4 blocks, the four candidates replicated over the blocks) in 2 passes - the same 32 MAC/clock.  The probe
// runs the hidden GEMM of the planner's MLP (512 x 512, weights streamed from L2 in fragment order, activations in LDS rows,
// one barrier per layer, 60 layers per launch) with CT = 4 / 8 / 12 / 16 candidates per workgroup on the 4x4x1 form and with
// 16 on the 16x16x4 form, and prints clocks per layer and workgroup: the matrix pipe's share (CT / 4 x 8192 clocks) against
// the weight stream's (1 MiB per layer and workgroup through a 64 B/clock L1: 16384 clocks).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/microtile tools/probes/microtile.hip && tools/probes/microtile
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int H = 512, NMAT = 3, LAYERS = 60, ROW = H + 4;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* base, long long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 ldw(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>()); static_for<I + 1, N>(f); }
}

// ---- 4x4x1 form: wave w owns the 64-unit tiles 2w, 2w + 1; CT / 4 micro-tiles of four candidates ---------------------------
// packed A: float (((T * 128 + g) * 64 + lane) * 4 + ii) = W[k = 4 g + ii][unit = 64 T + lane]
template <int CT, int NB>
__global__ void __launch_bounds__(256) micro_k(const float* wpk, float* out, unsigned long long* clocks) {
    constexpr int MT = CT / 4;
    extern __shared__ __attribute__((aligned(16))) float act_raw[];
    float (*act)[16][ROW] = reinterpret_cast<float (*)[16][ROW]>(act_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * 16 * ROW; i += 256) (&act[0][0][0])[i] = 0.001f * (float)((i * 7 + blockIdx.x) % 97);
    __syncthreads();
    unsigned long long t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    int cur = 0;
    for (int l = 0; l < LAYERS; ++l) {
        const __amdgpu_buffer_rsrc_t W = rsrc(wpk + (size_t)(l % NMAT) * H * H, (long long)H * H * 4);
        f32x4 acc[2][MT];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int c = 0; c < MT; ++c) acc[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* brow = &act[cur][lane & 3][0];
        constexpr int KG = H / 4;                   // k-groups of four; operand ring of NB slots, NB - 1 requests ahead
        static_assert(KG % NB == 0, "ring depth must divide the k-groups");
        f32x4 a[NB][2], b[NB][MT];
        auto issue = [&](int g, auto slot_tag) {
            constexpr int s = decltype(slot_tag)::value;
            const int gc = g < KG - 1 ? g : KG - 1;
#pragma unroll
            for (int t = 0; t < 2; ++t) a[s][t] = ldw(W, lane * 16, (((2 * wave + t) * KG + gc) * 64) * 16);
#pragma unroll
            for (int c = 0; c < MT; ++c) b[s][c] = *reinterpret_cast<const f32x4*>(brow + 4 * c * ROW + 4 * gc);
        };
        auto mfma = [&](auto slot_tag) {
            constexpr int s = decltype(slot_tag)::value;
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int c = 0; c < MT; ++c)
                        acc[t][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[s][t][ii], b[s][c][ii], acc[t][c], 0, 0, 0);
        };
        static_for<0, NB - 1>([&](auto i) { issue(decltype(i)::value, i); });
        for (int g0 = 0; g0 < KG; g0 += NB)
            static_for<0, NB>([&](auto i) {
                constexpr int I = decltype(i)::value;
                issue(g0 + I + NB - 1, std::integral_constant<int, (I + NB - 1) % NB>());
                mfma(i);
            });
        // D: lane (block b = lane >> 2, candidate lane & 3) holds units 64 T + 4 b + i -> the candidate's LDS row, relu
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int c = 0; c < MT; ++c) {
                f32x4 v = acc[t][c];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.0f) * 0.01f;
                *reinterpret_cast<f32x4*>(&act[cur ^ 1][4 * c + (lane & 3)][64 * (2 * wave + t) + 4 * (lane >> 2)]) = v;
            }
        __syncthreads();
        cur ^= 1;
    }
    unsigned long long t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (tid == 0) clocks[blockIdx.x] = t1 - t0;
    if (tid < CT) out[blockIdx.x * 16 + tid] = act[cur][tid][tid];
}

// ---- 16x16x4 form (today's tile): wave w owns the 16-unit tiles 8w .. 8w + 7, 16 candidates -----------------------------------
// packed A: float (((T * 32 + g) * 64 + lane) * 4 + ii) = W[k = 16 g + 4 (lane >> 4) + ii][unit = 16 T + (lane & 15)]
__global__ void __launch_bounds__(256) full_k(const float* wpk, float* out, unsigned long long* clocks) {
    extern __shared__ __attribute__((aligned(16))) float act_raw[];
    float (*act)[16][ROW] = reinterpret_cast<float (*)[16][ROW]>(act_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), jc = lane & 15, qq = lane >> 4;
    for (int i = tid; i < 2 * 16 * ROW; i += 256) (&act[0][0][0])[i] = 0.001f * (float)((i * 7 + blockIdx.x) % 97);
    __syncthreads();
    unsigned long long t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    int cur = 0;
    for (int l = 0; l < LAYERS; ++l) {
        const __amdgpu_buffer_rsrc_t W = rsrc(wpk + (size_t)(l % NMAT) * H * H, (long long)H * H * 4);
        f32x4 acc[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* brow = &act[cur][jc][4 * qq];
        constexpr int KG = H / 16, NB = 4;
        f32x4 a[NB][8], b[NB];
        auto issue = [&](int g, auto slot_tag) {
            constexpr int s = decltype(slot_tag)::value;
            const int gc = g < KG - 1 ? g : KG - 1;
#pragma unroll
            for (int t = 0; t < 8; ++t) a[s][t] = ldw(W, lane * 16, (((8 * wave + t) * KG + gc) * 64) * 16);
            b[s] = *reinterpret_cast<const f32x4*>(brow + 16 * gc);
        };
        auto mfma = [&](auto slot_tag) {
            constexpr int s = decltype(slot_tag)::value;
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][t][ii], b[s][ii], acc[t], 0, 0, 0);
        };
        using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
        using S2 = std::integral_constant<int, 2>; using S3 = std::integral_constant<int, 3>;
        issue(0, S0()); issue(1, S1()); issue(2, S2());
        for (int g0 = 0; g0 < KG; g0 += NB) {
            issue(g0 + 3, S3()); mfma(S0());
            issue(g0 + 4, S0()); mfma(S1());
            issue(g0 + 5, S1()); mfma(S2());
            issue(g0 + 6, S2()); mfma(S3());
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            f32x4 v = acc[t];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.0f) * 0.01f;
            *reinterpret_cast<f32x4*>(&act[cur ^ 1][jc][16 * (8 * wave + t) + 4 * qq]) = v;
        }
        __syncthreads();
        cur ^= 1;
    }
    unsigned long long t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (tid == 0) clocks[blockIdx.x] = t1 - t0;
    if (tid < 16) out[blockIdx.x * 16 + tid] = act[cur][tid][tid];
}

template <class K>
static void run(const char* name, K kernel, int cand, int wgs, const float* w, float* out, unsigned long long* clk) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    constexpr int SMEM = 2 * 16 * ROW * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(kernel, dim3(wgs), dim3(256), SMEM, 0, w, out, clk);   // clocks up
    (void)hipEventRecord(e0);
    const int reps = 50;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kernel, dim3(wgs), dim3(256), SMEM, 0, w, out, clk);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return; }
    std::vector<unsigned long long> h(wgs);
    (void)hipMemcpy(h.data(), clk, sizeof(unsigned long long) * wgs, hipMemcpyDeviceToHost);
    unsigned long long mx = 0; double mean = 0;
    for (auto v : h) { mx = v > mx ? v : mx; mean += (double)v / wgs; }
    const double per_layer = mean / LAYERS;
    const double flop = 2.0 * H * H * cand * (double)wgs * LAYERS;
    printf("{\"kernel\": \"%s\", \"candidates_per_workgroup\": %d, \"workgroups\": %d, \"launch_ms\": %.4f, \"clocks_per_layer\": %.0f, "
           "\"clocks_per_layer_max\": %.0f, \"mfma_clocks_per_layer_ideal\": %d, \"weight_bytes_per_clock\": %.1f, \"tflops\": %.2f}\n",
           name, cand, wgs, ms / reps, per_layer, (double)mx / LAYERS, cand == 16 && name[0] == 'f' ? 32768 : 8192 * (cand / 4),
           (double)H * H * 4 / per_layer, flop / (ms / reps) / 1e9);
}

int main(int argc, char** argv) {
    float *w, *out; unsigned long long* clk;
    (void)hipMalloc(&w, sizeof(float) * NMAT * H * H);
    (void)hipMalloc(&out, sizeof(float) * 16 * 1024);
    (void)hipMalloc(&clk, sizeof(unsigned long long) * 1024);
    std::vector<float> hw((size_t)NMAT * H * H);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0.02f * (float)((int)(i * 2654435761u % 201) - 100) / 100.0f;
    (void)hipMemcpy(w, hw.data(), sizeof(float) * hw.size(), hipMemcpyHostToDevice);
    // the shapes the idea is for: 500 candidates (config 1), 2500 (both 160-tile defaults), a full chip of today's tiles
    run("full 16x16x4", full_k, 16, 32, w, out, clk);
    run("full 16x16x4", full_k, 16, 160, w, out, clk);
    run("full 16x16x4", full_k, 16, 256, w, out, clk);
    run("micro 4x4x1 ring 4", micro_k<16, 4>, 16, 160, w, out, clk);
    run("micro 4x4x1 ring 4", micro_k<12, 4>, 12, 209, w, out, clk);
    run("micro 4x4x1 ring 4", micro_k<12, 4>, 12, 256, w, out, clk);
    run("micro 4x4x1 ring 4", micro_k<8, 4>, 8, 256, w, out, clk);
    run("micro 4x4x1 ring 4", micro_k<4, 4>, 4, 125, w, out, clk);
    run("micro 4x4x1 ring 4", micro_k<4, 4>, 4, 256, w, out, clk);
    run("micro 4x4x1 ring 8", micro_k<16, 8>, 16, 160, w, out, clk);
    run("micro 4x4x1 ring 8", micro_k<12, 8>, 12, 209, w, out, clk);
    run("micro 4x4x1 ring 8", micro_k<8, 8>, 8, 256, w, out, clk);
    run("micro 4x4x1 ring 8", micro_k<4, 8>, 4, 125, w, out, clk);
    run("micro 4x4x1 ring 16", micro_k<12, 16>, 12, 209, w, out, clk);
    run("micro 4x4x1 ring 16", micro_k<8, 16>, 8, 256, w, out, clk);
    run("micro 4x4x1 ring 16", micro_k<4, 16>, 4, 125, w, out, clk);
    return 0;
}
