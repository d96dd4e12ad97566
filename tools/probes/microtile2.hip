// microtile2.hip - round-4 probe: what does the inner loop of a 4-candidate-tile kernel (v_mfma_f32_4x4x1_16b_f32) need to reach
// the matrix rate?  Same experiment as microtile.hip (512 x 512 hidden GEMM, weights streamed from L2 in fragment order,
// activations in LDS rows, 60 layers per launch, clocks per layer and workgroup), but the loop is taken apart:
//
//   STYLE 0  plain operand ring, the compiler's own schedule (microtile.hip's loop)
//   STYLE 1  the ring with the memory operations spread between the MFMAs (sched_group_barrier), steps free to overlap
//   STYLE 2  as 1, every ring step pinned (sched_barrier(0) at its end)
//   STYLE 3  no memory operations in the loop at all (operands loaded once): the MFMA issue rate of the form itself
//   STYLE 4  weight loads only (activations loaded once)   STYLE 5  LDS reads only (weights loaded once)
//   NWAVE    4 = one wave per SIMD (two 64-unit tiles each), 8 = two per SIMD (one tile each)
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/microtile2 tools/probes/microtile2.hip && tools/probes/microtile2
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int H = 512, NMAT = 3, LAYERS = 60, ROW = H + 8;     // + 8: rows 8 banks apart, conflict-free 16-byte writes and reads

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

// packed A: float (((T * 128 + g) * 64 + lane) * 4 + ii) = W[k = 4 g + ii][unit = 64 T + lane]
template <int MT, int RING, int STYLE, int NWAVE>
__global__ void __launch_bounds__(64 * NWAVE) micro2_k(const float* wpk, float* out, unsigned long long* clocks) {
    constexpr int TW = 8 / NWAVE;           // 64-unit tiles per wave
    extern __shared__ __attribute__((aligned(16))) float act_raw[];
    float (*act)[16][ROW] = reinterpret_cast<float (*)[16][ROW]>(act_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * 16 * ROW; i += 64 * NWAVE) (&act[0][0][0])[i] = 0.001f * (float)((i * 7 + blockIdx.x) % 97);
    __syncthreads();
    unsigned long long t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    int cur = 0;
    for (int l = 0; l < LAYERS; ++l) {
        const __amdgpu_buffer_rsrc_t W = rsrc(wpk + (size_t)(l % NMAT) * H * H, (long long)H * H * 4);
        f32x4 acc[TW][MT];
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
            for (int c = 0; c < MT; ++c) acc[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* brow = &act[cur][lane & 3][0];
        constexpr int KG = H / 4;
        static_assert(KG % RING == 0, "ring depth must divide the k-groups");
        f32x4 a[RING][TW], b[RING][MT];
        constexpr bool LD_A = (STYLE != 3 && STYLE != 5), LD_B = (STYLE != 3 && STYLE != 4);
        auto issue_a = [&](int g, auto slot_tag) {
            constexpr int s = decltype(slot_tag)::value;
            const int gc = g < KG - 1 ? g : KG - 1;
#pragma unroll
            for (int t = 0; t < TW; ++t) a[s][t] = ldw(W, lane * 16, (((TW * wave + t) * KG + gc) * 64) * 16);
        };
        auto issue_b = [&](int g, auto slot_tag) {
            constexpr int s = decltype(slot_tag)::value;
            const int gc = g < KG - 1 ? g : KG - 1;
#pragma unroll
            for (int c = 0; c < MT; ++c) b[s][c] = *reinterpret_cast<const f32x4*>(brow + 4 * c * ROW + 4 * gc);
        };
        auto mfma = [&](auto slot_tag) {
            constexpr int s = decltype(slot_tag)::value;
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                for (int t = 0; t < TW; ++t)
#pragma unroll
                    for (int c = 0; c < MT; ++c)
                        acc[t][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[s][t][ii], b[s][c][ii], acc[t][c], 0, 0, 0);
        };
        // issue-order hint of one ring step: its TW weight loads and MT LDS reads spread evenly between its 4 TW MT MFMAs
        auto hint = [&]() {
            constexpr int NM = 4 * TW * MT, NOP = (LD_A ? TW : 0) + (LD_B ? MT : 0);
            if constexpr (NOP > 0) {
                constexpr int PER = NM / NOP;
                static_for<0, NOP>([&](auto it) {
                    constexpr int i = decltype(it)::value;
                    // weight loads and LDS reads alternate as long as both are left
                    constexpr bool vm = LD_A && (!LD_B || ((i & 1) == 0 ? (i / 2 < TW) : (i / 2 >= MT)));
                    if constexpr (vm) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, (i == NOP - 1) ? NM - PER * (NOP - 1) : PER, 0);
                });
            }
        };
        static_for<0, RING - 1>([&](auto i) { issue_a(decltype(i)::value, i); issue_b(decltype(i)::value, i); });
        if (!LD_A) issue_a(RING - 1, std::integral_constant<int, RING - 1>());
        if (!LD_B) issue_b(RING - 1, std::integral_constant<int, RING - 1>());
#pragma unroll 1
        for (int g0 = 0; g0 < KG; g0 += RING)
            static_for<0, RING>([&](auto i) {
                constexpr int I = decltype(i)::value;
                using NS = std::integral_constant<int, (I + RING - 1) % RING>;
                if (LD_A) issue_a(g0 + I + RING - 1, NS());
                if (LD_B) issue_b(g0 + I + RING - 1, NS());
                mfma(i);
                if (STYLE == 1 || STYLE == 2 || STYLE == 4 || STYLE == 5) hint();
                if (STYLE != 0 && STYLE != 1) __builtin_amdgcn_sched_barrier(0);
            });
        // D: lane (block b = lane >> 2, candidate lane & 3) holds units 64 T + 4 b + i -> the candidate's LDS row, relu
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
            for (int c = 0; c < MT; ++c) {
                f32x4 v = acc[t][c];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.0f) * 0.01f;
                *reinterpret_cast<f32x4*>(&act[cur ^ 1][4 * c + (lane & 3)][64 * (TW * wave + t) + 4 * (lane >> 2)]) = v;
            }
        __syncthreads();
        cur ^= 1;
    }
    unsigned long long t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (tid == 0) clocks[blockIdx.x] = t1 - t0;
    if (tid < 4 * MT) out[blockIdx.x * 16 + tid] = act[cur][tid][tid];
}


// ---- micro3: the loop without per-step address arithmetic -----------------------------------------------------------------
// Weight loads: one SGPR offset per four k-groups + the instruction's 12-bit immediate (0 / 1 / 2 / 3 KiB); LDS reads: one base
// VGPR per unrolled iteration + the 16-bit immediate; nothing is clamped (the arrays carry slack, requests past the last
// k-group read it).  Separate ring depths: RA k-groups of weights in flight (L2 latency), RB of activations (LDS latency;
// lgkmcnt is a 4-bit counter: more than 15 LDS reads in flight and the compiler has to wait for all of them).
template <int MT, int RA, int RB, int PIN>
__global__ void __launch_bounds__(256) micro3_k(const float* wpk, float* out, unsigned long long* clocks) {
    constexpr int TW = 2;
    static_assert(RA % 4 == 0 && RA % RB == 0, "ring depths");
    extern __shared__ __attribute__((aligned(16))) float act_raw[];
    float (*act)[16][ROW] = reinterpret_cast<float (*)[16][ROW]>(act_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * 16 * ROW + 512; i += 256) (&act[0][0][0])[i] = 0.001f * (float)((i * 7 + blockIdx.x) % 97);
    __syncthreads();
    unsigned long long t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    int cur = 0;
    constexpr int KG = H / 4;
    int voff[TW];
#pragma unroll
    for (int t = 0; t < TW; ++t) voff[t] = lane * 16 + (TW * wave + t) * KG * 1024;
    for (int l = 0; l < LAYERS; ++l) {
        const __amdgpu_buffer_rsrc_t W = rsrc(wpk + (size_t)(l % NMAT) * H * H, (long long)H * H * 4 + 65536);
        f32x4 acc[TW][MT];
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
            for (int c = 0; c < MT; ++c) acc[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* brow = &act[cur][lane & 3][0];
        f32x4 a[RA][TW], b[RB][MT];
        // group index = gbase (run time, multiple of 4) + GO (compile time)
        auto issue_a = [&](int gbase, auto go_tag, auto slot_tag) {
            constexpr int s = decltype(slot_tag)::value, GO = decltype(go_tag)::value;
#pragma unroll
            for (int t = 0; t < TW; ++t)
                a[s][t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(W, voff[t] + (GO & 3) * 1024, (gbase + (GO & ~3)) * 1024, 0));
        };
        auto issue_b = [&](const float* bp, auto go_tag, auto slot_tag) {
            constexpr int s = decltype(slot_tag)::value, GO = decltype(go_tag)::value;
#pragma unroll
            for (int c = 0; c < MT; ++c) b[s][c] = *reinterpret_cast<const f32x4*>(bp + 4 * c * ROW + 4 * GO);
        };
        auto mfma = [&](auto sa_tag, auto sb_tag) {
            constexpr int sa = decltype(sa_tag)::value, sb = decltype(sb_tag)::value;
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                for (int t = 0; t < TW; ++t)
#pragma unroll
                    for (int c = 0; c < MT; ++c)
                        acc[t][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[sa][t][ii], b[sb][c][ii], acc[t][c], 0, 0, 0);
        };
        auto hint = [&]() {
            constexpr int NM = 4 * TW * MT, NOP = TW + MT, PER = NM / NOP;
            static_for<0, NOP>([&](auto it) {
                constexpr int i = decltype(it)::value;
                constexpr bool vm = ((i & 1) == 0 ? (i / 2 < TW) : (i / 2 >= MT));
                if constexpr (vm) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, (i == NOP - 1) ? NM - PER * (NOP - 1) : PER, 0);
            });
        };
        static_for<0, RA - 1>([&](auto i) { issue_a(0, i, i); });
        static_for<0, RB - 1>([&](auto i) { issue_b(brow, i, i); });
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
        for (int g0 = 0; g0 < KG; g0 += RA) {
            const float* bp = brow + 4 * g0;
            static_for<0, RA>([&](auto i) {
                constexpr int I = decltype(i)::value;
                issue_a(g0, std::integral_constant<int, I + RA - 1>(), std::integral_constant<int, (I + RA - 1) % RA>());
                issue_b(bp, std::integral_constant<int, I + RB - 1>(), std::integral_constant<int, (I + RB - 1) % RB>());
                mfma(std::integral_constant<int, I % RA>(), std::integral_constant<int, I % RB>());
                hint();
                if (PIN) __builtin_amdgcn_sched_barrier(0);
            });
        }
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
            for (int c = 0; c < MT; ++c) {
                f32x4 v = acc[t][c];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.0f) * 0.01f;
                *reinterpret_cast<f32x4*>(&act[cur ^ 1][4 * c + (lane & 3)][64 * (TW * wave + t) + 4 * (lane >> 2)]) = v;
            }
        __syncthreads();
        cur ^= 1;
    }
    unsigned long long t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (tid == 0) clocks[blockIdx.x] = t1 - t0;
    if (tid < 4 * MT) out[blockIdx.x * 16 + tid] = act[cur][tid][tid];
}

template <class K>
static void run(const char* name, K kernel, int threads, int cand, int wgs, const float* w, float* out, unsigned long long* clk) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    constexpr int SMEM = 2 * 16 * ROW * 4 + 4096;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    for (int i = 0; i < 120; ++i) hipLaunchKernelGGL(kernel, dim3(wgs), dim3(threads), SMEM, 0, w, out, clk);   // clocks up
    (void)hipEventRecord(e0);
    const int reps = 30;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kernel, dim3(wgs), dim3(threads), SMEM, 0, w, out, clk);
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
    printf("{\"kernel\": \"%s\", \"candidates_per_workgroup\": %d, \"workgroups\": %d, \"launch_ms\": %.4f, \"clocks_per_layer\": %.0f, "
           "\"clocks_per_layer_max\": %.0f, \"mfma_clocks_per_layer_ideal\": %d, \"frac_of_matrix_time\": %.3f}\n",
           name, cand, wgs, ms / reps, per_layer, (double)mx / LAYERS, 8192 * (cand / 4), 8192.0 * (cand / 4) / per_layer);
    fflush(stdout);
}

#define RUN(MT, RING, STYLE, NWAVE, WGS) \
    run("micro2 MT=" #MT " ring=" #RING " style=" #STYLE " waves=" #NWAVE, micro2_k<MT, RING, STYLE, NWAVE>, 64 * NWAVE, 4 * MT, WGS, w, out, clk)
#define RUN3(MT, RA, RB, PIN, WGS) \
    run("micro3 MT=" #MT " RA=" #RA " RB=" #RB " pin=" #PIN, micro3_k<MT, RA, RB, PIN>, 256, 4 * MT, WGS, w, out, clk)

int main(int argc, char** argv) {
    float *w, *out; unsigned long long* clk;
    (void)hipMalloc(&w, sizeof(float) * NMAT * H * H + 131072);
    (void)hipMalloc(&out, sizeof(float) * 16 * 1024);
    (void)hipMalloc(&clk, sizeof(unsigned long long) * 1024);
    std::vector<float> hw((size_t)NMAT * H * H);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0.02f * (float)((int)(i * 2654435761u % 201) - 100) / 100.0f;
    (void)hipMemcpy(w, hw.data(), sizeof(float) * hw.size(), hipMemcpyHostToDevice);
    const bool all = argc > 1;
    if (all) {
        // what the form can issue at all, and what each operand stream costs on its own (12 candidates)
        RUN(3, 8, 3, 4, 256); RUN(3, 8, 4, 4, 256); RUN(3, 8, 5, 4, 256);
        RUN(2, 8, 3, 4, 256); RUN(1, 8, 3, 4, 256);
        // the full loop: compiler's schedule / spread / spread + pinned, ring depths
        RUN(3, 8, 0, 4, 256); RUN(3, 8, 1, 4, 256); RUN(3, 8, 2, 4, 256);
        RUN(3, 16, 0, 4, 256); RUN(3, 16, 1, 4, 256); RUN(3, 16, 2, 4, 256);
        RUN(3, 4, 2, 4, 256);
        RUN(2, 8, 0, 4, 256); RUN(2, 8, 1, 4, 256); RUN(2, 8, 2, 4, 256); RUN(2, 16, 2, 4, 256);
        RUN(1, 8, 0, 4, 256); RUN(1, 8, 2, 4, 256); RUN(1, 16, 2, 4, 256); RUN(1, 16, 2, 4, 125);
        // two waves per SIMD, one 64-unit tile each
        RUN(3, 8, 0, 8, 256); RUN(3, 8, 2, 8, 256); RUN(3, 16, 2, 8, 256); RUN(2, 8, 2, 8, 256); RUN(1, 8, 2, 8, 256); RUN(1, 16, 2, 8, 125);
    }
    RUN(3, 8, 3, 4, 256); RUN(3, 8, 1, 4, 256);
    RUN3(3, 8, 2, 0, 256); RUN3(3, 8, 2, 1, 256); RUN3(3, 8, 4, 0, 256); RUN3(3, 8, 4, 1, 256);
    RUN3(3, 12, 2, 0, 256); RUN3(3, 12, 4, 0, 256); RUN3(3, 16, 2, 0, 256); RUN3(3, 16, 4, 0, 256); RUN3(3, 4, 2, 0, 256); RUN3(3, 4, 4, 0, 256);
    RUN3(2, 8, 2, 0, 256); RUN3(2, 8, 4, 0, 256); RUN3(2, 12, 4, 0, 256); RUN3(2, 16, 4, 0, 256); RUN3(2, 16, 4, 1, 256);
    RUN3(1, 8, 4, 0, 256); RUN3(1, 16, 4, 0, 256); RUN3(1, 16, 8, 0, 125);
    return 0;
}
