// gridbar.hip - what does a barrier of the resident grid cost on MI355X?  (round 4: the fused adaptation step, csrc/l2a_adapt.h)
//   flat  : one device-scope counter, every workgroup adds one and polls it (sc1 loads)
//   tree  : one counter per XCD (workgroup i sits on XCD i % 8); the last arrival of an XCD adds one to the root; everybody
//           polls the root
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/gridbar tools/probes/gridbar.hip && tools/probes/gridbar
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void __launch_bounds__(512) bar_k(unsigned int* ctr, int rounds, unsigned int base, unsigned long long* out) {
    const unsigned int G = gridDim.x;
    unsigned long long t0 = 0, t1 = 0;
    if (threadIdx.x == 0) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    unsigned int target = base, xt = base / 8 * 0;
    const unsigned int xcd = blockIdx.x & 7, per = (G + 7 - xcd) / 8;      // workgroups of this XCD
    unsigned int xtarget = (base / G) * per;
    for (int r = 0; r < rounds; ++r) {
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        if (threadIdx.x == 0) {
            if (MODE == 0) {
                target += G;
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) __builtin_amdgcn_s_sleep(1);
            } else {
                target += 8;
                xtarget += per;
                const unsigned int old = __hip_atomic_fetch_add(ctr + 64 * (1 + xcd), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old + 1 == xtarget) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
    (void)xt;
    if (threadIdx.x == 0) {
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        out[blockIdx.x] = t1 - t0;
    }
}

int main() {
    unsigned int* ctr; unsigned long long* out;
    (void)hipMalloc(&ctr, 64 * 9 * 4); (void)hipMalloc(&out, 8 * 1024);
    for (int mode = 0; mode < 2; ++mode)
        for (int G : {40, 160, 256}) {
            (void)hipMemset(ctr, 0, 64 * 9 * 4);
            const int rounds = 200;
            unsigned int base = 0;
            void* args0[] = {&ctr, (void*)&rounds, &base, &out};
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
                (void)hipEventRecord(e0);
                hipError_t e = hipLaunchCooperativeKernel(mode == 0 ? (const void*)bar_k<0> : (const void*)bar_k<1>, dim3(G), dim3(512), args0, 0, 0);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
                if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); break; }
                if (ms < best) best = ms;
                base += (mode == 0 ? (unsigned)G : 8u) * rounds;
                if (mode == 1) { (void)hipMemset(ctr, 0, 64 * 9 * 4); base = 0; }
            }
            unsigned long long h[256]; (void)hipMemcpy(h, out, 8 * G, hipMemcpyDeviceToHost);
            unsigned long long mx = 0; for (int i = 0; i < G; ++i) mx = h[i] > mx ? h[i] : mx;
            printf("{\"barrier\": \"%s\", \"workgroups\": %d, \"rounds\": %d, \"launch_ms\": %.4f, \"us_per_barrier_in_kernel\": %.3f}\n",
                   mode == 0 ? "flat" : "per-XCD tree", G, rounds, best, (double)mx * 0.01 / rounds);
        }
    return 0;
}
