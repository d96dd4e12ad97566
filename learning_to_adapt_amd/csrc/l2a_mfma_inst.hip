// l2a_mfma_inst.hip - one translation unit per (L2A_INST_NT, L2A_INST_TPW): instantiates the MFMA
// rollout kernel for every (OT, KG0, GACT) and exports its launcher (see l2a_mfma_launch.h).
// With -DL2A_INST_FAN=1 the unit holds the member-fan instances instead (one workgroup per candidate tile and
// ensemble member, l2a_mfma.h "Member fan") and exports l2a_launch_mfma_fan_<NT>_<TPW>; with -DL2A_INST_FAN=2 the
// whole-tiles-only instances (no exchange, no half member: the double rounds of multi-round plans at width 512) as
// l2a_launch_mfma_whole_<NT>_<TPW>.
#include "l2a_mfma.h"

#ifndef L2A_INST_FAN
#define L2A_INST_FAN 0
#endif

#if !defined(L2A_INST_NT) || !defined(L2A_INST_TPW)
#error "compile with -DL2A_INST_NT=<1|2> -DL2A_INST_TPW=<2|4|8>"
#endif

namespace {

template <int OT, int KG0, bool GACT, int K0L = 4, bool N1 = false, bool O4 = false>
int launch_one(const L2AKParams* p, unsigned grid, int smem, hipStream_t stream) {
    // one hidden layer: its own instances (no hidden->hidden GEMM in them), generic activation code only
    if (!N1 && p->n_hidden == 1) return launch_one<OT, KG0, true, 4, true>(p, grid, smem, stream);
    auto kernel = l2a_rollout_mfma_k<L2A_INST_NT, L2A_INST_TPW, OT, KG0, GACT, K0L, N1, O4, L2A_INST_FAN == 1, L2A_INST_FAN == 2>;
    // the dynamic-LDS ceiling of this instance is raised once per device and size, not on every launch (~2 us each)
    static int smem_set[16] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 16 || smem > smem_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return (int)e;
        if (dev >= 0 && dev < 16) smem_set[dev] = smem;
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(64 * L2A_NW), smem, stream, *p);
    return 0;
}

template <bool GACT>
int launch_shape(int ot, int kg0, const L2AKParams* p, unsigned grid, int smem, hipStream_t stream) {
    // (OT, KG0) = (ceil(obs/16), ceil((obs+act)/16)); act_dim <= 16 makes KG0 either OT or OT + 1.
    switch (ot * 8 + kg0) {
        case 1 * 8 + 1: return launch_one<1, 1, GACT>(p, grid, smem, stream);
        case 1 * 8 + 2: return launch_one<1, 2, GACT>(p, grid, smem, stream);
        case 2 * 8 + 2:     // HalfCheetah: 20 = 16 + 4 observations - the last obs tile has four live units (O4 instance)
            if (p->n_hidden > 1 && p->obs_dim - 16 >= 1 && p->obs_dim - 16 <= 4)
                return launch_one<2, 2, GACT, 4, false, true>(p, grid, smem, stream);
            return launch_one<2, 2, GACT>(p, grid, smem, stream);
        case 2 * 8 + 3: return launch_one<2, 3, GACT>(p, grid, smem, stream);
        case 3 * 8 + 3: return launch_one<3, 3, GACT>(p, grid, smem, stream);
        case 3 * 8 + 4:     // Ant: 41 + 8 = 49 inputs end one feature into the last k-group (its own instance; relu / identity)
            if (!GACT && p->in_dim == 49) return launch_one<3, 4, GACT, 1>(p, grid, smem, stream);
            return launch_one<3, 4, GACT>(p, grid, smem, stream);
#if L2A_INST_NT == 1
        // 49 - 64 observation dims: NT = 1 only (choose_nt, l2a_api.hip: the NT = 2 instances spilled 43 - 58 VGPRs and lost every
        // plan measured; they are no longer built)
        case 4 * 8 + 4: return launch_one<4, 4, GACT>(p, grid, smem, stream);
        case 4 * 8 + 5: return launch_one<4, 5, GACT>(p, grid, smem, stream);
#endif
        default: return -100;
    }
}

}  // namespace

#define L2A_CAT3(a, b, c) a##b##_##c
#if L2A_INST_FAN == 2
#define L2A_NAME(nt, tpw) L2A_CAT3(l2a_launch_mfma_whole_, nt, tpw)
#elif L2A_INST_FAN
#define L2A_NAME(nt, tpw) L2A_CAT3(l2a_launch_mfma_fan_, nt, tpw)
#else
#define L2A_NAME(nt, tpw) L2A_CAT3(l2a_launch_mfma_, nt, tpw)
#endif

int L2A_NAME(L2A_INST_NT, L2A_INST_TPW)(int ot, int kg0, int gact, const L2AKParams* p, unsigned grid, int smem,
                                         hipStream_t stream) {
    return gact ? launch_shape<true>(ot, kg0, p, grid, smem, stream)
                : launch_shape<false>(ot, kg0, p, grid, smem, stream);
}
