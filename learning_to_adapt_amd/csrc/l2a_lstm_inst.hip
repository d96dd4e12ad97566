// l2a_lstm_inst.hip - one translation unit per L2A_INST_UTW (= units / 64): instantiates the MFMA
// LSTM rollout kernel for every (OT, KG0) and exports its launcher (see l2a_lstm_launch.h).
#include "l2a_lstm.h"

#if !defined(L2A_INST_UTW)
#error "compile with -DL2A_INST_UTW=<2|4|8>"
#endif

namespace {

template <int NT, int OT, int KG0, bool SPLIT>
int launch_split(const L2ALstmParams* p, unsigned grid, int smem, hipStream_t stream) {
    auto kernel = l2a_lstm_mfma_k<NT, L2A_INST_UTW, OT, KG0, SPLIT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(64 * L2A_NW), smem, stream, *p);
    return 0;
}

template <int NT, int OT, int KG0>
int launch_one(const L2ALstmParams* p, unsigned grid, int smem, hipStream_t stream) {
    return p->split ? launch_split<NT, OT, KG0, true>(p, grid, smem, stream)
                    : launch_split<NT, OT, KG0, false>(p, grid, smem, stream);
}

}  // namespace

#define L2A_CAT2(a, b) a##b
#define L2A_LNAME(utw) L2A_CAT2(l2a_launch_lstm_, utw)

int L2A_LNAME(L2A_INST_UTW)(int nt, int ot, int kg0, const L2ALstmParams* p, unsigned grid, int smem,
                            hipStream_t stream) {
    if (nt != 1) return -100;
    switch (ot * 8 + kg0) {
        case 1 * 8 + 1: return launch_one<1, 1, 1>(p, grid, smem, stream);
        case 1 * 8 + 2: return launch_one<1, 1, 2>(p, grid, smem, stream);
        case 2 * 8 + 2: return launch_one<1, 2, 2>(p, grid, smem, stream);
        case 2 * 8 + 3: return launch_one<1, 2, 3>(p, grid, smem, stream);
        case 3 * 8 + 3: return launch_one<1, 3, 3>(p, grid, smem, stream);
        case 3 * 8 + 4: return launch_one<1, 3, 4>(p, grid, smem, stream);
        case 4 * 8 + 4: return launch_one<1, 4, 4>(p, grid, smem, stream);
        case 4 * 8 + 5: return launch_one<1, 4, 5>(p, grid, smem, stream);
        default: return -100;
    }
}
