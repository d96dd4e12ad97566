// l2a_micro_inst.hip - the micro-tile kernels of l2a_micro.h as one translation unit; launchers in l2a_micro_launch.h.
#include "l2a_micro.h"
#include "l2a_micro_launch.h"

namespace {

template <int UW>
int launch_lstm(const L2ALstmParams* p, unsigned grid, int smem, hipStream_t stream) {
    auto kernel = l2a_lstm_micro_k<UW>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), smem, stream, *p);
    return 0;
}

template <int UW, bool GACT>
int launch_mlp(const L2AKParams* p, unsigned grid, int smem, hipStream_t stream) {
    auto kernel = l2a_mlp_micro_k<UW, GACT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), smem, stream, *p);
    return 0;
}

}  // namespace

int l2a_launch_lstm_micro(int units, const L2ALstmParams* p, unsigned grid, int smem, hipStream_t stream) {
    if (units == 256) return launch_lstm<1>(p, grid, smem, stream);
    if (units == 512) return launch_lstm<2>(p, grid, smem, stream);
    return -100;
}

int l2a_launch_mlp_micro(int hidden, int gact, const L2AKParams* p, unsigned grid, int smem, hipStream_t stream) {
    if (hidden == 256) return gact ? launch_mlp<1, true>(p, grid, smem, stream) : launch_mlp<1, false>(p, grid, smem, stream);
    if (hidden == 512) return gact ? launch_mlp<2, true>(p, grid, smem, stream) : launch_mlp<2, false>(p, grid, smem, stream);
    return -100;
}
