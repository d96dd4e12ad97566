// l2a_rnn_micro_inst.hip - the micro-tile kernels of the generic recurrent cells (l2a_rnn_micro.h) as one translation unit.
#include "l2a_rnn_micro.h"
#include "l2a_micro_launch.h"

namespace {

template <int UW, int CELL>
int launch_rnn(const L2ALstmParams* p, unsigned grid, int smem, hipStream_t stream) {
    auto kernel = l2a_rnn_micro_k<UW, CELL>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), smem, stream, *p);
    return 0;
}

}  // namespace

int l2a_launch_rnn_micro(int units, int cell_type, const L2ALstmParams* p, unsigned grid, int smem, hipStream_t stream) {
    if (units == 256) {
        if (cell_type == L2A_CELL_LSTM) return launch_rnn<1, L2A_CELL_LSTM>(p, grid, smem, stream);
        if (cell_type == L2A_CELL_GRU) return launch_rnn<1, L2A_CELL_GRU>(p, grid, smem, stream);
        if (cell_type == L2A_CELL_RNN) return launch_rnn<1, L2A_CELL_RNN>(p, grid, smem, stream);
    }
    if (units == 512) {
        // (no LSTM instance: one 512-unit LSTM layer has its tuned kernel, two do not fit the LDS)
        if (cell_type == L2A_CELL_GRU) return launch_rnn<2, L2A_CELL_GRU>(p, grid, smem, stream);
        if (cell_type == L2A_CELL_RNN) return launch_rnn<2, L2A_CELL_RNN>(p, grid, smem, stream);
    }
    return -100;
}
