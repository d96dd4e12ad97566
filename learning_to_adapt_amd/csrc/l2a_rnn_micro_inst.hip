// l2a_rnn_micro_inst.hip - the micro-tile kernels of the generic recurrent cells (l2a_rnn_micro.h) as one translation unit.
#include "l2a_rnn_micro.h"
#include "l2a_micro_launch.h"

namespace {

template <int UW, int CELL, int MTM>
int launch_rnn(const L2ALstmParams* p, unsigned grid, int smem, hipStream_t stream) {
    auto kernel = l2a_rnn_micro_k<UW, CELL, MTM>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), smem, stream, *p);
    return 0;
}

}  // namespace

int l2a_launch_rnn_micro(int units, int cell_type, int mtm, const L2ALstmParams* p, unsigned grid, int smem, hipStream_t stream) {
    if (units == 256 && mtm == 3) {
        if (cell_type == L2A_CELL_LSTM) return launch_rnn<1, L2A_CELL_LSTM, 3>(p, grid, smem, stream);
        if (cell_type == L2A_CELL_GRU) return launch_rnn<1, L2A_CELL_GRU, 3>(p, grid, smem, stream);
        if (cell_type == L2A_CELL_RNN) return launch_rnn<1, L2A_CELL_RNN, 3>(p, grid, smem, stream);
    }
    if (units == 256 && mtm == 4) {
        if (cell_type == L2A_CELL_LSTM) return launch_rnn<1, L2A_CELL_LSTM, 4>(p, grid, smem, stream);
        if (cell_type == L2A_CELL_GRU) return launch_rnn<1, L2A_CELL_GRU, 4>(p, grid, smem, stream);
        if (cell_type == L2A_CELL_RNN) return launch_rnn<1, L2A_CELL_RNN, 4>(p, grid, smem, stream);
    }
    if (units == 512 && mtm == 3) {
        // (no LSTM instance: one 512-unit LSTM layer has its tuned kernel, two do not fit the LDS; no four-tile instances: LDS)
        if (cell_type == L2A_CELL_GRU) return launch_rnn<2, L2A_CELL_GRU, 3>(p, grid, smem, stream);
        if (cell_type == L2A_CELL_RNN) return launch_rnn<2, L2A_CELL_RNN, 3>(p, grid, smem, stream);
    }
    return -100;
}
