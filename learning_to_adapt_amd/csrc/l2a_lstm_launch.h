// l2a_lstm_launch.h - host-side entry points of the MFMA LSTM kernel instances
// (l2a_lstm_inst.hip compiled once per UTW = units / 64).
#pragma once

#include <hip/hip_runtime.h>

struct L2ALstmParams;

// Returns 0, a hipError_t (> 0) or -100 when no instance exists for the shape.
int l2a_launch_lstm_2(int nt, int ot, int kg0, const L2ALstmParams* p, unsigned grid, int smem, hipStream_t stream);
int l2a_launch_lstm_4(int nt, int ot, int kg0, const L2ALstmParams* p, unsigned grid, int smem, hipStream_t stream);
int l2a_launch_lstm_8(int nt, int ot, int kg0, const L2ALstmParams* p, unsigned grid, int smem, hipStream_t stream);

inline int l2a_launch_lstm(int utw, int nt, int ot, int kg0, const L2ALstmParams* p, unsigned grid, int smem,
                           hipStream_t stream) {
    if (utw == 2) return l2a_launch_lstm_2(nt, ot, kg0, p, grid, smem, stream);
    if (utw == 4) return l2a_launch_lstm_4(nt, ot, kg0, p, grid, smem, stream);
    if (utw == 8) return l2a_launch_lstm_8(nt, ot, kg0, p, grid, smem, stream);
    return -100;
}
