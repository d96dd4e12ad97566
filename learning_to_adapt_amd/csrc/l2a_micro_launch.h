// l2a_micro_launch.h - host-side entry points of the micro-tile kernel instances (l2a_micro_inst.hip).
#pragma once

#include <hip/hip_runtime.h>

struct L2ALstmParams;
struct L2AKParams;

// Returns 0, a hipError_t (> 0) or -100 when no instance exists for the shape.
int l2a_launch_lstm_micro(int units, const L2ALstmParams* p, unsigned grid, int smem, hipStream_t stream);
// generic recurrent cells (l2a_rnn_micro.h): every layer `units` wide, cell_type L2A_CELL_*, mtm = micro tiles of the largest workgroup (3 | 4)
int l2a_launch_rnn_micro(int units, int cell_type, int mtm, const L2ALstmParams* p, unsigned grid, int smem, hipStream_t stream);
int l2a_launch_mlp_micro(int hidden, int gact, const L2AKParams* p, unsigned grid, int smem, hipStream_t stream);
