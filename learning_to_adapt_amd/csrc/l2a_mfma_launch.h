// l2a_mfma_launch.h - host-side entry points of the MFMA kernel instances.
//
// The kernel template (l2a_mfma.h) is instantiated per (NT, TPW, OT, KG0, GACT) combination; the instances are spread over
// one translation unit per (NT, TPW) pair (l2a_mfma_inst.hip compiled five times with different -D flags) so that the
// library builds in parallel.  There is no (2, 8) unit: at hidden width 512 two candidate tiles per workgroup spill
// registers and lose to NT = 1 (launch_rollout, l2a_api.hip).
#pragma once

#include <hip/hip_runtime.h>

struct L2AKParams;

#ifndef L2A_NW
#define L2A_NW 4
#endif

// Returns 0, a hipError_t (> 0) or -100 when no instance exists for the shape.
#define L2A_DECL_LAUNCH(NT_, TPW_)                                                               \
    int l2a_launch_mfma_##NT_##_##TPW_(int ot, int kg0, int gact, const L2AKParams* p, unsigned grid, \
                                       int smem, hipStream_t stream);
L2A_DECL_LAUNCH(1, 2) L2A_DECL_LAUNCH(1, 4) L2A_DECL_LAUNCH(1, 8)
L2A_DECL_LAUNCH(2, 2) L2A_DECL_LAUNCH(2, 4)
// member-fan instances (p->split == 3: one workgroup per candidate tile and ensemble member), NT = 1
L2A_DECL_LAUNCH(fan_1, 2) L2A_DECL_LAUNCH(fan_1, 4) L2A_DECL_LAUNCH(fan_1, 8) L2A_DECL_LAUNCH(fan_2, 8)
// whole-tiles-only instances (no exchange, no half member), two tiles per workgroup at width 512: double rounds
L2A_DECL_LAUNCH(whole_2, 8) L2A_DECL_LAUNCH(whole_1, 8)
#undef L2A_DECL_LAUNCH

// geo: 0 = general instances (whole tiles, tile split, tail split), 1 = member fan, 2 = whole tiles only
inline int l2a_launch_mfma(int nt, int tpw, int ot, int kg0, int gact, int geo, const L2AKParams* p, unsigned grid,
                           int smem, hipStream_t stream) {
    static_assert(L2A_NW == 4, "instances are generated for 4-wave workgroups (TPW = H / 64)");
    if (geo == 2) {
        if (nt == 2 && tpw == 8) return l2a_launch_mfma_whole_2_8(ot, kg0, gact, p, grid, smem, stream);
        if (nt == 1 && tpw == 8) return l2a_launch_mfma_whole_1_8(ot, kg0, gact, p, grid, smem, stream);
        return -100;
    }
    if (geo == 1) {
        if (nt == 1 && tpw == 2) return l2a_launch_mfma_fan_1_2(ot, kg0, gact, p, grid, smem, stream);
        if (nt == 1 && tpw == 4) return l2a_launch_mfma_fan_1_4(ot, kg0, gact, p, grid, smem, stream);
        if (nt == 1 && tpw == 8) return l2a_launch_mfma_fan_1_8(ot, kg0, gact, p, grid, smem, stream);
        if (nt == 2 && tpw == 8) return l2a_launch_mfma_fan_2_8(ot, kg0, gact, p, grid, smem, stream);
        return -100;
    }
    if (nt == 1 && tpw == 2) return l2a_launch_mfma_1_2(ot, kg0, gact, p, grid, smem, stream);
    if (nt == 1 && tpw == 4) return l2a_launch_mfma_1_4(ot, kg0, gact, p, grid, smem, stream);
    if (nt == 1 && tpw == 8) return l2a_launch_mfma_1_8(ot, kg0, gact, p, grid, smem, stream);
    if (nt == 2 && tpw == 2) return l2a_launch_mfma_2_2(ot, kg0, gact, p, grid, smem, stream);
    if (nt == 2 && tpw == 4) return l2a_launch_mfma_2_4(ot, kg0, gact, p, grid, smem, stream);
    return -100;
}
