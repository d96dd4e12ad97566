// l2a_mfma.h - the fp32 MFMA rollout kernel (gfx950 / CDNA4 only).  Included by l2a_mfma_inst.hip
// (one translation unit per (NT, TPW); launchers declared in l2a_mfma_launch.h).
//
// Workgroup = L2A_NW waves (4 = one per SIMD) owning NT tiles of 16 candidates of one env for the
// whole horizon.  Tiles never talk to each other; when a plan has few tiles, TWO workgroups share a
// tile and swap partial sums once per horizon step (see "Tile split").  One atomicMax per tile at
// the end.
//
// Data distribution: lane (j = l & 15, qq = l >> 4) holds, for candidate j of a tile, features
// 16c + 4qq + 0..3 of every 16-feature tile c - for the state, the deltas and the hidden
// activations alike.  That is at once the D-fragment layout of v_mfma_f32_16x16x4_f32 (computed
// as D = W^T-tile x activations) and, thanks to the k permutation baked into the packed weights
// (l2a_kernels.h), the B-fragment layout of the next layer: layers chain through LDS with one
// 16-byte write and one 16-byte read per lane and tile, no transposes, no bank conflicts.
//
// Per horizon step the sets a workgroup runs are processed in BATCHES of p.lb sets (l2a_set_batch; as many as the CU's
// LDS holds side by side when the model has two hidden layers, else one):
//   layer 0      : of every set of the batch, back to back - wave w computes hidden tiles [w*TPW, (w+1)*TPW); B =
//                  normalised [obs | act] built in registers from the state fragment; K = in_dim (KG0 k-groups);
//                  result -> LDS region of the set.  ONE barrier.
//   hidden GEMMs : set by set, same tile ownership; B fragments (activations) come from LDS; A fragments
//                  (weights, 1 KiB per wave-level buffer load) stream from L2 into registers, two
//                  k-groups ahead of the MFMAs, one load issued per four MFMAs
//   output layer : fused into the last hidden layer - hidden tile c IS k-group c of the output
//                  layer and its D fragment IS the B fragment, so every wave multiplies its own
//                  registers; chunk partials go to the set's LDS slot; the next set's GEMM follows without a barrier
//   reduce       : ONE barrier per batch, then every wave sums every set's chunk partials in the canonical order,
//                  so all waves hold bit-identical copies of state / return
//   every phase issues the FIRST operands of the next phase before it ends (and the epilogue's bias vectors under its
//   last MFMAs), so no phase starts with an exposed L2 or LDS round trip
// LDS: max(2, lb) activation regions of sa_elems f32x4 | chunk partials of a batch (lb > 1) | per-set constants
// (normalisation vectors and all biases) | 2 KiB per (candidate tile, obs tile) of exchange staging.
//
// Fixed summation order (every launch geometry and every batch size gives the same bits): sets are summed as (group A)
// + (group B), A = first ceil(E/2) sets; the output layer's K reduction is cut into 2 * L2A_NW
// chunks of TPW / 2 k-groups, summed as the balanced tree ((c0+c1)+(c2+c3)) + ((c4+c5)+(c6+c7)).
//
// Tile split (p.split): 1 = workgroup 0 runs group A, workgroup 1 group B; 2 = additionally the
// last set of group A is SHARED - both workgroups run it as a "half member": layer 0 and the inner
// hidden layers in full, the last hidden layer and the output layer for one half of the hidden
// tiles each (this balances odd ensembles and lets a single model use two CUs).  Partial sums
// travel through 16-byte self-validating {tag, v, tag, v} granules in global memory: write-through
// (sc1) stores, sc1 loads, no flag and no fence (cdna_hip_programming.md G16 recipe R2: the 8-byte {tag, value} pair is
// the unit whose single-copy atomicity the recipe rests on - a {tag, v0, v1, v2} granule would need 16-byte atomicity,
// which nothing documents).  While the records travel the waves take over the next step's actions.  Spins are bounded
// (status word, never a hang).
// Member fan (FAN instances, p.split == 3; l2a_set_fan): a mean ensemble of E = 3 .. 8 sets whose plan has so few tiles that
// E x tiles <= CUs (one rank's 500-candidate shard of BASELINE config 5: 32 tiles) runs E workgroups per tile, workgroup g
// = member g as a full member and nothing else.  Per horizon step every workgroup publishes its member's term
// (denormalised delta, the same {tag, v, tag, v} granules, one region per (tile, member, step parity)), wave w collects
// partners w, w + 4, .. and every wave adds all E terms from LDS in the unsplit launch's order, (0 + t_0 + .. + t_eh-1) +
// (0 + t_eh + .. + t_E-1): same bits.  The half-member code is compiled out of these instances (330 instead of 442 VGPRs
// on the HalfCheetah shape).  Config-5 shard, n = 500, h = 30, E = 5: 160 workgroups of one set, 0.597 ms, against the
// tile split's 64 workgroups of 2.5 sets, 1.399 ms (rocprofv3, profiles/r06_c5shard_kernel_stats.csv).
// Tail split (p.split_from >= 0): in a multi-round plan whose last round would fill under half of the
// chip only the left-over tiles are shared; their workgroup pairs are dispatched last, back to back.
//
// Template parameters: NT candidate tiles per workgroup (1|2), TPW hidden tiles per wave (hidden
// width = 16 * L2A_NW * TPW), OT = ceil(obs_dim / 16), KG0 = ceil((obs_dim + act_dim) / 16), GACT =
// generic activation functions (false: relu / identity only, branch-free), K0L = layer-0 MFMAs of the last input
// k-group that can see non-zero operands (4 unless in_dim mod 16 is 1..3), N1 = the model has ONE hidden layer (its
// own instances: compile-time, so that neither path keeps the other's operand registers alive - with the run-time
// test the register allocator carried the GEMM's prefetch registers through the whole kernel, 475 instead of ~410
// VGPRs on the HalfCheetah instance), O4 = the last obs tile has at most four live units (HalfCheetah: 20 = 16 + 4):
// it is computed with the 4x4x1 MFMA instead of a 16-row tile that is three quarters padding (l2a_out_phase).
#pragma once

#include <type_traits>

#include "l2a_kernels.h"

// Early exchange of the tile split (phase C of the kernel): -DL2A_XEARLY=0 builds the instances without it (A/B, tools/build_variant.py).
#ifndef L2A_XEARLY
#define L2A_XEARLY 1
#endif

// Waves per workgroup.  4 = one wave per SIMD: a lone wave issues its MFMAs back to back (32
// cycles each) with loads / LDS reads / address math slotted in between, whereas two waves on
// one SIMD were measured (tools/timeline.py) to leave the matrix pipe ~25 % idle - the older
// wave is paced to every other slot and the younger one catches only a third of the rest.
#ifndef L2A_NW
#define L2A_NW 4
#endif

// Weight fragments are fetched with raw buffer loads: address = descriptor base (SGPRs, wave
// uniform) + per-lane VGPR offset + SGPR offset + 12-bit immediate.  All address arithmetic of the
// hot loop then lives in one SALU add per iteration instead of two VALU adds per load (the VALU
// slots between MFMAs are not free: tools/timeline.py measured ~39 instead of 32 cycles per MFMA
// with flat 64-bit addressing).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t l2a_rsrc(const void* base, long long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 l2a_ldw(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}

// Activation.  GACT == false covers relu and identity branch-free as max(x, floor) with floor = 0
// or -inf: the generic path (tanh / sigmoid / swish) inlines ~10 KiB of code per call site, and
// jumping around it in every epilogue cost ~2k cycles per layer in instruction fetch.
template <bool GACT>
__device__ __forceinline__ f32x4 l2a_actv(f32x4 v, int kind, float floor) {
    if (GACT) return l2a_act4(v, kind);
    v.x = fmaxf(v.x, floor); v.y = fmaxf(v.y, floor);
    v.z = fmaxf(v.z, floor); v.w = fmaxf(v.w, floor);
    return v;
}

// x[l] + x[l ^ 16] and x[l] + x[l ^ 32] over the wave with the gfx950 lane-swap instructions (VALU rate, no LDS
// crossbar round trip like ds_bpermute): v_permlane16_swap exchanges the odd rows of its first operand with the even
// rows of the second, v_permlane32_swap the upper half of the first with the lower half of the second; called on two
// copies of x the two results are {own, other} in one half / row and {other, own} in the other - their sum is the same
// bits in every lane.
__device__ __forceinline__ float l2a_sum_xor16(float x) {
    const unsigned int u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float l2a_sum_xor32(float x) {
    const unsigned int u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// v_mfma_f32_4x4x1_16b_f32: sixteen independent 4x4 outer products (one k each) per instruction, block b on lanes
// 4b .. 4b + 3: lane 4b + i supplies A_b[i], lane 4b + j supplies B_b[j] and holds D_b[i][j] in result register i.
// Two passes (8 clocks) instead of the eight of the 16x16x4 tile.  Used for an obs tile of which only four units
// are alive (O4, below).
#define L2A_MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 0, 0, 0)

// Make `v` opaque to the optimiser at this point (no instructions emitted).  Used on the
// pre-loop fills of the software-pipeline registers: without it InstCombine folds the
// loop-carried phi(load, load) into load(phi(addr)) and the prefetch distance collapses to 0.
#define L2A_OPAQUE(v) asm volatile("" : "+v"(v))

// Phase timeline for tools/timeline.py: every wave of candidate tile 0 stamps the shader clock at
// phase boundaries, dbg[(((grp * h + t) * 8 + e) * 8 + wave) * 16 + slot].  One uniform branch per stamp - which
// also ends a basic block, i.e. keeps the scheduler from moving instructions across phase boundaries (measured:
// 0.6 % of a config-2 plan, 2.7 % of a config-1 plan).  Hence only in builds with -DL2A_TIMELINE
// (tools/build_variant.py timeline -DL2A_TIMELINE); the product library carries no stamps.
#ifndef L2A_TIMELINE
#define L2A_TS(slot)
#else
#define L2A_TS(slot)                                                                        \
    if (p.dbg && pairid == 0 && e < 8 && grp < 2) {                                         \
        unsigned long long ts_;                                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_) : : "memory");       \
        if (lane == 0) p.dbg[(((long long)(grp * p.h + t) * 8 + e) * 8 + wave) * 16 + (slot)] = ts_; \
    }
#endif

// ------------------------------------------------------------------------------------------
// Hidden->hidden layer GEMM for this wave's TPW output tiles (accumulators only; the caller
// owns the epilogue).
//
// Software pipeline over the HT k-groups, prefetch distance 2 groups, 4 register buffers in
// rotation (A, B, C, D <-> group index mod 4) so that a buffer is refilled only after its last
// use: no loop-carried register copies.  `sched_barrier(0)` pins each refill inside its stage
// (hipcc otherwise sinks the loads next to their consumers) and the incoming buffers are made
// opaque so InstCombine cannot fold phi(load, load) into load(phi(addr)) and collapse the
// prefetch distance.  aA / aB arrive PRELOADED with k-groups 0 and 1 (issued by the previous
// phase, before its barrier).  The last two refill slots of the loop - which would otherwise
// reload the final k-group - fetch the NEXT phase's first operands instead:
//   LAST == false : k-groups 0 / 1 of the next hidden layer (wnext) back into aA / aB
//   LAST == true  : the output layer's A fragments for this wave's k-groups into pfO
// ------------------------------------------------------------------------------------------
#define L2A_STAGE_MFMA(CA, CB)                                                             \
    _Pragma("unroll") for (int ii = 0; ii < 4; ++ii)                                       \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                  \
            _Pragma("unroll") for (int tt = TW - 1; tt >= 0; --tt)   /* last-loaded tile first: one vmcnt wait per stage */ \
                acc[nt][tt] = L2A_MFMA(CA[tt][ii], CB[nt][ii], acc[nt][tt]);

#define L2A_STAGE(CA, CB, FA, FB, SOFF, IMM, GF)                                           \
    {                                                                                      \
        _Pragma("unroll") for (int tt = 0; tt < TW; ++tt)                                  \
            FA[tt] = l2a_ldw(rs, voff[tt] + (IMM), (SOFF));                                \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                  \
            FB[nt] = hin[(nt * HT + (GF)) * 64 + lane];                                    \
        L2A_STAGE_MFMA(CA, CB)                                                             \
        /* interleave: one weight load per 4 NT MFMAs (8 back-to-back VMEM issues stall the pipe) */ \
        __builtin_amdgcn_sched_group_barrier(0x100, NT, 0);                                \
        _Pragma("unroll") for (int tt = 0; tt < TW; ++tt) {                                \
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT, 0);                        \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                             \
        }                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                 \
    }

// olane : per obs tile, the byte offset of the lane whose output-layer fragment this lane fetches (16 lane; see O4)
// bias_lds : this lane's bias slice of the wave's first output tile (LDS, + 16 floats per tile) -> bias[0 .. TW)
// TW    : output tiles this wave computes here (TPW, or TPW / 2 for a half member)
// rs    : descriptor of this layer's packed weights [tile][k-group][64 lanes][4]
// rsn   : descriptor of the next phase's packed weights (next hidden layer, or the output layer)
// voff  : byte offset of this lane in the layer for each of its tiles, 16 lane + tile * HT * 1024
// LAST == false: voffn / twn = offsets / count of the tiles this wave computes in the NEXT layer
// LAST == true : tile0 = global index of this wave's first tile (selects the output fragments)
// Output-layer fragments fetched a phase ahead: the first L2A_PFT tiles only (the rest are loaded
// when the output phase starts and land under the MFMAs of the first tiles) - register budget.
#define L2A_PFT_MAX 6
#define L2A_PFT(TW_, OT_) ((OT_) <= 2 ? ((TW_) < L2A_PFT_MAX ? (TW_) : L2A_PFT_MAX) : ((TW_) < 2 ? (TW_) : 2))

template <int NT, int TW, int TPW, int OT, bool LAST>
__device__ __forceinline__ void l2a_hidden_gemm(__amdgpu_buffer_rsrc_t rs, __amdgpu_buffer_rsrc_t rsn,
                                                const int (&voff)[TPW], const int (&voffn)[TPW], int twn,
                                                int tile0, const f32x4* hin,
                                                f32x4 (&aA)[TPW], f32x4 (&aB)[TPW],
                                                f32x4 (&pfO)[TPW][OT], f32x4 (&acc)[NT][TPW], int lane,
                                                const float* bias_lds, f32x4 (&bias)[TPW], const int (&olane)[OT]) {
    constexpr int HT = L2A_NW * TPW;
    static_assert(HT % 4 == 0, "the k-group pipeline is unrolled by 4");
    f32x4 aC[TW], aD[TW], bA[NT], bB[NT], bC[NT], bD[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        bA[nt] = hin[(nt * HT + 0) * 64 + lane];
        bB[nt] = hin[(nt * HT + 1) * 64 + lane];
#pragma unroll
        for (int tt = 0; tt < TW; ++tt) acc[nt][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // (the weight buffers come from buffer-load intrinsics, which InstCombine leaves alone; only the
    // LDS reads are plain loads that it would sink into the loop)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { L2A_OPAQUE(bA[nt]); L2A_OPAQUE(bB[nt]); }
    __builtin_amdgcn_sched_barrier(0);

#pragma unroll 1
    for (int g = 0; g < HT - 4; g += 4) {
        const int soff = (g + 2) * 1024;      // byte offset of k-group g + 2 inside a tile
        L2A_STAGE(aA, bA, aC, bC, soff, 0, g + 2)
        L2A_STAGE(aB, bB, aD, bD, soff, 1024, g + 3)
        L2A_STAGE(aC, bC, aA, bA, soff, 2048, g + 4)
        L2A_STAGE(aD, bD, aB, bB, soff, 3072, g + 5)
    }
    // ---- peeled last iteration (k-groups HT-4 .. HT-1) ------------------------------------
    L2A_STAGE(aA, bA, aC, bC, (HT - 2) * 1024, 0, HT - 2)
    L2A_STAGE(aB, bB, aD, bD, (HT - 2) * 1024, 1024, HT - 1)
    {   // stage 2: consume C; aA is free -> next phase's first operands
        if (LAST) {     // output layer: A fragment (obs tile c, k-group = hidden tile tile0 + tt)
#pragma unroll
            for (int tt = 0; tt < L2A_PFT(TW, OT); ++tt)
#pragma unroll
                for (int c = 0; c < OT; ++c)
                    if (((tt * OT + c) & 1) == 0)
                        pfO[tt][c] = l2a_ldw(rsn, olane[c] + (tile0 + tt) * 1024, c * HT * 1024);
        } else {
#pragma unroll
            for (int tt = 0; tt < TPW; ++tt)
                if (tt < twn) aA[tt] = l2a_ldw(rsn, voffn[tt], 0);
        }
        L2A_STAGE_MFMA(aC, bC)
        __builtin_amdgcn_sched_barrier(0);
    }
    {   // stage 3: consume D; aB is free.  The epilogue's bias vectors (LDS) are requested here, under the last MFMAs:
        // read where they are used they cost one exposed LDS round trip per tile (~1k clocks per GEMM, timeline r03)
#pragma unroll
        for (int tt = 0; tt < TW; ++tt) bias[tt] = *reinterpret_cast<const f32x4*>(bias_lds + 16 * tt);
        if (LAST) {
#pragma unroll
            for (int tt = 0; tt < L2A_PFT(TW, OT); ++tt)
#pragma unroll
                for (int c = 0; c < OT; ++c)
                    if (((tt * OT + c) & 1) == 1)
                        pfO[tt][c] = l2a_ldw(rsn, olane[c] + (tile0 + tt) * 1024, c * HT * 1024);
        } else {
#pragma unroll
            for (int tt = 0; tt < TPW; ++tt)
                if (tt < twn) aB[tt] = l2a_ldw(rsn, voffn[tt] + 1024, 0);
        }
        L2A_STAGE_MFMA(aD, bD)
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Output layer for the TW hidden tiles whose activations `hreg` this wave holds in registers
// (hidden tile == k-group of the output layer, D fragment == B fragment: no LDS round trip).
// The K reduction is cut into chunks of TPW / 2 k-groups, each its own MFMA chain, written to
// pbuf[(chunk * NT + nt) * OT + c]; a full member's wave owns chunks 2w and 2w + 1, a half
// member's wave owns chunk w of its workgroup's half.  `prefetch` issues the NPF operand loads of
// the phase that follows (the next set's hidden GEMM inside a batch, else the next layer 0), so
// that they are in flight across the barrier / under the MFMAs here.
//
// O4: only four units of the LAST obs tile are alive (HalfCheetah: 20 = 16 + 4).  Its sixteen-row MFMA tile would
// spend three quarters of its rows on padding; instead the tile is computed with the 4x4x1 MFMA: block b = lane / 4
// multiplies the four live units by the four candidates 4 (b & 3) .. + 3 for the hidden unit 16 tile + 4 (b >> 2) +
// ii - the B operand is the hidden fragment as it stands, the A operand the packed output fragment of lane
// (lane & 48) | (lane & 3) (same packed array, another lane's 16 bytes: `olane`).  A lane's accumulator then holds the
// sum over ITS quarter of the hidden units (qq = lane >> 4) for candidate lane & 15; the four quarters are added
// where the chunk partials are reduced (xor-16 / xor-32 lane sums, same order in every launch geometry).
template <int NT, int TW, int TPW, int OT, int NPF, bool O4, class PF>
__device__ __forceinline__ void l2a_out_phase(const f32x4 (&hreg)[NT][TPW], f32x4 (&pfO)[TPW][OT],
                                              __amdgpu_buffer_rsrc_t rs_out, int tile0, PF prefetch,
                                              f32x4* pbuf, int chunk0, int lane, const int (&olane)[OT]) {
    constexpr int HT = L2A_NW * TPW;
    constexpr int CS = TPW / 2;         // tiles per chunk
    constexpr int NCH = TW / CS;        // chunks this wave owns (2 = full member, 1 = half member)
    f32x4 a[TW][OT];
#pragma unroll
    for (int tt = 0; tt < TW; ++tt)
#pragma unroll
        for (int c = 0; c < OT; ++c)
            a[tt][c] = (tt < L2A_PFT(TW, OT)) ? pfO[tt][c]
                                              : l2a_ldw(rs_out, olane[c] + (tile0 + tt) * 1024, c * HT * 1024);
    prefetch();
    f32x4 acc[NCH][NT][OT];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int c = 0; c < OT; ++c) acc[ch][nt][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t2 = 0; t2 < CS; ++t2)
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int c = 0; c < OT; ++c)
                        acc[ch][nt][c] = (O4 && c == OT - 1)
                            ? L2A_MFMA4(a[ch * CS + t2][c][ii], hreg[nt][ch * CS + t2][ii], acc[ch][nt][c])
                            : L2A_MFMA(a[ch * CS + t2][c][ii], hreg[nt][ch * CS + t2][ii], acc[ch][nt][c]);
    // issue order hint: the late output fragments first (one per 2 MFMAs), then the prefetch loads
    // spread over the remaining MFMAs - 8 VMEM issues in a row would idle the matrix pipe
    {
        constexpr int NLATE = (TW - L2A_PFT(TW, OT)) * OT;
        constexpr int NMFMA = TW * 4 * NT * OT;
#pragma unroll
        for (int i = 0; i < NLATE; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
        constexpr int PER = (NMFMA - 2 * NLATE) / NPF > 0 ? (NMFMA - 2 * NLATE) / NPF : 1;
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
    }
    // one partial per wave: a full member adds its two chunks here (c_2w + c_2w+1, the first level of the canonical tree)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int c = 0; c < OT; ++c)
            pbuf[((chunk0 * NT + nt) * OT + c) * 64 + lane] = (NCH == 2) ? acc[0][nt][c] + acc[NCH - 1][nt][c] : acc[0][nt][c];
}

template <int NT, int TPW, int OT, int KG0, bool GACT, int K0L = 4, bool N1 = false, bool O4 = false, bool FAN = false, bool WHOLE = false>
__global__ void __launch_bounds__(64 * L2A_NW) l2a_rollout_mfma_k(const L2AKParams p) {
    static_assert(!(FAN && WHOLE), "an instance is general, member fan or whole-tiles-only");
    constexpr bool NOHALF = FAN || WHOLE;   // instances without the half-member paths (they cost ~110 VGPRs at width 512)
    constexpr int HT = L2A_NW * TPW;
    constexpr int TH = TPW / 2;         // tiles per wave of a half member
    static_assert(TPW % 2 == 0, "half members split a wave's tiles in two");
    // Per-set constants cached in LDS: [in_mu 16 KG0][in_inv 16 KG0][out_mu 16 OT][out_sd 16 OT]
    // [out_bias 16 OT][hidden biases n_hidden x H] - p.cst_set floats per weight set, one slot per set of this
    // workgroup's sequence (slot i = the i-th set it runs).
    constexpr int CST_BOUT = 32 * KG0 + 32 * OT;
    constexpr int CST_BHID = CST_BOUT + 16 * OT;
    constexpr int PS = L2A_NW * NT * OT * 64;           // f32x4 of output-layer partials per set: one per wave
    const int NRM_SET = p.cst_set;
    // LDS: activation regions [max(2, LB)][sa_elems] | (LB > 1: chunk partials of a batch) | constants | exchange staging.
    // LB = sets per batch (p.lb).  LB == 1: one set at a time, two regions used in turn (layer in / layer out, the
    // output partials in whichever is free).  LB > 1 (two hidden layers only): layer 0 of LB sets -> region j each,
    // ONE barrier, then every set's hidden GEMM + output layer back to back (partials -> their own area), ONE
    // barrier, all reduces - two barriers per batch instead of two per set, and the MFMA-light phases of
    // consecutive sets overlap each other's latencies.  Same arithmetic in the same order: bit-identical.
    const int LB = p.lb;
    extern __shared__ __attribute__((aligned(16))) char l2a_smem[];
    f32x4* buf0 = reinterpret_cast<f32x4*>(l2a_smem);
    f32x4* buf1 = buf0 + p.sa_elems;
    f32x4* pbase = buf0 + (LB > 1 ? LB : 2) * p.sa_elems;
    float* nrm = reinterpret_cast<float*>(buf0 + p.cst_off);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int jc = lane & 15;
    const int qq = lane >> 4;
    const int c0 = wave * TPW;
    int olane[OT];              // whose output-layer fragment this lane fetches, per obs tile (O4: see l2a_out_phase)
#pragma unroll
    for (int c = 0; c < OT; ++c) olane[c] = ((O4 && c == OT - 1) ? ((lane & 48) | (lane & 3)) : lane) * 16;

    // Geometry.  Uniform launches (p.split_from < 0): every tile is run by 1 (split 0) or 2 workgroups, ids
    // remapped XCD-aware.  Tail split (p.split_from >= 0): the first split_from hardware workgroups run whole
    // tiles, the remaining ones - dispatched last - are pairs that share the left-over tiles of a multi-round
    // plan (e.g. 625 tiles on 256 CUs: 512 whole + 113 shared instead of a third, 44 %-filled round).
    const int tail = p.split_from >= 0;
    const int in_tail = tail && (int)blockIdx.x >= p.split_from;
    const int split = FAN ? 3 : WHOLE ? 0 : (tail ? (in_tail ? p.split : 0) : p.split);   // this workgroup's split mode (3 = member fan)
    const int n_tiles = p.m * p.tiles_per_env;
    const int n_pairs = tail ? n_tiles - p.split_from : n_tiles;       // tiles that are shared by two workgroups
    int bid, grp, lpair, pairid;
    if (!tail && p.pl_units > 0) {
        // Every XCD (own 4 MB L2) serves ONE unit - workgroups that stream the same weights: group A of a split ensemble on
        // XCDs 0-3 and group B on 4-7, each environment of a per-block plan on its own XCD(s).  The contiguous remap cannot
        // do that when the counts do not divide: config 2 (125 tiles) left one workgroup of group B alone with its sets in
        // an L2 of group A, and its pair ended the launch 1 % late; config 3b (5 environments x 32 tiles, 2.3 MB of weights
        // each) had four XCDs thrash between two environments' sets, 5 % slower than the other four.  The grid is padded
        // (host: launch_rollout); hardware workgroup i sits on XCD i % 8, and the spare ones return here.
        const int hx = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3;
        const int xcd = hx;
        const int f = p.pl_f, wide = p.pl_r * (f + 1);
        int u, k, xu;
        if (xcd < wide) { u = xcd / (f + 1); k = xcd - u * (f + 1); xu = f + 1; }
        else { const int y = xcd - wide; u = y / f; k = y - u * f; u += p.pl_r; xu = f; }
        const int su = (p.pl_w + xu - 1) / xu;              // workgroups of this unit per XCD
        const int j = k * su + idx;                         // member of the unit
        if (idx >= su || j >= p.pl_w) return;
        if (p.mode == L2A_MODE_PER_BLOCK) {                 // unit = environment: its tiles, both workgroups of each
            grp = split ? j / p.tiles_per_env : 0;
            lpair = u * p.tiles_per_env + (j - grp * p.tiles_per_env);
        } else {                                            // unit = ensemble group
            grp = u;
            lpair = j;
        }
        bid = grp * n_pairs + lpair;
        pairid = lpair;
    } else if (!tail) {
        bid = l2a_logical_wg(blockIdx.x, gridDim.x);
        grp = split ? (bid / n_pairs) : 0;                  // which of the two workgroups of a tile
        lpair = bid - grp * n_pairs;
        pairid = lpair;
    } else if (!in_tail) {
        bid = l2a_logical_wg(blockIdx.x, p.split_from);
        grp = 0; lpair = 0; pairid = bid;
    } else {
        bid = (int)blockIdx.x - p.split_from;
        grp = bid & 1; lpair = bid >> 1;                    // partners are dispatched back to back
        pairid = p.split_from + lpair;
    }
#if defined(L2A_TIMELINE) || defined(L2A_WGREC)
    unsigned long long wg_t0_, wg_r0_;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(wg_t0_), "=s"(wg_r0_) : : "memory");
#endif
    const int env = pairid / p.tiles_per_env;
    const int tb = pairid - env * p.tiles_per_env;
    const int R = p.m * p.n;
    const int obs_dim = p.obs_dim, act_dim = p.act_dim;
    // N1: the model has ONE hidden layer (its own instances: no hidden->hidden GEMM, layer 0 feeds the output layer from
    // registers) - a compile-time fact, so that neither path keeps the other's operand registers alive
    const int n_hidden = N1 ? 1 : p.n_hidden;

    int cand[NT], row[NT];
    bool valid[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        cand[nt] = p.c_lo + tb * (16 * NT) + nt * 16 + jc;     // (a launch covers candidates [c_lo, c_hi) of every env)
        valid[nt] = cand[nt] < p.c_hi;
        row[nt] = env * p.n + (valid[nt] ? cand[nt] : p.c_hi - 1);
    }

    // Which sets this workgroup runs (split: 0 = all; 1 = group A | group B; 2 = as 1, but the
    // last set of group A is SHARED: both workgroups run it as a "half member" - layer 0 and the
    // inner hidden layers in full, the last hidden layer and the output layer for one half of the
    // hidden tiles each - which balances odd ensembles and lets a single model use two CUs).
    // Both workgroups run their full sets first and the shared one last.
    const bool per_block = (p.mode == L2A_MODE_PER_BLOCK);
    const int e_loop = (p.mode == L2A_MODE_MEAN) ? p.n_sets : 1;
    const int e_half = (e_loop + 1) >> 1;                   // group A = [0, e_half), B = [e_half, e_loop)
    // Member fan (FAN instances, split == 3): workgroup `grp` of a tile runs set `grp` alone, as a full member.
    const int e_shared = (!NOHALF && split == 2) ? e_half - 1 : -1;
    const int n_full = FAN ? 1 : (!split ? e_loop : (grp == 0 ? e_half - (split == 2 ? 1 : 0) : e_loop - e_half));
    const int n_seq = FAN ? 1 : n_full + (split == 2 ? 1 : 0);
    const int full0 = FAN ? grp : ((split && grp == 1) ? e_half : 0);       // first full set of this workgroup
    auto seq = [&](int i) { return (i < n_full) ? full0 + i : e_shared; };
    auto set_base = [&](int e) { return p.wblk + (long long)(per_block ? env : e) * p.set_stride; };

    // ---- constants of every set this workgroup uses -> LDS (slot i = set seq(i)) ---------------
    f32x4* xlds = reinterpret_cast<f32x4*>(nrm + p.n_cst * NRM_SET);   // [2][NT * OT][64], split only
    for (int i = tid; i < n_seq * NRM_SET; i += 64 * L2A_NW) {
        const int s = i / NRM_SET;
        const int o = i - s * NRM_SET;
        const float* src = set_base(seq(s));
        float v;
        if (o < CST_BOUT) v = src[p.nm_off + o];
        else if (o < CST_BHID) v = src[p.pk_bout + (o - CST_BOUT)];
        else {
            const int l = (o - CST_BHID) / (16 * HT);
            v = src[p.raw_b[l] + (o - CST_BHID - l * 16 * HT)];
        }
        nrm[i] = v;
    }

    // ---- state fragment -------------------------------------------------------------------
    f32x4 st[NT][OT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float* orow = p.obs0 + (p.obs_per_row ? (long long)row[nt] : (long long)env) * obs_dim;
#pragma unroll
        for (int c = 0; c < OT; ++c)
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int dim = 16 * c + 4 * qq + ii;
                const float v = orow[dim < obs_dim ? dim : obs_dim - 1];   // branch-free, in bounds
                st[nt][c][ii] = (dim < obs_dim) ? v : 0.0f;
            }
    }
    __syncthreads();

    // ---- actions occupy input k-groups ga0 and ga0 + 1 (act_dim <= 16) ----------------------
    // Fetched with raw buffer loads from a per-step descriptor (base = the step's [R, act_dim] slab, SALU only):
    // the lane's eight 32-bit byte offsets are loop invariants, and the slots of an input k-group that hold no
    // action (state features, padding) point past the slab - the bounds check returns 0.0 for them, so there
    // is neither a lane mask nor a select per element.
    const int ga0 = obs_dim >> 4;
    f32x4 av_next[NT][2];
    int aoff[NT][2][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int ka = 16 * (ga0 + s) + 4 * qq + ii - obs_dim;
                const bool in = (ka >= 0) && (ka < act_dim);
                aoff[nt][s][ii] = in ? (row[nt] * act_dim + ka) * 4 : 0x7ffffff0;
            }
    const long long a_step = (long long)R * act_dim;            // floats per horizon step
    auto load_actions = [&](int t, f32x4 (&dst)[NT][2]) {
        const __amdgpu_buffer_rsrc_t ars = l2a_rsrc(p.actions + (long long)t * a_step, a_step * 4);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int ii = 0; ii < 4; ++ii)
                    dst[nt][s][ii] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ars, aoff[nt][s][ii], 0, 0));
    };
    // The actions of step t + 1 are taken over (and those of step t + 2 requested) while step t's exchange
    // travels: nothing after the last layer 0 of a step reads `av`.
    f32x4 av[NT][2];
    float asq[NT];
    auto take_actions = [&](int t_fetch) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            av[nt][0] = av_next[nt][0];
            av[nt][1] = av_next[nt][1];
            float s = 0.0f;
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                s = fmaf(av[nt][0][ii], av[nt][0][ii], s);
                s = fmaf(av[nt][1][ii], av[nt][1][ii], s);
            }
            asq[nt] = s;
        }
        load_actions((t_fetch < p.h) ? t_fetch : p.h - 1, av_next);
    };
    load_actions(0, av_next);
    take_actions(1);

    float ret[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) ret[nt] = p.ret_in ? p.ret_in[(long long)env * p.n + (valid[nt] ? cand[nt] : p.c_hi - 1)] : 0.0f;

    f32x4* hcur = buf0;
    f32x4* hoth = buf1;
    const float e_count = (float)e_loop;
    const float e_inv = 1.0f / e_count;
    double disc_pow = p.disc0;  // discount ** t, carried in float64 like the reference (:126)
    unsigned int spin_left = p.spin_limit;      // exchange polls this workgroup may still spend (whole launch)
    const unsigned int xtag0 = p.xtag;
    const __amdgpu_buffer_rsrc_t xrs = l2a_rsrc(p.xbuf, FAN ? (long long)n_pairs * e_loop * 2 * (NT * OT * 2 * 64 * 16)
                                                        : split ? (long long)n_pairs * 8 * (NT * OT * 2 * 64 * 16) : 16);

    // Operands every phase receives preloaded from the phase before it (issued ahead of the
    // barrier that separates them, so a phase never starts with an exposed L2 round trip).
    // (every layer-0 k-group a phase ahead was measured neutral for 21 more registers: profiles/r03_ab_kernel_variants.jsonl)
    // Layer-0 k-groups fetched a phase ahead: the first one; in the fan instances (110 VGPRs below the tile-split ones) both
    // k-groups of an input of at most 32 features - config-5 shard 0.6008 -> 0.5932 ms (round 6 A/B, tools/ab_fan.sh; the same
    // prefetch was neutral in the tile-split instances at 442 VGPRs: profiles/r03_ab_kernel_variants.jsonl)
    constexpr int KG0P = (FAN && KG0 <= 2) ? KG0 : 1;
    f32x4 pfL0[KG0P][TPW];      // layer-0 A fragments of the upcoming (step, set)
    f32x4 pfA[TPW], pfB[TPW];   // k-groups 0 / 1 of the upcoming hidden->hidden layer
    f32x4 pfO[TPW][OT];         // output-layer A fragments of this wave's k-groups
    // this lane's byte offset of its tiles in a hidden matrix (full member / half member) and in layer 0
    int voff[TPW], voffh[TPW], voff0[TPW];
    const int tile0h = grp * (HT / 2) + wave * TH;          // first tile of this wave in a half member
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
        voff[tt] = lane * 16 + (c0 + tt) * HT * 1024;
        voffh[tt] = lane * 16 + (tile0h + (tt < TH ? tt : 0)) * HT * 1024;
        voff0[tt] = lane * 16 + (c0 + tt) * KG0 * 1024;
    }
    const long long w0_bytes = (long long)HT * KG0 * 1024, wm_bytes = (long long)HT * HT * 1024,
                    wo_bytes = (long long)OT * HT * 1024;
    {
        const __amdgpu_buffer_rsrc_t r0 = l2a_rsrc(set_base(seq(0)) + p.pk_w0, w0_bytes);
#pragma unroll
        for (int g = 0; g < KG0P; ++g)
#pragma unroll
            for (int tt = 0; tt < TPW; ++tt) pfL0[g][tt] = l2a_ldw(r0, voff0[tt] + g * 1024, 0);
    }

    for (int t = 0; t < p.h; ++t) {
        // dsum: finished group; dgrp: group being summed; qsh: this workgroup's half of the shared
        // set's raw output sum (split == 2)
        f32x4 dsum[NT][OT], dgrp[NT][OT], qsh[NT][OT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int c = 0; c < OT; ++c) {
                dsum[nt][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
                dgrp[nt][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
                qsh[nt][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }

        // Exchange records (split): region 0 = group sum of the full sets, region 1 = this half of
        // the shared set's raw output sum.  A value v of lane l travels as 16-byte self-validating
        // granules {tag, v.x, tag, v.y} {tag, v.z, tag, v.w}: write-through (sc1) stores, sc1 loads,
        // no flag and no fence - a granule is accepted when its tags match (G16 recipe R2).
        const unsigned int xtag = xtag0 + (unsigned int)(t + 1);
        constexpr int XREG = NT * OT * 2 * 64 * 16;             // bytes per region
        auto xbase = [&](int g, int region) { return (((lpair * 2 + g) * 2 + (t & 1)) * 2 + region) * XREG + lane * 16; };
        auto xput = [&](int region, const f32x4 (&v)[NT][OT]) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int c = 0; c < OT; ++c)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        u32x4 g;
                        g.x = xtag; g.y = __float_as_uint(v[nt][c][2 * hh]);
                        g.z = xtag; g.w = __float_as_uint(v[nt][c][2 * hh + 1]);
                        __builtin_amdgcn_raw_buffer_store_b128(g, xrs, xbase(grp, region) + ((nt * OT + c) * 2 + hh) * 1024, 0, L2A_SC1);
                    }
        };
        auto xget = [&](int region, f32x4 (&v)[NT][OT]) {
            bool ok = true;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int c = 0; c < OT; ++c)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const u32x4 g = __builtin_amdgcn_raw_buffer_load_b128(
                            xrs, xbase(grp ^ 1, region) + ((nt * OT + c) * 2 + hh) * 1024, 0, L2A_SC1);
                        v[nt][c][2 * hh] = __uint_as_float(g.y);
                        v[nt][c][2 * hh + 1] = __uint_as_float(g.w);
                        ok = ok && (g.x == xtag) && (g.z == xtag);
                    }
            return ok;
        };

        // Early exchange (round 6; tile split with a shared set whose last batch holds two or more sets - config 2: all three):
        // the records leave as early as the canonical order allows and the partner's are requested BEFORE this workgroup is
        // through its own epilogue.  Wave 0 reduces the shared set FIRST and publishes this half of it at once; wave 1 (natural
        // order) publishes the group sum of the full sets right after the last of them; wave 3 requests the partner's two
        // records straight after its own last reduce and looks at them only after the next step's actions are taken over - the
        // ~3.4k clocks a sweep takes from issue to return (whatever it finds) now run under the other waves' reduces and the
        // hand-over instead of behind them.  The shared set adds nothing to `dgrp` in phase C (its term is formed in the
        // combine), so its place in wave 0's order changes no bit.  Config 2: 1.4153 -> 1.3997 ms (profiles/r06_ab_xsweep.jsonl;
        // the records leaving early WITHOUT the early request gained nothing: r06_ab_xord.jsonl).
        bool xearly = false;        // this step's records left in phase C, wave 3 holds the requested partner records in `sw`
        u32x4 sw[2][NT][OT][2];
        for (int b0 = 0; b0 < n_seq; b0 += LB) {
            const int nb = (n_seq - b0 < LB) ? n_seq - b0 : LB;
            f32x4 hreg[NT][TPW];    // activations of the LAST hidden layer (stay in registers)

            // ======== phase A: layer 0 of every set of the batch ==================================
            // B = normalised [obs | act] built from the state fragment; result -> LDS region j (hcur when LB == 1)
            // (a generic lambda instantiated twice - "another layer 0 follows" / "the batch's first GEMM follows" - and the
            // last iteration peeled: on every path into the GEMM its first operands have just been requested, so no
            // register of one phase's prefetch is carried through the other's code)
            auto layer0 = [&](int j, auto last_tag) {
                constexpr bool last_l0 = decltype(last_tag)::value;
                const int i = b0 + j;
                const int e = seq(i);
                const float* wb = set_base(e);
                const float* nr = nrm + i * NRM_SET;
                const __amdgpu_buffer_rsrc_t rs_out = l2a_rsrc(wb + p.pk_wout, wo_bytes);
                L2A_TS(0)
                // k-group 0 arrived with the previous phase; the others land under its MFMAs
                const __amdgpu_buffer_rsrc_t r0 = l2a_rsrc(wb + p.pk_w0, w0_bytes);
                f32x4 a[KG0][TPW];
#pragma unroll
                for (int g = 0; g < KG0; ++g)
#pragma unroll
                    for (int tt = 0; tt < TPW; ++tt)
                        a[g][tt] = (g < KG0P) ? pfL0[g < KG0P ? g : 0][tt] : l2a_ldw(r0, voff0[tt] + g * 1024, 0);
                L2A_TS(8)
                // normalised inputs of every k-group up front: all 2 KG0 constant reads are issued together (the
                // register-starved scheduler otherwise puts each LDS round trip right in front of its MFMAs)
                constexpr bool XPRE = (KG0 <= 2);       // wider inputs: no registers to spare, per k-group as before
                auto norm_in = [&](int g, const f32x4& mu, const f32x4& iv, f32x4 (&x)[NT]) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        f32x4 sv = (f32x4){0.f, 0.f, 0.f, 0.f};
                        if (g < OT) sv = st[nt][g < OT ? g : 0];
                        f32x4 aa = (f32x4){0.f, 0.f, 0.f, 0.f};
                        if (g == ga0) aa = av[nt][0];
                        if (g == ga0 + 1) aa = av[nt][1];
                        // a slot holds a state feature OR an action OR padding: the state fragment is exactly zero
                        // outside the observation (zero-padded weights, biases and denormalisation), the action fragment
                        // exactly zero outside the action (bounds-checked loads) - their sum is the select, without a lane
                        // mask per element
#pragma unroll
                        for (int ii = 0; ii < 4; ++ii) x[nt][ii] = ((sv[ii] + aa[ii]) - mu[ii]) * iv[ii];
                    }
                };
                f32x4 xin[XPRE ? KG0 : 1][NT];
                if (XPRE) {
                    f32x4 mu[KG0], iv[KG0];
#pragma unroll
                    for (int g = 0; g < KG0; ++g) {
                        mu[g] = *reinterpret_cast<const f32x4*>(nr + 16 * g + 4 * qq);
                        iv[g] = *reinterpret_cast<const f32x4*>(nr + 16 * KG0 + 16 * g + 4 * qq);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int g = 0; g < KG0; ++g) norm_in(g, mu[g], iv[g], xin[XPRE ? g : 0]);
                }
                // the epilogue's bias vectors: requested now, in flight under the MFMAs (read in the epilogue they cost one
                // exposed LDS round trip per tile)
                f32x4 bias0[TPW];
#pragma unroll
                for (int tt = 0; tt < TPW; ++tt)
                    bias0[tt] = *reinterpret_cast<const f32x4*>(nr + CST_BHID + 16 * (c0 + tt) + 4 * qq);
                f32x4 acc[NT][TPW];
                // layer-0 MFMAs; called from every branch below so that each branch's operand
                // prefetch shares a basic block with them and can be interleaved (NPF loads)
                auto l0_mfma = [&](auto npf_tag) {
                    constexpr int NPF = decltype(npf_tag)::value;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int tt = 0; tt < TPW; ++tt) acc[nt][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int g = 0; g < KG0; ++g) {
                        f32x4 xg[NT];
                        if (XPRE) {
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) xg[nt] = xin[XPRE ? g : 0][nt];
                        } else {
                            const f32x4 mu = *reinterpret_cast<const f32x4*>(nr + 16 * g + 4 * qq);
                            const f32x4 iv = *reinterpret_cast<const f32x4*>(nr + 16 * KG0 + 16 * g + 4 * qq);
                            norm_in(g, mu, iv, xg);
                        }
#pragma unroll
                        for (int ii = 0; ii < 4; ++ii) {
                            // K0L: MFMA ii of the LAST k-group multiplies input features 16 (KG0 - 1) + 4 qq + ii; when the
                            // input layer ends within the group's first K0L features (Ant: 49 = 3 * 16 + 1) the other
                            // MFMAs see zero weights and zero inputs - skipping them leaves every bit as it was
                            if (g == KG0 - 1 && ii >= K0L) continue;
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                                for (int tt = 0; tt < TPW; ++tt)
                                    acc[nt][tt] = L2A_MFMA(a[g][tt][ii], xg[nt][ii], acc[nt][tt]);
                        }
                    }
                    // hint: the a[g >= 1] loads first, then one prefetch load per few MFMAs
                    constexpr int NM = ((KG0 - 1) * 4 + K0L) * NT * TPW;
                    constexpr int PER = NM / (NPF > 0 ? NPF : 1) > 0 ? NM / (NPF > 0 ? NPF : 1) : 1;
                    __builtin_amdgcn_sched_group_barrier(0x020, (KG0 - KG0P) * TPW, 0);
#pragma unroll
                    for (int k = 0; k < NPF; ++k) {
                        __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                };
                // operands of the phase after this one
                if (N1) {
#pragma unroll
                    for (int tt = 0; tt < L2A_PFT(TPW, OT); ++tt)
#pragma unroll
                        for (int c = 0; c < OT; ++c)
                            pfO[tt][c] = l2a_ldw(rs_out, olane[c] + (c0 + tt) * 1024, c * HT * 1024);
                    l0_mfma(std::integral_constant<int, L2A_PFT(TPW, OT) * OT>{});
                } else if (!last_l0) {
                    // the next set's layer 0 follows: its k-group 0
                    const __amdgpu_buffer_rsrc_t r0n = l2a_rsrc(set_base(seq(i + 1)) + p.pk_w0, w0_bytes);
#pragma unroll
                    for (int g = 0; g < KG0P; ++g)
#pragma unroll
                        for (int tt = 0; tt < TPW; ++tt) pfL0[g][tt] = l2a_ldw(r0n, voff0[tt] + g * 1024, 0);
                    l0_mfma(std::integral_constant<int, KG0P * TPW>{});
                } else {
                    // the first hidden GEMM of the batch follows (set seq(b0))
                    const int e0 = seq(b0);
                    const __amdgpu_buffer_rsrc_t rs1 = l2a_rsrc(set_base(e0) + p.pk_wmid, wm_bytes);
                    const bool next_half = !NOHALF && (e0 == e_shared) && n_hidden == 2;
                    if (next_half) {
#pragma unroll
                        for (int tt = 0; tt < TH; ++tt) {
                            pfA[tt] = l2a_ldw(rs1, voffh[tt], 0);
                            pfB[tt] = l2a_ldw(rs1, voffh[tt] + 1024, 0);
                        }
                        l0_mfma(std::integral_constant<int, 2 * TH>{});
                    } else {
#pragma unroll
                        for (int tt = 0; tt < TPW; ++tt) {
                            pfA[tt] = l2a_ldw(rs1, voff[tt], 0);
                            pfB[tt] = l2a_ldw(rs1, voff[tt] + 1024, 0);
                        }
                        l0_mfma(std::integral_constant<int, 2 * TPW>{});
                    }
                }
                L2A_TS(9)
#pragma unroll
                for (int tt = 0; tt < TPW; ++tt) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        hreg[nt][tt] = l2a_actv<GACT>(acc[nt][tt] + bias0[tt], p.hidden_act, p.hid_floor);
                }
                L2A_TS(10)
                if (!N1) {
                    f32x4* hdst = (LB > 1) ? buf0 + j * p.sa_elems : hcur;
#pragma unroll
                    for (int tt = 0; tt < TPW; ++tt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) hdst[(nt * HT + c0 + tt) * 64 + lane] = hreg[nt][tt];
                }
                L2A_TS(1)
            };

            // ======== phase B: hidden -> hidden layers and the fused output layer, set by set ======
            // the last hidden layer keeps its output in registers and feeds the output layer directly
            // (again a generic lambda, the batch's last set peeled: a set that is not the last of its batch is a full
            // member and hands over to the next set's GEMM, the last one hands over to a layer 0)
            auto member = [&](int j, auto last_tag) {
                constexpr bool last_set = decltype(last_tag)::value;
                const int i = b0 + j;
                const int e = seq(i);
                const bool is_half = !NOHALF && last_set && (e == e_shared);
                const float* wb = set_base(e);
                const float* nr = nrm + i * NRM_SET;
                const __amdgpu_buffer_rsrc_t rs_out = l2a_rsrc(wb + p.pk_wout, wo_bytes);
                f32x4* pb = (LB > 1) ? pbase + j * PS : hoth;      // LB == 1: re-read below, after the inner layers' swaps
                L2A_TS(2)
                if (!N1) {
                    f32x4 acc[NT][TPW], bias[TPW];
                    for (int l = 1; last_set && l < n_hidden - 1; ++l) {        // inner layers: LB == 1 only (the set is its batch)
                        const float* wl = wb + p.pk_wmid + (long long)(l - 1) * p.pk_wmid_stride;
                        const bool next_half = is_half && (l == n_hidden - 2);
                        l2a_hidden_gemm<NT, TPW, TPW, OT, false>(l2a_rsrc(wl, wm_bytes), l2a_rsrc(wl + p.pk_wmid_stride, wm_bytes),
                                                                 voff, next_half ? voffh : voff, next_half ? TH : TPW, 0,
                                                                 hcur, pfA, pfB, pfO, acc, lane,
                                                                 nr + CST_BHID + l * (16 * HT) + 16 * c0 + 4 * qq, bias, olane);
#pragma unroll
                        for (int tt = 0; tt < TPW; ++tt) {
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                hoth[(nt * HT + c0 + tt) * 64 + lane] = l2a_actv<GACT>(acc[nt][tt] + bias[tt], p.hidden_act, p.hid_floor);
                        }
                        __syncthreads();
                        f32x4* tmp = hcur; hcur = hoth; hoth = tmp;
                    }
                    if (LB == 1) pb = hoth;
                    L2A_TS(15)
                    const f32x4* hin = (LB > 1) ? buf0 + j * p.sa_elems : hcur;
                    const __amdgpu_buffer_rsrc_t rs_last =
                        l2a_rsrc(wb + p.pk_wmid + (long long)(n_hidden - 2) * p.pk_wmid_stride, wm_bytes);
                    const float* bl = nr + CST_BHID + (n_hidden - 1) * (16 * HT);
                    if (!is_half) {
                        l2a_hidden_gemm<NT, TPW, TPW, OT, true>(rs_last, rs_out, voff, voff, TPW, c0, hin, pfA, pfB, pfO,
                                                                acc, lane, bl + 16 * c0 + 4 * qq, bias, olane);
#pragma unroll
                        for (int tt = 0; tt < TPW; ++tt) {
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                hreg[nt][tt] = l2a_actv<GACT>(acc[nt][tt] + bias[tt], p.hidden_act, p.hid_floor);
                        }
                    } else {
                        l2a_hidden_gemm<NT, TH, TPW, OT, true>(rs_last, rs_out, voffh, voffh, TH, tile0h, hin, pfA, pfB,
                                                               pfO, acc, lane, bl + 16 * tile0h + 4 * qq, bias, olane);
#pragma unroll
                        for (int tt = 0; tt < TH; ++tt) {
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                hreg[nt][tt] = l2a_actv<GACT>(acc[nt][tt] + bias[tt], p.hidden_act, p.hid_floor);
                        }
                    }
                }
                L2A_TS(3)
                // output layer; its MFMAs cover the first operand loads of the phase that follows: the next set's hidden
                // GEMM inside a batch, else layer 0 of the next batch / step
                if (last_set) {
                    const __amdgpu_buffer_rsrc_t r0n = l2a_rsrc(set_base(seq((i + 1 < n_seq) ? i + 1 : 0)) + p.pk_w0, w0_bytes);
                    auto pf_l0 = [&]() {
#pragma unroll
                        for (int g = 0; g < KG0P; ++g)
#pragma unroll
                            for (int tt = 0; tt < TPW; ++tt) pfL0[g][tt] = l2a_ldw(r0n, voff0[tt] + g * 1024, 0);
                    };
                    if (is_half) l2a_out_phase<NT, TH, TPW, OT, KG0P * TPW, O4>(hreg, pfO, rs_out, tile0h, pf_l0, pb, wave, lane, olane);
                    else l2a_out_phase<NT, TPW, TPW, OT, KG0P * TPW, O4>(hreg, pfO, rs_out, c0, pf_l0, pb, wave, lane, olane);
                } else {
                    const int en = seq(i + 1);
                    const __amdgpu_buffer_rsrc_t rsn = l2a_rsrc(set_base(en) + p.pk_wmid, wm_bytes);   // nb > 1: two hidden layers
                    if (!NOHALF && en == e_shared) {
                        l2a_out_phase<NT, TPW, TPW, OT, 2 * TH, O4>(hreg, pfO, rs_out, c0, [&]() {
#pragma unroll
                            for (int tt = 0; tt < TH; ++tt) {
                                pfA[tt] = l2a_ldw(rsn, voffh[tt], 0);
                                pfB[tt] = l2a_ldw(rsn, voffh[tt] + 1024, 0);
                            }
                        }, pb, wave, lane, olane);
                    } else {
                        l2a_out_phase<NT, TPW, TPW, OT, 2 * TPW, O4>(hreg, pfO, rs_out, c0, [&]() {
#pragma unroll
                            for (int tt = 0; tt < TPW; ++tt) {
                                pfA[tt] = l2a_ldw(rsn, voff[tt], 0);
                                pfB[tt] = l2a_ldw(rsn, voff[tt] + 1024, 0);
                            }
                        }, pb, wave, lane, olane);
                    }
                }
                L2A_TS(4)
            };
            // (layer 0 of a batch's later sets carried inside the previous set's GEMM - one hidden tile per loop iteration - was
            // measured and lost: it lengthens the host GEMM by what the phase of its own costs, profiles/r03_ab_kernel_variants.jsonl)
            for (int j = 0; j + 1 < nb; ++j) layer0(j, std::false_type{});
            layer0(nb - 1, std::true_type{});
            if (!N1) __syncthreads();
            for (int j = 0; j + 1 < nb; ++j) member(j, std::false_type{});
            member(nb - 1, std::true_type{});

            // ======== phase C: canonical reduce of every set of the batch ===========================
            __syncthreads();
            // (the batch that ends with the shared set, if it holds another set too)
            // (instances of up to 32 input features: the requested records are 16 VGPRs per obs tile, and the Ant's instance -
            //  496 VGPRs without them - would spill)
            constexpr bool XE_OK = L2A_XEARLY && !NOHALF && NT == 1 && OT <= 2 && KG0 <= 2;
            const bool xe = XE_OK && split == 2 && nb >= 2 && b0 + nb == n_seq;
            for (int jj = 0; jj < nb; ++jj) {
                const int j = (xe && wave == 0) ? (jj == 0 ? nb - 1 : jj - 1) : jj;     // wave 0: the shared set first
                const int i = b0 + j;
                const int e = seq(i);
                const bool is_half = !NOHALF && (e == e_shared);
                const float* nr = nrm + i * NRM_SET;
                const f32x4* pb = (LB > 1) ? pbase + j * PS : hoth;
                L2A_TS(5)
                if (e == e_half && !split) {      // group A complete: park it, start group B
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int c = 0; c < OT; ++c) {
                            dsum[nt][c] = dgrp[nt][c];
                            dgrp[nt][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
                        }
                }
                // Sum the waves' partials in the canonical order - a balanced tree over the eight chunks of hidden units,
                // ((c0+c1)+(c2+c3)) + ((c4+c5)+(c6+c7)): a full member's wave w wrote c_2w + c_2w+1, so (p0+p1)+(p2+p3) is the
                // whole tree; a half member's wave w wrote chunk w of ITS half, so the same expression is its half of the
                // tree, and the other half arrives at the end of the step from the partner workgroup.  (Until round 3 the
                // order was ((c0+c1)+c2)+c3 + ((c4+c5)+c6)+c7, which forced a full member to keep its two chunks apart: eight
                // partials per set to write and to read back instead of four.)
                // All LDS reads of an obs tile are issued before the first add: left to itself the compiler,
                // which is out of registers kernel-wide, serialises them read-wait-add (~120 cycles each, 22 of
                // them: the reduce measured 1.4k cycles per set in tools/timeline.py).
                // (reading all obs tiles of a set up front - one exposed round trip instead of OT - was measured and lost:
                // 1.434 against 1.431 ms on config 2, profiles/r03_ab_kernel_variants.jsonl)
#pragma unroll
                for (int c = 0; c < OT; ++c) {
                    f32x4 part[L2A_NW][NT];
#pragma unroll
                    for (int ch = 0; ch < L2A_NW; ++ch)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) part[ch][nt] = pb[((ch * NT + nt) * OT + c) * 64 + lane];
                    const f32x4 bias = *reinterpret_cast<const f32x4*>(nr + CST_BOUT + 16 * c + 4 * qq);
                    const f32x4 omu = *reinterpret_cast<const f32x4*>(nr + 32 * KG0 + 16 * c + 4 * qq);
                    const f32x4 osd = *reinterpret_cast<const f32x4*>(nr + 32 * KG0 + 16 * OT + 16 * c + 4 * qq);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        static_assert(L2A_NW == 4, "the canonical tree below is written out for four waves");
                        f32x4 s = (part[0][nt] + part[1][nt]) + (part[2][nt] + part[3][nt]);
                        if (is_half) {
                            qsh[nt][c] = s;
                        } else {
                            if (O4 && c == OT - 1) {        // the four quarters of the hidden units (l2a_out_phase)
#pragma unroll
                                for (int ii = 0; ii < 4; ++ii) s[ii] = l2a_sum_xor32(l2a_sum_xor16(s[ii]));
                            }
                            s = l2a_actv<GACT>(s + bias, p.output_act, p.out_floor);
                            if (FAN) dgrp[nt][c] = s * osd + omu;      // this member's term as it stands: summed with the others' below
                            else dgrp[nt][c] += s * osd + omu;
                        }
                    }
                }
                // the group sum of the full sets leaves as soon as it exists (LB == 1: it travels under the half set)
                if (!NOHALF && split == 2 && i == n_full - 1 && wave == (XE_OK ? 1 : 0)) xput(0, dgrp);
                if (xe && wave == 0 && jj == 0) xput(1, qsh);
                if (xe && wave == 3 && jj == nb - 1) {
                    // single model (E == 1): there are no full sets, region 0 is never written - not requested
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                            for (int c = 0; c < OT; ++c)
#pragma unroll
                                for (int hh = 0; hh < 2; ++hh)
                                    sw[r][nt][c][hh] = (r == 1 || e_loop > 1)
                                        ? __builtin_amdgcn_raw_buffer_load_b128(xrs, xbase(grp ^ 1, r) + ((nt * OT + c) * 2 + hh) * 1024, 0, L2A_SC1)
                                        : (u32x4){xtag, 0u, xtag, 0u};
                }
                L2A_TS(6)
            }
            if (xe) xearly = true;
            // LDS hazards.  LB == 1: the partial sums live in `hoth`; `hcur` was last read by the final
            // hidden layer, before the barrier above.  n_hidden >= 2: the next writes are layer 0
            // -> `hcur` (free) and, only after the layer-0 barrier, `hoth` again (every wave has
            // finished these reads by then).  n_hidden == 1: there is no layer-0 barrier, the next
            // partial sums would land in `hoth` while slow waves still read it -> alternate regions.
            // LB > 1: regions and partial areas are rewritten only after the next batch's / step's
            // barriers, which every wave reaches after these reads.
            if (N1) { f32x4* tmp = hcur; hcur = hoth; hoth = tmp; }
        }

        // ---- combine the two workgroups of a tile ---------------------------------------------
        { const int e = 7; L2A_TS(9) }
        float asq_t[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) asq_t[nt] = asq[nt];
        if (FAN) {
            // ---- member fan: the E workgroups of a tile each hold ONE member's term; everybody publishes its own, collects
            // the E - 1 others (wave w polls partners w, w + 4, ..: with five members one partner per wave, all sweeps in
            // flight together) and adds all E in the canonical order - the unsplit launch's (0 + t_0 + .. ) + (0 + t_eh + ..).
            // One region per (tile, member, step parity): a member can be one step ahead of a partner at most (it needs the
            // partner's record of step t + 1 to get to step t + 2, and the partner publishes that after reading step t).
            constexpr int XREGF = NT * OT * 2 * 64 * 16;
            auto fbase = [&](int g) { return ((lpair * e_loop + g) * 2 + (t & 1)) * XREGF + lane * 16; };
            // (dealing the record's rows over the four waves instead of wave 0 measured +0.1 .. +0.6 %: round 6 A/B)
            if (wave == 0) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int c = 0; c < OT; ++c) {
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh) {
                            u32x4 g;
                            g.x = xtag; g.y = __float_as_uint(dgrp[nt][c][2 * hh]);
                            g.z = xtag; g.w = __float_as_uint(dgrp[nt][c][2 * hh + 1]);
                            __builtin_amdgcn_raw_buffer_store_b128(g, xrs, fbase(grp) + ((nt * OT + c) * 2 + hh) * 1024, 0, L2A_SC1);
                        }
                        xlds[((grp * NT + nt) * OT + c) * 64 + lane] = dgrp[nt][c];
                    }
            }
            { const int e = 7; L2A_TS(11) }
            take_actions(t + 2);
            for (int k = wave; k < e_loop - 1; k += L2A_NW) {
                int g2 = grp + 1 + k;
                if (g2 >= e_loop) g2 -= e_loop;
                f32x4 oth[NT][OT];
                while (true) {
                    bool ok = true;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int c = 0; c < OT; ++c)
#pragma unroll
                            for (int hh = 0; hh < 2; ++hh) {
                                const u32x4 g = __builtin_amdgcn_raw_buffer_load_b128(
                                    xrs, fbase(g2) + ((nt * OT + c) * 2 + hh) * 1024, 0, L2A_SC1);
                                oth[nt][c][2 * hh] = __uint_as_float(g.y);
                                oth[nt][c][2 * hh + 1] = __uint_as_float(g.w);
                                ok = ok && (g.x == xtag) && (g.z == xtag);
                            }
                    if (__all(ok)) break;
                    if (spin_left == 0) {               // partner never arrived: flag the launch, do not hang
                        if (lane == 0) __hip_atomic_fetch_or(p.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                    --spin_left;
                    __builtin_amdgcn_s_sleep(4);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int c = 0; c < OT; ++c) xlds[((g2 * NT + nt) * OT + c) * 64 + lane] = oth[nt][c];
            }
            { const int e = 7; L2A_TS(12) }
            __syncthreads();
            { const int e = 7; L2A_TS(13) }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int c = 0; c < OT; ++c) {
                    dsum[nt][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    dgrp[nt][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            {   // every term's LDS read first (slots past the ensemble re-read the last one and are not added), then the
                // adds in member order: one exposed LDS round trip instead of E (timeline r06: 3.4k -> 1.9k clocks)
                constexpr int EMAX = 8;
                f32x4 v[EMAX][NT][OT];
#pragma unroll
                for (int e = 0; e < EMAX; ++e) {
                    const int es = e < e_loop ? e : e_loop - 1;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int c = 0; c < OT; ++c) v[e][nt][c] = xlds[((es * NT + nt) * OT + c) * 64 + lane];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < EMAX; ++e)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int c = 0; c < OT; ++c) {
                            if (e < e_half) dsum[nt][c] += v[e][nt][c];
                            else if (e < e_loop) dgrp[nt][c] += v[e][nt][c];
                        }
            }
            // (the staging slots are rewritten after the next step's barriers, which every wave reaches after these reads)
        } else if (!WHOLE && split) {
            if (wave == 0 && !xearly) {
                if (split == 2) xput(1, qsh);
                else xput(0, dgrp);
            }
            { const int e = 7; L2A_TS(11) }
            // while the records travel (a write-through store + an sc1 load across two XCDs' L2s: ~4k clocks),
            // take over the next step's actions and request the ones after them
            take_actions(t + 2);
            if (xearly && wave == 3) {
                // the records requested in phase C: usually valid; a miss costs a further round trip, as before
                while (true) {
                    bool ok = true;
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                            for (int c = 0; c < OT; ++c)
#pragma unroll
                                for (int hh = 0; hh < 2; ++hh)
                                    ok = ok && (sw[r][nt][c][hh].x == xtag) && (sw[r][nt][c][hh].z == xtag);
                    if (__all(ok)) break;
                    if (spin_left == 0) {               // partner never arrived: flag the launch, do not hang
                        if (lane == 0) __hip_atomic_fetch_or(p.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                    --spin_left;
                    __builtin_amdgcn_s_sleep(4);
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                            for (int c = 0; c < OT; ++c)
#pragma unroll
                                for (int hh = 0; hh < 2; ++hh)
                                    if (r == 1 || e_loop > 1)
                                        sw[r][nt][c][hh] = __builtin_amdgcn_raw_buffer_load_b128(
                                            xrs, xbase(grp ^ 1, r) + ((nt * OT + c) * 2 + hh) * 1024, 0, L2A_SC1);
                }
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int c = 0; c < OT; ++c) {
                            f32x4 v;
                            v[0] = __uint_as_float(sw[r][nt][c][0].y); v[1] = __uint_as_float(sw[r][nt][c][0].w);
                            v[2] = __uint_as_float(sw[r][nt][c][1].y); v[3] = __uint_as_float(sw[r][nt][c][1].w);
                            xlds[((r * NT + nt) * OT + c) * 64 + lane] = v;
                        }
                { const int e = 7; L2A_TS(12) }
            }
            if (wave == 0 && !xearly) {
                f32x4 oth[2][NT][OT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int c = 0; c < OT; ++c) {
                        oth[0][nt][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
                        oth[1][nt][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    }
                // single model (E == 1): there are no full sets, region 0 is never written - skip it
                const bool want0 = (split == 1) || (e_loop > 1);
                unsigned int spins = 0;
                // Both regions in ONE round trip (an sc1 load sweep costs ~2-3k clocks whatever its size): the partner's records
                // become visible ~2.2k clocks after it published them; a sweep that misses them costs a whole further round trip
                // (timeline r03: 3.3k clocks from publish to swept when the first sweep hits, 4.9k when it is the second).
                auto give_up = [&]() {                  // partner never arrived: flag it, do not hang; the budget is per
                    ++spins;                            // launch, so the remaining steps give up after one poll each
                    if (spin_left == 0) {
                        if (lane == 0) __hip_atomic_fetch_or(p.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        return true;
                    }
                    --spin_left;
                    return false;
                };
                // Measured and lost (profiles/r03_ab_kernel_variants.jsonl; the code is in the history): two sweeps in flight half a
                // round trip apart (41 more registers, and loads retire in order: the sweep still in flight stalls the next step's
                // first operand wait); a learnt delay of the first sweep (every jitter-induced miss of the late workgroup then
                // costs 16 steps); streaming / system-scope granules; both workgroups of a pair on one XCD.
                while (true) {
                    bool ok = true;
                    if (split == 2) ok = xget(1, oth[1]);
                    if (want0) ok = xget(0, oth[0]) && ok;
                    if (__all(ok) || give_up()) break;
                    __builtin_amdgcn_s_sleep(4);
                }
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int c = 0; c < OT; ++c) xlds[((k * NT + nt) * OT + c) * 64 + lane] = oth[k][nt][c];
                { const int e = 7; L2A_TS(12) }
#ifdef L2A_TIMELINE
                if (p.dbg && pairid == 0 && lane == 0) p.dbg[(((long long)(grp * p.h + t) * 8 + 7) * 8 + wave) * 16 + 14] = spins;
#endif
            }
            { const int e = 7; L2A_TS(10) }
            __syncthreads();
            { const int e = 7; L2A_TS(13) }
            const float* nrs = nrm + (n_seq - 1) * NRM_SET;     // the shared set is the last of the sequence
            {   // every LDS read of the combine first, then the arithmetic (one exposed round trip, not four)
                f32x4 cb[OT], cm[OT], cs[OT], og[NT][OT], oq[NT][OT];
#pragma unroll
                for (int c = 0; c < OT; ++c) {
                    cb[c] = *reinterpret_cast<const f32x4*>(nrs + CST_BOUT + 16 * c + 4 * qq);
                    cm[c] = *reinterpret_cast<const f32x4*>(nrs + 32 * KG0 + 16 * c + 4 * qq);
                    cs[c] = *reinterpret_cast<const f32x4*>(nrs + 32 * KG0 + 16 * OT + 16 * c + 4 * qq);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        og[nt][c] = xlds[((0 * NT + nt) * OT + c) * 64 + lane];       // partner's group sum
                        oq[nt][c] = xlds[((1 * NT + nt) * OT + c) * 64 + lane];       // partner's half of the shared set
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < OT; ++c)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        f32x4 ga = (grp == 0) ? dgrp[nt][c] : og[nt][c];               // group A (without the shared set)
                        const f32x4 gb = (grp == 0) ? og[nt][c] : dgrp[nt][c];         // group B
                        if (split == 2) {
                            f32x4 s = qsh[nt][c] + oq[nt][c];                          // S1 + S2
                            if (O4 && c == OT - 1) {
#pragma unroll
                                for (int ii = 0; ii < 4; ++ii) s[ii] = l2a_sum_xor32(l2a_sum_xor16(s[ii]));
                            }
                            s = l2a_actv<GACT>(s + cb[c], p.output_act, p.out_floor);
                            ga += s * cs[c] + cm[c];        // the shared set is the last member of group A
                        }
                        dsum[nt][c] = ga;
                        dgrp[nt][c] = gb;
                    }
            }
        } else {
            take_actions(t + 2);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int c = 0; c < OT; ++c) dsum[nt][c] += dgrp[nt][c];     // group A + group B

        // ---- ensemble mean, reward, state update -------------------------------------------
        const float disc_t = (float)disc_pow;
        disc_pow *= p.discount;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float plin = ((qq == 0) ? p.rw.alive : 0.0f) - p.rw.ctrl_coef * asq_t[nt];
            float psq = 0.0f;
#pragma unroll
            for (int c = 0; c < OT; ++c) {
                f32x4 d = dsum[nt][c];
                if (e_loop > 1) {
                    // d / E, correctly rounded like the IEEE division it replaces (Markstein: q = RN(d y), r = d - q E exactly
                    // (fma), q' = RN(q + r y) with y = RN(1 / E)) - three operations instead of the ~11 of v_div_*.  A diverged
                    // candidate keeps its +-inf like the division would (the correction term of an infinite q is inf - inf = NaN,
                    // and NaN ranks highest in the arg-max key: ADVICE r3)
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) {
                        const float q = d[ii] * e_inv;
                        const float qc = fmaf(fmaf(-q, e_count, d[ii]), e_inv, q);
                        d[ii] = (fabsf(q) < INFINITY) ? qc : q;
                    }
                }
                const f32x4 nx = st[nt][c] + d;
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    const int dim = 16 * c + 4 * qq + ii;
                    if (dim == p.rw.vel_index) plin += p.rw.w_vel * d[ii] * p.rw.inv_dt;
                    const bool in_dist = (p.rw.dist_coef != 0.0f) && (dim >= p.rw.dist_index) &&
                                         (dim < p.rw.dist_index + 3) && (dim < obs_dim);
                    psq += in_dist ? nx[ii] * nx[ii] : 0.0f;
                }
                st[nt][c] = nx;
            }
            plin = l2a_sum_xor32(l2a_sum_xor16(plin));      // (r0 + r1) + (r2 + r3), as the xor-16 / xor-32 shuffles gave
            psq = l2a_sum_xor32(l2a_sum_xor16(psq));
            float r = plin;
            if (p.rw.dist_coef != 0.0f) r -= p.rw.dist_coef * sqrtf(psq);
            ret[nt] = fmaf(disc_t, r, ret[nt]);
        }
        { const int e = 7; L2A_TS(7) }     // end of the step (after exchange, reward, state update)
    }

    // ---- results: wave 0 writes returns, arg-max key and (for predict) the final state -----
    if (wave == 0 && grp == 0) {
        unsigned long long key = 0ull;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (valid[nt] && qq == 0) {
                if (p.returns_out) p.returns_out[(long long)env * p.n + cand[nt]] = ret[nt];
                const unsigned long long k = l2a_key_pack(ret[nt], p.cand_offset + cand[nt]);
                key = (k > key) ? k : key;
            }
            if (p.state_out && valid[nt]) {
                float* srow = p.state_out + ((long long)env * p.n + cand[nt]) * obs_dim;
#pragma unroll
                for (int c = 0; c < OT; ++c)
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) {
                        const int dim = 16 * c + 4 * qq + ii;
                        if (dim < obs_dim) srow[dim] = st[nt][c][ii];
                    }
            }
        }
        if (p.best_key) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const unsigned int hi = __shfl_xor((unsigned int)(key >> 32), off);
                const unsigned int lo = __shfl_xor((unsigned int)(key & 0xffffffffu), off);
                const unsigned long long other = ((unsigned long long)hi << 32) | lo;
                key = (other > key) ? other : key;
            }
            if (lane == 0) {
                if (key != 0ull) atomicMax(p.best_key + env, key);
                l2a_publish_result(p, p.done_total > 0 ? p.done_total : n_tiles);
            }
        }
    }
#if defined(L2A_TIMELINE) || defined(L2A_WGREC)
    if (p.dbg && wave == 0 && lane == 0 && grp < 2) {      // per-workgroup record behind the phase stamps: lifetime and placement
        unsigned long long wg_t1_, wg_r1_;
        unsigned int xcc_;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(wg_t1_), "=s"(wg_r1_) : : "memory");
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));
        unsigned long long* r = p.dbg + (long long)2 * p.h * 8 * 8 * 16 + (long long)(grp * n_tiles + pairid) * 6;
        r[0] = wg_t0_; r[1] = wg_t1_; r[2] = xcc_; r[3] = blockIdx.x; r[4] = wg_r0_; r[5] = wg_r1_;   // [4, 5]: 100 MHz real time
    }
#endif
}
