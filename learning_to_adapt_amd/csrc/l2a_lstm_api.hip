// l2a_lstm_api.hip - host side of the recurrent-planner entry points of include/l2a.h
// (l2a_lstm_*): model storage (raw TF-layout weights + MFMA-packed copies + normalisation), weight
// re-packing and the launch logic of the kernels in l2a_lstm.h.  Everything is enqueued on the
// caller's stream.

#define L2A_PACK_KERNELS 1      // this unit launches the weight re-packing kernels of l2a_micro_pack.h
#include "l2a_host.h"
#include "l2a_lstm.h"
#include "l2a_lstm_valu.h"
#include "l2a_lstm_launch.h"
#include "l2a_rnn_valu.h"
#include "l2a_rnn_mfma.h"
#include "l2a_micro_pack.h"
#include "l2a_micro_launch.h"

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct l2a_lstm {
    l2a_ctx* ctx = nullptr;
    int obs_dim = 0, act_dim = 0, in_dim = 0, units = 0;
    int cell_act = L2A_ACT_TANH, output_act = L2A_ACT_IDENTITY;
    bool mfma_ok = false;
    bool micro_ok = false;                        // the micro-tile kernel of l2a_micro.h has an instance (256 / 512 units)
    long long pk_mg = 0, pk_mo = 0;               // its copies of the gate matrix / output layer
    int UTW = 0, KG0 = 0, OT = 0;
    float* wblk = nullptr;
    long long total = 0;
    long long raw_wk = 0, raw_bk = 0, raw_wo = 0, raw_bo = 0, pk_wg = 0, pk_wout = 0, pk_bout = 0, nm_off = 0;
    bool weights_set = false, norm_set = false;
    std::vector<float> norm_stage;
    unsigned long long* xbuf = nullptr;           // unit-tile split exchange granules
    long long xbuf_bytes = 0;
    unsigned int launch_nonce = 0;
    // generic stacks (l2a_rnn_create): any cell type / several layers -> l2a_rnn_valu_k; `units` = sum(lunits)
    bool generic = false;
    int n_layers = 1, cell_type = L2A_CELL_LSTM;
    int lunits[L2A_RNN_MAX_LAYERS] = {0};
    long long lw[L2A_RNN_MAX_LAYERS][2] = {{0}}, lb[L2A_RNN_MAX_LAYERS][2] = {{0}};
    long long lpk[L2A_RNN_MAX_LAYERS][2] = {{0}};  // the kernels again in MFMA fragment order (l2a_rnn_mfma.h); pk_wout alike
    bool gmicro_ok = false;                       // ... and in the micro-tile kernel's (l2a_rnn_micro.h: every layer 256 or every layer 512 units wide)
    long long lmk[L2A_RNN_MAX_LAYERS][2] = {{0}};
    bool gmicro4_ok = false;                      // ... with workgroups of four micro tiles (plans beyond three per CU)
    float* adv_buf = nullptr;                     // l2a_lstm_plan_rs_sync: [64, act_dim] chosen actions + [64, obs_dim] next obs
};

// First action of every env's winning candidate: out[i] = actions[step 0][i * n + (index(best_key[i]) - cand_offset)]
// (key layout: l2a_key_pack; the index is clamped, a launch flagged invalid may have left anything in the key).
__global__ void l2a_gather_best_k(const unsigned long long* best_key, const float* actions, int m, int n, int cand_offset,
                                  int act_dim, float* out) {
    const int i = blockIdx.x;
    if (i >= m) return;
    const unsigned int low = (unsigned int)(best_key[i] & 0x7fffffffull);
    int idx = (int)(0x7fffffffu - low) - cand_offset;
    idx = idx < 0 ? 0 : (idx >= n ? n - 1 : idx);
    for (int d = threadIdx.x; d < act_dim; d += blockDim.x)      // any action width (the generic cells take act_dim > 64)
        out[i * act_dim + d] = actions[((long long)i * n + idx) * act_dim + d];
}

namespace {

bool lstm_mfma_eligible(int obs_dim, int act_dim, int units) {
    if (units != 128 && units != 256 && units != 512) return false;
    if (obs_dim < 1 || obs_dim > 16 * L2A_OTMAX) return false;
    if (act_dim < 1 || act_dim > 16) return false;
    if (obs_dim + act_dim > 16 * L2A_KG0MAX) return false;
    return true;
}

void fill(const l2a_lstm* md, L2ALstmParams& p) {
    std::memset(&p, 0, sizeof(p));
    p.wblk = md->wblk;
    p.raw_wk = md->raw_wk; p.raw_bk = md->raw_bk; p.raw_wo = md->raw_wo; p.raw_bo = md->raw_bo;
    p.pk_wg = md->pk_wg; p.pk_wout = md->pk_wout; p.pk_bout = md->pk_bout; p.nm_off = md->nm_off;
    p.pk_mg = md->pk_mg; p.pk_mo = md->pk_mo;
    p.obs_dim = md->obs_dim; p.act_dim = md->act_dim; p.in_dim = md->in_dim; p.units = md->units;
    p.cell_act = md->cell_act; p.output_act = md->output_act;
    p.KG0 = md->KG0; p.OT = md->OT;
    p.n_layers = md->n_layers; p.cell_type = md->cell_type;
    for (int l = 0; l < L2A_RNN_MAX_LAYERS; ++l) {
        p.layer_units[l] = md->lunits[l];
        for (int q = 0; q < 2; ++q) {
            p.layer_w[l][q] = md->lw[l][q]; p.layer_b[l][q] = md->lb[l][q]; p.layer_pk[l][q] = md->lpk[l][q]; p.layer_mk[l][q] = md->lmk[l][q];
        }
    }
    p.disc0 = 1.0;
}

// generic models (stacks / GRU / BasicRNN / odd LSTM widths): does the launch take the matrix-core kernel of l2a_rnn_mfma.h?
// (unless the caller asked for the VALU one, or its padded LDS rows do not fit: the VALU kernel's dense rows may still)
long long generic_mfma_smem(const l2a_lstm* md) {
    const int gates = md->cell_type == L2A_CELL_LSTM ? 4 : (md->cell_type == L2A_CELL_GRU ? 3 : 1);
    return 4 * l2a_rnn_mfma_lds_floats(md->in_dim, md->obs_dim, md->n_layers, md->lunits, gates);
}
bool generic_uses_mfma(const l2a_lstm* md) {
    return md->generic && md->ctx->kernel_kind != L2A_KERNEL_VALU && generic_mfma_smem(md) <= md->ctx->lds_per_block;
}

int launch(l2a_lstm* md, L2ALstmParams& p, void* stream_v, bool allow_split = true) {
    l2a_ctx* ctx = md->ctx;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    l2a_device_guard guard(ctx->device);
    if (!md->weights_set) return l2a_fail(ctx, L2A_ESTATE, "LSTM weights were never set");
    if (!md->norm_set) return l2a_fail(ctx, L2A_ESTATE, "LSTM normalisation was never set");
    p.dbg = ctx->dbg;
    if (md->generic) {
        // stacks / GRU / BasicRNN: the matrix-core kernel of l2a_rnn_mfma.h unless the caller asked for the VALU one (or
        // its padded LDS rows do not fit: the VALU kernel's dense rows may still)
        p.tiles_per_env = l2a_ceil_div(p.n, L2A_LVT);
        const long long smem_m = generic_mfma_smem(md);
        const int smem_v = (md->in_dim + 3 * md->units + 2 * md->obs_dim + 1) * L2A_LVT * 4;
        if (ctx->kernel_kind != L2A_KERNEL_VALU && md->gmicro_ok) {
            // Micro tiles (l2a_rnn_micro.h; geometry and conditions: the tuned LSTM kernel's branch below).  These models have no
            // unit-tile split, so every plan that leaves CUs idle under 16-candidate tiles - and every plan of at most three micro
            // tiles per CU - takes them (policy 1); policy 0 keeps the 16-candidate kernel, 2 forces micro tiles where eligible.
            const int cus_m = ctx->num_cu > 0 ? ctx->num_cu : 256;
            const int quads = l2a_ceil_div(p.n, 4);
            int W = cus_m / p.m;
            if (W > quads) W = quads;
            int hi = W > 0 ? l2a_ceil_div(quads, W) : 99;
            // Larger plans: workgroups of FOUR micro tiles, ceil(quads / 4) per env, as many rounds as that takes (the instances
            // with LDS rows for sixteen candidates: 256-unit layers, where the stack still fits)
            int mtm = 3;
            if (hi > 3 && md->gmicro4_ok) { hi = 4; mtm = 4; }
            if (hi >= 1 && hi <= 4) {
                W = l2a_ceil_div(quads, hi);
                // `hi` forced down to four is not the natural ceil(quads / W) of the W it yields (quads = 5: W = 2 would leave
                // 5 - 2 * 3 = -1 workgroups of four): take the natural one, so that 1 <= mc_r <= W always holds
                hi = l2a_ceil_div(quads, W);
            }
            const int mc_r = quads - W * (hi - 1);
            const bool eligible = !p.obs_per_row && !p.state_out && !p.c_out && !p.h_out && hi <= mtm && (p.returns_out || p.best_key) &&
                                  W >= 1 && mc_r >= 1 && mc_r <= W;
            if (eligible && ctx->micro_policy != 0) {
                p.mc_w = W;
                p.mc_hi = hi;
                p.mc_r = quads - W * (hi - 1);
                int smem_g = (int)l2a_rnn_micro_smem(md->cell_type, md->n_layers, md->lunits[0], md->KG0, mtm);
                if (smem_g < 84 * 1024) smem_g = 84 * 1024;     // more than half a CU's LDS: one workgroup per CU
                const int rc = l2a_launch_rnn_micro(md->lunits[0], md->cell_type, mtm, &p, (unsigned)(p.m * W), smem_g, stream);
                if (rc != 0) return l2a_fail(ctx, L2A_EHIP, std::string("micro-tile recurrent kernel launch: ") +
                                                            (rc > 0 ? hipGetErrorString((hipError_t)rc) : "no instance"));
                L2A_HIP(ctx, hipGetLastError());
                return L2A_OK;
            }
        }
        bool mfma = ctx->kernel_kind != L2A_KERNEL_VALU;
        if (mfma && smem_m > ctx->lds_per_block) {
            if (ctx->kernel_kind == L2A_KERNEL_MFMA)
                return l2a_fail(ctx, L2A_EINVAL, "LDS budget exceeded by the matrix-core recurrent kernel (" + std::to_string(smem_m) + " B)");
            mfma = false;
        }
        if (!mfma && smem_v > ctx->lds_per_block)
            return l2a_fail(ctx, L2A_EINVAL, "LDS budget exceeded by the generic recurrent kernel (" + std::to_string(smem_v) +
                                             " B): the layers' units may sum to about 800 at most");
        if (mfma) {
            L2A_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(l2a_rnn_mfma_k),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_m));
            hipLaunchKernelGGL(l2a_rnn_mfma_k, dim3((unsigned)(p.m * p.tiles_per_env)), dim3(256), (size_t)smem_m, stream, p);
        } else {
            L2A_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(l2a_rnn_valu_k),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, smem_v));
            hipLaunchKernelGGL(l2a_rnn_valu_k, dim3((unsigned)(p.m * p.tiles_per_env)), dim3(256), smem_v, stream, p);
        }
        L2A_HIP(ctx, hipGetLastError());
        return L2A_OK;
    }
    int kind = ctx->kernel_kind;
    if (kind == L2A_KERNEL_AUTO) kind = md->mfma_ok ? L2A_KERNEL_MFMA : L2A_KERNEL_VALU;
    if (kind == L2A_KERNEL_MFMA && !md->mfma_ok)
        return l2a_fail(ctx, L2A_EINVAL, "LSTM shape is not eligible for the MFMA kernel "
                                         "(needs a single LSTM layer, units in {128, 256, 512}, obs_dim <= 64, act_dim <= 16)");
    if (kind == L2A_KERNEL_MFMA) {
        const int nt = 1;
        const int UT = L2A_NW * md->UTW, U = md->units;
        p.tiles_per_env = l2a_ceil_div(p.n, 16 * nt);
        {
            // Micro tiles (l2a_micro.h): every env's ceil(n / 4) candidate tiles of four dealt to W workgroups of at most
            // three - one workgroup per CU, none idle, no exchange.  Plans only (no per-row states, no state written out:
            // those launches are one step long or chunk continuations and keep the 16-candidate kernel - same bits).
            const int cus_m = ctx->num_cu > 0 ? ctx->num_cu : 256;
            const long long tiles16 = (long long)p.m * p.tiles_per_env;
            const int quads = l2a_ceil_div(p.n, 4);
            int W = cus_m / p.m;
            if (W > quads) W = quads;
            const int hi = W > 0 ? l2a_ceil_div(quads, W) : 99;
            // ... and the fewest workgroups that keep the largest one at `hi` micro tiles (the chip is power limited under this
            // kernel, see launch_rollout in l2a_api.hip)
            if (hi >= 1 && hi <= 3) W = l2a_ceil_div(quads, hi);
            const bool eligible = md->micro_ok && !p.obs_per_row && !p.state_out && !p.c_out && !p.h_out && hi <= 3 &&
                                  (p.returns_out || p.best_key);
            // automatic (profiles/r04_ab_micro.jsonl): the plans the 16-candidate geometries cannot fill - more than CUs / 2
            // tiles (no unit-tile split) and fewer than CUs (the ReBAL default: 0.91 of the 16-candidate launch; 512 units: 0.85)
            // - and plans of at most one micro tile per CU (0.87 - 0.95 of the unit-tile split); in between the unit-tile
            // split wins (1.04 - 1.16).
            const bool unfilled = 2 * tiles16 > cus_m && tiles16 < cus_m;
            const bool wanted = ctx->micro_policy == 2 || (ctx->micro_policy == 1 && (unfilled || hi == 1));
            if (eligible && wanted) {
                p.mc_w = W;
                p.mc_hi = hi;
                p.mc_r = quads - W * (hi - 1);          // workgroups that take `hi` micro tiles
                int smem_m = l2a_lstm_micro_smem(U, md->KG0);
                if (smem_m < 84 * 1024) smem_m = 84 * 1024;     // more than half a CU's LDS: one workgroup per CU
                const int rc = l2a_launch_lstm_micro(U, &p, (unsigned)(p.m * W), smem_m, stream);
                if (rc != 0) return l2a_fail(ctx, L2A_EHIP, std::string("micro-tile LSTM kernel launch: ") +
                                                            (rc > 0 ? hipGetErrorString((hipError_t)rc) : "no instance"));
                L2A_HIP(ctx, hipGetLastError());
                return L2A_OK;
            }
        }
        const int smem = 2 * nt * UT * 64 * 16 + 2 * (L2A_NW * nt * md->OT * 64) * 16 +
                         (32 * md->KG0 + 48 * md->OT + 4 * U) * 4;
        if (smem > ctx->lds_per_block)
            return l2a_fail(ctx, L2A_EINVAL, "LDS budget exceeded (" + std::to_string(smem) + " B)");
        const long long tiles = (long long)p.m * p.tiles_per_env;
        const int cus = ctx->num_cu > 0 ? ctx->num_cu : 256;
        // Unit-tile split: two workgroups per candidate tile when that still fits one workgroup per CU (both must
        // be resident: they swap their halves of h once per step).  Same bits as the unsplit launch.
        p.split = (allow_split && ctx->split_policy != 0 && 2 * tiles <= cus && p.h < 4096) ? 1 : 0;
        if (p.split) {
            const long long need = tiles * 4 * (long long)(UT / 2 + md->OT) * 2048;
            if (need > md->xbuf_bytes) {
                if (md->xbuf) { L2A_HIP(ctx, hipStreamSynchronize(stream)); L2A_HIP(ctx, hipFree(md->xbuf)); md->xbuf = nullptr; }
                L2A_HIP(ctx, l2a_xbuf_alloc(reinterpret_cast<void**>(&md->xbuf), (size_t)need));
                L2A_HIP(ctx, hipMemsetAsync(md->xbuf, 0, (size_t)need, stream));
                md->xbuf_bytes = need;
                md->launch_nonce = 0;
            }
            md->launch_nonce += 1;
            if (md->launch_nonce >= (1u << 20)) {     // tag space exhausted: wipe stale tags, restart
                L2A_HIP(ctx, hipMemsetAsync(md->xbuf, 0, (size_t)md->xbuf_bytes, stream));
                md->launch_nonce = 1;
            }
            p.xtag = md->launch_nonce << 12;
            p.xbuf = md->xbuf;
            p.status = ctx->status_dev;
            p.spin_limit = ctx->spin_limit;
        }
        const unsigned grid = (unsigned)(tiles * (p.split ? 2 : 1));
        const int rc = l2a_launch_lstm(md->UTW, nt, md->OT, md->KG0, &p, grid, smem, stream);
        if (rc == -100) return l2a_fail(ctx, L2A_EINVAL, "no MFMA LSTM kernel instance for this (obs_dim, act_dim, units)");
        if (rc != 0) return l2a_fail(ctx, L2A_EHIP, std::string("MFMA LSTM kernel launch: ") + hipGetErrorString((hipError_t)rc));
    } else {
        p.tiles_per_env = l2a_ceil_div(p.n, L2A_LVT);
        const int smem = (md->in_dim + 3 * md->units + 2 * md->obs_dim + 1) * L2A_LVT * 4;
        if (smem > ctx->lds_per_block)
            return l2a_fail(ctx, L2A_EINVAL, "LDS budget exceeded by the VALU LSTM kernel (" + std::to_string(smem) + " B)");
        L2A_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(l2a_lstm_valu_k),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        hipLaunchKernelGGL(l2a_lstm_valu_k, dim3((unsigned)(p.m * p.tiles_per_env)), dim3(256), smem, stream, p);
    }
    L2A_HIP(ctx, hipGetLastError());
    return L2A_OK;
}

// The controller's own state step on the small-rows kernel (single LSTM layer, units a multiple of 16): see l2a_lstm_advance_k.
bool advance_kernel_ok(const l2a_lstm* md) {
    return !md->generic && (md->units % 16) == 0 && md->units >= 16 && md->ctx->kernel_kind != L2A_KERNEL_VALU &&
           (md->in_dim + md->units) * L2A_ADV_ROWS * 4 + 4 * 64 * L2A_ADV_ROWS * 4 <= 64 * 1024;
}

int launch_advance(l2a_lstm* md, const float* obs, const float* act, const unsigned long long* best_key, const float* actions, int n,
                   int cand_offset, const float* c0, const float* h0, float* c1, float* h1, int m, hipStream_t stream) {
    l2a_ctx* ctx = md->ctx;
    if (!md->weights_set) return l2a_fail(ctx, L2A_ESTATE, "LSTM weights were never set");
    if (!md->norm_set) return l2a_fail(ctx, L2A_ESTATE, "LSTM normalisation was never set");
    L2ALstmAdvParams a;
    std::memset(&a, 0, sizeof(a));
    a.wblk = md->wblk; a.raw_wk = md->raw_wk; a.raw_bk = md->raw_bk; a.nm_off = md->nm_off;
    a.obs_dim = md->obs_dim; a.act_dim = md->act_dim; a.in_dim = md->in_dim; a.units = md->units; a.KG0 = md->KG0;
    a.cell_act = md->cell_act;
    a.obs = obs; a.act = act; a.best_key = best_key; a.actions = actions; a.n = n; a.cand_offset = cand_offset;
    a.c0 = c0; a.h0 = h0; a.c1 = c1; a.h1 = h1; a.m = m;
    const int smem = (md->in_dim + md->units) * L2A_ADV_ROWS * 4 + 4 * 64 * L2A_ADV_ROWS * 4;
    hipLaunchKernelGGL(l2a_lstm_advance_k, dim3((unsigned)(md->units / 16)), dim3(256), smem, stream, a);
    L2A_HIP(ctx, hipGetLastError());
    return L2A_OK;
}

}  // namespace

extern "C" {

int l2a_lstm_mfma_eligible(int obs_dim, int act_dim, int units) {
    return lstm_mfma_eligible(obs_dim, act_dim, units) ? 1 : 0;
}

int l2a_lstm_create(l2a_ctx* ctx, int obs_dim, int act_dim, int units, int cell_act, int output_act,
                    l2a_lstm** out) {
    if (!ctx) return L2A_EINVAL;
    if (!out) return l2a_fail(ctx, L2A_EINVAL, "l2a_lstm_create: out is null");
    *out = nullptr;
    if (obs_dim < 1 || act_dim < 1) return l2a_fail(ctx, L2A_EINVAL, "obs_dim and act_dim must be >= 1");
    if (units < 1 || units > 1024) return l2a_fail(ctx, L2A_EINVAL, "units must be in [1, 1024]");
    if (cell_act < 0 || cell_act > L2A_ACT_SWISH || output_act < 0 || output_act > L2A_ACT_SWISH)
        return l2a_fail(ctx, L2A_EINVAL, "unsupported nonlinearity");
    l2a_lstm* md = new l2a_lstm();
    md->ctx = ctx;
    md->obs_dim = obs_dim; md->act_dim = act_dim; md->in_dim = obs_dim + act_dim; md->units = units;
    md->cell_act = cell_act; md->output_act = output_act;
    md->KG0 = l2a_ceil_div(md->in_dim, 16);
    md->OT = l2a_ceil_div(obs_dim, 16);
    md->mfma_ok = lstm_mfma_eligible(obs_dim, act_dim, units);
    md->UTW = md->mfma_ok ? units / (16 * L2A_NW) : 0;
    long long off = 0;
    auto take = [&off](long long n) { long long o = off; off += (n + 15) / 16 * 16; return o; };
    md->raw_wk = take((long long)(md->in_dim + units) * 4 * units);
    md->raw_bk = take(4LL * units);
    md->raw_wo = take((long long)units * obs_dim);
    md->raw_bo = take(obs_dim);
    if (md->mfma_ok) {
        const int UT = units / 16;
        md->pk_wg = take(4LL * UT * (md->KG0 + UT) * 256);
        md->pk_wout = take((long long)md->OT * UT * 256);
        md->micro_ok = (units == 256 || units == 512);
        if (md->micro_ok) {
            md->pk_mg = take(l2a_lstm_micro_gate_floats(units, md->KG0));
            md->pk_mo = take((long long)(units / 4) * 256);
        }
    }
    md->pk_bout = take(16 * md->OT);
    md->nm_off = take(32 * md->KG0 + 32 * md->OT);
    md->total = off;
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&md->wblk), (size_t)off * sizeof(float));
    if (e == hipSuccess) e = hipMemset(md->wblk, 0, (size_t)off * sizeof(float));
    if (e != hipSuccess) {
        std::string msg = std::string("allocating LSTM model storage: ") + hipGetErrorString(e);
        delete md;
        return l2a_fail(ctx, L2A_EHIP, msg);
    }
    *out = md;
    return L2A_OK;
}

int l2a_rnn_create(l2a_ctx* ctx, int obs_dim, int act_dim, int n_layers, const int* units, int cell_type,
                   int cell_act, int output_act, l2a_lstm** out) {
    if (!ctx) return L2A_EINVAL;
    if (!out || !units) return l2a_fail(ctx, L2A_EINVAL, "l2a_rnn_create: null argument");
    *out = nullptr;
    if (n_layers < 1 || n_layers > L2A_RNN_MAX_LAYERS)
        return l2a_fail(ctx, L2A_EINVAL, "l2a_rnn_create: 1 to " + std::to_string(L2A_RNN_MAX_LAYERS) + " layers");
    if (cell_type != L2A_CELL_LSTM && cell_type != L2A_CELL_GRU && cell_type != L2A_CELL_RNN)
        return l2a_fail(ctx, L2A_EINVAL, "l2a_rnn_create: unknown cell type");
    // the run_rebal.py configuration, at the widths its tuned kernel is instantiated for: the model of l2a_lstm_create;
    // one LSTM layer of any other width takes the generic matrix-core kernel like the stacks below
    // (L2A_FORCE_GENERIC: developer switch - A/B of the generic kernels against the tuned one on its own shape)
    if (n_layers == 1 && cell_type == L2A_CELL_LSTM && lstm_mfma_eligible(obs_dim, act_dim, units[0]) && !getenv("L2A_FORCE_GENERIC"))
        return l2a_lstm_create(ctx, obs_dim, act_dim, units[0], cell_act, output_act, out);
    if (obs_dim < 1 || act_dim < 1) return l2a_fail(ctx, L2A_EINVAL, "obs_dim and act_dim must be >= 1");
    if (cell_act < 0 || cell_act > L2A_ACT_SWISH || output_act < 0 || output_act > L2A_ACT_SWISH)
        return l2a_fail(ctx, L2A_EINVAL, "unsupported nonlinearity");
    int width = 0;
    for (int l = 0; l < n_layers; ++l) {
        if (units[l] < 1 || units[l] > 1024) return l2a_fail(ctx, L2A_EINVAL, "units must be in [1, 1024]");
        width += units[l];
    }
    l2a_lstm* md = new l2a_lstm();
    md->ctx = ctx;
    md->generic = true;
    md->n_layers = n_layers; md->cell_type = cell_type;
    md->obs_dim = obs_dim; md->act_dim = act_dim; md->in_dim = obs_dim + act_dim; md->units = width;
    md->cell_act = cell_act; md->output_act = output_act;
    md->KG0 = l2a_ceil_div(md->in_dim, 16);
    md->OT = l2a_ceil_div(obs_dim, 16);
    long long off = 0;
    auto take = [&off](long long n) { long long o = off; off += (n + 15) / 16 * 16; return o; };
    // micro tiles (l2a_rnn_micro.h): every layer 256 (or every layer 512) units wide, the input shapes of the tuned LSTM kernel,
    // LDS for the whole stack (256 units: up to three layers; 512: one GRU or BasicRNN layer)
    md->gmicro_ok = lstm_mfma_eligible(obs_dim, act_dim, 256) && (units[0] == 256 || (units[0] == 512 && cell_type != L2A_CELL_LSTM));
    for (int l = 0; l < n_layers; ++l) md->gmicro_ok = md->gmicro_ok && units[l] == units[0];
    md->gmicro_ok = md->gmicro_ok && l2a_rnn_micro_smem(cell_type, n_layers, units[0], md->KG0, 3) <= ctx->lds_per_block;
    // (and only where a 16-candidate kernel fits too - the matrix-core one or, for three-layer stacks, the VALU one: one-step
    // launches - predict, chunk continuations - stay with those)
    if (md->gmicro_ok) {
        const int gates = cell_type == L2A_CELL_LSTM ? 4 : (cell_type == L2A_CELL_GRU ? 3 : 1);
        const long long smem_v = (long long)(md->in_dim + 3 * width + 2 * obs_dim + 1) * L2A_LVT * 4;
        md->gmicro_ok = 4 * l2a_rnn_mfma_lds_floats(md->in_dim, obs_dim, n_layers, units, gates) <= ctx->lds_per_block ||
                        smem_v <= ctx->lds_per_block;
    }
    md->gmicro4_ok = md->gmicro_ok && units[0] == 256 && l2a_rnn_micro_smem(cell_type, n_layers, 256, md->KG0, 4) <= ctx->lds_per_block;
    int kin = md->in_dim;
    for (int l = 0; l < n_layers; ++l) {
        const int U = units[l];
        md->lunits[l] = U;
        const int cols0 = (cell_type == L2A_CELL_LSTM) ? 4 * U : (cell_type == L2A_CELL_GRU ? 2 * U : U);
        md->lw[l][0] = take((long long)(kin + U) * cols0);
        md->lb[l][0] = take(cols0);
        md->lpk[l][0] = take(l2a_rnn_pack_floats(kin, U, cols0 / U, true));
        if (cell_type == L2A_CELL_GRU) {
            md->lw[l][1] = take((long long)(kin + U) * U);
            md->lb[l][1] = take(U);
            md->lpk[l][1] = take(l2a_rnn_pack_floats(kin, U, 1, true));
        }
        if (md->gmicro_ok) {
            md->lmk[l][0] = take(l2a_rnn_micro_floats(kin, U, cols0 / U));
            if (cell_type == L2A_CELL_GRU) md->lmk[l][1] = take(l2a_rnn_micro_floats(kin, U, 1));
        }
        kin = U;
    }
    md->raw_wo = take((long long)kin * obs_dim);
    md->raw_bo = take(obs_dim);
    md->pk_wout = take(l2a_rnn_pack_floats(kin, obs_dim, 1, false));
    if (md->gmicro_ok) md->pk_mo = take((long long)(kin / 4) * 256);
    md->pk_bout = take(16 * md->OT);
    md->nm_off = take(32 * md->KG0 + 32 * md->OT);
    md->total = off;
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&md->wblk), (size_t)off * sizeof(float));
    if (e == hipSuccess) e = hipMemset(md->wblk, 0, (size_t)off * sizeof(float));
    if (e != hipSuccess) {
        std::string msg = std::string("allocating recurrent model storage: ") + hipGetErrorString(e);
        delete md;
        return l2a_fail(ctx, L2A_EHIP, msg);
    }
    *out = md;
    return L2A_OK;
}

void l2a_lstm_destroy(l2a_lstm* md) {
    if (!md) return;
    if (md->wblk) {
        (void)hipDeviceSynchronize();
        (void)hipFree(md->wblk);
    }
    if (md->xbuf) (void)hipFree(md->xbuf);
    if (md->adv_buf) (void)hipFree(md->adv_buf);
    delete md;
}

int l2a_lstm_set_weights(l2a_lstm* md, const void* const* device_ptrs, void* stream_v) {
    if (!md) return L2A_EINVAL;
    l2a_ctx* ctx = md->ctx;
    if (!device_ptrs) return l2a_fail(ctx, L2A_EINVAL, "device_ptrs is null");
    for (int i = 0; i < 4 && !md->generic; ++i)
        if (!device_ptrs[i]) return l2a_fail(ctx, L2A_EINVAL, "null LSTM parameter pointer " + std::to_string(i));
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    l2a_device_guard guard(ctx->device);
    if (md->generic) {
        // per layer: lstm (kernel, bias) | gru (gates/kernel, gates/bias, candidate/kernel, candidate/bias) |
        // rnn (kernel, bias); then output/kernel, output/bias
        int pi = 0, kin = md->in_dim;
        auto copy = [&](long long off, size_t count) -> int {
            const void* src = device_ptrs[pi++];
            if (!src) return l2a_fail(ctx, L2A_EINVAL, "null recurrent parameter pointer " + std::to_string(pi - 1));
            L2A_HIP(ctx, hipMemcpyAsync(md->wblk + off, src, sizeof(float) * count, hipMemcpyDeviceToDevice, stream));
            return L2A_OK;
        };
        for (int l = 0; l < md->n_layers; ++l) {
            const int Ul = md->lunits[l];
            const int cols0 = (md->cell_type == L2A_CELL_LSTM) ? 4 * Ul : (md->cell_type == L2A_CELL_GRU ? 2 * Ul : Ul);
            // (the fragment-order copies are made from the block's own raw copies, in stream order behind them)
            auto pack = [&](long long raw, long long dst, int rows_in, int Uc, int G, int recurrent) {
                const long long total = l2a_rnn_pack_floats(rows_in, Uc, G, recurrent != 0);
                hipLaunchKernelGGL(l2a_rnn_pack_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                                   md->wblk + raw, rows_in, Uc, G, recurrent, total, md->wblk + dst);
            };
            auto pack_micro = [&](long long raw, long long dst, int rows_in, int Uc, int G) {
                if (!md->gmicro_ok) return;
                const long long total = l2a_rnn_micro_floats(rows_in, Uc, G);
                hipLaunchKernelGGL(l2a_rnn_micro_pack_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                                   md->wblk + raw, rows_in, Uc, G, total, md->wblk + dst);
            };
            int rc = copy(md->lw[l][0], (size_t)(kin + Ul) * cols0);
            if (rc == L2A_OK) { pack(md->lw[l][0], md->lpk[l][0], kin, Ul, cols0 / Ul, 1); pack_micro(md->lw[l][0], md->lmk[l][0], kin, Ul, cols0 / Ul); }
            if (rc == L2A_OK) rc = copy(md->lb[l][0], (size_t)cols0);
            if (rc == L2A_OK && md->cell_type == L2A_CELL_GRU) {
                rc = copy(md->lw[l][1], (size_t)(kin + Ul) * Ul);
                if (rc == L2A_OK) { pack(md->lw[l][1], md->lpk[l][1], kin, Ul, 1, 1); pack_micro(md->lw[l][1], md->lmk[l][1], kin, Ul, 1); }
                if (rc == L2A_OK) rc = copy(md->lb[l][1], (size_t)Ul);
            }
            if (rc != L2A_OK) return rc;
            kin = Ul;
        }
        int rc = copy(md->raw_wo, (size_t)kin * md->obs_dim);
        if (rc == L2A_OK) {
            const long long total = l2a_rnn_pack_floats(kin, md->obs_dim, 1, false);
            hipLaunchKernelGGL(l2a_rnn_pack_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                               md->wblk + md->raw_wo, kin, md->obs_dim, 1, 0, total, md->wblk + md->pk_wout);
            if (md->gmicro_ok) {
                const long long total_m = (long long)(kin / 4) * 256;
                hipLaunchKernelGGL(l2a_lstm_micro_pack_out_k, dim3((unsigned)((total_m + 255) / 256)), dim3(256), 0, stream,
                                   md->wblk + md->raw_wo, kin, md->obs_dim, total_m, md->wblk + md->pk_mo);
            }
        }
        if (rc == L2A_OK) rc = copy(md->raw_bo, (size_t)md->obs_dim);
        if (rc != L2A_OK) return rc;
        L2A_HIP(ctx, hipGetLastError());
        md->weights_set = true;
        return L2A_OK;
    }
    const int U = md->units;
    const float* wk = static_cast<const float*>(device_ptrs[0]);
    const float* wo = static_cast<const float*>(device_ptrs[2]);
    L2A_HIP(ctx, hipMemcpyAsync(md->wblk + md->raw_wk, wk, sizeof(float) * (size_t)(md->in_dim + U) * 4 * U,
                                hipMemcpyDeviceToDevice, stream));
    L2A_HIP(ctx, hipMemcpyAsync(md->wblk + md->raw_bk, device_ptrs[1], sizeof(float) * 4 * (size_t)U,
                                hipMemcpyDeviceToDevice, stream));
    L2A_HIP(ctx, hipMemcpyAsync(md->wblk + md->raw_wo, wo, sizeof(float) * (size_t)U * md->obs_dim,
                                hipMemcpyDeviceToDevice, stream));
    L2A_HIP(ctx, hipMemcpyAsync(md->wblk + md->raw_bo, device_ptrs[3], sizeof(float) * (size_t)md->obs_dim,
                                hipMemcpyDeviceToDevice, stream));
    L2A_HIP(ctx, hipMemcpyAsync(md->wblk + md->pk_bout, device_ptrs[3], sizeof(float) * (size_t)md->obs_dim,
                                hipMemcpyDeviceToDevice, stream));
    if (md->mfma_ok) {
        const int UT = U / 16;
        long long total = 4LL * UT * (md->KG0 + UT) * 256;
        hipLaunchKernelGGL(l2a_lstm_pack_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, wk,
                           md->KG0, UT, md->in_dim, total, md->wblk + md->pk_wg);
        L2A_HIP(ctx, hipGetLastError());
        total = (long long)md->OT * UT * 256;
        hipLaunchKernelGGL(l2a_lstm_pack_out_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, wo, U,
                           md->obs_dim, UT, total, md->wblk + md->pk_wout);
        L2A_HIP(ctx, hipGetLastError());
        if (md->micro_ok) {
            total = l2a_lstm_micro_gate_floats(U, md->KG0);
            hipLaunchKernelGGL(l2a_lstm_micro_pack_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, wk,
                               md->in_dim, U, md->KG0, total, md->wblk + md->pk_mg);
            total = (long long)(U / 4) * 256;
            hipLaunchKernelGGL(l2a_lstm_micro_pack_out_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, wo, U,
                               md->obs_dim, total, md->wblk + md->pk_mo);
            L2A_HIP(ctx, hipGetLastError());
        }
    }
    md->weights_set = true;
    return L2A_OK;
}

int l2a_lstm_set_norm(l2a_lstm* md, const double* mean_obs, const double* std_obs, const double* mean_act,
                      const double* std_act, const double* mean_delta, const double* std_delta, void* stream_v) {
    if (!md) return L2A_EINVAL;
    l2a_ctx* ctx = md->ctx;
    const int n_null = !mean_obs + !std_obs + !mean_act + !std_act + !mean_delta + !std_delta;
    if (n_null != 0 && n_null != 6)
        return l2a_fail(ctx, L2A_EINVAL, "pass all six normalisation vectors, or none for identity");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    l2a_device_guard guard(ctx->device);
    const int KG0 = md->KG0, OT = md->OT;
    std::vector<float>& st = md->norm_stage;
    if (!st.empty()) L2A_HIP(ctx, hipStreamSynchronize(stream));
    st.assign((size_t)(32 * KG0 + 32 * OT), 0.0f);
    float* in_mu = st.data();
    float* in_iv = in_mu + 16 * KG0;
    float* out_mu = in_iv + 16 * KG0;
    float* out_sd = out_mu + 16 * OT;
    const double eps = 1e-10;   // rnn_dynamics.py:329-334
    for (int k = 0; k < md->in_dim; ++k) {
        if (n_null) { in_mu[k] = 0.0f; in_iv[k] = 1.0f; continue; }
        const double mu = (k < md->obs_dim) ? mean_obs[k] : mean_act[k - md->obs_dim];
        const double sd = (k < md->obs_dim) ? std_obs[k] : std_act[k - md->obs_dim];
        in_mu[k] = (float)mu;
        in_iv[k] = (float)(1.0 / (sd + eps));
    }
    for (int d = 0; d < md->obs_dim; ++d) {
        out_mu[d] = n_null ? 0.0f : (float)mean_delta[d];
        out_sd[d] = n_null ? 1.0f : (float)(std_delta[d] + eps);
    }
    L2A_HIP(ctx, hipMemcpyAsync(md->wblk + md->nm_off, st.data(), st.size() * sizeof(float), hipMemcpyHostToDevice, stream));
    md->norm_set = true;
    return L2A_OK;
}

int l2a_lstm_plan_rs(l2a_lstm* md, const float* obs0, const float* c0, const float* h0, const float* actions,
                     int m, int n, int h, double discount, const l2a_reward* reward, int cand_offset,
                     float* returns_out, unsigned long long* best_key, void* stream_v) {
    if (!md) return L2A_EINVAL;
    l2a_ctx* ctx = md->ctx;
    if (!obs0 || !c0 || !h0 || !actions || !reward)
        return l2a_fail(ctx, L2A_EINVAL, "l2a_lstm_plan_rs: null obs0/c0/h0/actions/reward");
    if (!best_key && !returns_out) return l2a_fail(ctx, L2A_EINVAL, "l2a_lstm_plan_rs: nothing to write");
    if (m < 1 || n < 1 || h < 1) return l2a_fail(ctx, L2A_EINVAL, "l2a_lstm_plan_rs: m, n and h must be >= 1");
    if ((long long)m * n > 0x3fffffffLL || cand_offset < 0 || (long long)cand_offset + n > 0x7fffffffLL)
        return l2a_fail(ctx, L2A_EINVAL, "l2a_lstm_plan_rs: too many candidates");
    if (reward->w_vel != 0.0f && (reward->vel_index < 0 || reward->vel_index >= md->obs_dim))
        return l2a_fail(ctx, L2A_EINVAL, "reward.vel_index out of range");
    if (reward->dist_coef != 0.0f && (reward->dist_index < 0 || reward->dist_index >= md->obs_dim))
        return l2a_fail(ctx, L2A_EINVAL, "reward.dist_index out of range");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    l2a_device_guard guard(ctx->device);
    if (best_key) L2A_HIP(ctx, hipMemsetAsync(best_key, 0, sizeof(unsigned long long) * (size_t)m, stream));
    L2ALstmParams p;
    fill(md, p);
    p.obs0 = obs0; p.c0 = c0; p.h0 = h0; p.actions = actions;
    p.returns_out = returns_out; p.best_key = best_key;
    p.m = m; p.n = n; p.h = h; p.cand_offset = cand_offset; p.discount = discount; p.rw = *reward;
    return launch(md, p, stream_v);
}

int l2a_lstm_plan_rs_sync(l2a_lstm* md, const float* obs_host, const float* c0, const float* h0, const float* actions,
                          int m, int n, int h, double discount, const l2a_reward* reward, int cand_offset,
                          unsigned long long* keys_host_out, float* c_next, float* h_next, void* stream_v) {
    return l2a_lstm_plan_rs_sync_hook(md, obs_host, c0, h0, actions, m, n, h, discount, reward, cand_offset, keys_host_out, c_next,
                                      h_next, stream_v, nullptr, nullptr);
}

void l2a_lstm_facts(const l2a_lstm* md, l2a_ctx** ctx, int* obs_dim, int* act_dim, int* units) {
    *ctx = md->ctx; *obs_dim = md->obs_dim; *act_dim = md->act_dim; *units = md->units;
}

int l2a_lstm_plan_rs_sync_hook(l2a_lstm* md, const float* obs_host, const float* c0, const float* h0, const float* actions,
                               int m, int n, int h, double discount, const l2a_reward* reward, int cand_offset,
                               unsigned long long* keys_host_out, float* c_next, float* h_next, void* stream_v,
                               l2a_after_launch_fn hook, void* hook_arg, l2a_mail_pending* pending) {
    if (!md) return L2A_EINVAL;
    l2a_ctx* ctx = md->ctx;
    ctx->stamps_us[0] = l2a_now_us();
    if (!obs_host || !c0 || !h0 || !actions || !reward || (!keys_host_out && !pending))
        return l2a_fail(ctx, L2A_EINVAL, "l2a_lstm_plan_rs_sync: null obs / c0 / h0 / actions / reward / keys_host_out");
    if ((!c_next) != (!h_next)) return l2a_fail(ctx, L2A_EINVAL, "l2a_lstm_plan_rs_sync: pass c_next and h_next together");
    if (c_next && (c_next == c0 || h_next == h0))
        return l2a_fail(ctx, L2A_EINVAL, "l2a_lstm_plan_rs_sync: the next state must not alias the current one");
    if (m < 1 || n < 1 || h < 1) return l2a_fail(ctx, L2A_EINVAL, "l2a_lstm_plan_rs_sync: m, n and h must be >= 1");
    if ((long long)m * n > 0x3fffffffLL || cand_offset < 0 || (long long)cand_offset + n > 0x7fffffffLL)
        return l2a_fail(ctx, L2A_EINVAL, "l2a_lstm_plan_rs_sync: too many candidates");
    if (reward->w_vel != 0.0f && (reward->vel_index < 0 || reward->vel_index >= md->obs_dim))
        return l2a_fail(ctx, L2A_EINVAL, "reward.vel_index out of range");
    if (reward->dist_coef != 0.0f && (reward->dist_index < 0 || reward->dist_index >= md->obs_dim))
        return l2a_fail(ctx, L2A_EINVAL, "reward.dist_index out of range");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    l2a_device_guard guard(ctx->device);
    int kind = ctx->kernel_kind;
    if (kind == L2A_KERNEL_AUTO) kind = md->mfma_ok ? L2A_KERNEL_MFMA : L2A_KERNEL_VALU;
    // only the matrix-core kernels have the mailbox epilogue
    const bool publish = md->generic ? generic_uses_mfma(md) : (kind == L2A_KERNEL_MFMA);
    if (c_next && !md->adv_buf)
        L2A_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&md->adv_buf), sizeof(float) * L2A_MAIL_KEYS * (md->act_dim + md->obs_dim)));
    l2a_mail_ticket tk;
    int rc = l2a_mail_begin(ctx, m, obs_host, (long long)m * md->obs_dim, stream, &tk);
    if (rc != L2A_OK) return rc;
    tk.shape = (unsigned long long)(size_t)md ^ ((unsigned long long)m << 48) ^ ((unsigned long long)n << 24) ^ (unsigned long long)h;
    L2ALstmParams p;
    fill(md, p);
    p.obs0 = tk.obs_dev; p.c0 = c0; p.h0 = h0; p.actions = actions;
    p.best_key = tk.keys_dev;
    p.m = m; p.n = n; p.h = h; p.cand_offset = cand_offset; p.discount = discount; p.rw = *reward;
    if (publish) {
        p.done_ctr = ctx->done_ctr;
        p.mail_keys = ctx->mail_dev->keys;
        p.mail_seq_ptr = &ctx->mail_dev->seq;
        p.mail_seq = tk.seq;
        p.next_keys = tk.next_keys;
    }
    ctx->stamps_us[1] = l2a_now_us();
    rc = launch(md, p, stream_v);
    if (rc == L2A_OK && c_next && advance_kernel_ok(md)) {
        // the controller's own state moves on with the chosen action (rnn_mpc_controller.py:63) - in stream order behind the
        // plan, without waiting for the host to learn the arg-max: ONE small launch that gathers the winners' first actions
        // through the keys and spreads the gate matrix over units / 16 workgroups (l2a_lstm_advance_k)
        rc = launch_advance(md, tk.obs_dev, nullptr, tk.keys_dev, actions, n, cand_offset, c0, h0, c_next, h_next, m, stream);
    } else if (rc == L2A_OK && c_next) {
        // generic cells / stacks: a gather launch and a one-step launch of the rollout kernel
        float* act_sel = md->adv_buf;
        float* obs_next = md->adv_buf + (size_t)L2A_MAIL_KEYS * md->act_dim;
        hipLaunchKernelGGL(l2a_gather_best_k, dim3((unsigned)m), dim3(64), 0, stream, tk.keys_dev, actions, m, n, cand_offset,
                           md->act_dim, act_sel);
        L2ALstmParams q;
        fill(md, q);
        q.obs0 = tk.obs_dev; q.c0 = c0; q.h0 = h0; q.actions = act_sel;
        q.state_out = obs_next; q.c_out = c_next; q.h_out = h_next;
        q.obs_per_row = 1; q.hid_per_row = 1;
        q.m = 1; q.n = m; q.h = 1; q.discount = 1.0;
        // NEVER tile-split: l2a_mail_end reads the status word as soon as the PLAN has published, while this launch may
        // still run - an exchange timeout in it would be noticed one call late and the state it wrote adopted
        rc = launch(md, q, stream_v, false);
    }
    ctx->stamps_us[2] = l2a_now_us();
    if (rc == L2A_OK && hook) hook(hook_arg);
    ctx->stamps_us[3] = l2a_now_us();
    if (pending && rc == L2A_OK) {
        pending->tk = tk; pending->publish = publish; pending->m = m; pending->stream = stream;
        pending->who = "l2a_lstm_plan_rs_sync"; pending->live = true;
        return L2A_OK;
    }
    rc = l2a_mail_end(ctx, tk, m, publish, rc, stream, keys_host_out, "l2a_lstm_plan_rs_sync");
    ctx->stamps_us[4] = l2a_now_us();
    return rc;
}

int l2a_lstm_plan_rs_chunk(l2a_lstm* md, const float* state, const float* c, const float* h, int per_row,
                           const float* actions, int m, int n, int h_chunk, int t0, double discount,
                           const l2a_reward* reward, int cand_offset, const float* returns_in, float* returns_out,
                           float* state_out, float* c_out, float* h_out, unsigned long long* best_key, void* stream_v) {
    if (!md) return L2A_EINVAL;
    l2a_ctx* ctx = md->ctx;
    if (!state || !c || !h || !actions || !reward)
        return l2a_fail(ctx, L2A_EINVAL, "l2a_lstm_plan_rs_chunk: null state/c/h/actions/reward");
    if (!returns_out) return l2a_fail(ctx, L2A_EINVAL, "l2a_lstm_plan_rs_chunk: returns_out is required");
    if (m < 1 || n < 1 || h_chunk < 1 || t0 < 0)
        return l2a_fail(ctx, L2A_EINVAL, "l2a_lstm_plan_rs_chunk: bad m / n / h_chunk / t0");
    if (t0 > 0 && (!returns_in || !per_row))
        return l2a_fail(ctx, L2A_EINVAL, "l2a_lstm_plan_rs_chunk: a continuation needs returns_in and per-row states");
    if ((!state_out) != (!c_out) || (!state_out) != (!h_out))
        return l2a_fail(ctx, L2A_EINVAL, "l2a_lstm_plan_rs_chunk: pass state_out, c_out and h_out together");
    if ((long long)m * n > 0x3fffffffLL || cand_offset < 0 || (long long)cand_offset + n > 0x7fffffffLL)
        return l2a_fail(ctx, L2A_EINVAL, "l2a_lstm_plan_rs_chunk: too many candidates");
    if (reward->w_vel != 0.0f && (reward->vel_index < 0 || reward->vel_index >= md->obs_dim))
        return l2a_fail(ctx, L2A_EINVAL, "reward.vel_index out of range");
    if (reward->dist_coef != 0.0f && (reward->dist_index < 0 || reward->dist_index >= md->obs_dim))
        return l2a_fail(ctx, L2A_EINVAL, "reward.dist_index out of range");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    l2a_device_guard guard(ctx->device);
    if (best_key) L2A_HIP(ctx, hipMemsetAsync(best_key, 0, sizeof(unsigned long long) * (size_t)m, stream));
    L2ALstmParams p;
    fill(md, p);
    p.obs0 = state; p.c0 = c; p.h0 = h; p.actions = actions;
    p.obs_per_row = per_row ? 1 : 0; p.hid_per_row = per_row ? 1 : 0;
    p.returns_out = returns_out; p.best_key = best_key;
    p.state_out = state_out; p.c_out = c_out; p.h_out = h_out;
    p.ret_in = (t0 > 0) ? returns_in : nullptr;
    double d0 = 1.0;
    for (int t = 0; t < t0; ++t) d0 *= discount;
    p.disc0 = d0;
    p.m = m; p.n = n; p.h = h_chunk; p.cand_offset = cand_offset; p.discount = discount; p.rw = *reward;
    return launch(md, p, stream_v);
}

int l2a_lstm_advance(l2a_lstm* md, const float* obs, const float* act, const float* c, const float* h, int rows,
                     float* c_out, float* h_out, void* stream_v) {
    if (!md) return L2A_EINVAL;
    l2a_ctx* ctx = md->ctx;
    if (!obs || !act || !c || !h || !c_out || !h_out) return l2a_fail(ctx, L2A_EINVAL, "l2a_lstm_advance: null pointer");
    if (rows < 1) return l2a_fail(ctx, L2A_EINVAL, "l2a_lstm_advance: rows must be >= 1");
    if (c_out == c || h_out == h) return l2a_fail(ctx, L2A_EINVAL, "l2a_lstm_advance: the next state must not alias the current one");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    l2a_device_guard guard(ctx->device);
    if (advance_kernel_ok(md)) return launch_advance(md, obs, act, nullptr, nullptr, 0, 0, c, h, c_out, h_out, rows, stream);
    // generic cells / stacks (and the VALU kernel when asked for): one step of the rollout kernel, the predicted observation dropped
    if ((long long)rows > L2A_MAIL_KEYS) return l2a_fail(ctx, L2A_EINVAL, "l2a_lstm_advance: at most 64 rows (use l2a_lstm_predict)");
    if (!md->adv_buf)
        L2A_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&md->adv_buf), sizeof(float) * L2A_MAIL_KEYS * (md->act_dim + md->obs_dim)));
    L2ALstmParams p;
    fill(md, p);
    p.obs0 = obs; p.c0 = c; p.h0 = h; p.actions = act;
    p.state_out = md->adv_buf + (size_t)L2A_MAIL_KEYS * md->act_dim; p.c_out = c_out; p.h_out = h_out;
    p.obs_per_row = 1; p.hid_per_row = 1;
    p.m = 1; p.n = rows; p.h = 1; p.discount = 1.0;
    return launch(md, p, stream_v, false);
}

int l2a_lstm_predict(l2a_lstm* md, const float* obs, const float* act, const float* c, const float* h, int rows,
                     float* next_obs_out, float* c_out, float* h_out, void* stream_v) {
    if (!md) return L2A_EINVAL;
    l2a_ctx* ctx = md->ctx;
    if (!obs || !act || !c || !h || !next_obs_out || !c_out || !h_out)
        return l2a_fail(ctx, L2A_EINVAL, "l2a_lstm_predict: null pointer");
    if (rows < 1) return l2a_fail(ctx, L2A_EINVAL, "l2a_lstm_predict: rows must be >= 1");
    L2ALstmParams p;
    fill(md, p);
    p.obs0 = obs; p.c0 = c; p.h0 = h; p.actions = act;
    p.state_out = next_obs_out; p.c_out = c_out; p.h_out = h_out;
    p.obs_per_row = 1; p.hid_per_row = 1;
    p.m = 1; p.n = rows; p.h = 1; p.discount = 1.0;
    return launch(md, p, stream_v);
}

}  // extern "C"
