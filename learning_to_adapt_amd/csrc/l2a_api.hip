// l2a_api.hip - host side of libl2a_hip.so: the C ABI declared in include/l2a.h.
//
// Owns: the context (device facts, kernel selection), the model (one HBM block per weight set
// holding raw + MFMA-packed weights and the padded normalisation vectors) and the launch logic
// of the fused rollout kernels in l2a_kernels.h.  Everything is enqueued on the caller's stream.

#define L2A_PACK_KERNELS 1      // this unit launches the weight re-packing kernels of l2a_micro_pack.h
#include "l2a_host.h"
#include "l2a_kernels.h"
#include "l2a_valu.h"
#include "l2a_adapt.h"
#include "l2a_mfma_launch.h"
#include "l2a_micro_pack.h"
#include "l2a_micro_launch.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <immintrin.h>
#include <unistd.h>

namespace {

std::string g_init_error;

}  // namespace

int l2a_fail(const l2a_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg; else g_init_error = msg;
    return code;
}

struct l2a_model {
    l2a_ctx* ctx = nullptr;
    int obs_dim = 0, act_dim = 0, in_dim = 0;
    int n_hidden = 0;
    int hidden[L2A_MAX_LAYERS] = {0};
    int hidden_act = L2A_ACT_RELU, output_act = L2A_ACT_IDENTITY;
    int n_sets = 1, mode = L2A_MODE_SINGLE;
    bool mfma_ok = false;
    int H = 0, TPW = 0, KG0 = 0, OT = 0, hmax = 0;
    float* wblk = nullptr;
    long long set_stride = 0;
    long long raw_w[L2A_MAX_LAYERS] = {0};
    long long raw_b[L2A_MAX_LAYERS] = {0};
    long long pk_w0 = 0, pk_wmid = 0, pk_wmid_stride = 0, pk_wout = 0, pk_bout = 0, nm_off = 0;
    bool micro_ok = false;                        // the micro-tile kernel of l2a_micro.h has an instance (hidden width 256 / 512)
    int m_o4 = 0;                                 // ... and sums dims 16 .. 19 per quarter of the hidden units, like the 16-candidate O4 instance
    long long pk_m = 0;                           // its copy of a set's weights (wave-stream order, l2a_micro_pack.h)
    std::vector<char> weights_set, norm_set;
    std::vector<std::vector<float>> norm_stage;   // host staging, kept alive for async H2D
    unsigned long long* xbuf = nullptr;           // member-split exchange granules
    long long xbuf_granules = 0;
    unsigned int launch_nonce = 0;
    float* adapt_scratch = nullptr;               // l2a_model_adapt_sgd: layer inputs and dZ of every task
    long long adapt_scratch_floats = 0;
    // l2a_model_adapt_sgd_host: two staging slots (host-mapped, read by the kernels directly)
    struct adapt_slot {
        float* stage_host = nullptr;
        float* stage_dev = nullptr;
        long long stage_floats = 0;
        hipEvent_t done = nullptr;
        bool pending = false;                     // `done` was recorded and not waited for yet
    } aslot[2];
    int aslot_next = 0;
};

namespace {

inline bool fast_act(int a) { return a == L2A_ACT_RELU || a == L2A_ACT_IDENTITY; }

inline int fail(const l2a_ctx* ctx, int code, const std::string& msg) { return l2a_fail(ctx, code, msg); }

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

bool mfma_eligible(int obs_dim, int act_dim, int n_hidden, const int* hidden) {
    if (n_hidden < 1 || n_hidden > L2A_MAX_LAYERS - 1) return false;
    const int H = hidden[0];
    if (H != 128 && H != 256 && H != 512) return false;
    for (int i = 1; i < n_hidden; ++i)
        if (hidden[i] != H) return false;
    if (obs_dim < 1 || obs_dim > 16 * L2A_OTMAX) return false;
    if (act_dim < 1 || act_dim > 16) return false;
    if (obs_dim + act_dim > 16 * L2A_KG0MAX) return false;
    return true;
}

long long packed_floats(int k_in, int n_out) {
    return (long long)ceil_div(n_out, 16) * ceil_div(k_in, 16) * 256;
}

template <typename K>
int allow_big_lds(const l2a_ctx* ctx, K kernel, int bytes) {
    L2A_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    return L2A_OK;
}

// Which tile-split flavour a model / policy allows: 2 = groups + shared middle (or only) set, 1 = whole
// sets, 0 = none.
int split_mode_for(const l2a_model* md, int e_loop) {
    const int policy = md->ctx->split_policy;
    if (policy == 0) return 0;
    if ((e_loop & 1) && md->n_hidden >= 2 && policy == 1) return 2;
    return (e_loop >= 2) ? 1 : 0;
}

// Launch geometry of the MFMA kernel for (m, n): NT candidate tiles per workgroup.  Costs in units of
// "one workgroup running one tile for the whole horizon": NT = 2 costs 1.9, a shared tail round ~0.8.
int choose_nt(const l2a_model* md, int m, int n, int e_loop, int sa_bytes_nt2, int other_bytes_nt2) {
    const int cus = md->ctx->num_cu > 0 ? md->ctx->num_cu : 256;
    if (2 * sa_bytes_nt2 + other_bytes_nt2 > md->ctx->lds_per_block) return 1;
    // Two candidate tiles per workgroup pay only where their instance keeps its registers: at hidden width 512 every NT = 2
    // instance spilled (55 - 388 VGPRs to scratch) and lost to NT = 1 on every multi-round plan measured - Ant 2 x 512,
    // n = 8000: 2.51 against 2.13 ms, config 4's 16 000 candidates: 3.53 against 3.50 ms per 10 steps (profiles/r04_ab_nt.jsonl)
    // - and so do the 49 - 64-dimensional observations at width 256.  Those shapes run NT = 1 (whole rounds + a shared tail);
    // the (2, 8) instances are no longer built.
    if (md->TPW >= 8 || md->OT >= 4) return 1;      // (no NT = 2 instance exists for OT >= 4: l2a_mfma_inst.hip)
    static const int force_nt = [] { const char* e = std::getenv("L2A_FORCE_NT"); return (e && (e[0] == '1' || e[0] == '2')) ? e[0] - '0' : 0; }();
    if (force_nt) return force_nt;      // developer A/B (tools/ab_nt.py)
    const long long wg1 = (long long)m * ceil_div(n, 16);
    const long long wg2 = (long long)m * ceil_div(n, 32);
    const long long t = wg1 % cus;
    double cost1 = (double)((wg1 + cus - 1) / cus);
    if (wg1 > cus && t != 0 && 2 * t <= cus && split_mode_for(md, e_loop) != 0) cost1 = (double)(wg1 / cus) + 0.8;
    const double cost2 = (double)((wg2 + cus - 1) / cus) * 1.9;
    return (cost2 < cost1) ? 2 : 1;
}

int sa_elems_for(const l2a_model* md, int nt) {
    const int ht = md->H / 16;
    const int a = nt * ht;
    const int b = L2A_NW * nt * md->OT;         // output-layer partials (one per wave)
    return (a > b ? a : b) * 64;
}

int launch_rollout(l2a_model* md, L2AKParams& p, void* stream_v) {
    l2a_ctx* ctx = md->ctx;
    l2a_device_guard guard(ctx->device);
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    const int sets_needed = ctx->dry ? 0 : (p.mode == L2A_MODE_PER_BLOCK) ? (p.m < md->n_sets ? p.m : md->n_sets) : md->n_sets;
    for (int e = 0; e < sets_needed; ++e) {
        if (!md->weights_set[e]) return fail(ctx, L2A_ESTATE, "weight set " + std::to_string(e) + " was never set");
        if (!md->norm_set[e]) return fail(ctx, L2A_ESTATE, "normalisation of set " + std::to_string(e) + " was never set");
    }
    if (p.mode == L2A_MODE_PER_BLOCK && p.m > md->n_sets)
        return fail(ctx, L2A_EINVAL, "per-block mode needs one weight set per env/block");
    p.c_lo = 0; p.c_hi = p.n; p.done_total = 0;     // one launch covers the plan unless the MFMA branch below cuts it in two

    int kind = ctx->kernel_kind;
    if (kind == L2A_KERNEL_AUTO) kind = md->mfma_ok ? L2A_KERNEL_MFMA : L2A_KERNEL_VALU;
    if (kind == L2A_KERNEL_MFMA && !md->mfma_ok)
        return fail(ctx, L2A_EINVAL, "model shape is not eligible for the MFMA kernel "
                                     "(needs equal hidden widths of 128/256/512, obs_dim<=64, act_dim<=16)");

    // ---- matrix-core kernels: where the plan is cut, and which geometry each part takes -----------------------------------
    const int e_loop0 = (p.mode == L2A_MODE_MEAN) ? md->n_sets : 1;
    const int cus0 = ctx->num_cu > 0 ? ctx->num_cu : 256;
    const int cst_set = 32 * md->KG0 + 48 * md->OT + md->n_hidden * md->H;
    // LDS plan of a launch whose busiest workgroup runs `nseq` sets in sequence: sets per batch, start of the constants,
    // bytes (l2a_mfma.h: activation regions | chunk partials of a batch (lb > 1) | constants | exchange staging).
    // lb = sets per batch: with two hidden layers, layer 0 of up to lb sets runs back to back, then their hidden
    // GEMMs + output layers, then their reduces - two barriers per batch instead of two per set.  The largest lb
    // (<= 4, <= the longest set sequence of a workgroup) that fits the CU's LDS; 1 = one set at a time.
    auto lds_plan = [&](int nt_, int nseq, int x_bytes, int* lb_out, int* cst_off_out) {
        const int sa = sa_elems_for(md, nt_);
        const int ps_bytes = L2A_NW * nt_ * md->OT * 64 * 16;       // the waves' output-layer partials of one set
        const int cst_bytes_all = nseq * cst_set * 4;
        int lb_ = 1, part_bytes = 0;
        if (md->n_hidden == 2 && ctx->batch_sets != 1) {
            const int cap = ctx->batch_sets > 0 ? ctx->batch_sets : 4;
            for (int lb = (nseq < cap ? nseq : cap); lb >= 2; --lb) {
                const int pbytes = lb * ps_bytes;
                if (lb * sa * 16 + pbytes + cst_bytes_all + x_bytes <= ctx->lds_per_block) {
                    lb_ = lb;
                    part_bytes = pbytes;
                    break;
                }
            }
        }
        *lb_out = lb_;
        *cst_off_out = ((lb_ > 1 ? lb_ : 2) * sa * 16 + part_bytes) / 16;
        return *cst_off_out * 16 + cst_bytes_all + x_bytes;
    };
    // Double rounds (round 6): a plan of at least two rounds of 16-candidate tiles at hidden width 512 runs its first
    // 2 x CUs x D tiles as D rounds of DOUBLE tiles on the whole-tiles-only instances (two candidate tiles per workgroup:
    // every weight fragment feeds both, the step's fixed costs - barriers, phase hand-overs, operand waits - are paid once
    // for 32 candidates; these instances carry neither exchange nor half-member code, so unlike the general NT = 2 instances
    // at this width they keep their registers).  The rest of every env's candidates follows in a second launch with whichever
    // geometry fits it (micro tiles, whole round, tile split); a rest of more than a round and a half joins the double tiles.
    // Per tile a double round costs ~0.93 of a single one (profiles/r06_ab_double.jsonl).  Same arithmetic per candidate:
    // bit-identical results.
    int front_k = 0;                    // double tiles per env of the front launch
    if (kind == L2A_KERNEL_MFMA) {
        const long long tiles1 = (long long)p.m * ceil_div(p.n, 16);
        int lb2 = 1, off2 = 0;
        if (ctx->double_policy != 0 && md->TPW == 8 && md->OT <= 3 && tiles1 >= 2LL * cus0 &&
            lds_plan(2, e_loop0, 0, &lb2, &off2) <= ctx->lds_per_block) {
            const long long D = tiles1 / (2LL * cus0);
            const long long R = tiles1 - 2LL * cus0 * D;
            if (R == 0 || 2 * R > 3LL * cus0) front_k = ceil_div(p.n, 32);       // everything on double tiles
            else front_k = (int)((cus0 * D) / p.m);
            if (32LL * front_k >= p.n) front_k = ceil_div(p.n, 32);
        }
    }
    const int n_front = (32LL * front_k >= p.n) ? (front_k ? p.n : 0) : 32 * front_k;
    const int n_rest = p.n - n_front;

    // Micro tiles (l2a_micro.h) for the plan - or for the rest behind a double round: every env's ceil(n / 4) candidate tiles of
    // four dealt to W workgroups of at most three - one workgroup per CU, none idle, no exchange between workgroups.  Plans only
    // (no per-row start states, no state written out: those launches are one step long or chunk continuations and keep the
    // 16-candidate kernel - same bits).
    bool micro = false;
    int mc_W = 0, mc_hi = 0, mc_quads = 0, smem_m = 0;
    long long mc_span = 0;
    if (kind == L2A_KERNEL_MFMA && n_rest > 0 && md->micro_ok && ctx->micro_policy != 0) {
        const long long tiles16 = (long long)p.m * ceil_div(n_rest, 16);
        const int quads = ceil_div(n_rest, 4);
        int W = cus0 / p.m;
        if (W > quads) W = quads;
        const int hi = W > 0 ? ceil_div(quads, W) : 99;
        // ... and the FEWEST workgroups that keep the largest one at `hi` micro tiles: under this kernel the chip is power limited
        // (2.1 GHz with 255 busy CUs against 2.37 with 125: tools/timeline_micro.py), and a workgroup of fewer micro tiles streams
        // the same weights for less work - 2 500 candidates on 5 x 42 workgroups of 12 instead of 5 x 51 of 12 and 8: c3b 0.325 ->
        // 0.304 ms, the ReBAL default 0.195 -> 0.183 ms (profiles/r04_ab_micro.jsonl)
        if (hi >= 1 && hi <= 3) W = ceil_div(quads, hi);
        const int smem_need = l2a_mlp_micro_smem(md->H, md->KG0, md->n_hidden, e_loop0);
        const long long span = (long long)((p.mode == L2A_MODE_PER_BLOCK ? p.m : e_loop0) - 1) * md->set_stride * 4 +
                               l2a_mlp_micro_floats(md->H, md->KG0, md->n_hidden) * 4;
        const bool eligible = !p.obs_per_row && !p.state_out && hi <= 3 && (p.returns_out || p.best_key) &&
                              smem_need <= ctx->lds_per_block && span < (1LL << 31);
        // automatic (profiles/r04_ab_micro.jsonl, hidden width 512): the plans the 16-candidate geometries cannot fill - more
        // than CUs / 2 tiles (no tile split) and fewer than CUs: 0.80 - 0.90 of the 16-candidate launch (per-block 3 x 512 /
        // 2 x 512, mean E = 5, single) - and plans of at most CUs / 2 tiles that run ONE set per candidate (single model,
        // per-block), whose tile split is bound by its per-step exchange: 0.77 - 0.91 (config 3's rest behind its double round,
        // 5 x 23 tiles: 0.312 -> 0.266 ms, profiles/r06_ab_rest.jsonl).  Small ENSEMBLE plans stay with the tile
        // split (every workgroup would stream all E sets: 1.2x), and so does hidden width 256 (one 64-unit tile per wave: 1.12x).
        const bool unfilled = 2 * tiles16 > cus0 && tiles16 < cus0;
        const bool small_one_set = e_loop0 == 1 && 2 * tiles16 <= cus0;
        const bool wanted = ctx->micro_policy == 2 || (ctx->micro_policy == 1 && md->H == 512 && (unfilled || small_one_set));
        if (eligible && wanted) {
            micro = true;
            mc_W = W; mc_hi = hi; mc_quads = quads; mc_span = span;
            smem_m = smem_need;
            if (smem_m < 84 * 1024) smem_m = 84 * 1024;     // more than half a CU's LDS: one workgroup per CU
        }
    }
    const int cst_bytes = e_loop0 * cst_set * 4;
    int nt = (kind == L2A_KERNEL_MFMA && n_rest > 0 && !micro)
                 ? choose_nt(md, p.m, n_rest, e_loop0, sa_elems_for(md, 2) * 16, cst_bytes + 4 * md->OT * 64 * 16) : 1;
    bool front_has_dbg = false;
    int front_wg = 0;
    if (front_k) {
        L2AKParams pa = p;
        pa.c_lo = 0; pa.c_hi = n_front;
        pa.sa_elems = sa_elems_for(md, 2);
        pa.tiles_per_env = front_k;
        pa.cst_set = cst_set;
        pa.split = 0; pa.split_from = -1; pa.pl_units = 0;
        pa.n_cst = e_loop0;
        // (phase stamps, tools/timeline.py: one launch writes them - the rest by default, the double tiles with L2A_DBG_FRONT=1)
        static const bool dbg_front = [] { const char* e = std::getenv("L2A_DBG_FRONT"); return e && e[0] == '1'; }();
        pa.dbg = (n_rest == 0 || dbg_front) ? ctx->dbg : nullptr;
        front_has_dbg = pa.dbg != nullptr;
        const int smem_a = lds_plan(2, e_loop0, 0, &pa.lb, &pa.cst_off);
        const long long wg_a = (long long)p.m * front_k;
        // what the mailbox counts: one per workgroup that owns a tile's (or micro-tile group's) key, over both launches
        const long long rest_units = n_rest == 0 ? 0 : micro ? (long long)p.m * mc_W : (long long)p.m * ceil_div(n_rest, 16 * nt);
        pa.done_total = (int)(wg_a + rest_units);
        p.done_total = pa.done_total;
        if (ctx->dry) {
            if (n_rest == 0) {
                const int g[12] = {1, 2, 0, -1, 0, (int)wg_a, smem_a, pa.lb, 0, 0, 0, 1};
                std::memcpy(ctx->dry, g, sizeof(g));
                return L2A_OK;
            }
        } else {
            const bool gact_a = !(fast_act(md->hidden_act) && fast_act(md->output_act));
            const int rc_a = l2a_launch_mfma(2, md->TPW, md->OT, md->KG0, gact_a ? 1 : 0, 2, &pa, (unsigned)wg_a, smem_a, stream);
            if (rc_a == -100) return fail(ctx, L2A_EINVAL, "no whole-tile MFMA kernel instance for this (obs_dim, act_dim, hidden)");
            if (rc_a != 0) return fail(ctx, L2A_EHIP, std::string("MFMA kernel launch (double rounds): ") + hipGetErrorString((hipError_t)rc_a));
            L2A_HIP(ctx, hipGetLastError());
            if (n_rest == 0) return L2A_OK;
        }
        p.c_lo = n_front;
        front_wg = (int)wg_a;
    }
    if (micro) {
        p.mc_w = mc_W;
        p.mc_hi = mc_hi;
        p.mc_r = mc_quads - mc_W * (mc_hi - 1);          // workgroups that take `hi` micro tiles
        p.m_bytes = mc_span;
        p.dbg = front_has_dbg ? nullptr : ctx->dbg;
        const bool gact_m = !(fast_act(md->hidden_act) && fast_act(md->output_act)) || md->n_hidden == 1;
        if (ctx->dry) {     // [kind, nt, split, split_from, fan, workgroups, lds bytes, sets per batch, micro tiles of the largest workgroup, placement units, double-tile workgroups in front, 0]
            const int g[12] = {2, 0, 0, -1, 0, p.m * mc_W, smem_m, 1, mc_hi, 0, front_wg, 0};
            std::memcpy(ctx->dry, g, sizeof(g));
            return L2A_OK;
        }
        const int rc = l2a_launch_mlp_micro(md->H, gact_m ? 1 : 0, &p, (unsigned)(p.m * mc_W), smem_m, stream);
        if (rc != 0) return fail(ctx, L2A_EHIP, std::string("micro-tile MLP kernel launch: ") +
                                                    (rc > 0 ? hipGetErrorString((hipError_t)rc) : "no instance"));
        L2A_HIP(ctx, hipGetLastError());
        return L2A_OK;
    }
    if (kind == L2A_KERNEL_MFMA) {
        // Member fan (l2a_mfma.h): E workgroups per tile, one set each - small mean-ensemble plans, e.g. one rank's shard of
        // config 5 (n = 500: 32 tiles -> 160 workgroups of one set instead of 64 of 2.5 sets).  With two candidate tiles per
        // workgroup (width 512, up to 48 observation dims) it reaches twice as far: one of FOUR ranks' shard of config 5
        // (n = 1000: 32 double tiles x 5) - each weight fragment feeds two tiles and the step's fixed costs are paid once for both.
        int fan_nt = 0;
        if (!front_k && ctx->split_policy != 0 && ctx->fan_policy != 0 && p.h < 4096 && p.mode == L2A_MODE_MEAN && e_loop0 >= 3 && e_loop0 <= 8) {
            if ((long long)p.m * ceil_div(p.n, 16) * e_loop0 <= cus0) fan_nt = 1;
            else if (md->TPW == 8 && md->OT <= 3 && (long long)p.m * ceil_div(p.n, 32) * e_loop0 <= cus0 &&
                     2 * sa_elems_for(md, 2) * 16 + (32 * md->KG0 + 48 * md->OT + md->n_hidden * md->H) * 4 + e_loop0 * 2 * md->OT * 64 * 16 <= ctx->lds_per_block)
                fan_nt = 2;
        }
        if (fan_nt) nt = fan_nt;
        p.sa_elems = sa_elems_for(md, nt);
        p.tiles_per_env = ceil_div(n_rest, 16 * nt);
        const int e_loop = (p.mode == L2A_MODE_MEAN) ? md->n_sets : 1;
        p.cst_set = cst_set;
        // Member split: two workgroups per candidate tile (group A | group B of the ensemble) when
        // that still fits one workgroup per CU - e.g. config 2: 125 tiles -> 250 workgroups.
        const long long pairs = (long long)p.m * p.tiles_per_env;
        const int cus = ctx->num_cu > 0 ? ctx->num_cu : 256;
        // 1 = group A | group B; 2 = additionally the middle set of an odd ensemble (or the single
        // set of a lone model) is shared: each workgroup runs the last hidden layer and the output
        // layer for one half of its hidden tiles (needs >= 2 hidden layers).
        p.split = 0;
        p.split_from = -1;
        long long split_pairs = 0;          // tiles shared by two workgroups
        auto split_mode = [&]() { return split_mode_for(md, e_loop); };
        const bool fan = fan_nt != 0;
        if (fan) {
            p.split = 3;
            split_pairs = pairs;
        } else if (ctx->split_policy != 0 && nt == 1 && p.h < 4096) {
            if (2 * pairs <= cus) {
                p.split = split_mode();
                split_pairs = p.split ? pairs : 0;
            } else if (pairs > cus && pairs % cus != 0 && 2 * (pairs % cus) <= cus && split_mode() != 0) {
                // Tail split: a multi-round plan whose last round would fill less than half of the chip -
                // the left-over tiles are shared by two workgroups each (dispatched last, see l2a_mfma.h).
                p.split = split_mode();
                split_pairs = pairs % cus;
                p.split_from = (int)(pairs - split_pairs);
            }
        }
        if (p.split && !ctx->dry) {
            // in 8-byte units: per shared tile 2 workgroups x 2 step parities x 2 regions (fan: E workgroups x 2 parities)
            const long long need = split_pairs * (fan ? e_loop : 4) * 2 * (long long)(nt * md->OT * 2 * 64 * 16) / 8;
            if (need > md->xbuf_granules) {
                if (md->xbuf) { L2A_HIP(ctx, hipStreamSynchronize(stream)); L2A_HIP(ctx, hipFree(md->xbuf)); md->xbuf = nullptr; }
                L2A_HIP(ctx, l2a_xbuf_alloc(reinterpret_cast<void**>(&md->xbuf), (size_t)need * 8));
                L2A_HIP(ctx, hipMemsetAsync(md->xbuf, 0, (size_t)need * 8, stream));   // ordered before the launch
                md->xbuf_granules = need;
                md->launch_nonce = 0;
            }
            md->launch_nonce += 1;
            if (md->launch_nonce >= (1u << 20)) {     // tag space exhausted: wipe stale tags, restart
                L2A_HIP(ctx, hipMemsetAsync(md->xbuf, 0, (size_t)md->xbuf_granules * 8, stream));
                md->launch_nonce = 1;
            }
            p.xtag = md->launch_nonce << 12;
            p.xbuf = md->xbuf;
            p.status = ctx->status_dev;
            p.spin_limit = ctx->spin_limit;
        }
        p.dbg = front_has_dbg ? nullptr : ctx->dbg;
        const bool uniform_split = p.split && p.split_from < 0;
        const int e_half = (e_loop + 1) / 2;
        const int nseq = fan ? 1 : uniform_split ? e_half : e_loop;  // sets the busiest workgroup runs in sequence
        const int x_bytes = (fan ? e_loop : 2) * nt * md->OT * 64 * 16;   // exchange staging: the partner's two regions / every member's term
        p.n_cst = nseq;
        const int smem = lds_plan(nt, nseq, x_bytes, &p.lb, &p.cst_off);
        if (smem > ctx->lds_per_block)
            return fail(ctx, L2A_EINVAL, "LDS budget exceeded (" + std::to_string(smem) + " B)");
        long long n_wg = fan ? pairs * e_loop : pairs + (p.split ? split_pairs : 0);
        p.pl_units = 0;
        if (ctx->xcd_align && p.split_from < 0 && n_wg <= cus) {
            // one unit of weight-sharing workgroups per XCD (see the kernel's geometry): the two groups of a split
            // ensemble, the environments of a per-block plan (both workgroups of a split tile stream the same set)
            int units = 0;
            long long w = 0;
            if (p.mode == L2A_MODE_PER_BLOCK) { units = p.m; w = (long long)p.tiles_per_env * (p.split ? 2 : 1); }
            else if (fan) { units = e_loop; w = pairs; }      // every member's workgroups on their own XCD(s)
            else if (p.split && e_loop > 1) { units = 2; w = pairs; }
            if (units >= 2 && units <= 8) {
                const int f = 8 / units;
                const long long slots = (w + f - 1) / f;       // per XCD, of the units with f XCDs
                if (8 * slots <= cus) {
                    p.pl_units = units; p.pl_f = f; p.pl_r = 8 - units * f; p.pl_w = (int)w;
                    n_wg = 8 * slots;
                }
            }
        }
        const dim3 grid((unsigned)n_wg), block(64 * L2A_NW);
        const bool gact = !(fast_act(md->hidden_act) && fast_act(md->output_act));
        if (ctx->dry) {
            const bool whole1_dry = !fan && ctx->double_policy != 0 && e_loop == 1 && nt == 1 && md->TPW == 8 && md->OT <= 3 &&
                                    p.split == 0 && p.split_from < 0;
            const int g[12] = {1, nt, p.split, p.split_from, fan ? 1 : 0, (int)n_wg, smem, p.lb, 0, p.pl_units, front_wg, whole1_dry ? 1 : 0};
            std::memcpy(ctx->dry, g, sizeof(g));
            return L2A_OK;
        }
        // Whole tiles of ONE set per candidate (single model, per-block sets) at width 512 also run on the whole-tiles-only
        // instances (332 - 368 VGPRs instead of 442 - 496: -3 .. -4 % per round, profiles/r06_ab_whole1.jsonl); ensembles keep the
        // general instances (config 5's iteration, five sets in batches of three: +3.4 % on the whole-tiles-only one)
        const bool whole1 = !fan && ctx->double_policy != 0 && e_loop == 1 && nt == 1 && md->TPW == 8 && md->OT <= 3 &&
                            p.split == 0 && p.split_from < 0;
        const int geo = fan ? 1 : whole1 ? 2 : 0;
        int rc = l2a_launch_mfma(nt, md->TPW, md->OT, md->KG0, gact ? 1 : 0, geo, &p, grid.x, smem, stream);
        if (rc == -100) return fail(ctx, L2A_EINVAL, "no MFMA kernel instance for this (obs_dim, act_dim, hidden)");
        if (rc != 0) return fail(ctx, L2A_EHIP, std::string("MFMA kernel launch: ") + hipGetErrorString((hipError_t)rc));
        rc = L2A_OK;
        if (rc != L2A_OK) return rc;
    } else {
        p.tiles_per_env = ceil_div(p.n, L2A_VT);
        const int smem = (md->in_dim + 2 * md->hmax + 3 * md->obs_dim + md->act_dim + 1) * L2A_VT * 4;
        if (smem > ctx->lds_per_block)
            return fail(ctx, L2A_EINVAL, "LDS budget exceeded by the VALU kernel (" + std::to_string(smem) + " B)");
        if (ctx->dry) {
            const int g[12] = {0, 0, 0, -1, 0, p.m * p.tiles_per_env, smem, 1, 0, 0, 0, 0};
            std::memcpy(ctx->dry, g, sizeof(g));
            return L2A_OK;
        }
        int rc = allow_big_lds(ctx, l2a_rollout_valu_k, smem);
        if (rc != L2A_OK) return rc;
        const dim3 grid((unsigned)(p.m * p.tiles_per_env)), block(256);
        hipLaunchKernelGGL(l2a_rollout_valu_k, grid, block, smem, stream, p);
    }
    L2A_HIP(ctx, hipGetLastError());
    return L2A_OK;
}

// What the launch logic needs to know of a model's shape (no storage): shared by l2a_model_create and l2a_plan_geometry.
void model_shape(l2a_model* md, int obs_dim, int act_dim, int n_hidden, const int* hidden, int hidden_act, int output_act,
                 int n_sets, int mode) {
    md->obs_dim = obs_dim; md->act_dim = act_dim; md->in_dim = obs_dim + act_dim;
    md->n_hidden = n_hidden;
    md->hmax = 0;
    for (int i = 0; i < n_hidden; ++i) {
        md->hidden[i] = hidden[i];
        if (hidden[i] > md->hmax) md->hmax = hidden[i];
    }
    md->hidden_act = hidden_act; md->output_act = output_act;
    md->n_sets = n_sets; md->mode = mode;
    md->KG0 = ceil_div(md->in_dim, 16);
    md->OT = ceil_div(obs_dim, 16);
    md->mfma_ok = mfma_eligible(obs_dim, act_dim, n_hidden, hidden);
    md->H = md->mfma_ok ? hidden[0] : 0;
    md->TPW = md->mfma_ok ? hidden[0] / (16 * L2A_NW) : 0;
    md->micro_ok = md->mfma_ok && (md->H == 256 || md->H == 512);
    // the quarter sums of the O4 instance (l2a_mfma_inst.hip picks it for exactly these shapes)
    md->m_o4 = (md->micro_ok && md->OT == 2 && md->KG0 == 2 && n_hidden > 1 && obs_dim - 16 <= 4) ? 1 : 0;
}

void fill_model_params(const l2a_model* md, L2AKParams& p) {
    std::memset(&p, 0, sizeof(p));
    p.wblk = md->wblk;
    p.set_stride = md->set_stride;
    for (int i = 0; i < L2A_MAX_LAYERS; ++i) {
        p.raw_w[i] = md->raw_w[i];
        p.raw_b[i] = md->raw_b[i];
        p.hidden[i] = md->hidden[i];
    }
    p.pk_w0 = md->pk_w0; p.pk_wmid = md->pk_wmid; p.pk_wmid_stride = md->pk_wmid_stride;
    p.pk_wout = md->pk_wout; p.pk_bout = md->pk_bout; p.nm_off = md->nm_off;
    p.pk_m = md->pk_m; p.m_o4 = md->m_o4;
    p.m_nrec = md->micro_ok ? l2a_mlp_micro_nrec(md->H, md->KG0, md->n_hidden) : 0;
    p.n_hidden = md->n_hidden;
    p.obs_dim = md->obs_dim; p.act_dim = md->act_dim; p.in_dim = md->in_dim;
    p.hidden_act = md->hidden_act; p.output_act = md->output_act;
    p.mode = md->mode; p.n_sets = md->n_sets;
    p.KG0 = md->KG0; p.OT = md->OT; p.hmax = md->hmax;
    p.hid_floor = (md->hidden_act == L2A_ACT_RELU) ? 0.0f : -INFINITY;
    p.out_floor = (md->output_act == L2A_ACT_RELU) ? 0.0f : -INFINITY;
}

}  // namespace

extern "C" {

int l2a_init(int device, l2a_ctx** out) {
    if (!out) return fail(nullptr, L2A_EINVAL, "l2a_init: out is null");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, L2A_ENODEV, std::string("no HIP device visible: ") +
                                             (e != hipSuccess ? hipGetErrorString(e) : "device count is 0"));
    if (device < 0 || device >= count)
        return fail(nullptr, L2A_EINVAL, "device index " + std::to_string(device) + " out of range");
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return fail(nullptr, L2A_EHIP, std::string("hipGetDeviceProperties: ") + hipGetErrorString(e));
    std::string arch(prop.gcnArchName);
    if (arch.rfind("gfx950", 0) != 0)
        return fail(nullptr, L2A_ENODEV, "libl2a_hip.so is built for gfx950 only; device reports " + arch);
    e = hipSetDevice(device);
    if (e != hipSuccess) return fail(nullptr, L2A_EHIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    l2a_ctx* ctx = new l2a_ctx();
    ctx->device = device;
    ctx->num_cu = prop.multiProcessorCount;
    ctx->lds_per_block = (int)prop.sharedMemPerBlock;
    // gfx950 lets one workgroup take the whole 160 KiB of a CU (MI355X_MICROARCH.md, LDS).
    if (ctx->lds_per_block < 160 * 1024) ctx->lds_per_block = 160 * 1024;
    ctx->clock_khz = prop.clockRate;
    ctx->arch = arch;
    ctx->name = prop.name;
    e = hipHostMalloc(reinterpret_cast<void**>(&ctx->status_host), sizeof(unsigned int), hipHostMallocMapped);
    if (e == hipSuccess) {
        *ctx->status_host = 0;
        e = hipHostGetDevicePointer(reinterpret_cast<void**>(&ctx->status_dev), ctx->status_host, 0);
    }
    if (e != hipSuccess) {
        std::string msg = std::string("allocating the launch status word: ") + hipGetErrorString(e);
        delete ctx;
        return fail(nullptr, L2A_EHIP, msg);
    }
    const char* sp = std::getenv("L2A_SPLIT");
    if (sp && sp[0] >= '0' && sp[0] <= '2') ctx->split_policy = sp[0] - '0';
    const char* xa = std::getenv("L2A_XCD_ALIGN");
    if (xa && (xa[0] == '0' || xa[0] == '1')) ctx->xcd_align = xa[0] - '0';
    const char* fn = std::getenv("L2A_FAN");
    if (fn && (fn[0] == '0' || fn[0] == '1')) ctx->fan_policy = fn[0] - '0';
    const char* db = std::getenv("L2A_DOUBLE");
    if (db && (db[0] == '0' || db[0] == '1')) ctx->double_policy = db[0] - '0';
    const char* bs = std::getenv("L2A_BATCH");
    if (bs && bs[0] >= '0' && bs[0] <= '4') ctx->batch_sets = bs[0] - '0';
    const char* mc = std::getenv("L2A_MICRO");
    if (mc && mc[0] >= '0' && mc[0] <= '2') ctx->micro_policy = mc[0] - '0';
    *out = ctx;
    return L2A_OK;
}

void l2a_destroy(l2a_ctx* ctx) {
    if (!ctx) return;
    (void)l2a_comm_destroy(ctx);
    if (ctx->status_host) (void)hipHostFree(ctx->status_host);
    if (ctx->mail_host) (void)hipHostFree(ctx->mail_host);
    if (ctx->done_ctr) (void)hipFree(ctx->done_ctr);
    if (ctx->key_ring) (void)hipFree(ctx->key_ring);
    delete ctx;
}

int l2a_plan_geometry(int obs_dim, int act_dim, int n_hidden, const int* hidden, int n_sets, int mode, int m, int n, int h,
                      const int* policy, int* out) {
    if (!hidden || !out || obs_dim < 1 || act_dim < 1 || n_hidden < 1 || n_hidden > L2A_MAX_LAYERS - 1 || n_sets < 1 || m < 1 || n < 1 || h < 1)
        return L2A_EINVAL;
    l2a_ctx ctx;                                    // never touches a device: the launcher stops before its first HIP call
    ctx.num_cu = 256;
    ctx.lds_per_block = 160 * 1024;
    if (policy) {
        if (policy[0] >= 0) ctx.split_policy = policy[0];
        if (policy[1] >= 0) ctx.fan_policy = policy[1];
        if (policy[2] >= 0) ctx.micro_policy = policy[2];
        if (policy[3] > 0) ctx.num_cu = policy[3];
        if (policy[4] >= 0) ctx.double_policy = policy[4];
    }
    l2a_model md;
    md.ctx = &ctx;
    model_shape(&md, obs_dim, act_dim, n_hidden, hidden, L2A_ACT_RELU, L2A_ACT_IDENTITY, n_sets, mode);
    L2AKParams p;
    fill_model_params(&md, p);
    p.m = m; p.n = n; p.h = h; p.discount = 1.0; p.disc0 = 1.0;
    unsigned long long dummy = 0;
    p.best_key = &dummy;                            // "a plan": returns or keys are wanted (never dereferenced here)
    int g[12] = {0};
    ctx.dry = g;
    const int rc = launch_rollout(&md, p, nullptr);
    ctx.dry = nullptr;
    if (rc != L2A_OK) return rc;
    for (int i = 0; i < 12; ++i) out[i] = g[i];
    return L2A_OK;
}

int l2a_set_split(l2a_ctx* ctx, int policy) {
    if (!ctx) return L2A_EINVAL;
    if (policy < 0 || policy > 2) return fail(ctx, L2A_EINVAL, "split policy must be 0, 1 or 2");
    ctx->split_policy = policy;
    return L2A_OK;
}

int l2a_set_fan(l2a_ctx* ctx, int on) {
    if (!ctx) return L2A_EINVAL;
    if (on != 0 && on != 1) return fail(ctx, L2A_EINVAL, "member fan must be 0 or 1");
    ctx->fan_policy = on;
    return L2A_OK;
}

int l2a_set_double_rounds(l2a_ctx* ctx, int on) {
    if (!ctx) return L2A_EINVAL;
    if (on != 0 && on != 1) return fail(ctx, L2A_EINVAL, "double rounds must be 0 or 1");
    ctx->double_policy = on;
    return L2A_OK;
}

int l2a_set_batch(l2a_ctx* ctx, int sets) {
    if (!ctx) return L2A_EINVAL;
    if (sets < 0 || sets > 4) return fail(ctx, L2A_EINVAL, "sets per batch must be 0 (automatic) .. 4");
    ctx->batch_sets = sets;
    return L2A_OK;
}

int l2a_set_micro(l2a_ctx* ctx, int policy) {
    if (!ctx) return L2A_EINVAL;
    if (policy < 0 || policy > 2) return fail(ctx, L2A_EINVAL, "micro-tile policy must be 0, 1 or 2");
    ctx->micro_policy = policy;
    return L2A_OK;
}

int l2a_set_xcd_align(l2a_ctx* ctx, int on) {
    if (!ctx) return L2A_EINVAL;
    if (on != 0 && on != 1) return fail(ctx, L2A_EINVAL, "xcd alignment must be 0 or 1");
    ctx->xcd_align = on;
    return L2A_OK;
}

int l2a_inject_status(l2a_ctx* ctx, int bits) {
    if (!ctx) return L2A_EINVAL;
    *ctx->status_host |= (unsigned int)bits;
    return L2A_OK;
}

int l2a_set_spin_limit(l2a_ctx* ctx, unsigned int polls) {
    if (!ctx) return L2A_EINVAL;
    ctx->spin_limit = polls ? polls : (1u << 18);
    return L2A_OK;
}

int l2a_set_debug_buffer(l2a_ctx* ctx, void* device_ptr) {
    if (!ctx) return L2A_EINVAL;
    ctx->dbg = static_cast<unsigned long long*>(device_ptr);
    return L2A_OK;
}

int l2a_launch_status(l2a_ctx* ctx, int* status_out) {
    if (!ctx || !status_out) return L2A_EINVAL;
    *status_out = (int)(*ctx->status_host);
    *ctx->status_host = 0;
    return L2A_OK;
}

const char* l2a_last_error(const l2a_ctx* ctx) { return ctx ? ctx->err.c_str() : g_init_error.c_str(); }

int l2a_device_info(const l2a_ctx* ctx, char* buf, int cap) {
    if (!ctx || !buf || cap <= 0) return L2A_EINVAL;
    std::snprintf(buf, (size_t)cap,
                  "{\"device\": %d, \"name\": \"%s\", \"arch\": \"%s\", \"compute_units\": %d, "
                  "\"clock_khz\": %d, \"lds_per_block\": %d}",
                  ctx->device, ctx->name.c_str(), ctx->arch.c_str(), ctx->num_cu, ctx->clock_khz,
                  ctx->lds_per_block);
    return L2A_OK;
}

int l2a_set_kernel(l2a_ctx* ctx, int kind) {
    if (!ctx) return L2A_EINVAL;
    if (kind != L2A_KERNEL_AUTO && kind != L2A_KERNEL_MFMA && kind != L2A_KERNEL_VALU)
        return fail(ctx, L2A_EINVAL, "unknown kernel kind");
    ctx->kernel_kind = kind;
    return L2A_OK;
}

int l2a_mfma_eligible(int obs_dim, int act_dim, int n_hidden, const int* hidden) {
    return (hidden && mfma_eligible(obs_dim, act_dim, n_hidden, hidden)) ? 1 : 0;
}

long long l2a_packed_layer_floats(int k_in, int n_out) { return packed_floats(k_in, n_out); }

int l2a_pack_layer_host(const float* w, int k_in, int n_out, float* out) {
    if (!w || !out || k_in < 1 || n_out < 1) return L2A_EINVAL;
    const int KG = ceil_div(k_in, 16);
    const long long total = packed_floats(k_in, n_out);
    for (long long idx = 0; idx < total; ++idx) {
        int k, u;
        l2a_pack_decode(idx, KG, &k, &u);
        out[idx] = (k < k_in && u < n_out) ? w[(long long)k * n_out + u] : 0.0f;
    }
    return L2A_OK;
}

long long l2a_micro_layout_floats(int obs_dim, int act_dim, int n_hidden, int hidden) {
    const int hid[L2A_MAX_LAYERS] = {hidden, hidden, hidden, hidden, hidden, hidden, hidden, hidden, hidden};
    if (n_hidden < 1 || n_hidden > L2A_MAX_LAYERS - 1 || !mfma_eligible(obs_dim, act_dim, n_hidden, hid)) return 0;
    if (hidden != 256 && hidden != 512) return 0;
    return l2a_mlp_micro_floats(hidden, ceil_div(obs_dim + act_dim, 16), n_hidden);
}

int l2a_micro_pack_layer_host(const float* w, int obs_dim, int act_dim, int n_hidden, int hidden, int layer, float* out) {
    if (!w || !out || layer < 0 || layer > n_hidden || l2a_micro_layout_floats(obs_dim, act_dim, n_hidden, hidden) == 0)
        return L2A_EINVAL;
    const int KG0 = ceil_div(obs_dim + act_dim, 16), OT = ceil_div(obs_dim, 16);
    const int o4 = (OT == 2 && KG0 == 2 && n_hidden > 1 && obs_dim - 16 <= 4) ? 1 : 0;
    const int k_in = layer == 0 ? obs_dim + act_dim : hidden, n_out = layer == n_hidden ? obs_dim : hidden;
    for (int k = 0; k < k_in; ++k)
        for (int u = 0; u < n_out; ++u)
            out[l2a_mlp_micro_index(hidden, KG0, n_hidden, o4, layer, k, u)] = w[(long long)k * n_out + u];
    return L2A_OK;
}

unsigned long long l2a_key_encode(float ret, int index) { return l2a_key_pack(ret, index); }

void l2a_key_decode(unsigned long long key, float* ret, int* index) {
    const unsigned int ord = (unsigned int)(key >> 31);
    const unsigned int idx = 0x7fffffffu - (unsigned int)(key & 0x7fffffffull);
    const unsigned int u = (ord & 0x80000000u) ? (ord & 0x7fffffffu) : ~ord;
    union { float f; unsigned int u; } cv;
    cv.u = u;
    if (ret) *ret = cv.f;
    if (index) *index = (int)idx;
}

int l2a_model_create(l2a_ctx* ctx, int obs_dim, int act_dim, int n_hidden, const int* hidden,
                     int hidden_act, int output_act, int n_sets, int mode, l2a_model** out) {
    if (!ctx) return L2A_EINVAL;
    if (!out || !hidden) return fail(ctx, L2A_EINVAL, "l2a_model_create: null argument");
    *out = nullptr;
    if (obs_dim < 1 || act_dim < 1) return fail(ctx, L2A_EINVAL, "obs_dim and act_dim must be >= 1");
    if (n_hidden < 1 || n_hidden > L2A_MAX_LAYERS - 1)
        return fail(ctx, L2A_EINVAL, "n_hidden must be in [1, 8]");
    for (int i = 0; i < n_hidden; ++i)
        if (hidden[i] < 1 || hidden[i] > 1024) return fail(ctx, L2A_EINVAL, "hidden sizes must be in [1, 1024]");
    if (hidden_act < 0 || hidden_act > L2A_ACT_SWISH || output_act < 0 || output_act > L2A_ACT_SWISH)
        return fail(ctx, L2A_EINVAL, "unsupported nonlinearity");
    if (mode != L2A_MODE_SINGLE && mode != L2A_MODE_PER_BLOCK && mode != L2A_MODE_MEAN)
        return fail(ctx, L2A_EINVAL, "unknown mode");
    if (n_sets < 1 || (mode == L2A_MODE_SINGLE && n_sets != 1))
        return fail(ctx, L2A_EINVAL, "n_sets must be >= 1 (and exactly 1 in single mode)");

    l2a_model* md = new l2a_model();
    md->ctx = ctx;
    model_shape(md, obs_dim, act_dim, n_hidden, hidden, hidden_act, output_act, n_sets, mode);

    // ---- lay out one weight-set block (offsets in floats, every region 64-B aligned) ------
    long long off = 0;
    auto take = [&off](long long n) { long long o = off; off += (n + 15) / 16 * 16; return o; };
    int k_in = md->in_dim;
    for (int l = 0; l <= n_hidden; ++l) {
        const int n_out = (l < n_hidden) ? hidden[l] : obs_dim;
        md->raw_w[l] = take((long long)k_in * n_out);
        md->raw_b[l] = take(n_out);
        k_in = n_out;
    }
    if (md->mfma_ok) {
        md->pk_w0 = take(packed_floats(md->in_dim, md->H));
        md->pk_wmid_stride = packed_floats(md->H, md->H);
        md->pk_wmid = take(md->pk_wmid_stride * (n_hidden - 1));
        md->pk_wout = take(packed_floats(md->H, obs_dim));
        if (md->micro_ok) md->pk_m = take(l2a_mlp_micro_floats(md->H, md->KG0, n_hidden));
    }
    md->pk_bout = take(16 * md->OT);
    md->nm_off = take(32 * md->KG0 + 32 * md->OT);
    md->set_stride = off;

    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&md->wblk), (size_t)off * n_sets * sizeof(float));
    if (e == hipSuccess) e = hipMemset(md->wblk, 0, (size_t)off * n_sets * sizeof(float));
    if (e != hipSuccess) {
        std::string msg = std::string("allocating model storage: ") + hipGetErrorString(e);
        delete md;
        return fail(ctx, L2A_EHIP, msg);
    }
    md->weights_set.assign(n_sets, 0);
    md->norm_set.assign(n_sets, 0);
    md->norm_stage.resize(n_sets);
    *out = md;
    return L2A_OK;
}

void l2a_model_destroy(l2a_model* md) {
    if (!md) return;
    l2a_device_guard guard(md->ctx->device);
    if (md->wblk) {
        (void)hipDeviceSynchronize();
        (void)hipFree(md->wblk);
    }
    if (md->xbuf) (void)hipFree(md->xbuf);
    if (md->adapt_scratch) (void)hipFree(md->adapt_scratch);
    for (auto& sl : md->aslot) {
        if (sl.done) (void)hipEventDestroy(sl.done);
        if (sl.stage_host) (void)hipHostFree(sl.stage_host);
    }
    delete md;
}

int l2a_model_set_weights_strided(l2a_model* md, int first_set, int count, const void* const* device_ptrs,
                                  const long long* set_strides, void* stream_v) {
    if (!md) return L2A_EINVAL;
    l2a_ctx* ctx = md->ctx;
    if (count < 1 || first_set < 0 || first_set + count > md->n_sets)
        return fail(ctx, L2A_EINVAL, "weight set range out of bounds");
    if (!device_ptrs) return fail(ctx, L2A_EINVAL, "device_ptrs is null");
    if (count > 1 && !set_strides) return fail(ctx, L2A_EINVAL, "set_strides is null");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    l2a_device_guard guard(ctx->device);
    float* blk = md->wblk + (long long)first_set * md->set_stride;
    const size_t dpitch = sizeof(float) * (size_t)md->set_stride;
    // One strided device-to-device copy of `n` floats per set (hipMemcpy2DAsync: `count` rows).
    auto copy_sets = [&](float* dst, const float* src, long long src_stride, size_t n) -> hipError_t {
        if (count == 1) return hipMemcpyAsync(dst, src, sizeof(float) * n, hipMemcpyDeviceToDevice, stream);
        return hipMemcpy2DAsync(dst, dpitch, src, sizeof(float) * (size_t)src_stride, sizeof(float) * n,
                                (size_t)count, hipMemcpyDeviceToDevice, stream);
    };
    int k_in = md->in_dim;
    for (int l = 0; l <= md->n_hidden; ++l) {
        const int n_out = (l < md->n_hidden) ? md->hidden[l] : md->obs_dim;
        const float* w = static_cast<const float*>(device_ptrs[2 * l]);
        const float* b = static_cast<const float*>(device_ptrs[2 * l + 1]);
        if (!w || !b) return fail(ctx, L2A_EINVAL, "null parameter pointer for layer " + std::to_string(l));
        const long long ws = (count > 1) ? set_strides[2 * l] : 0, bs = (count > 1) ? set_strides[2 * l + 1] : 0;
        if (count > 1 && (ws < (long long)k_in * n_out || bs < n_out))
            return fail(ctx, L2A_EINVAL, "set stride smaller than the parameter of layer " + std::to_string(l));
        L2A_HIP(ctx, copy_sets(blk + md->raw_w[l], w, ws, (size_t)k_in * n_out));
        L2A_HIP(ctx, copy_sets(blk + md->raw_b[l], b, bs, (size_t)n_out));
        if (md->mfma_ok) {
            float* dst;
            if (l == 0) dst = blk + md->pk_w0;
            else if (l < md->n_hidden) dst = blk + md->pk_wmid + (long long)(l - 1) * md->pk_wmid_stride;
            else dst = blk + md->pk_wout;
            const long long total = packed_floats(k_in, n_out);
            const int KG = ceil_div(k_in, 16);
            const dim3 grid((unsigned)((total + 255) / 256), (unsigned)count);
            hipLaunchKernelGGL(l2a_pack_layer_k, grid, dim3(256), 0, stream, w, ws, k_in, n_out, KG, total, dst,
                               md->set_stride);
            L2A_HIP(ctx, hipGetLastError());
            if (md->micro_ok) {
                const dim3 mgrid((unsigned)(((long long)k_in * n_out + 255) / 256), (unsigned)count);
                hipLaunchKernelGGL(l2a_mlp_micro_pack_k, mgrid, dim3(256), 0, stream, w, ws, k_in, n_out, l, md->H, md->KG0,
                                   md->n_hidden, md->m_o4, blk + md->pk_m, md->set_stride);
                L2A_HIP(ctx, hipGetLastError());
            }
        }
        if (l == md->n_hidden)      // padded copy of the output bias (the tail beyond obs_dim stays zero)
            L2A_HIP(ctx, copy_sets(blk + md->pk_bout, b, bs, (size_t)n_out));
        k_in = n_out;
    }
    for (int e = first_set; e < first_set + count; ++e) md->weights_set[e] = 1;
    return L2A_OK;
}

int l2a_model_set_weights(l2a_model* md, int e, const void* const* device_ptrs, void* stream_v) {
    return l2a_model_set_weights_strided(md, e, 1, device_ptrs, nullptr, stream_v);
}

namespace {

// Argument checks + parameter block of one adaptation step; grows the scratch buffer (may synchronise `stream`).
int adapt_prepare(l2a_model* md, const void* const* base_ptrs, const float* x, const float* y, int m, int rows,
                  hipStream_t stream, L2AAdaptParams& ap) {
    l2a_ctx* ctx = md->ctx;
    if (!base_ptrs || !x || !y) return fail(ctx, L2A_EINVAL, "l2a_model_adapt_sgd: null pointer");
    if (md->mode != L2A_MODE_PER_BLOCK) return fail(ctx, L2A_EINVAL, "l2a_model_adapt_sgd needs a per-block model");
    if (m < 1 || m > md->n_sets) return fail(ctx, L2A_EINVAL, "l2a_model_adapt_sgd: m must be in [1, n_sets]");
    if (rows < 1 || rows > L2A_AR)
        return fail(ctx, L2A_EINVAL, "l2a_model_adapt_sgd: rows must be in [1, " + std::to_string(L2A_AR) + "]");
    if (md->output_act != L2A_ACT_IDENTITY || md->hidden_act == L2A_ACT_SWISH)
        return fail(ctx, L2A_EINVAL, "l2a_model_adapt_sgd: needs an identity output layer and a relu / tanh / "
                                     "sigmoid / identity hidden nonlinearity");
    std::memset(&ap, 0, sizeof(ap));
    const int L = md->n_hidden + 1;
    ap.n_layers = L;
    ap.dims[0] = md->in_dim;
    for (int l = 0; l < md->n_hidden; ++l) ap.dims[l + 1] = md->hidden[l];
    ap.dims[L] = md->obs_dim;
    long long off = 0;
    int hmax = 0;
    for (int l = 0; l <= L; ++l) hmax = ap.dims[l] > hmax ? ap.dims[l] : hmax;
    for (int l = 0; l < L; ++l) { ap.a_off[l] = off; off += (long long)ap.dims[l] * L2A_AR; }
    for (int l = 1; l <= L; ++l) { ap.z_off[l] = off; off += (long long)ap.dims[l] * L2A_AR; }
    ap.y_off = off; off += (long long)ap.dims[L] * L2A_AR;
    ap.scratch_stride = off;
    ap.hmax = hmax;
    for (int l = 0; l < L; ++l) {
        ap.w[l] = static_cast<const float*>(base_ptrs[2 * l]);
        ap.b[l] = static_cast<const float*>(base_ptrs[2 * l + 1]);
        if (!ap.w[l] || !ap.b[l]) return fail(ctx, L2A_EINVAL, "null base parameter pointer for layer " + std::to_string(l));
    }
    ap.hidden_act = md->hidden_act;
    ap.rows = rows;
    ap.x = x; ap.y = y;
    const long long need = off * md->n_sets;
    if (need > md->adapt_scratch_floats) {
        if (md->adapt_scratch) { L2A_HIP(ctx, hipStreamSynchronize(stream)); L2A_HIP(ctx, hipFree(md->adapt_scratch)); md->adapt_scratch = nullptr; }
        L2A_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&md->adapt_scratch), (size_t)need * sizeof(float)));
        md->adapt_scratch_floats = need;
    }
    ap.scratch = md->adapt_scratch;
    ap.dbg = ctx->dbg;
    return L2A_OK;
}

// One adaptation step on `stream`: one launch per phase.
int adapt_enqueue(l2a_model* md, const L2AAdaptParams& ap, int m, float lr, hipStream_t stream) {
    l2a_ctx* ctx = md->ctx;
    const int L = ap.n_layers;
    L2AAdaptDst d;
    std::memset(&d, 0, sizeof(d));
    d.blk = md->wblk;
    d.set_stride = md->set_stride;
    d.has_pk = md->mfma_ok ? 1 : 0;
    d.has_mk = md->micro_ok ? 1 : 0;
    d.mk = md->pk_m; d.mk_H = md->H; d.mk_KG0 = md->KG0; d.mk_o4 = md->m_o4;
    d.pk_bout = md->pk_bout;
    d.lr = lr;
    for (int l = 0; l < L; ++l) {
        d.raw_w[l] = md->raw_w[l];
        d.raw_b[l] = md->raw_b[l];
        if (md->mfma_ok) {
            if (l == 0) d.pk[l] = md->pk_w0;
            else if (l < md->n_hidden) d.pk[l] = md->pk_wmid + (long long)(l - 1) * md->pk_wmid_stride;
            else d.pk[l] = md->pk_wout;
        }
    }
    const dim3 block(64 * L2A_AW);
    auto slices = [&](int dim) { return (unsigned)((dim + 63) / 64); };
    hipLaunchKernelGGL(l2a_adapt_fwd0_k, dim3(slices(ap.dims[1]), (unsigned)m), block, 0, stream, ap);
    for (int l = 1; l < L; ++l)
        hipLaunchKernelGGL(l2a_adapt_fwd_k, dim3(slices(ap.dims[l + 1]), (unsigned)m), block, 0, stream, ap, l);
    // backward through layer l beside the update of layer l; the last of them (l = 1) also updates layer 0
    for (int l = L - 1; l >= 1; --l) {
        const int ub = ((ap.dims[l + 1] + 255) / 256) * ((ap.dims[l] + L2A_UK - 1) / L2A_UK);
        hipLaunchKernelGGL(l2a_adapt_bwdu_k, dim3(slices(ap.dims[l]) + (unsigned)((ub + 1) / 2), (unsigned)m), block, 0, stream, ap, d, l);
    }
    L2A_HIP(ctx, hipGetLastError());
    return L2A_OK;
}

}  // namespace

int l2a_model_adapt_sgd(l2a_model* md, const void* const* base_ptrs, const float* x, const float* y, int m,
                        int rows, float lr, void* stream_v) {
    if (!md) return L2A_EINVAL;
    l2a_ctx* ctx = md->ctx;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    l2a_device_guard guard(ctx->device);
    L2AAdaptParams ap;
    const int rc = adapt_prepare(md, base_ptrs, x, y, m, rows, stream, ap);
    if (rc != L2A_OK) return rc;
    const int rc2 = adapt_enqueue(md, ap, m, lr, stream);
    if (rc2 != L2A_OK) return rc2;
    for (int e = 0; e < m; ++e) md->weights_set[e] = 1;
    return L2A_OK;
}

extern "C++" {
namespace {

// Shared by the two host-array entry points: stage `bytes` (gathered by `fill` into the slot's host-mapped buffer),
// run the step with the parameters `bind` derives from the staging's device alias.
template <class Fill, class Bind>
int adapt_staged(l2a_model* md, const void* const* base_ptrs, int m, int rows, float lr, hipStream_t stream,
                 size_t bytes, Fill fill, Bind bind, const char* who) {
    l2a_ctx* ctx = md->ctx;
    if (m < 1 || m > md->n_sets || rows < 1 || rows > L2A_AR)
        return fail(ctx, L2A_EINVAL, std::string(who) + ": m must be in [1, n_sets], rows in [1, 16]");
    auto& sl = md->aslot[md->aslot_next];
    md->aslot_next ^= 1;
    // the launch that last read this slot's staging must be through before the host overwrites it
    if (sl.pending) { L2A_HIP(ctx, hipEventSynchronize(sl.done)); sl.pending = false; }
    if ((long long)bytes > sl.stage_floats * (long long)sizeof(float)) {
        if (sl.stage_host) { L2A_HIP(ctx, hipHostFree(sl.stage_host)); sl.stage_host = nullptr; }
        // room for either layout: fp32 [x | y] or float64 [obs | act | next | six normalisation vectors]
        const long long cap = 2LL * md->n_sets * L2A_AR * (md->in_dim + md->obs_dim) + 4LL * (md->in_dim + md->obs_dim) + 16;
        L2A_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&sl.stage_host), (size_t)cap * sizeof(float), hipHostMallocMapped));
        L2A_HIP(ctx, hipHostGetDevicePointer(reinterpret_cast<void**>(&sl.stage_dev), sl.stage_host, 0));
        sl.stage_floats = cap;
    }
    if (!sl.done) L2A_HIP(ctx, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    fill(sl.stage_host);
    L2AAdaptParams ap;
    const float* dummy = sl.stage_dev;
    const int rc = adapt_prepare(md, base_ptrs, dummy, dummy, m, rows, stream, ap);
    if (rc != L2A_OK) return rc;
    bind(ap, sl.stage_dev);
    const int rc2 = adapt_enqueue(md, ap, m, lr, stream);
    if (rc2 != L2A_OK) return rc2;
    L2A_HIP(ctx, hipEventRecord(sl.done, stream));
    sl.pending = true;
    for (int e = 0; e < m; ++e) md->weights_set[e] = 1;
    return L2A_OK;
}

}  // namespace
}  // extern "C++"

int l2a_model_adapt_sgd_host(l2a_model* md, const void* const* base_ptrs, const float* x_host, const float* y_host,
                             int m, int rows, float lr, void* stream_v) {
    if (!md) return L2A_EINVAL;
    l2a_ctx* ctx = md->ctx;
    l2a_device_guard guard(ctx->device);
    if (!x_host || !y_host) return fail(ctx, L2A_EINVAL, "l2a_model_adapt_sgd_host: null pointer");
    const long long mm = m > 0 ? m : 0, rr = rows > 0 ? rows : 0;
    const long long xf = mm * rr * md->in_dim, yf = mm * rr * md->obs_dim;
    return adapt_staged(md, base_ptrs, m, rows, lr, reinterpret_cast<hipStream_t>(stream_v), (size_t)(xf + yf) * sizeof(float),
        [&](float* host) {
            std::memcpy(host, x_host, (size_t)xf * sizeof(float));
            std::memcpy(host + xf, y_host, (size_t)yf * sizeof(float));
        },
        [&](L2AAdaptParams& ap, const float* dev) { ap.x = dev; ap.y = dev + xf; },
        "l2a_model_adapt_sgd_host");
}

int l2a_model_adapt_sgd_raw(l2a_model* md, const void* const* base_ptrs, const double* obs_host, const double* act_host,
                            const double* next_obs_host, const double* mean_obs, const double* std_obs,
                            const double* mean_act, const double* std_act, const double* mean_delta,
                            const double* std_delta, int m, int rows, float lr, void* stream_v) {
    if (!md) return L2A_EINVAL;
    l2a_ctx* ctx = md->ctx;
    l2a_device_guard guard(ctx->device);
    if (!obs_host || !act_host || !next_obs_host || !mean_obs || !std_obs || !mean_act || !std_act || !mean_delta || !std_delta)
        return fail(ctx, L2A_EINVAL, "l2a_model_adapt_sgd_raw: null pointer");
    if (md->in_dim > L2A_XS_MAX)
        return fail(ctx, L2A_EINVAL, "l2a_model_adapt_sgd_raw: input layers of at most " + std::to_string(L2A_XS_MAX) +
                                     " features (normalise on the host and use l2a_model_adapt_sgd_host)");
    const int od = md->obs_dim, ad = md->act_dim;
    const long long mm = m > 0 ? m : 0, rr = rows > 0 ? rows : 0;
    const long long no = mm * rr * od, na = mm * rr * ad;
    const long long doubles = 2 * no + na + 4 * od + 2 * ad;
    return adapt_staged(md, base_ptrs, m, rows, lr, reinterpret_cast<hipStream_t>(stream_v), (size_t)doubles * sizeof(double),
        [&](float* host_f) {
            double* host = reinterpret_cast<double*>(host_f);
            std::memcpy(host, obs_host, (size_t)no * sizeof(double));
            std::memcpy(host + no, act_host, (size_t)na * sizeof(double));
            std::memcpy(host + no + na, next_obs_host, (size_t)no * sizeof(double));
            double* nv = host + 2 * no + na;
            std::memcpy(nv, mean_obs, sizeof(double) * od);            std::memcpy(nv + od, std_obs, sizeof(double) * od);
            std::memcpy(nv + 2 * od, mean_act, sizeof(double) * ad);   std::memcpy(nv + 2 * od + ad, std_act, sizeof(double) * ad);
            std::memcpy(nv + 2 * od + 2 * ad, mean_delta, sizeof(double) * od);
            std::memcpy(nv + 3 * od + 2 * ad, std_delta, sizeof(double) * od);
        },
        [&](L2AAdaptParams& ap, const float* dev_f) {
            const double* dev = reinterpret_cast<const double*>(dev_f);
            ap.x = nullptr; ap.y = nullptr;
            ap.raw_obs = dev; ap.raw_act = dev + no; ap.raw_next = dev + no + na; ap.raw_norm = dev + 2 * no + na;
            ap.obs_dim = od; ap.act_dim = ad;
        },
        "l2a_model_adapt_sgd_raw");
}

int l2a_model_get_weights(l2a_model* md, int e, void* const* device_ptrs_out, void* stream_v) {
    if (!md) return L2A_EINVAL;
    l2a_ctx* ctx = md->ctx;
    if (e < 0 || e >= md->n_sets) return fail(ctx, L2A_EINVAL, "weight set index out of range");
    if (!device_ptrs_out) return fail(ctx, L2A_EINVAL, "device_ptrs_out is null");
    if (!md->weights_set[e]) return fail(ctx, L2A_ESTATE, "weight set " + std::to_string(e) + " was never set");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    l2a_device_guard guard(ctx->device);
    const float* blk = md->wblk + (long long)e * md->set_stride;
    int k_in = md->in_dim;
    for (int l = 0; l <= md->n_hidden; ++l) {
        const int n_out = (l < md->n_hidden) ? md->hidden[l] : md->obs_dim;
        if (!device_ptrs_out[2 * l] || !device_ptrs_out[2 * l + 1])
            return fail(ctx, L2A_EINVAL, "null output pointer for layer " + std::to_string(l));
        L2A_HIP(ctx, hipMemcpyAsync(device_ptrs_out[2 * l], blk + md->raw_w[l], sizeof(float) * (size_t)k_in * n_out,
                                    hipMemcpyDeviceToDevice, stream));
        L2A_HIP(ctx, hipMemcpyAsync(device_ptrs_out[2 * l + 1], blk + md->raw_b[l], sizeof(float) * (size_t)n_out,
                                    hipMemcpyDeviceToDevice, stream));
        k_in = n_out;
    }
    return L2A_OK;
}

int l2a_model_set_norm(l2a_model* md, int e, const double* mean_obs, const double* std_obs,
                       const double* mean_act, const double* std_act, const double* mean_delta,
                       const double* std_delta, void* stream_v) {
    if (!md) return L2A_EINVAL;
    l2a_ctx* ctx = md->ctx;
    if (e < 0 || e >= md->n_sets) return fail(ctx, L2A_EINVAL, "weight set index out of range");
    const int n_null = !mean_obs + !std_obs + !mean_act + !std_act + !mean_delta + !std_delta;
    if (n_null != 0 && n_null != 6)
        return fail(ctx, L2A_EINVAL, "pass all six normalisation vectors, or none for identity");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    l2a_device_guard guard(ctx->device);
    const int KG0 = md->KG0, OT = md->OT;
    std::vector<float>& st = md->norm_stage[e];
    // a previous async copy from this staging buffer must have drained before we overwrite it
    if (!st.empty()) L2A_HIP(ctx, hipStreamSynchronize(stream));
    st.assign((size_t)(32 * KG0 + 32 * OT), 0.0f);
    float* in_mu = st.data();
    float* in_iv = in_mu + 16 * KG0;
    float* out_mu = in_iv + 16 * KG0;
    float* out_sd = out_mu + 16 * OT;
    const double eps = 1e-10;   // mlp_dynamics.py:265-270
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
    float* dst = md->wblk + (long long)e * md->set_stride + md->nm_off;
    L2A_HIP(ctx, hipMemcpyAsync(dst, st.data(), st.size() * sizeof(float), hipMemcpyHostToDevice, stream));
    md->norm_set[e] = 1;
    return L2A_OK;
}

int l2a_plan_rs(l2a_model* md, const float* obs0, const float* actions, int m, int n, int h,
                double discount, const l2a_reward* reward, int cand_offset, float* returns_out,
                unsigned long long* best_key, void* stream_v) {
    if (!md) return L2A_EINVAL;
    l2a_ctx* ctx = md->ctx;
    if (!obs0 || !actions || !reward) return fail(ctx, L2A_EINVAL, "l2a_plan_rs: null obs0/actions/reward");
    if (!best_key && !returns_out) return fail(ctx, L2A_EINVAL, "l2a_plan_rs: nothing to write (best_key and returns_out are null)");
    if (m < 1 || n < 1 || h < 1) return fail(ctx, L2A_EINVAL, "l2a_plan_rs: m, n and h must be >= 1");
    if ((long long)m * n > 0x3fffffffLL || cand_offset < 0 || (long long)cand_offset + n > 0x7fffffffLL)
        return fail(ctx, L2A_EINVAL, "l2a_plan_rs: too many candidates");
    if (reward->w_vel != 0.0f && (reward->vel_index < 0 || reward->vel_index >= md->obs_dim))
        return fail(ctx, L2A_EINVAL, "reward.vel_index out of range");
    if (reward->dist_coef != 0.0f && (reward->dist_index < 0 || reward->dist_index >= md->obs_dim))
        return fail(ctx, L2A_EINVAL, "reward.dist_index out of range");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    l2a_device_guard guard(ctx->device);
    if (best_key) L2A_HIP(ctx, hipMemsetAsync(best_key, 0, sizeof(unsigned long long) * (size_t)m, stream));
    L2AKParams p;
    fill_model_params(md, p);
    p.obs0 = obs0; p.actions = actions; p.returns_out = returns_out; p.best_key = best_key;
    p.state_out = nullptr; p.obs_per_row = 0;
    p.ret_in = nullptr; p.disc0 = 1.0;
    p.m = m; p.n = n; p.h = h; p.cand_offset = cand_offset; p.discount = discount; p.rw = *reward;
    return launch_rollout(md, p, stream_v);
}

// Allocates the context's result mailbox on first use.
static int ensure_mail(l2a_ctx* ctx) {
    if (ctx->mail_host) return L2A_OK;
    l2a_mail* host = nullptr;
    L2A_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&host), sizeof(l2a_mail), hipHostMallocMapped));
    std::memset(host, 0, sizeof(l2a_mail));
    l2a_mail* dev = nullptr;
    hipError_t e = hipHostGetDevicePointer(reinterpret_cast<void**>(&dev), host, 0);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&ctx->done_ctr), sizeof(unsigned int));
    if (e == hipSuccess) e = hipMemset(ctx->done_ctr, 0, sizeof(unsigned int));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&ctx->key_ring), 2 * L2A_MAIL_KEYS * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(ctx->key_ring, 0, 2 * L2A_MAIL_KEYS * sizeof(unsigned long long));
    if (e != hipSuccess) {
        (void)hipHostFree(host);
        return fail(ctx, L2A_EHIP, std::string("allocating the result mailbox: ") + hipGetErrorString(e));
    }
    ctx->mail_host = host;
    ctx->mail_dev = dev;
    ctx->ring_dirty[0] = ctx->ring_dirty[1] = 0;
    return L2A_OK;
}

int l2a_mail_begin(l2a_ctx* ctx, int m, const float* obs_host, long long obs_floats, hipStream_t stream,
                   l2a_mail_ticket* tk) {
    if (m < 1 || m > L2A_MAIL_KEYS || obs_floats < 0 || obs_floats > L2A_MAIL_OBS)
        return fail(ctx, L2A_EINVAL, "blocking plan: at most 64 envs / 4096 observation floats");
    const int rc = ensure_mail(ctx);
    if (rc != L2A_OK) return rc;
    tk->seq = ++ctx->mail_seq;
    tk->slot = (int)(tk->seq & 1ull);
    std::memcpy(ctx->mail_host->obs[tk->slot], obs_host, sizeof(float) * (size_t)obs_floats);
    tk->obs_dev = ctx->mail_dev->obs[tk->slot];
    tk->keys_dev = ctx->key_ring + (size_t)tk->slot * L2A_MAIL_KEYS;
    tk->next_keys = ctx->key_ring + (size_t)(tk->slot ^ 1) * L2A_MAIL_KEYS;
    // Key-slot bookkeeping: ring_dirty[s] = how many leading entries of slot s may be non-zero.  A mailbox launch
    // with m envs zeroes entries [0, m) of the OTHER slot in its epilogue; anything beyond needs a memset.
    if (ctx->ring_dirty[tk->slot] > 0)
        L2A_HIP(ctx, hipMemsetAsync(tk->keys_dev, 0, sizeof(unsigned long long) * L2A_MAIL_KEYS, stream));
    ctx->ring_dirty[tk->slot] = m;
    tk->t0_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
    return L2A_OK;
}

int l2a_mail_end(l2a_ctx* ctx, const l2a_mail_ticket& tk, int m, bool published, int launch_rc, hipStream_t stream,
                 unsigned long long* keys_host_out, const char* who) {
    auto now_us = [] {
        return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
    };
    auto dirty = [&] { ctx->ring_dirty[0] = ctx->ring_dirty[1] = L2A_MAIL_KEYS; };
    if (launch_rc != L2A_OK) { dirty(); return launch_rc; }
    l2a_mail* mh = ctx->mail_host;
    if (published) {
        if (m >= ctx->ring_dirty[tk.slot ^ 1]) ctx->ring_dirty[tk.slot ^ 1] = 0;   // zeroed by this launch's last tile
        // Sleep through most of the expected duration, then poll the mailbox word (host-mapped memory: no copy,
        // no hipStreamSynchronize wake-up latency).
        static const int sleep_mode = [] { const char* e = std::getenv("L2A_SYNC_SLEEP"); return e ? std::atoi(e) : 1; }();
        // The estimate belongs to a plan SHAPE: it is reset whenever the previous blocking launch of the context had
        // another one (a 1.5 ms plan's estimate made the first ~25 launches of a 0.2 ms plan oversleep, the estimate
        // decaying by 5 % per launch: +40 us per step averaged over 200), and whenever a launch was overslept.
        if (ctx->sync_shape != tk.shape) { ctx->sync_shape = tk.shape; ctx->sync_ema_us = 0.0; }
        const volatile unsigned long long* seqp = &mh->seq;
        bool overslept = false;
        // (a plan that had already published when the host came to wait - l2a_controller_finish after other host work - says
        // nothing about its duration: the estimate is left alone)
        const bool ready_at_entry = (__atomic_load_n(seqp, __ATOMIC_ACQUIRE) == tk.seq);
        if (sleep_mode && ctx->sync_ema_us > 400.0) {
            // (what is left of the expected duration: l2a_controller_finish may be called long after the launch)
            const double left = ctx->sync_ema_us * 0.8 - 100.0 - (now_us() - tk.t0_us);
            if (left > 20.0 && __atomic_load_n(seqp, __ATOMIC_ACQUIRE) != tk.seq) {
                usleep((useconds_t)left);
                overslept = (__atomic_load_n(seqp, __ATOMIC_ACQUIRE) == tk.seq);
            }
        }
        unsigned long long spins = 0;
        while (__atomic_load_n(seqp, __ATOMIC_ACQUIRE) != tk.seq) {
            _mm_pause();
            if ((++spins & 0xffffull) == 0) {
                const hipError_t q = hipStreamQuery(stream);
                if (q != hipSuccess && q != hipErrorNotReady) {
                    dirty();
                    return fail(ctx, L2A_EHIP, std::string(who) + ": " + hipGetErrorString(q));
                }
                const double waited = (now_us() - tk.t0_us) * 1e-6;
                if (q == hipSuccess && __atomic_load_n(seqp, __ATOMIC_ACQUIRE) != tk.seq && waited > 1.0) {
                    dirty();
                    return fail(ctx, L2A_EHIP, std::string(who) + ": the stream drained but the mailbox was never written");
                }
                if (waited > 60.0) {
                    dirty();
                    return fail(ctx, L2A_EHIP, std::string(who) + ": timed out waiting for the plan");
                }
            }
        }
        const double us = now_us() - tk.t0_us;
        if (!ready_at_entry)
            ctx->sync_ema_us = overslept ? 0.0 : (ctx->sync_ema_us == 0.0) ? us : 0.75 * ctx->sync_ema_us + 0.25 * us;
        for (int i = 0; i < m; ++i) keys_host_out[i] = mh->keys[i];
    } else {
        L2A_HIP(ctx, hipMemcpyAsync(mh->keys, tk.keys_dev, sizeof(unsigned long long) * (size_t)m, hipMemcpyDeviceToHost, stream));
        L2A_HIP(ctx, hipStreamSynchronize(stream));
        for (int i = 0; i < m; ++i) keys_host_out[i] = mh->keys[i];
    }
    if (*ctx->status_host != 0) {       // a tile-split partner never arrived: the caller relaunches unsplit
        *ctx->status_host = 0;
        return fail(ctx, L2A_ESPLIT, "a tile-split exchange timed out (relaunch with l2a_set_split(ctx, 0))");
    }
    return L2A_OK;
}

int l2a_plan_rs_sync(l2a_model* md, const float* obs_host, const float* actions, int m, int n, int h,
                     double discount, const l2a_reward* reward, int cand_offset, float* returns_out,
                     unsigned long long* keys_host_out, void* stream_v) {
    return l2a_plan_rs_sync_hook(md, obs_host, actions, m, n, h, discount, reward, cand_offset, returns_out, keys_host_out,
                                 stream_v, nullptr, nullptr);
}

void l2a_model_facts(const l2a_model* md, l2a_ctx** ctx, int* obs_dim, int* act_dim) {
    *ctx = md->ctx; *obs_dim = md->obs_dim; *act_dim = md->act_dim;
}

int l2a_plan_rs_sync_hook(l2a_model* md, const float* obs_host, const float* actions, int m, int n, int h,
                          double discount, const l2a_reward* reward, int cand_offset, float* returns_out,
                          unsigned long long* keys_host_out, void* stream_v, l2a_after_launch_fn hook, void* hook_arg,
                          l2a_mail_pending* pending) {
    if (!md) return L2A_EINVAL;
    l2a_ctx* ctx = md->ctx;
    ctx->stamps_us[0] = l2a_now_us();
    if (!obs_host || !actions || !reward || (!keys_host_out && !pending))
        return fail(ctx, L2A_EINVAL, "l2a_plan_rs_sync: null obs / actions / reward / keys_host_out");
    if (m < 1 || n < 1 || h < 1) return fail(ctx, L2A_EINVAL, "l2a_plan_rs_sync: m, n and h must be >= 1");
    if (m > L2A_MAIL_KEYS || (long long)m * md->obs_dim > L2A_MAIL_OBS)
        return fail(ctx, L2A_EINVAL, "l2a_plan_rs_sync: at most 64 envs / 4096 observation floats (use l2a_plan_rs)");
    if ((long long)m * n > 0x3fffffffLL || cand_offset < 0 || (long long)cand_offset + n > 0x7fffffffLL)
        return fail(ctx, L2A_EINVAL, "l2a_plan_rs_sync: too many candidates");
    if (reward->w_vel != 0.0f && (reward->vel_index < 0 || reward->vel_index >= md->obs_dim))
        return fail(ctx, L2A_EINVAL, "reward.vel_index out of range");
    if (reward->dist_coef != 0.0f && (reward->dist_index < 0 || reward->dist_index >= md->obs_dim))
        return fail(ctx, L2A_EINVAL, "reward.dist_index out of range");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    l2a_device_guard guard(ctx->device);
    int kind = ctx->kernel_kind;
    if (kind == L2A_KERNEL_AUTO) kind = md->mfma_ok ? L2A_KERNEL_MFMA : L2A_KERNEL_VALU;
    const bool publish = (kind == L2A_KERNEL_MFMA);        // the VALU kernel has no mailbox epilogue
    l2a_mail_ticket tk;
    int rc = l2a_mail_begin(ctx, m, obs_host, (long long)m * md->obs_dim, stream, &tk);
    if (rc != L2A_OK) return rc;
    tk.shape = (unsigned long long)(size_t)md ^ ((unsigned long long)m << 48) ^ ((unsigned long long)n << 24) ^ (unsigned long long)h;
    L2AKParams p;
    fill_model_params(md, p);
    p.obs0 = tk.obs_dev; p.actions = actions; p.returns_out = returns_out; p.best_key = tk.keys_dev;
    p.state_out = nullptr; p.obs_per_row = 0;
    p.ret_in = nullptr; p.disc0 = 1.0;
    p.m = m; p.n = n; p.h = h; p.cand_offset = cand_offset; p.discount = discount; p.rw = *reward;
    if (publish) {
        p.done_ctr = ctx->done_ctr;
        p.mail_keys = ctx->mail_dev->keys;
        p.mail_seq_ptr = &ctx->mail_dev->seq;
        p.mail_seq = tk.seq;
        p.next_keys = tk.next_keys;
    }
    ctx->stamps_us[1] = l2a_now_us();
    rc = launch_rollout(md, p, stream_v);
    ctx->stamps_us[2] = l2a_now_us();
    if (rc == L2A_OK && hook) hook(hook_arg);
    ctx->stamps_us[3] = l2a_now_us();
    if (pending && rc == L2A_OK) {
        pending->tk = tk; pending->publish = publish; pending->m = m; pending->stream = stream;
        pending->who = "l2a_plan_rs_sync"; pending->live = true;
        return L2A_OK;
    }
    rc = l2a_mail_end(ctx, tk, m, publish, rc, stream, keys_host_out, "l2a_plan_rs_sync");
    ctx->stamps_us[4] = l2a_now_us();
    return rc;
}

int l2a_plan_finish(l2a_ctx* ctx, l2a_mail_pending* pending, unsigned long long* keys_host_out) {
    if (!pending || !pending->live) return fail(ctx, L2A_ESTATE, "no plan is in flight");
    pending->live = false;
    l2a_device_guard guard(ctx->device);
    const int rc = l2a_mail_end(ctx, pending->tk, pending->m, pending->publish, L2A_OK, pending->stream, keys_host_out, pending->who);
    ctx->stamps_us[4] = l2a_now_us();
    return rc;
}

int l2a_plan_rs_chunk(l2a_model* md, const float* state, int state_per_row, const float* actions, int m, int n,
                      int h_chunk, int t0, double discount, const l2a_reward* reward, int cand_offset,
                      const float* returns_in, float* returns_out, float* state_out, unsigned long long* best_key,
                      void* stream_v) {
    if (!md) return L2A_EINVAL;
    l2a_ctx* ctx = md->ctx;
    if (!state || !actions || !reward) return fail(ctx, L2A_EINVAL, "l2a_plan_rs_chunk: null state/actions/reward");
    if (!returns_out) return fail(ctx, L2A_EINVAL, "l2a_plan_rs_chunk: returns_out is required");
    if (m < 1 || n < 1 || h_chunk < 1 || t0 < 0) return fail(ctx, L2A_EINVAL, "l2a_plan_rs_chunk: bad m / n / h_chunk / t0");
    if (t0 > 0 && (!returns_in || !state_per_row))
        return fail(ctx, L2A_EINVAL, "l2a_plan_rs_chunk: a continuation needs returns_in and per-row states");
    if ((long long)m * n > 0x3fffffffLL || cand_offset < 0 || (long long)cand_offset + n > 0x7fffffffLL)
        return fail(ctx, L2A_EINVAL, "l2a_plan_rs_chunk: too many candidates");
    if (reward->w_vel != 0.0f && (reward->vel_index < 0 || reward->vel_index >= md->obs_dim))
        return fail(ctx, L2A_EINVAL, "reward.vel_index out of range");
    if (reward->dist_coef != 0.0f && (reward->dist_index < 0 || reward->dist_index >= md->obs_dim))
        return fail(ctx, L2A_EINVAL, "reward.dist_index out of range");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    l2a_device_guard guard(ctx->device);
    if (best_key) L2A_HIP(ctx, hipMemsetAsync(best_key, 0, sizeof(unsigned long long) * (size_t)m, stream));
    L2AKParams p;
    fill_model_params(md, p);
    p.obs0 = state; p.obs_per_row = state_per_row ? 1 : 0;
    p.actions = actions; p.returns_out = returns_out; p.best_key = best_key; p.state_out = state_out;
    p.ret_in = (t0 > 0) ? returns_in : nullptr;
    double d0 = 1.0;    // discount ** t0 by the kernel's own recurrence (bit-identical continuation)
    for (int t = 0; t < t0; ++t) d0 *= discount;
    p.disc0 = d0;
    p.m = m; p.n = n; p.h = h_chunk; p.cand_offset = cand_offset; p.discount = discount; p.rw = *reward;
    return launch_rollout(md, p, stream_v);
}

int l2a_predict(l2a_model* md, const float* obs, const float* act, int rows, int n_blocks,
                float* next_obs_out, void* stream_v) {
    if (!md) return L2A_EINVAL;
    l2a_ctx* ctx = md->ctx;
    if (!obs || !act || !next_obs_out) return fail(ctx, L2A_EINVAL, "l2a_predict: null pointer");
    if (rows < 1 || n_blocks < 1 || rows % n_blocks != 0)
        return fail(ctx, L2A_EINVAL, "l2a_predict: rows must be a positive multiple of n_blocks");
    if (md->mode != L2A_MODE_PER_BLOCK && n_blocks != 1)
        return fail(ctx, L2A_EINVAL, "l2a_predict: n_blocks > 1 needs per-block mode");
    L2AKParams p;
    fill_model_params(md, p);
    p.obs0 = obs; p.actions = act; p.returns_out = nullptr; p.best_key = nullptr;
    p.state_out = next_obs_out; p.obs_per_row = 1;
    p.ret_in = nullptr; p.disc0 = 1.0;
    p.m = n_blocks; p.n = rows / n_blocks; p.h = 1; p.cand_offset = 0; p.discount = 1.0;
    std::memset(&p.rw, 0, sizeof(p.rw));
    return launch_rollout(md, p, stream_v);
}

}  // extern "C"
