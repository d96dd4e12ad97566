// l2a_lstm_api.hip - host side of the recurrent-planner entry points of include/l2a.h
// (l2a_lstm_*): model storage (raw TF-layout weights + MFMA-packed copies + normalisation), weight
// re-packing and the launch logic of the kernels in l2a_lstm.h.  Everything is enqueued on the
// caller's stream.

#include "l2a_host.h"
#include "l2a_lstm.h"
#include "l2a_lstm_valu.h"
#include "l2a_lstm_launch.h"

#include <cstring>
#include <string>
#include <vector>

struct l2a_lstm {
    l2a_ctx* ctx = nullptr;
    int obs_dim = 0, act_dim = 0, in_dim = 0, units = 0;
    int cell_act = L2A_ACT_TANH, output_act = L2A_ACT_IDENTITY;
    bool mfma_ok = false;
    int UTW = 0, KG0 = 0, OT = 0;
    float* wblk = nullptr;
    long long total = 0;
    long long raw_wk = 0, raw_bk = 0, raw_wo = 0, raw_bo = 0, pk_wg = 0, pk_wout = 0, pk_bout = 0, nm_off = 0;
    bool weights_set = false, norm_set = false;
    std::vector<float> norm_stage;
};

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
    p.obs_dim = md->obs_dim; p.act_dim = md->act_dim; p.in_dim = md->in_dim; p.units = md->units;
    p.cell_act = md->cell_act; p.output_act = md->output_act;
    p.KG0 = md->KG0; p.OT = md->OT;
    p.disc0 = 1.0;
}

int launch(l2a_lstm* md, L2ALstmParams& p, void* stream_v) {
    l2a_ctx* ctx = md->ctx;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    l2a_device_guard guard(ctx->device);
    if (!md->weights_set) return l2a_fail(ctx, L2A_ESTATE, "LSTM weights were never set");
    if (!md->norm_set) return l2a_fail(ctx, L2A_ESTATE, "LSTM normalisation was never set");
    int kind = ctx->kernel_kind;
    if (kind == L2A_KERNEL_AUTO) kind = md->mfma_ok ? L2A_KERNEL_MFMA : L2A_KERNEL_VALU;
    if (kind == L2A_KERNEL_MFMA && !md->mfma_ok)
        return l2a_fail(ctx, L2A_EINVAL, "LSTM shape is not eligible for the MFMA kernel "
                                         "(needs units in {128, 256, 512}, obs_dim <= 64, act_dim <= 16)");
    p.dbg = ctx->dbg;
    if (kind == L2A_KERNEL_MFMA) {
        const int nt = 1;
        const int UT = L2A_NW * md->UTW, U = md->units;
        p.tiles_per_env = l2a_ceil_div(p.n, 16 * nt);
        const int smem = 2 * nt * UT * 64 * 16 + 2 * (2 * L2A_NW * nt * md->OT * 64) * 16 +
                         (32 * md->KG0 + 48 * md->OT + 4 * U) * 4;
        if (smem > ctx->lds_per_block)
            return l2a_fail(ctx, L2A_EINVAL, "LDS budget exceeded (" + std::to_string(smem) + " B)");
        const unsigned grid = (unsigned)((long long)p.m * p.tiles_per_env);
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

void l2a_lstm_destroy(l2a_lstm* md) {
    if (!md) return;
    if (md->wblk) {
        (void)hipDeviceSynchronize();
        (void)hipFree(md->wblk);
    }
    delete md;
}

int l2a_lstm_set_weights(l2a_lstm* md, const void* const* device_ptrs, void* stream_v) {
    if (!md) return L2A_EINVAL;
    l2a_ctx* ctx = md->ctx;
    if (!device_ptrs) return l2a_fail(ctx, L2A_EINVAL, "device_ptrs is null");
    for (int i = 0; i < 4; ++i)
        if (!device_ptrs[i]) return l2a_fail(ctx, L2A_EINVAL, "null LSTM parameter pointer " + std::to_string(i));
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    l2a_device_guard guard(ctx->device);
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
