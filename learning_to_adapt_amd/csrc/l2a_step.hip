// l2a_step.hip - the controller step as ONE C-ABI call (include/l2a.h: l2a_controller_*).
//
// What `MPCController.get_actions` (policies/mpc_controller.py:59-69,108-129) and `RNNMPCController.get_actions`
// (policies/rnn_mpc_controller.py:57-65,112-134) are to their caller in parity mode: float64 observations in, the float64 first
// action of the best candidate out, NumPy's global generator left exactly where the reference's own draw leaves it.  Until
// round 5 the pieces of a step were glued together in Python (a Condition shared with a worker thread under the GIL for the
// candidates drawn ahead, ~10 ctypes / torch calls around the launch): ~0.1 ms per call beside a 0.18 - 1.4 ms kernel.  Here:
//   take   the block of candidates the chain of csrc/l2a_rng.c drew (and this file's callback uploaded) while the previous plan
//          ran - adopted only if the global generator's words are still the state the block started from
//   launch the fused rollout from host-mapped observations (l2a_plan_rs_sync's path; recurrent: + the state advance)
//   kick   the producer of the next block (between launch and wait)
//   wait   for the mailbox word, decode the keys, gather the winners' float64 first actions from the block's `cand_a`
// When no valid block is waiting (first call, a foreign consumer of np.random between two steps) the step draws the candidates
// itself - the reference's draw from the GLOBAL generator, on the helper's threads, into the idle slot, uploaded on the launch
// stream in front of the kernel - and re-arms the chain behind that draw (L2A_STEP_DREW); while steps keep missing it re-arms only
// every 16th step (every block drawn ahead would be thrown away).  L2A_STEP_MISS is left for a controller that cannot serve the
// call at all (a forked child: its producer thread and its HIP state did not come along).
#include <hip/hip_runtime.h>

#include <cstring>
#include <new>
#include <string>

#include "l2a_host.h"
#include "l2a_philox.h"
#include "l2a_rng.h"

struct l2a_controller {
    l2a_ctx* ctx = nullptr;
    l2a_model* mlp = nullptr;
    l2a_lstm* rnn = nullptr;
    int m = 0, n = 0, h = 0, obs_dim = 0, act_dim = 0, units = 0;
    double discount = 1.0;
    l2a_reward rw;
    void* np_addr = nullptr;
    size_t act_floats = 0;
    float* pin[2] = {nullptr, nullptr};         // page-locked staging of the fp32 candidate tensor [h, m * n, act_dim]
    float* dev[2] = {nullptr, nullptr};         // its copy in HBM (what the rollout reads)
    double* c64[2] = {nullptr, nullptr};        // float64 `cand_a` = the first horizon step's rows [m * n, act_dim]
    hipStream_t side = nullptr;                 // the producer's upload stream
    l2a_ahead* chain = nullptr;
    int slot = -1;                              // block of the latest successful step
    bool producer_bound = false;                // the producer thread has made ctx->device current
    std::string upload_err;
    double stage_us[8] = {0};
    unsigned long long steps = 0, relaunches = 0, sync_draws = 0;
    double low[16], high[16];
    int rng_threads = 1;
    int misses_in_row = 0;
    unsigned long long cooldown = 0;
    // device-RNG mode (`rng="device"`): no chain - the candidates are drawn by a Philox kernel in front of the plan
    bool device_rng = false;
    unsigned long long seed = 0, calls = 0;
    float* lowr_dev = nullptr;                  // [2][16]: low | high - low, fp32
    // a step between l2a_controller_begin and l2a_controller_finish
    l2a_mail_pending pending;
    bool in_flight = false;
    int result = L2A_OK;                        // what the finished step reports (L2A_OK / L2A_STEP_DREW / L2A_STEP_UNSPLIT)
    unsigned long long offset = 0;              // device-RNG mode: the step's first stream element
    float obs32[L2A_MAIL_OBS];                  // the observations as staged (a relaunch in `finish` stages them again)
    const float *c0 = nullptr, *h0 = nullptr;   // recurrent: the caller's state pointers of the step in flight
    float *c1 = nullptr, *h1 = nullptr;
    void* stream = nullptr;
    double t_begin = 0.0, t_taken = 0.0;
    // sharded plan (l2a_controller_create_sharded): this rank rolls out candidates [lo, hi) of every env and the ranks' keys meet
    // in ONE int64 MAX all-reduce of [keys (m), launch flag, digest, MASK - digest] per step
    bool sharded = false;                       // (also with world = 1: the same code path with a one-rank collective)
    int rank = 0, world = 1, lo = 0, hi = 0;
    l2a_reduce_fn reduce = nullptr;             // null: RCCL through the context's communicator (l2a_comm_init)
    void* reduce_arg = nullptr;
    float* obs_map_host = nullptr;              // host-mapped observation staging (read by the kernel directly)
    float* obs_map_dev = nullptr;
    unsigned long long* keys_dev = nullptr;     // [m]
    unsigned long long* payload_dev = nullptr;  // [m + 3]
    unsigned long long* payload_host = nullptr; // page-locked [m + 3]
    hipEvent_t payload_ev = nullptr;
    unsigned long long digest = 0;
    size_t glob_floats = 0;                     // h * m * n * act_dim: the WHOLE plan's candidate tensor (the device stream's step)
};

extern "C" unsigned long long l2a_mt19937_state_digest(const void* addr);      // csrc/l2a_rng.c

// Device-RNG mode: the candidate tensor [h, m * n, act_dim] from the counter-based stream (seed, offset + element): four elements
// per thread (one Philox block), the action dimension of an element = its index modulo act_dim.
__global__ void __launch_bounds__(256) l2a_uniform_fill_k(unsigned long long seed, unsigned long long offset, long long total,
                                                          int act_dim, const float* __restrict__ lowr, float* __restrict__ out) {
    const long long b = (long long)blockIdx.x * 256 + threadIdx.x;      // block of four elements
    if (4 * b >= total) return;
    unsigned int c[4];
    l2a_philox4x32_10(seed, (offset >> 2) + (unsigned long long)b, L2A_PHILOX_UNIFORM, c);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long long e = 4 * b + i;
        if (e < total) {
            const int k = (int)(e % act_dim);
            out[e] = l2a_uniform_from_word(c[i], lowr[k], lowr[16 + k]);
        }
    }
}

// The same stream for ONE rank of a sharded plan: this rank's candidates [lo, lo + n_local) of every env, local tensor
// [h, m * n_local, act_dim]; an element takes the value of its GLOBAL position ((t m + i) n + j) act_dim + k in the stream, so the
// candidates - and with them the plan - do not depend on the number of ranks.  One element per thread (its Philox block computed
// whole: four times the rounds of the contiguous fill, on a tensor an eighth of the size).
__global__ void __launch_bounds__(256) l2a_uniform_fill_shard_k(unsigned long long seed, unsigned long long offset, long long total_local,
                                                                int n, int lo, int n_local, int act_dim, const float* __restrict__ lowr,
                                                                float* __restrict__ out) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total_local) return;
    const int k = (int)(e % act_dim);
    const long long r = e / act_dim;                // local row (t m + i) n_local + jl
    const long long ti = r / n_local;
    const int jl = (int)(r - ti * n_local);
    const unsigned long long g = (unsigned long long)((ti * n + lo + jl) * act_dim + k);
    out[e] = l2a_philox_uniform(seed, offset + g, lowr[k], lowr[16 + k]);
}

namespace {

int fail(l2a_ctx* ctx, int code, const std::string& msg) { return l2a_fail(ctx, code, msg); }

// page-locked / device words of a sharded controller (both RNG modes)
hipError_t alloc_sharded(l2a_controller* c) {
    hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&c->obs_map_host), sizeof(float) * L2A_MAIL_OBS, hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void**>(&c->obs_map_dev), c->obs_map_host, 0);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->keys_dev), sizeof(unsigned long long) * (size_t)c->m);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->payload_dev), sizeof(unsigned long long) * (size_t)(c->m + 3));
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&c->payload_host), sizeof(unsigned long long) * (size_t)(c->m + 3), hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->payload_ev, hipEventDisableTiming);
    return e;
}

// Producer thread, after the block's draw: one H2D copy on the side stream, completed before the block is marked ready - the
// consumer neither waits on an event nor launches behind an unfinished copy.
int upload_block(void* arg, int slot) {
    l2a_controller* c = static_cast<l2a_controller*>(arg);
    if (!c->producer_bound) {
        if (hipSetDevice(c->ctx->device) != hipSuccess) { c->upload_err = "hipSetDevice on the producer thread failed"; return -1; }
        c->producer_bound = true;
    }
    if (c->hi == c->lo) return 0;                               // more ranks than candidates: this rank rolls nothing out
    hipError_t e = hipMemcpyAsync(c->dev[slot], c->pin[slot], c->act_floats * sizeof(float), hipMemcpyHostToDevice, c->side);
    if (e == hipSuccess) e = hipStreamSynchronize(c->side);
    if (e != hipSuccess) { c->upload_err = std::string("uploading a candidate block: ") + hipGetErrorString(e); return -1; }
    return 0;
}

void kick_next(void* arg) { (void)l2a_ahead_next(static_cast<l2a_controller*>(arg)->chain); }

// After a synchronous draw: the chain restarts at the generator's new state - unless steps keep missing (a consumer of np.random
// runs between the controller's steps): then only every 16th step tries again.
void kick_arm(void* arg) {
    l2a_controller* c = static_cast<l2a_controller*>(arg);
    c->misses_in_row += 1;
    if (c->misses_in_row > 2 && (++c->cooldown % 16) != 0) return;
    (void)l2a_ahead_arm(c->chain, c->np_addr);
}

int create(l2a_ctx* ctx, l2a_model* mlp, l2a_lstm* rnn, int obs_dim, int act_dim, int units, int m, int n, int h,
           const double* low, const double* high, double discount, const l2a_reward* reward, void* np_state_addr,
           int rng_threads, l2a_controller** out, bool device_rng = false, unsigned long long seed = 0, int rank = 0,
           int world = 1, l2a_reduce_fn reduce = nullptr, void* reduce_arg = nullptr, bool sharded = false) {
    if (!out) return fail(ctx, L2A_EINVAL, "l2a_controller_create: out is null");
    *out = nullptr;
    if (!low || !high || !reward || (!np_state_addr && !device_rng))
        return fail(ctx, L2A_EINVAL, "l2a_controller_create: null low / high / reward / generator state address");
    if (m < 1 || m > L2A_MAIL_KEYS || (long long)m * obs_dim > L2A_MAIL_OBS || n < 1 || h < 1)
        return fail(ctx, L2A_EINVAL, "l2a_controller_create: needs 1 <= m <= 64 envs (at most 4096 observation floats), n >= 1, h >= 1");
    if (act_dim < 1 || act_dim > 16) return fail(ctx, L2A_EINVAL, "l2a_controller_create: the host draw takes 1 <= act_dim <= 16");
    if ((long long)m * n > 0x3fffffffLL) return fail(ctx, L2A_EINVAL, "l2a_controller_create: too many candidates");
    if (l2a_rng_version() < 8) return fail(ctx, L2A_ESTATE, "l2a_controller_create: libl2a_rng.so is older than this library");
    l2a_controller* c = new (std::nothrow) l2a_controller();
    if (!c) return fail(ctx, L2A_EHIP, "l2a_controller_create: out of memory");
    c->ctx = ctx; c->mlp = mlp; c->rnn = rnn;
    c->m = m; c->n = n; c->h = h; c->obs_dim = obs_dim; c->act_dim = act_dim; c->units = units;
    c->discount = discount; c->rw = *reward; c->np_addr = np_state_addr;
    for (int k = 0; k < act_dim; ++k) { c->low[k] = low[k]; c->high[k] = high[k]; }
    c->rng_threads = rng_threads < 1 ? 1 : rng_threads;
    c->rank = rank; c->world = world; c->reduce = reduce; c->reduce_arg = reduce_arg; c->sharded = sharded;
    c->lo = (int)((long long)rank * n / world);                 // contiguous candidate ranges (MPCController._shard_range)
    c->hi = (int)((long long)(rank + 1) * n / world);
    const int n_local = c->hi - c->lo;
    c->act_floats = (size_t)h * m * (n_local > 0 ? n_local : 1) * act_dim;
    c->glob_floats = (size_t)h * m * n * act_dim;
    l2a_device_guard guard(ctx->device);
    c->device_rng = device_rng; c->seed = seed;
    hipError_t e = hipSuccess;
    if (device_rng) {
        // the stream's elements are addressed in blocks of four: a step's tensor starts on a block boundary
        e = hipMalloc(reinterpret_cast<void**>(&c->dev[0]), ((c->act_floats + 3) / 4 * 4) * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->lowr_dev), 32 * sizeof(float));
        if (e == hipSuccess) {
            float lr[32] = {0};
            for (int k = 0; k < act_dim; ++k) { lr[k] = (float)low[k]; lr[16 + k] = (float)high[k] - (float)low[k]; }
            e = hipMemcpy(c->lowr_dev, lr, sizeof(lr), hipMemcpyHostToDevice);
        }
        if (e == hipSuccess && sharded) e = alloc_sharded(c);
        if (e != hipSuccess) {
            const std::string msg = std::string("l2a_controller_create_device: ") + hipGetErrorString(e);
            l2a_controller_destroy(c);
            return fail(ctx, L2A_EHIP, msg);
        }
        *out = c;
        return L2A_OK;
    }
    e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
    for (int s = 0; s < 2 && e == hipSuccess; ++s) {
        e = hipHostMalloc(reinterpret_cast<void**>(&c->pin[s]), c->act_floats * sizeof(float), hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->dev[s]), c->act_floats * sizeof(float));
        if (e == hipSuccess) {
            c->c64[s] = static_cast<double*>(std::malloc(sizeof(double) * (size_t)m * n * act_dim));
            if (!c->c64[s]) e = hipErrorOutOfMemory;
        }
    }
    if (e == hipSuccess) {
        // rows of the reference's draw: h * n * m (mpc_controller.py:114), row r <-> candidate r % n; the whole env-major tensor
        // goes up (one GPU: every candidate is local); the first n * m rows are kept in float64 (`cand_a`, :118)
        // (a sharded plan: every rank consumes the generator for ALL h * n * m rows and keeps candidates [lo, hi) of every env)
        c->chain = l2a_ahead_create((long long)h * n * m, act_dim, low, high, n, c->lo, c->hi, (long long)n * m, c->pin[0], c->pin[1],
                                    c->c64[0], c->c64[1], rng_threads, upload_block, c);
        if (!c->chain) e = hipErrorInvalidValue;
    }
    if (e == hipSuccess && sharded) e = alloc_sharded(c);
    if (e != hipSuccess) {
        const std::string msg = std::string("l2a_controller_create: ") + hipGetErrorString(e);
        l2a_controller_destroy(c);
        return fail(ctx, L2A_EHIP, msg);
    }
    *out = c;
    return L2A_OK;
}

// Sharded plan: this rank's launch, the payload packed on the device behind it, the ONE collective of the step, and the copy
// of the reduced words to page-locked memory - all in stream order, nothing on the host waits (policies/mpc_controller.py
// `_combine_keys` did the same from Python with torch.distributed).
int launch_sharded(l2a_controller* c, bool first) {
    l2a_ctx* ctx = c->ctx;
    l2a_device_guard guard(ctx->device);
    hipStream_t stream = reinterpret_cast<hipStream_t>(c->stream);
    const int n_local = c->hi - c->lo;
    std::memcpy(c->obs_map_host, c->obs32, sizeof(float) * (size_t)c->m * c->obs_dim);
    ctx->stamps_us[1] = l2a_now_us();
    if (n_local > 0) {
        const int rc = l2a_plan_rs(c->mlp, c->obs_map_dev, c->dev[c->slot], c->m, n_local, c->h, c->discount, &c->rw, c->lo, nullptr,
                                   c->keys_dev, c->stream);
        if (rc != L2A_OK) return rc;
    } else {
        L2A_HIP(ctx, hipMemsetAsync(c->keys_dev, 0, sizeof(unsigned long long) * (size_t)c->m, stream));    // the neutral key
    }
    int rc = l2a_plan_payload(ctx, c->keys_dev, c->m, c->digest, c->payload_dev, c->stream);
    if (rc != L2A_OK) return rc;
    ctx->stamps_us[2] = l2a_now_us();
    if (first && !c->device_rng) (c->result == L2A_STEP_DREW ? kick_arm : kick_next)(c);
    ctx->stamps_us[3] = l2a_now_us();
    rc = c->reduce ? c->reduce(c->reduce_arg, c->payload_dev, c->m + 3, c->stream)
                   : l2a_allreduce_best(ctx, c->payload_dev, c->m + 3, c->stream);       // RCCL: uint64 MAX over xGMI
    if (rc != L2A_OK) return c->reduce ? fail(ctx, L2A_EHIP, "l2a_controller_step: the caller's reduce function failed") : rc;
    L2A_HIP(ctx, hipMemcpyAsync(c->payload_host, c->payload_dev, sizeof(unsigned long long) * (size_t)(c->m + 3), hipMemcpyDeviceToHost, stream));
    L2A_HIP(ctx, hipEventRecord(c->payload_ev, stream));
    return L2A_OK;
}

// First half of a step: everything that touches the generator (take / draw, re-arm), the staging and the launch.  Returns
// L2A_OK (plan in flight), L2A_STEP_MISS (nothing consumed or launched) or a negative code.
int begin(l2a_controller* c, const double* obs, const float* c0, const float* h0, float* c1, float* h1, void* stream) {
    l2a_ctx* ctx = c->ctx;
    if (!obs) return fail(ctx, L2A_EINVAL, "l2a_controller_begin: null obs");
    if (c->in_flight) return fail(ctx, L2A_ESTATE, "l2a_controller_begin: the previous step was not finished (l2a_controller_finish)");
    const double t0 = l2a_now_us();
    int slot = 0;
    bool drew = false;
    if (c->device_rng) {
        // candidates of this step: elements [offset, offset + h m n act_dim) of the stream (seed) - drawn on the launch stream
        const unsigned long long per_step = (unsigned long long)((c->glob_floats + 3) / 4 * 4);
        c->offset = c->calls * per_step;
        l2a_device_guard guard(ctx->device);
        if (!c->sharded || c->world == 1) {
            const long long total = (long long)c->glob_floats;
            hipLaunchKernelGGL(l2a_uniform_fill_k, dim3((unsigned)((total + 1023) / 1024)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                               c->seed, c->offset, total, c->act_dim, c->lowr_dev, c->dev[0]);
        } else if (c->hi > c->lo) {
            const long long total = (long long)c->h * c->m * (c->hi - c->lo) * c->act_dim;
            hipLaunchKernelGGL(l2a_uniform_fill_shard_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                               c->seed, c->offset, total, c->n, c->lo, c->hi - c->lo, c->act_dim, c->lowr_dev, c->dev[0]);
        }
        L2A_HIP(ctx, hipGetLastError());
    } else if ((slot = l2a_ahead_take(c->chain, c->np_addr)) < 0) {
        if (!c->upload_err.empty()) { const std::string msg = c->upload_err; c->upload_err.clear(); return fail(ctx, L2A_EHIP, msg); }
        slot = l2a_ahead_idle_slot(c->chain);
        if (slot < 0) return L2A_STEP_MISS;                     // (a forked child, or a chain somebody else is driving)
        // the reference's own draw (mpc_controller.py:67-69,114) from the global generator, advanced in place
        struct np_state { unsigned int key[624]; int pos; };
        np_state* g = static_cast<np_state*>(c->np_addr);
        if (l2a_mt19937_uniform_rows(g->key, &g->pos, (long long)c->h * c->n * c->m, c->act_dim, c->low, c->high, c->n, c->lo, c->hi,
                                     c->pin[slot], (long long)c->n * c->m, c->c64[slot], c->rng_threads) != 0)
            return fail(ctx, L2A_EINVAL, "l2a_controller_step: the generator state at np_state_addr is not a legacy MT19937 state");
        l2a_device_guard guard(ctx->device);
        if (c->hi > c->lo)
            L2A_HIP(ctx, hipMemcpyAsync(c->dev[slot], c->pin[slot], c->act_floats * sizeof(float), hipMemcpyHostToDevice,
                                        reinterpret_cast<hipStream_t>(stream)));
        drew = true;
        c->sync_draws += 1;
    } else {
        c->misses_in_row = 0;
    }
    c->slot = slot;
    c->t_begin = t0;
    c->t_taken = l2a_now_us();
    const int no = c->m * c->obs_dim;
    for (int i = 0; i < no; ++i) c->obs32[i] = (float)obs[i];       // np.float64 -> np.float32 (round to nearest even), as the host cast
    c->c0 = c0; c->h0 = h0; c->c1 = c1; c->h1 = h1; c->stream = stream;
    c->result = drew ? L2A_STEP_DREW : L2A_OK;
    if (c->sharded) {
        // what this rank's candidates were drawn from: the generator as this step's draw left it (every rank must agree)
        // (device mode: the stream's seed and position - ranks seeded differently, or out of step, would plan on different candidates)
        c->digest = c->device_rng ? (c->seed * 0x9E3779B97F4A7C15ull) ^ (c->calls + 1ull) : l2a_mt19937_state_digest(c->np_addr);
        ctx->stamps_us[0] = t0;
        const int rc = launch_sharded(c, true);
        if (rc != L2A_OK) return rc;
        c->in_flight = true;
        return L2A_OK;
    }
    l2a_after_launch_fn hook = c->device_rng ? nullptr : (drew ? kick_arm : kick_next);
    int rc;
    if (c->mlp)
        rc = l2a_plan_rs_sync_hook(c->mlp, c->obs32, c->dev[slot], c->m, c->n, c->h, c->discount, &c->rw, 0, nullptr, nullptr, stream,
                                   hook, c, &c->pending);
    else
        rc = l2a_lstm_plan_rs_sync_hook(c->rnn, c->obs32, c0, h0, c->dev[slot], c->m, c->n, c->h, c->discount, &c->rw, 0, nullptr,
                                        c1, h1, stream, hook, c, &c->pending);
    if (rc != L2A_OK) return rc;
    c->in_flight = true;
    return L2A_OK;
}

// Second half: wait for the keys (a launch that lost its tile-split partner is repeated unsplit - same bits; the generator is not
// touched again), decode, gather the winners' float64 first actions.
int finish(l2a_controller* c, double* action_out, long long* index_out, float* return_out) {
    l2a_ctx* ctx = c->ctx;
    if (!action_out) return fail(ctx, L2A_EINVAL, "l2a_controller_finish: null action_out");
    if (!c->in_flight) return fail(ctx, L2A_ESTATE, "l2a_controller_finish: no step is in flight (l2a_controller_begin)");
    c->in_flight = false;
    const int slot = c->slot;
    unsigned long long keys[L2A_MAIL_KEYS];
    int result = c->result;
    int rc = L2A_OK;
    if (c->sharded) {
        l2a_device_guard guard(ctx->device);
        for (int attempt = 0; attempt < 2; ++attempt) {
            L2A_HIP(ctx, hipEventSynchronize(c->payload_ev));
            ctx->stamps_us[4] = l2a_now_us();
            const unsigned long long* v = c->payload_host;
            if (v[c->m + 1] + v[c->m + 2] != L2A_DIGEST_MASK)
                return fail(ctx, L2A_ESTATE, c->device_rng
                    ? "candidate sharding needs identical seeds and step counts on every rank (device RNG: build every rank's controller "
                      "with the same seed at the same step)"
                    : "candidate sharding needs identical np.random global state on every rank (seed all ranks alike and "
                      "keep other consumers of the generator off the planning process; the shards themselves are disjoint)");
            if (v[c->m] == 0) break;
            // SOME rank's launch lost its tile-split partner: the reduced flag is the same on every rank, so all of them switch to the
            // unsplit geometry (bit-identical results) and repeat launch + collective together
            // (also a rank that runs unsplit already: it must stay in step with the others' collective)
            *ctx->status_host = 0;
            if (attempt == 1) return fail(ctx, L2A_ESPLIT, "l2a_controller_step: some rank's rollout was flagged invalid twice");
            (void)l2a_set_split(ctx, 0);
            c->relaunches += 1;
            result = L2A_STEP_UNSPLIT;
            rc = launch_sharded(c, false);
            if (rc != L2A_OK) return rc;
        }
        for (int i = 0; i < c->m; ++i) keys[i] = c->payload_host[i];
    } else {
        rc = l2a_plan_finish(ctx, &c->pending, keys);
    }
    if (rc == L2A_ESPLIT) {
        // a tile-split partner was not co-resident: the unsplit geometry gives the same bits (the caller is told: L2A_STEP_UNSPLIT)
        if (ctx->split_policy == 0)
            return fail(ctx, L2A_ESPLIT, "l2a_controller_step: the rollout was flagged invalid with the tile split disabled");
        (void)l2a_set_split(ctx, 0);
        c->relaunches += 1;
        result = L2A_STEP_UNSPLIT;
        if (c->mlp)
            rc = l2a_plan_rs_sync_hook(c->mlp, c->obs32, c->dev[slot], c->m, c->n, c->h, c->discount, &c->rw, 0, nullptr, keys, c->stream,
                                       nullptr, nullptr);
        else
            rc = l2a_lstm_plan_rs_sync_hook(c->rnn, c->obs32, c->c0, c->h0, c->dev[slot], c->m, c->n, c->h, c->discount, &c->rw, 0, keys,
                                            c->c1, c->h1, c->stream, nullptr, nullptr);
        if (rc == L2A_ESPLIT)
            return fail(ctx, L2A_ESPLIT, "l2a_controller_step: the rollout was flagged invalid with the tile split disabled");
    }
    if (rc != L2A_OK) return rc;
    const double t2 = l2a_now_us();
    for (int i = 0; i < c->m; ++i) {
        float ret = 0.0f;
        int idx = 0;
        l2a_key_decode(keys[i], &ret, &idx);
        if (idx < 0 || idx >= c->n) return fail(ctx, L2A_EHIP, "l2a_controller_step: the arg-max key holds no candidate index");
        if (index_out) index_out[i] = idx;
        if (return_out) return_out[i] = ret;
        if (c->device_rng) {
            // the winner's first action, recomputed from the counter-based stream: element (row i n + idx of step 0, dim k) -
            // the fp32 value the kernel planned on, as float64 (no gather launch, no copy back)
            for (int k = 0; k < c->act_dim; ++k) {
                const unsigned long long e = c->offset + ((unsigned long long)i * c->n + idx) * c->act_dim + k;
                action_out[(size_t)i * c->act_dim + k] =
                    (double)l2a_philox_uniform(c->seed, e, (float)c->low[k], (float)c->high[k] - (float)c->low[k]);
            }
        } else {
            std::memcpy(action_out + (size_t)i * c->act_dim, c->c64[slot] + ((size_t)i * c->n + idx) * c->act_dim,
                        sizeof(double) * (size_t)c->act_dim);                // cand_a[i, idx] (:118,129)
        }
    }
    const double t3 = l2a_now_us();
    c->steps += 1;
    c->calls += 1;
    const double* st = ctx->stamps_us;
    c->stage_us[0] = c->t_taken - c->t_begin;   // take (compare + adopt the block; waits only if the producer is late)
    c->stage_us[1] = st[1] - c->t_taken;        // observation cast + staging
    c->stage_us[2] = st[2] - st[1];             // launch call(s)
    c->stage_us[3] = st[3] - st[2];             // producer kick
    c->stage_us[4] = st[4] - st[3];             // wait for the keys (begin -> finish: whatever the host did in between is in here)
    c->stage_us[5] = t3 - t2;                   // decode + gather
    c->stage_us[6] = t3 - c->t_begin;           // whole step
    return result;
}

int step(l2a_controller* c, const double* obs, const float* c0, const float* h0, float* c1, float* h1, double* action_out,
         long long* index_out, float* return_out, void* stream) {
    if (!obs || !action_out) return fail(c->ctx, L2A_EINVAL, "l2a_controller_step: null obs / action_out");
    const int rc = begin(c, obs, c0, h0, c1, h1, stream);
    if (rc != L2A_OK) return rc;
    return finish(c, action_out, index_out, return_out);
}

}  // namespace

extern "C" {

int l2a_controller_create(l2a_model* model, int m, int n, int h, const double* low, const double* high, double discount,
                          const l2a_reward* reward, void* np_state_addr, int rng_threads, l2a_controller** out) {
    if (!model) return L2A_EINVAL;
    l2a_ctx* ctx = nullptr;
    int obs_dim = 0, act_dim = 0;
    l2a_model_facts(model, &ctx, &obs_dim, &act_dim);
    return create(ctx, model, nullptr, obs_dim, act_dim, 0, m, n, h, low, high, discount, reward, np_state_addr, rng_threads, out);
}

int l2a_controller_create_sharded(l2a_model* model, int m, int n, int h, const double* low, const double* high, double discount,
                                  const l2a_reward* reward, void* np_state_addr, int rng_threads, int rank, int world,
                                  l2a_reduce_fn reduce, void* reduce_arg, l2a_controller** out) {
    if (!model) return L2A_EINVAL;
    l2a_ctx* ctx = nullptr;
    int obs_dim = 0, act_dim = 0;
    l2a_model_facts(model, &ctx, &obs_dim, &act_dim);
    if (world < 1 || rank < 0 || rank >= world) return fail(ctx, L2A_EINVAL, "l2a_controller_create_sharded: bad rank / world");
    if (!reduce && (!ctx->comm || ctx->comm_world != world || ctx->comm_rank != rank))
        return fail(ctx, L2A_ESTATE, "l2a_controller_create_sharded: no reduce function and no communicator of this rank / world (l2a_comm_init)");
    return create(ctx, model, nullptr, obs_dim, act_dim, 0, m, n, h, low, high, discount, reward, np_state_addr, rng_threads, out,
                  false, 0, rank, world, reduce, reduce_arg, true);
}

int l2a_controller_create_sharded_device(l2a_model* model, int m, int n, int h, const double* low, const double* high, double discount,
                                         const l2a_reward* reward, unsigned long long seed, int rank, int world,
                                         l2a_reduce_fn reduce, void* reduce_arg, l2a_controller** out) {
    if (!model) return L2A_EINVAL;
    l2a_ctx* ctx = nullptr;
    int obs_dim = 0, act_dim = 0;
    l2a_model_facts(model, &ctx, &obs_dim, &act_dim);
    if (world < 1 || rank < 0 || rank >= world) return fail(ctx, L2A_EINVAL, "l2a_controller_create_sharded_device: bad rank / world");
    if (!reduce && (!ctx->comm || ctx->comm_world != world || ctx->comm_rank != rank))
        return fail(ctx, L2A_ESTATE, "l2a_controller_create_sharded_device: no reduce function and no communicator of this rank / world (l2a_comm_init)");
    return create(ctx, model, nullptr, obs_dim, act_dim, 0, m, n, h, low, high, discount, reward, nullptr, 1, out, true, seed, rank, world,
                  reduce, reduce_arg, true);
}

int l2a_lstm_controller_create(l2a_lstm* model, int m, int n, int h, const double* low, const double* high, double discount,
                               const l2a_reward* reward, void* np_state_addr, int rng_threads, l2a_controller** out) {
    if (!model) return L2A_EINVAL;
    l2a_ctx* ctx = nullptr;
    int obs_dim = 0, act_dim = 0, units = 0;
    l2a_lstm_facts(model, &ctx, &obs_dim, &act_dim, &units);
    return create(ctx, nullptr, model, obs_dim, act_dim, units, m, n, h, low, high, discount, reward, np_state_addr, rng_threads, out);
}

int l2a_controller_create_device(l2a_model* model, int m, int n, int h, const double* low, const double* high, double discount,
                                 const l2a_reward* reward, unsigned long long seed, l2a_controller** out) {
    if (!model) return L2A_EINVAL;
    l2a_ctx* ctx = nullptr;
    int obs_dim = 0, act_dim = 0;
    l2a_model_facts(model, &ctx, &obs_dim, &act_dim);
    return create(ctx, model, nullptr, obs_dim, act_dim, 0, m, n, h, low, high, discount, reward, nullptr, 1, out, true, seed);
}

int l2a_lstm_controller_create_device(l2a_lstm* model, int m, int n, int h, const double* low, const double* high, double discount,
                                      const l2a_reward* reward, unsigned long long seed, l2a_controller** out) {
    if (!model) return L2A_EINVAL;
    l2a_ctx* ctx = nullptr;
    int obs_dim = 0, act_dim = 0, units = 0;
    l2a_lstm_facts(model, &ctx, &obs_dim, &act_dim, &units);
    return create(ctx, nullptr, model, obs_dim, act_dim, units, m, n, h, low, high, discount, reward, nullptr, 1, out, true, seed);
}

void l2a_controller_destroy(l2a_controller* c) {
    if (!c) return;
    if (c->chain) l2a_ahead_destroy(c->chain);                // joins the producer: no upload is in flight afterwards
    l2a_device_guard guard(c->ctx->device);
    if (c->in_flight) (void)hipDeviceSynchronize();           // a step begun and never finished: its launch still reads these buffers
    for (int s = 0; s < 2; ++s) {
        if (c->pin[s]) (void)hipHostFree(c->pin[s]);
        if (c->dev[s]) (void)hipFree(c->dev[s]);
        std::free(c->c64[s]);
    }
    if (c->lowr_dev) (void)hipFree(c->lowr_dev);
    if (c->obs_map_host) (void)hipHostFree(c->obs_map_host);
    if (c->keys_dev) (void)hipFree(c->keys_dev);
    if (c->payload_dev) (void)hipFree(c->payload_dev);
    if (c->payload_host) (void)hipHostFree(c->payload_host);
    if (c->payload_ev) (void)hipEventDestroy(c->payload_ev);
    if (c->side) (void)hipStreamDestroy(c->side);
    delete c;
}

int l2a_controller_step(l2a_controller* c, const double* obs, double* action_out, long long* index_out, float* return_out,
                        void* stream) {
    if (!c) return L2A_EINVAL;
    if (!c->mlp) return fail(c->ctx, L2A_EINVAL, "l2a_controller_step: this controller plans with a recurrent model (l2a_lstm_controller_step)");
    return step(c, obs, nullptr, nullptr, nullptr, nullptr, action_out, index_out, return_out, stream);
}

int l2a_lstm_controller_step(l2a_controller* c, const double* obs, const float* c0, const float* h0, float* c_next, float* h_next,
                             double* action_out, long long* index_out, float* return_out, void* stream) {
    if (!c) return L2A_EINVAL;
    if (!c->rnn) return fail(c->ctx, L2A_EINVAL, "l2a_lstm_controller_step: this controller plans with an MLP model (l2a_controller_step)");
    if (!c0 || !h0) return fail(c->ctx, L2A_EINVAL, "l2a_lstm_controller_step: null c0 / h0");
    return step(c, obs, c0, h0, c_next, h_next, action_out, index_out, return_out, stream);
}

int l2a_controller_begin(l2a_controller* c, const double* obs, void* stream) {
    if (!c) return L2A_EINVAL;
    if (!c->mlp) return fail(c->ctx, L2A_EINVAL, "l2a_controller_begin: this controller plans with a recurrent model (l2a_lstm_controller_begin)");
    return begin(c, obs, nullptr, nullptr, nullptr, nullptr, stream);
}

int l2a_lstm_controller_begin(l2a_controller* c, const double* obs, const float* c0, const float* h0, float* c_next, float* h_next,
                              void* stream) {
    if (!c) return L2A_EINVAL;
    if (!c->rnn) return fail(c->ctx, L2A_EINVAL, "l2a_lstm_controller_begin: this controller plans with an MLP model (l2a_controller_begin)");
    if (!c0 || !h0) return fail(c->ctx, L2A_EINVAL, "l2a_lstm_controller_begin: null c0 / h0");
    return begin(c, obs, c0, h0, c_next, h_next, stream);
}

int l2a_controller_finish(l2a_controller* c, double* action_out, long long* index_out, float* return_out) {
    if (!c) return L2A_EINVAL;
    return finish(c, action_out, index_out, return_out);
}

int l2a_controller_rearm(l2a_controller* c) {
    if (!c) return L2A_EINVAL;
    if (c->device_rng) return L2A_OK;
    if (l2a_ahead_arm(c->chain, c->np_addr) != 0) return fail(c->ctx, L2A_ESTATE, "l2a_controller_rearm: the producer thread could not be started");
    return L2A_OK;
}

const float* l2a_controller_actions(const l2a_controller* c) {
    return (c && c->slot >= 0) ? c->dev[c->slot] : nullptr;
}

int l2a_controller_stats(l2a_controller* c, double* out, int cap) {
    if (!c || !out || cap < 1) return L2A_EINVAL;
    double v[16] = {0};
    for (int i = 0; i < 7; ++i) v[i] = c->stage_us[i];
    double ch[6] = {0, 0, 0, 0, 0, 0};
    if (c->chain) l2a_ahead_stats(c->chain, ch);
    v[7] = (double)c->steps; v[8] = (double)c->relaunches;
    for (int i = 0; i < 6; ++i) v[9 + i] = ch[i];
    v[15] = (double)c->sync_draws;
    for (int i = 0; i < cap && i < 16; ++i) out[i] = v[i];
    return L2A_OK;
}

}  // extern "C"
