// l2a_host.h - host-side definitions shared by the translation units of libl2a_hip.so
// (l2a_api.hip: MLP models; l2a_lstm.hip: recurrent models).  Not part of the C ABI.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <ctime>
#include <string>

#include "../../include/l2a.h"

#define L2A_MAIL_KEYS 64
#define L2A_MAIL_OBS 4096
struct l2a_mail {
    unsigned long long seq;                     // last published launch
    unsigned long long keys[L2A_MAIL_KEYS];     // its arg-max keys
    float obs[2][L2A_MAIL_OBS];                 // observation staging, slot = launch parity (read by the kernel)
};

struct l2a_ctx {
    int device = 0;
    int kernel_kind = L2A_KERNEL_AUTO;
    int split_policy = 1;                 // 1 = split members over two workgroups when it fills the chip
    int xcd_align = 1;                    // uniform tile split: pad the grid so that each ensemble group owns four XCDs (L2A_XCD_ALIGN=0: off)
    int fan_policy = 1;                   // member fan (one workgroup per candidate tile and ensemble member): 0 = never, 1 = where E x tiles fit the chip (L2A_FAN=0: off)
    int double_policy = 1;                // double rounds (two candidate tiles per workgroup on the whole-tiles-only instances in front of a multi-round plan at width 512): 0 = never (L2A_DOUBLE=0: off)
    int batch_sets = 0;                   // MFMA rollout: sets per batch; 0 = as many as fit the LDS, 1 = one at a time
    int micro_policy = 1;                 // micro-tile kernels (l2a_micro.h): 0 = never, 1 = where they fill the chip better, 2 = whenever a plan is eligible
    unsigned int* status_host = nullptr;  // pinned, device-visible launch status word
    unsigned int* status_dev = nullptr;
    unsigned long long* dbg = nullptr;    // optional timeline buffer (l2a_set_debug_buffer)
    int* dry = nullptr;                   // l2a_plan_geometry: the launcher records its decisions here and launches nothing
    unsigned int spin_limit = 1u << 18;   // exchange polls per workgroup and launch before it gives up (~0.5 s)
    // result mailbox of l2a_plan_rs_sync (allocated on first use)
    struct l2a_mail* mail_host = nullptr; // host-mapped
    struct l2a_mail* mail_dev = nullptr;  // device alias of mail_host
    unsigned int* done_ctr = nullptr;     // device
    unsigned long long* key_ring = nullptr;   // device [2][L2A_MAIL_KEYS]
    int ring_dirty[2] = {L2A_MAIL_KEYS, L2A_MAIL_KEYS};   // leading entries of each slot that may be non-zero
    unsigned long long mail_seq = 0;
    double sync_ema_us = 0.0;             // expected duration of the next blocking plan (sleep-then-spin)
    double stamps_us[6] = {0, 0, 0, 0, 0, 0};   // latest blocking plan, host clock: entry | staged | launched | hook done | keys seen | (free)
    unsigned long long sync_shape = 0;    // plan shape the estimate belongs to (l2a_mail_ticket::shape)
    // RCCL communicator of sharded plans (l2a_comm.hip); null = single GPU
    void* comm = nullptr;
    int comm_rank = 0, comm_world = 0;
    int num_cu = 0;
    int lds_per_block = 0;
    int clock_khz = 0;
    std::string arch;
    std::string name;
    mutable std::string err;
};

// Records `msg` on the context (or as the init error when ctx is null) and returns `code`.
int l2a_fail(const l2a_ctx* ctx, int code, const std::string& msg);

#define L2A_HIP(ctx, call)                                                                        \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return l2a_fail((ctx), L2A_EHIP, std::string(#call) + ": " + hipGetErrorString(e_));  \
    } while (0)

inline int l2a_ceil_div(int a, int b) { return (a + b - 1) / b; }

// Exchange granules of the tile splits.  Experiment switch L2A_XBUF_MODE: 0 / unset = hipMalloc (coarse-grained, the kernels'
// sc1 stores and loads make the granules visible across XCDs), 1 = fine-grained, 3 = uncached device memory.
inline hipError_t l2a_xbuf_alloc(void** ptr, size_t bytes) {
    static int mode = -1;
    if (mode < 0) {
        const char* e = std::getenv("L2A_XBUF_MODE");
        mode = (e && (e[0] == '1' || e[0] == '3')) ? e[0] - '0' : 0;
    }
    if (mode == 0) return hipMalloc(ptr, bytes);
    return hipExtMallocWithFlags(ptr, bytes, mode == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached);
}

// ---- blocking launches through the result mailbox (l2a_plan_rs_sync, l2a_lstm_plan_rs_sync; defined in l2a_api.hip) ----
struct l2a_mail_ticket {
    unsigned long long seq = 0;                 // value the kernel publishes
    int slot = 0;                               // observation staging / key slot of this launch
    const float* obs_dev = nullptr;             // device alias of the staged observations
    unsigned long long* keys_dev = nullptr;     // this launch's key slot (zeroed)
    unsigned long long* next_keys = nullptr;    // the next launch's key slot (the last tile zeroes [0, m))
    double t0_us = 0.0;                         // host clock when the ticket was drawn
    unsigned long long shape = 0;               // caller's digest of (model, m, n, h): keys the wait-time estimate
};
// Stage `obs_floats` observation floats, pick and clear the key slot.  Kernel parameters of a publishing launch:
// best_key = keys_dev, done_ctr = ctx->done_ctr, mail_keys = ctx->mail_dev->keys, mail_seq_ptr = &ctx->mail_dev->seq,
// mail_seq = seq, next_keys.
#define L2A_HIDDEN __attribute__((visibility("hidden")))        // internal to libl2a_hip.so: not part of the C ABI, not exported
extern "C" L2A_HIDDEN int l2a_mail_begin(l2a_ctx* ctx, int m, const float* obs_host, long long obs_floats, hipStream_t stream,
                                         l2a_mail_ticket* ticket);
// After the launch (`launch_rc` = its return code): wait for the keys - mailbox word when the kernel publishes, copy +
// stream synchronisation otherwise - and check the status word (L2A_ESPLIT).
extern "C" L2A_HIDDEN int l2a_mail_end(l2a_ctx* ctx, const l2a_mail_ticket& ticket, int m, bool published, int launch_rc,
                                       hipStream_t stream, unsigned long long* keys_host_out, const char* who);

// ---- the blocking plans with a hook between launch and wait (l2a_step.hip: the controller step kicks the producer of the NEXT
//      step's candidates there - after the launch is on its way, before the host starts waiting) ----
typedef void (*l2a_after_launch_fn)(void* arg);
// With `pending` the hook variants return right after the hook - launch on its way, nothing waited for - and the caller ends
// the plan later with l2a_plan_finish (l2a_controller_begin / _finish).
struct l2a_mail_pending {
    l2a_mail_ticket tk;
    bool publish = false;
    int m = 0;
    hipStream_t stream = nullptr;
    const char* who = "";
    bool live = false;
};
extern "C" L2A_HIDDEN int l2a_plan_finish(l2a_ctx* ctx, l2a_mail_pending* pending, unsigned long long* keys_host_out);
extern "C" L2A_HIDDEN int l2a_plan_rs_sync_hook(l2a_model* md, const float* obs_host, const float* actions, int m, int n, int h,
                                                double discount, const l2a_reward* reward, int cand_offset, float* returns_out,
                                                unsigned long long* keys_host_out, void* stream, l2a_after_launch_fn hook,
                                                void* hook_arg, l2a_mail_pending* pending = nullptr);
extern "C" L2A_HIDDEN int l2a_lstm_plan_rs_sync_hook(l2a_lstm* md, const float* obs_host, const float* c0, const float* h0,
                                                     const float* actions, int m, int n, int h, double discount,
                                                     const l2a_reward* reward, int cand_offset, unsigned long long* keys_host_out,
                                                     float* c_next, float* h_next, void* stream, l2a_after_launch_fn hook,
                                                     void* hook_arg, l2a_mail_pending* pending = nullptr);
extern "C" L2A_HIDDEN void l2a_model_facts(const l2a_model* md, l2a_ctx** ctx, int* obs_dim, int* act_dim);
extern "C" L2A_HIDDEN void l2a_lstm_facts(const l2a_lstm* md, l2a_ctx** ctx, int* obs_dim, int* act_dim, int* units);
inline double l2a_now_us() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec * 1e6 + (double)ts.tv_nsec * 1e-3;
}

// Makes the context's device current for the duration of an entry point and restores the caller's (a process
// driving several GPUs, or a torch thread whose current device differs from the model's).
struct l2a_device_guard {
    int prev = -1;
    bool switched = false;
    explicit l2a_device_guard(int device) {
        if (hipGetDevice(&prev) == hipSuccess && prev != device) switched = (hipSetDevice(device) == hipSuccess);
    }
    ~l2a_device_guard() { if (switched) (void)hipSetDevice(prev); }
};

