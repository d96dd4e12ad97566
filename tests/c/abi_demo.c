/*
 * abi_demo.c - libl2a_hip.so driven from plain C (no Python, no torch): the binding a non-Python host of
 * the reference's planner would write against include/l2a.h.  Test infrastructure (tests/test_c_abi.py
 * compiles it with gcc and runs it on the GPU box).
 *
 * usage: abi_demo <case.bin> [rank world idfile] ; without the three extra arguments one process plans the whole
 * case.  With them, `world` processes (one per GPU: device = rank % visible devices) each plan the candidate
 * shard [rank * n / world, (rank + 1) * n / world) with global indices (cand_offset) and combine their keys with
 * l2a_allreduce_best (RCCL): rank 0 creates the communicator id and writes it to `idfile`, the others wait for it.
 * Every rank prints the same lines.  The case file is written by the test:
 *   int32 header[8] = {obs_dim, act_dim, n_hidden, hidden0, m, n, h, vel_index}
 *   float  dt, ctrl_coef
 *   per layer: kernel [in, out] fp32, bias [out] fp32          (reference parameter order)
 *   double norm[6][...]: mean_obs, std_obs, mean_act, std_act, mean_delta, std_delta
 *   float  obs0[m * obs_dim], actions[h * m * n * act_dim]
 * prints: one line per env "env <i> index <idx> return <ret>"
 *
 * usage: abi_demo <case.bin> ctrl <seed> <steps> <state.bin> - the controller step as ONE call, from plain C: the program owns
 * an MT19937 state in NumPy's layout ({uint32 key[624]; int pos;}), seeds it the way `np.random.seed(seed)` does
 * (init_genrand, pos = 624), and calls l2a_controller_create / _step (x steps) / _stats / _rearm / _destroy on the case's
 * observations.  Prints per step and env "step <s> env <i> index <idx> return <ret> action <act_dim hex doubles>" and
 * writes the generator state it is left with to <state.bin> (the test loads it into a RandomState and draws from it).
 */
#define _DEFAULT_SOURCE   /* usleep */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "l2a.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_L2A(ctx, x) do { int rc_ = (x); if (rc_ != L2A_OK) { \
    fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, l2a_last_error(ctx)); return 3; } } while (0)

static void* rd(FILE* f, size_t bytes) {
    void* p = malloc(bytes ? bytes : 1);
    if (fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "short read\n"); exit(4); }
    return p;
}

static void* to_device(const void* host, size_t bytes) {
    void* d = NULL;
    if (hipMalloc(&d, bytes) != hipSuccess || hipMemcpy(d, host, bytes, hipMemcpyHostToDevice) != hipSuccess) {
        fprintf(stderr, "device upload failed\n");
        exit(5);
    }
    return d;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s case.bin\n", argv[0]); return 1; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    int* hd = (int*)rd(f, 8 * sizeof(int));
    const int obs_dim = hd[0], act_dim = hd[1], n_hidden = hd[2], width = hd[3], m = hd[4], n = hd[5], h = hd[6];
    float* rw_par = (float*)rd(f, 2 * sizeof(float));

    const int ctrl_mode = (argc >= 6 && strcmp(argv[2], "ctrl") == 0);
    const int sharded = (argc >= 5) && !ctrl_mode;
    const int rank = sharded ? atoi(argv[2]) : 0, world = sharded ? atoi(argv[3]) : 1;
    int n_dev = 1;
    CHECK_HIP(hipGetDeviceCount(&n_dev));
    l2a_ctx* ctx = NULL;
    if (l2a_init(rank % n_dev, &ctx) != L2A_OK) { fprintf(stderr, "l2a_init: %s\n", l2a_last_error(NULL)); return 3; }
    int hidden[8];
    for (int i = 0; i < n_hidden; ++i) hidden[i] = width;
    l2a_model* model = NULL;
    CHECK_L2A(ctx, l2a_model_create(ctx, obs_dim, act_dim, n_hidden, hidden, L2A_ACT_RELU, L2A_ACT_IDENTITY, 1,
                                    L2A_MODE_SINGLE, &model));

    const void* ptrs[2 * 9];
    int k_in = obs_dim + act_dim;
    for (int l = 0; l <= n_hidden; ++l) {
        const int n_out = (l < n_hidden) ? width : obs_dim;
        float* w = (float*)rd(f, sizeof(float) * (size_t)k_in * n_out);
        float* b = (float*)rd(f, sizeof(float) * (size_t)n_out);
        ptrs[2 * l] = to_device(w, sizeof(float) * (size_t)k_in * n_out);
        ptrs[2 * l + 1] = to_device(b, sizeof(float) * (size_t)n_out);
        free(w); free(b);
        k_in = n_out;
    }
    CHECK_L2A(ctx, l2a_model_set_weights(model, 0, ptrs, NULL));
    double* nm[6];
    for (int i = 0; i < 6; ++i) nm[i] = (double*)rd(f, sizeof(double) * (size_t)((i == 2 || i == 3) ? act_dim : obs_dim));
    CHECK_L2A(ctx, l2a_model_set_norm(model, 0, nm[0], nm[1], nm[2], nm[3], nm[4], nm[5], NULL));

    float* obs0 = (float*)rd(f, sizeof(float) * (size_t)m * obs_dim);
    float* acts = (float*)rd(f, sizeof(float) * (size_t)h * m * n * act_dim);
    fclose(f);
    float* d_obs0 = (float*)to_device(obs0, sizeof(float) * (size_t)m * obs_dim);
    unsigned long long* d_key = NULL;
    CHECK_HIP(hipMalloc((void**)&d_key, sizeof(unsigned long long) * (size_t)m));

    /* this rank's shard of the candidate tensor [h, m * n, act_dim] -> [h, m * n_loc, act_dim] */
    const int lo = (int)((long long)rank * n / world), hi = (int)((long long)(rank + 1) * n / world), n_loc = hi - lo;
    float* shard = (float*)malloc(sizeof(float) * (size_t)h * m * (n_loc > 0 ? n_loc : 1) * act_dim);
    for (int t = 0; t < h; ++t)
        for (int i = 0; i < m; ++i)
            memcpy(shard + ((size_t)(t * m + i) * n_loc) * act_dim, acts + ((size_t)(t * m + i) * n + lo) * act_dim,
                   sizeof(float) * (size_t)n_loc * act_dim);
    float* d_acts = (float*)to_device(shard, sizeof(float) * (size_t)h * m * (n_loc > 0 ? n_loc : 1) * act_dim);

    if (sharded) {
        char id[128];
        if (rank == 0) {
            CHECK_L2A(ctx, l2a_comm_unique_id(id));
            char tmp[1024];
            snprintf(tmp, sizeof(tmp), "%s.tmp", argv[4]);
            FILE* g = fopen(tmp, "wb");
            if (!g || fwrite(id, 1, 128, g) != 128) { perror("idfile"); return 7; }
            fclose(g);
            rename(tmp, argv[4]);
        } else {
            FILE* g = NULL;
            for (int tries = 0; tries < 600 && !(g = fopen(argv[4], "rb")); ++tries) usleep(100000);
            if (!g || fread(id, 1, 128, g) != 128) { fprintf(stderr, "no communicator id in %s\n", argv[4]); return 7; }
            fclose(g);
        }
        CHECK_L2A(ctx, l2a_comm_init(ctx, rank, world, id));
    }

    l2a_reward rw = {1.0f, 1.0f / rw_par[0], 0.0f, rw_par[1], 0.0f, hd[7], 0, 0};      /* half_cheetah_env.py:58-65 */
    if (ctrl_mode) {
        /* the reference's call sequence from a C host: np.random.seed(seed); for s in steps: policy.get_actions(obs) */
        struct { unsigned int key[624]; int pos; } mt;                  /* numpy/random/src/mt19937/mt19937.h: mt19937_state */
        mt.key[0] = (unsigned int)strtoul(argv[3], NULL, 10);
        for (int i = 1; i < 624; ++i) mt.key[i] = 1812433253u * (mt.key[i - 1] ^ (mt.key[i - 1] >> 30)) + (unsigned int)i;
        mt.pos = 624;
        const int steps = atoi(argv[4]);
        double low[16], high[16];
        for (int k = 0; k < act_dim; ++k) { low[k] = -1.0; high[k] = 1.0; }     /* half_cheetah_env.py:40 */
        double* obs64 = (double*)malloc(sizeof(double) * (size_t)m * obs_dim);
        for (int i = 0; i < m * obs_dim; ++i) obs64[i] = (double)obs0[i];
        l2a_controller* c = NULL;
        CHECK_L2A(ctx, l2a_controller_create(model, m, n, h, low, high, 1.0, &rw, &mt, 2, &c));
        double* act = (double*)malloc(sizeof(double) * (size_t)m * act_dim);
        long long* idx = (long long*)malloc(sizeof(long long) * (size_t)m);
        float* ret = (float*)malloc(sizeof(float) * (size_t)m);
        for (int s = 0; s < steps; ++s) {
            const int rc = l2a_controller_step(c, obs64, act, idx, ret, NULL);
            if (rc < 0 || rc == L2A_STEP_MISS) { fprintf(stderr, "l2a_controller_step -> %d: %s\n", rc, l2a_last_error(ctx)); return 3; }
            for (int i = 0; i < m; ++i) {
                printf("step %d env %d index %lld return %.9g action", s, i, idx[i], ret[i]);
                for (int k = 0; k < act_dim; ++k) printf(" %a", act[i * act_dim + k]);
                printf(" rc %d\n", rc);
            }
            if (s == steps / 2) CHECK_L2A(ctx, l2a_controller_rearm(c));       /* legal at any time: restart the chain here */
        }
        double st[16];
        CHECK_L2A(ctx, l2a_controller_stats(c, st, 16));
        printf("stats steps %.0f hits %.0f sync_draws %.0f\n", st[7], st[9], st[15]);
        l2a_controller_destroy(c);
        FILE* g = fopen(argv[5], "wb");
        if (!g || fwrite(&mt, sizeof(mt), 1, g) != 1) { perror("state file"); return 7; }
        fclose(g);
        l2a_model_destroy(model);
        l2a_destroy(ctx);
        return 0;
    }
    if (n_loc > 0) CHECK_L2A(ctx, l2a_plan_rs(model, d_obs0, d_acts, m, n_loc, h, 1.0, &rw, lo, NULL, d_key, NULL));
    else CHECK_HIP(hipMemset(d_key, 0, sizeof(unsigned long long) * (size_t)m));            /* neutral key */
    if (sharded) CHECK_L2A(ctx, l2a_allreduce_best(ctx, d_key, m, NULL));
    unsigned long long* key = (unsigned long long*)malloc(sizeof(unsigned long long) * (size_t)m);
    CHECK_HIP(hipMemcpy(key, d_key, sizeof(unsigned long long) * (size_t)m, hipMemcpyDeviceToHost));   /* = the sync */
    int status = 0;
    CHECK_L2A(ctx, l2a_launch_status(ctx, &status));
    if (status != 0) { fprintf(stderr, "launch status 0x%x\n", status); return 6; }
    for (int i = 0; i < m; ++i) {
        float ret; int idx;
        l2a_key_decode(key[i], &ret, &idx);
        printf("env %d index %d return %.9g\n", i, idx, ret);
    }
    l2a_model_destroy(model);
    l2a_destroy(ctx);
    return 0;
}
