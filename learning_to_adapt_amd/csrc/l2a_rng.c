/*
 * l2a_rng.c - host helper (plain C, no GPU): NumPy's legacy global generator, faster and on several threads.
 *
 * Parity mode draws its candidate actions from `np.random` exactly as the reference does:
 *   random shooting  np.random.uniform(low, high, (h*n*m, act_dim))        policies/mpc_controller.py:67-69,114
 *   CEM              np.random.normal(size=(n, m, h*act_dim))              policies/mpc_controller.py:85
 * Both consume the MT19937 word stream of the legacy `RandomState` (Matsumoto & Nishimura's reference
 * generator, numpy/random/src/mt19937/mt19937.c):
 *   random_sample :  (a * 67108864 + b) / 9007199254740992,  a = next32() >> 5,  b = next32() >> 6
 *   uniform       :  low + (high - low) * random_sample           (two roundings, no FMA)
 *   legacy_gauss  :  polar method on pairs x = 2 * random_sample - 1, r2 = x1*x1 + x2*x2, rejected when
 *                    r2 >= 1 or r2 == 0; f = sqrt(-2 log(r2) / r2); returns f*x2, caches f*x1 for the next call
 *                    (numpy/random/src/legacy/legacy-distributions.c)
 * This file restates them so that
 *   - the state-update loops and the tempering vectorise (dependency distances 227 / 397 words),
 *   - T threads generate DISJOINT SLICES OF THE SAME STREAM: a slice's start state is reached by fast-forwarding
 *     a copy of the state (state updates only, no tempering / conversion: ~0.1 ns per word, an order of magnitude
 *     cheaper than producing the numbers), or, for long distances, by `l2a_mt19937_jump` (polynomial jump-ahead),
 *   - the Gaussian's rejection loop becomes a data-parallel compaction: every attempt consumes exactly one pair of
 *     doubles, so attempt k always reads doubles 2k, 2k+1 of the stream whatever happened to attempts < k,
 *   - the float64 -> fp32 cast, the rank's candidate slice and the [h, rows, act] transposition of the CEM
 *     samples are fused into the draw (the planner's pinned staging buffer is written directly).
 * The Python side (`learning_to_adapt_amd/utils/fast_rng.py`) moves the state out of `np.random.get_state()`, calls
 * in here and puts the advanced state back; it trusts each entry point only after it has reproduced NumPy's own
 * call on the running machine (libm's `log` is the same shared object NumPy calls).
 *
 * Built with -ffp-contract=off: NumPy's baseline build has no FMA, a fused x1*x1 + x2*x2 would change bits.
 */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#define MT_N 624
#define MT_M 397
#define MT_A 0x9908b0dfU
#define MT_UP 0x80000000U
#define MT_LO 0x7fffffffU

/* runtime dispatch: the AVX2 / AVX-512 clone is picked on CPUs that have it (the library is built on one machine
 * and runs on another, so no -march=native) */
#define L2A_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))

typedef struct { uint32_t key[MT_N]; int pos; } mt_t;

L2A_CLONES static void mt_regen(uint32_t* mt) {
    int kk;
#pragma GCC ivdep
    for (kk = 0; kk < MT_N - MT_M; ++kk) {
        const uint32_t y = (mt[kk] & MT_UP) | (mt[kk + 1] & MT_LO);
        mt[kk] = mt[kk + MT_M] ^ (y >> 1) ^ ((uint32_t)(-(int32_t)(y & 1U)) & MT_A);
    }
#pragma GCC ivdep
    for (kk = MT_N - MT_M; kk < MT_N - 1; ++kk) {
        const uint32_t y = (mt[kk] & MT_UP) | (mt[kk + 1] & MT_LO);
        mt[kk] = mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ ((uint32_t)(-(int32_t)(y & 1U)) & MT_A);
    }
    {
        const uint32_t y = (mt[MT_N - 1] & MT_UP) | (mt[0] & MT_LO);
        mt[MT_N - 1] = mt[MT_M - 1] ^ (y >> 1) ^ ((uint32_t)(-(int32_t)(y & 1U)) & MT_A);
    }
}

static inline uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680U;
    y ^= (y << 15) & 0xefc60000U;
    y ^= (y >> 18);
    return y;
}


/* ---- polynomial jump-ahead ------------------------------------------------------------------------------------
 * The state s_k = (top bit of z_k, z_{k+1}, ..., z_{k+623}) of the raw word sequence z evolves linearly over GF(2):
 * s_{k+1} = F s_k, with minimal polynomial phi(t) of degree 19937.  With g(t) = t^n mod phi(t),
 * s_n = g(F) s_0 = sum_i g_i s_i, i.e. word-wise z_{n+j} = XOR over the set bits i of g of z_{i+j}: an XOR of at most
 * 19937 windows of the next 19937 + 624 raw words - O(1) in n, against O(n) for regenerating the blocks in between
 * (Haramoto, Matsumoto, Nishimura, Panneton, L'Ecuyer, "Efficient jump ahead for F2-linear random number generators",
 * 2008 - the sliding-window form).  phi is found once per process by Berlekamp-Massey on 2 * 19937 output bits (its
 * degree is checked); g is cached per distance (the planner's slice offsets repeat every controller step).
 * mt_skip uses the jump beyond L2A_JUMP_MIN_WORDS, where it beats block regeneration (~0.14 ns per word). */
#define MT_DEG 19937
#define PW ((MT_DEG + 64) / 64)                 /* 64-bit words of a polynomial of degree <= MT_DEG: 312 */
#define L2A_JUMP_MIN_WORDS (3LL << 20)

typedef struct { uint64_t w[PW]; } poly_t;
static poly_t g_phi;                            /* minimal polynomial, bit i = coefficient of t^i */
static int g_phi_ready = 0;
static pthread_mutex_t g_jump_mu = PTHREAD_MUTEX_INITIALIZER;
#define JUMP_CACHE 64
static struct { long long n; poly_t g; } g_jump_cache[JUMP_CACHE];
static int g_jump_cached = 0;

static inline int pbit(const uint64_t* p, int i) { return (int)((p[i >> 6] >> (i & 63)) & 1u); }

/* Berlekamp-Massey over GF(2): minimal polynomial of the bit sequence seq[0..n). */
static int berlekamp_massey(const uint8_t* seq, int n, uint64_t* out, int words) {
    uint64_t* C = (uint64_t*)calloc((size_t)words, 8);
    uint64_t* B = (uint64_t*)calloc((size_t)words, 8);
    uint64_t* T = (uint64_t*)calloc((size_t)words, 8);
    uint64_t* S = (uint64_t*)calloc((size_t)words, 8);         /* reversed sliding window of the sequence */
    if (!C || !B || !T || !S) { free(C); free(B); free(T); free(S); return -1; }
    C[0] = B[0] = 1;
    int L = 0, m = 1;
    for (int N = 0; N < n; ++N) {
        /* S = bits s_N, s_{N-1}, ... at positions 0, 1, ...: shift left by one, insert s_N */
        uint64_t carry = seq[N];
        for (int k = 0; k < words; ++k) { const uint64_t nc = S[k] >> 63; S[k] = (S[k] << 1) | carry; carry = nc; }
        /* discrepancy d = sum_{i=0..L} C_i s_{N-i} */
        uint64_t acc = 0;
        const int lw = (L >> 6) + 1;
        for (int k = 0; k < lw && k < words; ++k) acc ^= C[k] & S[k];
        const int d = __builtin_parityll(acc);
        if (!d) { ++m; continue; }
        const int grow = (2 * L <= N);
        if (grow) memcpy(T, C, (size_t)words * 8);
        /* C ^= B << m */
        const int ws = m >> 6, bs = m & 63;
        for (int k = words - 1; k >= ws; --k) {
            uint64_t v = B[k - ws] << bs;
            if (bs && k - ws - 1 >= 0) v |= B[k - ws - 1] >> (64 - bs);
            C[k] ^= v;
        }
        if (grow) { L = N + 1 - L; memcpy(B, T, (size_t)words * 8); m = 1; } else { ++m; }
    }
    /* connection polynomial C(x) = 1 + c_1 x + ... + c_L x^L  <->  minimal polynomial t^L + c_1 t^(L-1) + ... + c_L */
    memset(out, 0, (size_t)words * 8);
    for (int i = 0; i <= L; ++i)
        if (pbit(C, i)) out[(L - i) >> 6] |= 1ull << ((L - i) & 63);
    free(C); free(B); free(T); free(S);
    return L;
}

static int phi_init(void) {
    if (g_phi_ready) return 0;
    mt_t s;
    uint32_t x = 19650218u;                                     /* any non-degenerate state will do */
    for (int k = 0; k < MT_N; ++k) { s.key[k] = x; x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)(k + 1); }
    s.pos = MT_N;
    const int n = 2 * MT_DEG + 64;
    uint8_t* bits = (uint8_t*)malloc((size_t)n);
    if (!bits) return -1;
    for (int k = 0; k < n; ++k) {                               /* bit 0 of every raw word: a linear output */
        if (s.pos >= MT_N) { mt_regen(s.key); s.pos = 0; }
        bits[k] = (uint8_t)(s.key[s.pos++] & 1u);
    }
    uint64_t tmp[2 * PW + 4];
    const int L = berlekamp_massey(bits, n, tmp, 2 * PW + 4);
    free(bits);
    if (L != MT_DEG) return -2;
    memcpy(g_phi.w, tmp, sizeof(g_phi.w));
    g_phi_ready = 1;
    return 0;
}

/* r = (a * b) mod phi, degrees < MT_DEG */
static void poly_mulmod(const poly_t* a, const poly_t* b, poly_t* r) {
    uint64_t prod[2 * PW + 1];
    memset(prod, 0, sizeof(prod));
    for (int i = 0; i < MT_DEG; ++i) {
        if (!pbit(a->w, i)) continue;
        const int ws = i >> 6, bs = i & 63;
        for (int k = 0; k < PW; ++k) {
            prod[k + ws] ^= b->w[k] << bs;
            if (bs) prod[k + ws + 1] ^= b->w[k] >> (64 - bs);
        }
    }
    for (int d = 2 * MT_DEG - 2; d >= MT_DEG; --d) {            /* reduce: t^d = t^(d - DEG) * (phi - t^DEG) */
        if (!pbit(prod, d)) continue;
        const int sh = d - MT_DEG, ws = sh >> 6, bs = sh & 63;
        for (int k = 0; k < PW; ++k) {
            prod[k + ws] ^= g_phi.w[k] << bs;
            if (bs) prod[k + ws + 1] ^= g_phi.w[k] >> (64 - bs);
        }
    }
    memcpy(r->w, prod, sizeof(r->w));
    r->w[PW - 1] &= (1ull << (MT_DEG & 63)) - 1;                /* bits >= DEG are gone after the reduction */
}

/* g = t^n mod phi (cached) */
static int jump_poly(long long n, poly_t* out) {
    pthread_mutex_lock(&g_jump_mu);
    int rc = phi_init();
    if (rc == 0) {
        int hit = -1;
        for (int k = 0; k < g_jump_cached; ++k)
            if (g_jump_cache[k].n == n) { hit = k; break; }
        if (hit < 0) {
            poly_t result, base, tmp;
            memset(&result, 0, sizeof(result)); result.w[0] = 1;     /* 1 */
            memset(&base, 0, sizeof(base)); base.w[0] = 2;           /* t */
            for (long long e = n; e > 0; e >>= 1) {
                if (e & 1) { poly_mulmod(&result, &base, &tmp); result = tmp; }
                if (e > 1) { poly_mulmod(&base, &base, &tmp); base = tmp; }
            }
            hit = (g_jump_cached < JUMP_CACHE) ? g_jump_cached++ : (int)((unsigned long long)n % JUMP_CACHE);
            g_jump_cache[hit].n = n;
            g_jump_cache[hit].g = result;
        }
        *out = g_jump_cache[hit].g;
    }
    pthread_mutex_unlock(&g_jump_mu);
    return rc;
}

/* Replace the block s->key by the block `blocks` regenerations later (blocks >= 1), in O(1). */
L2A_CLONES static int mt_jump_blocks(mt_t* s, long long blocks) {
    poly_t g;
    const long long n = blocks * MT_N - 1;      /* s_{W-1}: (top bit of z_{W-1}, z_W .. z_{W+622}), W = 624 * blocks */
    if (jump_poly(n, &g) != 0) return -1;
    enum { RAW = MT_DEG + MT_N + MT_N };
    uint32_t* z = (uint32_t*)malloc(sizeof(uint32_t) * RAW);
    if (!z) return -1;
    memcpy(z, s->key, sizeof(s->key));
    for (int b = 1; (b + 1) * MT_N <= RAW; ++b) {               /* the next raw words, block by block */
        memcpy(z + b * MT_N, z + (b - 1) * MT_N, sizeof(s->key));
        mt_regen(z + b * MT_N);
    }
    uint32_t acc[MT_N];
    memset(acc, 0, sizeof(acc));
    for (int w = 0; w < PW; ++w) {
        uint64_t bitsw = g.w[w];
        while (bitsw) {
            const int i = (w << 6) + __builtin_ctzll(bitsw);
            bitsw &= bitsw - 1;
            const uint32_t* win = z + i;
            for (int j = 0; j < MT_N; ++j) acc[j] ^= win[j];
        }
    }
    free(z);
    /* acc[0] = z_{W-1} (only its top bit is meaningful), acc[1..623] = z_W .. z_{W+622}; one more recurrence step
     * gives z_{W+623} = z_{W+396} ^ twist(top(z_{W-1}) | low(z_W)) */
    const uint32_t y = (acc[0] & MT_UP) | (acc[1] & MT_LO);
    const uint32_t last = acc[MT_M] ^ (y >> 1) ^ ((uint32_t)(-(int32_t)(y & 1U)) & MT_A);
    memcpy(s->key, acc + 1, sizeof(uint32_t) * (MT_N - 1));
    s->key[MT_N - 1] = last;
    return 0;
}

/* Advance the state by `words` 32-bit outputs without producing them. */
static void mt_skip(mt_t* s, long long words) {
    if (words >= L2A_JUMP_MIN_WORDS && s->pos <= MT_N) {
        /* whole blocks by the polynomial jump, the remainder (< 624 words) by position */
        long long target = (long long)s->pos + words;       /* index in the z sequence of the current block */
        if (s->pos == MT_N) { mt_regen(s->key); s->pos = 0; target -= MT_N; }
        const long long blocks = target / MT_N;
        if (blocks >= 1) {
            mt_t saved = *s;
            if (mt_jump_blocks(s, blocks) == 0) {
                s->pos = (int)(target - blocks * MT_N);
                return;
            }
            *s = saved;                                      /* (allocation failure: fall through to regeneration) */
        }
    }
    while (words > 0) {
        if (s->pos >= MT_N) { mt_regen(s->key); s->pos = 0; }
        long long take = MT_N - s->pos;
        if (take > words) take = words;
        s->pos += (int)take;
        words -= take;
    }
}

/* The next n doubles of the stream. */
L2A_CLONES static void mt_fill_double(mt_t* s, double* out, long long n) {
    long long i = 0;
    int p = s->pos;
    uint32_t* key = s->key;
    uint32_t t[MT_N];
    while (i < n) {
        if (p >= MT_N) { mt_regen(key); p = 0; }
        /* temper the rest of this block at once, then pair the words up */
        const int avail = MT_N - p;
        int k;
        for (k = 0; k < avail; ++k) t[k] = mt_temper(key[p + k]);
        int pairs = avail / 2;
        if ((long long)pairs > n - i) pairs = (int)(n - i);
        for (k = 0; k < pairs; ++k) {
            const uint32_t a = t[2 * k] >> 5, b = t[2 * k + 1] >> 6;
            out[i + k] = ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
        }
        i += pairs;
        p += 2 * pairs;
        if (i < n && MT_N - p == 1) {      /* a pair straddles the block boundary */
            const uint32_t a = t[avail - 1] >> 5;
            mt_regen(key);
            const uint32_t b = mt_temper(key[0]) >> 6;
            out[i++] = ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
            p = 1;
        }
    }
    s->pos = p;
}

/* ---- a small persistent thread pool ------------------------------------------------------------------------
 * Jobs are fork-join; the caller runs slice 0.  Workers sleep on a condition variable between jobs.  If the
 * process forked after the pool was created (the reference forks its env workers, samplers/sampler.py:37), the
 * child has no worker threads: the pid check rebuilds the pool there. */
#define L2A_MAXT 32
typedef void (*job_fn)(void* arg, int tid, int nthreads);
typedef struct l2a_pool {
    pthread_mutex_t mu;
    pthread_cond_t cv_start, cv_done;
    pthread_t th[L2A_MAXT];
    int n_workers;
    pid_t pid;
    job_fn fn;
    void* arg;
    int nthreads;       /* participants of the current job (incl. the caller) */
    unsigned gen;       /* job generation */
    int pending;
    unsigned seen0[L2A_MAXT];   /* generation each worker starts from (set by its creator) */
    pthread_mutex_t call_mu;    /* one parallel job at a time per pool */
} l2a_pool;
/* Two pools: the thread that first calls in (the controller's main thread: synchronous draws, l2a_cem_samples)
 * owns pool 0, every other thread (the draw-ahead worker) shares pool 1 - a long draw-ahead job must not make the
 * main thread's short passes queue behind it (measured: +4.7 ms per config-5 plan step with a single pool). */
static l2a_pool g_pools[2] = {
    {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, {0}, 0, 0, 0, 0, 0, 0, 0, {0}, PTHREAD_MUTEX_INITIALIZER},
    {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, {0}, 0, 0, 0, 0, 0, 0, 0, {0}, PTHREAD_MUTEX_INITIALIZER}};
static pthread_t g_main_thread;
static int g_have_main = 0;
static pthread_mutex_t g_main_mu = PTHREAD_MUTEX_INITIALIZER;

typedef struct { l2a_pool* pool; int id; } worker_arg;
static worker_arg g_worker_args[2][L2A_MAXT];

static void* pool_worker(void* argp) {
    l2a_pool* pl = ((worker_arg*)argp)->pool;
    const int id = ((worker_arg*)argp)->id;         /* 1 .. n_workers */
    pthread_mutex_lock(&pl->mu);
    unsigned seen = pl->seen0[id];
    for (;;) {
        while (pl->gen == seen) pthread_cond_wait(&pl->cv_start, &pl->mu);
        seen = pl->gen;
        if (id < pl->nthreads) {
            job_fn fn = pl->fn;
            void* arg = pl->arg;
            const int nt = pl->nthreads;
            pthread_mutex_unlock(&pl->mu);
            fn(arg, id, nt);
            pthread_mutex_lock(&pl->mu);
            if (--pl->pending == 0) pthread_cond_signal(&pl->cv_done);
        }
    }
    return NULL;
}

static void pool_atfork_child(void) {        /* the child owns no worker threads and no held locks */
    for (int q = 0; q < 2; ++q) {
        pthread_mutex_init(&g_pools[q].call_mu, NULL);
        g_pools[q].pid = 0;
    }
    pthread_mutex_init(&g_main_mu, NULL);
    g_have_main = 0;
}

static void pool_register_atfork(void) { pthread_atfork(NULL, NULL, pool_atfork_child); }

static l2a_pool* pool_of_caller(void) {
    pthread_mutex_lock(&g_main_mu);
    if (!g_have_main) { g_main_thread = pthread_self(); g_have_main = 1; }
    const int q = pthread_equal(g_main_thread, pthread_self()) ? 0 : 1;
    pthread_mutex_unlock(&g_main_mu);
    return &g_pools[q];
}

static void run_parallel(job_fn fn, void* arg, int nthreads) {
    static pthread_once_t once = PTHREAD_ONCE_INIT;
    if (nthreads > L2A_MAXT) nthreads = L2A_MAXT;
    if (nthreads <= 1) { fn(arg, 0, 1); return; }
    pthread_once(&once, pool_register_atfork);
    l2a_pool* pl = pool_of_caller();
    const int q = (int)(pl - g_pools);
    pthread_mutex_lock(&pl->call_mu);
    if (pl->pid != getpid()) {              /* first use, or a forked child */
        pthread_mutex_init(&pl->mu, NULL);
        pthread_cond_init(&pl->cv_start, NULL);
        pthread_cond_init(&pl->cv_done, NULL);
        pl->n_workers = 0;
        pl->gen = 0;
        pl->pid = getpid();
    }
    while (pl->n_workers < nthreads - 1) {
        pthread_attr_t at;
        pthread_attr_init(&at);
        pthread_attr_setdetachstate(&at, PTHREAD_CREATE_DETACHED);
        const int id = pl->n_workers + 1;
        pl->seen0[id] = pl->gen;            /* jobs are only posted below, under call_mu */
        g_worker_args[q][id].pool = pl;
        g_worker_args[q][id].id = id;
        if (pthread_create(&pl->th[id], &at, pool_worker, &g_worker_args[q][id]) != 0) {
            pthread_attr_destroy(&at);
            nthreads = pl->n_workers + 1;               /* make do with what we have */
            break;
        }
        pthread_attr_destroy(&at);
        pl->n_workers = id;
    }
    if (nthreads <= 1) { pthread_mutex_unlock(&pl->call_mu); fn(arg, 0, 1); return; }
    pthread_mutex_lock(&pl->mu);
    pl->fn = fn; pl->arg = arg; pl->nthreads = nthreads; pl->pending = nthreads - 1;
    pl->gen += 1;
    pthread_cond_broadcast(&pl->cv_start);
    pthread_mutex_unlock(&pl->mu);
    fn(arg, 0, nthreads);
    pthread_mutex_lock(&pl->mu);
    while (pl->pending != 0) pthread_cond_wait(&pl->cv_done, &pl->mu);
    pthread_mutex_unlock(&pl->mu);
    pthread_mutex_unlock(&pl->call_mu);
}

static inline long long slice_lo(long long n, int tid, int nt) { return n * tid / nt; }

/* ---- plain doubles ----------------------------------------------------------------------------------------- */
typedef struct { const mt_t* s0; double* out; long long n; mt_t end; } fill_job;

static void fill_slice(void* argp, int tid, int nt) {
    fill_job* j = (fill_job*)argp;
    const long long lo = slice_lo(j->n, tid, nt), hi = slice_lo(j->n, tid + 1, nt);
    mt_t s = *j->s0;
    mt_skip(&s, 2 * lo);
    mt_fill_double(&s, j->out + lo, hi - lo);
    if (tid == nt - 1) j->end = s;
}

/* Fill out[0..n) with the next n doubles of the stream; key[624] and *pos are NumPy's legacy state and are
 * advanced in place.  Returns 0, or -1 on a bad argument. */
int l2a_mt19937_fill_double_mt(uint32_t* key, int* pos, double* out, long long n, int nthreads) {
    if (!key || !pos || !out || n < 0 || *pos < 0 || *pos > MT_N) return -1;
    mt_t s0;
    memcpy(s0.key, key, sizeof(s0.key));
    s0.pos = *pos;
    if (n < 65536) nthreads = 1;
    fill_job j = {&s0, out, n, s0};
    if (n > 0) run_parallel(fill_slice, &j, nthreads);
    memcpy(key, j.end.key, sizeof(s0.key));
    *pos = j.end.pos;
    return 0;
}

int l2a_mt19937_fill_double(uint32_t* key, int* pos, double* out, long long n) {
    return l2a_mt19937_fill_double_mt(key, pos, out, n, 1);
}

/* Advance the state by `words` 32-bit outputs (a double consumes two). */
int l2a_mt19937_skip(uint32_t* key, int* pos, long long words) {
    if (!key || !pos || words < 0 || *pos < 0 || *pos > MT_N) return -1;
    mt_t s;
    memcpy(s.key, key, sizeof(s.key));
    s.pos = *pos;
    mt_skip(&s, words);
    memcpy(key, s.key, sizeof(s.key));
    *pos = s.pos;
    return 0;
}

/* ---- the random-shooting draw, fused -------------------------------------------------------------------------
 * `get_random_action(h*n*m)` + `.reshape(h, n*m, act_dim)` (policies/mpc_controller.py:67-69,114): rows of act_dim
 * doubles, value = low[k] + (high[k] - low[k]) * u.  Row r of the stream belongs to candidate j = r % period
 * (period = n); rows with sel_lo <= j < sel_hi (this rank's shard) are cast to fp32 and written compactly to
 * out_f32 (row-major, the layout l2a_plan_rs reads).  The first rows64 rows are also kept in float64 (out_f64):
 * `cand_a = a[0]`, whose winner the controller returns (:118,129).  Consumes rows * act_dim doubles. */
typedef struct {
    const mt_t* s0; mt_t end;
    long long rows; int act_dim; const double* low; const double* range;
    long long period, sel_lo, sel_hi; float* out_f32;
    long long rows64; double* out_f64;
} uni_job;

L2A_CLONES static void uniform_rows(const uni_job* j, mt_t* s, long long r0, long long r1) {
    enum { BLK = 1024 };
    const int ad = j->act_dim;
    const long long nsel = j->sel_hi - j->sel_lo;
    double u[BLK * 16], rng_t[BLK * 16], low_t[BLK * 16];
    const long long rows_per_blk = (BLK * 16) / ad;
    for (long long q = 0; q < rows_per_blk; ++q)        /* bounds tiled over a block: the affine map vectorises */
        for (int k = 0; k < ad; ++k) { rng_t[q * ad + k] = j->range[k]; low_t[q * ad + k] = j->low[k]; }
    for (long long r = r0; r < r1; r += rows_per_blk) {
        const long long nr = (r1 - r < rows_per_blk) ? r1 - r : rows_per_blk;
        mt_fill_double(s, u, nr * ad);
        for (long long e = 0; e < nr * ad; ++e) {
            const double v = u[e] * rng_t[e];
            u[e] = v + low_t[e];
        }
        if (j->out_f64 && r < j->rows64) {
            const long long n64 = (j->rows64 - r < nr) ? j->rows64 - r : nr;
            memcpy(j->out_f64 + r * ad, u, sizeof(double) * (size_t)(n64 * ad));
        }
        if (j->out_f32) {
            long long q = 0;
            while (q < nr) {
                const long long row = r + q;
                const long long cand = row % j->period, blk = row / j->period;
                if (cand >= j->sel_hi) { q += j->period - cand; continue; }
                if (cand < j->sel_lo) { q += j->sel_lo - cand; continue; }
                long long run = j->sel_hi - cand;           /* consecutive selected rows */
                if (run > nr - q) run = nr - q;
                float* dst = j->out_f32 + (blk * nsel + (cand - j->sel_lo)) * ad;
                const double* src = u + q * ad;
                for (long long e = 0; e < run * ad; ++e) dst[e] = (float)src[e];
                q += run;
            }
        }
    }
}

static void uniform_slice(void* argp, int tid, int nt) {
    uni_job* j = (uni_job*)argp;
    const long long lo = slice_lo(j->rows, tid, nt), hi = slice_lo(j->rows, tid + 1, nt);
    mt_t s = *j->s0;
    mt_skip(&s, 2 * lo * j->act_dim);
    uniform_rows(j, &s, lo, hi);
    if (tid == nt - 1) j->end = s;
}

int l2a_mt19937_uniform_rows(uint32_t* key, int* pos, long long rows, int act_dim, const double* low,
                             const double* high, long long period, long long sel_lo, long long sel_hi,
                             float* out_f32, long long rows64, double* out_f64, int nthreads) {
    if (!key || !pos || !low || !high || rows < 0 || act_dim < 1 || act_dim > 16 || *pos < 0 || *pos > MT_N) return -1;
    if (period < 1 || sel_lo < 0 || sel_hi < sel_lo || sel_hi > period || rows64 < 0 || rows64 > rows) return -1;
    double range[16];
    for (int k = 0; k < act_dim; ++k) range[k] = high[k] - low[k];
    mt_t s0;
    memcpy(s0.key, key, sizeof(s0.key));
    s0.pos = *pos;
    uni_job j = {&s0, s0, rows, act_dim, low, range, period, sel_lo, sel_hi, (sel_hi > sel_lo) ? out_f32 : NULL,
                 rows64, out_f64};
    if (rows * act_dim < 65536) nthreads = 1;
    if (rows > 0) run_parallel(uniform_slice, &j, nthreads);
    memcpy(key, j.end.key, sizeof(s0.key));
    *pos = j.end.pos;
    return 0;
}

/* ---- legacy Gaussian ----------------------------------------------------------------------------------------- */
typedef struct {
    const mt_t* s0;
    long long attempts;         /* attempts of this round, split over the threads */
    double* tmp;                /* [attempts][2] accepted values, thread slices at 2 * slice_lo */
    int* tmp_idx;               /* [attempts] attempt index of every accepted pair */
    long long count[L2A_MAXT];
    /* phase 2 */
    double* out; long long out_pairs; long long prefix[L2A_MAXT + 1];
} gauss_job;

L2A_CLONES static long long gauss_attempts(mt_t* s, long long a0, long long a1, double* vals, int* idx) {
    enum { BLK = 2048 };
    double u[2 * BLK], r2v[BLK], c1[BLK], c2[BLK], cr[BLK], lg[BLK];
    long long c = 0;
    for (long long a = a0; a < a1; a += BLK) {
        const int na = (int)((a1 - a < BLK) ? a1 - a : BLK);
        mt_fill_double(s, u, 2 * (long long)na);
        for (int k = 0; k < na; ++k) {
            const double x1 = 2.0 * u[2 * k] - 1.0, x2 = 2.0 * u[2 * k + 1] - 1.0;
            u[2 * k] = x1; u[2 * k + 1] = x2;
            r2v[k] = x1 * x1 + x2 * x2;
        }
        /* branch-free compaction of the accepted attempts, then the transcendental on a dense array */
        int nc = 0;
        for (int k = 0; k < na; ++k) {
            const double r2 = r2v[k];
            c1[nc] = u[2 * k]; c2[nc] = u[2 * k + 1]; cr[nc] = r2;
            idx[c + nc] = (int)(a + k - a0);
            nc += (int)((r2 < 1.0) & (r2 != 0.0));
        }
        for (int k = 0; k < nc; ++k) lg[k] = log(cr[k]);
        for (int k = 0; k < nc; ++k) {
            const double f = sqrt(-2.0 * lg[k] / cr[k]);
            vals[2 * (c + k)] = f * c2[k];          /* returned first */
            vals[2 * (c + k) + 1] = f * c1[k];      /* cached, returned by the next call */
        }
        c += nc;
    }
    return c;
}

static void gauss_phase1(void* argp, int tid, int nt) {
    gauss_job* j = (gauss_job*)argp;
    const long long lo = slice_lo(j->attempts, tid, nt), hi = slice_lo(j->attempts, tid + 1, nt);
    mt_t s = *j->s0;
    mt_skip(&s, 4 * lo);
    j->count[tid] = gauss_attempts(&s, lo, hi, j->tmp + 2 * lo, j->tmp_idx + lo);
}

static void gauss_phase2(void* argp, int tid, int nt) {
    gauss_job* j = (gauss_job*)argp;
    const long long lo = slice_lo(j->attempts, tid, nt);
    long long from = j->prefix[tid], to = j->prefix[tid + 1];
    if (to > j->out_pairs) to = j->out_pairs;
    if (to > from) memcpy(j->out + 2 * from, j->tmp + 2 * lo, sizeof(double) * 2 * (size_t)(to - from));
}

/* out[0..n) = the next n values of `np.random.normal()` / `standard_normal()` of the legacy generator (loc 0,
 * scale 1); key / pos / has_gauss / gauss are the legacy state (np.random.get_state()) and are advanced. */
int l2a_mt19937_fill_gauss(uint32_t* key, int* pos, int* has_gauss, double* gauss, double* out, long long n,
                           int nthreads) {
    if (!key || !pos || !has_gauss || !gauss || !out || n < 0 || *pos < 0 || *pos > MT_N) return -1;
    long long i = 0;
    if (n > 0 && *has_gauss) { out[i++] = *gauss; *has_gauss = 0; *gauss = 0.0; }
    long long pairs = (n - i + 1) / 2;             /* accepted attempts still needed */
    if (pairs == 0) return 0;
    const int odd = (int)((n - i) & 1);
    mt_t s;
    memcpy(s.key, key, sizeof(s.key));
    s.pos = *pos;
    double last_cached = 0.0;
    while (pairs > 0) {
        long long attempts = (long long)((double)pairs * 1.2742 * 1.01) + 64;     /* 4/pi, + margin */
        if (attempts > 0x7fffff00LL) attempts = 0x7fffff00LL;
        int nt = (attempts < 16384) ? 1 : nthreads;
        if (nt > L2A_MAXT) nt = L2A_MAXT;
        if (nt < 1) nt = 1;
        gauss_job j;
        memset(&j, 0, sizeof(j));
        j.s0 = &s; j.attempts = attempts;
        /* per-thread scratch kept between calls: a fresh 15 MB allocation per draw would be page-faulted in by
         * every call (the planner draws 720 k normals five times per controller step) */
        static __thread double* tl_tmp = NULL;
        static __thread int* tl_idx = NULL;
        static __thread long long tl_cap = 0;
        if (attempts > tl_cap) {
            free(tl_tmp); free(tl_idx);
            tl_tmp = (double*)malloc(sizeof(double) * 2 * (size_t)attempts);
            tl_idx = (int*)malloc(sizeof(int) * (size_t)attempts);
            tl_cap = (tl_tmp && tl_idx) ? attempts : 0;
            if (!tl_cap) { free(tl_tmp); free(tl_idx); tl_tmp = NULL; tl_idx = NULL; return -2; }
        }
        j.tmp = tl_tmp;
        j.tmp_idx = tl_idx;
        /* run_parallel may fall back to fewer threads only when nt collapses to 1; slices are indexed by the
         * nt actually used, which the job functions receive */
        j.count[0] = -1;
        for (int t = 0; t < L2A_MAXT; ++t) j.count[t] = -1;
        run_parallel(gauss_phase1, &j, nt);
        int used = 0;
        while (used < L2A_MAXT && j.count[used] >= 0) ++used;
        j.prefix[0] = 0;
        for (int t = 0; t < used; ++t) j.prefix[t + 1] = j.prefix[t] + j.count[t];
        const long long got = j.prefix[used];
        const long long take = (got < pairs) ? got : pairs;
        /* when n is odd the very last pair only contributes its first value */
        const int ends_here = (take == pairs);
        const long long full = (ends_here && odd) ? take - 1 : take;
        j.out = out + i; j.out_pairs = full;
        run_parallel(gauss_phase2, &j, used);
        long long used_attempts = attempts;
        if (ends_here) {
            /* locate the last accepted pair: thread tl, local index ll */
            int tl = 0;
            while (j.prefix[tl + 1] < take) ++tl;
            const long long ll = take - 1 - j.prefix[tl];
            const long long lo = slice_lo(attempts, tl, used);
            used_attempts = lo + j.tmp_idx[lo + ll] + 1;
            if (odd) {
                out[i + 2 * full] = j.tmp[2 * (lo + ll)];
                last_cached = j.tmp[2 * (lo + ll) + 1];
            }
        }
        i += 2 * full + ((ends_here && odd) ? 1 : 0);
        pairs -= take;
        mt_skip(&s, 4 * used_attempts);
    }
    memcpy(key, s.key, sizeof(s.key));
    *pos = s.pos;
    if (odd) { *has_gauss = 1; *gauss = last_cached; }
    return 0;
}

/* ---- one CEM iteration's samples, fused ------------------------------------------------------------------------
 * `rows` consecutive rows (global flat rows row_base .. row_base + rows - 1 of the n * m drawn by
 * `np.random.normal(size=(n, m, D))`, D = h * act_dim, policies/mpc_controller.py:85; flat row g = j * m + i) whose
 * standard normals are already in `z` [rows, D]:
 *   a        = mean_i + z * std_i             (:86, float64, two roundings)      -> a_out [rows, D] (may alias z)
 *   a_clip   = clip(a, low, high)             (:87)                              -> clip_out [rows, D] or NULL
 *   seq_f32  fp32 [h, m * nsel, act_dim], the tensor l2a_plan_rs reads: row of the plan = g in the reference's
 *            reading (it feeds the candidate-major rows as if they were env-major, :92-96) or i * n + j
 *            (env_major); of those rows only candidates sel_lo <= row % n < sel_hi are written, compactly
 *            (this rank's shard, or the candidate chunk being pipelined).  Values: the UNCLIPPED samples in the
 *            reference (:92-96), the clipped ones when use_clipped.
 * mean / std: [m, D]. */
typedef struct {
    const double* z; long long rows, row_base; int h, act_dim; const double* mean; const double* std; int m;
    const double* low; const double* high; double* a_out; double* clip_out; float* seq;
    long long n, sel_lo, sel_hi; int env_major, use_clipped;
    int t0, t1;         /* horizon steps [t0, t1) of every row (the rollout can be pipelined along the horizon) */
} cem_job;

L2A_CLONES static void cem_slice(void* argp, int tid, int nt) {
    cem_job* j = (cem_job*)argp;
    const long long lo = slice_lo(j->rows, tid, nt), hi = slice_lo(j->rows, tid + 1, nt);
    const int D = j->h * j->act_dim, ad = j->act_dim;
    const long long nsel = j->sel_hi - j->sel_lo, seq_rows = (long long)j->m * nsel;
    double av[4096], cv[4096];
    for (long long r = lo; r < hi; ++r) {
        const long long g = j->row_base + r;
        const int env = (int)(g % j->m);
        const double* zr = j->z + r * D;
        const double* mu = j->mean + (long long)env * D;
        const double* sd = j->std + (long long)env * D;
        const int d0 = j->t0 * ad, d1 = j->t1 * ad;
        for (int t = j->t0; t < j->t1; ++t)
            for (int k = 0; k < ad; ++k) {
                const int d = t * ad + k;
                const double zs = zr[d] * sd[d];
                const double v = mu[d] + zs;
                double c = v;
                if (c < j->low[k]) c = j->low[k];
                if (c > j->high[k]) c = j->high[k];
                av[d] = v; cv[d] = c;
            }
        memcpy(j->a_out + r * D + d0, av + d0, sizeof(double) * (size_t)(d1 - d0));
        if (j->clip_out) memcpy(j->clip_out + r * D + d0, cv + d0, sizeof(double) * (size_t)(d1 - d0));
        if (j->seq) {
            const long long prow = j->env_major ? (long long)env * j->n + g / j->m : g;
            const long long cand = prow % j->n, blk = prow / j->n;
            if (cand >= j->sel_lo && cand < j->sel_hi) {
                const long long srow = blk * nsel + (cand - j->sel_lo);
                const double* src = j->use_clipped ? cv : av;
                for (int t = j->t0; t < j->t1; ++t)
                    for (int k = 0; k < ad; ++k)
                        j->seq[((long long)t * seq_rows + srow) * ad + k] = (float)src[t * ad + k];
            }
        }
    }
}

/* Horizon steps [t0, t1) only: columns t0 * act_dim .. t1 * act_dim - 1 of a_out / clip_out and steps t0 .. t1 - 1 of
 * seq_f32 (whose layout stays [h, m * nsel, act_dim]) - one slice of the rollout's horizon pipeline. */
int l2a_cem_samples_steps(const double* z, long long rows, long long row_base, int h, int act_dim, const double* mean,
                          const double* std, int m, const double* low, const double* high, double* a_out,
                          double* clip_out, float* seq_f32, long long n, long long sel_lo, long long sel_hi,
                          int env_major, int use_clipped, int t0, int t1, int nthreads) {
    if (!z || !mean || !std || !low || !high || !a_out || rows < 0 || row_base < 0 || h < 1 || act_dim < 1 || m < 1)
        return -1;
    if (h * act_dim > 4096 || n < 1 || sel_lo < 0 || sel_hi < sel_lo || sel_hi > n || row_base + rows > n * m) return -1;
    if (t0 < 0 || t1 < t0 || t1 > h) return -1;
    cem_job j = {z, rows, row_base, h, act_dim, mean, std, m, low, high, a_out, clip_out,
                 (sel_hi > sel_lo) ? seq_f32 : NULL, n, sel_lo, sel_hi, env_major, use_clipped, t0, t1};
    if (rows * (long long)(t1 - t0) * act_dim < 65536) nthreads = 1;
    if (rows > 0 && t1 > t0) run_parallel(cem_slice, &j, nthreads);
    return 0;
}

int l2a_cem_samples(const double* z, long long rows, long long row_base, int h, int act_dim, const double* mean,
                    const double* std, int m, const double* low, const double* high, double* a_out,
                    double* clip_out, float* seq_f32, long long n, long long sel_lo, long long sel_hi,
                    int env_major, int use_clipped, int nthreads) {
    return l2a_cem_samples_steps(z, rows, row_base, h, act_dim, mean, std, m, low, high, a_out, clip_out, seq_f32, n,
                                 sel_lo, sel_hi, env_major, use_clipped, 0, h, nthreads);
}

/* ---- direct access to the global generator's words ------------------------------------------------------------
 * `addr` = np.random.mtrand._rand._bit_generator.ctypes.state_address: NumPy's `mt19937_state`
 * { uint32_t key[624]; int pos; } (numpy/random/src/mt19937/mt19937.h).  Comparing / storing 2.5 KB here replaces
 * np.random.get_state() / set_state() (tens of microseconds each) on the controller's per-step path.  The Python
 * side verifies the layout against get_state() before it trusts these. */
typedef struct { uint32_t key[MT_N]; int pos; } np_mt19937_state;

int l2a_mt19937_state_equal(const void* addr, const uint32_t* key, int pos) {
    const np_mt19937_state* s = (const np_mt19937_state*)addr;
    return (s->pos == pos && memcmp(s->key, key, sizeof(s->key)) == 0) ? 1 : 0;
}

void l2a_mt19937_state_store(void* addr, const uint32_t* key, int pos) {
    np_mt19937_state* s = (np_mt19937_state*)addr;
    memcpy(s->key, key, sizeof(s->key));
    s->pos = pos;
}

/* 64-bit FNV-1a fingerprint of (key[624], pos): what the ranks of a sharded plan compare every step to notice that one
 * of them consumed the global generator on its own (~0.6 us). */
unsigned long long l2a_mt19937_state_digest(const void* addr) {
    const np_mt19937_state* s = (const np_mt19937_state*)addr;
    unsigned long long h = 1469598103934665603ull;
    for (int i = 0; i < MT_N; ++i) { h ^= s->key[i]; h *= 1099511628211ull; }
    h ^= (unsigned int)s->pos; h *= 1099511628211ull;
    return h;
}

void l2a_mt19937_state_load(const void* addr, uint32_t* key, int* pos) {
    const np_mt19937_state* s = (const np_mt19937_state*)addr;
    memcpy(key, s->key, sizeof(s->key));
    *pos = s->pos;
}

/* Test hook: the same skip with the jump forced on (use_jump = 1, any distance >= 624) or off (0). */
int l2a_mt19937_skip_mode(uint32_t* key, int* pos, long long words, int use_jump) {
    if (!key || !pos || words < 0 || *pos < 0 || *pos > MT_N) return -1;
    mt_t s;
    memcpy(s.key, key, sizeof(s.key));
    s.pos = *pos;
    if (use_jump) {
        long long target = (long long)s.pos + words;
        if (s.pos == MT_N) { mt_regen(s.key); s.pos = 0; target -= MT_N; }
        const long long blocks = target / MT_N;
        if (blocks >= 1) {
            if (mt_jump_blocks(&s, blocks) != 0) return -2;
            s.pos = (int)(target - blocks * MT_N);
        } else {
            s.pos = (int)target;
        }
    } else {
        long long left = words;
        while (left > 0) {
            if (s.pos >= MT_N) { mt_regen(s.key); s.pos = 0; }
            long long take = MT_N - s.pos;
            if (take > left) take = left;
            s.pos += (int)take;
            left -= take;
        }
    }
    memcpy(key, s.key, sizeof(s.key));
    *pos = s.pos;
    return 0;
}

/* ---- CEM elite statistics --------------------------------------------------------------------------------------------
 * `elites = a_stacked[elites_idx]; np.mean(elites, axis=0); np.std(elites, axis=0)` (policies/mpc_controller.py:101-104) in two
 * passes over the masked rows instead of a gather and five temporaries.  Same bits as NumPy: its reduction over the leading axis
 * of a C-ordered array adds row after row (pairwise summation applies along the contiguous axis only), so per dimension d
 *   mean = (((x_0 + x_1) + x_2) + ...) / N ;  std = sqrt((((x_0 - mean)^2 + (x_1 - mean)^2) + ...) / N)
 * with the rows in index order - float64, no contraction (-ffp-contract=off).  The Python side checks that against np.mean /
 * np.std on the running machine before it trusts it.  a [rows, D] row-major, mask [rows] (non-zero = elite).  Returns the elite
 * count, or -1 on a bad argument. */
long long l2a_cem_elite_stats(const double* a, const unsigned char* mask, long long rows, int D, double* mean_out, double* std_out) {
    if (!a || !mask || !mean_out || !std_out || rows < 0 || D < 1) return -1;
    long long cnt = 0;
    for (int d = 0; d < D; ++d) { mean_out[d] = 0.0; std_out[d] = 0.0; }
    for (long long r = 0; r < rows; ++r) {
        if (!mask[r]) continue;
        const double* x = a + r * D;
        if (cnt == 0) { for (int d = 0; d < D; ++d) mean_out[d] = x[d]; }       /* (NumPy starts from the first row, not from 0.0 + x) */
        else { for (int d = 0; d < D; ++d) mean_out[d] = mean_out[d] + x[d]; }
        ++cnt;
    }
    if (cnt == 0) return 0;
    const double N = (double)cnt;
    for (int d = 0; d < D; ++d) mean_out[d] = mean_out[d] / N;
    long long seen = 0;
    for (long long r = 0; r < rows; ++r) {
        if (!mask[r]) continue;
        const double* x = a + r * D;
        if (seen == 0) { for (int d = 0; d < D; ++d) { const double t = x[d] - mean_out[d]; std_out[d] = t * t; } }
        else { for (int d = 0; d < D; ++d) { const double t = x[d] - mean_out[d]; std_out[d] = std_out[d] + t * t; } }
        ++seen;
    }
    for (int d = 0; d < D; ++d) std_out[d] = sqrt(std_out[d] / N);
    return cnt;
}

/* ---- draw-ahead chain: the NEXT controller step's candidates, drawn while the GPU runs the current plan ----------
 * The reference draws `get_random_action(h*n*m)` at the top of every controller step (policies/mpc_controller.py:
 * 67-69,114) from the global generator; the stream is a pure function of the generator state, so the block of step
 * k + 1 is known as soon as step k has consumed its numbers.  A chain owns ONE producer thread that fills the block
 * (l2a_mt19937_uniform_rows' job, on the pool of the non-main threads) from a PRIVATE copy of the state, then calls
 * `post(arg, slot)` (libl2a_hip.so: the upload of the block on a side stream) and marks the block ready.  The consumer
 *   l2a_ahead_take : waits for the block in flight; adopts it only if the global generator's words are still EXACTLY
 *                    the state the block started from (nobody else drew from np.random in between) - the global state
 *                    is then set to the block's end state, i.e. where the reference's own draw would have left it -
 *                    and returns the slot (0 / 1); anything else drops the chain and returns -1 (the caller draws
 *                    synchronously, as before)
 *   l2a_ahead_next : after a successful take (and after the caller's launch): the block after the one just taken is
 *                    produced into the other slot
 *   l2a_ahead_arm  : (re)start the chain at the CURRENT global state (after a synchronous draw)
 * Until round 5 this lived in Python (policies/draw_ahead.py: a threading.Condition shared with a worker thread
 * under the GIL, 55 us per controller step); numbers, order and the state left behind are the reference's either way.
 * Two slots suffice: the consumer is done with slot s (its plan has completed) before block k + 2 is requested.
 * One consumer thread per chain; a forked child starts with an idle chain and a fresh producer thread. */
typedef int (*l2a_ahead_post_fn)(void* arg, int slot);
typedef struct l2a_ahead {
    pthread_mutex_t mu;
    pthread_cond_t cv_work, cv_done;
    pthread_t th;
    int have_thread, quit;
    pid_t pid;
    /* the request (fixed at creation) */
    long long rows, period, sel_lo, sel_hi, rows64;
    int act_dim, nthreads;
    double low[16], high[16];
    float* out_f32[2];
    double* out_f64[2];
    l2a_ahead_post_fn post;
    void* post_arg;
    /* the chain */
    int armed;              /* a block has been requested and not been taken / dropped */
    int want;               /* request posted, not yet picked up by the producer */
    int busy;               /* producer at work */
    int ready;              /* the requested block is complete (and `end` valid) */
    int error;
    int slot;               /* slot of the requested block */
    unsigned gen;           /* bumped by every drop: results of older generations are discarded */
    mt_t base, end;
    /* diagnostics */
    unsigned long long hits, misses, produced;
    double produce_us, wait_us;
} l2a_ahead;

static double ahead_now_us(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec * 1e6 + (double)ts.tv_nsec * 1e-3;
}

static void* ahead_worker(void* argp) {
    l2a_ahead* a = (l2a_ahead*)argp;
    pthread_mutex_lock(&a->mu);
    for (;;) {
        while (!a->want && !a->quit) pthread_cond_wait(&a->cv_work, &a->mu);
        if (a->quit) break;
        a->want = 0;
        a->busy = 1;
        const unsigned gen = a->gen;
        const int slot = a->slot;
        mt_t s0 = a->base;
        pthread_mutex_unlock(&a->mu);
        const double t0 = ahead_now_us();
        double range[16];
        for (int k = 0; k < a->act_dim; ++k) range[k] = a->high[k] - a->low[k];
        uni_job j = {&s0, s0, a->rows, a->act_dim, a->low, range, a->period, a->sel_lo, a->sel_hi,
                     (a->sel_hi > a->sel_lo) ? a->out_f32[slot] : NULL, a->rows64, a->out_f64[slot]};
        int nt = a->nthreads;
        if (a->rows * a->act_dim < 65536) nt = 1;
        if (a->rows > 0) run_parallel(uniform_slice, &j, nt);
        const int rc = a->post ? a->post(a->post_arg, slot) : 0;
        const double t1 = ahead_now_us();
        pthread_mutex_lock(&a->mu);
        a->busy = 0;
        a->produced += 1;
        a->produce_us += t1 - t0;
        if (gen == a->gen) {
            a->end = j.end;
            a->error = rc;
            a->ready = 1;
        }
        pthread_cond_broadcast(&a->cv_done);
    }
    pthread_mutex_unlock(&a->mu);
    return NULL;
}

/* Takes a->mu; a forked child first rebuilds the locks (they may have been held at fork time) and forgets the parent's
 * producer thread and block.  Then makes sure the producer exists.  Returns 0 with the lock held, -1 without. */
static int ahead_lock(l2a_ahead* a) {
    if (a->pid != getpid()) {
        pthread_mutex_init(&a->mu, NULL);
        pthread_cond_init(&a->cv_work, NULL);
        pthread_cond_init(&a->cv_done, NULL);
        a->armed = a->want = a->busy = a->ready = 0;
        a->gen += 1;
        a->have_thread = 0;
        a->pid = getpid();
    }
    pthread_mutex_lock(&a->mu);
    if (!a->have_thread) {
        if (pthread_create(&a->th, NULL, ahead_worker, a) != 0) { pthread_mutex_unlock(&a->mu); return -1; }
        a->have_thread = 1;
    }
    return 0;
}

l2a_ahead* l2a_ahead_create(long long rows, int act_dim, const double* low, const double* high, long long period,
                            long long sel_lo, long long sel_hi, long long rows64, float* out_f32_slot0,
                            float* out_f32_slot1, double* out_f64_slot0, double* out_f64_slot1, int nthreads,
                            l2a_ahead_post_fn post, void* post_arg) {
    if (!low || !high || rows < 0 || act_dim < 1 || act_dim > 16) return NULL;
    if (period < 1 || sel_lo < 0 || sel_hi < sel_lo || sel_hi > period || rows64 < 0 || rows64 > rows) return NULL;
    if (sel_hi > sel_lo && (!out_f32_slot0 || !out_f32_slot1)) return NULL;
    if (rows64 > 0 && (!out_f64_slot0 || !out_f64_slot1)) return NULL;
    l2a_ahead* a = (l2a_ahead*)calloc(1, sizeof(l2a_ahead));
    if (!a) return NULL;
    pthread_mutex_init(&a->mu, NULL);
    pthread_cond_init(&a->cv_work, NULL);
    pthread_cond_init(&a->cv_done, NULL);
    a->pid = getpid();
    a->rows = rows; a->act_dim = act_dim; a->period = period; a->sel_lo = sel_lo; a->sel_hi = sel_hi; a->rows64 = rows64;
    for (int k = 0; k < act_dim; ++k) { a->low[k] = low[k]; a->high[k] = high[k]; }
    a->out_f32[0] = out_f32_slot0; a->out_f32[1] = out_f32_slot1;
    a->out_f64[0] = rows64 > 0 ? out_f64_slot0 : NULL; a->out_f64[1] = rows64 > 0 ? out_f64_slot1 : NULL;
    a->nthreads = nthreads < 1 ? 1 : nthreads;
    a->post = post; a->post_arg = post_arg;
    a->slot = 1;
    return a;
}

/* Stops the producer (waits for a block in flight: its post callback may be using the caller's buffers). */
void l2a_ahead_destroy(l2a_ahead* a) {
    if (!a) return;
    if (a->pid == getpid() && a->have_thread) {
        pthread_mutex_lock(&a->mu);
        a->quit = 1;
        pthread_cond_broadcast(&a->cv_work);
        pthread_mutex_unlock(&a->mu);
        pthread_join(a->th, NULL);
    }
    free(a);
}

static void ahead_request(l2a_ahead* a, const mt_t* base) {      /* under a->mu */
    a->base = *base;
    a->slot ^= 1;
    a->armed = 1; a->ready = 0; a->error = 0; a->want = 1;
    pthread_cond_signal(&a->cv_work);
}

/* Drop whatever the chain holds and produce the block that starts at the global generator's current state. */
int l2a_ahead_arm(l2a_ahead* a, const void* np_state_addr) {
    if (!a || !np_state_addr) return -1;
    if (ahead_lock(a) != 0) return -1;
    const np_mt19937_state* g = (const np_mt19937_state*)np_state_addr;
    mt_t base;
    memcpy(base.key, g->key, sizeof(base.key));
    base.pos = g->pos;
    a->gen += 1;            /* a block of an older generation still in production is discarded when it completes */
    ahead_request(a, &base);
    pthread_mutex_unlock(&a->mu);
    return 0;
}

/* >= 0: the slot of the adopted block (global state advanced to its end); -1: no valid block (chain dropped). */
int l2a_ahead_take(l2a_ahead* a, void* np_state_addr) {
    if (!a || !np_state_addr || a->pid != getpid()) return -1;
    pthread_mutex_lock(&a->mu);
    if (!a->armed) { pthread_mutex_unlock(&a->mu); return -1; }
    if (!a->ready) {
        const double t0 = ahead_now_us();
        while (!a->ready) pthread_cond_wait(&a->cv_done, &a->mu);
        a->wait_us += ahead_now_us() - t0;
    }
    np_mt19937_state* g = (np_mt19937_state*)np_state_addr;
    const int same = a->error == 0 && g->pos == a->base.pos && memcmp(g->key, a->base.key, sizeof(g->key)) == 0;
    a->armed = 0;
    if (!same) {
        a->gen += 1;
        a->ready = 0;
        a->misses += 1;
        pthread_mutex_unlock(&a->mu);
        return -1;
    }
    memcpy(g->key, a->end.key, sizeof(g->key));
    g->pos = a->end.pos;
    a->hits += 1;
    const int slot = a->slot;
    pthread_mutex_unlock(&a->mu);
    return slot;
}

/* After a successful take: request the block that follows it (into the other slot). */
int l2a_ahead_next(l2a_ahead* a) {
    if (!a || a->pid != getpid()) return -1;
    pthread_mutex_lock(&a->mu);
    if (a->armed || !a->ready) { pthread_mutex_unlock(&a->mu); return -1; }      /* nothing was taken */
    const mt_t base = a->end;
    ahead_request(a, &base);
    pthread_mutex_unlock(&a->mu);
    return 0;
}

/* The slot a synchronous draw of the caller may use while the chain is idle (nothing requested, nothing in production): the one
 * the chain wrote last, so that the next arm / next - which toggles - takes the other.  -1 while a block is requested. */
int l2a_ahead_idle_slot(l2a_ahead* a) {
    if (!a || a->pid != getpid()) return -1;
    pthread_mutex_lock(&a->mu);
    const int slot = (a->armed || a->want || a->busy) ? -1 : a->slot;
    pthread_mutex_unlock(&a->mu);
    return slot;
}

/* out[0..5] = hits, misses, blocks produced, producer us per block, consumer wait us per take, armed */
void l2a_ahead_stats(l2a_ahead* a, double* out) {
    if (!a || !out) return;
    if (a->pid != getpid()) { for (int i = 0; i < 6; ++i) out[i] = 0.0; return; }
    pthread_mutex_lock(&a->mu);
    out[0] = (double)a->hits; out[1] = (double)a->misses; out[2] = (double)a->produced;
    out[3] = a->produced ? a->produce_us / (double)a->produced : 0.0;
    out[4] = (a->hits + a->misses) ? a->wait_us / (double)(a->hits + a->misses) : 0.0;
    out[5] = (double)a->armed;
    pthread_mutex_unlock(&a->mu);
}

int l2a_rng_version(void) { return 8; }
