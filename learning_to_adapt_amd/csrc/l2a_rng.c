/*
 * l2a_rng.c - host helper (plain C, no GPU): the double stream of NumPy's legacy global generator, faster.
 *
 * Parity mode draws its candidate actions from `np.random` exactly as the reference does
 * (policies/mpc_controller.py:67-69).  The legacy `RandomState.random_sample` produces
 *     (a * 67108864 + b) / 9007199254740992,  a = next32() >> 5,  b = next32() >> 6
 * from the MT19937 sequence (Matsumoto & Nishimura's reference generator, as in
 * numpy/random/src/mt19937/mt19937.c); this file restates that generator with the two state-update loops and
 * the tempering written so that gcc vectorises them (dependency distances are 227 / 397 words), ~3x the
 * speed of the scalar loop.  The Python side (`learning_to_adapt_amd/utils/fast_rng.py`) takes the state out of
 * `np.random.get_state()`, calls `l2a_mt19937_fill_double` and puts the advanced state back, and verifies the
 * stream against `np.random.random_sample` once per process before trusting it.
 */
#include <stdint.h>
#include <stddef.h>

#define MT_N 624
#define MT_M 397
#define MT_A 0x9908b0dfU
#define MT_UP 0x80000000U
#define MT_LO 0x7fffffffU

/* runtime dispatch: the AVX2 clone is picked on CPUs that have it (the library is built on one machine and
 * runs on another, so no -march=native) */
#define L2A_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))

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

/* Fill out[0..n) with the next n doubles of the stream; key[624] and *pos are NumPy's legacy state and are
 * advanced in place.  Returns 0, or -1 on a bad argument. */
L2A_CLONES int l2a_mt19937_fill_double(uint32_t* key, int* pos, double* out, long long n) {
    if (!key || !pos || !out || n < 0 || *pos < 0 || *pos > MT_N) return -1;
    long long i = 0;
    int p = *pos;
    uint32_t t[MT_N];
    while (i < n) {
        if (p >= MT_N) {
            mt_regen(key);
            p = 0;
        }
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
    *pos = p;
    return 0;
}
