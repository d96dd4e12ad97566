// l2a_philox.h - Philox4x32-10 (Salmon et al., SC'11) for the device-RNG controller step: host AND device evaluate the same
// integer rounds and the same fp32 map, so the host can recompute any candidate's action from (seed, element index) - the winner's
// first action needs no gather launch and no device-to-host copy.  (l2a_cem.hip keeps its own copy for the CEM normals; the uniform
// stream carries another domain word, the two never collide.)
#pragma once

__host__ __device__ inline void l2a_philox4x32_10(unsigned long long seed, unsigned long long ctr, unsigned int domain,
                                                  unsigned int (&c)[4]) {
    c[0] = (unsigned int)ctr; c[1] = (unsigned int)(ctr >> 32); c[2] = domain; c[3] = 0u;
    unsigned int k0 = (unsigned int)seed, k1 = (unsigned int)(seed >> 32);
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
        const unsigned int hi0 = (unsigned int)(p0 >> 32), lo0 = (unsigned int)p0;
        const unsigned int hi1 = (unsigned int)(p1 >> 32), lo1 = (unsigned int)p1;
        c[0] = hi1 ^ c[1] ^ k0; c[1] = lo1; c[2] = hi0 ^ c[3] ^ k1; c[3] = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

#define L2A_PHILOX_UNIFORM 0x756e6966u      // 'unif'

// word -> low + range * u, u = (word >> 8) / 2^24 in [0, 1): exact conversions and ONE fused multiply-add (v_fma_f32 on the device,
// libm's correctly rounded fmaf on the host): the same bits on both sides
__host__ __device__ inline float l2a_uniform_from_word(unsigned int w, float low, float range) {
    const float u = (float)(w >> 8) * (1.0f / 16777216.0f);
    return fmaf(range, u, low);
}

// element e of the stream that starts at `offset` (both in elements; four elements share a Philox block)
__host__ __device__ inline float l2a_philox_uniform(unsigned long long seed, unsigned long long e, float low, float range) {
    unsigned int c[4];
    l2a_philox4x32_10(seed, e >> 2, L2A_PHILOX_UNIFORM, c);
    return l2a_uniform_from_word(c[e & 3ull], low, range);
}
