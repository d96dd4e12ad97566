/* l2a_rng.h - what libl2a_hip.so uses of libl2a_rng.so (csrc/l2a_rng.c: NumPy's legacy MT19937 stream on several threads).
 * The draw-ahead chain: see the comment above `l2a_ahead` in l2a_rng.c. */
#ifndef L2A_RNG_H_
#define L2A_RNG_H_
#ifdef __cplusplus
extern "C" {
#endif
typedef struct l2a_ahead l2a_ahead;
typedef int (*l2a_ahead_post_fn)(void* arg, int slot);
l2a_ahead* l2a_ahead_create(long long rows, int act_dim, const double* low, const double* high, long long period,
                            long long sel_lo, long long sel_hi, long long rows64, float* out_f32_slot0,
                            float* out_f32_slot1, double* out_f64_slot0, double* out_f64_slot1, int nthreads,
                            l2a_ahead_post_fn post, void* post_arg);
void l2a_ahead_destroy(l2a_ahead* a);
int l2a_ahead_arm(l2a_ahead* a, const void* np_state_addr);
int l2a_ahead_take(l2a_ahead* a, void* np_state_addr);
int l2a_ahead_next(l2a_ahead* a);
int l2a_ahead_idle_slot(l2a_ahead* a);
void l2a_ahead_stats(l2a_ahead* a, double* out6);
/* `rows` rows of act_dim uniforms low + (high - low) * u from the legacy MT19937 state (key[624], *pos), advanced in place: rows
 * whose row % period lies in [sel_lo, sel_hi) as fp32 into out_f32 (compact), the first rows64 rows as float64 into out_f64. */
int l2a_mt19937_uniform_rows(unsigned int* key, int* pos, long long rows, int act_dim, const double* low, const double* high,
                             long long period, long long sel_lo, long long sel_hi, float* out_f32, long long rows64,
                             double* out_f64, int nthreads);
int l2a_rng_version(void);
#ifdef __cplusplus
}
#endif
#endif
