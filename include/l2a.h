/*
 * l2a.h - C ABI of libl2a_hip.so: the MI355X (gfx950) MPC random-shooting rollout.
 *
 * The reference (iclavera/learning_to_adapt) has no FFI: its planner is Python that calls
 * `dynamics_model.predict` (-> TensorFlow `sess.run`) once per horizon step.  This header is
 * the boundary a maintainer would bind instead; every entry point names the reference code
 * it replaces.  All pointers marked "device" are HIP device pointers (in practice
 * `tensor.data_ptr()` of PyTorch-ROCm tensors - torch only stores the bytes); everything else
 * is host memory.  No torch types, no C++ types and no exceptions cross this boundary.
 *
 * Error model: functions return 0 on success and a negative L2A_E* code on failure;
 * `l2a_last_error` returns a human-readable message for the last failure on that context.
 * Threading: one context per (process, device); calls on one context are not re-entrant.
 * All device work is enqueued on the caller's stream; the library never synchronises except
 * inside l2a_init / l2a_model_create / l2a_model_destroy / l2a_destroy.
 */
#ifndef L2A_H_
#define L2A_H_

#ifdef __cplusplus
extern "C" {
#endif

typedef struct l2a_ctx l2a_ctx;
typedef struct l2a_model l2a_model;
typedef struct l2a_lstm l2a_lstm;
typedef struct l2a_controller l2a_controller;

/* ---- error codes --------------------------------------------------------------------- */
#define L2A_OK 0
#define L2A_EINVAL (-1)      /* bad argument / unsupported shape                           */
#define L2A_EHIP (-2)        /* a HIP runtime call failed (message has hipGetErrorString)  */
#define L2A_ENODEV (-3)      /* no usable gfx950 device                                    */
#define L2A_ESTATE (-4)      /* call order violated (e.g. plan before weights were set)    */
#define L2A_ESPLIT (-5)      /* l2a_plan_rs_sync only: a tile-split exchange timed out, the result is
                                invalid; call l2a_set_split(ctx, 0) and repeat the call              */

/* ---- enums ---------------------------------------------------------------------------- */
/* hidden / output nonlinearity: dynamics/mlp_dynamics.py:16-23 (`_activations`).           */
#define L2A_ACT_IDENTITY 0
#define L2A_ACT_RELU 1
#define L2A_ACT_TANH 2
#define L2A_ACT_SIGMOID 3
#define L2A_ACT_SWISH 4

/* how the E weight sets of a model are used                                                 */
#define L2A_MODE_SINGLE 0     /* E == 1: MLPDynamicsModel.predict (mlp_dynamics.py:204-222)   */
#define L2A_MODE_PER_BLOCK 1  /* row block i (env i) <-> set i: MetaMLPDynamicsModel._predict
                                 with adapted weights (meta_mlp_dynamics.py:296-306,143-163)  */
#define L2A_MODE_MEAN 2       /* delta = mean_e denorm_e(MLP_e(norm_e(.))) - build-defined
                                 ensemble of BASELINE.json (not in the reference)            */

/* which rollout kernel l2a_plan_rs / l2a_predict dispatch to                                */
#define L2A_KERNEL_AUTO 0     /* MFMA kernel when the shape allows it, else the VALU kernel  */
#define L2A_KERNEL_MFMA 1     /* fail with L2A_EINVAL when the shape is not MFMA-eligible    */
#define L2A_KERNEL_VALU 2     /* generic fp32 VALU kernel (any layer sizes)                  */

/*
 * Closed-form reward, evaluated in-kernel once per (candidate, horizon step).  Replaces the
 * host call `self.unwrapped_env.reward(observation, a[t], next_observation)`
 * (policies/mpc_controller.py:125) for the reference's env rewards:
 *   envs/half_cheetah_env.py:58-65, envs/ant_env.py:56-66, envs/arm_7dof_env.py:91-99.
 *
 *   r = w_vel * (next[vel_index] - obs[vel_index]) * inv_dt + alive
 *       - ctrl_coef * sum_k act[k]^2 - dist_coef * || next[dist_index .. dist_index+2] ||
 */
typedef struct l2a_reward {
    float w_vel;
    float inv_dt;
    float alive;
    float ctrl_coef;
    float dist_coef;
    int vel_index;
    int dist_index;
    int reserved;
} l2a_reward;

/* ---- context -------------------------------------------------------------------------- */
/* Create a context on HIP device `device`.  Lazy by design: the reference forks its env
 * workers (samplers/sampler.py:37) before it creates the TF session
 * (trainers/mb_trainer.py:46-48); callers must likewise call this after forking.            */
int l2a_init(int device, l2a_ctx** out);
void l2a_destroy(l2a_ctx* ctx);
/* Message of the last failed call on `ctx` (or of a failed l2a_init when ctx == NULL).      */
const char* l2a_last_error(const l2a_ctx* ctx);
/* Library / device facts: writes up to `cap` bytes of a JSON object (arch, CU count, ...).  */
int l2a_device_info(const l2a_ctx* ctx, char* buf, int cap);
/* Select the kernel (L2A_KERNEL_*) used by subsequent launches on this context.             */
int l2a_set_kernel(l2a_ctx* ctx, int kind);
/* Tile-split policy of the MFMA kernel.  When a plan has at most (CUs / 2) candidate tiles, two
 * workgroups can share each tile and exchange partial sums once per horizon step through global
 * memory: 0 = never; 2 = split the ensemble into two groups of whole sets (needs >= 2 sets);
 * 1 (default) = as 2, and additionally the middle set of an odd ensemble - or the only set of a
 * single / per-block model - is shared, each workgroup computing the last hidden layer and the
 * output layer for one half of the hidden units (needs >= 2 hidden layers).  Results are
 * bit-identical under all three policies: the summation order is fixed (sets: group A + group B;
 * output layer: eight chunks of hidden units in a balanced tree, ((c0+c1)+(c2+c3)) + ((c4+c5)+(c6+c7))).  */
int l2a_set_split(l2a_ctx* ctx, int policy);
/* Set batching of the MFMA kernel (models with two hidden layers).  A workgroup that runs several weight sets per
 * horizon step (an ensemble member group) can run layer 0 of `sets` sets back to back, then their hidden GEMMs and
 * output layers, then their reduces - two workgroup barriers per batch instead of two per set, the activations of a
 * batch held in LDS side by side.  0 (default) = as many as fit the CU's LDS (at most 4), 1 = one set at a time.
 * Arithmetic and summation order do not depend on it: results are bit-identical for every value.       */
int l2a_set_batch(l2a_ctx* ctx, int sets);
/* Placement of a uniformly split launch (two workgroups per candidate tile).  1 (default): the grid is padded to a
 * multiple of eight workgroups so that the workgroups of ensemble group A land on XCDs 0-3 and those of group B on XCDs
 * 4-7 exactly (hardware workgroup id % 8 = XCD); the up to six spare workgroups return at once.  Without it a tile count
 * that is not a multiple of four leaves one workgroup alone with its weight sets in a foreign L2, and its tile ends the
 * launch late (config 2, 125 tiles: 1.431 -> 1.417 ms).  0: the contiguous remap of 2 x tiles workgroups.  Placement
 * only: results are bit-identical.                                                                          */
int l2a_set_xcd_align(l2a_ctx* ctx, int on);
/* Member fan of the MFMA kernel (mean ensembles of 3 .. 8 sets).  A small plan - E x candidate tiles <= CUs, e.g. one
 * rank's 500-candidate shard of BASELINE config 5: 32 tiles - runs ONE workgroup per (candidate tile, ensemble member):
 * every workgroup streams one weight set, and once per horizon step the E workgroups of a tile swap their members' terms
 * (the tagged granules of the tile split, E - 1 partners instead of one) and add them in the unsplit launch's order.  The
 * tile split runs such a plan on 2 x tiles workgroups of 2.5 sets each; the fan on E x tiles of one set each.  1 (default)
 * = wherever it fits (and l2a_set_split is not 0: a flagged launch degrades to the unsplit geometry as before), 0 = never.
 * Geometry only: results are bit-identical.                                                                     */
int l2a_set_fan(l2a_ctx* ctx, int on);
/* Double rounds of the MFMA kernel (hidden width 512, up to 48 observation dims).  A plan of at least two rounds of
 * 16-candidate tiles (2 x CUs tiles: BASELINE config 3's 625, config 4 on one GPU, run_mb_mpc.py's default 1250) runs its
 * first 2 x CUs x D tiles as D rounds of DOUBLE tiles - two candidate tiles per workgroup on kernel instances that carry
 * neither exchange nor half-member code (and so keep their registers at this width): every weight fragment feeds both tiles
 * and a step's fixed costs are paid once for 32 candidates, ~0.945 of a single round's cost per tile.  The rest of every
 * env's candidates follows in a second launch on the same stream with the ordinary geometry (whole round, tail split); a rest
 * of more than a round and a half joins the double tiles.  The same instances also run WHOLE single tiles of plans with one
 * weight set per candidate (single model, per-block sets), 3 - 4 % faster per round than the general instances.
 * 1 (default) = wherever they apply, 0 = never (general instances only).  Geometry only: results are bit-identical.   */
int l2a_set_double_rounds(l2a_ctx* ctx, int on);
/* Micro tiles (csrc/l2a_micro.h).  A plan whose 16-candidate tiles would leave CUs idle - e.g. the reference's own default
 * plans, run_grbal.py:84-85 / run_rebal.py:77-78: 5 x 500 candidates = 160 tiles on 256 CUs - can run in candidate tiles of
 * FOUR on v_mfma_f32_4x4x1_16b_f32 instead: workgroups of 4, 8 or 12 candidates, every CU busy, no exchange between
 * workgroups.  0 = never; 1 (default) = where it shortens the launch; 2 = whenever the plan is eligible (testing).  The
 * arithmetic is ordered like the 16-candidate MATRIX-CORE kernels': results are bit-identical under all three policies wherever
 * policy 0 runs such a kernel.  The one exception: a generic recurrent stack whose 16-candidate matrix-core kernel does not fit
 * the LDS runs the VALU kernel under policy 0 (and for one-step / predict launches) - a different rounding, so there the
 * policies, and ranks of a sharded plan whose shard widths select different kernels, agree to the fp32 tolerance only.   */
int l2a_set_micro(l2a_ctx* ctx, int policy);
/* Status word of the launches issued since the last call (caller must have synchronised the
 * stream): 0 = fine, bit 0 = a member-split exchange timed out (results are invalid; relaunch
 * with l2a_set_split(ctx, 0)).  Reading clears it.                                            */
int l2a_launch_status(l2a_ctx* ctx, int* status_out);
/* How many polls a split workgroup may spend waiting for its partner, per launch, before it gives up and sets
 * status bit 0 (0 = default 2^18, about half a second).  Tests use a tiny value to force the condition. */
int l2a_set_spin_limit(l2a_ctx* ctx, unsigned int polls);
/* Test hook (host only): OR `bits` into the status word, as if a launch had reported them. */
int l2a_inject_status(l2a_ctx* ctx, int bits);
/* Developer aid (tools/timeline.py): when `device_ptr` is non-NULL the MFMA kernel's first
 * candidate tile stamps the shader clock at its phase boundaries into it as u64
 * [group 2][step h][set 8][slot 8].  NULL (default) disables it.                               */
int l2a_set_debug_buffer(l2a_ctx* ctx, void* device_ptr);

/* ---- model ----------------------------------------------------------------------------- */
/* Describe the MLP dynamics model: replaces the graph construction of
 * MLPDynamicsModel.__init__ (dynamics/mlp_dynamics.py:61-89) / the post-update graph of
 * MetaMLPDynamicsModel (meta_mlp_dynamics.py:143-163) and dynamics/core/utils.py:111-142.
 * `hidden` has `n_hidden` entries; `n_sets` weight sets are allocated (E).                  */
int l2a_model_create(l2a_ctx* ctx, int obs_dim, int act_dim, int n_hidden, const int* hidden,
                     int hidden_act, int output_act, int n_sets, int mode, l2a_model** out);
void l2a_model_destroy(l2a_model* model);

/* Load weight set `e` from fp32 DEVICE arrays in the reference's parameter order
 * (dynamics/core/layers.py:160-163; forward_mlp, dynamics/core/utils.py:250,273-274):
 *   ptrs = { hidden_0/kernel [in,h0], hidden_0/bias [h0], ..., output/kernel [hL,obs],
 *            output/bias [obs] },  kernels row-major [in, out].
 * Replaces feeding `network_params_feed_dict` on every sess.run (meta_mlp_dynamics.py:429-432).
 * The library re-packs into its own HBM layout (MFMA fragment order) on `stream`; the caller
 * keeps ownership of the source arrays and may free them once `stream` has passed this call. */
int l2a_model_set_weights(l2a_model* model, int e, const void* const* device_ptrs,
                          void* stream);

/* Normalisation statistics of set `e` (float64 HOST vectors, as `compute_normalization`
 * stores them, mlp_dynamics.py:253-262).  The +1e-10 of normalize/denormalize
 * (mlp_dynamics.py:265-270) is applied inside.  Passing NULL for all six = identity
 * (normalize_input=False).                                                                  */
int l2a_model_set_norm(l2a_model* model, int e, const double* mean_obs, const double* std_obs,
                       const double* mean_act, const double* std_act,
                       const double* mean_delta, const double* std_delta, void* stream);

/* ---- the hot path ------------------------------------------------------------------------ */
/* One random-shooting plan step: replaces the whole horizon loop of
 * MPCController.get_rs_action (policies/mpc_controller.py:116-129) and of one CEM iteration
 * (:92-100) - normalise, MLP (all sets), denormalise, state update, reward, discounted
 * return, arg-max - in one kernel launch.
 *
 *   obs0      device fp32 [m, obs_dim]        current observation of each env
 *   actions   device fp32 [h, m*n, act_dim]   candidate action sequences, row r = i*n + j
 *                                             (env i, candidate j) - the layout of
 *                                             `a.reshape(h, n*m, -1)` (mpc_controller.py:114)
 *   returns_out  device fp32 [m, n] or NULL   discounted return of every candidate
 *   best_key  device u64 [m]                  packed arg-max, see l2a_key_decode.  The library
 *                                             zeroes it on `stream` before the launch.
 *   cand_offset                               global index of local candidate 0 (multi-GPU
 *                                             sharding: indices in best_key are global)
 * Rows whose env index is i use weight set i in L2A_MODE_PER_BLOCK (needs n_sets >= m).
 * `discount` is float64 like the reference's `self.discount ** t` (:126); the kernel carries the power in
 * float64 by repeated multiplication and rounds it to fp32 once per step, where it meets the fp32 reward. */
int l2a_plan_rs(l2a_model* model, const float* obs0, const float* actions, int m, int n, int h,
                double discount, const l2a_reward* reward, int cand_offset, float* returns_out,
                unsigned long long* best_key, void* stream);

/* The blocking form of l2a_plan_rs - what `get_rs_action` (policies/mpc_controller.py:108-129) is to its caller:
 * observations in (HOST fp32 [m, obs_dim], m <= 64), arg-max keys out (HOST u64 [m]), nothing else to do.  The
 * library stages the observations in host-mapped memory that the kernel reads directly, keeps its own key slot
 * (zeroed by the previous launch) and lets the last candidate tile of the launch publish the keys and a sequence
 * number to a host-mapped mailbox which this call polls: no H2D copy, no memset, no D2H copy and no
 * hipStreamSynchronize on the path.  `actions` (device) must already be ordered before `stream`.  Single GPU only
 * (sharded plans need the keys on the device for the all-reduce: l2a_plan_rs).  Returns L2A_ESPLIT when the launch
 * was flagged (see l2a_launch_status); the status word is consumed.                                          */
int l2a_plan_rs_sync(l2a_model* model, const float* obs_host, const float* actions, int m, int n, int h,
                     double discount, const l2a_reward* reward, int cand_offset, float* returns_out,
                     unsigned long long* keys_host_out, void* stream);

/* A plan step cut along the horizon: launch k covers horizon steps t0 .. t0 + h_chunk - 1 of
 * get_rs_action's loop (policies/mpc_controller.py:116-127) and hands the per-candidate state and the
 * accumulated returns to launch k + 1.  Lets the host draw / upload the NEXT chunk of candidate actions (the
 * reference's `get_random_action`, :67-69,114, consumes its RNG stream horizon-major) while the GPU rolls out the
 * current one; the chain is bit-identical to one l2a_plan_rs over the whole horizon.
 *   state:       t0 == 0: obs0 [m, obs_dim] (state_per_row = 0) ; t0 > 0: state_out of the previous chunk
 *                [m * n, obs_dim] (state_per_row = 1)
 *   actions:     [h_chunk, m * n, act_dim] - the rows of this chunk only
 *   returns_in:  returns_out of the previous chunk (ignored when t0 == 0); returns_out [m, n] is required
 *   state_out:   [m * n, obs_dim] or NULL (last chunk); best_key: [m] or NULL (give it on the last chunk)   */
int l2a_plan_rs_chunk(l2a_model* model, const float* state, int state_per_row, const float* actions, int m, int n,
                      int h_chunk, int t0, double discount, const l2a_reward* reward, int cand_offset,
                      const float* returns_in, float* returns_out, float* state_out, unsigned long long* best_key,
                      void* stream);

/* One-step batched prediction: MLPDynamicsModel.predict / MetaMLPDynamicsModel.predict
 * (mlp_dynamics.py:204-222, meta_mlp_dynamics.py:276-294).
 *   obs device fp32 [R, obs_dim], act device fp32 [R, act_dim] -> next_obs device fp32 [R, obs_dim]
 * In L2A_MODE_PER_BLOCK the rows are split into `n_blocks` equal blocks (block i <-> set i);
 * pass n_blocks = 1 otherwise.                                                               */
int l2a_predict(l2a_model* model, const float* obs, const float* act, int rows, int n_blocks,
                float* next_obs_out, void* stream);

/* best_key packing (host helpers, pure functions):
 *   key = (orderable_u32(return) << 31) | (0x7fffffff - global_index), top bit always 0, so
 *   unsigned max == signed max == "largest return, ties -> lowest index" = np.argmax
 *   (mpc_controller.py:129).  A max all-reduce of the keys over ranks (RCCL, int64 MAX) is the
 *   only collective of a plan step.                                                          */
unsigned long long l2a_key_encode(float ret, int index);
void l2a_key_decode(unsigned long long key, float* ret, int* index);

/* Batched form of l2a_model_set_weights for `count` consecutive weight sets first_set .. first_set+count-1
 * whose parameters are stacked along a leading axis: device_ptrs[i] points at parameter i of set first_set
 * and set_strides[i] (in floats) is the distance to the same parameter of the next set.  This is the shape
 * GrBAL's inner adaptation produces (`_adapted_param_values`, meta_mlp_dynamics.py:344,429-432: one dict of
 * arrays per task) when all tasks are adapted in one batched step; 2 * (n_hidden + 1) strided copies and
 * n_hidden + 1 pack launches in total instead of per set.                                              */
int l2a_model_set_weights_strided(l2a_model* model, int first_set, int count, const void* const* device_ptrs,
                                  const long long* set_strides, void* stream);

/* GrBAL's inner adaptation on the device: for every task i < m one SGD step
 *   theta_i = theta - lr * grad mean((f_theta(x_i) - y_i)^2)
 * (MetaMLPDynamicsModel.adapt / _adapt_sym, dynamics/meta_mlp_dynamics.py:321-345,409-421; loss :118), written
 * straight into weight sets 0 .. m-1 of `model` (L2A_MODE_PER_BLOCK) in both the reference layout and the
 * kernel's packed layout - replaces the sess.run + host round trip of the adapted parameters (:344) and their
 * re-feed on every later predict (:429-432).  base_ptrs: the pre-update parameters theta in the order of
 * l2a_model_set_weights (device fp32).  x [m, rows, obs_dim + act_dim] and y [m, rows, obs_dim] (device fp32):
 * the NORMALISED inputs and target deltas of each task's adaptation batch (:324-326; rows <= 16 =
 * adapt_batch_size of run_scripts/run_grbal.py).  Requires an identity output layer and a relu / tanh /
 * sigmoid / identity hidden nonlinearity.                                                             */
int l2a_model_adapt_sgd(l2a_model* model, const void* const* base_ptrs, const float* x, const float* y, int m,
                        int rows, float lr, void* stream);

/* The same step with the adaptation batches handed over as HOST arrays - which is how the reference's caller holds
 * them (samplers/sampler.py:81-90 passes lists of NumPy arrays to dynamics_model.adapt).  x_host / y_host are copied
 * into host-mapped staging that the kernels read directly (no H2D copy on the stream; two slots, so the call only
 * waits when the launch two steps back is still running).  The arrays may be reused as soon as the call returns.
 * Results are bit-identical to l2a_model_adapt_sgd.                                                       */
int l2a_model_adapt_sgd_host(l2a_model* model, const void* const* base_ptrs, const float* x_host, const float* y_host,
                             int m, int rows, float lr, void* stream);

/* The same step from the UN-normalised transitions, as `dynamics_model.adapt(obs, act, next_obs)` receives them
 * (meta_mlp_dynamics.py:321-326): float64 host arrays obs / next_obs [m, rows, obs_dim], act [m, rows, act_dim] and the six
 * float64 normalisation vectors (`self.normalization`).  The first forward launch normalises inputs and target deltas
 * itself - (v - mean) / (std + 1e-10) in float64, then the cast to fp32, i.e. the host's arithmetic bit for bit
 * (mlp_dynamics.py:265-266) - so the host keeps nothing but the copy into staging.  Input layers of at most 128 features. */
int l2a_model_adapt_sgd_raw(l2a_model* model, const void* const* base_ptrs, const double* obs_host,
                            const double* act_host, const double* next_obs_host, const double* mean_obs,
                            const double* std_obs, const double* mean_act, const double* std_act,
                            const double* mean_delta, const double* std_delta, int m, int rows, float lr, void* stream);

/* Copy weight set `e` out of the model in the reference's parameter order and layout (device fp32 buffers of
 * the sizes l2a_model_set_weights takes) - e.g. to read back adapted sets (`_adapted_param_values`).    */
int l2a_model_get_weights(l2a_model* model, int e, void* const* device_ptrs_out, void* stream);

/* ---- cross-entropy-method planner: the per-iteration work around the rollout, on the device -----------------------
 * `MPCController.get_cem_action` (policies/mpc_controller.py:71-106).  reference = 1 keeps the reference's semantics
 * (rollouts on the UNCLIPPED samples with its candidate-major rows read as env-major, :92-96; "elites" = the boolean
 * rank mask of :101 pooled over the envs), 0 = clipped rollouts, env-major rows, true top-k per env.
 *
 * l2a_cem_sample: a = mean + z * std (:86), clip (:87) for all n * m sample rows of one iteration.  z [n, m, D]
 * (device fp32, D = h * act_dim) or NULL: standard normals from Philox4x32-10 + Box-Muller, element e of the iteration
 * drawing counter (offset + e) under `seed` - every rank of a sharded plan generates the same numbers.  mean / std
 * [m, D], low / high [act_dim] (device).  Outputs: a_clip [n, m, D], a_raw [n, m, D] (optional, the unclipped samples)
 * and seq [h, m * (hi - lo), act_dim], the candidate tensor l2a_plan_rs reads for candidates lo <= j < hi (this
 * rank's shard; optional).
 *
 * l2a_cem_refit: elites of `returns` [m, n] and the update mean = alpha * mean + (1 - alpha) * mean(elites),
 * std = std(elites) (:101-104; biased) on mean / std [m, D] in place; elite_rows [m * num_elites] is scratch.  No sort:
 * the mask of :101 only needs the ranks of the first num_elites candidates.                                     */
int l2a_cem_sample(l2a_ctx* ctx, const float* z, unsigned long long seed, unsigned long long offset,
                   const float* mean, const float* std, const float* low, const float* high, int n, int m, int h,
                   int act_dim, int reference, int lo, int hi, float* a_clip, float* a_raw, float* seq, void* stream);
int l2a_cem_refit(l2a_ctx* ctx, const float* returns, const float* a_clip, int n, int m, int D, int num_elites,
                  int reference, float alpha, int* elite_rows, float* mean, float* std, void* stream);
/* l2a_cem_pick: what the plan returns (:106), in one buffer for one read-back: per env the arg-max of the LAST iteration's
 * `returns` [m, n] (first maximum), the first action of that candidate in `cand` - the unclipped samples read as [m, n, D]
 * (reference = 1, :92-96) or the clipped samples [n, m, D] (0) - and its return; behind them the final mean / std.
 * out: m x (act_dim + 2) floats (action | return | index as the bits of an int32), then mean [m, D], std [m, D].  NaN
 * returns are ordered as np.argmax orders them (a NaN is the maximum, the first one wins): a diverged plan reports NaN.  */
int l2a_cem_pick(l2a_ctx* ctx, const float* returns, const float* cand, const float* mean, const float* std, int n, int m,
                 int D, int act_dim, int reference, float* out, void* stream);

/* ---- sharded plans: the one collective ------------------------------------------------------
 * Candidates are independent, so G GPUs (one process and one context each) plan disjoint shards of one candidate
 * tensor (l2a_plan_rs with cand_offset = first global index of the shard) and combine the per-shard keys with ONE
 * in-place MAX all-reduce of m u64 words over RCCL / xGMI - the reference has no collective (single process);
 * the semantics preserved are np.argmax's "first maximum" over the whole candidate set
 * (policies/mpc_controller.py:128-129), which the key packing encodes (see l2a_key_encode).
 *   l2a_comm_unique_id : rank 0 fills a 128-byte id (ncclGetUniqueId) and hands it to the other ranks by any means
 *                        (file, pipe, MPI, torch.distributed store)
 *   l2a_comm_init      : every rank, same id; binds the communicator to the context's device (ncclCommInitRank)
 *   l2a_allreduce_best : best_key device u64 [m], in place, enqueued on `stream` after the plan
 * RCCL is loaded at run time on the first of these calls; without it they fail with L2A_ENODEV and single-GPU
 * planning is unaffected.  (The Python drop-in uses torch.distributed - the same RCCL - for this step.)          */
int l2a_comm_unique_id(char id_out[128]);
int l2a_comm_init(l2a_ctx* ctx, int rank, int world, const char id[128]);
int l2a_comm_destroy(l2a_ctx* ctx);
int l2a_allreduce_best(l2a_ctx* ctx, unsigned long long* best_key, int m, void* stream);
/* What the ranks of a sharded plan all-reduce, packed on the device behind the plan launch (no host synchronisation
 * between the launch and the collective): payload [m + 3] u64 =
 *   [0, m)   the shard's arg-max keys (copied from best_key)
 *   [m]      1 when this context's launch status word is set (a tile-split exchange timed out: the keys are invalid)
 *   [m + 1]  digest & L2A_DIGEST_MASK, [m + 2]  L2A_DIGEST_MASK - (digest & L2A_DIGEST_MASK)
 * After an in-place MAX all-reduce of the m + 3 words (l2a_allreduce_best with m + 3, or torch.distributed) every rank
 * holds the global keys, knows whether ANY rank has to repeat its launch unsplit ([m] != 0: all ranks do, together),
 * and whether every rank planned on the same candidate tensor ([m + 1] + [m + 2] == L2A_DIGEST_MASK iff all digests
 * were equal; the digest is the caller's fingerprint of its RNG position).  One device-to-host copy of m + 3 words
 * ends the step.  Every word is below 2^63, so signed 64-bit MAX (torch.int64) reduces them correctly.            */
#define L2A_DIGEST_MASK 0x7fffffffffffull
int l2a_plan_payload(l2a_ctx* ctx, const unsigned long long* best_key, int m, unsigned long long digest,
                     unsigned long long* payload, void* stream);

/* ---- recurrent planner (ReBAL) --------------------------------------------------------------
 * Single-layer LSTM dynamics model: `RNNDynamicsModel` (dynamics/rnn_dynamics.py:11-100) built by
 * `create_rnn` (dynamics/core/utils.py:192-236) with cell_type='lstm' - the configuration of
 * run_scripts/run_rebal.py:98-99.  `cell_act` is the cell's `activation` (hidden_nonlinearity,
 * default tanh, rnn_dynamics.py:20), `output_act` the output layer's nonlinearity.
 * Fused MFMA kernel for units in {128, 256, 512}, obs_dim <= 64, act_dim <= 16; generic VALU
 * kernel otherwise (l2a_set_kernel applies).                                                    */
int l2a_lstm_create(l2a_ctx* ctx, int obs_dim, int act_dim, int units, int cell_act, int output_act,
                    l2a_lstm** out);
void l2a_lstm_destroy(l2a_lstm* model);

/* The other recurrent models `create_rnn` (dynamics/core/utils.py:192-236) can build: `cell_type` 'lstm' / 'gru' /
 * 'rnn' and several stacked layers (`len(hidden_sizes) > 1` -> tf.nn.rnn_cell.MultiRNNCell).  One LSTM layer is
 * the model of l2a_lstm_create (its own MFMA kernel); everything else runs on the generic matrix-core kernel of
 * l2a_rnn_mfma.h (l2a_set_kernel VALU: the fp32 VALU kernel of l2a_rnn_valu.h; the layers' units may sum to about 800 -
 * one candidate tile's states live in a CU's LDS).  Cell arithmetic: tensorflow==1.13.1 LSTMCell / GRUCell; 'rnn' = BasicRNNCell
 * (h = act([x | h] K + b)) - the reference passes the abstract `tf.nn.rnn_cell.RNNCell` there (:209), which cannot
 * be instantiated, so this is the evident intent, not a measured behaviour.
 * Parameters (l2a_lstm_set_weights), in `get_params()` order, kernels row-major [in_l + U_l, .] with the layer's
 * input first (in_0 = obs_dim + act_dim, in_l = U_(l-1)):
 *   lstm layer: kernel [., 4 U] (gates i j f o), bias [4 U]
 *   gru layer:  gates/kernel [., 2 U] (r | u), gates/bias [2 U], candidate/kernel [., U], candidate/bias [U]
 *   rnn layer:  kernel [., U], bias [U]
 * then output/kernel [U_top, obs_dim], output/bias [obs_dim].
 * State: wherever the l2a_lstm_* entry points take c / h [rows, units], a stack takes the layers' states
 * concatenated, [rows, sum(U_l)]; c is read and written for LSTM stacks only (pass any valid buffer otherwise). */
#define L2A_CELL_LSTM 0
#define L2A_CELL_GRU 1
#define L2A_CELL_RNN 2
int l2a_rnn_create(l2a_ctx* ctx, int obs_dim, int act_dim, int n_layers, const int* units, int cell_type,
                   int cell_act, int output_act, l2a_lstm** out);

/* Upload the four trainable variables in the reference's order (rnn.get_params(),
 * dynamics/core/layers.py:219-221): rnn/lstm_cell/kernel [obs_dim + act_dim + units, 4 * units]
 * (TF gate order i, j, f, o), rnn/lstm_cell/bias [4 * units], output/kernel [units, obs_dim],
 * output/bias [obs_dim]; fp32 device pointers, row-major.  Replaces `Layer.set_params`
 * (core/layers.py:81-94).                                                                        */
int l2a_lstm_set_weights(l2a_lstm* model, const void* const* device_ptrs, void* stream);

/* `self.normalization` of RNNDynamicsModel (rnn_dynamics.py:295-307; host float64 vectors, eps 1e-10
 * applied inside as in :329-334); all NULL = identity.                                           */
int l2a_lstm_set_norm(l2a_lstm* model, const double* mean_obs, const double* std_obs,
                      const double* mean_act, const double* std_act, const double* mean_delta,
                      const double* std_delta, void* stream);

/* One recurrent plan step = the loop of RNNMPCController.get_rs_action
 * (policies/rnn_mpc_controller.py:112-134; also the rollout of one CEM iteration, :92-104):
 * `repeat_hidden` (:165-187) of the per-env LSTM state (c0, h0: device fp32 [m, units]) to the n
 * candidates of each env, h x `dynamics_model.predict(obs, a[t], hidden)` + `env.reward` + return
 * accumulation + arg-max.  Other arguments as l2a_plan_rs.                                       */
int l2a_lstm_plan_rs(l2a_lstm* model, const float* obs0, const float* c0, const float* h0,
                     const float* actions, int m, int n, int h, double discount, const l2a_reward* reward,
                     int cand_offset, float* returns_out, unsigned long long* best_key, void* stream);

/* Blocking form of l2a_lstm_plan_rs for one GPU (see l2a_plan_rs_sync): `obs_host` [m, obs_dim] is a HOST array,
 * the arg-max keys arrive in `keys_host_out` [m] through the host-mapped mailbox (MFMA kernel; the generic kernels copy
 * and synchronise).  With c_next / h_next (device, [m, state width], not aliasing c0 / h0) the controller's own state
 * is moved on as well: state' = cell(obs, first action of the winning candidate, state) - what
 * `RNNMPCController.get_actions` does after planning (rnn_mpc_controller.py:57-65) - enqueued behind the plan without a
 * host round trip (the action is gathered from `actions` on the device; its fp32 value is the one a host would pass).
 * Returns L2A_ESPLIT like l2a_plan_rs_sync; c_next / h_next are then invalid too and the repeated call rewrites them. */
int l2a_lstm_plan_rs_sync(l2a_lstm* model, const float* obs_host, const float* c0, const float* h0,
                          const float* actions, int m, int n, int h, double discount, const l2a_reward* reward,
                          int cand_offset, unsigned long long* keys_host_out, float* c_next, float* h_next,
                          void* stream);

/* l2a_lstm_plan_rs cut along the horizon (see l2a_plan_rs_chunk): launch k covers steps t0 .. t0 + h_chunk - 1
 * and hands per-candidate observation, LSTM state and accumulated returns to launch k + 1; the chain is
 * bit-identical to one launch.  t0 == 0: state [m, obs_dim], c / h [m, units] (per_row = 0); t0 > 0: the
 * state_out / c_out / h_out of the previous chunk, [m * n, ...] (per_row = 1).  state_out, c_out, h_out: all
 * three or none (last chunk); best_key on the last chunk.                                               */
int l2a_lstm_plan_rs_chunk(l2a_lstm* model, const float* state, const float* c, const float* h, int per_row,
                           const float* actions, int m, int n, int h_chunk, int t0, double discount,
                           const l2a_reward* reward, int cand_offset, const float* returns_in, float* returns_out,
                           float* state_out, float* c_out, float* h_out, unsigned long long* best_key, void* stream);

/* RNNDynamicsModel.predict (rnn_dynamics.py:233-252): one step for `rows` independent rows, each
 * with its own LSTM state.  obs [rows, obs_dim], act [rows, act_dim], c / h [rows, units] ->
 * next_obs_out [rows, obs_dim], c_out / h_out [rows, units] (all device fp32).  Used by
 * RNNMPCController.get_actions (:63) to advance the controller's hidden state with the chosen action. */
int l2a_lstm_predict(l2a_lstm* model, const float* obs, const float* act, const float* c, const float* h,
                     int rows, float* next_obs_out, float* c_out, float* h_out, void* stream);

/* The controller's OWN state step - `_, self._hidden_state = self.dynamics_model.predict(observations, actions,
 * self._hidden_state)` (rnn_mpc_controller.py:63), whose predicted observation the caller discards: c_out / h_out [rows, state
 * width] from obs [rows, obs_dim], act [rows, act_dim] (the chosen actions), c / h (all device fp32; the outputs must not alias
 * the inputs).  For one LSTM layer a dedicated small-rows kernel (the gate matrix cut over units / 16 workgroups, no output
 * layer) - the launch l2a_lstm_plan_rs_sync enqueues behind its plan; other cells take one step of their rollout kernel
 * (rows <= 64).  Agrees with l2a_lstm_predict's states to fp32 rounding (another summation order), not bit for bit.           */
int l2a_lstm_advance(l2a_lstm* model, const float* obs, const float* act, const float* c, const float* h, int rows,
                     float* c_out, float* h_out, void* stream);

/* 1 when (obs_dim, act_dim, units) is eligible for the MFMA LSTM kernel, else 0.                 */
int l2a_lstm_mfma_eligible(int obs_dim, int act_dim, int units);

/* ---- the controller step as one call ----------------------------------------------------------
 * `MPCController.get_actions(observations)` in random-shooting mode (policies/mpc_controller.py:59-69,108-129) and
 * `RNNMPCController.get_actions` (policies/rnn_mpc_controller.py:57-65,112-134), as their caller sees them: float64
 * observations in, the float64 first action of every env's best candidate out, NumPy's legacy global generator left
 * exactly where the reference's `np.random.uniform(low, high, (h*n*m, act_dim))` (:67-69,114) leaves it.  One GPU.
 *
 * A controller owns two page-locked / HBM buffer pairs for the candidate tensor [h, m*n, act_dim] and a producer
 * thread (csrc/l2a_rng.c, `l2a_ahead_*`) that draws the NEXT step's candidates from a private copy of the generator
 * state and uploads them while the GPU runs the current plan.
 *   np_state_addr  address of the global generator's `mt19937_state` { uint32 key[624]; int pos; } - in Python
 *                  `np.random.mtrand._rand._bit_generator.ctypes.state_address` (who may touch it when: the threading
 *                  contract at l2a_controller_begin below)
 *   low / high     the action bounds (`env.action_space.low / high`, float64 [act_dim], act_dim <= 16)
 *   rng_threads    threads of one draw (the stream is cut into disjoint slices, same numbers for every count)
 * l2a_controller_step:
 *   obs          HOST float64 [m, obs_dim]        (cast to fp32 like the host would, staged in host-mapped memory)
 *   action_out   HOST float64 [m, act_dim]        cand_a[i, argmax_i] (:118,129) - the float64 values NumPy drew
 *   index_out    HOST int64 [m] or NULL           the winning candidate of each env
 *   return_out   HOST fp32 [m] or NULL            its (fp32) return
 *   returns      L2A_OK; L2A_STEP_DREW (= OK: no valid block was waiting - first call, somebody else drew from np.random
 *                since the last step - so the step drew the candidates itself, the reference's own draw from the global
 *                generator on the helper's threads, and re-armed the chain behind it); L2A_STEP_UNSPLIT (= OK, but a
 *                tile-split launch lost its partner and the plan was repeated unsplit - same bits - with the context
 *                switched to l2a_set_split(ctx, 0)); L2A_STEP_MISS when this controller cannot serve the call (a forked
 *                child): NOTHING was consumed or launched, the caller plans the ordinary way (l2a_plan_rs_sync) and may
 *                call l2a_controller_rearm afterwards; a negative L2A_E* on failure.
 * l2a_lstm_controller_step additionally takes the controller's recurrent state c0 / h0 (device fp32 [m, state width])
 * and, with c_next / h_next, advances it with the winning first actions behind the plan (l2a_lstm_plan_rs_sync).
 * l2a_controller_rearm: restart the chain at the CURRENT global generator state (after a synchronous draw).
 * l2a_controller_actions: the device tensor [h, m*n, act_dim] the latest successful step planned on (diagnostics).
 * l2a_controller_stats: out[0..6] host microseconds of the latest step - take | stage obs | launch | kick | wait |
 * decode + gather | whole call; [7] steps, [8] unsplit relaunches; [9..14] chain: hits, misses, blocks produced,
 * producer us per block, consumer wait us per take, armed; [15] steps that drew synchronously.                  */
#define L2A_STEP_MISS 1
#define L2A_STEP_UNSPLIT 2
#define L2A_STEP_DREW 3
int l2a_controller_create(l2a_model* model, int m, int n, int h, const double* low, const double* high, double discount,
                          const l2a_reward* reward, void* np_state_addr, int rng_threads, l2a_controller** out);
int l2a_lstm_controller_create(l2a_lstm* model, int m, int n, int h, const double* low, const double* high,
                               double discount, const l2a_reward* reward, void* np_state_addr, int rng_threads,
                               l2a_controller** out);
/* The SHARDED step (`MPCController._shard_range / _combine_keys`): rank r of `world` rolls out candidates
 * [r n / world, (r + 1) n / world) of every env - every rank consumes the generator for ALL h*n*m rows, so the shards are
 * slices of the one candidate tensor the reference draws (policies/mpc_controller.py:114) - and the step's ONE collective, an
 * int64 MAX all-reduce of [keys (m) | launch flag | digest | L2A_DIGEST_MASK - digest] packed on the device behind the
 * launch (l2a_plan_payload), runs in stream order; one page-locked copy of m + 3 words brings the result back.  MAX on the
 * packed keys = max return, ties -> lowest GLOBAL index = np.argmax over all candidates (:128-129); a set flag makes every
 * rank repeat launch + collective unsplit together (L2A_STEP_UNSPLIT); digests that differ (ranks seeded differently, a
 * foreign consumer of the generator on one rank) fail the step on every rank (L2A_ESTATE).
 *   reduce       NULL: RCCL over xGMI through the context's communicator (l2a_comm_init; this is l2a_allreduce_best on
 *                m + 3 words).  Otherwise the caller's collective: all-reduce `words` uint64 at `payload_dev` in place with
 *                MAX, ordered on `stream`; return 0 on success (torch.distributed behind a ctypes callback; tests).
 * Steps through l2a_controller_step / _begin / _finish like an unsharded controller; action_out = the float64 first action
 * of the GLOBAL winner (every rank holds every candidate's float64 first step).  MLP models.                          */
typedef int (*l2a_reduce_fn)(void* arg, unsigned long long* payload_dev, int words, void* stream);
int l2a_controller_create_sharded(l2a_model* model, int m, int n, int h, const double* low, const double* high,
                                  double discount, const l2a_reward* reward, void* np_state_addr, int rng_threads, int rank,
                                  int world, l2a_reduce_fn reduce, void* reduce_arg, l2a_controller** out);
/* The sharded step with the candidates drawn ON THE DEVICE (`MPCController(rng="device")` at N > 1): every rank fills its slice of
 * the SAME counter-based Philox stream (an element's value is that of its position in the whole plan's tensor [h, m*n, act_dim]),
 * so the candidates - and the chosen action - do not depend on the number of ranks, and every rank recomputes the winner's first
 * action from the stream on the host: one collective per step, no gather.  Ranks must be built with the same seed at the same
 * step (the digest pair of the collective carries seed and step count: L2A_ESTATE otherwise).                              */
int l2a_controller_create_sharded_device(l2a_model* model, int m, int n, int h, const double* low, const double* high,
                                         double discount, const l2a_reward* reward, unsigned long long seed, int rank, int world,
                                         l2a_reduce_fn reduce, void* reduce_arg, l2a_controller** out);
/* The same step with the candidates drawn ON THE DEVICE (`MPCController(rng="device")`: statistically equivalent to the reference's
 * draw, not its numbers; NumPy's generator is not touched): every step a Philox4x32-10 kernel fills the candidate tensor from the
 * counter-based stream (seed, steps so far) in front of the plan, and the winners' first actions are recomputed on the host from
 * the same stream (csrc/l2a_philox.h: identical integer rounds and fp32 map on both sides) - no upload, no gather, no copy back.
 * Steps through l2a_controller_step / l2a_lstm_controller_step as above (they never return L2A_STEP_DREW / _MISS here). */
int l2a_controller_create_device(l2a_model* model, int m, int n, int h, const double* low, const double* high, double discount,
                                 const l2a_reward* reward, unsigned long long seed, l2a_controller** out);
int l2a_lstm_controller_create_device(l2a_lstm* model, int m, int n, int h, const double* low, const double* high,
                                      double discount, const l2a_reward* reward, unsigned long long seed, l2a_controller** out);
void l2a_controller_destroy(l2a_controller* controller);
int l2a_controller_step(l2a_controller* controller, const double* obs, double* action_out, long long* index_out,
                        float* return_out, void* stream);
int l2a_lstm_controller_step(l2a_controller* controller, const double* obs, const float* c0, const float* h0,
                             float* c_next, float* h_next, double* action_out, long long* index_out, float* return_out,
                             void* stream);
/* The step in two halves.  l2a_controller_begin / l2a_lstm_controller_begin: take (or draw) the candidates, stage the
 * observations, launch, kick the producer - and return while the GPU plans (L2A_OK; L2A_STEP_MISS / negative codes as above,
 * nothing in flight then).  l2a_controller_finish: wait for the keys, decode, gather - returns what l2a_controller_step
 * returns (L2A_OK / L2A_STEP_DREW / L2A_STEP_UNSPLIT); between the two the host is free (the next env step's bookkeeping).
 * One step in flight per controller; the recurrent state pointers must stay valid until the step is finished.
 * THREADING CONTRACT of `np_state_addr`: the generator's words are read and written ONLY inside l2a_controller_begin (and
 * therefore the first half of l2a_controller_step) and l2a_controller_rearm, on the calling thread; the producer thread
 * works on a private copy of the state and never touches the caller's; l2a_controller_finish / _stats / _actions / _destroy
 * do not access it.  A caller whose OTHER threads may draw from the same generator holds that generator's lock (NumPy:
 * `_bit_generator.lock`) around _begin / _rearm - not around the wait for the GPU; a single-threaded caller (the
 * reference's run scripts) needs no lock.                                                                            */
int l2a_controller_begin(l2a_controller* controller, const double* obs, void* stream);
int l2a_lstm_controller_begin(l2a_controller* controller, const double* obs, const float* c0, const float* h0,
                              float* c_next, float* h_next, void* stream);
int l2a_controller_finish(l2a_controller* controller, double* action_out, long long* index_out, float* return_out);
int l2a_controller_rearm(l2a_controller* controller);
const float* l2a_controller_actions(const l2a_controller* controller);
int l2a_controller_stats(l2a_controller* controller, double* out, int cap);

/* ---- introspection used by tests (no GPU needed) ------------------------------------------ */
/* Which launch geometry the library picks for a plan of m envs x n candidates x h steps of an MLP model of this shape (relu /
 * identity, `n_sets` weight sets in `mode`) on a 256-CU device - the launcher's own decision code, stopped before its first HIP
 * call.  policy: NULL or {split, fan, micro, compute units, double rounds} (negative / 0 = the default: 1, 1, 1, 256, 1).  out[12] =
 * {kernel (0 generic fp32, 1 matrix core on 16-candidate tiles, 2 micro tiles), candidate tiles per workgroup, split mode
 * (0 none, 1 whole sets, 2 + shared half member, 3 member fan), first shared tile of a tail split or -1, member fan (0 / 1),
 * workgroups launched (incl. the placement's spare ones), LDS bytes per workgroup, sets per batch, micro tiles of the largest
 * workgroup, XCD placement units, double-tile workgroups of the launch IN FRONT of the described one (l2a_set_double_rounds:
 * the described launch then covers the rest of every env's candidates; 0 = the plan is one launch), 1 when the described launch
 * runs on a whole-tiles-only kernel instance (double tiles; whole single tiles of one set per candidate)}.  Results do not depend
 * on the geometry (bit-identical); this is how tests/test_host_logic.py pins the routing table without a GPU.          */
int l2a_plan_geometry(int obs_dim, int act_dim, int n_hidden, const int* hidden, int n_sets, int mode, int m, int n, int h,
                      const int* policy, int* out);
/* 1 when (obs_dim, act_dim, hidden[]) is eligible for the MFMA kernel, else 0.
 * Eligible shapes whose kernel instance keeps part of its working set in scratch memory (spilled VGPRs; correct, slower per
 * MFMA than their neighbours - tools/isa_notes.sh, profiles/r05_scratch_table.txt; none is a BASELINE.json shape):
 *   - hidden width 512 with 33 .. 64 observation dims, other than 49 inputs with relu / identity (the Ant instance is clean):
 *     6 - 8 VGPRs at 33 .. 48 observation dims, 40 - 132 at 49 .. 64.  The spill comes from the half-member paths of the tile
 *     split; the member-fan instances of the same shapes (l2a_set_fan) have none.  Alternative: the generic VALU kernel
 *     (l2a_set_kernel(ctx, L2A_KERNEL_VALU)), ~35 x slower.
 *   - l2a_lstm_mfma_eligible shapes with 512 units: 10 - 134 VGPRs (the 512-unit micro-tile instance: 14); 256 and 128 units are
 *     clean.  Alternative: the VALU kernel, ~40 x slower.
 * They stay selectable because every alternative is an order of magnitude slower.                                  */
int l2a_mfma_eligible(int obs_dim, int act_dim, int n_hidden, const int* hidden);
/* Host implementation of the weight re-packing used by the device pack kernel (same index
 * function).  Packs kernel W [k_in, n_out] (row-major) into `out`, which must hold
 * l2a_packed_layer_floats(k_in, n_out) floats.                                               */
long long l2a_packed_layer_floats(int k_in, int n_out);
int l2a_pack_layer_host(const float* w, int k_in, int n_out, float* out);
/* The micro-tile kernels' copy of a weight set (csrc/l2a_micro_pack.h: wave-stream order), on the host - what tests/
 * micro_emulator.py consumes.  l2a_micro_layout_floats: floats per set for an MLP obs_dim + act_dim -> n_hidden x hidden ->
 * obs_dim (0: no micro-tile instance for the shape).  l2a_micro_pack_layer_host: scatter layer `layer`'s row-major
 * [k_in, n_out] kernel (layer == n_hidden: the output layer) into `out`, which the caller zero-initialised.     */
long long l2a_micro_layout_floats(int obs_dim, int act_dim, int n_hidden, int hidden);
int l2a_micro_pack_layer_host(const float* w, int obs_dim, int act_dim, int n_hidden, int hidden, int layer, float* out);

#ifdef __cplusplus
}
#endif
#endif /* L2A_H_ */
