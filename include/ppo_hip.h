/*
 * ppo_hip.h -- C ABI of libppo_hip.so, the MI355X (gfx950) engine behind the
 * ProximalPolicyOptimization.jl rollout-and-update path.
 *
 * The reference has NO FFI: its boundary is Julia multiple dispatch on the generic functions of
 * module ProximalPolicyOptimization (src/ProximalPolicyOptimization.jl:16-30).  Each entry point
 * below names the reference function it stands in for; the Julia-side binding (ccall methods for
 * HipVecEnv / HipPolicy / HipRollouts) is in julia/ProximalPolicyOptimizationHIP.jl and
 * INTEGRATION.md.  The Python mirror used by the tests is proximalpolicyoptimization.jl_amd/.
 *
 * Conventions
 *  - every function returns int32 status: 0 = ok, <0 = error; ppo_last_error() gives the text.
 *    (reference: ErrorException / AssertionError, SURVEY.md 8(b) "Errors")
 *  - every pointer is a HOST pointer owned by the caller unless the name ends in _dev;
 *    getters copy into caller buffers; handles own all device memory.
 *  - action indices crossing this ABI are 0-based int32 (Julia shim adds/subtracts 1).
 *  - layouts: rollout columns are time-major [T,N]; states are int8 [.,H,F] (row = half-edge,
 *    F = 72 features: 36 template scores then 36 template degrees); the action mask travels as
 *    the active-quad bit mask (bit q set = quad q active; entries a with quad a/16 inactive are
 *    -Inf in the reference, test/quad_game_utilities.jl:39-44).
 *  - policy parameters are one flat float32 vector in Flux order (W1,b1, the num_hidden_layers-1 hidden->hidden
 *    (W,b) pairs, W_out,b_out), W stored [out,in] column-major, so Flux.params(policy) round-trips (test/policy.jl:9-19).
 *  - one engine per HOST THREAD: the device (ppo_device_init), the stream (ppo_set_stream), the kernel timers, the RCCL
 *    communicator and the error text are thread-local, so a process may run several engines side by side (one thread per
 *    GPU); a handle belongs to the thread that created it and is not re-entrant.  The ppo_set_* tuning knobs are
 *    process-wide.
 */
#ifndef PPO_HIP_H
#define PPO_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ppo_env_s* ppo_env_t;
typedef struct ppo_policy_s* ppo_policy_t;
typedef struct ppo_adam_s* ppo_adam_t;
typedef struct ppo_rollouts_s* ppo_rollouts_t;

#define PPO_OK 0
#define PPO_ERR_ARG (-1)        /* bad argument / failed @assert-equivalent            */
#define PPO_ERR_HIP (-2)        /* HIP runtime error                                   */
#define PPO_ERR_DEVICE_FLAG (-3)/* device-side error flag (zero-prob action sampled,   */
                                /* invalid action index, stepping a terminated env)    */
#define PPO_ERR_UNSUPPORTED (-4)

/* advantage plugin modes (batch_advantage, src/ProximalPolicyOptimization.jl:29; no
 * implementation exists in the reference, old scripts use raw returns) */
#define PPO_ADV_RETURNS 0            /* advantage = returns (what the reference's old scripts do)            */
#define PPO_ADV_RETURNS_NORMALISED 1 /* (R - mean) / (std + 1e-8) over the minibatch (population std, fp64 stats) */
#define PPO_ADV_GAE 2                /* the GAE(gamma, lambda) column of ppo_rollouts_compute_gae (host-supplied state values) */
#define PPO_ADV_GAE_NORMALISED 3     /* the same, normalised over the minibatch like mode 1 */

/* ---------------------------------------------------------------- library / device */
int32_t ppo_version(void);
int32_t ppo_last_error(char* buf, int64_t cap);
int32_t ppo_device_init(int32_t device_ordinal);          /* hipSetDevice + private stream     */
int32_t ppo_set_stream(void* hip_stream);                 /* run on an external hipStream_t     */
int32_t ppo_device_synchronize(void);
int32_t ppo_device_count(int32_t* out);
/* rollout execution: 0 = three launches per step (observe, policy forward + sample, step!), 1 = ONE launch for the
 * whole T-step rollout wherever the shape is covered (envs are independent: each wavefront walks its envs through all
 * steps with the env state in LDS), -1 (default) = automatic: one launch for Q = 8 envs, per-step launches otherwise.
 * Same results bit for bit; per-step launches are always used while a disk sink is attached. */
int32_t ppo_set_rollout_persistent(int32_t mode);
/* state storage of engine-collected rollouts (R4, src/rollout_buffer.jl:1-22: the reference boxes every state):
 * 0 = expanded observation rows [H][F] int8 per transition (2304 B for Q = 8), 1 = compact: the env snapshot the rows
 * are derived from (score[V] + degree[V] int8 = 64 B for Q = 8, plus the active-quad word) -- the train forward
 * re-derives the rows exactly as state(env) does, the getters expand on demand; -1 (default) = automatic: compact when
 * the expanded rollout would exceed 32 GiB (PPO_COMPACT_AUTO_BYTES; below that the rows are kept because re-deriving
 * them costs the train forward 3 % in fp32 and 17 % in bf16 mode) or while a disk sink is attached (the streamed record
 * shrinks 28x).  Same results bit for bit.  Host-supplied rollouts (ppo_rollouts_set) are always expanded. */
int32_t ppo_set_rollout_compact(int32_t mode);

/* ---------------------------------------------------------------- standalone ops (parity entry points) */
/* compute_returns(rewards, terminal, discount)            src/collect_rollouts.jl:26-42
 * flat concatenated-episodes semantics; discount_is_f32=0 reproduces the Float64 running value */
int32_t ppo_compute_returns(const float* rewards, const uint8_t* terminal, int64_t n,
                            double discount, int32_t discount_is_f32, float* out);
/* same scan on time-major [T,N] columns (v=0 at each column tail) -- the engine's hot layout */
int32_t ppo_compute_returns_tn(const float* rewards, const uint8_t* done, int64_t T, int64_t N,
                               double discount, int32_t discount_is_f32, float* out);
/* GAE(gamma,lambda) extension (values [T+1,N]); lambda=1, V=0 == ppo_compute_returns_tn */
int32_t ppo_gae_tn(const float* rewards, const uint8_t* done, const float* values, int64_t T,
                   int64_t N, double gamma, double lambda, float* adv_out, float* ret_out);
/* rand(Categorical(p)) for B rows of A probabilities given B uniforms   src/collect_rollouts.jl:6-7
 * actions 0-based; err[b]=1 when the sampled entry has p==0 (the reference's @assert) */
int32_t ppo_categorical_sample(const float* probs, const float* u, int64_t B, int64_t A,
                               int32_t* actions, float* p_sel, int32_t* err);
/* get_linear_action_index (1-based in, 1-based out)        src/train.jl:48-52 */
int32_t ppo_linear_action_index(const int64_t* actions1, int64_t B, int64_t A, int64_t* out);
/* ppo_loss_with_entropy on given probs [A,B] (column-major), 1-based linear indices
 *                                                          src/train.jl:21-26,35-46 */
int32_t ppo_loss_with_entropy(const float* probs, const int64_t* lin_idx1, const float* p_old,
                              const float* adv, int64_t B, int64_t A, double epsilon,
                              double* ppoloss, double* entropyloss);
/* counter RNG exposed for tests: Philox4x32-10 */
int32_t ppo_philox4x32_10(const uint32_t* ctr4, const uint32_t* key2, int64_t n, uint32_t* out4);

/* ---------------------------------------------------------------- env plugin (batched) */
/* PPO.state / reward / is_terminal / reset! / step!        src/ProximalPolicyOptimization.jl:16-20
 * kind 0 = synthetic rand-poly-shaped env (Q quad slots, H=4Q half-edges, A=16Q actions) */
int32_t ppo_env_create(int32_t kind, int64_t num_envs, int64_t global_env_offset, int32_t Q,
                       int32_t max_actions, float no_action_reward, uint64_t seed, ppo_env_t* out);
int32_t ppo_env_destroy(ppo_env_t env);
int32_t ppo_env_dims(ppo_env_t env, int64_t* N, int32_t* H, int32_t* F, int32_t* A);
int32_t ppo_env_reset(ppo_env_t env);                                   /* reset!      :19 */
int32_t ppo_env_step(ppo_env_t env, const int32_t* actions0);           /* step!       :20 */
int32_t ppo_env_get_state(ppo_env_t env, int8_t* obs, uint32_t* active);/* state       :16 */
int32_t ppo_env_get_reward(ppo_env_t env, float* out);                  /* reward      :17 */
int32_t ppo_env_get_terminal(ppo_env_t env, uint8_t* out);              /* is_terminal :18 */
int32_t ppo_env_get_internal(ppo_env_t env, int8_t* score, int8_t* degree, int32_t* steps,
                             uint32_t* episode, uint32_t* tick);
int32_t ppo_env_check_errors(ppo_env_t env, int32_t* flags_or_null);
/* rand(Categorical(ap)) walks the CDF sequentially; when the fp32 sum of ap stops short of u (u within 2^-24 of 1) the
 * walk ends on the LAST action, and if that action is masked the reference's `@assert ap[a] > 0.0` throws
 * (src/collect_rollouts.jl:7) -- about 3 in 10^8 samples.  Default (strict = 0): the residue goes to the last action
 * with ap > 0 and flag bit 32 records it.  strict = 1: the rollout call fails with PPO_ERR_DEVICE_FLAG like the
 * reference's AssertionError. */
int32_t ppo_env_set_strict_sampling(ppo_env_t env, int32_t strict);

/* ---------------------------------------------------------------- policy plugin */
/* SimplePolicy.Policy(in, hidden, num_hidden_layers, out)  test/policy.jl:9-19
 * in = 72 or 216 features, hidden = 1..256 (widths other than 128 / 256 run zero-padded on the next wider kernel, exact),
 * num_hidden_layers = 1..4 (2: the fused kernels; 1, 3, 4: the layer-looped forms of the same kernels, fp32 only),
 * out = 4 (the quad game's actions per edge, test/quad_game_utilities.jl:95).  Anything else: PPO_ERR_UNSUPPORTED. */
int32_t ppo_policy_create(int32_t F, int32_t hidden, int32_t num_hidden_layers,
                          int32_t out_per_edge, ppo_policy_t* out);
/* arithmetic type of the policy MLP (BASELINE config 5: "bf16 MLP on MFMA + fp32 GAE").  PPO_DTYPE_F32 (default):
 * exact fp32 MFMA.  PPO_DTYPE_BF16: weights and layer inputs rounded to bfloat16 (RNE), fp32 accumulation, bias,
 * leakyrelu, softmax, sampling, loss and Adam (fp32 master parameters) unchanged; saved activations and the
 * gradient signals dZ are bf16.  Call before training; the reference has no counterpart (Flux Float32 only). */
#define PPO_DTYPE_F32 0
#define PPO_DTYPE_BF16 1
int32_t ppo_policy_set_dtype(ppo_policy_t pol, int32_t dtype);
int32_t ppo_policy_get_dtype(ppo_policy_t pol, int32_t* dtype);
int32_t ppo_policy_destroy(ppo_policy_t pol);
int32_t ppo_policy_num_params(ppo_policy_t pol, int64_t* n);
int32_t ppo_policy_set_params(ppo_policy_t pol, const float* flat);     /* Flux.params order */
int32_t ppo_policy_get_params(ppo_policy_t pol, float* flat);
/* batch_action_probabilities(policy, state) -> probs [A,B] column-major (B=1: action_probabilities)
 *                                                          test/quad_game_utilities.jl:65-79 */
int32_t ppo_policy_forward(ppo_policy_t pol, const int8_t* states, const uint32_t* active,
                           int64_t B, int32_t H, float* probs);
/* flat gradient of the last ppo_forward_backward (Flux order) */
int32_t ppo_policy_get_grad(ppo_policy_t pol, float* flat);
/* device pointer to the flat gradient buffer [num_params + 2] (grad, ppo-sum, entropy-sum):
 * the hand-off point for the data-parallel all-reduce (RCCL through torch.distributed) */
int32_t ppo_policy_grad_buffer_dev(ppo_policy_t pol, void** dev_ptr, int64_t* n_floats);

/* ---------------------------------------------------------------- optimiser */
/* Flux.Optimiser(Adam(eta,(beta1,beta2),eps)) + Flux.update!            src/train.jl:81,155-158 */
int32_t ppo_adam_create(ppo_policy_t pol, double eta, double beta1, double beta2, double eps,
                        ppo_adam_t* out);
int32_t ppo_adam_destroy(ppo_adam_t opt);
int32_t ppo_adam_get_lr(ppo_adam_t opt, double* eta);                    /* get_optimizer_learning_rate */
int32_t ppo_adam_set_lr(ppo_adam_t opt, double eta);
int32_t ppo_adam_get_state(ppo_adam_t opt, float* m, float* v, double* beta_pow2);
int32_t ppo_adam_set_state(ppo_adam_t opt, const float* m, const float* v, const double* beta_pow2);
/* number of epochs this optimiser has trained through ppo_train.  Together with the caller's seed it keys the device
 * minibatch permutation (the stand-in for randperm, src/train.jl:93), so there is no hidden process-global state: the
 * same seed with a fresh optimiser reproduces a run, and a checkpointed optimiser (m, v, beta powers, epoch count)
 * resumes one. */
int32_t ppo_adam_get_epoch_count(ppo_adam_t opt, int64_t* epochs);
int32_t ppo_adam_set_epoch_count(ppo_adam_t opt, int64_t epochs);

/* ---------------------------------------------------------------- rollout buffer */
/* BufferRollouts()                                          src/rollout_buffer.jl:1-22 */
int32_t ppo_rollouts_create(ppo_env_t env, int64_t capacity_T, ppo_rollouts_t* out);
/* the same for states that do not come from the built-in env (a user env's state(env) converted by the host, e.g. the
 * reference's 216-feature level-4 template, test/output/catmull-clark-policy-l4.bson): N columns of [H][F] int8 rows,
 * H = 32 or 128, F a multiple of 8 the policy was created for; filled with ppo_rollouts_set */
int32_t ppo_rollouts_create_shape(int64_t num_envs, int32_t H, int32_t F, int64_t capacity_T, ppo_rollouts_t* out);
int32_t ppo_rollouts_destroy(ppo_rollouts_t ro);
int32_t ppo_rollouts_len(ppo_rollouts_t ro, int64_t* n);                 /* Base.length :40-48 */
int32_t ppo_rollouts_dims(ppo_rollouts_t ro, int64_t* T, int64_t* N);
/* collect_rollouts!(rollouts, env, policy, ., discount)     src/rollout_buffer.jl:66-79
 * fixed-T vectorised form: T steps of all N envs with auto-reset, then compute_state_value!
 * (returns overwrite rewards, :55-64).  record_probs!=0 additionally keeps the full [T,N,A]
 * probabilities (tests only). */
int32_t ppo_collect_rollouts(ppo_rollouts_t ro, ppo_env_t env, ppo_policy_t pol, int64_t T,
                             double discount, int32_t discount_is_f32, int32_t record_probs);
/* num_episodes form (the reference's own signature): EXACTLY num_episodes whole episodes enter the buffer
 * (src/rollout_buffer.jl:73-77).  The N resident envs play them in parallel, episode e on env e mod N (env n plays
 * ceil((num_episodes - n) / N) episodes, reset! before each; envs beyond num_episodes stay idle).  Dataset order =
 * env-major concatenation of the whole episodes; length = number of valid transitions. */
int32_t ppo_collect_rollouts_episodes(ppo_rollouts_t ro, ppo_env_t env, ppo_policy_t pol,
                                      int64_t num_episodes, double discount,
                                      int32_t discount_is_f32);
/* column getters, time-major [T,N] (dataset getindex, src/rollout_buffer.jl:103-133) */
int32_t ppo_rollouts_get_states(ppo_rollouts_t ro, int8_t* states, uint32_t* active);
int32_t ppo_rollouts_get_actions(ppo_rollouts_t ro, int32_t* actions0);
int32_t ppo_rollouts_get_probs(ppo_rollouts_t ro, float* p_sel);
int32_t ppo_rollouts_get_returns(ppo_rollouts_t ro, float* returns);     /* "rewards" after :55-64 */
int32_t ppo_rollouts_get_raw_rewards(ppo_rollouts_t ro, float* rewards);
int32_t ppo_rollouts_get_terminal(ppo_rollouts_t ro, uint8_t* terminal);
int32_t ppo_rollouts_get_valid(ppo_rollouts_t ro, uint8_t* valid);
int32_t ppo_rollouts_get_full_probs(ppo_rollouts_t ro, float* probs);   /* [T,N,A], record_probs only */
/* dataset order: flat index list of the valid transitions (t*N+n), length = ppo_rollouts_len */
int32_t ppo_rollouts_get_index(ppo_rollouts_t ro, int64_t* idx);
/* batch_advantage as GAE(gamma, lambda) (north_star "GAE advantage reverse scan"; the reference declares the plugin at
 * src/ProximalPolicyOptimization.jl:29 and implements nothing).  values: [T+1][N] state values from the caller's critic
 * (row T = bootstrap value behind the last step; the reference has no value head).  Runs the LDS-tiled fp64 scan over
 * the buffer's raw rewards / terminal flags and keeps the advantage column on the device for adv_mode PPO_ADV_GAE*;
 * optional host copies of the advantages and of the lambda-returns adv + V.  lambda = 1 and V = 0 reproduce
 * compute_returns (src/collect_rollouts.jl:26-42) bit for bit. */
int32_t ppo_rollouts_compute_gae(ppo_rollouts_t ro, const float* values, double gamma, double lambda,
                                 float* adv_out_or_null, float* lambda_returns_out_or_null);
/* load columns from host (tests / generic host-side envs) */
int32_t ppo_rollouts_set(ppo_rollouts_t ro, int64_t T, const int8_t* states, const uint32_t* active,
                         const int32_t* actions0, const float* p_sel, const float* returns,
                         const uint8_t* terminal);

/* ---------------------------------------------------------------- training */
/* forward + loss + backward of one minibatch given dataset positions `sample_idx` (0-based
 * positions into the dataset order).  Leaves the flat gradient (mean over B_global) in the
 * policy's gradient buffer.  B_global = minibatch size across all data-parallel ranks.
 *                                                          src/train.jl:35-46,65-79 */
int32_t ppo_forward_backward(ppo_policy_t pol, ppo_rollouts_t ro, const int64_t* sample_idx,
                             int64_t B, int64_t B_global, double epsilon, double entropy_weight,
                             int32_t adv_mode);
/* Which backward kernel a minibatch takes (same gradient, different reduction tree: results agree to fp32 rounding):
 * up to `tiles` 32-row tiles the three-product form (dZ kernel + output-stationary split-K weight-gradient kernel: no
 * per-workgroup gradient slabs, the fixed cost that dominates a small optimiser step), above it the fused kernel that
 * keeps every weight gradient resident in MFMA accumulators.  Default 384 (-1 restores it), 0 = always fused.  While the
 * split-fp32 training pass is on (ppo_set_bwd_split_bf16, the default) its fused backward is faster at every size and the
 * threshold that applies is PPO_BWD_SMALL_MAX_TILES_SPLIT (default 0). */
int32_t ppo_set_bwd_small_max_tiles(int64_t tiles);
/* fp32 policies, fused backward (Policy(72, h, 2, 4)): 1 = its three big products (dH1 = dZ2 W2, dW2 += dZ2^T H1,
 * dW1 += dZ1^T X) run on the bf16 matrix pipe as SPLIT-fp32 products -- every fp32 operand is the exact sum of three
 * bfloat16 pieces, six piece products (fp32 accumulation) carry a product to one fp32 rounding -- instead of on the 16x
 * slower fp32 matrix instructions; 0 = the pure fp32-MFMA kernel; -1 = the default (environment PPO_BWD_SPLIT_BF16, else
 * on).  Same inputs, same gradient slab; gradients agree with the other form to fp32 rounding and meet the same 2e-5 max|g|
 * bar against the float64 restatement. */
int32_t ppo_set_bwd_split_bf16(int32_t mode);
/* Small minibatches (the per-GPU shard of a strong-scaling run): up to `tiles` 32-row tiles the train forward, the loss and
 * the backward-data pass of a tile run in ONE workgroup (k_policy_train_tile: nothing but the operands of the weight-
 * gradient products leaves the CU) followed by the split-K weight-gradient kernel, instead of the separate forward and
 * backward launches.  Same gradient to fp32 rounding (the layer-3 partial sums are added in another order), bitwise
 * reproducible.  -1 restores the default, 0 = never. */
int32_t ppo_set_train_tile_max_tiles(int64_t tiles);
/* Likewise for the train forward: minibatches of up to `states` states (H = 32) give every state to 2 or 4 waves instead
 * of one, so a minibatch smaller than the chip's 1024 SIMDs still fills it (logits agree with the one-wave kernel to
 * fp32 rounding: the layer-3 partial sums are added in a different order).  Default 512 (-1 restores it), 0 = never. */
int32_t ppo_set_fwd_split_max_states(int64_t states);
/* And for the one-launch rollout: up to `envs` resident envs (Q = 8, fp32) every env is walked by 2 or 4 waves instead of
 * one.  Unlike the train forward this split is BIT-EXACT (the layer-3 fmaf chain is handed from wave to wave in tile
 * order), so it changes nothing but the time.  Default 512 (-1 restores it), 0 = never. */
int32_t ppo_set_rollout_split_max_envs(int64_t envs);
/* Flux.update!(optimizer, weights, grad)                    src/train.jl:81 */
int32_t ppo_adam_apply(ppo_adam_t opt, ppo_policy_t pol);
/* losses of the last forward_backward (after any all-reduce): (ppoloss, entropy_weight*entropyloss)
 *                                                          src/train.jl:83 */
int32_t ppo_last_losses(ppo_policy_t pol, double* ppoloss, double* entropyloss);
/* step_batch!                                               src/train.jl:54-84 */
int32_t ppo_step_batch(ppo_policy_t pol, ppo_adam_t opt, ppo_rollouts_t ro,
                       const int64_t* sample_idx, int64_t B, double epsilon, double entropy_weight,
                       int32_t adv_mode, double* ppoloss, double* entropyloss);
/* all-reduce hook: called once per optimiser step between backward and Adam with the device
 * gradient buffer (sum over ranks expected on return, enqueued on / ordered with the stream).
 * Contract: the hook must reduce EXACTLY the buffer it is handed on each call -- `grad_dev`, `n_floats` floats -- and it
 * must SUM (not average): ppo_train also calls it on two small non-gradient buffers at its start (the shard-length
 * exchange, 2 * world floats, and a 1-float status agreement), so a hook that ignores the pointer / count and reduces a
 * cached gradient tensor, or one that divides by the world size, breaks the exchange (ppo_train then fails on every
 * rank alike with "shard-length exchange is inconsistent").  Return 0 on success. */
typedef int32_t (*ppo_allreduce_fn)(void* ctx, void* grad_dev, int64_t n_floats);
/* ppo_train!(policy, optimizer, dataset, epsilon, batch_size, num_epochs, entropy_weight)
 *                                                          src/train.jl:86-153
 * perm: NULL -> device Feistel permutation keyed by (seed, ppo_adam_get_epoch_count), else num_epochs
 * explicit 0-based permutations of length len (randperm, :93).  hist arrays have num_epochs
 * entries: mean per-batch losses and the learning rate (:127,144-150).
 * Data parallel: rank / world = this process's place among the ranks that each hold an env shard (1 process: 0 / 1).
 * The minibatch of a step is the union of `batch_size` samples per rank.  Shards may differ in length: the ranks
 * exchange their dataset lengths once per call (through the hook), all run max_r ceil(len_r / batch_size) steps per
 * epoch, each step's gradient is the exact mean over the samples all ranks contributed to it, and a rank whose shard
 * is exhausted contributes zeros -- so every rank issues the same sequence of collectives.  batch_size must not
 * exceed the shortest shard (the reference's @assert, on every rank alike). */
int32_t ppo_train(ppo_policy_t pol, ppo_adam_t opt, ppo_rollouts_t ro, double epsilon, int64_t batch_size,
                  int32_t num_epochs, double entropy_weight, int32_t adv_mode, const int64_t* perm, uint64_t seed,
                  int32_t rank, int32_t world, ppo_allreduce_fn allreduce, void* allreduce_ctx, double* ppo_hist,
                  double* entropy_hist, double* lr_hist);

/* Native hook (the default of bench.py and DataParallel): the same all-reduce as ONE RCCL call made by the library
 * itself on the engine's stream -- no host-language callback per optimiser step.  RCCL is resolved with dlopen at first
 * use (a host that already carries an RCCL, e.g. torch, shares its copy).  The engine owns no rendezvous: the host
 * distributes the 128-byte unique id between its ranks (rank 0 calls ppo_rccl_unique_id, every rank ppo_rccl_init
 * after ppo_device_init), runs ppo_rccl_self_test collectively, and passes ppo_rccl_allreduce as the `allreduce`
 * argument of ppo_train.  Replaces, on the reference side, nothing: the reference is single-process. */
int32_t ppo_rccl_probe(void);                      /* local: can librccl be resolved?  agree on it before ppo_rccl_init */
int32_t ppo_rccl_unique_id(uint8_t* out128);
int32_t ppo_rccl_init(int32_t rank, int32_t world, const uint8_t* id128);
int32_t ppo_rccl_allreduce(void* ctx, void* grad_dev, int64_t n_floats);      /* a ppo_allreduce_fn */
int32_t ppo_rccl_comm_info(int32_t* rank, int32_t* world);                    /* as the communicator reports them */
int32_t ppo_rccl_self_test(int32_t* ok);                                      /* collective: known-answer all-reduce */
int32_t ppo_rccl_finalize(void);

/* ---------------------------------------------------------------- out-of-core rollout store
 * DiskRollouts / DiskDataset (src/rollouts_to_disk.jl:1-171, src/dataset.jl:1-82) for rollouts that should not
 * stay resident: while ppo_collect_rollouts runs, every finished step [N] is copied device -> pinned host with
 * hipMemcpyAsync on a copy stream (ordered by events, overlapping the next step's kernels) and a writer thread
 * appends it to <dir>/rollout.bin; the returns column is appended when the scan has run (the reference rewrites
 * trajectory.csv at the same point, :106-132).  One fixed-size binary shard instead of one BSON file per state;
 * the reference's CSV + per-state BSON layout is produced by the host-side exporter for small runs.
 * attach wipes and recreates <dir> like the DiskRollouts constructor (:7-13,23-45). */
int32_t ppo_rollouts_attach_disk(ppo_rollouts_t ro, const char* dir, int32_t pinned_slots);
int32_t ppo_rollouts_detach_disk(ppo_rollouts_t ro);
/* Deferred finish of a streamed collection.  ppo_set_disk_async(1) (PPO_DISK_ASYNC=1; default 0): the pinned ring is sized to
 * hold EVERY step of the collection (up to PPO_DISK_ASYNC_MAX_BYTES, default 1 GiB), ppo_collect_rollouts returns as soon as the
 * last step and the returns column are on the copy stream, and the writer thread finishes <dir>/rollout.bin on its own while
 * the caller trains on the columns that stayed in HBM.  The file is complete when ppo_rollouts_disk_sync returns (the next
 * collection into the same buffer, detach and destroy wait for it too).  With mode 0 the file is complete when
 * ppo_collect_rollouts returns, as before (the reference's rollouts_to_disk is synchronous: src/rollouts_to_disk.jl:73-132). */
int32_t ppo_set_disk_async(int32_t mode);
int32_t ppo_rollouts_disk_sync(ppo_rollouts_t ro);
/* DiskDataset: read <dir>/rollout.bin back into the (device) rollout buffer; shapes must match the env it was
 * created for.  All transitions are valid afterwards. */
int32_t ppo_rollouts_load_disk(ppo_rollouts_t ro, const char* dir);

/* average_returns(policy, env, num_trajectories) -> (mean, sample std [n-1]) of the UNdiscounted episode return
 *                                                          src/evaluate.jl:1-25
 * exactly num_trajectories whole episodes, trajectory e on resident env e mod N (reset! before each, stochastic policy);
 * `scratch` is a rollout buffer created for this env (its contents are overwritten). */
int32_t ppo_average_returns(ppo_policy_t pol, ppo_env_t env, ppo_rollouts_t scratch, int64_t num_trajectories,
                            double* mean, double* std);

/* Evaluator variants (test/quad_game_utilities.jl:280-307,369-387), same trajectory-to-env assignment as
 * ppo_average_returns.  env.current_score of the synthetic env = sum of |vertex score| over the active quads,
 * env.opt_score = |sum of vertex scores| (the two numbers its termination test compares).
 *   average_best_returns:       mean / sample std of  initial_score - min(score seen on the trajectory)      :280-307
 *   average_normalized_returns: the same divided by maxreturn = initial_score - opt_score; a trajectory whose
 *                               maxreturn is 0 counts 1.0 and is NOT played (:369-378)                        :369-387
 * ppo_evaluate_trajectories returns the per-trajectory values themselves (kind 1 = single_trajectory_return
 * src/evaluate.jl:1-16, 2 = best_single_trajectory_return, 3 = single_trajectory_normalized_return), env-major:
 * env n's trajectories (e = n, n + N, ...) are consecutive. */
int32_t ppo_average_best_returns(ppo_policy_t pol, ppo_env_t env, ppo_rollouts_t scratch, int64_t num_trajectories,
                                 double* mean, double* std);
int32_t ppo_average_normalized_returns(ppo_policy_t pol, ppo_env_t env, ppo_rollouts_t scratch,
                                       int64_t num_trajectories, double* mean, double* std);
int32_t ppo_evaluate_trajectories(ppo_policy_t pol, ppo_env_t env, ppo_rollouts_t scratch, int64_t num_trajectories,
                                  int32_t kind, double* values);

/* timing of the dominant kernels of the last ppo_train / ppo_collect_rollouts call, measured with
 * HIP events on the engine's stream (bench.py roofline leg) */
/* measurement helper: run the return scan `iters` times on device-resident synthetic [T,N] columns (no host
 * copies in the timed region) and report the average kernel time (HIP events).  K6 roofline leg of bench.py. */
int32_t ppo_profile_returns(int64_t T, int64_t N, double discount, int32_t iters, double* avg_ms);
int32_t ppo_profile_gae(int64_t T, int64_t N, double gamma, double lambda, int32_t iters, double* avg_ms);
int32_t ppo_profile_enable(int32_t on);
int32_t ppo_profile_get(const char* kernel_name, double* total_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif
