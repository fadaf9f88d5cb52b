/*
 * ppo_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE ONLY).
 *
 * Plain-C restatement of the ProximalPolicyOptimization.jl hot path
 * (collect_rollouts! / rollout_buffer / train.jl) used ONLY as the checker in
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing
 * under proximalpolicyoptimization.jl_amd/ may include, link or call this.
 *
 * Parity status: the Julia reference cannot run in the build container (no
 * julia binary, empty reference test-suite).  The oracle is pinned by the
 * reference's own known answers (tests/golden/, see tests/test_oracle_golden.py):
 *   - returns:  /root/reference/output/trajectory.csv:1-7 and
 *               /root/reference/test/test_rollout_buffer.jl:41-50
 *   - action decode / mask pattern / masked-softmax zeros: SURVEY.md 8(c) (4)-(6)
 * Flux forward/backward, Adam and Categorical sampling have NO reference-held
 * vectors: for those the header of each function says "parity unpinned" and the
 * restatement follows the published upstream semantics.
 *
 * Every function cites the reference file:line it restates.
 */
#ifndef PPO_ORACLE_H
#define PPO_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_OUT 4          /* actions per half-edge  (test/quad_game_utilities.jl:39,95) */
#define ORC_TPL 36         /* template rows          (level-4 template)    */

/* ---------------- returns / GAE (src/collect_rollouts.jl:26-42) ---------------- */
void orc_compute_returns(const float* rewards, const uint8_t* terminal, int64_t n,
                         double discount, int32_t discount_is_f32, float* out);
/* time-major [T,N] columns, v=0 at the column tail (vectorised layout of the same scan) */
void orc_compute_returns_tn(const float* rewards, const uint8_t* done, int64_t T, int64_t N,
                            double discount, int32_t discount_is_f32, float* out);
/* GAE(gamma,lambda) extension; values is [T+1,N]; lambda=1,V=0 == returns */
void orc_gae_tn(const float* rewards, const uint8_t* done, const float* values, int64_t T,
                int64_t N, double gamma, double lambda, float* adv_out, float* ret_out);

/* ---------------- counter RNG (Philox4x32-10, Salmon et al. SC'11) ---------------- */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
float orc_u01(uint32_t w); /* (w>>8) * 2^-24  in [0,1) */
/* Feistel bijection on [0,n): minibatch permutation used when no explicit permutation
 * is supplied (stands in for randperm, src/train.jl:93) */
int64_t orc_feistel_perm(int64_t i, int64_t n, uint64_t seed, uint32_t epoch);

/* ---------------- synthetic rand-poly-shaped env (plugin contract:
 *   src/ProximalPolicyOptimization.jl:16-20; shapes test/quad_game_utilities.jl:39-59,95-110) */
typedef struct {
    int32_t Q, H, A, V, F;      /* quad slots, half-edges=4Q, actions=16Q, vertices=4Q, features=72 */
    int32_t max_actions;
    float no_action_reward;
    int64_t N;                  /* envs in this shard */
    int64_t global_offset;      /* global id of env 0 (multi-GPU sharding) */
    uint64_t seed;
    int8_t* score;              /* [N][V] */
    int8_t* degree;             /* [N][V] */
    uint32_t* active;           /* [N] */
    int32_t* steps;             /* [N] */
    float* reward;              /* [N] */
    uint8_t* done;              /* [N] */
    uint32_t* episode;          /* [N] resets so far */
    uint32_t* tick;             /* [N] env steps so far (sampling counter) */
    int32_t* err;               /* [N] error flags */
} orc_env;

orc_env* orc_env_create(int32_t Q, int32_t max_actions, float no_action_reward, int64_t N,
                        int64_t global_offset, uint64_t seed);
void orc_env_destroy(orc_env* e);
void orc_env_reset_one(orc_env* e, int64_t n);
void orc_env_reset(orc_env* e);
void orc_env_step_one(orc_env* e, int64_t n, int32_t action); /* 0-based action */
void orc_env_observe_one(const orc_env* e, int64_t n, int8_t* obs /*[H][F]*/);
int32_t orc_env_template(int32_t Q, int32_t h, int32_t t); /* vertex id or -1 */
/* (quad,edge,type) 1-based decode of a 1-based index: test/quad_game_utilities.jl:95-105 */
void orc_index_to_action(int32_t index1, int32_t actions_per_edge, int32_t* quad, int32_t* edge, int32_t* type);
void orc_index_to_action_edges(int32_t index1, int32_t edges, int32_t actions_per_edge, int32_t* elem, int32_t* edge,
                               int32_t* type);
/* action mask over A entries: 0 or -Inf (test/quad_game_utilities.jl:39-44) */
void orc_action_mask(const uint8_t* active_quad, int32_t Q, int32_t actions_per_edge, float* mask_out);

/* ---------------- policy MLP (test/policy.jl:9-31, test/quad_game_utilities.jl:65-79) --------- */
/* flat params in Flux order (W1,b1,W2,b2,...,Wout,bout), W is [out,in] column-major */
int64_t orc_mlp_num_params(int32_t F, int32_t HID, int32_t n_hidden);
/* reference-order fp32: y_i = sum_k W[i,k] x_k (k ascending), + b, leakyrelu(0.01) */
void orc_mlp_logits_ref(const float* params, int32_t F, int32_t HID, int32_t n_hidden,
                        const int8_t* x /*[H][F]*/, int32_t H, float* logits /*[H*4] type fastest*/);
void orc_mlp_logits_f64(const float* params, int32_t F, int32_t HID, int32_t n_hidden,
                        const int8_t* x, int32_t H, double* logits);
/* device-order fp32: the exact fmaf chain order of the gfx950 MFMA kernels (orc_mlp_logits_dev: n_hidden == 2) */
void orc_mlp_logits_dev_n(const float* params, int32_t F, int32_t HID, int32_t n_hidden,
                          const int8_t* x, int32_t H, float* logits);
void orc_mlp_logits_dev(const float* params, int32_t F, int32_t HID,
                        const int8_t* x, int32_t H, float* logits);
/* softmax(logits + mask) over A = 4H entries; mask from active-quad bits (quad = a/16) */
void orc_masked_softmax_ref(const float* logits, uint32_t active, int32_t A, float* probs);
void orc_masked_softmax_dev(const float* logits, uint32_t active, int32_t A, float* probs);
float orc_exp_dev(float x);
/* rand(Categorical(p)) given u in [0,1): sequential fp32 inverse CDF (src/collect_rollouts.jl:6);
 * returns 0-based index; *err=1 if p[idx]==0 (src/collect_rollouts.jl:7) */
int32_t orc_categorical_sample(const float* p, int32_t A, float u, int32_t* err);

/* ---------------- rollout (src/collect_rollouts.jl:1-24, src/rollout_buffer.jl:24-79) -------- */
/* fixed-T vectorised collection with auto-reset.  Outputs time-major [T,N]. mode_dev!=0 uses the
 * device-order forward so results are bit-comparable with the HIP engine. */
void orc_collect_rollouts_tn(orc_env* e, const float* params, int32_t HID, int32_t n_hidden,
                             int64_t T, int32_t mode_dev,
                             int8_t* states /*[T,N,H,F]*/, uint32_t* active /*[T,N]*/,
                             float* p_sel, int32_t* actions, float* rewards, uint8_t* done);

/* ---------------- loss + gradient (src/train.jl:1-84) ---------------- */
/* get_linear_action_index: 1-based a + (0:A:(B-1)A)  (src/train.jl:48-52) */
void orc_linear_action_index(const int64_t* a1, int64_t B, int64_t A, int64_t* out);
double orc_simplified_ppo_clip(double adv, double eps); /* src/train.jl:1-7 */
/* forward-only on given probs [A,B] column-major; fp32 probs, Float64 loss like the reference
 * when epsilon is Float64 (src/train.jl:35-46) */
void orc_ppo_loss_with_entropy(const float* probs, const int64_t* lin_idx1, const float* p_old,
                               const float* adv, int64_t B, int64_t A, double eps,
                               double* ppoloss, double* entropyloss);
/* full minibatch forward+backward in float64 ("truth" for gradient parity; parity unpinned vs
 * Zygote -- cross-checked against torch autograd in tests/test_oracle_crosscheck.py).
 * grads_out is flat Flux order, double.  losses: ppoloss and entropy_weight*entropyloss. */
void orc_step_batch_grad_f64(const float* params, int32_t F, int32_t HID, int32_t n_hidden,
                             const int8_t* states /*[B,H,F]*/, const uint32_t* active /*[B]*/,
                             const int32_t* actions0 /*[B] 0-based*/, const float* p_old,
                             const float* adv, int64_t B, int32_t H, double eps,
                             double entropy_weight, double* grads_out,
                             double* ppoloss, double* entropyloss);
/* Flux legacy Adam: Float32 state, Float64 hyper-parameters; beta_pow = {b1^t,b2^t} */
void orc_adam_step(float* params, const float* grad, float* m, float* v, double* beta_pow,
                   int64_t n, double eta, double beta1, double beta2, double eps);

#ifdef __cplusplus
}
#endif
#endif
