// ppo_internal.h -- shared internals of libppo_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <map>
#include "../../include/ppo_hip.h"

#define PPO_OUT 4   // actions per half-edge (test/quad_game_utilities.jl:39,95)
#define PPO_TPL 36  // template rows; F = 72
#define PPO_PACK_PAD (16 * 64 * 4)   // floats of zero tail padding behind each packed weight stream (prefetch over-read)

// ---------------------------------------------------------------- host-side error plumbing
void ppo_set_error(const std::string& msg);
hipStream_t ppo_stream();
int ppo_hip_fail(hipError_t e, const char* what, const char* file, int line);

#define HIP_TRY(expr)                                                        \
    do {                                                                     \
        hipError_t _e = (expr);                                              \
        if (_e != hipSuccess) return ppo_hip_fail(_e, #expr, __FILE__, __LINE__); \
    } while (0)
#define ARG_CHECK(cond, msg)                                                 \
    do {                                                                     \
        if (!(cond)) { ppo_set_error(std::string("AssertionError: ") + msg + " [" #cond "]"); return PPO_ERR_ARG; } \
    } while (0)
#define PPO_TRY(expr)                                                        \
    do { int32_t _s = (expr); if (_s != PPO_OK) return _s; } while (0)

// kernel timing registry (bench roofline leg)
struct ProfScope {
    const char* name; hipEvent_t e0, e1; bool on;
    ProfScope(const char* n);
    ~ProfScope();
};

// ---------------------------------------------------------------- device buffers
template <typename T>
struct DevBuf {
    T* p = nullptr; size_t n = 0;
    int32_t alloc(size_t count) {
        if (count <= n && p) return PPO_OK;
        release();
        if (count == 0) return PPO_OK;
        HIP_TRY(hipMalloc((void**)&p, count * sizeof(T)));
        n = count;
        return PPO_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
};

// ---------------------------------------------------------------- handles
struct ppo_env_s {
    int32_t kind, Q, H, A, V, F, max_actions;
    int32_t strict_sampling = 0;       // 1: a CDF residue landing on a masked action is an error (reference @assert)
    float no_action_reward;
    int64_t N, global_offset;
    uint64_t seed;
    DevBuf<int8_t> score, degree;      // [N][V]
    DevBuf<uint32_t> active;           // [N]
    DevBuf<int32_t> steps;             // [N]
    DevBuf<float> reward;              // [N]
    DevBuf<uint8_t> done;              // [N]
    DevBuf<uint32_t> episode, tick;    // [N]
    DevBuf<int32_t> err;               // [1] OR of error flags
    DevBuf<int32_t> actions_tmp;       // [N]
    DevBuf<int8_t> obs_tmp;            // [N][H][F]
    DevBuf<int32_t> episodes_left;     // [N] (episodes mode)
    DevBuf<int8_t> tmpl;               // [H][36] template vertex ids (env_template), -1 = missing: looked up by k_env_observe
};

// k-slot order of the packed W2^T fragments (backward: dH1^T = W2^T dZ2^T).  Row form: component e of fragment group g,
// lane half hh <-> contraction feature 8g + 4hh + e, so the B operands of four MFMAs are ONE 16-byte LDS read of a
// row-major dZ2 tile; else feature 8g + 2e + hh against a feature-major tile, one ds_read_b32 per MFMA (round-1 form).
// Measured (gpurun_out/r2x, alternating): HID = 128 backward 0.1170 -> 0.1156 ms with the row form; at HID = 256 it is
// 0.3 % SLOWER (0.3423 -> 0.3432: LDS instructions beside an fp32 MFMA are not what that kernel waits for, and phase C
// then needs four 4-byte reads per A operand), so the row form is used up to this width only.  Shared by ppo_optim.hip
// (packing), ppo_policy_bwd.hip and ppo_policy_bwd_small.hip.
#ifndef PPO_BWD_Z2ROW_MAX_HID
#define PPO_BWD_Z2ROW_MAX_HID 128
#endif
#define PPO_BWD_Z2ROW_AT(HID) ((HID) <= PPO_BWD_Z2ROW_MAX_HID)

struct ppo_policy_s {
    int32_t F, HID, L, OUT;            // HID: the width the kernels run (128 or 256); L = hidden layers (test/policy.jl:9-19):
                                       // Dense(F,HID) + (L-1) x Dense(HID,HID) + Dense(HID,OUT).  L == 2 runs the fused
                                       // kernels; L in {1, 3, 4} the layer-looped ("deep") forms of the same kernels
    int32_t hid_user = 0;              // hidden_channels the caller asked for (<= HID): the missing units are zero-padded
    int64_t np_user = 0;               // parameter count of the caller's Policy (what crosses the ABI)
    int64_t np;                        // parameter count at width HID (device buffers, all-reduce)
    // canonical flat parameters (Flux order) + packed MFMA-fragment copies
    DevBuf<float> params;              // [np]
    DevBuf<float> w1p, w2p, w2tp;      // A-operand fragment order; w2p / w2tp: the L-1 hidden->hidden layers back to back (HID*HID each)
    DevBuf<float> b1p, b2p, w3p, b3;   // accumulator-init / VALU packs; b2p: L-1 x HID
    // bf16 compute mode (ppo_policy_set_dtype): bf16 fragment streams of the same parameters, rewritten by k_adam
    int32_t dtype = 0;                 // PPO_DTYPE_F32 / PPO_DTYPE_BF16
    DevBuf<uint16_t> w1b, w2b, w2tb;   // [HID/32][KS][64][8] A-operand fragments of v_mfma_f32_32x32x16_bf16
    DevBuf<uint16_t> w3c, w3tb;        // layer 3 forward (compact rows 0..3) / backward ([HID][4])
    // split-fp32 backward (ppo_policy_bwd_x6.hip): W2 as three bf16 pieces, B-operand fragments of dH1 = dZ2 W2,
    // [in-feature tile][k-step][piece: lo, mid, hi][64 lanes][8]; L == 2, F == 72 only; rewritten by k_adam
    DevBuf<uint16_t> w2x;
    // split-fp32 train forward (ppo_policy_fwd_x6.hip): A-operand piece fragments [feature tile][k-step][piece][64][8] of
    // layer 1 (5 k-steps of natural input order, zero padded from 72 to 80) and layer 2 (k order of packed accumulators)
    DevBuf<uint16_t> w1x, w2fx;
    DevBuf<float> grad;                // [np + 2]  (+ ppo sum, entropy sum)
    // training workspace
    DevBuf<float> act1, act2;          // saved activations, D-fragment order [tiles][HID/32][4][64] float4: the FIRST and the LAST hidden layer
    DevBuf<float> actm;                // L > 2: the L-2 hidden layers in between, [L-2][tiles]... back to back
    DevBuf<float> dz2f, dz1f;          // three-product backward: dZ of the last / first hidden layer in fragment order (like act2 / act1)
    DevBuf<float> dzm;                 // L > 2: dZ of the layers in between, like actm
    DevBuf<float> dY;                  // [tiles][32][4]
    DevBuf<double> loss_terms;         // [tiles][2]
    DevBuf<float> slabs;               // [nwg][slab]
    DevBuf<int32_t> idx;               // gathered transition ids of the minibatch
    DevBuf<int8_t> xs;                 // [tiles][32][F] state rows of the minibatch re-derived by the forward (compact rollouts)
    DevBuf<float> adv_col;             // batch_advantage scratch column [T*N] (PPO_ADV_RETURNS_NORMALISED)
    DevBuf<int32_t> err;               // device error flag
    int64_t cap_tiles = 0;
    int32_t nwg_bwd = 0;               // slabs holding weight-gradient partials of the last backward
    int32_t nwg_small = 0;             // slabs holding its small-gradient tails (0: the same slabs)
    int64_t last_B = 0;
    double last_entropy_weight = 0.0;
};

struct ppo_adam_s {
    ppo_policy_s* pol;
    double eta, beta1, beta2, eps;
    double beta_pow[2];
    int64_t epochs_done = 0;           // epochs trained through ppo_train: keys the minibatch permutation with the seed
    DevBuf<float> m, v;
};

struct DiskSink;   // ppo_disk.hip

struct ppo_rollouts_s {
    int64_t N, capT, T;    // T = steps currently held
    int32_t H, F, A;
    int64_t len;           // valid transitions
    // state storage, one of two forms (ppo_set_rollout_compact):
    //   expanded: the observation rows themselves, 2304 B per transition for Q = 8 (host-supplied rollouts always)
    //   compact:  the env snapshot the rows are derived from (score[V] then degree[V], int8: 64 B for Q = 8) -- the
    //             train forward re-derives the rows like the persistent rollout does (k_policy_fwd MODE 4)
    bool compact = false;
    int32_t V = 0;             // vertices per env (4Q)
    DevBuf<int8_t> states;     // [T][N][H][F]        (expanded form)
    DevBuf<int8_t> cstate;     // [T][N][2V]          (compact form)
    DevBuf<int8_t> tmpl;       // [H][36] template vertex ids of the env the buffer was created for
    DevBuf<int8_t> expand_tmp; // getters of the compact form: expanded observations, built on demand
    DevBuf<uint32_t> active;   // [T][N]
    DevBuf<int32_t> actions;   // [T][N]
    DevBuf<float> p_sel;       // [T][N]
    DevBuf<float> rewards;     // [T][N] raw
    DevBuf<float> returns;     // [T][N]
    DevBuf<uint8_t> done;      // [T][N]
    DevBuf<uint8_t> valid;     // [T][N]
    DevBuf<int32_t> index;     // [len] transition ids in dataset order
    DevBuf<float> full_probs;  // [T][N][A] optional
    DevBuf<float> values;      // [T+1][N] host-supplied state values (ppo_rollouts_compute_gae)
    DevBuf<float> adv;         // [T][N] GAE(gamma, lambda) advantages (PPO_ADV_GAE*)
    DevBuf<float> lam_ret;     // [T][N] lambda-returns adv + V
    int64_t adv_T = -1;        // T the adv column was computed for (-1: none)
    bool all_valid = true;
    DiskSink* sink = nullptr;  // optional out-of-core store (ppo_rollouts_attach_disk)
    ~ppo_rollouts_s();
};

// rollout buffer internals shared with the disk loader (ppo_api.hip)
extern "C" int32_t rollouts_reserve(ppo_rollouts_s* r, int64_t T, bool compact);
extern "C" int32_t set_index_all(ppo_rollouts_s* r);

// out-of-core store hooks used by ppo_collect_rollouts (ppo_disk.hip)
int32_t disk_sink_begin(ppo_rollouts_s* ro, int64_t T);
int disk_sink_slots(const ppo_rollouts_s* ro);                // pinned records in the ring (0: no sink)
int disk_sink_chunk(const ppo_rollouts_s* ro);                // steps per persistent launch of a streamed collection
int32_t disk_sink_step(ppo_rollouts_s* ro, int64_t t);       // after the kernels of step t were enqueued
int32_t disk_sink_finish(ppo_rollouts_s* ro);                // after the return scan: appends returns, flushes
void disk_sink_destroy(DiskSink* s);

// ---------------------------------------------------------------- packed layout sizes
// one gradient slab: [dW of the L-1 hidden->hidden layers][dW1, input padded to 32][db1][db of the L-1 layers][dW3][db3 + pad]
// (L == 2: exactly the round-1 layout W2, W1, b1, b2, W3, b3)
static inline size_t slab_floats(int F, int HID, int L = 2) {
    const int FP = ((F + 31) / 32) * 32;
    return (size_t)(L - 1) * HID * HID + (size_t)HID * FP + (size_t)HID * L + (size_t)HID * PPO_OUT + 64;
}

// ---------------------------------------------------------------- kernel launchers (defined in the .hip files)
int32_t launch_returns_tn(const float* r, const uint8_t* done, float* out, int64_t T, int64_t N,
                          double discount, int f32mode);
int32_t launch_returns_flat(const float* r, const uint8_t* term, float* out, int64_t n, double discount, int f32mode);
int32_t launch_gae_tn(const float* r, const uint8_t* done, const float* values, float* adv, float* ret,
                      int64_t T, int64_t N, double gamma, double lambda);

int32_t launch_env_reset(ppo_env_s* e, int only_done);
// evaluator bookkeeping folded into the env step of the episodes mode (test/quad_game_utilities.jl:280-307,369-387,
// src/evaluate.jl:1-25): per-env running values of the episode in flight, one result per finished episode
struct EvalView {
    int32_t kind = 0;              // 0 off, 1 return (sum of rewards), 2 best return (initial score - lowest score seen),
                                   // 3 normalised best return (best / (initial score - optimum score); 1.0 when that is 0)
    double* ep_ret = nullptr;      // [N] reward sum of the episode in flight
    int32_t* ep_init = nullptr;    // [N] score at its reset
    int32_t* ep_min = nullptr;     // [N] lowest score seen so far
    int32_t* ep_maxret = nullptr;  // [N] initial score - optimum score
    int32_t* ep_count = nullptr;   // [N] episodes this env has finished
    double* out = nullptr;         // [num_traj] results, env-major (env n's episodes are consecutive)
    int64_t num_traj = 0;
};
int32_t launch_env_step(ppo_env_s* e, const int32_t* actions_dev, float* reward_out, uint8_t* done_out,
                        uint8_t* valid_out, int auto_reset, int episodes_mode, const EvalView* ev = nullptr);
int32_t launch_env_observe(ppo_env_s* e, int8_t* obs_out, uint32_t* active_out);
// compact rollout storage: env snapshot of every env (score[V] then degree[V]) -> cstate_out [N][2V]
int32_t launch_env_snapshot(ppo_env_s* e, int8_t* cstate_out);
// snapshots -> observations (the same arithmetic as state(env)): count records of 2V bytes -> [count][H][F]
// idx (optional): output record n = transition idx[n] (a gathered minibatch) instead of record n
int32_t launch_expand_states(const int8_t* cstate, const uint32_t* active, const int8_t* tmpl, int64_t count, int32_t Q,
                             int8_t* obs_out, const int32_t* idx = nullptr);

int32_t launch_pack_params(ppo_policy_s* p);
int32_t launch_policy_probs(ppo_policy_s* p, const int8_t* states_dev, const uint32_t* active_dev, int64_t B,
                            int32_t H, float* probs_dev);
int32_t launch_policy_rollout(ppo_policy_s* p, ppo_env_s* e, const int8_t* states_dev, const uint32_t* active_dev,
                              int32_t* actions_out, float* psel_out, float* full_probs_or_null);
// adv_col: the advantage column indexed by transition id (ro->returns for PPO_ADV_RETURNS)
int32_t launch_policy_rollout_persistent(ppo_policy_s* p, ppo_env_s* e, ppo_rollouts_s* ro, int64_t T, int record_probs,
                                         int64_t t0 = 0);
int32_t launch_policy_train_fwd(ppo_policy_s* p, ppo_rollouts_s* ro, const int32_t* idx_dev, int64_t B,
                                int64_t B_global, double eps, double entropy_weight, const float* adv_col);
int32_t launch_adv_normalise(const float* returns, const int32_t* idx_dev, int64_t B, float* adv_col);
int32_t launch_policy_bwd(ppo_policy_s* p, ppo_rollouts_s* ro, const int32_t* idx_dev, int64_t B);
// the same fused backward with its row contractions (dW2, dW1) as split-fp32 products on the bf16 matrix pipe
// (ppo_policy_bwd_x6.hip); PPO_ERR_UNSUPPORTED: shape not covered
int32_t launch_policy_bwd_x6(ppo_policy_s* p, ppo_rollouts_s* ro, const int32_t* idx_dev, int64_t B);
extern "C" int ppo_bwd_split_enabled();
// three-product backward for small minibatches (ppo_policy_bwd_small.hip); PPO_ERR_UNSUPPORTED: not covered
int32_t launch_policy_bwd_small(ppo_policy_s* p, ppo_rollouts_s* ro, const int32_t* idx_dev, int64_t B);
// small minibatches: forward + loss + backward-data of a tile in one workgroup, then the weight-gradient kernel
// (ppo_policy_train_tile.hip); PPO_ERR_UNSUPPORTED: not covered
int32_t launch_policy_train_tile(ppo_policy_s* p, ppo_rollouts_s* ro, const int32_t* idx_dev, int64_t B, int64_t B_global,
                                 double eps, double entropy_weight, const float* adv_col);
// bf16 compute mode (ppo_policy_bf16.hip); MODE as in k_policy_fwd: 0 probs, 1 rollout, 2 train
struct FwdArgs;
// one-launch rollout with 2 or 4 waves per env for few envs, bit-identical results (ppo_policy_rollout_split.hip)
int32_t launch_rollout_split(ppo_policy_s* p, FwdArgs& a, int64_t N, int tps, int V, int env);
// train forward with 2 or 4 waves per state for small minibatches (ppo_policy_fwd_split.hip)
int32_t launch_policy_train_fwd_split(ppo_policy_s* p, FwdArgs& a, int64_t B, int tps, bool compact);
// train forward with its Dense products as split-fp32 MFMAs (ppo_policy_fwd_x6.hip); PPO_ERR_UNSUPPORTED: not covered / switched off
int32_t launch_policy_train_fwd_x6(ppo_policy_s* p, FwdArgs& a, int64_t B, int tps, bool compact);
int32_t launch_policy_fwd_bf16(ppo_policy_s* p, FwdArgs& args, int mode, int64_t B, int tps);
int32_t launch_policy_rollout_persistent_bf16(ppo_policy_s* p, FwdArgs& args, int64_t N, int tps, int V);
int32_t launch_policy_bwd_bf16(ppo_policy_s* p, ppo_rollouts_s* ro, const int32_t* idx_dev, int64_t B);
static inline int bf16_ks1(int F) { return (F + 15) / 16; }     // layer-1 k-steps of 16 (zero padded)
// fuse (optional): apply Adam + re-pack in the same launch (single-rank training); hist2: the per-batch loss pair of that step
int32_t launch_grad_reduce(ppo_policy_s* p, int64_t B, int64_t B_global, double entropy_weight, ppo_adam_s* fuse = nullptr, float* hist2 = nullptr);
int32_t launch_adam(ppo_adam_s* o, float* hist2_or_null);
int32_t launch_categorical(const float* probs, const float* u, int64_t B, int64_t A, int32_t* actions, float* psel,
                           int32_t* err);
int32_t launch_feistel_index(const int32_t* index_dev, int64_t len, uint64_t seed, uint32_t epoch, int32_t* out_dev);
