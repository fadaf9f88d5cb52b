/*
 * ppo_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; see ppo_oracle.h).
 * Plain scalar C restatement of the reference path.  Build: make -C oracle
 * (gcc -O2 -ffp-contract=off: no implicit FMA contraction, explicit fmaf only).
 */
#include "ppo_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ===================================================================== returns */

/* src/collect_rollouts.jl:26-42.  `v` starts as zero(T)=Float32; `discount*v` promotes to
 * Float64 when discount is a Float64 (every call site in the reference passes 1.0/0.99
 * literals, i.e. Float64), so the running value is Float64 until the next terminal resets it
 * to Float32 zero; only the STORED value is rounded to Float32 (values[idx] = v). */
void orc_compute_returns(const float* rewards, const uint8_t* terminal, int64_t n,
                         double discount, int32_t discount_is_f32, float* out) {
    if (discount_is_f32) {
        float g = (float)discount, v = 0.0f;
        for (int64_t i = n - 1; i >= 0; --i) {
            if (terminal[i]) v = 0.0f;
            float gv = g * v;          /* Float32*Float32 */
            v = rewards[i] + gv;
            out[i] = v;
        }
    } else {
        double v = 0.0;
        for (int64_t i = n - 1; i >= 0; --i) {
            if (terminal[i]) v = 0.0;
            double gv = discount * v;
            v = (double)rewards[i] + gv;
            out[i] = (float)v;
        }
    }
}

void orc_compute_returns_tn(const float* rewards, const uint8_t* done, int64_t T, int64_t N,
                            double discount, int32_t discount_is_f32, float* out) {
    for (int64_t n = 0; n < N; ++n) {
        if (discount_is_f32) {
            float g = (float)discount, v = 0.0f;
            for (int64_t t = T - 1; t >= 0; --t) {
                if (done[t * N + n]) v = 0.0f;
                float gv = g * v;
                v = rewards[t * N + n] + gv;
                out[t * N + n] = v;
            }
        } else {
            double v = 0.0;
            for (int64_t t = T - 1; t >= 0; --t) {
                if (done[t * N + n]) v = 0.0;
                double gv = discount * v;
                v = (double)rewards[t * N + n] + gv;
                out[t * N + n] = (float)v;
            }
        }
    }
}

/* GAE extension (no reference counterpart; lambda=1, V=0 reduces to orc_compute_returns_tn).
 * delta_t = r_t + gamma*V_{t+1}*(1-done_t) - V_t ; A_t = delta_t + gamma*lambda*(1-done_t)*A_{t+1}
 * fp64 running values, fp32 stores. */
void orc_gae_tn(const float* rewards, const uint8_t* done, const float* values, int64_t T,
                int64_t N, double gamma, double lambda, float* adv_out, float* ret_out) {
    for (int64_t n = 0; n < N; ++n) {
        double a = 0.0;
        for (int64_t t = T - 1; t >= 0; --t) {
            double nd = done[t * N + n] ? 0.0 : 1.0;
            double vnext = (double)values[(t + 1) * N + n] * nd;
            double gvn = gamma * vnext;
            double delta = ((double)rewards[t * N + n] + gvn) - (double)values[t * N + n];
            double gl = gamma * lambda;
            double carry = (gl * nd) * a;
            a = delta + carry;
            adv_out[t * N + n] = (float)a;
            ret_out[t * N + n] = (float)(a + (double)values[t * N + n]);
        }
    }
}

/* ===================================================================== RNG */

static inline void mulhilo(uint32_t a, uint32_t b, uint32_t* hi, uint32_t* lo) {
    uint64_t p = (uint64_t)a * (uint64_t)b;
    *hi = (uint32_t)(p >> 32);
    *lo = (uint32_t)p;
}

/* Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11).  Known-answer vectors from the paper's
 * reference implementation are checked in tests/test_oracle_golden.py. */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        mulhilo(0xD2511F53u, c0, &hi0, &lo0);
        mulhilo(0xCD9E8D57u, c2, &hi1, &lo1);
        uint32_t n0 = hi1 ^ c1 ^ k0;
        uint32_t n1 = lo1;
        uint32_t n2 = hi0 ^ c3 ^ k1;
        uint32_t n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

float orc_u01(uint32_t w) { return (float)(w >> 8) * (1.0f / 16777216.0f); }

static inline uint32_t feistel_round_fn(uint32_t x, uint32_t k) {
    x = x * 0x9E3779B1u + k;
    x ^= x >> 15; x *= 0x85EBCA77u;
    x ^= x >> 13; x *= 0xC2B2AE3Du;
    x ^= x >> 16;
    return x;
}

int64_t orc_feistel_perm(int64_t i, int64_t n, uint64_t seed, uint32_t epoch) {
    if (n <= 1) return 0;
    int bits = 2;
    while (((int64_t)1 << bits) < n) bits += 2;      /* even number of bits */
    int hb = bits / 2;
    uint32_t hmask = (uint32_t)(((uint64_t)1 << hb) - 1);
    uint64_t x = (uint64_t)i;
    do {
        uint32_t L = (uint32_t)(x >> hb) & hmask, R = (uint32_t)x & hmask;
        for (uint32_t r = 0; r < 6; ++r) {
            uint32_t k = (uint32_t)(seed >> ((r & 1) ? 32 : 0)) ^ (epoch * 0x9E3779B9u) ^ (r * 0x7F4A7C15u);
            uint32_t nL = R;
            uint32_t nR = L ^ (feistel_round_fn(R, k) & hmask);
            L = nL; R = nR;
        }
        x = ((uint64_t)L << hb) | R;
    } while ((int64_t)x >= n);                      /* cycle-walk into [0,n) */
    return (int64_t)x;
}

/* ===================================================================== synthetic env */

orc_env* orc_env_create(int32_t Q, int32_t max_actions, float no_action_reward, int64_t N,
                        int64_t global_offset, uint64_t seed) {
    orc_env* e = (orc_env*)calloc(1, sizeof(orc_env));
    e->Q = Q; e->H = 4 * Q; e->A = 16 * Q; e->V = 4 * Q; e->F = 2 * ORC_TPL;
    e->max_actions = max_actions; e->no_action_reward = no_action_reward;
    e->N = N; e->global_offset = global_offset; e->seed = seed;
    e->score = (int8_t*)calloc((size_t)N * e->V, 1);
    e->degree = (int8_t*)calloc((size_t)N * e->V, 1);
    e->active = (uint32_t*)calloc((size_t)N, 4);
    e->steps = (int32_t*)calloc((size_t)N, 4);
    e->reward = (float*)calloc((size_t)N, 4);
    e->done = (uint8_t*)calloc((size_t)N, 1);
    e->episode = (uint32_t*)calloc((size_t)N, 4);
    e->tick = (uint32_t*)calloc((size_t)N, 4);
    e->err = (int32_t*)calloc((size_t)N, 4);
    return e;
}

void orc_env_destroy(orc_env* e) {
    if (!e) return;
    free(e->score); free(e->degree); free(e->active); free(e->steps); free(e->reward);
    free(e->done); free(e->episode); free(e->tick); free(e->err); free(e);
}

/* reset!(env): src/ProximalPolicyOptimization.jl:19 contract; synthetic initial state:
 * the first 3Q/4 quad slots active, vertex scores uniform in [-2,2], desired degree 3 or 4. */
void orc_env_reset_one(orc_env* e, int64_t n) {
    const int V = e->V, Q = e->Q;
    const int nact = (3 * Q) / 4;
    int8_t* sc = e->score + n * V;
    int8_t* dg = e->degree + n * V;
    uint64_t g = (uint64_t)(e->global_offset + n);
    uint32_t key[2] = {(uint32_t)e->seed, (uint32_t)(e->seed >> 32)};
    for (int q = 0; q < Q; ++q) {
        uint32_t ctr[4] = {(uint32_t)g, e->episode[n], 1u, (uint32_t)q}, w[4];
        orc_philox4x32_10(ctr, key, w);
        for (int i = 0; i < 4; ++i) {
            int v = 4 * q + i;
            if (q < nact) {
                int s = (int)(w[i] % 5u) - 2;
                int desired = 3 + (int)((w[i] >> 8) & 1u);
                sc[v] = (int8_t)s;
                dg[v] = (int8_t)(desired - s);
            } else { sc[v] = 0; dg[v] = 0; }
        }
    }
    e->active[n] = (nact >= 32) ? 0xFFFFFFFFu : ((1u << nact) - 1u);
    e->steps[n] = 0; e->reward[n] = 0.0f; e->done[n] = 0;
    e->episode[n] += 1u;
}

void orc_env_reset(orc_env* e) { for (int64_t n = 0; n < e->N; ++n) orc_env_reset_one(e, n); }

int32_t orc_env_template(int32_t Q, int32_t h, int32_t t) {
    int V = 4 * Q, q = h / 4, ed = h % 4;
    if (t < 4) return 4 * q + (ed + t) % 4;
    int c = (h * 5 + t * 7 + 3) % (V + 6);
    return c >= V ? -1 : c;
}

/* state(env): src/ProximalPolicyOptimization.jl:16; matrix = vcat(vertex_score[template],
 * degree[template]) with 0 for missing entries (test/quad_game_utilities.jl:35-37,46-59);
 * stored row-major per half-edge: obs[h][f], f<36 scores, f>=36 degrees. */
void orc_env_observe_one(const orc_env* e, int64_t n, int8_t* obs) {
    const int V = e->V, H = e->H, F = e->F;
    const int8_t* sc = e->score + n * V;
    const int8_t* dg = e->degree + n * V;
    uint32_t act = e->active[n];
    for (int h = 0; h < H; ++h) {
        int own_active = (act >> (h / 4)) & 1u;
        for (int t = 0; t < ORC_TPL; ++t) {
            int v = orc_env_template(e->Q, h, t);
            int ok = own_active && v >= 0 && ((act >> (v / 4)) & 1u);
            obs[h * F + t] = ok ? sc[v] : 0;
            obs[h * F + ORC_TPL + t] = ok ? dg[v] : 0;
        }
    }
}

static int total_abs(const int8_t* sc, uint32_t act, int Q) {
    int s = 0;
    for (int q = 0; q < Q; ++q) if ((act >> q) & 1u)
        for (int i = 0; i < 4; ++i) { int x = sc[4 * q + i]; s += x < 0 ? -x : x; }
    return s;
}
static int total_sum(const int8_t* sc, uint32_t act, int Q) {
    int s = 0;
    for (int q = 0; q < Q; ++q) if ((act >> q) & 1u)
        for (int i = 0; i < 4; ++i) s += sc[4 * q + i];
    return s;
}
static int deg_ok(int d) { return d >= 2 && d <= 7; }

/* step!(env, action): src/ProximalPolicyOptimization.jl:20; decode test/quad_game_utilities.jl:95-105
 * (0-based here); invalid move => reward = no_action_reward, state unchanged (:151,175). */
void orc_env_step_one(orc_env* e, int64_t n, int32_t a) {
    const int V = e->V, Q = e->Q;
    int8_t* sc = e->score + n * V;
    int8_t* dg = e->degree + n * V;
    uint32_t act = e->active[n];
    e->tick[n] += 1u;
    if (e->done[n]) { e->err[n] |= 4; return; }            /* stepping a terminated env */
    if (a < 0 || a >= e->A) { e->err[n] |= 2; a = 0; }      /* invalid index */
    int q = a / 16, ed = (a % 16) / 4, type = a % 4;
    int old_total = total_abs(sc, act, Q);
    int valid = 0;
    if (!((act >> q) & 1u)) {
        e->err[n] |= 1;                                    /* acting on an inactive quad */
    } else {
        int v0 = 4 * q + ed, v1 = 4 * q + (ed + 1) % 4, v2 = 4 * q + (ed + 2) % 4, v3 = 4 * q + (ed + 3) % 4;
        int nq = (q + 1 + ed) % Q;
        int w0 = 4 * nq + ed, w1 = 4 * nq + (ed + 1) % 4;
        int nq_ok = (nq != q) && ((act >> nq) & 1u);
        if (type == 0 || type == 1) {
            int p = (type == 0) ? v3 : v2, r = (type == 0) ? w0 : w1;
            if (nq_ok && deg_ok(dg[v0] - 1) && deg_ok(dg[v1] - 1) && deg_ok(dg[p] + 1) && deg_ok(dg[r] + 1)) {
                dg[v0]--; sc[v0]++; dg[v1]--; sc[v1]++;
                dg[p]++; sc[p]--; dg[r]++; sc[r]--;
                valid = 1;
            }
        } else if (type == 2) {
            int f = -1;
            for (int s = 0; s < Q; ++s) if (!((act >> s) & 1u)) { f = s; break; }
            if (f >= 0 && deg_ok(dg[v0] + 1) && deg_ok(dg[v2] + 1)) {
                dg[v0]++; sc[v0]--; dg[v2]++; sc[v2]--;
                for (int i = 0; i < 4; ++i) { sc[4 * f + i] = 0; dg[4 * f + i] = 4; }
                act |= (1u << f);
                valid = 1;
            }
        } else {
            int cnt = 0;
            for (int s = 0; s < Q; ++s) cnt += (act >> s) & 1u;
            if (nq_ok && cnt > Q / 2 && deg_ok(dg[w0] - 1) && deg_ok(dg[w1] - 1)) {
                dg[w0]--; sc[w0]++; dg[w1]--; sc[w1]++;
                for (int i = 0; i < 4; ++i) { sc[4 * q + i] = 0; dg[4 * q + i] = 0; }
                act &= ~(1u << q);
                valid = 1;
            }
        }
    }
    e->active[n] = act;
    int new_total = total_abs(sc, act, Q);
    e->reward[n] = valid ? (float)(old_total - new_total) : e->no_action_reward;
    e->steps[n] += 1;
    int sum = total_sum(sc, act, Q);
    int opt = sum < 0 ? -sum : sum;
    e->done[n] = (uint8_t)((new_total == opt) || (e->steps[n] >= e->max_actions));
}

/* index_to_action for elements with `edges` edges (4: the quad game, test/quad_game_utilities.jl:95-105;
 * 3: the triangle variant of the tutorial notebook, whose printed answers pin it: learn_flip.ipynb:19669-19699) */
void orc_index_to_action_edges(int32_t index1, int32_t edges, int32_t actions_per_edge, int32_t* elem, int32_t* edge,
                               int32_t* type) {
    int ape = edges * actions_per_edge;            /* actions per element */
    *elem = (index1 - 1) / ape + 1;
    int ea = (index1 - 1) % ape;
    *edge = ea / actions_per_edge + 1;
    *type = ea % actions_per_edge + 1;
}

void orc_index_to_action(int32_t index1, int32_t actions_per_edge, int32_t* quad, int32_t* edge, int32_t* type) {
    orc_index_to_action_edges(index1, 4, actions_per_edge, quad, edge, type);
}

void orc_action_mask(const uint8_t* active_quad, int32_t Q, int32_t actions_per_edge, float* mask_out) {
    int apq = 4 * actions_per_edge;
    for (int q = 0; q < Q; ++q)
        for (int i = 0; i < apq; ++i) mask_out[q * apq + i] = active_quad[q] ? 0.0f : -INFINITY;
}

/* ===================================================================== policy MLP */

int64_t orc_mlp_num_params(int32_t F, int32_t HID, int32_t n_hidden) {
    int64_t n = (int64_t)HID * F + HID;
    for (int l = 1; l < n_hidden; ++l) n += (int64_t)HID * HID + HID;
    n += (int64_t)ORC_OUT * HID + ORC_OUT;
    return n;
}

static inline float lrelu_f(float x) { return x > 0.0f ? x : 0.01f * x; }
static inline double lrelu_d(double x) { return x > 0.0 ? x : 0.01 * x; }

/* Dense chain in fp32, natural order (parity unpinned vs Flux/Julia generic matmul order). */
void orc_mlp_logits_ref(const float* params, int32_t F, int32_t HID, int32_t n_hidden,
                        const int8_t* x, int32_t H, float* logits) {
    float* a = (float*)malloc(sizeof(float) * (size_t)HID);
    float* b = (float*)malloc(sizeof(float) * (size_t)HID);
    for (int h = 0; h < H; ++h) {
        const float* p = params;
        for (int i = 0; i < HID; ++i) {
            float acc = 0.0f;
            for (int k = 0; k < F; ++k) acc += p[i + (size_t)HID * k] * (float)x[h * F + k];
            a[i] = lrelu_f(acc + p[(size_t)HID * F + i]);
        }
        p += (size_t)HID * F + HID;
        for (int l = 1; l < n_hidden; ++l) {
            for (int i = 0; i < HID; ++i) {
                float acc = 0.0f;
                for (int k = 0; k < HID; ++k) acc += p[i + (size_t)HID * k] * a[k];
                b[i] = lrelu_f(acc + p[(size_t)HID * HID + i]);
            }
            float* t = a; a = b; b = t;
            p += (size_t)HID * HID + HID;
        }
        for (int o = 0; o < ORC_OUT; ++o) {
            float acc = 0.0f;
            for (int k = 0; k < HID; ++k) acc += p[o + (size_t)ORC_OUT * k] * a[k];
            logits[h * ORC_OUT + o] = acc + p[(size_t)ORC_OUT * HID + o];
        }
    }
    free(a); free(b);
}

void orc_mlp_logits_f64(const float* params, int32_t F, int32_t HID, int32_t n_hidden,
                        const int8_t* x, int32_t H, double* logits) {
    double* a = (double*)malloc(sizeof(double) * (size_t)HID);
    double* b = (double*)malloc(sizeof(double) * (size_t)HID);
    for (int h = 0; h < H; ++h) {
        const float* p = params;
        for (int i = 0; i < HID; ++i) {
            double acc = 0.0;
            for (int k = 0; k < F; ++k) acc += (double)p[i + (size_t)HID * k] * (double)x[h * F + k];
            a[i] = lrelu_d(acc + (double)p[(size_t)HID * F + i]);
        }
        p += (size_t)HID * F + HID;
        for (int l = 1; l < n_hidden; ++l) {
            for (int i = 0; i < HID; ++i) {
                double acc = 0.0;
                for (int k = 0; k < HID; ++k) acc += (double)p[i + (size_t)HID * k] * a[k];
                b[i] = lrelu_d(acc + (double)p[(size_t)HID * HID + i]);
            }
            double* t = a; a = b; b = t;
            p += (size_t)HID * HID + HID;
        }
        for (int o = 0; o < ORC_OUT; ++o) {
            double acc = 0.0;
            for (int k = 0; k < HID; ++k) acc += (double)p[o + (size_t)ORC_OUT * k] * a[k];
            logits[h * ORC_OUT + o] = acc + (double)p[(size_t)ORC_OUT * HID + o];
        }
    }
    free(a); free(b);
}

/* feature index held by accumulator register r of a 32x32 MFMA tile in lane-half hh
 * (gfx950 C/D map: row = (r&3) + 8*(r>>2) + 4*hh). */
static inline int dfeat(int tile, int r, int hh) { return 32 * tile + (r & 3) + 8 * (r >> 2) + 4 * hh; }

/* Device-order forward: reproduces, as a CPU fmaf chain, the exact accumulation order of the
 * gfx950 kernels (v_mfma_f32_32x32x2_f32: D = fma(a_k1,b_k1, fma(a_k0,b_k0, C)), accumulators
 * initialised with the bias; layer-1 k-steps pair feature s (lane-half 0) with F/2+s
 * (lane-half 1); layer-2 k-steps walk the accumulator registers of layer 1; layer 3 is a VALU
 * fmaf chain per lane-half, halves added, then bias). */
void orc_mlp_logits_dev_n(const float* params, int32_t F, int32_t HID, int32_t n_hidden,
                          const int8_t* x, int32_t H, float* logits) {
    /* any num_hidden_layers (test/policy.jl:9-19): every hidden->hidden layer walks its contraction in the
     * accumulator-register order of the previous layer's tiles, exactly like layer 2 (k_policy_fwd, DEEP form) */
    const float* W1 = params;
    const float* b1 = W1 + (size_t)HID * F;
    const float* Wh = b1 + HID;                                  /* n_hidden - 1 blocks of (W [HID,HID], b [HID]) */
    const float* W3 = Wh + (size_t)(n_hidden - 1) * ((size_t)HID * HID + HID);
    const float* b3 = W3 + (size_t)ORC_OUT * HID;
    float* h1 = (float*)malloc(sizeof(float) * (size_t)HID);
    float* h2 = (float*)malloc(sizeof(float) * (size_t)HID);
    const int F2 = F / 2, T = HID / 32;
    for (int h = 0; h < H; ++h) {
        const int8_t* xr = x + (size_t)h * F;
        for (int i = 0; i < HID; ++i) {
            float acc = b1[i];
            for (int s = 0; s < F2; ++s) {
                acc = fmaf(W1[i + (size_t)HID * s], (float)xr[s], acc);
                acc = fmaf(W1[i + (size_t)HID * (F2 + s)], (float)xr[F2 + s], acc);
            }
            h1[i] = lrelu_f(acc);
        }
        for (int l = 1; l < n_hidden; ++l) {
            const float* W2 = Wh + (size_t)(l - 1) * ((size_t)HID * HID + HID);
            const float* b2 = W2 + (size_t)HID * HID;
            for (int i = 0; i < HID; ++i) {
                float acc = b2[i];
                for (int t = 0; t < T; ++t)
                    for (int r = 0; r < 16; ++r) {
                        int k0 = dfeat(t, r, 0), k1 = dfeat(t, r, 1);
                        acc = fmaf(W2[i + (size_t)HID * k0], h1[k0], acc);
                        acc = fmaf(W2[i + (size_t)HID * k1], h1[k1], acc);
                    }
                h2[i] = lrelu_f(acc);
            }
            float* t = h1; h1 = h2; h2 = t;
        }
        for (int o = 0; o < ORC_OUT; ++o) {
            float part[2];
            for (int hh = 0; hh < 2; ++hh) {
                float acc = 0.0f;
                for (int t = 0; t < T; ++t)
                    for (int r = 0; r < 16; ++r) {
                        int k = dfeat(t, r, hh);
                        acc = fmaf(W3[o + (size_t)ORC_OUT * k], h1[k], acc);
                    }
                part[hh] = acc;
            }
            logits[h * ORC_OUT + o] = (part[0] + part[1]) + b3[o];
        }
    }
    free(h1); free(h2);
}

void orc_mlp_logits_dev(const float* params, int32_t F, int32_t HID,
                        const int8_t* x, int32_t H, float* logits) {
    orc_mlp_logits_dev_n(params, F, HID, 2, x, H, logits);
}

/* softmax(logits + mask): subtract max, exp, divide by sum (NNlib semantics);
 * masked entries are exactly 0. */
void orc_masked_softmax_ref(const float* logits, uint32_t active, int32_t A, float* probs) {
    float m = -INFINITY;
    for (int a = 0; a < A; ++a) if ((active >> (a / 16)) & 1u) m = fmaxf(m, logits[a]);
    float s = 0.0f;
    for (int a = 0; a < A; ++a) {
        float e = ((active >> (a / 16)) & 1u) ? expf(logits[a] - m) : 0.0f;
        probs[a] = e; s += e;
    }
    for (int a = 0; a < A; ++a) probs[a] = probs[a] / s;
}

/* exp for x<=0 as an explicit fmaf sequence shared bit-for-bit with the HIP kernels
 * (Cephes expf polynomial, round-to-nearest-even range reduction). */
float orc_exp_dev(float x) {
    if (!(x >= -87.0f)) return 0.0f;
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500E-4f;
    p = fmaf(p, r, 1.3981999507E-3f);
    p = fmaf(p, r, 8.3334519073E-3f);
    p = fmaf(p, r, 4.1665795894E-2f);
    p = fmaf(p, r, 1.6666665459E-1f);
    p = fmaf(p, r, 5.0000001201E-1f);
    float z = r * r;
    float y = fmaf(p, z, r);
    y = y + 1.0f;
    int32_t e = (int32_t)n + 127;               /* n in [-126,0] */
    union { uint32_t u; float f; } sc; sc.u = (uint32_t)e << 23;
    return y * sc.f;
}

/* Device-order masked softmax: lane j owns logits 4j..4j+3; max is exact in any order; the sum
 * is ((e0+e1)+e2)+e3 per lane, then an xor-butterfly over 32 lanes (offsets 16,8,4,2,1); for
 * A > 128 (several 32-row tiles per state) tile partials are added in tile order first. */
void orc_masked_softmax_dev(const float* logits, uint32_t active, int32_t A, float* probs) {
    int H = A / 4, NT = (H + 31) / 32;
    float m = -INFINITY;
    for (int a = 0; a < A; ++a) if ((active >> (a / 16)) & 1u) m = fmaxf(m, logits[a]);
    float lane[32];
    for (int j = 0; j < 32; ++j) lane[j] = 0.0f;
    for (int t = 0; t < NT; ++t)
        for (int j = 0; j < 32; ++j) {
            int h = 32 * t + j;
            float s = 0.0f;
            if (h < H) {
                float e[4];
                for (int o = 0; o < 4; ++o) {
                    int a = 4 * h + o;
                    e[o] = ((active >> (a / 16)) & 1u) ? orc_exp_dev(logits[a] - m) : 0.0f;
                    probs[a] = e[o];
                }
                s = ((e[0] + e[1]) + e[2]) + e[3];
            }
            lane[j] = (t == 0) ? s : lane[j] + s;
        }
    for (int off = 16; off >= 1; off >>= 1) {
        float nl[32];
        for (int j = 0; j < 32; ++j) nl[j] = lane[j] + lane[j ^ off];
        memcpy(lane, nl, sizeof(nl));
    }
    float S = lane[0];
    for (int a = 0; a < A; ++a) probs[a] = probs[a] / S;
}

/* rand(Categorical(p)) (Distributions.jl, parity unpinned): u in [0,1) in eltype(p);
 * cp = p[1]; i = 1; while cp <= u && i < n: i += 1; cp += p[i].  0-based result. */
int32_t orc_categorical_sample(const float* p, int32_t A, float u, int32_t* err) {
    float cp = p[0];
    int32_t i = 0;
    while (cp <= u && i < A - 1) { i += 1; cp = cp + p[i]; }
    if (err) *err = !(p[i] > 0.0f);   /* @assert ap[a] > 0.0  src/collect_rollouts.jl:7 */
    return i;
}

/* ===================================================================== rollout */

/* src/collect_rollouts.jl:1-15 per env and step, vectorised over N envs with auto-reset after a
 * terminal transition (src/rollout_buffer.jl:74-76 resets before each episode). */
void orc_collect_rollouts_tn(orc_env* e, const float* params, int32_t HID, int32_t n_hidden,
                             int64_t T, int32_t mode_dev,
                             int8_t* states, uint32_t* active,
                             float* p_sel, int32_t* actions, float* rewards, uint8_t* done) {
    const int H = e->H, F = e->F, A = e->A;
    const int64_t N = e->N;
    float* logits = (float*)malloc(sizeof(float) * (size_t)A);
    float* probs = (float*)malloc(sizeof(float) * (size_t)A);
    uint32_t key[2] = {(uint32_t)e->seed, (uint32_t)(e->seed >> 32)};
    for (int64_t t = 0; t < T; ++t)
        for (int64_t n = 0; n < N; ++n) {
            int64_t idx = t * N + n;
            int8_t* obs = states + (size_t)idx * H * F;
            orc_env_observe_one(e, n, obs);                                   /* :2  state(env) */
            active[idx] = e->active[n];
            if (mode_dev) {
                orc_mlp_logits_dev_n(params, F, HID, n_hidden, obs, H, logits);
                orc_masked_softmax_dev(logits, e->active[n], A, probs);
            } else {
                orc_mlp_logits_ref(params, F, HID, n_hidden, obs, H, logits);  /* :5 */
                orc_masked_softmax_ref(logits, e->active[n], A, probs);
            }
            uint32_t ctr[4] = {(uint32_t)(e->global_offset + n), e->tick[n], 0u, 0u}, w[4];
            orc_philox4x32_10(ctr, key, w);
            int32_t err = 0;
            int32_t a = orc_categorical_sample(probs, A, orc_u01(w[0]), &err);  /* :6-7 */
            if (err) {
                /* the walk ran off the end of a masked distribution (fp32 sum of the probabilities < u): the engine
                 * gives the rounding residue to the last action with p > 0 instead of throwing like the reference's
                 * @assert would (flag 32, informational); no positive entry at all is still an error (flag 8) */
                int32_t best = -1;
                for (int32_t q = 0; q < A; ++q) if (probs[q] > 0.0f) best = q;
                if (best >= 0) { a = best; e->err[n] |= 32; } else e->err[n] |= 8;
            }
            orc_env_step_one(e, n, a);                                        /* :9 */
            p_sel[idx] = probs[a]; actions[idx] = a;                          /* :14 update! */
            rewards[idx] = e->reward[n]; done[idx] = e->done[n];              /* :11-12 */
            if (e->done[n]) orc_env_reset_one(e, n);
        }
    free(logits); free(probs);
}

/* ===================================================================== loss / grad */

void orc_linear_action_index(const int64_t* a1, int64_t B, int64_t A, int64_t* out) {
    for (int64_t b = 0; b < B; ++b) out[b] = a1[b] + b * A;     /* src/train.jl:48-52 */
}

double orc_simplified_ppo_clip(double adv, double eps) {        /* src/train.jl:1-7 */
    return adv >= 0 ? (1.0 + eps) * adv : (1.0 - eps) * adv;
}

/* src/train.jl:21-26,35-46: Float32 probs; gain in Float32; clip and min/mean promote to
 * Float64 because epsilon is a Float64 at every reference call site. */
void orc_ppo_loss_with_entropy(const float* probs, const int64_t* lin_idx1, const float* p_old,
                               const float* adv, int64_t B, int64_t A, double eps,
                               double* ppoloss, double* entropyloss) {
    double acc = 0.0;
    for (int64_t b = 0; b < B; ++b) {
        float ps = probs[lin_idx1[b] - 1];
        float gain = ps / p_old[b] * adv[b];
        double clip = orc_simplified_ppo_clip((double)adv[b], eps);
        double m = (double)gain < clip ? (double)gain : clip;
        acc += m;
    }
    *ppoloss = -(acc / (double)B);
    const float smooth = 1e-8f;
    const float one_minus = 1.0f - smooth;             /* == 1.0f in Float32 */
    const float add = smooth / (float)A;
    float hsum = 0.0f;
    for (int64_t b = 0; b < B; ++b) {
        float h = 0.0f;
        for (int64_t a = 0; a < A; ++a) {
            float sp = one_minus * probs[b * A + a] + add;
            h += sp * logf(sp);
        }
        hsum += -h;
    }
    *entropyloss = -(double)(hsum / (float)B);
}

/* Whole step_batch! forward + analytic backward in Float64. */
void orc_step_batch_grad_f64(const float* params, int32_t F, int32_t HID, int32_t n_hidden,
                             const int8_t* states, const uint32_t* active,
                             const int32_t* actions0, const float* p_old,
                             const float* adv, int64_t B, int32_t H, double eps,
                             double entropy_weight, double* g,
                             double* ppoloss, double* entropyloss) {
    const int A = 4 * H, L = n_hidden;
    const int64_t np = orc_mlp_num_params(F, HID, L);
    memset(g, 0, sizeof(double) * (size_t)np);
    /* parameter offsets */
    int64_t* offW = (int64_t*)malloc(sizeof(int64_t) * (size_t)(L + 1));
    int64_t* offb = (int64_t*)malloc(sizeof(int64_t) * (size_t)(L + 1));
    int64_t o = 0;
    for (int l = 0; l <= L; ++l) {
        int in = (l == 0) ? F : HID, out = (l == L) ? ORC_OUT : HID;
        offW[l] = o; o += (int64_t)out * in; offb[l] = o; o += out;
    }
    double* act = (double*)malloc(sizeof(double) * (size_t)L * H * HID);   /* post-activation */
    double* logit = (double*)malloc(sizeof(double) * (size_t)A);
    double* p = (double*)malloc(sizeof(double) * (size_t)A);
    double* dlogit = (double*)malloc(sizeof(double) * (size_t)A);
    double* dcur = (double*)malloc(sizeof(double) * (size_t)HID);
    double* dprev = (double*)malloc(sizeof(double) * (size_t)HID);
    double lp = 0.0, le = 0.0;
    const double s = (double)1e-8f, sA = s / (double)A;
    for (int64_t b = 0; b < B; ++b) {
        const int8_t* xs = states + (size_t)b * H * F;
        /* forward */
        for (int h = 0; h < H; ++h) {
            for (int l = 0; l < L; ++l) {
                int in = (l == 0) ? F : HID;
                const float* W = params + offW[l];
                const float* bb = params + offb[l];
                double* out = act + ((size_t)l * H + h) * HID;
                const double* inp = (l == 0) ? NULL : act + ((size_t)(l - 1) * H + h) * HID;
                for (int i = 0; i < HID; ++i) {
                    double acc = (double)bb[i];
                    for (int k = 0; k < in; ++k)
                        acc += (double)W[i + (size_t)HID * k] * (l == 0 ? (double)xs[h * F + k] : inp[k]);
                    out[i] = lrelu_d(acc);
                }
            }
            const float* W = params + offW[L];
            const float* bb = params + offb[L];
            const double* inp = act + ((size_t)(L - 1) * H + h) * HID;
            for (int oo = 0; oo < ORC_OUT; ++oo) {
                double acc = (double)bb[oo];
                for (int k = 0; k < HID; ++k) acc += (double)W[oo + (size_t)ORC_OUT * k] * inp[k];
                logit[h * ORC_OUT + oo] = acc;
            }
        }
        uint32_t am = active[b];
        double m = -INFINITY;
        for (int a = 0; a < A; ++a) if ((am >> (a / 16)) & 1u) m = fmax(m, logit[a]);
        double Z = 0.0;
        for (int a = 0; a < A; ++a) { p[a] = ((am >> (a / 16)) & 1u) ? exp(logit[a] - m) : 0.0; Z += p[a]; }
        for (int a = 0; a < A; ++a) p[a] /= Z;
        /* loss (src/train.jl:35-46) */
        double advb = (double)adv[b], pob = (double)p_old[b];
        int ab = actions0[b];
        double gain = p[ab] / pob * advb;
        double clip = orc_simplified_ppo_clip(advb, eps);
        lp += gain < clip ? gain : clip;
        double Hb = 0.0;
        for (int a = 0; a < A; ++a) { double sp = (1.0 - s) * p[a] + sA; Hb -= sp * log(sp); }
        le += Hb;
        /* dL/dp, then softmax backward */
        double dot = 0.0;
        for (int a = 0; a < A; ++a) {
            double sp = (1.0 - s) * p[a] + sA;
            double d = (entropy_weight / (double)B) * (1.0 - s) * (log(sp) + 1.0);
            if (a == ab && gain < clip) d += -(1.0 / (double)B) * advb / pob;
            dlogit[a] = d; dot += p[a] * d;
        }
        for (int a = 0; a < A; ++a) dlogit[a] = p[a] * (dlogit[a] - dot);
        /* MLP backward per half-edge */
        for (int h = 0; h < H; ++h) {
            const float* W = params + offW[L];
            const double* inp = act + ((size_t)(L - 1) * H + h) * HID;
            for (int k = 0; k < HID; ++k) dcur[k] = 0.0;
            for (int oo = 0; oo < ORC_OUT; ++oo) {
                double dy = dlogit[h * ORC_OUT + oo];
                g[offb[L] + oo] += dy;
                for (int k = 0; k < HID; ++k) {
                    g[offW[L] + oo + (int64_t)ORC_OUT * k] += dy * inp[k];
                    dcur[k] += (double)W[oo + (size_t)ORC_OUT * k] * dy;
                }
            }
            for (int l = L - 1; l >= 0; --l) {
                int in = (l == 0) ? F : HID;
                const float* Wl = params + offW[l];
                const double* outv = act + ((size_t)l * H + h) * HID;
                const double* inv = (l == 0) ? NULL : act + ((size_t)(l - 1) * H + h) * HID;
                for (int i = 0; i < HID; ++i) dcur[i] *= (outv[i] > 0.0 ? 1.0 : 0.01);
                if (l > 0) for (int k = 0; k < HID; ++k) dprev[k] = 0.0;
                for (int i = 0; i < HID; ++i) {
                    double dz = dcur[i];
                    g[offb[l] + i] += dz;
                    for (int k = 0; k < in; ++k) {
                        double xin = (l == 0) ? (double)xs[h * F + k] : inv[k];
                        g[offW[l] + i + (int64_t)HID * k] += dz * xin;
                        if (l > 0) dprev[k] += (double)Wl[i + (size_t)HID * k] * dz;
                    }
                }
                if (l > 0) { double* t = dcur; dcur = dprev; dprev = t; }
            }
        }
    }
    *ppoloss = -(lp / (double)B);
    *entropyloss = entropy_weight * (-(le / (double)B));
    free(offW); free(offb); free(act); free(logit); free(p); free(dlogit); free(dcur); free(dprev);
}

/* Flux legacy Adam apply! + update! (parity unpinned).  Element arithmetic in
 * Float64 (Float64 hyper-parameters broadcast against Float32 arrays), stores rounded to Float32. */
void orc_adam_step(float* params, const float* grad, float* m, float* v, double* beta_pow,
                   int64_t n, double eta, double beta1, double beta2, double eps) {
    for (int64_t i = 0; i < n; ++i) {
        double gd = (double)grad[i];
        m[i] = (float)(beta1 * (double)m[i] + (1.0 - beta1) * gd);
        v[i] = (float)(beta2 * (double)v[i] + ((1.0 - beta2) * gd) * gd);
        double delta = (double)m[i] / (1.0 - beta_pow[0]) / (sqrt((double)v[i] / (1.0 - beta_pow[1])) + eps) * eta;
        float df = (float)delta;
        params[i] = params[i] - df;
    }
    beta_pow[0] *= beta1;
    beta_pow[1] *= beta2;
}
