"""ctypes binding of the CPU ORACLE (oracle/ppo_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (proximalpolicyoptimization.jl_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "libppo_oracle.so")


def build(force=False):
    src = os.path.join(_DIR, "ppo_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _DIR, "-s"])
    return _SO


_lib = None

c_f32p = C.POINTER(C.c_float)
c_f64p = C.POINTER(C.c_double)
c_u8p = C.POINTER(C.c_uint8)
c_i8p = C.POINTER(C.c_int8)
c_i32p = C.POINTER(C.c_int32)
c_u32p = C.POINTER(C.c_uint32)
c_i64p = C.POINTER(C.c_int64)


def _p(a, t):
    return a.ctypes.data_as(t)


class _Env(C.Structure):
    _fields_ = [("Q", C.c_int32), ("H", C.c_int32), ("A", C.c_int32), ("V", C.c_int32), ("F", C.c_int32),
                ("max_actions", C.c_int32), ("no_action_reward", C.c_float), ("N", C.c_int64),
                ("global_offset", C.c_int64), ("seed", C.c_uint64),
                ("score", c_i8p), ("degree", c_i8p), ("active", c_u32p), ("steps", c_i32p),
                ("reward", c_f32p), ("done", c_u8p), ("episode", c_u32p), ("tick", c_u32p), ("err", c_i32p)]


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_u01.restype = C.c_float
        L.orc_u01.argtypes = [C.c_uint32]
        L.orc_feistel_perm.restype = C.c_int64
        L.orc_feistel_perm.argtypes = [C.c_int64, C.c_int64, C.c_uint64, C.c_uint32]
        L.orc_env_create.restype = C.POINTER(_Env)
        L.orc_env_create.argtypes = [C.c_int32, C.c_int32, C.c_float, C.c_int64, C.c_int64, C.c_uint64]
        L.orc_env_destroy.argtypes = [C.POINTER(_Env)]
        L.orc_env_reset.argtypes = [C.POINTER(_Env)]
        L.orc_env_reset_one.argtypes = [C.POINTER(_Env), C.c_int64]
        L.orc_env_step_one.argtypes = [C.POINTER(_Env), C.c_int64, C.c_int32]
        L.orc_env_observe_one.argtypes = [C.POINTER(_Env), C.c_int64, c_i8p]
        L.orc_env_template.restype = C.c_int32
        L.orc_env_template.argtypes = [C.c_int32, C.c_int32, C.c_int32]
        L.orc_mlp_num_params.restype = C.c_int64
        L.orc_mlp_num_params.argtypes = [C.c_int32, C.c_int32, C.c_int32]
        L.orc_exp_dev.restype = C.c_float
        L.orc_exp_dev.argtypes = [C.c_float]
        L.orc_categorical_sample.restype = C.c_int32
        L.orc_categorical_sample.argtypes = [c_f32p, C.c_int32, C.c_float, c_i32p]
        L.orc_simplified_ppo_clip.restype = C.c_double
        L.orc_simplified_ppo_clip.argtypes = [C.c_double, C.c_double]
        L.orc_compute_returns.argtypes = [c_f32p, c_u8p, C.c_int64, C.c_double, C.c_int32, c_f32p]
        L.orc_compute_returns_tn.argtypes = [c_f32p, c_u8p, C.c_int64, C.c_int64, C.c_double, C.c_int32, c_f32p]
        L.orc_gae_tn.argtypes = [c_f32p, c_u8p, c_f32p, C.c_int64, C.c_int64, C.c_double, C.c_double, c_f32p, c_f32p]
        L.orc_philox4x32_10.argtypes = [c_u32p, c_u32p, c_u32p]
        L.orc_index_to_action.argtypes = [C.c_int32, C.c_int32, c_i32p, c_i32p, c_i32p]
        L.orc_index_to_action_edges.argtypes = [C.c_int32, C.c_int32, C.c_int32, c_i32p, c_i32p, c_i32p]
        L.orc_action_mask.argtypes = [c_u8p, C.c_int32, C.c_int32, c_f32p]
        L.orc_mlp_logits_ref.argtypes = [c_f32p, C.c_int32, C.c_int32, C.c_int32, c_i8p, C.c_int32, c_f32p]
        L.orc_mlp_logits_f64.argtypes = [c_f32p, C.c_int32, C.c_int32, C.c_int32, c_i8p, C.c_int32, c_f64p]
        L.orc_mlp_logits_dev.argtypes = [c_f32p, C.c_int32, C.c_int32, c_i8p, C.c_int32, c_f32p]
        L.orc_mlp_logits_dev_n.argtypes = [c_f32p, C.c_int32, C.c_int32, C.c_int32, c_i8p, C.c_int32, c_f32p]
        L.orc_masked_softmax_ref.argtypes = [c_f32p, C.c_uint32, C.c_int32, c_f32p]
        L.orc_masked_softmax_dev.argtypes = [c_f32p, C.c_uint32, C.c_int32, c_f32p]
        L.orc_collect_rollouts_tn.argtypes = [C.POINTER(_Env), c_f32p, C.c_int32, C.c_int32, C.c_int64, C.c_int32,
                                              c_i8p, c_u32p, c_f32p, c_i32p, c_f32p, c_u8p]
        L.orc_linear_action_index.argtypes = [c_i64p, C.c_int64, C.c_int64, c_i64p]
        L.orc_ppo_loss_with_entropy.argtypes = [c_f32p, c_i64p, c_f32p, c_f32p, C.c_int64, C.c_int64, C.c_double,
                                                c_f64p, c_f64p]
        L.orc_step_batch_grad_f64.argtypes = [c_f32p, C.c_int32, C.c_int32, C.c_int32, c_i8p, c_u32p, c_i32p, c_f32p,
                                              c_f32p, C.c_int64, C.c_int32, C.c_double, C.c_double, c_f64p, c_f64p,
                                              c_f64p]
        L.orc_adam_step.argtypes = [c_f32p, c_f32p, c_f32p, c_f32p, c_f64p, C.c_int64, C.c_double, C.c_double,
                                    C.c_double, C.c_double]
        _lib = L
    return _lib


# ------------------------------------------------------------------ thin numpy wrappers
def compute_returns(rewards, terminal, discount, discount_is_f32=False):
    r = np.ascontiguousarray(rewards, np.float32)
    t = np.ascontiguousarray(terminal, np.uint8)
    out = np.empty_like(r)
    lib().orc_compute_returns(_p(r, c_f32p), _p(t, c_u8p), r.size, float(discount), int(discount_is_f32), _p(out, c_f32p))
    return out


def compute_returns_tn(rewards, done, discount, discount_is_f32=False):
    r = np.ascontiguousarray(rewards, np.float32)
    d = np.ascontiguousarray(done, np.uint8)
    T, N = r.shape
    out = np.empty_like(r)
    lib().orc_compute_returns_tn(_p(r, c_f32p), _p(d, c_u8p), T, N, float(discount), int(discount_is_f32), _p(out, c_f32p))
    return out


def gae_tn(rewards, done, values, gamma, lam):
    r = np.ascontiguousarray(rewards, np.float32)
    d = np.ascontiguousarray(done, np.uint8)
    v = np.ascontiguousarray(values, np.float32)
    T, N = r.shape
    assert v.shape == (T + 1, N)
    adv = np.empty_like(r)
    ret = np.empty_like(r)
    lib().orc_gae_tn(_p(r, c_f32p), _p(d, c_u8p), _p(v, c_f32p), T, N, float(gamma), float(lam), _p(adv, c_f32p), _p(ret, c_f32p))
    return adv, ret


def philox(ctr, key):
    c = np.asarray(ctr, np.uint32)
    k = np.asarray(key, np.uint32)
    o = np.zeros(4, np.uint32)
    lib().orc_philox4x32_10(_p(c, c_u32p), _p(k, c_u32p), _p(o, c_u32p))
    return o


def u01(w):
    return float(lib().orc_u01(int(w)))


def feistel_perm(n, seed, epoch):
    L = lib()
    return np.array([L.orc_feistel_perm(i, n, seed, epoch) for i in range(n)], np.int64)


def index_to_action(index1, actions_per_edge=4, edges=4):
    """edges=4: the quad game (test/quad_game_utilities.jl:95-105); edges=3: the notebook's triangle variant."""
    q, e, t = C.c_int32(), C.c_int32(), C.c_int32()
    lib().orc_index_to_action_edges(index1, edges, actions_per_edge, C.byref(q), C.byref(e), C.byref(t))
    return q.value, e.value, t.value


def action_mask(active_quad, actions_per_edge=4):
    aq = np.ascontiguousarray(active_quad, np.uint8)
    out = np.empty(aq.size * 4 * actions_per_edge, np.float32)
    lib().orc_action_mask(_p(aq, c_u8p), aq.size, actions_per_edge, _p(out, c_f32p))
    return out


def mlp_num_params(F, HID, n_hidden=2):
    return int(lib().orc_mlp_num_params(F, HID, n_hidden))


def mlp_logits(params, F, HID, x, mode="ref", n_hidden=2):
    p = np.ascontiguousarray(params, np.float32)
    x = np.ascontiguousarray(x, np.int8)
    H = x.shape[0]
    assert x.shape == (H, F)
    if mode == "f64":
        out = np.empty(H * 4, np.float64)
        lib().orc_mlp_logits_f64(_p(p, c_f32p), F, HID, n_hidden, _p(x, c_i8p), H, _p(out, c_f64p))
        return out
    out = np.empty(H * 4, np.float32)
    if mode == "dev":
        lib().orc_mlp_logits_dev_n(_p(p, c_f32p), F, HID, n_hidden, _p(x, c_i8p), H, _p(out, c_f32p))
    else:
        lib().orc_mlp_logits_ref(_p(p, c_f32p), F, HID, n_hidden, _p(x, c_i8p), H, _p(out, c_f32p))
    return out


def masked_softmax(logits, active, mode="ref"):
    l = np.ascontiguousarray(logits, np.float32)
    out = np.empty_like(l)
    fn = lib().orc_masked_softmax_dev if mode == "dev" else lib().orc_masked_softmax_ref
    fn(_p(l, c_f32p), int(active), l.size, _p(out, c_f32p))
    return out


def exp_dev(x):
    return float(lib().orc_exp_dev(float(x)))


def categorical_sample(p, u):
    p = np.ascontiguousarray(p, np.float32)
    err = C.c_int32(0)
    a = lib().orc_categorical_sample(_p(p, c_f32p), p.size, float(u), C.byref(err))
    return int(a), int(err.value)


def action_probabilities(params, F, HID, x, active, mode="ref", n_hidden=2):
    """PPO.action_probabilities (test/quad_game_utilities.jl:65-71) for one state."""
    return masked_softmax(mlp_logits(params, F, HID, x, mode, n_hidden), active, "dev" if mode == "dev" else "ref")


class Env:
    """Synthetic rand-poly-shaped env (oracle side)."""

    def __init__(self, Q=8, max_actions=128, no_action_reward=-4.0, N=1, global_offset=0, seed=1234):
        self.L = lib()
        self.e = self.L.orc_env_create(Q, max_actions, no_action_reward, N, global_offset, seed)
        c = self.e.contents
        self.Q, self.H, self.A, self.V, self.F, self.N = c.Q, c.H, c.A, c.V, c.F, c.N

    def __del__(self):
        try:
            self.L.orc_env_destroy(self.e)
        except Exception:
            pass

    def reset(self):
        self.L.orc_env_reset(self.e)

    def reset_one(self, n):
        self.L.orc_env_reset_one(self.e, n)

    def step_one(self, n, a):
        self.L.orc_env_step_one(self.e, n, int(a))

    def step(self, actions):
        for n, a in enumerate(actions):
            self.step_one(n, a)

    def observe_one(self, n):
        obs = np.empty((self.H, self.F), np.int8)
        self.L.orc_env_observe_one(self.e, n, _p(obs, c_i8p))
        return obs

    def observe(self):
        return np.stack([self.observe_one(n) for n in range(self.N)])

    def _arr(self, name, dtype, per=1):
        c = self.e.contents
        ptr = getattr(c, name)
        return np.ctypeslib.as_array(ptr, shape=(self.N * per,)).view(dtype).reshape(self.N, per) if per > 1 else \
            np.ctypeslib.as_array(ptr, shape=(self.N,))

    @property
    def score(self):
        return self._arr("score", np.int8, self.V)

    @property
    def degree(self):
        return self._arr("degree", np.int8, self.V)

    @property
    def active(self):
        return self._arr("active", np.uint32)

    @property
    def reward(self):
        return self._arr("reward", np.float32)

    @property
    def done(self):
        return self._arr("done", np.uint8)

    @property
    def steps(self):
        return self._arr("steps", np.int32)

    @property
    def err(self):
        return self._arr("err", np.int32)

    @property
    def tick(self):
        return self._arr("tick", np.uint32)

    @property
    def episode(self):
        return self._arr("episode", np.uint32)


def collect_rollouts_tn(env, params, HID, T, mode_dev=True, n_hidden=2):
    p = np.ascontiguousarray(params, np.float32)
    N, H, F = env.N, env.H, env.F
    states = np.empty((T, N, H, F), np.int8)
    active = np.empty((T, N), np.uint32)
    p_sel = np.empty((T, N), np.float32)
    actions = np.empty((T, N), np.int32)
    rewards = np.empty((T, N), np.float32)
    done = np.empty((T, N), np.uint8)
    lib().orc_collect_rollouts_tn(env.e, _p(p, c_f32p), HID, n_hidden, T, int(mode_dev), _p(states, c_i8p),
                                  _p(active, c_u32p), _p(p_sel, c_f32p), _p(actions, c_i32p), _p(rewards, c_f32p),
                                  _p(done, c_u8p))
    return dict(states=states, active=active, p_sel=p_sel, actions=actions, rewards=rewards, done=done)


def linear_action_index(a1, A):
    a = np.ascontiguousarray(a1, np.int64)
    out = np.empty_like(a)
    lib().orc_linear_action_index(_p(a, c_i64p), a.size, A, _p(out, c_i64p))
    return out


def simplified_ppo_clip(adv, eps):
    return float(lib().orc_simplified_ppo_clip(adv, eps))


def ppo_loss_with_entropy(probs_AB, lin_idx1, p_old, adv, eps):
    """probs_AB: [A,B] column-major semantics -> pass as numpy [B,A] C-order (same memory)."""
    pr = np.ascontiguousarray(probs_AB, np.float32)
    B, A = pr.shape
    li = np.ascontiguousarray(lin_idx1, np.int64)
    po = np.ascontiguousarray(p_old, np.float32)
    ad = np.ascontiguousarray(adv, np.float32)
    a, b = C.c_double(), C.c_double()
    lib().orc_ppo_loss_with_entropy(_p(pr, c_f32p), _p(li, c_i64p), _p(po, c_f32p), _p(ad, c_f32p), B, A, float(eps),
                                    C.byref(a), C.byref(b))
    return a.value, b.value


def step_batch_grad_f64(params, F, HID, states, active, actions0, p_old, adv, eps, entropy_weight, n_hidden=2):
    p = np.ascontiguousarray(params, np.float32)
    s = np.ascontiguousarray(states, np.int8)
    B, H, _ = s.shape
    am = np.ascontiguousarray(active, np.uint32)
    a0 = np.ascontiguousarray(actions0, np.int32)
    po = np.ascontiguousarray(p_old, np.float32)
    ad = np.ascontiguousarray(adv, np.float32)
    g = np.empty(p.size, np.float64)
    lp, le = C.c_double(), C.c_double()
    lib().orc_step_batch_grad_f64(_p(p, c_f32p), F, HID, n_hidden, _p(s, c_i8p), _p(am, c_u32p), _p(a0, c_i32p),
                                  _p(po, c_f32p), _p(ad, c_f32p), B, H, float(eps), float(entropy_weight),
                                  _p(g, c_f64p), C.byref(lp), C.byref(le))
    return g, lp.value, le.value


def adam_step(params, grad, m, v, beta_pow, eta=1e-4, beta1=0.9, beta2=0.999, eps=1e-8):
    """In-place on float32 arrays params/m/v and float64[2] beta_pow."""
    g = np.ascontiguousarray(grad, np.float32)
    lib().orc_adam_step(_p(params, c_f32p), _p(g, c_f32p), _p(m, c_f32p), _p(v, c_f32p), _p(beta_pow, c_f64p),
                        params.size, eta, beta1, beta2, eps)


def glorot_params(F, HID, n_hidden=2, seed=0):
    """Glorot-uniform weights, zero bias, flat Flux order (W [out,in] column-major)."""
    rng = np.random.default_rng(seed)
    parts = []
    dims = [(HID, F)] + [(HID, HID)] * (n_hidden - 1) + [(4, HID)]
    for (o, i) in dims:
        lim = np.sqrt(6.0 / (o + i))
        W = rng.uniform(-lim, lim, size=(o, i)).astype(np.float32)
        parts.append(np.asfortranarray(W).ravel(order="F"))
        parts.append(np.zeros(o, np.float32))
    return np.concatenate(parts).astype(np.float32)
