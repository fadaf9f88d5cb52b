"""Host-side mirror of module ProximalPolicyOptimization (reference: src/ProximalPolicyOptimization.jl)
over the C ABI of libppo_hip.so (include/ppo_hip.h).

Julia is not available in the build image, so the host language above the C ABI is Python; the
Julia `ccall` shim a maintainer would use is julia/ProximalPolicyOptimizationHIP.jl.
Names, argument order and error behaviour follow the reference:

    Julia                                   here
    ------------------------------------    -------------------------------------------
    PPO.state / reward / is_terminal        state(env) / reward(env) / is_terminal(env)
    PPO.reset!(env) / PPO.step!(env, a)     reset_(env) / step_(env, a)          (a is 1-based)
    PPO.action_probabilities(policy, s)     action_probabilities(policy, s)
    PPO.batch_action_probabilities          batch_action_probabilities(policy, s)
    PPO.batch_state / number_of_actions_per_state / batch_advantage / save_loss   same names
    PPO.BufferRollouts()                    BufferRollouts()
    PPO.collect_rollouts!(r, env, p, n, g)  collect_rollouts_(r, env, p, n, g)
    PPO.compute_returns(r, t, g)            compute_returns(r, t, g)
    PPO.construct_dataset(r)                construct_dataset(r);  len(ds);  ds[i] / ds[[i,...]] (1-based)
    PPO.ppo_train!(...)                     ppo_train_(...)
    PPO.ppo_iterate!(...)                   ppo_iterate_(...)
    Flux.Optimiser(Adam(1e-4))              Optimiser(Adam(1e-4))

Plugin functions are generic: calling one on an object that does not overload it raises
`PPOError("Function <name> needs to be overloaded")` like src/ProximalPolicyOptimization.jl:12-14.
Indices that cross this API are 1-based like the reference; the C ABI below is 0-based.
All compute runs in the HIP library; nothing here falls back to the CPU.
"""
import ctypes as C
import functools
import os
import shutil
import sys

import numpy as np

from . import _lib
from . import checkpoint  # noqa: F401
from . import generic  # noqa: F401
from .generic import HostDataset, HostRollouts  # noqa: F401
from ._lib import PPOError, call, lib
from .disk import (DiskDataset, DiskRollouts, bson_decode_state, bson_encode_state, export_reference_layout,  # noqa: F401
                   update_, write_returns_to_disk)

__all__ = [
    "PPOError", "HipVecEnv", "HipPolicy", "Adam", "Optimiser", "StateData", "BufferRollouts", "BufferDataset",
    "state", "reward", "is_terminal", "reset_", "step_", "action_probabilities", "batch_action_probabilities",
    "batch_state", "number_of_actions_per_state", "batch_advantage", "save_loss", "compute_returns",
    "compute_returns_tn", "gae_tn", "compute_gae_", "profile_gae", "collect_rollouts_", "collect_rollouts_steps_", "construct_dataset",
    "simplified_ppo_clip", "get_linear_action_index", "ppo_loss_with_entropy", "categorical_sample", "step_batch_",
    "ppo_train_", "ppo_iterate_", "get_optimizer_learning_rate", "index_to_action", "action_mask", "DataParallel",
    "device_count", "average_returns", "average_best_returns", "average_normalized_returns",
    "evaluate_trajectories", "DiskRollouts", "DiskDataset", "update_", "write_returns_to_disk",
    "export_reference_layout", "load_disk_rollouts", "bson_encode_state", "bson_decode_state", "philox4x32_10", "profile_enable", "profile_get", "synchronize",
    "HostRollouts", "HostDataset", "update_rollouts_", "compute_state_value_", "collect_step_data_", "collect_episode_data_",
    "permute_", "shuffle_", "single_trajectory_return", "smoothed_entropy", "clamped_entropy", "ppo_loss", "step_epoch_",
    "save_policy", "load_policy", "forward_backward",
]


def _p(a, t):
    return a.ctypes.data_as(t)


def _not_implemented(name):
    raise PPOError(-1, "Function %s needs to be overloaded" % name)       # src/ProximalPolicyOptimization.jl:12-14


def _generic(name):
    @functools.singledispatch
    def f(obj, *a, **k):
        _not_implemented(name)
    f.__name__ = name
    return f


# ---- plugin generic functions (src/ProximalPolicyOptimization.jl:16-30)
state = _generic("state")
reward = _generic("reward")
is_terminal = _generic("is_terminal")
reset_ = _generic("reset!")
step_ = _generic("step!")
action_probabilities = _generic("action_probabilities")
batch_action_probabilities = _generic("batch_action_probabilities")
batch_state = _generic("batch_state")
number_of_actions_per_state = _generic("number_of_actions_per_state")
batch_advantage = _generic("batch_advantage")
save_loss = _generic("save_loss")


def device_count():
    return _lib.device_count()


def set_rollout_persistent(on=None):
    """Rollout execution: True = one launch per rollout (each wavefront walks its envs through all T steps) wherever
    covered, False = three launches per step, None = automatic (default: one launch for Q = 8 envs).  Bit-identical
    results either way."""
    call("ppo_set_rollout_persistent", -1 if on is None else int(bool(on)))


def set_rollout_compact(on=None):
    """State storage of engine-collected rollouts: True = compact env snapshots (64 B per transition for Q = 8; the train
    forward re-derives the observation rows), False = expanded observations (2304 B), None = automatic (compact above
    32 GiB of expanded states or while streaming to disk).  Bit-identical results either way."""
    call("ppo_set_rollout_compact", -1 if on is None else int(bool(on)))


def set_bwd_small_max_tiles(tiles=None):
    """Minibatches of up to `tiles` 32-row tiles use the three-product backward (no per-workgroup gradient slabs), larger
    ones the fused kernel.  None = default (384), 0 = always the fused kernel."""
    call("ppo_set_bwd_small_max_tiles", -1 if tiles is None else int(tiles))


def set_bwd_split_bf16(mode=None):
    """fp32 policies: True = the fused backward's three big products as split-fp32 ("bf16x6") products on the bf16 matrix
    pipe, False = the pure fp32-MFMA kernel, None = default (PPO_BWD_SPLIT_BF16, else on)."""
    call("ppo_set_bwd_split_bf16", -1 if mode is None else int(bool(mode)))


def set_train_tile_max_tiles(tiles=None):
    """Minibatches of up to `tiles` 32-row tiles run forward + loss + backward-data of each tile in one workgroup
    (k_policy_train_tile) followed by the split-K weight-gradient kernel.  None = default, 0 = never."""
    call("ppo_set_train_tile_max_tiles", -1 if tiles is None else int(tiles))


def set_fwd_split_max_states(states=None):
    """Minibatches of up to `states` states use the train forward that gives each state to 2 or 4 waves.  None = default
    (512), 0 = always one wave per state."""
    call("ppo_set_fwd_split_max_states", -1 if states is None else int(states))


def set_rollout_split_max_envs(envs=None):
    """One-launch rollouts of up to `envs` envs give every env to 2 or 4 waves (bit-identical results).  None = default
    (512), 0 = always one wave per env."""
    call("ppo_set_rollout_split_max_envs", -1 if envs is None else int(envs))


def synchronize():
    call("ppo_device_synchronize")


def profile_enable(on=True):
    call("ppo_profile_enable", int(bool(on)))


def profile_returns(T, N, discount=1.0, iters=20):
    """Average device time (ms) of the return scan on resident [T,N] columns (K6 roofline measurement)."""
    ms = C.c_double(0)
    call("ppo_profile_returns", int(T), int(N), float(discount), int(iters), C.byref(ms))
    return ms.value


def profile_gae(T, N, gamma=0.99, lam=0.95, iters=20):
    """Average device time (ms) of the GAE scan on resident [T,N] columns (17 B/transition roofline measurement)."""
    ms = C.c_double(0)
    call("ppo_profile_gae", int(T), int(N), float(gamma), float(lam), int(iters), C.byref(ms))
    return ms.value


def profile_get(name):
    ms, n = C.c_double(0), C.c_int64(0)
    call("ppo_profile_get", name.encode(), C.byref(ms), C.byref(n))
    return ms.value, n.value


# ------------------------------------------------------------------ state container (test/quad_game_utilities.jl:17-20)
class StateData:
    """vertex_score: int8 [H,F] (single) or [B,H,F] (batched); action_mask: active-quad bit mask(s).

    The reference stores the [F,H] Int matrix and a Float32 {0,-Inf} mask vector; here the same
    information travels as the row-per-half-edge int8 matrix and the active-quad bits the mask is
    built from (action_mask(), test/quad_game_utilities.jl:39-44)."""

    def __init__(self, vertex_score, action_mask):
        self.vertex_score = vertex_score
        self.action_mask = action_mask

    def mask_vector(self, actions_per_edge=4):
        bits = np.atleast_1d(np.asarray(self.action_mask, np.uint32))
        Hh = self.vertex_score.shape[-2]
        q = np.arange(Hh * actions_per_edge) // (4 * actions_per_edge)
        m = np.where((bits[:, None] >> q[None, :]) & 1, np.float32(0), np.float32(-np.inf))
        return m[0] if np.ndim(self.action_mask) == 0 else m


@batch_state.register(list)
def _(states):
    """PPO.batch_state (test/quad_game_utilities.jl:26-33): cat along the batch dimension."""
    if not states or not all(isinstance(x, StateData) for x in states):
        _not_implemented("batch_state")          # the reference dispatches on the element type: no method, no batch
    vs = np.stack([np.asarray(s.vertex_score, np.int8) for s in states])
    am = np.array([np.uint32(s.action_mask) for s in states], np.uint32)
    return StateData(vs, am)


@number_of_actions_per_state.register(StateData)
def _(s):
    return int(s.vertex_score.shape[-2]) * 4          # default: size(mask, 1)  


@batch_advantage.register(StateData)
def _(s, returns):
    return np.asarray(returns, np.float32)             # reference scripts use raw returns as advantage


def index_to_action(index, actions_per_edge=4):
    """test/quad_game_utilities.jl:95-105 (1-based)."""
    apq = 4 * actions_per_edge
    quad = (index - 1) // apq + 1
    qa = (index - 1) % apq
    return quad, qa // actions_per_edge + 1, qa % actions_per_edge + 1


def action_mask(active_quad, actions_per_edge=4):
    """test/quad_game_utilities.jl:39-44."""
    req = np.repeat(~np.asarray(active_quad, bool), 4 * actions_per_edge)
    return np.where(req, -np.inf, 0.0).astype(np.float32)


# ------------------------------------------------------------------ env
class HipVecEnv:
    """N synthetic rand-poly-shaped envs resident on the GPU (ppo_env_*).  Not QuadMeshGame: the
    mesh dynamics are not in the reference tree ('Synthetic env')."""

    def __init__(self, num_envs=1, Q=8, max_actions=128, no_action_reward=-4.0, seed=1234, global_offset=0,
                 strict_sampling=False):
        h = C.c_void_p()
        call("ppo_env_create", 0, int(num_envs), int(global_offset), int(Q), int(max_actions), float(no_action_reward),
             int(seed), C.byref(h))
        self._h = h
        self.N, self.Q, self.H, self.F, self.A = int(num_envs), Q, 4 * Q, 72, 16 * Q
        self.max_actions = max_actions
        if strict_sampling:      # reference behaviour: a CDF walk that ends on a masked action raises (@assert ap[a] > 0.0)
            call("ppo_env_set_strict_sampling", h, 1)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().ppo_env_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def error_flags(self):
        """OR of the device-side flags since creation: 1 action on an inactive quad, 2 index out of range, 4 step! on a
        terminated env, 8 sampled action with probability 0 (the reference's @assert), 32 informational: a CDF
        rounding residue was handed to the last unmasked action (numerics contract)."""
        f = C.c_int32(0)
        lib().ppo_env_check_errors(self._h, C.byref(f))
        return f.value

    def internal(self):
        V = 4 * self.Q
        sc = np.empty((self.N, V), np.int8)
        dg = np.empty((self.N, V), np.int8)
        st = np.empty(self.N, np.int32)
        ep = np.empty(self.N, np.uint32)
        tk = np.empty(self.N, np.uint32)
        call("ppo_env_get_internal", self._h, _p(sc, _lib.c_i8p), _p(dg, _lib.c_i8p), _p(st, _lib.c_i32p),
             _p(ep, _lib.c_u32p), _p(tk, _lib.c_u32p))
        return dict(score=sc, degree=dg, steps=st, episode=ep, tick=tk)


@state.register(HipVecEnv)
def _(env):
    obs = np.empty((env.N, env.H, env.F), np.int8)
    act = np.empty(env.N, np.uint32)
    call("ppo_env_get_state", env._h, _p(obs, _lib.c_i8p), _p(act, _lib.c_u32p))
    return StateData(obs[0], act[0]) if env.N == 1 else StateData(obs, act)


@reward.register(HipVecEnv)
def _(env):
    r = np.empty(env.N, np.float32)
    call("ppo_env_get_reward", env._h, _p(r, _lib.c_f32p))
    return float(r[0]) if env.N == 1 else r


@is_terminal.register(HipVecEnv)
def _(env):
    d = np.empty(env.N, np.uint8)
    call("ppo_env_get_terminal", env._h, _p(d, _lib.c_u8p))
    return bool(d[0]) if env.N == 1 else d.astype(bool)


@reset_.register(HipVecEnv)
def _(env):
    call("ppo_env_reset", env._h)


@step_.register(HipVecEnv)
def _(env, action):
    a = np.atleast_1d(np.asarray(action, np.int64))
    if a.size != env.N:
        raise PPOError(-1, "AssertionError: step! expects one action per env")
    if np.any(a < 1) or np.any(a > env.A):
        raise PPOError(-1, "AssertionError: Expected 0 < action_index <= %d" % env.A)   # quad_game_utilities.jl:178
    a0 = (a - 1).astype(np.int32)
    call("ppo_env_step", env._h, _p(a0, _lib.c_i32p))


# ------------------------------------------------------------------ policy
class HipPolicy:
    """SimplePolicy.Policy(in_channels, hidden_channels, num_hidden_layers, num_output) (test/policy.jl:9-19)
    with parameters resident on the GPU.  `params` is the flat Flux.params vector (W [out,in] column-major)."""

    DTYPES = {"f32": 0, "bf16": 1}

    def __init__(self, in_channels, hidden_channels, num_hidden_layers, num_output, seed=0, dtype="f32"):
        """dtype: arithmetic of the MLP's three Dense products -- "f32" (exact fp32 MFMA, the reference's Float32)
        or "bf16" (bf16 MFMA with fp32 accumulation, BASELINE config 5; ppo_policy_set_dtype)."""
        if dtype not in self.DTYPES:
            raise PPOError(-1, "AssertionError: dtype must be 'f32' or 'bf16'")
        h = C.c_void_p()
        call("ppo_policy_create", int(in_channels), int(hidden_channels), int(num_hidden_layers), int(num_output),
             C.byref(h))
        self._h = h
        self.dtype = dtype
        if dtype != "f32":
            call("ppo_policy_set_dtype", h, self.DTYPES[dtype])
        self.in_channels, self.hidden_channels = in_channels, hidden_channels
        self.num_hidden_layers, self.num_output = num_hidden_layers, num_output
        n = C.c_int64(0)
        call("ppo_policy_num_params", h, C.byref(n))
        self.num_params = n.value
        self.params = glorot_uniform_params(in_channels, hidden_channels, num_hidden_layers, num_output, seed)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().ppo_policy_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def params(self):
        out = np.empty(self.num_params, np.float32)
        call("ppo_policy_get_params", self._h, _p(out, _lib.c_f32p))
        return out

    @params.setter
    def params(self, flat):
        flat = np.ascontiguousarray(flat, np.float32)
        if flat.size != self.num_params:
            raise PPOError(-1, "AssertionError: expected %d parameters, got %d" % (self.num_params, flat.size))
        call("ppo_policy_set_params", self._h, _p(flat, _lib.c_f32p))

    def grad(self):
        out = np.empty(self.num_params, np.float32)
        call("ppo_policy_get_grad", self._h, _p(out, _lib.c_f32p))
        return out

    def grad_buffer_dev(self):
        ptr, n = C.c_void_p(), C.c_int64(0)
        call("ppo_policy_grad_buffer_dev", self._h, C.byref(ptr), C.byref(n))
        return ptr.value, n.value

    def layer_shapes(self):
        d = [(self.hidden_channels, self.in_channels)] + [(self.hidden_channels, self.hidden_channels)] * \
            (self.num_hidden_layers - 1) + [(self.num_output, self.hidden_channels)]
        return d


def glorot_uniform_params(F, HID, n_hidden, out, seed=0):
    """Flux default init: Glorot-uniform weights, zero bias; flat Flux order."""
    rng = np.random.default_rng(seed)
    parts = []
    for (o, i) in [(HID, F)] + [(HID, HID)] * (n_hidden - 1) + [(out, HID)]:
        lim = np.sqrt(6.0 / (o + i))
        W = rng.uniform(-lim, lim, size=(o, i)).astype(np.float32)
        parts += [W.ravel(order="F"), np.zeros(o, np.float32)]
    return np.concatenate(parts).astype(np.float32)


def _forward(policy, vs, bits):
    vs = np.ascontiguousarray(vs, np.int8)
    bits = np.ascontiguousarray(bits, np.uint32)
    B, Hh, F = vs.shape
    probs = np.empty((B, Hh * 4), np.float32)
    call("ppo_policy_forward", policy._h, _p(vs, _lib.c_i8p), _p(bits, _lib.c_u32p), B, Hh, _p(probs, _lib.c_f32p))
    return probs


@action_probabilities.register(HipPolicy)
def _(policy, s):
    """test/quad_game_utilities.jl:65-71 -> vector of length A."""
    return _forward(policy, np.asarray(s.vertex_score)[None], np.atleast_1d(np.uint32(s.action_mask)))[0]


@batch_action_probabilities.register(HipPolicy)
def _(policy, s):
    """test/quad_game_utilities.jl:73-79 -> [A,B] (returned as the transposed view of the C-order [B,A])."""
    return _forward(policy, s.vertex_score, s.action_mask).T


# ------------------------------------------------------------------ optimiser
class Adam:
    """Flux legacy Adam(eta, beta, epsilon)."""

    def __init__(self, eta=1e-3, beta=(0.9, 0.999), epsilon=1e-8):
        self.eta, self.beta, self.epsilon = float(eta), (float(beta[0]), float(beta[1])), float(epsilon)
        self._h = None
        self._policy = None

    def _bind(self, policy):
        if self._h is None:
            h = C.c_void_p()
            call("ppo_adam_create", policy._h, self.eta, self.beta[0], self.beta[1], self.epsilon, C.byref(h))
            self._h, self._policy = h, policy
        elif self._policy is not policy:
            raise PPOError(-1, "AssertionError: optimiser state belongs to another policy")
        call("ppo_adam_set_lr", self._h, float(self.eta))
        return self._h

    def __del__(self):
        try:
            if self._h:
                lib().ppo_adam_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def get_state(self):
        n = self._policy.num_params
        m, v, bp = np.empty(n, np.float32), np.empty(n, np.float32), np.empty(2, np.float64)
        call("ppo_adam_get_state", self._h, _p(m, _lib.c_f32p), _p(v, _lib.c_f32p), _p(bp, _lib.c_f64p))
        return m, v, bp


class Optimiser:
    """Flux.Optimiser(...): iterable composite (get_optimizer_learning_rate iterates it, src/train.jl:155-158)."""

    def __init__(self, *members):
        self.members = list(members)

    def __iter__(self):
        return iter(self.members)

    def _adam(self):
        adams = [m for m in self.members if isinstance(m, Adam)]
        if len(adams) != 1 or len(self.members) != 1:
            raise PPOError(-4, "only Optimiser(Adam(...)) is implemented on the device")
        return adams[0]


def get_optimizer_learning_rate(optimizer):
    lr = 1.0
    for opt in optimizer:          # a bare Adam is not iterable, like the reference
        lr *= opt.eta
    return lr


# ------------------------------------------------------------------ standalone ops
def compute_returns(rewards, terminal, discount):
    """src/collect_rollouts.jl:26-42.  A Python float discount is a Float64 (running value in fp64);
    pass np.float32(discount) for the all-Float32 variant."""
    r = np.ascontiguousarray(rewards, np.float32)
    t = np.ascontiguousarray(terminal, np.uint8)
    if r.shape != t.shape:
        raise PPOError(-1, "AssertionError: rewards and terminal differ in length")
    out = np.empty_like(r)
    call("ppo_compute_returns", _p(r, _lib.c_f32p), _p(t, _lib.c_u8p), r.size, float(discount),
         int(isinstance(discount, np.float32)), _p(out, _lib.c_f32p))
    return out


def compute_returns_tn(rewards, done, discount):
    r = np.ascontiguousarray(rewards, np.float32)
    d = np.ascontiguousarray(done, np.uint8)
    T, N = r.shape
    out = np.empty_like(r)
    call("ppo_compute_returns_tn", _p(r, _lib.c_f32p), _p(d, _lib.c_u8p), T, N, float(discount),
         int(isinstance(discount, np.float32)), _p(out, _lib.c_f32p))
    return out


def gae_tn(rewards, done, values, gamma, lam):
    r = np.ascontiguousarray(rewards, np.float32)
    d = np.ascontiguousarray(done, np.uint8)
    v = np.ascontiguousarray(values, np.float32)
    T, N = r.shape
    adv, ret = np.empty_like(r), np.empty_like(r)
    call("ppo_gae_tn", _p(r, _lib.c_f32p), _p(d, _lib.c_u8p), _p(v, _lib.c_f32p), T, N, float(gamma), float(lam),
         _p(adv, _lib.c_f32p), _p(ret, _lib.c_f32p))
    return adv, ret


def categorical_sample(probs, u):
    """rand(Categorical(p)) for rows of probs [B,A] with uniforms u[B]; returns 1-based actions,
    selected probabilities and the ap[a] > 0 assertion flags (src/collect_rollouts.jl:6-7)."""
    p = np.ascontiguousarray(probs, np.float32)
    uu = np.ascontiguousarray(u, np.float32)
    B, A = p.shape
    a, ps, err = np.empty(B, np.int32), np.empty(B, np.float32), np.empty(B, np.int32)
    call("ppo_categorical_sample", _p(p, _lib.c_f32p), _p(uu, _lib.c_f32p), B, A, _p(a, _lib.c_i32p),
         _p(ps, _lib.c_f32p), _p(err, _lib.c_i32p))
    return a.astype(np.int64) + 1, ps, err


def simplified_ppo_clip(advantage, epsilon):
    """src/train.jl:1-7 (scalar helper; the device path fuses it into the loss kernel)."""
    return (1.0 + epsilon) * advantage if advantage >= 0 else (1.0 - epsilon) * advantage


def get_linear_action_index(selected_actions, num_actions_per_state):
    """src/train.jl:48-52 (1-based in, 1-based out)."""
    a = np.ascontiguousarray(selected_actions, np.int64)
    out = np.empty_like(a)
    call("ppo_linear_action_index", _p(a, _lib.c_i64p), a.size, int(num_actions_per_state), _p(out, _lib.c_i64p))
    return out


def ppo_loss_with_entropy(probs_AB, linear_action_index, old_action_probabilities, advantage, epsilon):
    """Forward-only loss on given probabilities [A,B] (src/train.jl:35-46) -> (ppoloss, entropyloss)."""
    pr = np.ascontiguousarray(np.asarray(probs_AB, np.float32).T)          # [B,A] C-order == [A,B] column-major
    B, A = pr.shape
    li = np.ascontiguousarray(linear_action_index, np.int64)
    po = np.ascontiguousarray(old_action_probabilities, np.float32)
    ad = np.ascontiguousarray(advantage, np.float32)
    a, b = C.c_double(0), C.c_double(0)
    call("ppo_loss_with_entropy", _p(pr, _lib.c_f32p), _p(li, _lib.c_i64p), _p(po, _lib.c_f32p), _p(ad, _lib.c_f32p),
         B, A, float(epsilon), C.byref(a), C.byref(b))
    return a.value, b.value


def philox4x32_10(ctr, key):
    c = np.ascontiguousarray(ctr, np.uint32).reshape(-1, 4)
    k = np.ascontiguousarray(key, np.uint32)
    out = np.empty_like(c)
    call("ppo_philox4x32_10", _p(c, _lib.c_u32p), _p(k, _lib.c_u32p), c.shape[0], _p(out, _lib.c_u32p))
    return out


# ------------------------------------------------------------------ rollouts / dataset
class _Shape:
    """Stand-in for the env of a rollout buffer created by shape (N columns of [H][F] int8 state rows)."""

    def __init__(self, N, H, F):
        self.N, self.H, self.F, self.A = int(N), int(H), int(F), 4 * int(H)


class BufferRollouts:
    """PPO.BufferRollouts() (src/rollout_buffer.jl:1-22): device-resident SoA columns [T,N]."""

    def __init__(self):
        self._h = None
        self._env = None
        self._host = None        # generic (user-defined env) mode: the reference's own AoS columns (generic.HostRollouts)

    def _ensure(self, env, T):
        if self._h is None:
            h = C.c_void_p()
            if isinstance(env, _Shape):      # states of a user env (any F the policy kernels take): ppo_rollouts_create_shape
                call("ppo_rollouts_create_shape", env.N, env.H, env.F, int(T), C.byref(h))
            else:
                call("ppo_rollouts_create", env._h, int(T), C.byref(h))
            self._h, self._env = h, env
        return self._h

    def __del__(self):
        try:
            if self._h:
                lib().ppo_rollouts_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def __len__(self):                                     # Base.length (src/rollout_buffer.jl:40-48)
        if self._h is None and self._host is not None:
            return len(self._host)
        if self._h is None:
            return 0
        n = C.c_int64(0)
        call("ppo_rollouts_len", self._h, C.byref(n))
        return n.value

    def dims(self):
        T, N = C.c_int64(0), C.c_int64(0)
        call("ppo_rollouts_dims", self._h, C.byref(T), C.byref(N))
        return T.value, N.value

    def _get(self, fn, dtype, ctype, extra=()):
        T, N = self.dims()
        out = np.empty((T, N) + tuple(extra), dtype)
        call(fn, self._h, _p(out, ctype))
        return out

    # columns (time-major [T,N]); names follow the reference struct fields
    @property
    def selected_actions(self):
        return self._get("ppo_rollouts_get_actions", np.int32, _lib.c_i32p).astype(np.int64) + 1

    @property
    def selected_action_probabilities(self):
        return self._get("ppo_rollouts_get_probs", np.float32, _lib.c_f32p)

    @property
    def rewards(self):
        """After collect_rollouts! this column holds the RETURNS (compute_state_value! overwrites it,
        src/rollout_buffer.jl:55-64)."""
        return self._get("ppo_rollouts_get_returns", np.float32, _lib.c_f32p)

    @property
    def raw_rewards(self):
        return self._get("ppo_rollouts_get_raw_rewards", np.float32, _lib.c_f32p)

    @property
    def terminal(self):
        return self._get("ppo_rollouts_get_terminal", np.uint8, _lib.c_u8p).astype(bool)

    @property
    def valid(self):
        return self._get("ppo_rollouts_get_valid", np.uint8, _lib.c_u8p).astype(bool)

    @property
    def state_data(self):
        T, N = self.dims()
        env = self._env
        st = np.empty((T, N, env.H, env.F), np.int8)
        act = np.empty((T, N), np.uint32)
        call("ppo_rollouts_get_states", self._h, _p(st, _lib.c_i8p), _p(act, _lib.c_u32p))
        return st, act

    def full_probs(self):
        T, N = self.dims()
        out = np.empty((T, N, self._env.A), np.float32)
        call("ppo_rollouts_get_full_probs", self._h, _p(out, _lib.c_f32p))
        return out

    def index(self):
        """Dataset order: flat transition ids t*N+n of the valid transitions."""
        idx = np.empty(len(self), np.int64)
        call("ppo_rollouts_get_index", self._h, _p(idx, _lib.c_i64p))
        return idx

    def set_columns(self, env, states, active, actions1, p_sel, returns, terminal=None):
        """Load columns from the host (generic host-side envs, tests).  env = None: the buffer takes its shape from
        `states` [T,N,H,F] (a user env whose state rows are not the built-in env's 72 features)."""
        st = np.ascontiguousarray(states, np.int8)
        T = st.shape[0]
        if env is None:
            env = _Shape(st.shape[1], st.shape[2], st.shape[3])
        self._ensure(env, T)
        ac = np.ascontiguousarray(active, np.uint32)
        a0 = np.ascontiguousarray(np.asarray(actions1, np.int64) - 1, np.int32)
        ps = np.ascontiguousarray(p_sel, np.float32)
        rt = np.ascontiguousarray(returns, np.float32)
        tm = None if terminal is None else np.ascontiguousarray(terminal, np.uint8)
        call("ppo_rollouts_set", self._h, T, _p(st, _lib.c_i8p), _p(ac, _lib.c_u32p), _p(a0, _lib.c_i32p),
             _p(ps, _lib.c_f32p), _p(rt, _lib.c_f32p), _p(tm, _lib.c_u8p) if tm is not None else None)


def _discount_args(discount):
    return float(discount), int(isinstance(discount, np.float32))


def collect_rollouts_(rollouts, env, policy, num_episodes, discount):
    """PPO.collect_rollouts!(rollouts, env, policy, num_episodes, discount) (src/rollout_buffer.jl:66-79,
    src/rollouts_to_disk.jl:134-147).  Exactly num_episodes whole episodes enter the buffer: the N resident envs play
    them in parallel, episode e on env e mod N (reset! before each).  A DiskRollouts target additionally gets the
    reference's CSV + BSON layout."""
    if not isinstance(env, HipVecEnv):
        # generic method: the reference's own per-step control flow over the user's plugin methods (an env without a
        # `state` method raises "Function state needs to be overloaded" from inside it, like the reference)
        if isinstance(rollouts, DiskRollouts):
            raise PPOError(-4, "DiskRollouts with a user-defined env: use update_ / write_returns_to_disk (disk.py)")
        if rollouts._host is None:
            rollouts._host = HostRollouts()
        generic.collect_rollouts_host_(_this_module(), rollouts._host, env, policy, num_episodes, discount)
        return
    if not isinstance(policy, HipPolicy):
        _not_implemented("action_probabilities")
    if isinstance(rollouts, DiskRollouts):
        print("\n\nCOLLECTING ROLLOUTS :")                              # src/rollouts_to_disk.jl:141
        dev = BufferRollouts()
        collect_rollouts_(dev, env, policy, num_episodes, discount)
        rollouts._device = dev
        export_reference_layout(rollouts, dev)
        return
    if num_episodes < 1:
        raise PPOError(-1, "AssertionError: num_episodes must be >= 1")
    per_env = -(-int(num_episodes) // env.N)
    h = rollouts._ensure(env, per_env * env.max_actions)
    g, f32 = _discount_args(discount)
    call("ppo_collect_rollouts_episodes", h, env._h, policy._h, int(num_episodes), g, f32)


def collect_rollouts_steps_(rollouts, env, policy, num_steps, discount, record_probs=False, pinned_slots=16):
    """Vectorised fixed-T form: num_steps steps of all N envs with auto-reset (the throughput path).
    With a DiskRollouts target every finished step is streamed device -> pinned host -> <dir>/rollout.bin
    while the next step runs (ppo_rollouts_attach_disk)."""
    if isinstance(rollouts, DiskRollouts):
        dev = BufferRollouts()
        h = dev._ensure(env, 0)          # capacity comes with the collection (its storage form depends on the sink)
        # the constructor already wiped the directory; attach recreates it (same semantics) for the shard
        call("ppo_rollouts_attach_disk", h, rollouts.state_data_directory.encode(), int(pinned_slots))
        g, f32 = _discount_args(discount)
        call("ppo_collect_rollouts", h, env._h, policy._h, int(num_steps), g, f32, 0)
        if not _DISK_ASYNC[0]:
            call("ppo_rollouts_detach_disk", h)      # (deferred finish: disk_sync detaches once the file is complete)
        rollouts._device = dev
        rollouts.num_samples = len(dev)
        with open(rollouts.trajectory_filename, "w", newline="") as f:          # attach wiped it: keep the header
            f.write(",".join(DiskRollouts.FINAL) + "\n")
        return
    h = rollouts._ensure(env, int(num_steps))
    g, f32 = _discount_args(discount)
    call("ppo_collect_rollouts", h, env._h, policy._h, int(num_steps), g, f32, int(bool(record_probs)))


def compute_gae_(rollouts, values, gamma, lam):
    """batch_advantage as GAE(gamma, lambda): `values` [T+1, N] are the caller's state values (row T = bootstrap).  The
    advantage column stays on the device for ppo_train_(..., advantage="gae"); returns (advantages, lambda_returns)."""
    T, N = rollouts.dims()
    v = np.ascontiguousarray(values, np.float32)
    if v.shape != (T + 1, N):
        raise PPOError(-1, "AssertionError: values must be [T+1, N] = [%d, %d]" % (T + 1, N))
    adv, ret = np.empty((T, N), np.float32), np.empty((T, N), np.float32)
    call("ppo_rollouts_compute_gae", rollouts._h, _p(v, _lib.c_f32p), float(gamma), float(lam), _p(adv, _lib.c_f32p),
         _p(ret, _lib.c_f32p))
    return adv, ret


class BufferDataset:
    """src/rollout_buffer.jl:95-147.  Non-owning view; indices are 1-based like the reference."""

    def __init__(self, rollouts):
        self.rollouts = rollouts
        self._cache = None

    def __len__(self):
        return len(self.rollouts)

    def _columns(self):
        if self._cache is None:
            r = self.rollouts
            st, act = r.state_data
            idx = r.index()
            self._cache = dict(idx=idx, st=st.reshape(-1, st.shape[2], st.shape[3]), act=act.reshape(-1),
                               a=r.selected_actions.reshape(-1), p=r.selected_action_probabilities.reshape(-1),
                               ret=r.rewards.reshape(-1))
        return self._cache

    def __getitem__(self, idx):
        c = self._columns()
        n = len(self)
        if isinstance(idx, (int, np.integer)):
            if not (1 <= idx <= n):
                raise PPOError(-1, "AssertionError: 1 <= idx <= length(rollouts)")         # :105-106
            t = c["idx"][idx - 1]
            return {"state": StateData(c["st"][t], c["act"][t]), "selected_action": int(c["a"][t]),
                    "selected_action_probability": float(c["p"][t]), "returns": float(c["ret"][t])}
        if isinstance(idx, (list, tuple, np.ndarray)):
            ii = np.asarray(idx, np.int64)
            if ii.size and (ii.min() < 1 or ii.max() > n):
                raise PPOError(-1, "AssertionError: dataset index out of range")
            t = c["idx"][ii - 1]
            return {"state": StateData(c["st"][t], c["act"][t]), "selected_action": c["a"][t],
                    "selected_action_probability": c["p"][t], "returns": c["ret"][t]}
        raise PPOError(-1, "Dataset index should be Int or Array, got %s" % type(idx).__name__)   # :141


def _this_module():
    import sys
    return sys.modules[__name__]


def _upload_host_rollouts(rollouts):
    """HostRollouts of StateData states -> device buffer (ppo_rollouts_set), so ppo_train_ runs the MFMA path."""
    h = rollouts._host
    vs = np.stack([np.asarray(s.vertex_score, np.int8) for s in h.state_data])
    if vs.ndim != 3 or vs.shape[2] != 72 or vs.shape[1] not in (32, 128):
        raise PPOError(-4, "states must be [H,72] int8 with H in {32,128} for the gfx950 kernels")
    carrier = HipVecEnv(num_envs=1, Q=vs.shape[1] // 4, max_actions=max(1, len(h)))
    dev = BufferRollouts()
    dev.set_columns(carrier, vs[:, None], np.array([[np.uint32(s.action_mask)] for s in h.state_data], np.uint32),
                    np.asarray(h.selected_actions, np.int64)[:, None],
                    np.asarray(h.selected_action_probabilities, np.float32)[:, None],
                    np.asarray(h.rewards, np.float32)[:, None], np.asarray(h.terminal, np.uint8)[:, None])
    return dev


def construct_dataset(rollouts):
    """construct_dataset (src/rollout_buffer.jl:145-147, src/rollouts_to_disk.jl:169-171)."""
    if isinstance(rollouts, BufferRollouts) and rollouts._h is None and rollouts._host is not None:
        if len(rollouts._host) and all(isinstance(s, StateData) for s in rollouts._host.state_data):
            rollouts._device = _upload_host_rollouts(rollouts)
            return BufferDataset(rollouts._device)
        return HostDataset(_this_module(), rollouts._host)
    if isinstance(rollouts, DiskRollouts):
        if rollouts._device is not None:
            return BufferDataset(rollouts._device)       # columns are still resident: no reload needed
        return DiskDataset(rollouts.state_data_directory)
    return BufferDataset(rollouts)


def set_disk_async(on=None):
    """Streamed collections (DiskRollouts) return without waiting for the file: the pinned ring holds the whole collection and
    the writer thread finishes rollout.bin while training runs; disk_sync(rollouts) (or the next collection) waits for it.
    None / False = the file is complete when collect_rollouts_ returns (default, the reference's behaviour)."""
    call("ppo_set_disk_async", -1 if on is None else int(bool(on)))
    _DISK_ASYNC[0] = bool(on)


_DISK_ASYNC = [False]


def disk_sync(rollouts):
    """Wait until the rollout file of a streamed collection is complete, then detach the store (only needed after
    set_disk_async(True); a no-op otherwise)."""
    dev = getattr(rollouts, "_device", None) or rollouts
    if getattr(dev, "_h", None):
        call("ppo_rollouts_disk_sync", dev._h)
        call("ppo_rollouts_detach_disk", dev._h)


def load_disk_rollouts(state_data_dir, env):
    """DiskDataset over the engine's streaming shard: <dir>/rollout.bin -> device rollout buffer."""
    ro = BufferRollouts()
    h = ro._ensure(env, 1)
    call("ppo_rollouts_load_disk", h, state_data_dir.encode())
    return ro


# ------------------------------------------------------------------ training
ADVANTAGE_MODES = {"returns": 0, "returns_normalised": 1, "gae": 2, "gae_normalised": 3}


def _adv_mode(advantage):
    """batch_advantage plugin (src/ProximalPolicyOptimization.jl:29; no implementation in the reference): "returns"
    (identity, what the reference's scripts do), "returns_normalised" ((R - mean) / (std + 1e-8) per minibatch), "gae"
    (the GAE(gamma, lambda) column of compute_gae_) or "gae_normalised"."""
    if advantage not in ADVANTAGE_MODES:
        raise PPOError(-1, "AssertionError: advantage must be one of %s" % sorted(ADVANTAGE_MODES))
    return ADVANTAGE_MODES[advantage]


def step_batch_(policy, optimizer, dataset, batch_indices, epsilon, entropy_weight, advantage="returns"):
    """One optimiser step on dataset[batch_indices] (1-based): gather + batch_advantage (=returns) +
    get_linear_action_index + step_batch! (src/train.jl:98-120, 54-84).  Returns (ppoloss, entropy_weight*entropyloss)."""
    adam = optimizer._adam()
    oh = adam._bind(policy)
    ii = np.ascontiguousarray(np.asarray(batch_indices, np.int64) - 1)
    a, b = C.c_double(0), C.c_double(0)
    call("ppo_step_batch", policy._h, oh, dataset.rollouts._h, _p(ii, _lib.c_i64p), ii.size, float(epsilon),
         float(entropy_weight), _adv_mode(advantage), C.byref(a), C.byref(b))
    return a.value, b.value


def forward_backward(policy, dataset, batch_indices, epsilon, entropy_weight, B_global=None, advantage="returns"):
    """Gradient only (no update): leaves the flat gradient in policy.grad(); returns the two losses."""
    ii = np.ascontiguousarray(np.asarray(batch_indices, np.int64) - 1)
    call("ppo_forward_backward", policy._h, dataset.rollouts._h, _p(ii, _lib.c_i64p), ii.size,
         int(B_global or ii.size), float(epsilon), float(entropy_weight), _adv_mode(advantage))
    a, b = C.c_double(0), C.c_double(0)
    call("ppo_last_losses", policy._h, C.byref(a), C.byref(b))
    return a.value, b.value


def ppo_train_(policy, optimizer, dataset, epsilon, batch_size, num_epochs, entropy_weight, perm=None, seed=0,
               parallel=None, verbose=True, advantage="returns"):
    """PPO.ppo_train!(policy, optimizer, dataset, epsilon, batch_size, num_epochs, entropy_weight)
    (src/train.jl:130-153) -> (ppo_loss_history, entropy_loss_history, lr_history).
    perm: optional [num_epochs, len] 1-based permutations standing in for randperm (:93)."""
    adam = optimizer._adam()
    oh = adam._bind(policy)
    n = len(dataset)
    pp = None
    if perm is not None:
        pp = np.ascontiguousarray(np.asarray(perm, np.int64).reshape(num_epochs, n) - 1)
    ph, eh, lh = (np.zeros(num_epochs, np.float64) for _ in range(3))
    rank, world, fn, keep = 0, 1, _lib.ALLREDUCE_FN(0), None
    if parallel is not None and (parallel.world > 1 or parallel.force_hook):
        rank, world = parallel.rank, parallel.world
        keep = parallel.make_hook(policy)
        fn = keep
    # the data-parallel shards may differ in length: the batch_size assert is then made inside ppo_train on the
    # shortest shard, identically on every rank (a local raise here would leave the other ranks in a collective)
    if world == 1 and not (1 <= batch_size <= n):
        raise PPOError(-1, "AssertionError: 1 <= batch_size <= num_data")                    # :88
    call("ppo_train", policy._h, oh, dataset.rollouts._h, float(epsilon), int(batch_size), int(num_epochs),
         float(entropy_weight), _adv_mode(advantage), _p(pp, _lib.c_i64p) if pp is not None else None, int(seed),
         int(rank), int(world), fn, None,
         _p(ph, _lib.c_f64p), _p(eh, _lib.c_f64p), _p(lh, _lib.c_f64p))
    if verbose:
        for e in range(num_epochs):                                                         # :146
            print("EPOCH : %d \t PPO LOSS : %1.4f\t ENTROPY LOSS : %1.4f \t LR : %1.1e" % (e + 1, ph[e], eh[e], lh[e]))
    return list(ph), list(eh), list(lh)                     # lr = get_optimizer_learning_rate per epoch (:144)


def ppo_iterate_(policy, env, optimizer, episodes_per_iteration, minibatch_size, num_ppo_iterations, evaluator,
                 epochs_per_iteration, discount, epsilon, entropy_weight, state_data_path=None, verbose=True):
    """PPO.ppo_iterate! (src/train.jl:164-249), positional argument order preserved: 11 arguments = in-memory
    method, a 12th `state_data_path` = the disk method (rollouts through DiskRollouts, directory removed at
    the end, :198-201)."""
    loss = {"ppo": [], "entropy": [], "lr": []}
    for it in range(1, num_ppo_iterations + 1):
        evaluator(policy, env, optimizer)                                                    # :181,226
        if verbose:
            print("\nPPO ITERATION : %d" % it)
        rollouts = BufferRollouts() if state_data_path is None else DiskRollouts(state_data_path)   # :185,230
        collect_rollouts_(rollouts, env, policy, episodes_per_iteration, discount)
        dataset = construct_dataset(rollouts)
        p, e, lr = ppo_train_(policy, optimizer, dataset, epsilon, minibatch_size, epochs_per_iteration,
                              entropy_weight, verbose=verbose)
        loss["ppo"] += p
        loss["entropy"] += e
        loss["lr"] += lr
        save_loss(evaluator, loss)       # :196,247 -- like the reference, an evaluator without a save_loss method throws
    if state_data_path is not None and os.path.isdir(state_data_path):
        if verbose:
            print("\n\nCLEARING DATA IN ROLLOUTS FOLDER :")                                 # :199
        shutil.rmtree(state_data_path)
    return loss


def average_returns(policy, env, num_trajectories):
    """PPO.average_returns(policy, env, num_trajectories) (src/evaluate.jl:18-25) -> (mean, std) of the
    undiscounted episode return under the stochastic policy (std with the n-1 correction like Flux.std)."""
    if not isinstance(env, HipVecEnv):
        return generic.average_returns_host(_this_module(), policy, env, num_trajectories)
    scratch = BufferRollouts()
    per_env = -(-int(num_trajectories) // env.N)
    h = scratch._ensure(env, per_env * env.max_actions)
    m, s = C.c_double(0), C.c_double(0)
    call("ppo_average_returns", policy._h, env._h, h, int(num_trajectories), C.byref(m), C.byref(s))
    return m.value, s.value


def _evaluator_scratch(env):
    scratch = BufferRollouts()
    return scratch, scratch._ensure(env, 1)


def average_best_returns(env, policy, num_trajectories):
    """average_best_returns(wrapper, policy, num_trajectories) (test/quad_game_utilities.jl:299-307; note the
    reference's argument order: env first) -> (mean, std) of initial_score - lowest score seen on the trajectory."""
    scratch, h = _evaluator_scratch(env)
    m, s = C.c_double(0), C.c_double(0)
    call("ppo_average_best_returns", policy._h, env._h, h, int(num_trajectories), C.byref(m), C.byref(s))
    return m.value, s.value


def average_normalized_returns(env, policy, num_trajectories):
    """average_normalized_returns(wrapper, policy, num_trajectories) (test/quad_game_utilities.jl:380-387) ->
    (mean, std) of best return / (initial_score - opt_score); a trajectory that starts at its optimum counts 1.0."""
    scratch, h = _evaluator_scratch(env)
    m, s = C.c_double(0), C.c_double(0)
    call("ppo_average_normalized_returns", policy._h, env._h, h, int(num_trajectories), C.byref(m), C.byref(s))
    return m.value, s.value


def evaluate_trajectories(env, policy, num_trajectories, kind):
    """Per-trajectory values behind the three evaluators: kind "return" (single_trajectory_return, src/evaluate.jl:1-16),
    "best" (best_single_trajectory_return, test/quad_game_utilities.jl:280-296) or "normalized"
    (single_trajectory_normalized_return, :369-378).  Env-major order: env n's trajectories are consecutive."""
    k = {"return": 1, "best": 2, "normalized": 3}[kind]
    scratch, h = _evaluator_scratch(env)
    out = np.zeros(int(num_trajectories), np.float64)
    call("ppo_evaluate_trajectories", policy._h, env._h, h, int(num_trajectories), k, _p(out, _lib.c_f64p))
    return out


# ------------------------------------------------------------------ data parallel (one process per GPU)
class _DevArray:
    """Exposes a raw device pointer through __cuda_array_interface__ so torch can alias it."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 3,
                                         "strides": None}


class DataParallel:
    """Env shards are data-parallel across ranks; the only exchange is one all-reduce (sum) of the flat
    gradient buffer [num_params + 2] per optimiser step, through torch.distributed (backend "nccl" = RCCL
    over xGMI on MI355X; "gloo" for the CPU rehearsal of the host logic)."""

    def __init__(self, rank=0, world=1, force_hook=False):
        self.rank, self.world = int(rank), int(world)
        self.force_hook = bool(force_hook)      # exercise the all-reduce hook even with one rank (tests)
        self.hook_kind = None                   # set by make_hook: "native-rccl" or "torch.distributed/<backend>"

    def env_shard(self, total_envs):
        """Contiguous shard [offset, offset+n) of this rank; RNG uses global env ids."""
        base, rem = divmod(int(total_envs), self.world)
        n = base + (1 if self.rank < rem else 0)
        off = self.rank * base + min(self.rank, rem)
        return off, n

    @staticmethod
    def allreduce_(tensor):
        """Sum all-reduce in place.  "nccl" (= RCCL) reduces the device tensor on the current stream; with the "gloo"
        backend (CPU rehearsal: several ranks sharing one GPU in the tests) a device tensor takes a host round trip."""
        import torch.distributed as dist
        if tensor.is_cuda and dist.get_backend() == "gloo":
            host = tensor.cpu()                       # synchronises with the stream the engine runs on
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            tensor.copy_(host)
            return tensor
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
        return tensor

    def _native_rccl_hook(self):
        global _NATIVE_RCCL_HOOK                   # one in-library communicator per process, created on first use
        if _NATIVE_RCCL_HOOK is False:
            _NATIVE_RCCL_HOOK = _native_rccl_hook_impl(self.rank, self.world)
        return _NATIVE_RCCL_HOOK

    def make_hook(self, policy):
        """The ppo_allreduce_fn handed to ppo_train.  Default: the library's own RCCL all-reduce (ppo_rccl_*), adopted
        only if EVERY rank brought its communicator up and passed the known-answer all-reduce (the decision is itself
        collective, so the ranks can never end up on different collectives); otherwise -- or with PPO_NATIVE_RCCL=0, or
        on the gloo rehearsal backend -- one torch.distributed all-reduce per step on a tensor aliasing the buffer."""
        import torch
        import torch.distributed as dist
        # the engine must run on the stream torch orders its collectives against
        call("ppo_set_stream", C.c_void_p(torch.cuda.current_stream().cuda_stream))
        ptr, n = policy.grad_buffer_dev()
        if os.environ.get("PPO_NATIVE_RCCL", "1") != "0" and dist.get_backend() != "gloo":
            native = self._native_rccl_hook()
            if native is not None:
                self.hook_kind = "native-rccl"
                return native
        self.hook_kind = "torch.distributed/" + dist.get_backend()

        views = {}

        def reduce(dev_ptr, n_floats):             # also serves ppo_train's small shard-length exchange
            t = views.get((dev_ptr, n_floats)) if dev_ptr == ptr else None
            if t is None:
                t = torch.as_tensor(_DevArray(dev_ptr, n_floats), device="cuda")
                if dev_ptr == ptr:                 # the gradient buffer lives as long as the policy: keep its view
                    views[(dev_ptr, n_floats)] = t
            self.allreduce_(t)

        def hook(ctx, dev_ptr, n_floats):
            try:
                reduce(dev_ptr, n_floats)
                return 0
            except Exception:                      # noqa: BLE001 -- reported through the status code
                return 1
        return _lib.ALLREDUCE_FN(hook)


_NATIVE_RCCL_HOOK = False                          # False: not tried yet; None: not adopted (torch hook stays)


def _native_rccl_hook_impl(rank, world):
    """ppo_rccl_* (include/ppo_hip.h): the unique id travels over the host's process group, the all-reduce itself is
    one RCCL call made by the library on its own stream.  Every stage is followed by a MIN all-reduce of a success
    flag over the host group, and rank 0 always reaches the broadcast (sending a zero id if it could not make one):
    either all ranks adopt the native communicator or all of them keep the torch.distributed hook."""
    import torch
    import torch.distributed as dist

    def all_ok(ok):
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    why = ""
    uid = np.zeros(128, np.uint8)
    # stage 0: every rank must be able to resolve RCCL before any rank enters the collective ncclCommInitRank
    probe = lib().ppo_rccl_probe() == 0
    if not all_ok(probe):
        sys.stderr.write("rank %d: native RCCL hook not adopted (librccl %s here); every rank uses the torch.distributed hook\n"
                         % (rank, "resolved" if probe else "not found"))
        return None
    if rank == 0:
        try:
            call("ppo_rccl_unique_id", uid.ctypes.data_as(C.c_void_p))
        except Exception as e:                  # noqa: BLE001
            uid[:] = 0
            why = str(e)
    t = torch.from_numpy(uid).cuda()
    dist.broadcast(t, 0)
    uid = np.ascontiguousarray(t.cpu().numpy())
    ok = bool(uid.any())
    if ok:
        try:
            call("ppo_rccl_init", int(rank), int(world), uid.ctypes.data_as(C.c_void_p))
        except Exception as e:                  # noqa: BLE001
            ok, why = False, str(e)
    if all_ok(ok):
        try:                                    # collective known-answer all-reduce on the new communicator
            good = C.c_int32(0)
            call("ppo_rccl_self_test", C.byref(good))
            r_, w_ = C.c_int32(-1), C.c_int32(-1)
            call("ppo_rccl_comm_info", C.byref(r_), C.byref(w_))
            ok = bool(good.value) and r_.value == rank and w_.value == world
            if not ok:
                why = "known-answer all-reduce / communicator size check failed"
        except Exception as e:                  # noqa: BLE001
            ok, why = False, str(e)
        if all_ok(ok):
            return _lib.ALLREDUCE_FN(C.cast(_lib.lib().ppo_rccl_allreduce, C.c_void_p).value)
    try:
        lib().ppo_rccl_finalize()
    except Exception:                           # noqa: BLE001
        pass
    sys.stderr.write("rank %d: native RCCL hook not adopted (%s); every rank uses the torch.distributed hook\n"
                     % (rank, why or "another rank failed"))
    return None


def rccl_finalize():
    """Destroy the library's communicator (if any) and forget the cached native hook, so a later process group can
    build a new one."""
    global _NATIVE_RCCL_HOOK
    _NATIVE_RCCL_HOOK = False
    call("ppo_rccl_finalize")


def rccl_comm_info():
    """(rank, world) as the library's own communicator reports them, or None when the native hook is not in use."""
    r_, w_ = C.c_int32(-1), C.c_int32(-1)
    if lib().ppo_rccl_comm_info(C.byref(r_), C.byref(w_)) != 0:
        return None
    return r_.value, w_.value


# ---------------------------------------------------------------- checkpoints (BSON.@save / BSON.@load of the policy)
def save_policy(path, policy):
    """BSON.@save path policy (examples/triangle/distance_weighted/triangle_utilities.jl:370-373): the document
    BSON.jl writes for SimplePolicy.Policy, readable by the reference."""
    checkpoint.save_policy(path, policy)


def load_policy(path, seed=0, dtype="f32"):
    """BSON.@load path policy -> HipPolicy holding the saved Float32 weights (e.g. the reference's test/output/*.bson)."""
    return checkpoint.load_policy(path, HipPolicy, seed=seed, dtype=dtype)


# ---------------------------------------------------------------- remaining reference entry points (host mirrors)
def update_rollouts_(buffer, state, action_probability, action, reward, terminal):
    """update!(buffer::BufferRollouts, ...) (src/rollout_buffer.jl:24-38) for user-driven collection."""
    if buffer._host is None:
        buffer._host = HostRollouts()
    generic.update_host_(buffer._host, state, action_probability, action, reward, terminal)


def compute_state_value_(rollouts, discount):
    """compute_state_value!(rollouts, discount) (src/rollout_buffer.jl:55-64), generic (host-collected) rollouts."""
    generic.compute_state_value_(_this_module(), rollouts._host, discount)


def collect_step_data_(buffer, env, policy, rng=None):
    """collect_step_data!(buffer, env, policy) (src/collect_rollouts.jl:1-15)."""
    if buffer._host is None:
        buffer._host = HostRollouts()
    generic.collect_step_data_(_this_module(), buffer._host, env, policy, rng or np.random.default_rng())


def collect_episode_data_(buffer, env, policy, rng=None):
    """collect_episode_data!(buffer, env, policy) (src/collect_rollouts.jl:17-24)."""
    if buffer._host is None:
        buffer._host = HostRollouts()
    generic.collect_episode_data_(_this_module(), buffer._host, env, policy, rng or np.random.default_rng())


def permute_(rollouts, idx):
    """permute!(rollouts, idx) (src/rollout_buffer.jl:81-88)."""
    generic.permute_(rollouts._host, idx)


def shuffle_(rollouts, rng=None):
    """shuffle!(rollouts) (src/rollout_buffer.jl:90-93)."""
    generic.shuffle_(rollouts._host, rng)


def single_trajectory_return(policy, env, rng=None):
    """single_trajectory_return(policy, env) (src/evaluate.jl:1-16)."""
    return generic.single_trajectory_return(_this_module(), policy, env, rng)


def smoothed_entropy(probs_AB, smooth=np.float32(1e-8)):
    """smoothed_entropy(probs, smooth) (src/train.jl:21-26) on probs [A,B] -> mean entropy (Float32 arithmetic)."""
    p = np.asarray(probs_AB, np.float32)
    sp = (np.float32(1.0) - np.float32(smooth)) * p + np.float32(smooth) / np.float32(p.shape[0])
    return float(np.mean(-np.sum(sp * np.log(sp), axis=0, dtype=np.float32), dtype=np.float32))


def clamped_entropy(probs_AB, clamp=np.float32(1e-8)):
    """clamped_entropy(probs, clamp) (src/train.jl:28-33; unused by the reference's training loop)."""
    p = np.clip(np.asarray(probs_AB, np.float32), np.float32(clamp), None)
    return float(np.mean(-np.sum(p * np.log(p), axis=0, dtype=np.float32), dtype=np.float32))


def ppo_loss(probs_AB, linear_action_index, old_action_probabilities, advantage, epsilon):
    """ppo_loss (src/train.jl:9-19): the clipped-surrogate term alone (ppo_loss_with_entropy without the entropy)."""
    return ppo_loss_with_entropy(probs_AB, linear_action_index, old_action_probabilities, advantage, epsilon)[0]


def step_epoch_(policy, optimizer, dataset, epsilon, batch_size, entropy_weight, perm=None, seed=0, advantage="returns"):
    """step_epoch!(policy, optimizer, dataset, epsilon, batch_size, entropy_weight) (src/train.jl:86-128):
    one pass over a fresh permutation; returns the unweighted means of the per-batch losses (:127)."""
    p, e, _ = ppo_train_(policy, optimizer, dataset, epsilon, batch_size, 1, entropy_weight,
                         perm=None if perm is None else np.asarray(perm)[None, :], seed=seed, verbose=False,
                         advantage=advantage)
    return p[0], e[0]
