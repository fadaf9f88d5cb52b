"""GPU parity tests (run on a real MI355X: `pytest -m gpu`).  Every test drives the HIP engine through
the C ABI (via the Python mirror) and checks it against the CPU oracle on the same seeded inputs.
Bar: bit-exact for integer / index / byte work and for everything on the rollout path (the oracle's
device-order mode mirrors the kernels' fp32 operation order); stated tolerances for the loss / gradient
(compared against the float64 oracle)."""
import csv
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P(ppo):
    if ppo.device_count() < 1:
        pytest.fail("no HIP device: the gpu-marked tests must run on the GPU box")
    yield ppo
    ppo.set_bwd_split_bf16(None)          # (a test that switches the training-pass arithmetic and fails leaves the default behind)


# ---------------------------------------------------------------- K6 returns / GAE
@pytest.mark.parametrize("T,N", [(1, 1), (6, 1), (128, 1), (129, 65), (127, 64), (300, 200), (128, 4096),
                                 (130, 16384), (33, 16388)])
@pytest.mark.parametrize("gamma", [1.0, 0.99, np.float32(0.99)])
def test_returns_tn_bitexact(P, orc, T, N, gamma):
    rng = np.random.default_rng(T * 1000 + N)
    r = (rng.normal(size=(T, N)) * 3).astype(np.float32)
    d = (rng.random((T, N)) < 0.03).astype(np.uint8)
    got = P.compute_returns_tn(r, d, gamma)
    want = orc.compute_returns_tn(r, d, float(gamma), isinstance(gamma, np.float32))
    assert np.array_equal(got, want)


def test_returns_flat_golden_and_random(P, orc, golden_dir):
    rows = list(csv.DictReader(open(os.path.join(golden_dir, "trajectory.csv"))))
    expect = np.array([float(r["returns"]) for r in rows], np.float32)
    assert np.array_equal(P.compute_returns(np.ones(6, np.float32), [0, 0, 0, 0, 0, 1], 1.0), expect)
    assert np.array_equal(P.compute_returns(np.ones(6, np.float32), [0, 0, 0, 0, 0, 0], 1.0), expect)
    k = json.load(open(os.path.join(golden_dir, "known_answers.json")))["returns_test_env"]
    term = np.zeros(100, np.uint8)
    term[9::10] = 1
    assert np.array_equal(P.compute_returns(np.ones(100, np.float32), term, 1.0),
                          np.tile(np.array(k["returns_per_episode"], np.float32), 10))
    rng = np.random.default_rng(3)
    for n in (1, 2, 1000, 50000):
        r = rng.normal(size=n).astype(np.float32)
        t = (rng.random(n) < 0.02).astype(np.uint8)
        for g in (1.0, 0.97, np.float32(0.97)):
            assert np.array_equal(P.compute_returns(r, t, g), orc.compute_returns(r, t, float(g), isinstance(g, np.float32)))
    assert P.compute_returns(np.zeros(0, np.float32), np.zeros(0, np.uint8), 1.0).size == 0      # empty input


def test_returns_rel_tolerance_vs_pure_f32(P):
    # BASELINE: returns within 1e-5 (relative) of the reference definition in either float mode
    rng = np.random.default_rng(9)
    r = rng.normal(size=(128, 512)).astype(np.float32)
    d = np.zeros((128, 512), np.uint8)
    a, b = P.compute_returns_tn(r, d, 0.99), P.compute_returns_tn(r, d, np.float32(0.99))
    assert np.allclose(a, b, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("T,N", [(77, 130), (1, 1), (128, 16384), (45, 16388), (33, 65536), (16, 262144)])
def test_gae(P, orc, T, N):
    """GAE(gamma, lambda) scan, both kernels (one lane per column below 16384 columns, the LDS-tiled wide form above:
    128- and 256-column workgroups, ragged last pass, partly filled last workgroup): bit-exact against the fp64 oracle;
    lambda = 1, V = 0 is compute_returns (src/collect_rollouts.jl:26-42) bit for bit."""
    rng = np.random.default_rng(T + N)
    r = rng.normal(size=(T, N)).astype(np.float32)
    d = (rng.random((T, N)) < 0.05).astype(np.uint8)
    v = rng.normal(size=(T + 1, N)).astype(np.float32)
    adv, ret = P.gae_tn(r, d, v, 0.99, 0.95)
    oadv, oret = orc.gae_tn(r, d, v, 0.99, 0.95)
    assert np.array_equal(adv, oadv) and np.array_equal(ret, oret)
    adv1, _ = P.gae_tn(r, d, np.zeros_like(v), 0.99, 1.0)           # lambda=1, V=0 == returns
    assert np.array_equal(adv1, P.compute_returns_tn(r, d, 0.99))
    assert np.array_equal(adv1, orc.compute_returns_tn(r, d, 0.99))


def test_gae_advantage_mode_in_training(P, orc):
    """batch_advantage = GAE (PPO_ADV_GAE): values from the host, scan on the device over the buffer's own raw rewards,
    the advantage column consumed by the fused loss.  Gradient vs the float64 oracle fed with the oracle's GAE column;
    with lambda = 1, V = 0 the mode reproduces the returns mode bit for bit."""
    env = P.HipVecEnv(num_envs=24, Q=8, max_actions=9, seed=8)
    pol = P.HipPolicy(72, 128, 2, 4, seed=6)
    ro = P.BufferRollouts()
    P.collect_rollouts_steps_(ro, env, pol, 12, 0.97)
    ds = P.construct_dataset(ro)
    T, N = ro.dims()
    sel = np.random.default_rng(3).permutation(len(ds))[:150] + 1
    with pytest.raises(P.PPOError, match="compute_gae"):
        P.forward_backward(pol, ds, sel, 0.05, 0.01, advantage="gae")
    values = np.random.default_rng(4).normal(size=(T + 1, N)).astype(np.float32)
    adv, lret = P.compute_gae_(ro, values, 0.97, 0.9)
    oadv, oret = orc.gae_tn(ro.raw_rewards, ro.terminal.astype(np.uint8), values, 0.97, 0.9)
    assert np.array_equal(adv, oadv) and np.array_equal(lret, oret)
    st, act = ro.state_data
    s0 = sel - 1
    cols = (st.reshape(-1, 32, 72)[s0], act.reshape(-1)[s0], (ro.selected_actions.reshape(-1)[s0] - 1).astype(np.int32),
            ro.selected_action_probabilities.reshape(-1)[s0])
    for mode, a in (("gae", oadv.reshape(-1)[s0]),
                    ("gae_normalised", None)):
        if a is None:
            x = oadv.reshape(-1)[s0].astype(np.float64)
            a = ((x - x.mean()) / (x.std() + 1e-8)).astype(np.float32)
        lp, le = P.forward_backward(pol, ds, sel, 0.05, 0.01, advantage=mode)
        g64, olp, ole = orc.step_batch_grad_f64(pol.params, 72, 128, *cols, a, 0.05, 0.01)
        assert np.abs(pol.grad() - g64).max() <= 2e-5 * np.abs(g64).max() + 1e-9
        assert abs(lp - olp) <= 1e-5 * (1 + abs(olp)) and abs(le - ole) <= 1e-5 * (1 + abs(ole))
    # lambda = 1, V = 0: the GAE column IS the returns column
    adv1, _ = P.compute_gae_(ro, np.zeros((T + 1, N), np.float32), 0.97, 1.0)
    assert np.array_equal(adv1, ro.rewards)
    P.forward_backward(pol, ds, sel, 0.05, 0.01, advantage="gae")
    g_gae = pol.grad()
    P.forward_backward(pol, ds, sel, 0.05, 0.01, advantage="returns")
    assert np.array_equal(g_gae, pol.grad())
    # a whole ppo_train! epoch in GAE mode runs and moves the parameters
    before = pol.params.copy()
    P.ppo_train_(pol, P.Optimiser(P.Adam(1e-3)), ds, 0.05, 64, 1, 0.01, seed=3, verbose=False, advantage="gae")
    assert not np.array_equal(before, pol.params) and np.all(np.isfinite(pol.params))
    # new rollouts invalidate the column
    P.collect_rollouts_steps_(ro, env, pol, 12, 0.97)
    with pytest.raises(P.PPOError, match="compute_gae"):
        P.forward_backward(pol, P.construct_dataset(ro), sel, 0.05, 0.01, advantage="gae")


# ---------------------------------------------------------------- RNG / sampling / index ops
def test_philox_kat(P, orc, golden_dir):
    for c in json.load(open(os.path.join(golden_dir, "known_answers.json")))["philox4x32_10_kat"]["cases"]:
        exp = np.array([int(x, 16) for x in c["out"]], np.uint32)
        assert np.array_equal(P.philox4x32_10(c["ctr"], c["key"])[0], exp)
    rng = np.random.default_rng(0)
    ctr = rng.integers(0, 2**32, size=(100, 4), dtype=np.uint64).astype(np.uint32)
    key = np.array([1234, 99], np.uint32)
    got = P.philox4x32_10(ctr, key)
    for i in range(100):
        assert np.array_equal(got[i], orc.philox(ctr[i], key))


def test_categorical_bitexact(P, orc):
    rng = np.random.default_rng(1)
    B, A = 600, 128
    p = rng.random((B, A)).astype(np.float32)
    p[rng.random((B, A)) < 0.3] = 0
    p[:, 7] += 0.1
    p = (p / p.sum(axis=1, keepdims=True, dtype=np.float32)).astype(np.float32)
    p[0] = 0
    p[0, 0] = 0.5                     # clamp case: walk runs off the end onto a zero entry
    u = rng.random(B).astype(np.float32)
    u[0] = 0.75
    a, ps, err = P.categorical_sample(p, u)
    for b in range(B):
        oa, oerr = orc.categorical_sample(p[b], u[b])
        assert a[b] == oa + 1 and err[b] == oerr and ps[b] == p[b, oa]
    assert err[0] == 1 and a[0] == A


def test_linear_action_index_and_loss(P, orc):
    rng = np.random.default_rng(2)
    B, A = 37, 128
    a1 = rng.integers(1, A + 1, B)
    lin = P.get_linear_action_index(a1, A)
    assert np.array_equal(lin, a1 + np.arange(B) * A)
    logits = rng.normal(size=(B, A)).astype(np.float32) * 2
    probs = np.stack([orc.masked_softmax(logits[b], 0x3F) for b in range(B)])
    a1 = rng.integers(1, 97, B)
    lin = P.get_linear_action_index(a1, A)
    p_old = (probs.reshape(-1)[lin - 1] * rng.uniform(0.8, 1.25, B)).astype(np.float32)
    adv = rng.normal(size=B).astype(np.float32)
    lp, le = P.ppo_loss_with_entropy(probs.T, lin, p_old, adv, 0.05)
    olp, ole = orc.ppo_loss_with_entropy(probs, lin, p_old, adv, 0.05)
    assert abs(lp - olp) <= 1e-6 * (1 + abs(olp))        # fp32 gain, fp64 clip/mean: same arithmetic
    assert abs(le - ole) <= 1e-5 * (1 + abs(ole))        # fp32 entropy sum, different summation order


# ---------------------------------------------------------------- K1 env
def test_env_step_bitexact(P, orc):
    N = 96
    env = P.HipVecEnv(num_envs=N, Q=8, max_actions=20, seed=42, global_offset=1000)
    oenv = orc.Env(Q=8, max_actions=20, N=N, seed=42, global_offset=1000)
    oenv.reset()
    rng = np.random.default_rng(0)

    def compare():
        it = env.internal()
        s = P.state(env)
        assert np.array_equal(it["score"], oenv.score) and np.array_equal(it["degree"], oenv.degree)
        assert np.array_equal(it["steps"], oenv.steps) and np.array_equal(it["tick"], oenv.tick)
        assert np.array_equal(s.action_mask, oenv.active)
        assert np.array_equal(s.vertex_score, oenv.observe())

    compare()
    for t in range(60):
        done = P.is_terminal(env)
        assert np.array_equal(done, oenv.done.astype(bool))
        if done.any():                         # reset! (all envs, like a vectorised reset)
            P.reset_(env)
            oenv.reset()
            compare()
        acts = np.zeros(N, np.int64)
        for n in range(N):
            q = rng.choice([i for i in range(8) if (int(oenv.active[n]) >> i) & 1])
            acts[n] = 16 * q + rng.integers(0, 16) + 1
        P.step_(env, acts)
        oenv.step(acts - 1)
        assert np.array_equal(P.reward(env), oenv.reward)
        compare()
    with pytest.raises(P.PPOError):
        P.step_(env, np.zeros(N, np.int64))          # 0 < action_index <= A


def test_env_inactive_quad_flag(P):
    env = P.HipVecEnv(num_envs=2, Q=8, max_actions=20, seed=1)
    with pytest.raises(P.PPOError, match="inactive quad"):
        P.step_(env, np.array([128, 1]))             # quad 8 is inactive after reset


# ---------------------------------------------------------------- K2/K3 policy forward
@pytest.mark.parametrize("F,HID,fixture", [(72, 128, "poly-30-policy"), (72, 128, "catmull-clark-policy"),
                                           (216, 128, "catmull-clark-policy-l4"), (72, 256, None), (72, 128, None)])
def test_policy_forward(P, orc, golden_dir, F, HID, fixture):
    rng = np.random.default_rng(F + HID)
    pol = P.HipPolicy(F, HID, 2, 4, seed=3)
    if fixture:
        pol.params = np.load(os.path.join(golden_dir, fixture + ".npz"))["params"]       # reference-trained weights
    else:
        pol.params = pol.params + (rng.normal(size=pol.num_params) * 0.02).astype(np.float32)
    params = pol.params
    B = 50
    states = rng.integers(-3, 7, size=(B, 32, F)).astype(np.int8)
    active = rng.integers(1, 256, size=B).astype(np.uint32)
    probs = P.batch_action_probabilities(pol, P.StateData(states, active)).T        # [B,A]
    assert probs.shape == (B, 128)
    for b in range(B):
        dev = orc.action_probabilities(params, F, HID, states[b], active[b], "dev")
        assert np.array_equal(probs[b], dev), "device-order oracle must match bit for bit"
        mask = orc.action_mask([(int(active[b]) >> q) & 1 for q in range(8)])
        assert np.all(probs[b][np.isneginf(mask)] == 0.0)
        assert abs(float(probs[b].sum()) - 1.0) < 1e-5
        if not fixture:        # random-init weights: compare with the natural-order fp32 restatement
            ref = orc.action_probabilities(params, F, HID, states[b], active[b], "ref")
            assert np.allclose(probs[b], ref, rtol=2e-5, atol=1e-8)
        # INDEPENDENT check, every weight set (the reference-trained ones have logits of O(10-100), so probabilities are
        # compared in logit space): float64 restatement in natural order -> masked log-softmax.  Tolerance: fp32
        # accumulation over 72..256-term dot products, |d log p| <= 1e-4 * max(1, max|logit|).
        l64 = orc.mlp_logits(params, F, HID, states[b], "f64")
        on = ~np.isneginf(mask)
        lse = np.log(np.exp(l64[on] - l64[on].max()).sum()) + l64[on].max()
        live = on & (probs[b] > 1e-30)
        assert live.sum() >= 1
        tol = 1e-4 * max(1.0, float(np.abs(l64[on]).max()))
        assert np.abs(np.log(probs[b][live].astype(np.float64)) - (l64[live] - lse)).max() <= tol
        dead = on & ~live                       # underflowed on the device: the f64 value must be negligible too
        assert np.all(l64[dead] - lse < -60.0)
    single = P.action_probabilities(pol, P.StateData(states[0], active[0]))
    assert np.array_equal(single, probs[0])


@pytest.mark.parametrize("hid", [64, 96, 160, 224, 50, 1])
def test_policy_any_hidden_width(P, orc, hid):
    """SimplePolicy.Policy(72, hidden, 2, 4) for hidden widths the kernels are not built for (test/policy.jl:9-19 takes any):
    the engine runs them zero-padded on the 128 / 256 kernels, which is exact -- multiples of 32 reproduce the
    device-order oracle at the caller's width bit for bit (rollout: actions, probabilities), every width matches the
    float64 oracle's gradient, and what crosses the ABI (parameters, gradient, Adam moments) has the caller's layout."""
    env = P.HipVecEnv(num_envs=16, Q=8, max_actions=9, seed=23)
    pol = P.HipPolicy(72, hid, 2, 4, seed=hid)
    n_user = 72 * hid + hid + hid * hid + hid + 4 * hid + 4
    assert pol.num_params == n_user
    rng = np.random.default_rng(hid)
    p0 = (pol.params + (rng.normal(size=n_user) * 0.05).astype(np.float32)).astype(np.float32)
    pol.params = p0
    assert np.array_equal(pol.params, p0), "parameters round-trip in the caller's Flux layout"
    ro = P.BufferRollouts()
    P.collect_rollouts_steps_(ro, env, pol, 10, 1.0)
    st, act = ro.state_data
    if hid % 32 == 0:
        oenv = orc.Env(Q=8, max_actions=9, N=16, seed=23)
        oenv.reset()
        ref = orc.collect_rollouts_tn(oenv, p0, hid, 10, mode_dev=True)
        assert np.array_equal(ro.selected_actions - 1, ref["actions"]) and np.array_equal(ro.selected_action_probabilities, ref["p_sel"])
    else:
        for t, n in ((0, 0), (5, 3), (9, 15)):
            pr = orc.action_probabilities(p0, 72, hid, st[t, n], act[t, n], "ref")
            a = int(ro.selected_actions[t, n]) - 1
            assert abs(ro.selected_action_probabilities[t, n] - pr[a]) <= 2e-5 * pr[a] + 1e-8
    ds = P.construct_dataset(ro)
    sel = rng.permutation(len(ds))[:100] + 1
    for small in (0, 4096):                       # both kernel sets
        P.set_bwd_small_max_tiles(small); P.set_fwd_split_max_states(min(small, 512))
        try:
            lp, le = P.forward_backward(pol, ds, sel, 0.05, 0.01)
        finally:
            P.set_bwd_small_max_tiles(None); P.set_fwd_split_max_states(None)
        g = pol.grad()
        assert g.shape == (n_user,)
        s0 = sel - 1
        g64, olp, ole = orc.step_batch_grad_f64(p0, 72, hid, st.reshape(-1, 32, 72)[s0], act.reshape(-1)[s0],
                                                (ro.selected_actions.reshape(-1)[s0] - 1).astype(np.int32),
                                                ro.selected_action_probabilities.reshape(-1)[s0], ro.rewards.reshape(-1)[s0], 0.05, 0.01)
        assert np.abs(g - g64).max() <= 2e-5 * np.abs(g64).max() + 1e-9
        assert abs(lp - olp) <= 1e-5 * (1 + abs(olp)) and abs(le - ole) <= 1e-5 * (1 + abs(ole))
    # training: Adam on the caller's parameters equals the oracle's Adam on them (the padded units never move)
    opt = P.Optimiser(P.Adam(1e-3))
    P.step_batch_(pol, opt, ds, sel, 0.05, 0.01)
    m, v, bp = opt.members[0].get_state()
    assert m.shape == v.shape == (n_user,)
    pp, mm, vv, bb = p0.copy(), np.zeros_like(p0), np.zeros_like(p0), np.array([0.9, 0.999])
    orc.adam_step(pp, pol.grad(), mm, vv, bb, 1e-3)
    assert np.array_equal(pol.params, pp) and np.array_equal(m, mm) and np.array_equal(v, vv)
    with pytest.raises(P.PPOError):
        P.HipPolicy(72, 300, 2, 4)
    with pytest.raises(P.PPOError):
        P.HipPolicy(72, 128, 5, 4)             # num_hidden_layers 1..4 (tests/test_gpu_deep_policy.py)
    with pytest.raises(P.PPOError):
        P.HipPolicy(72, 128, 2, 5)             # the quad game has 4 actions per edge


# ---------------------------------------------------------------- rollout (K1-K6 end to end)
@pytest.mark.parametrize("N,T,HID,max_actions", [(64, 40, 128, 16), (8, 24, 256, 10), (300, 5, 256, 4), (300, 6, 128, 5), (1, 30, 256, 12)])
def test_rollout_bitexact(P, orc, N, T, HID, max_actions, rollout_mode):
    _storage(P, (N + T) % 2)
    env = P.HipVecEnv(num_envs=N, Q=8, max_actions=max_actions, seed=77, global_offset=5)
    pol = P.HipPolicy(72, HID, 2, 4, seed=11)
    ro = P.BufferRollouts()
    P.collect_rollouts_steps_(ro, env, pol, T, 0.99, record_probs=True)
    oenv = orc.Env(Q=8, max_actions=max_actions, N=N, seed=77, global_offset=5)
    oenv.reset()
    ref = orc.collect_rollouts_tn(oenv, pol.params, HID, T, mode_dev=True)
    st, act = ro.state_data
    assert np.array_equal(st, ref["states"]) and np.array_equal(act, ref["active"])
    assert np.array_equal(ro.selected_actions - 1, ref["actions"])
    assert np.array_equal(ro.selected_action_probabilities, ref["p_sel"])
    assert np.array_equal(ro.raw_rewards, ref["rewards"])
    assert np.array_equal(ro.terminal, ref["done"].astype(bool))
    assert np.array_equal(ro.rewards, orc.compute_returns_tn(ref["rewards"], ref["done"], 0.99))
    assert ref["done"].sum() > 0 and len(ro) == N * T
    fp = ro.full_probs()
    assert np.allclose(fp.sum(axis=2), 1.0, atol=1e-5)
    # a second call continues the same envs (fresh RNG counters): still bit-exact
    P.collect_rollouts_steps_(ro, env, pol, 8, 1.0)
    ref2 = orc.collect_rollouts_tn(oenv, pol.params, HID, 8, mode_dev=True)
    assert np.array_equal(ro.selected_actions - 1, ref2["actions"])
    P.set_rollout_compact(None)


@pytest.fixture(params=["per-step", "per-step-split", "persistent", "persistent-split"])
def rollout_mode(request, P):
    """All rollout executions: three launches per step and the whole T-step rollout in one launch (MODE 3), each with one
    wave per env and with 2 / 4 waves per env (what few envs take by default; Q = 8 fp32, one wave otherwise)."""
    P.set_rollout_persistent(request.param not in ("per-step", "per-step-split"))
    P.set_rollout_split_max_envs(None if request.param.endswith("split") else 0)
    yield request.param
    P.set_rollout_persistent(None)
    P.set_rollout_split_max_envs(None)


@pytest.fixture(params=["per-step", "persistent"])
def rollout_mode2(request, P):
    """Per-step launches vs the one-launch rollout, for shapes the 2 / 4-waves-per-env kernels do not cover (Q = 32)."""
    P.set_rollout_persistent(request.param == "persistent")
    yield request.param
    P.set_rollout_persistent(None)


def _storage(P, compact):
    """Pick a state-storage form inside a test whose parameters already multiply (undone by the caller)."""
    P.set_rollout_compact(bool(compact))


@pytest.fixture(params=[False, True], ids=["expanded", "compact"])
def storage_mode(request, P):
    """Both state-storage forms of engine-collected rollouts: the expanded observation rows, and the env snapshots the
    train forward re-derives them from (ppo_set_rollout_compact)."""
    P.set_rollout_compact(request.param)
    yield request.param
    P.set_rollout_compact(None)


@pytest.fixture(params=[0, -2, 4096, -1], ids=["large-batch-kernels", "large-batch-fp32-mfma", "small-batch-kernels", "train-tile-kernels"])
def bwd_form(request, P):
    """All kernel sets at test sizes: the fused backward (its products as split-fp32 MFMAs on the bf16 pipe: the default;
    and as pure fp32 MFMAs: ppo_set_bwd_split_bf16(0)) + one-wave-per-state train forward (forced); the three-product
    backward and the 2 / 4-waves-per-state train forward (ppo_set_bwd_small_max_tiles, ppo_set_fwd_split_max_states); and
    the one-workgroup-per-tile training pass + operand-layout weight-gradient kernel (ppo_set_train_tile_max_tiles; it
    covers Q = 8 states in the expanded storage form and falls back to the others elsewhere)."""
    if request.param == -1:
        P.set_train_tile_max_tiles(1 << 20)
    else:
        P.set_train_tile_max_tiles(0)
        P.set_bwd_small_max_tiles(max(request.param, 0))
        P.set_fwd_split_max_states(min(max(request.param, 0), 512))
        if request.param in (-2, 4096):      # the fp32-MFMA forms of the large- and the small-minibatch kernels
            P.set_bwd_split_bf16(False)
    yield request.param
    P.set_bwd_split_bf16(None)
    P.set_bwd_small_max_tiles(None)
    P.set_fwd_split_max_states(None)
    P.set_train_tile_max_tiles(None)


@pytest.mark.parametrize("case", range(10))
def test_rollout_bitexact_fuzz(P, orc, case, rollout_mode):
    """Randomised shapes / seeds / horizons / global offsets: whole rollouts stay bit-identical to the device-order
    oracle (states, masks, sampled actions, probabilities, rewards, done flags, returns in both discount types)."""
    rng = np.random.default_rng(1000 + case)
    _storage(P, case % 2)                               # odd cases keep env snapshots, even ones the observation rows
    N = int(rng.integers(1, 70))
    T = int(rng.integers(1, 40))
    HID = int(rng.choice([128, 256]))
    Q = int(rng.choice([8, 8, 32]))
    max_actions = int(rng.integers(2, 30))
    seed = int(rng.integers(0, 2 ** 31))
    off = int(rng.integers(0, 2 ** 20))
    gamma = [1.0, 0.99, np.float32(0.97)][case % 3]
    env = P.HipVecEnv(num_envs=N, Q=Q, max_actions=max_actions, seed=seed, global_offset=off)
    pol = P.HipPolicy(72, HID, 2, 4, seed=case)
    pol.params = pol.params + (rng.normal(size=pol.num_params) * 0.05).astype(np.float32)
    ro = P.BufferRollouts()
    P.collect_rollouts_steps_(ro, env, pol, T, gamma)
    oenv = orc.Env(Q=Q, max_actions=max_actions, N=N, seed=seed, global_offset=off)
    oenv.reset()
    ref = orc.collect_rollouts_tn(oenv, pol.params, HID, T, mode_dev=True)
    st, act = ro.state_data
    assert np.array_equal(st, ref["states"]) and np.array_equal(act, ref["active"])
    assert np.array_equal(ro.selected_actions - 1, ref["actions"])
    assert np.array_equal(ro.selected_action_probabilities, ref["p_sel"])
    assert np.array_equal(ro.raw_rewards, ref["rewards"]) and np.array_equal(ro.terminal, ref["done"].astype(bool))
    assert np.array_equal(ro.rewards, orc.compute_returns_tn(ref["rewards"], ref["done"], float(gamma),
                                                             isinstance(gamma, np.float32)))
    assert env.error_flags() & ~32 == 0
    P.set_rollout_compact(None)


@pytest.mark.parametrize("B", [1, 2, 3, 31, 33, 255, 257])
def test_gradient_ragged_batch_sizes(P, orc, B, bwd_form):
    """Minibatches that do not fill the persistent grid (B = 1 ... 257 tiles on 256 workgroups) and repeat samples."""
    if bwd_form == -2 and B not in (1, 257):
        pytest.skip("the fp32-MFMA form of the large-minibatch kernels: the two ends of the range only (suite time)")
    env, pol, ro, ds = _make_dataset(P, orc, 30, 10, 256, seed=77)      # 300 samples
    rng = np.random.default_rng(B)
    sel = rng.integers(1, len(ds) + 1, size=B)
    lp, le = P.forward_backward(pol, ds, sel, 0.05, 0.01)
    g64, olp, ole = _oracle_grad(orc, pol.params, 256, ro, sel - 1, 0.05, 0.01)
    assert np.abs(pol.grad() - g64).max() <= 2e-5 * np.abs(g64).max() + 1e-9
    assert abs(lp - olp) <= 1e-5 * (1 + abs(olp)) and abs(le - ole) <= 1e-5 * (1 + abs(ole))


def test_cdf_residue_goes_to_last_unmasked_action(P, orc, rollout_mode):
    """u = 1 - 2^-24 (the largest uniform) against a distribution whose sequential fp32 sum stops at or below it: the
    walk runs off the end onto a masked action (p = 0), where the reference's `@assert ap[a] > 0.0` would throw
    (src/collect_rollouts.jl:7).  The engine and the oracle give the residue to the last action with p > 0 (flag 32).
    The (seed, env id, tick) triples below were found by scanning Philox4x32-10 for w0 >> 8 == 0xFFFFFF."""
    fired = 0
    for seed, gid, t in [(1, 7173697, 1), (2, 2149053, 1), (3, 2350790, 0)]:
        assert orc.u01(orc.philox([gid, t, 0, 0], [seed, 0])[0]) == 1.0 - 2.0 ** -24
        for pseed in range(6):
            N, T = 3, t + 1
            env = P.HipVecEnv(num_envs=N, Q=8, max_actions=20, seed=seed, global_offset=gid - 1)
            pol = P.HipPolicy(72, 128, 2, 4, seed=pseed)
            ro = P.BufferRollouts()
            P.collect_rollouts_steps_(ro, env, pol, T, 1.0, record_probs=True)     # must not raise
            oenv = orc.Env(Q=8, max_actions=20, N=N, seed=seed, global_offset=gid - 1)
            oenv.reset()
            ref = orc.collect_rollouts_tn(oenv, pol.params, 128, T, mode_dev=True)
            assert np.array_equal(ro.selected_actions - 1, ref["actions"])
            assert np.array_equal(ro.selected_action_probabilities, ref["p_sel"])
            assert np.all(ro.selected_action_probabilities > 0)
            probs = ro.full_probs()[t, 1]
            if env.error_flags() & 32:
                fired += 1
                strict = P.HipVecEnv(num_envs=N, Q=8, max_actions=20, seed=seed, global_offset=gid - 1, strict_sampling=True)
                with pytest.raises(P.PPOError, match="ap\\[a\\] > 0.0"):       # reference semantics on request
                    P.collect_rollouts_steps_(P.BufferRollouts(), strict, pol, T, 1.0)
                a = int(ro.selected_actions[t, 1]) - 1
                assert a == int(np.nonzero(probs > 0)[0].max()) and oenv.err[1] & 32
                csum = np.float32(0)
                for x in probs:
                    csum = np.float32(csum + x)
                assert csum <= np.float32(1.0 - 2.0 ** -24)
            assert env.error_flags() & ~32 == 0
    assert fired >= 1, "none of the candidates had a short fp32 sum: extend the candidate list"


def test_rollout_episodes_mode_plumbing(P, orc):
    """BASELINE config 1: 1 env x 128-step rollout; whole episodes only (reference semantics)."""
    env = P.HipVecEnv(num_envs=1, Q=8, max_actions=128, seed=5)
    pol = P.HipPolicy(72, 128, 2, 4, seed=2)
    ro = P.BufferRollouts()
    P.collect_rollouts_(ro, env, pol, 2, 1.0)
    ds = P.construct_dataset(ro)
    n = len(ds)
    assert 2 <= n <= 256
    term = ro.terminal.reshape(-1)[ro.index()]
    assert term.sum() == 2 and term[-1]                    # two complete episodes, buffer ends on a terminal
    # oracle replay of the same two episodes
    oenv = orc.Env(Q=8, max_actions=128, N=1, seed=5)
    oenv.episode[0] = 1                                   # create() consumed episode 0, collect resets again
    acts, rews = [], []
    for ep in range(2):
        oenv.reset_one(0)
        while not oenv.done[0]:
            obs = oenv.observe_one(0)
            p = orc.action_probabilities(pol.params, 72, 128, obs, oenv.active[0], "dev")
            w = orc.philox([0, int(oenv.tick[0]), 0, 0], [5, 0])
            a, err = orc.categorical_sample(p, orc.u01(w[0]))
            oenv.step_one(0, a)
            acts.append(a)
            rews.append(float(oenv.reward[0]))
    assert n == len(acts)
    idx = ro.index()
    assert np.array_equal((ro.selected_actions.reshape(-1)[idx]) - 1, acts)
    sample = ds[1]
    assert sample["selected_action"] == acts[0] + 1 and sample["state"].vertex_score.shape == (32, 72)
    batch = ds[[1, 2]]
    assert batch["state"].vertex_score.shape == (2, 32, 72)
    with pytest.raises(P.PPOError):
        ds[0]
    with pytest.raises(P.PPOError):
        ds["a"]


# ---------------------------------------------------------------- K8-K12 training
def _make_dataset(P, orc, N, T, HID, seed, max_actions=12):
    env = P.HipVecEnv(num_envs=N, Q=8, max_actions=max_actions, seed=seed)
    pol = P.HipPolicy(72, HID, 2, 4, seed=seed + 1)
    rng = np.random.default_rng(seed)
    pol.params = pol.params + (rng.normal(size=pol.num_params) * 0.02).astype(np.float32)
    ro = P.BufferRollouts()
    P.collect_rollouts_steps_(ro, env, pol, T, 1.0)
    return env, pol, ro, P.construct_dataset(ro)


def _oracle_grad(orc, pol_params, HID, ro, sel0, eps, ew):
    st, act = ro.state_data
    st = st.reshape(-1, 32, 72)[sel0]
    act = act.reshape(-1)[sel0]
    a0 = (ro.selected_actions.reshape(-1)[sel0] - 1).astype(np.int32)
    return orc.step_batch_grad_f64(pol_params, 72, HID, st, act, a0, ro.selected_action_probabilities.reshape(-1)[sel0],
                                   ro.rewards.reshape(-1)[sel0], eps, ew)


@pytest.mark.parametrize("HID,B", [(128, 24), (128, 300), (256, 40), (256, 520)])
def test_gradient_vs_f64_oracle(P, orc, HID, B, storage_mode, bwd_form):
    if bwd_form == -2 and storage_mode:
        pytest.skip("the fp32-MFMA form of the large-minibatch kernels: expanded storage only (suite time)")
    N, T = 40, 16
    env, pol, ro, ds = _make_dataset(P, orc, N, T, HID, seed=B)
    rng = np.random.default_rng(B)
    sel = rng.choice(len(ds), size=B, replace=B > len(ds)) + 1
    # perturb p_old so ratios are not exactly 1 (and never exactly 1 +- eps: SURVEY Appendix A tie rule)
    lp, le = P.forward_backward(pol, ds, sel, 0.05, 0.01)
    g = pol.grad()
    g64, olp, ole = _oracle_grad(orc, pol.params, HID, ro, sel - 1, 0.05, 0.01)
    scale = np.abs(g64).max()
    assert scale > 0
    assert np.abs(g - g64).max() <= 2e-5 * scale + 1e-9, "gradient tolerance: 2e-5 of max|g| vs float64 oracle"
    assert abs(lp - olp) <= 1e-5 * (1 + abs(olp)) and abs(le - ole) <= 1e-5 * (1 + abs(ole))
    # run-to-run bitwise reproducibility (fixed-order slab reduction, no float atomics)
    P.forward_backward(pol, ds, sel, 0.05, 0.01)
    assert np.array_equal(g, pol.grad())


@pytest.mark.parametrize("fixture", ["poly-30-policy", "catmull-clark-policy"])
def test_gradient_with_reference_trained_weights(P, orc, golden_dir, fixture, bwd_form):
    """Zygote's gradient (src/train.jl:65-79) restated in float64, on the weights the REFERENCE trained
    (test/output/*.bson decoded into tests/golden/*.npz): rollout with those weights on the engine, then loss and
    gradient of a minibatch against orc_step_batch_grad_f64 -- same tolerance as for random-init weights."""
    params = np.load(os.path.join(golden_dir, fixture + ".npz"))["params"]
    env = P.HipVecEnv(num_envs=32, Q=8, max_actions=12, seed=17)
    pol = P.HipPolicy(72, 128, 2, 4, seed=0)
    pol.params = params
    ro = P.BufferRollouts()
    P.collect_rollouts_steps_(ro, env, pol, 12, 1.0)
    ds = P.construct_dataset(ro)
    sel = np.random.default_rng(2).permutation(len(ds))[:200] + 1
    for eps, ew in ((0.05, 0.01), (0.2, 0.0)):
        lp, le = P.forward_backward(pol, ds, sel, eps, ew)
        g = pol.grad()
        g64, olp, ole = _oracle_grad(orc, params, 128, ro, sel - 1, eps, ew)
        scale = np.abs(g64).max()
        assert scale > 0 and np.abs(g - g64).max() <= 2e-5 * scale + 1e-9
        assert abs(lp - olp) <= 1e-5 * (1 + abs(olp)) and abs(le - ole) <= 1e-5 * (1 + abs(ole))
    # the rollout itself (sampled actions, probabilities) is bit-exact against the device-order oracle with these weights
    oenv = orc.Env(Q=8, max_actions=12, N=32, seed=17)
    oenv.reset()
    ref = orc.collect_rollouts_tn(oenv, params, 128, 12, mode_dev=True)
    assert np.array_equal(ro.selected_actions - 1, ref["actions"]) and np.array_equal(ro.selected_action_probabilities, ref["p_sel"])


@pytest.mark.parametrize("split", [False, True], ids=["fp32-mfma", "split-fp32"])
@pytest.mark.parametrize("B", [256, 257, 384, 385, 512, 513])
def test_gradient_at_the_kernel_switch_points(P, orc, B, split):
    """Default kernel selection by minibatch size.  fp32-MFMA training pass (ppo_set_bwd_split_bf16(0)): 4 waves per state up
    to 256 states, 2 up to 512, one above; the three-product backward up to 384 tiles, the fused kernel above.  Split-fp32
    pass (the default): one workgroup per state and the fused backward at every size.  Every size around the switch points
    gives the float64 oracle's gradient (same tolerance)."""
    if split and B in (257, 384, 512):
        pytest.skip("the split pass has no switch points of its own: three of the six sizes (suite time)")
    P.set_bwd_split_bf16(split)
    env, pol, ro, ds = _make_dataset(P, orc, 48, 12, 256, seed=91)
    assert len(ds) == 576
    sel = np.random.default_rng(B).permutation(len(ds))[:B] + 1
    lp, le = P.forward_backward(pol, ds, sel, 0.05, 0.01)
    g = pol.grad()
    g64, olp, ole = _oracle_grad(orc, pol.params, 256, ro, sel - 1, 0.05, 0.01)
    assert np.abs(g - g64).max() <= 2e-5 * np.abs(g64).max() + 1e-9
    assert abs(lp - olp) <= 1e-5 * (1 + abs(olp)) and abs(le - ole) <= 1e-5 * (1 + abs(ole))
    P.forward_backward(pol, ds, sel, 0.05, 0.01)
    same = np.array_equal(g, pol.grad())
    P.set_bwd_split_bf16(None)
    assert same, "bitwise reproducible run to run"


def test_gradient_clipped_branch(P, orc):
    """Huge epsilon => never clipped; tiny epsilon with old probabilities scaled => clipped samples carry
    zero policy gradient (only the entropy term remains)."""
    env, pol, ro, ds = _make_dataset(P, orc, 16, 8, 128, seed=3)
    sel = np.arange(1, 65)
    for eps in (10.0, 1e-3):
        P.forward_backward(pol, ds, sel, eps, 0.02)
        g64, _, _ = _oracle_grad(orc, pol.params, 128, ro, sel - 1, eps, 0.02)
        assert np.abs(pol.grad() - g64).max() <= 2e-5 * np.abs(g64).max() + 1e-9


def test_normalised_advantage_mode(P, orc):
    """batch_advantage plugin, "returns_normalised": (R - mean) / (std + 1e-8) over the minibatch (population std,
    fp64 statistics, fp32 result).  The reference declares the plugin and ships no implementation
    (src/ProximalPolicyOptimization.jl:29), so the oracle is the f64 gradient fed with advantages normalised on the host."""
    env, pol, ro, ds = _make_dataset(P, orc, 24, 12, 128, seed=5)
    rng = np.random.default_rng(1)
    for B in (7, 64, 200):
        sel = rng.permutation(len(ds))[:B] + 1
        lp, le = P.forward_backward(pol, ds, sel, 0.05, 0.01, advantage="returns_normalised")
        g = pol.grad()
        st, act = ro.state_data
        s0 = sel - 1
        R = ro.rewards.reshape(-1)[s0].astype(np.float64)
        advn = ((R - R.mean()) / (R.std() + 1e-8)).astype(np.float32)
        g64, olp, ole = orc.step_batch_grad_f64(pol.params, 72, 128, st.reshape(-1, 32, 72)[s0], act.reshape(-1)[s0],
                                                (ro.selected_actions.reshape(-1)[s0] - 1).astype(np.int32),
                                                ro.selected_action_probabilities.reshape(-1)[s0], advn, 0.05, 0.01)
        assert np.abs(g - g64).max() <= 2e-5 * np.abs(g64).max() + 1e-9
        assert abs(lp - olp) <= 1e-5 * (1 + abs(olp)) and abs(le - ole) <= 1e-5 * (1 + abs(ole))
    with pytest.raises(P.PPOError):
        P.forward_backward(pol, ds, sel, 0.05, 0.01, advantage="td0")
    # a whole ppo_train! epoch in this mode runs and changes the parameters
    before = pol.params.copy()
    P.ppo_train_(pol, P.Optimiser(P.Adam(1e-3)), ds, 0.05, 64, 1, 0.01, seed=3, verbose=False, advantage="returns_normalised")
    assert np.all(np.isfinite(pol.params)) and not np.array_equal(before, pol.params)


def test_step_batch_adam_bitexact(P, orc):
    """Flux.update! with legacy Adam: parameters after the step are bit-identical to the oracle's Adam
    applied to the device gradient (element arithmetic in fp64, fp32 stores)."""
    env, pol, ro, ds = _make_dataset(P, orc, 16, 8, 128, seed=9)
    opt = P.Optimiser(P.Adam(1e-4))
    p = pol.params.copy()
    m = np.zeros_like(p)
    v = np.zeros_like(p)
    bp = np.array([0.9, 0.999])
    rng = np.random.default_rng(0)
    for it in range(3):
        sel = rng.permutation(len(ds))[:32] + 1
        P.forward_backward(pol, ds, sel, 0.05, 0.01)
        g = pol.grad()
        lp, le = P.step_batch_(pol, opt, ds, sel, 0.05, 0.01)
        orc.adam_step(p, g, m, v, bp, 1e-4)
        assert np.array_equal(pol.params, p)
        dm, dv, dbp = opt.members[0].get_state()
        assert np.array_equal(dm, m) and np.array_equal(dv, v) and np.allclose(dbp, bp, rtol=0, atol=0)
    assert P.get_optimizer_learning_rate(opt) == 1e-4
    with pytest.raises(TypeError):
        P.get_optimizer_learning_rate(P.Adam(1e-4))          # a bare Adam is not iterable (src/train.jl:155-158)


def test_ppo_train_epochs_with_explicit_perm(P, orc):
    """ppo_train! (src/train.jl:86-153) with explicitly supplied permutations (stand-in for randperm):
    per-epoch mean losses and final parameters vs the oracle loop (f64 grad -> f32 -> oracle Adam)."""
    env, pol, ro, ds = _make_dataset(P, orc, 12, 10, 128, seed=21)
    n = len(ds)
    E, Bsz = 2, 50                                            # last batch of each epoch is short (120 = 50+50+20)
    rng = np.random.default_rng(4)
    perm = np.stack([rng.permutation(n) + 1 for _ in range(E)])
    p = pol.params.copy()
    m, v, bp = np.zeros_like(p), np.zeros_like(p), np.array([0.9, 0.999])
    want_p, want_e = [], []
    for e in range(E):
        lps, les = [], []
        for s in range(0, n, Bsz):
            sel0 = perm[e, s:s + Bsz] - 1
            g64, lp, le = _oracle_grad(orc, p, 128, ro, sel0, 0.05, 0.01)
            orc.adam_step(p, g64.astype(np.float32), m, v, bp, 1e-4)
            lps.append(lp)
            les.append(le)
        want_p.append(np.mean(lps))
        want_e.append(np.mean(les))
    opt = P.Optimiser(P.Adam(1e-4))
    ph, eh, lh = P.ppo_train_(pol, opt, ds, 0.05, Bsz, E, 0.01, perm=perm, verbose=False)
    assert np.allclose(ph, want_p, rtol=1e-4, atol=1e-6) and np.allclose(eh, want_e, rtol=1e-4, atol=1e-7)
    assert lh == [1e-4] * E
    # Adam's first steps are sign-like (|delta| ~ eta): near-zero gradient entries can flip, so compare
    # with an absolute tolerance of a few eta-steps on a handful of entries and tightly elsewhere
    diff = np.abs(pol.params - p)
    assert np.quantile(diff, 0.999) <= 2e-6 and diff.max() <= 6 * 1e-4 * 2
    with pytest.raises(P.PPOError):
        P.ppo_train_(pol, opt, ds, 0.05, n + 1, 1, 0.01, verbose=False)     # @assert 1 <= batch_size <= num_data


def test_feistel_minibatch_order_is_a_permutation(P, orc):
    env, pol, ro, ds = _make_dataset(P, orc, 10, 10, 128, seed=5)
    opt = P.Optimiser(P.Adam(1e-4))
    before = pol.params.copy()
    ph, eh, lh = P.ppo_train_(pol, opt, ds, 0.05, 32, 1, 0.01, seed=77, verbose=False)
    assert np.isfinite(ph).all() and np.isfinite(eh).all()
    assert not np.array_equal(before, pol.params)


def test_learning_signal(P):
    """The engine actually learns: mean return per step rises over PPO iterations on the synthetic env."""
    env = P.HipVecEnv(num_envs=256, Q=8, max_actions=32, seed=3)
    pol = P.HipPolicy(72, 128, 2, 4, seed=0)
    opt = P.Optimiser(P.Adam(3e-4))
    means = []
    for it in range(12):
        ro = P.BufferRollouts()
        P.collect_rollouts_steps_(ro, env, pol, 32, 1.0)
        means.append(float(ro.raw_rewards.mean()))
        ds = P.construct_dataset(ro)
        P.ppo_train_(pol, opt, ds, 0.1, 1024, 2, 0.01, seed=it, verbose=False)
    assert np.mean(means[-3:]) > np.mean(means[:3]) + 0.05, means


# ---------------------------------------------------------------- BASELINE-size properties (config 2)
def test_full_size_properties(P, orc):
    """4096 envs x 128 steps, 2x256 MLP: size-independent properties + teacher-forced oracle spot checks."""
    N, T = 4096, 128
    env = P.HipVecEnv(num_envs=N, Q=8, max_actions=128, seed=1234)
    pol = P.HipPolicy(72, 256, 2, 4, seed=0)
    ro = P.BufferRollouts()
    P.collect_rollouts_steps_(ro, env, pol, T, 1.0)
    assert len(ro) == N * T
    a = ro.selected_actions
    p = ro.selected_action_probabilities
    st, act = ro.state_data
    assert a.min() >= 1 and a.max() <= 128 and np.all(p > 0) and np.all(p <= 1)
    quad = (a - 1) // 16
    assert np.all((act >> quad.astype(np.uint32)) & 1), "sampled action on an inactive quad"
    assert np.array_equal(ro.rewards, orc.compute_returns_tn(ro.raw_rewards, ro.terminal, 1.0))
    rng = np.random.default_rng(0)
    params = pol.params
    for _ in range(40):                     # teacher-forced: recorded state -> oracle probs -> same sample
        t, n = int(rng.integers(0, T)), int(rng.integers(0, N))
        pr = orc.action_probabilities(params, 72, 256, st[t, n], act[t, n], "dev")
        assert p[t, n] == pr[a[t, n] - 1]
    ds = P.construct_dataset(ro)
    opt = P.Optimiser(P.Adam(1e-4))
    ph, eh, _ = P.ppo_train_(pol, opt, ds, 0.05, 4096, 1, 0.01, seed=1, verbose=False)
    assert np.isfinite(ph[0]) and np.isfinite(eh[0])
    assert np.isfinite(pol.params).all()


# ---------------------------------------------------------------- BASELINE config 4 shape: Q=32 -> H=128, A=512
def test_policy_forward_q32(P, orc):
    rng = np.random.default_rng(128)
    for HID in (128, 256):
        pol = P.HipPolicy(72, HID, 2, 4, seed=5)
        pol.params = pol.params + (rng.normal(size=pol.num_params) * 0.02).astype(np.float32)
        B = 20
        states = rng.integers(-3, 7, size=(B, 128, 72)).astype(np.int8)
        active = rng.integers(1, 2**32, size=B, dtype=np.uint64).astype(np.uint32)
        probs = P.batch_action_probabilities(pol, P.StateData(states, active)).T
        assert probs.shape == (B, 512)
        for b in range(B):
            dev = orc.action_probabilities(pol.params, 72, HID, states[b], active[b], "dev")
            assert np.array_equal(probs[b], dev)
            ref = orc.action_probabilities(pol.params, 72, HID, states[b], active[b], "ref")
            assert np.allclose(probs[b], ref, rtol=2e-5, atol=1e-8)
            q = np.arange(512) // 16
            assert np.all(probs[b][((int(active[b]) >> q) & 1) == 0] == 0.0)


@pytest.mark.parametrize("HID", [128, 256])
def test_rollout_and_gradient_q32(P, orc, HID, rollout_mode2, storage_mode):
    """square_mesh-sized action space (Q=32 quads, 512 masked actions), variable-length episodes."""
    N, T, max_actions = 12, 20, 9
    env = P.HipVecEnv(num_envs=N, Q=32, max_actions=max_actions, seed=31)
    pol = P.HipPolicy(72, HID, 2, 4, seed=8)
    ro = P.BufferRollouts()
    P.collect_rollouts_steps_(ro, env, pol, T, 1.0)
    oenv = orc.Env(Q=32, max_actions=max_actions, N=N, seed=31)
    oenv.reset()
    ref = orc.collect_rollouts_tn(oenv, pol.params, HID, T, mode_dev=True)
    st, act = ro.state_data
    assert st.shape == (T, N, 128, 72)
    assert np.array_equal(st, ref["states"]) and np.array_equal(act, ref["active"])
    assert np.array_equal(ro.selected_actions - 1, ref["actions"])
    assert np.array_equal(ro.selected_action_probabilities, ref["p_sel"])
    assert np.array_equal(ro.raw_rewards, ref["rewards"]) and np.array_equal(ro.terminal, ref["done"].astype(bool))
    assert ref["done"].sum() >= N                          # episodes end early / at max_actions: variable length
    ds = P.construct_dataset(ro)
    sel = np.random.default_rng(1).choice(len(ds), size=30, replace=False) + 1
    lp, le = P.forward_backward(pol, ds, sel, 0.05, 0.01)
    a0 = (ro.selected_actions.reshape(-1)[sel - 1] - 1).astype(np.int32)
    g64, olp, ole = orc.step_batch_grad_f64(pol.params, 72, HID, st.reshape(-1, 128, 72)[sel - 1], act.reshape(-1)[sel - 1],
                                            a0, ro.selected_action_probabilities.reshape(-1)[sel - 1],
                                            ro.rewards.reshape(-1)[sel - 1], 0.05, 0.01)
    assert np.abs(pol.grad() - g64).max() <= 2e-5 * np.abs(g64).max() + 1e-9
    assert abs(lp - olp) <= 1e-5 * (1 + abs(olp)) and abs(le - ole) <= 1e-5 * (1 + abs(ole))
    P.set_bwd_small_max_tiles(0)                           # the same minibatch through the fused backward (120 tiles)
    try:
        P.forward_backward(pol, ds, sel, 0.05, 0.01)
    finally:
        P.set_bwd_small_max_tiles(None)
    assert np.abs(pol.grad() - g64).max() <= 2e-5 * np.abs(g64).max() + 1e-9
    opt = P.Optimiser(P.Adam(1e-4))
    ph, eh, _ = P.ppo_train_(pol, opt, ds, 0.05, 64, 1, 0.01, seed=3, verbose=False)
    assert np.isfinite(ph[0]) and np.isfinite(eh[0])


def test_config4_size_properties(P, orc, rollout_mode2):
    """BASELINE config 4 size: 8192 envs, Q=32 (A=512), masked actions, variable-length episodes; both rollout
    executions (per-step launches and the one-launch persistent rollout)."""
    N, T = 8192, 16
    env = P.HipVecEnv(num_envs=N, Q=32, max_actions=12, seed=4)
    pol = P.HipPolicy(72, 128, 2, 4, seed=0)
    ro = P.BufferRollouts()
    P.collect_rollouts_steps_(ro, env, pol, T, 0.99)
    a, p = ro.selected_actions, ro.selected_action_probabilities
    st, act = ro.state_data
    assert a.min() >= 1 and a.max() <= 512 and np.all(p > 0)
    assert np.all((act >> ((a - 1) // 16).astype(np.uint32)) & 1)
    assert np.array_equal(ro.rewards, orc.compute_returns_tn(ro.raw_rewards, ro.terminal, 0.99))
    assert ro.terminal.sum() >= N
    rng = np.random.default_rng(0)
    for _ in range(10):
        t, n = int(rng.integers(0, T)), int(rng.integers(0, N))
        pr = orc.action_probabilities(pol.params, 72, 128, st[t, n], act[t, n], "dev")
        assert p[t, n] == pr[a[t, n] - 1]


# ---------------------------------------------------------------- data-parallel plumbing on one GPU
@pytest.mark.parametrize("kind", ["torch.distributed/nccl", "native-rccl"])
def test_allreduce_hook_single_rank_rccl(P, orc, kind, monkeypatch):
    """One-rank RCCL group, both hook kinds DataParallel.make_hook can hand to ppo_train: the library's own RCCL
    all-reduce (default) and a torch.distributed all-reduce on a tensor aliasing the gradient buffer (PPO_NATIVE_RCCL=0).
    With one rank the sum is the identity, so training must be bit-identical to the hook-free run (validates
    aliasing, stream ordering, the C callback and the seed-only minibatch order)."""
    import socket
    monkeypatch.setenv("PPO_NATIVE_RCCL", "1" if kind == "native-rccl" else "0")
    import torch
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        res = []
        for hook in (False, True):
            env = P.HipVecEnv(num_envs=32, Q=8, max_actions=10, seed=2)
            pol = P.HipPolicy(72, 128, 2, 4, seed=4)
            ro = P.BufferRollouts()
            P.collect_rollouts_steps_(ro, env, pol, 8, 1.0)
            ds = P.construct_dataset(ro)
            opt = P.Optimiser(P.Adam(1e-3))
            dp = P.DataParallel(0, 1, force_hook=hook)
            ph, eh, _ = P.ppo_train_(pol, opt, ds, 0.05, 64, 2, 0.01, seed=11, parallel=dp, verbose=False)
            torch.cuda.synchronize()
            res.append((pol.params.copy(), ph, eh))
            assert dp.hook_kind == (kind if hook else None)
        assert np.array_equal(res[0][0], res[1][0]) and res[0][1] == res[1][1] and res[0][2] == res[1][2]
    finally:
        P.rccl_finalize()
        dist.destroy_process_group()


def test_collect_rollouts_exact_num_episodes(P):
    """src/rollout_buffer.jl:73-77: exactly num_episodes episodes enter the buffer, also when the env batch does not
    divide num_episodes (episode e runs on env e mod N; envs beyond num_episodes stay idle)."""
    for N, num_episodes in ((6, 8), (6, 3), (4, 9), (5, 5)):
        env = P.HipVecEnv(num_envs=N, Q=8, max_actions=6, seed=21)
        pol = P.HipPolicy(72, 128, 2, 4, seed=4)
        ro = P.BufferRollouts()
        P.collect_rollouts_(ro, env, pol, num_episodes, 1.0)
        valid, term = ro.valid, ro.terminal
        assert int((valid & term).sum()) == num_episodes
        per_env = (valid & term).sum(axis=0)
        assert list(per_env) == [len(range(n, num_episodes, N)) for n in range(N)]
        idx = ro.index()
        assert len(idx) == len(ro) == int(valid.sum()) and term.reshape(-1)[idx][-1]
        # whole episodes only: in every env column the valid flags form a prefix that ends on a terminal
        for n in range(N):
            k = int(valid[:, n].sum())
            assert valid[:k, n].all() and not valid[k:, n].any() and (k == 0 or term[k - 1, n])


def test_average_returns_evaluator(P, orc):
    """src/evaluate.jl:18-25 -- mean / sample-std of undiscounted episode returns, checked against an oracle replay.
    11 trajectories on 6 envs: exactly 11 are played (envs 0..4 two each, env 5 one)."""
    N, num_traj = 6, 11
    env = P.HipVecEnv(num_envs=N, Q=8, max_actions=7, seed=13)
    pol = P.HipPolicy(72, 128, 2, 4, seed=3)
    mean, std = P.average_returns(pol, env, num_traj)
    oenv = orc.Env(Q=8, max_actions=7, N=N, seed=13)
    oenv.episode[:] = 1                                  # create() consumed episode 0 on the device side
    rets = []
    for n in range(N):
        for ep in range(len(range(n, num_traj, N))):
            oenv.reset_one(n)
            acc = 0.0
            while not oenv.done[n]:
                p = orc.action_probabilities(pol.params, 72, 128, oenv.observe_one(n), oenv.active[n], "dev")
                w = orc.philox([n, int(oenv.tick[n]), 0, 0], [13, 0])
                a, err = orc.categorical_sample(p, orc.u01(w[0]))
                oenv.step_one(n, a)
                acc += float(oenv.reward[n])
            rets.append(acc)
    assert abs(mean - np.mean(rets)) < 1e-9 and abs(std - np.std(rets, ddof=1)) < 1e-9


def _oracle_scores(oenv, n):
    """env.current_score / env.opt_score of the synthetic env: sum |vertex score| over the active quads, |sum|."""
    act = int(oenv.active[n])
    sc = oenv.score[n].astype(np.int64)
    on = np.array([(act >> (v >> 2)) & 1 for v in range(sc.size)], bool)
    return int(np.abs(sc[on]).sum()), int(abs(sc[on].sum()))


def _oracle_trajectories(orc, pol, N, num_traj, seed, max_actions, global_offset, kind):
    """The reference's evaluator loops (test/quad_game_utilities.jl:280-307,369-387, src/evaluate.jl:1-25) replayed on
    the oracle env, trajectory e on env e mod N, env-major result order."""
    oenv = orc.Env(Q=8, max_actions=max_actions, N=N, seed=seed, global_offset=global_offset)
    oenv.episode[:] = 1                                  # create() consumed episode 0 on the device side
    out = []
    for n in range(N):
        for _ in range(len(range(n, num_traj, N))):
            oenv.reset_one(n)                            # PPO.reset!(wrapper)
            cur, opt = _oracle_scores(oenv, n)
            init, minscore, maxret, ret = cur, cur, cur - opt, 0.0
            if kind == "normalized" and maxret == 0:
                out.append(1.0)                          # :372-373, the trajectory is not played
                continue
            while not oenv.done[n]:
                p = orc.action_probabilities(pol.params, 72, 128, oenv.observe_one(n), oenv.active[n], "dev")
                w = orc.philox([global_offset + n, int(oenv.tick[n]), 0, 0], [seed, 0])
                a, err = orc.categorical_sample(p, orc.u01(w[0]))
                oenv.step_one(n, a)
                ret += float(oenv.reward[n])
                minscore = min(minscore, _oracle_scores(oenv, n)[0])
            best = init - minscore
            out.append(ret if kind == "return" else (float(best) if kind == "best" else best / maxret))
    return np.array(out, np.float64)


@pytest.mark.parametrize("kind", ["return", "best", "normalized"])
def test_evaluator_variants(P, orc, kind):
    """best_single_trajectory_return / average_best_returns (test/quad_game_utilities.jl:280-307) and
    single_trajectory_normalized_return / average_normalized_returns (:369-387) on the device, per trajectory and as
    (mean, sample std), against the reference's loops replayed on the oracle.  Global env id 107433 with seed 13 resets
    (episode counter 1) onto its optimum: the `maxreturn == 0 -> 1.0` branch, which does not play the trajectory."""
    N, num_traj, seed, goff, T = 3, 8, 13, 107433, 9
    pol = P.HipPolicy(72, 128, 2, 4, seed=5)
    want = _oracle_trajectories(orc, pol, N, num_traj, seed, T, goff, kind)
    env = P.HipVecEnv(num_envs=N, Q=8, max_actions=T, seed=seed, global_offset=goff)
    got = P.evaluate_trajectories(env, pol, num_traj, kind)
    assert np.array_equal(got, want), (got, want)
    if kind == "normalized":
        assert want[0] == 1.0                            # env 0's first trajectory starts at its optimum
    env = P.HipVecEnv(num_envs=N, Q=8, max_actions=T, seed=seed, global_offset=goff)
    if kind == "return":
        mean, std = P.average_returns(pol, env, num_traj)
    elif kind == "best":
        mean, std = P.average_best_returns(env, pol, num_traj)
    else:
        mean, std = P.average_normalized_returns(env, pol, num_traj)
    assert abs(mean - want.mean()) < 1e-9 and abs(std - want.std(ddof=1)) < 1e-9


# ---------------------------------------------------------------- disk rollout store (config 5 / src/rollouts_to_disk.jl)
def test_write_returns_to_disk_reproduces_reference_csv(P, golden_dir, tmp_path):
    """test/write_action_history.jl + output/trajectory.csv: six update! calls then write_returns_to_disk(.,1.0)
    must give the reference's own trajectory.csv byte for byte."""
    d = P.DiskRollouts(str(tmp_path / "output"))
    for k in range(6):
        P.update_(d, np.array([1, 2, 3, 4, 5], np.int64), 0.5, 4, 1, k == 5)
    P.write_returns_to_disk(d, 1.0)
    want = open(os.path.join(golden_dir, "trajectory.csv")).read().replace("\r\n", "\n")
    assert open(d.trajectory_filename).read() == want
    assert open(os.path.join(d.state_data_directory, "states", "sample_6.bson"), "rb").read() == \
        open(os.path.join(golden_dir, "sample_1.bson"), "rb").read()
    ds = P.DiskDataset(d.state_data_directory)
    assert len(ds) == 6 and ds[1]["returns"] == 6.0 and ds[6]["returns"] == 1.0


@pytest.mark.parametrize("stream_form", ["compact", "expanded"])
def test_streamed_disk_rollouts_roundtrip(P, orc, tmp_path, stream_form):
    """Steps are streamed device -> pinned host -> rollout.bin while collection runs; the shard read back through
    ppo_rollouts_load_disk must reproduce every column, and training from it must match training from memory.  Default
    while streaming: the record carries the 64-byte env snapshot instead of the 2304-byte observation (file version 2);
    ppo_set_rollout_compact(0) keeps the observation rows (version 1)."""
    if stream_form == "expanded":
        P.set_rollout_compact(False)
    try:
        _streamed_roundtrip(P, tmp_path, 32 * 72 if stream_form == "expanded" else 64)
    finally:
        P.set_rollout_compact(None)


def _streamed_roundtrip(P, tmp_path, state_bytes):
    N, T = 48, 12
    res = {}
    for mode in ("memory", "disk"):
        env = P.HipVecEnv(num_envs=N, Q=8, max_actions=9, seed=17)
        pol = P.HipPolicy(72, 128, 2, 4, seed=6)
        if mode == "memory":
            ro = P.BufferRollouts()
            P.collect_rollouts_steps_(ro, env, pol, T, 0.99)
        else:
            disk = P.DiskRollouts(str(tmp_path / "store"))
            P.collect_rollouts_steps_(disk, env, pol, T, 0.99, pinned_slots=2)      # 2 slots: exercises back-pressure
            assert os.path.getsize(os.path.join(disk.state_data_directory, "rollout.bin")) == \
                40 + (16 if state_bytes == 64 else 0) + T * (N * state_bytes + N * 17) + T * N * 4   # header (+ template tag of the snapshot form), T step records, returns column
            assert len(disk) == N * T
            ro = P.load_disk_rollouts(disk.state_data_directory, env)               # DiskDataset path
        st, act = ro.state_data
        ds = P.construct_dataset(ro)
        opt = P.Optimiser(P.Adam(1e-3))
        perm = np.stack([np.random.default_rng(3).permutation(len(ds)) + 1])
        ph, eh, _ = P.ppo_train_(pol, opt, ds, 0.05, 128, 1, 0.01, perm=perm, verbose=False)
        res[mode] = (st, act, ro.selected_actions, ro.selected_action_probabilities, ro.rewards, ro.raw_rewards,
                     ro.terminal, pol.params, ph, eh)
    for a, b in zip(res["memory"], res["disk"]):
        assert np.array_equal(np.asarray(a), np.asarray(b))


def test_streamed_disk_rollouts_full_width(P, tmp_path):
    """BASELINE config 5 width: 65536 envs streamed to disk (T small), bf16 policy.  The stream carries env snapshots
    (81 B per env-step); the shard read back is the resident rollout column for column, and the observations re-derived
    from it match the expanded form of a second, identical collection."""
    N, T = 65536, 4
    cols = {}
    for mode in ("disk", "memory-expanded"):
        env = P.HipVecEnv(num_envs=N, Q=8, max_actions=128, seed=3)
        pol = P.HipPolicy(72, 256, 2, 4, seed=0, dtype="bf16")
        if mode == "disk":
            disk = P.DiskRollouts(str(tmp_path / "wide"))
            P.collect_rollouts_steps_(disk, env, pol, T, 1.0)
            sz = os.path.getsize(os.path.join(disk.state_data_directory, "rollout.bin"))
            assert sz == 40 + 16 + T * N * (64 + 17) + T * N * 4
            ro = P.load_disk_rollouts(disk.state_data_directory, env)
        else:
            P.set_rollout_compact(False)
            try:
                ro = P.BufferRollouts()
                P.collect_rollouts_steps_(ro, env, pol, T, 1.0)
            finally:
                P.set_rollout_compact(None)
        st, act = ro.state_data
        cols[mode] = (st, act, ro.selected_actions, ro.selected_action_probabilities, ro.rewards, ro.terminal)
        if mode == "disk":          # train straight from the reloaded snapshots (MODE 4 forward), one minibatch of 65536
            h = P.ppo_train_(pol, P.Optimiser(P.Adam(1e-4)), P.construct_dataset(ro), 0.05, 65536, 1, 0.01, seed=0, verbose=False)
            assert np.isfinite(h[0][0]) and np.isfinite(h[1][0])
    for a, b in zip(cols["disk"], cols["memory-expanded"]):
        assert np.array_equal(a, b)


def test_ppo_iterate_disk_method(P, tmp_path):
    """ppo_iterate! 12-argument method (src/train.jl:164-202): rollouts through DiskRollouts, folder cleared."""
    env = P.HipVecEnv(num_envs=4, Q=8, max_actions=6, seed=1)
    pol = P.HipPolicy(72, 128, 2, 4, seed=0)
    opt = P.Optimiser(P.Adam(1e-4))

    class Evaluator:                                   # the reference's evaluator objects are callable structs
        def __init__(self):
            self.calls, self.saved = 0, None

        def __call__(self, policy, env_, optimizer):
            self.calls += 1

    # save_loss is a plugin with no method anywhere in the reference: an evaluator type without one throws (:196,247)
    with pytest.raises(P.PPOError, match="Function save_loss needs to be overloaded"):
        P.ppo_iterate_(pol, env, opt, 8, 8, 1, Evaluator(), 1, 1.0, 0.05, 0.01, verbose=False)

    @P.save_loss.register(Evaluator)
    def _(ev, loss):
        ev.saved = {k: list(v) for k, v in loss.items()}

    ev = Evaluator()
    path = str(tmp_path / "iter_store")
    loss = P.ppo_iterate_(pol, env, opt, 8, 8, 2, ev, 1, 1.0, 0.05, 0.01, path, verbose=False)
    assert ev.calls == 2 and len(loss["ppo"]) == 2 and not os.path.exists(path) and ev.saved == loss
    loss2 = P.ppo_iterate_(pol, env, opt, 8, 8, 1, Evaluator(), 2, 1.0, 0.05, 0.01, verbose=False)
    assert len(loss2["entropy"]) == 2 and loss2["lr"] == [1e-4, 1e-4]


def test_example_driver_script_learns_and_checkpoints(P, tmp_path, monkeypatch):
    """examples/train_rand_poly.py: the reference's driver-script shape (evaluator object + save_loss plugin +
    ppo_iterate! + BSON checkpoint of the best policy) runs end to end and improves the average return."""
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("train_rand_poly", os.path.join(root, "examples", "train_rand_poly.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = str(tmp_path / "best.bson")
    monkeypatch.setattr(sys, "argv", ["train_rand_poly.py", "--iterations", "6", "--envs", "256", "--out", out])
    ev = mod.main()
    assert os.path.exists(out) and ev.loss is not None and len(ev.loss["ppo"]) == 6 * 4
    assert ev.best_return > ev.mean_returns[0] + 1.0, ev.mean_returns
