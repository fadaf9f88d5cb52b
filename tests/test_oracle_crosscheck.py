"""Two independent CPU restatements (C and numpy/torch) must agree before anything is golden."""
import os

import numpy as np
import pytest

from oracle import np_oracle


@pytest.fixture(scope="module")
def small(orc):
    rng = np.random.default_rng(7)
    F, HID, H, B = 72, 128, 32, 6
    params = orc.glorot_params(F, HID, 2, seed=3)
    params += (rng.normal(size=params.size) * 0.01).astype(np.float32)     # non-zero biases
    states = rng.integers(-3, 6, size=(B, H, F)).astype(np.int8)
    active = np.array([0x3F, 0xFF, 0x0F, 0x3F, 0x7F, 0x1F], np.uint32)
    return dict(F=F, HID=HID, H=H, B=B, params=params, states=states, active=active, rng=rng)


def test_returns_random(orc):
    rng = np.random.default_rng(0)
    for gamma in (1.0, 0.99, 0.5):
        r = rng.normal(size=500).astype(np.float32)
        t = (rng.random(500) < 0.05).astype(np.uint8)
        assert np.array_equal(orc.compute_returns(r, t, gamma), np_oracle.compute_returns(r, t, gamma))
        assert np.array_equal(orc.compute_returns(r, t, gamma, True), np_oracle.compute_returns_f32(r, t, gamma))
    # f64-running vs f32-running semantics differ in general (SURVEY hard part 3) but stay within 1e-5 rel
    r = rng.normal(size=128).astype(np.float32)
    t = np.zeros(128, np.uint8)
    a, b = orc.compute_returns(r, t, 0.99), orc.compute_returns(r, t, 0.99, True)
    assert np.allclose(a, b, rtol=1e-5, atol=1e-5)


def test_gae_reduces_to_returns(orc):
    rng = np.random.default_rng(1)
    T, N = 40, 7
    r = rng.normal(size=(T, N)).astype(np.float32)
    d = (rng.random((T, N)) < 0.1).astype(np.uint8)
    adv, ret = orc.gae_tn(r, d, np.zeros((T + 1, N), np.float32), 0.97, 1.0)
    assert np.array_equal(adv, orc.compute_returns_tn(r, d, 0.97))
    assert np.array_equal(ret, adv)


def test_mlp_forward_modes(orc, small):
    s = small
    for b in range(s["B"]):
        x = s["states"][b]
        f64 = orc.mlp_logits(s["params"], s["F"], s["HID"], x, "f64")
        npv = np_oracle.mlp_logits(s["params"], s["F"], s["HID"], x)
        assert np.allclose(f64, npv, rtol=1e-12, atol=1e-12)
        ref = orc.mlp_logits(s["params"], s["F"], s["HID"], x, "ref")
        dev = orc.mlp_logits(s["params"], s["F"], s["HID"], x, "dev")
        scale = np.abs(f64).max() + 1.0
        assert np.abs(ref - f64).max() / scale < 2e-6
        assert np.abs(dev - f64).max() / scale < 2e-6


def test_exp_dev_accuracy(orc):
    xs = np.concatenate([-np.logspace(-6, np.log10(86.9), 400), [0.0, -87.0, -87.5, -100.0]])
    for x in xs:
        got = orc.exp_dev(np.float32(x))
        want = float(np.exp(np.float64(np.float32(x))))
        if x < -87.0:
            assert got == 0.0
        else:
            assert abs(got - want) <= 2.5e-7 * want


def test_softmax_modes(orc, small):
    s = small
    for b in range(s["B"]):
        logits = orc.mlp_logits(s["params"], s["F"], s["HID"], s["states"][b], "ref")
        mask = orc.action_mask([(int(s["active"][b]) >> q) & 1 for q in range(8)])
        want = np_oracle.masked_softmax(logits, mask)
        for mode in ("ref", "dev"):
            got = orc.masked_softmax(logits, s["active"][b], mode)
            assert np.allclose(got, want, rtol=2e-6, atol=1e-9)
            assert np.all(got[np.isneginf(mask)] == 0.0)


def test_grad_c_vs_torch(orc, small):
    s = small
    rng = s["rng"]
    B, H = s["B"], s["H"]
    actions0 = np.array([rng.integers(0, 16 * bin(int(a)).count("1")) for a in s["active"]], np.int32)
    p_old = rng.uniform(0.005, 0.02, B).astype(np.float32)
    adv = rng.normal(size=B).astype(np.float32) * 3
    masks = np.stack([orc.action_mask([(int(a) >> q) & 1 for q in range(8)]) for a in s["active"]])
    for eps, ew in ((0.05, 0.01), (0.2, 0.0), (10.0, 0.1)):
        g_c, lp_c, le_c = orc.step_batch_grad_f64(s["params"], s["F"], s["HID"], s["states"], s["active"], actions0,
                                                  p_old, adv, eps, ew)
        g_t, lp_t, le_t = np_oracle.step_batch_grad_torch(s["params"], s["F"], s["HID"], s["states"], masks, actions0,
                                                          p_old, adv, eps, ew)
        assert abs(lp_c - lp_t) < 1e-10 * (1 + abs(lp_t))
        assert abs(le_c - le_t) < 1e-10 * (1 + abs(le_t))
        assert np.abs(g_c - g_t).max() < 1e-9 * (1 + np.abs(g_t).max())
        assert np.abs(g_t).max() > 0


def test_loss_forward_f32_vs_f64(orc, small):
    s = small
    rng = np.random.default_rng(5)
    B, A = s["B"], 128
    probs = np.stack([orc.action_probabilities(s["params"], s["F"], s["HID"], s["states"][b], s["active"][b])
                      for b in range(B)])
    a1 = np.array([1, 17, 3, 40, 90, 2], np.int64)
    lin = orc.linear_action_index(a1, A)
    assert np.array_equal(lin, a1 + np.arange(B) * A)          # src/train.jl:48-52
    p_old = probs.reshape(-1)[lin - 1] * rng.uniform(0.8, 1.25, B).astype(np.float32)
    adv = rng.normal(size=B).astype(np.float32)
    lp, le = orc.ppo_loss_with_entropy(probs, lin, p_old, adv, 0.05)
    _, lp64, le64 = orc.step_batch_grad_f64(s["params"], s["F"], s["HID"], s["states"], s["active"],
                                            (a1 - 1).astype(np.int32), p_old, adv, 0.05, 1.0)
    assert abs(lp - lp64) < 1e-5 * (1 + abs(lp64))
    assert abs(le - le64) < 1e-5 * (1 + abs(le64))


def test_adam_c_vs_numpy(orc):
    rng = np.random.default_rng(3)
    n = 1000
    p = rng.normal(size=n).astype(np.float32)
    m = np.zeros(n, np.float32)
    v = np.zeros(n, np.float32)
    bp = np.array([0.9, 0.999])
    p2, m2, v2, bp2 = p.copy(), m.copy(), v.copy(), bp.copy()
    for _ in range(5):
        g = rng.normal(size=n).astype(np.float32)
        orc.adam_step(p, g, m, v, bp, 1e-4)
        p2, m2, v2, bp2 = np_oracle.adam_step(p2, g, m2, v2, bp2, 1e-4)
        assert np.array_equal(p, p2) and np.array_equal(m, m2) and np.array_equal(v, v2)
        assert np.allclose(bp, bp2, rtol=0, atol=0)
    # first step magnitude ~ eta (bias-corrected)
    assert np.all(np.abs(p - p2) == 0)


def test_feistel_is_permutation(orc):
    for n in (1, 2, 7, 100, 1000, 4096):
        for epoch in (0, 1):
            p = orc.feistel_perm(n, 1234, epoch)
            assert np.array_equal(np.sort(p), np.arange(n))
    assert not np.array_equal(orc.feistel_perm(1000, 1234, 0), orc.feistel_perm(1000, 1234, 1))


def test_env_invariants(orc):
    env = orc.Env(Q=8, max_actions=16, N=4, seed=99)
    env.reset()
    assert np.all(env.active == 0x3F)
    assert np.all(np.abs(env.score[:, :24]) <= 2) and np.all(env.score[:, 24:] == 0)
    obs = env.observe()
    assert obs.shape == (4, 32, 72) and np.all(obs[:, 24:, :] == 0)
    rng = np.random.default_rng(0)
    ndone = 0
    for t in range(40):
        for n in range(4):
            if env.done[n]:
                ndone += 1
                env.reset_one(n)
            act = int(env.active[n])
            q = rng.choice([i for i in range(8) if (act >> i) & 1])
            a = 16 * q + int(rng.integers(0, 16))
            env.step_one(n, a)
            assert env.err[n] == 0
            assert -8.0 <= float(env.reward[n]) <= 16.0   # collapse removes a whole quad from the total
            # desired degree = score + degree is conserved: 3 or 4 on active quads, 0 elsewhere
            des = env.score[n].astype(int) + env.degree[n].astype(int)
            actq = np.repeat([(int(env.active[n]) >> q) & 1 for q in range(8)], 4).astype(bool)
            assert np.all((des[actq] == 3) | (des[actq] == 4)) and np.all(des[~actq] == 0)
            assert np.all((env.degree[n][actq] >= 1) & (env.degree[n][actq] <= 7))   # reset can give degree 1
            assert env.steps[n] <= 16
    assert ndone > 0


def test_rollout_ref_vs_dev_modes(orc):
    """Teacher-forced agreement of the two forward orders on recorded states."""
    params = orc.glorot_params(72, 128, 2, seed=1)
    env = orc.Env(Q=8, max_actions=12, N=3, seed=5)
    env.reset()
    ro = orc.collect_rollouts_tn(env, params, 128, 20, mode_dev=True)
    assert np.all(env.err == 0)
    T, N = ro["actions"].shape
    for t in range(T):
        for n in range(N):
            p_ref = orc.action_probabilities(params, 72, 128, ro["states"][t, n], ro["active"][t, n], "ref")
            p_dev = orc.action_probabilities(params, 72, 128, ro["states"][t, n], ro["active"][t, n], "dev")
            assert np.allclose(p_ref, p_dev, rtol=5e-6, atol=1e-9)
            assert ro["p_sel"][t, n] == p_dev[ro["actions"][t, n]]
    assert ro["done"].sum() >= 3
