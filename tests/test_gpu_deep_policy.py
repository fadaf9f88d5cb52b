"""SimplePolicy.Policy(in, hidden, num_hidden_layers, out) for EVERY depth the constructor takes in practice
(test/policy.jl:9-19: Dense(in,h) + (L-1) x Dense(h,h) + Dense(h,out)) and for F = 216 at hidden = 256: the layer-looped
("deep") forms of the forward, backward-data and weight-gradient kernels against the oracle -- device-order fp32 bit for
bit, natural-order fp32 and float64 in logit space for the forward; float64 for the gradient; Adam bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P(ppo):
    if ppo.device_count() < 1:
        pytest.fail("no HIP device: the gpu-marked tests must run on the GPU box")
    return ppo


def _n_params(F, hid, L):
    return hid * F + hid + (L - 1) * (hid * hid + hid) + 4 * hid + 4


CASES = [(72, 128, 1), (72, 128, 3), (72, 128, 4), (72, 256, 1), (72, 256, 3), (72, 256, 4), (216, 128, 3), (216, 256, 2),
         (216, 256, 3), (72, 96, 3), (72, 200, 4)]


@pytest.mark.parametrize("F,hid,L", CASES)
def test_deep_policy_forward(P, orc, F, hid, L):
    rng = np.random.default_rng(F + hid + L)
    pol = P.HipPolicy(F, hid, L, 4, seed=L)
    assert pol.num_params == _n_params(F, hid, L) == orc.mlp_num_params(F, hid, L)
    p0 = (pol.params + (rng.normal(size=pol.num_params) * 0.03).astype(np.float32)).astype(np.float32)
    pol.params = p0
    assert np.array_equal(pol.params, p0), "parameters round-trip in Flux.params order"
    B = 37
    states = rng.integers(-3, 7, size=(B, 32, F)).astype(np.int8)
    active = rng.integers(1, 256, size=B).astype(np.uint32)
    probs = P.batch_action_probabilities(pol, P.StateData(states, active)).T
    assert probs.shape == (B, 128)
    for b in range(B):
        if hid % 32 == 0:
            dev = orc.action_probabilities(p0, F, hid, states[b], active[b], "dev", L)
            assert np.array_equal(probs[b], dev), "device-order oracle must match bit for bit"
        ref = orc.action_probabilities(p0, F, hid, states[b], active[b], "ref", L)       # natural-order fp32
        assert np.allclose(probs[b], ref, rtol=2e-5, atol=1e-8)
        mask = orc.action_mask([(int(active[b]) >> q) & 1 for q in range(8)])
        assert np.all(probs[b][np.isneginf(mask)] == 0.0) and abs(float(probs[b].sum()) - 1.0) < 1e-5
        l64 = orc.mlp_logits(p0, F, hid, states[b], "f64", L)                               # float64, logit space
        on = ~np.isneginf(mask)
        lse = np.log(np.exp(l64[on] - l64[on].max()).sum()) + l64[on].max()
        live = on & (probs[b] > 1e-30)
        tol = 1e-4 * max(1.0, float(np.abs(l64[on]).max()))
        assert np.abs(np.log(probs[b][live].astype(np.float64)) - (l64[live] - lse)).max() <= tol


@pytest.mark.parametrize("hid,L", [(128, 1), (128, 3), (256, 4), (256, 1)])
@pytest.mark.parametrize("compact", [False, True], ids=["expanded", "compact"])
def test_deep_policy_rollout_bitexact(P, orc, hid, L, compact):
    """collect_rollouts! with a deep policy: per-step launches (the one-launch rollout covers L = 2 only), both state
    storage forms; actions, probabilities, rewards, flags and returns bit for bit against the oracle."""
    N, T, ma = 40, 14, 6
    P.set_rollout_compact(compact)
    try:
        env = P.HipVecEnv(num_envs=N, Q=8, max_actions=ma, seed=91, global_offset=3)
        pol = P.HipPolicy(72, hid, L, 4, seed=7)
        ro = P.BufferRollouts()
        P.collect_rollouts_steps_(ro, env, pol, T, 0.99)
        oenv = orc.Env(Q=8, max_actions=ma, N=N, seed=91, global_offset=3)
        oenv.reset()
        ref = orc.collect_rollouts_tn(oenv, pol.params, hid, T, mode_dev=True, n_hidden=L)
        st, act = ro.state_data
        assert np.array_equal(st, ref["states"]) and np.array_equal(act, ref["active"])
        assert np.array_equal(ro.selected_actions - 1, ref["actions"])
        assert np.array_equal(ro.selected_action_probabilities, ref["p_sel"])
        assert np.array_equal(ro.raw_rewards, ref["rewards"]) and np.array_equal(ro.terminal, ref["done"].astype(bool))
        assert np.array_equal(ro.rewards, orc.compute_returns_tn(ref["rewards"], ref["done"], 0.99))
        # one optimiser step from these rollouts (compact form: the train forward re-derives the rows)
        ds = P.construct_dataset(ro)
        sel = np.random.default_rng(1).permutation(len(ds))[:200] + 1
        lp, le = P.forward_backward(pol, ds, sel, 0.05, 0.01)
        s0 = sel - 1
        g64, olp, ole = orc.step_batch_grad_f64(pol.params, 72, hid, st.reshape(-1, 32, 72)[s0], act.reshape(-1)[s0],
                                                ref["actions"].reshape(-1)[s0], ref["p_sel"].reshape(-1)[s0],
                                                ro.rewards.reshape(-1)[s0], 0.05, 0.01, n_hidden=L)
        g = pol.grad()
        assert np.abs(g - g64).max() <= 2e-5 * np.abs(g64).max() + 1e-9
        assert abs(lp - olp) <= 1e-5 * (1 + abs(olp)) and abs(le - ole) <= 1e-5 * (1 + abs(ole))
    finally:
        P.set_rollout_compact(None)


def _off_the_kink(params, F, hid, L, states, delta=1e-5):
    """True per state when no hidden unit's pre-activation lies within `delta` of leakyrelu's kink.  There the fp32 and the
    float64 forward can disagree about the SIGN of the pre-activation (fp32 error of a 72..256-term dot product: ~1e-6),
    and the derivative jumps from 0.01 to 1: such a unit changes a whole row of its layer's gradient by O(1e-3 max|g|) --
    a property of comparing two precisions across a kink (about one unit per 10^7), not of the kernels, and it would make
    the 2e-5 bar a lottery over seeds."""
    from oracle import np_oracle
    a = states.reshape(-1, F).astype(np.float64).T
    ok = np.ones(states.shape[0], bool)
    for (W, b) in np_oracle.unpack_params(params, F, hid, L)[:-1]:
        z = W.astype(np.float64) @ a + b.astype(np.float64)[:, None]
        ok &= (np.abs(z).min(axis=0).reshape(states.shape[0], 32).min(axis=1) >= delta)
        a = np.where(z > 0, z, 0.01 * z)
    return ok


def _random_batch(P, pol, rng, B, F):
    """A minibatch as PPO sees it: random states (none with a hidden unit on the leakyrelu kink, see above), actions
    sampled from the policy's own probabilities and `old` probabilities near the current ones (ratios around 1: the regime
    step_batch! runs in -- an arbitrary p_old would make the per-sample terms huge and the test one of fp32 cancellation
    rather than of the kernels)."""
    cand = rng.integers(-3, 7, size=(2 * B + 8, 32, F)).astype(np.int8)
    keep = _off_the_kink(pol.params, F, pol.hidden_channels, pol.num_hidden_layers, cand)
    assert keep.sum() >= B, "too many states on the kink?"
    states = np.ascontiguousarray(cand[keep][:B])
    active = rng.integers(1, 256, size=B).astype(np.uint32)
    probs = P.batch_action_probabilities(pol, P.StateData(states, active)).T.astype(np.float64)     # [B,128]
    actions = np.zeros(B, np.int32)
    p_old = np.zeros(B, np.float32)
    for b in range(B):
        pr = probs[b] / probs[b].sum()
        actions[b] = rng.choice(128, p=pr)
        p_old[b] = np.float32(probs[b][actions[b]] * rng.uniform(0.8, 1.25))
    adv = rng.normal(size=B).astype(np.float32) * 3
    return states, active, actions, p_old, adv


# (the float64 oracle is scalar C on one host core: ~30 ms per state at HID = 256, L = 3 -- the batch sizes are chosen so the
# whole file stays under two minutes; 600 > 512 tiles runs several tiles per workgroup in every kernel of the path)
GRAD_CASES = [(F, hid, L, B) for (F, hid, L) in CASES for B in (5, 200)] + \
    [(72, 128, 3, 600), (72, 256, 3, 600), (216, 256, 3, 600), (72, 256, 1, 600)]


@pytest.mark.parametrize("F,hid,L,B", GRAD_CASES)
def test_deep_policy_gradient_vs_f64(P, orc, F, hid, L, B):
    """step_batch! gradient (src/train.jl:54-84) of a deep policy from host-supplied rollouts (any F): float64 oracle,
    tolerance 2e-5 of max|g| like the L = 2 kernels; ragged tile counts and more tiles than workgroups."""
    rng = np.random.default_rng(B + F + hid + L)
    pol = P.HipPolicy(F, hid, L, 4, seed=2)
    p0 = (pol.params + (rng.normal(size=pol.num_params) * 0.03).astype(np.float32)).astype(np.float32)
    pol.params = p0
    states, active, actions, p_old, adv = _random_batch(P, pol, rng, B, F)
    ro = P.BufferRollouts()
    ro.set_columns(None, states[None], active[None], actions[None].astype(np.int64) + 1, p_old[None], adv[None])   # by shape: any F
    ds = P.construct_dataset(ro)
    sel = rng.permutation(B) + 1
    lp, le = P.forward_backward(pol, ds, sel, 0.05, 0.01)
    g = pol.grad()
    s0 = sel - 1
    g64, olp, ole = orc.step_batch_grad_f64(p0, F, hid, states[s0], active[s0], actions[s0], p_old[s0], adv[s0], 0.05, 0.01,
                                            n_hidden=L)
    assert g.shape == g64.shape
    assert np.abs(g - g64).max() <= 2e-5 * np.abs(g64).max() + 1e-9, np.abs(g - g64).max() / np.abs(g64).max()
    assert abs(lp - olp) <= 1e-5 * (1 + abs(olp)) and abs(le - ole) <= 1e-5 * (1 + abs(ole))
    g2 = None
    P.forward_backward(pol, ds, sel, 0.05, 0.01)
    g2 = pol.grad()
    assert np.array_equal(g, g2), "fixed-order reductions: the gradient is bitwise reproducible"


@pytest.mark.parametrize("hid,L", [(128, 3), (256, 1), (160, 4)])
def test_deep_policy_adam_and_training(P, orc, hid, L):
    """Flux.update! with legacy Adam on a deep policy: bit for bit against the oracle's Adam given the device gradient
    (the caller's layout at any padded width), then ppo_train! epochs with explicit permutations against the oracle loop."""
    rng = np.random.default_rng(hid + L)
    B, F = 96, 72
    pol = P.HipPolicy(F, hid, L, 4, seed=4)
    p0 = pol.params.copy()
    states, active, actions, p_old, adv = _random_batch(P, pol, rng, B, F)
    env = P.HipVecEnv(num_envs=B, Q=8, max_actions=4, seed=1)
    ro = P.BufferRollouts()
    ro.set_columns(env, states[None], active[None], actions[None].astype(np.int64) + 1, p_old[None], adv[None])
    ds = P.construct_dataset(ro)
    opt = P.Optimiser(P.Adam(1e-3))
    pp, mm, vv, bb = p0.copy(), np.zeros_like(p0), np.zeros_like(p0), np.array([0.9, 0.999])
    for step in range(3):
        sel = rng.permutation(B)[:64] + 1
        P.step_batch_(pol, opt, ds, sel, 0.05, 0.01)
        orc.adam_step(pp, pol.grad(), mm, vv, bb, 1e-3)
        assert np.array_equal(pol.params, pp), "Adam on the device gradient is bit-exact"
    m, v, _ = opt.members[0].get_state()
    assert np.array_equal(m, mm) and np.array_equal(v, vv)
    # ppo_train!: 2 epochs, explicit permutations, short last batch; oracle loop with float64 gradients -> close parameters
    pol.params = p0
    opt = P.Optimiser(P.Adam(1e-3))
    perm = np.stack([rng.permutation(B) + 1 for _ in range(2)])
    ph, eh, _ = P.ppo_train_(pol, opt, ds, 0.05, 40, 2, 0.01, perm=perm, verbose=False)
    pp, mm, vv, bb = p0.copy(), np.zeros_like(p0), np.zeros_like(p0), np.array([0.9, 0.999])
    for ep in range(2):
        losses = []
        for s in range(0, B, 40):
            s0 = perm[ep][s:s + 40] - 1
            g64, lp, le = orc.step_batch_grad_f64(pp, F, hid, states[s0], active[s0], actions[s0], p_old[s0], adv[s0], 0.05, 0.01,
                                                  n_hidden=L)
            losses.append(lp)
            orc.adam_step(pp, g64.astype(np.float32), mm, vv, bb, 1e-3)
        assert abs(ph[ep] - np.mean(losses)) <= 1e-4 * (1 + abs(np.mean(losses)))
    assert np.abs(pol.params - pp).max() <= 1e-4                     # Adam normalises the step: fp32-vs-fp64 gradient sign flips near zero


def test_deep_policy_checkpoint_and_rejections(P, tmp_path):
    """BSON.@save / @load round trip in Flux.params order at depth 3 (checkpoint.py), and what stays rejected."""
    pol = P.HipPolicy(72, 128, 3, 4, seed=9)
    path = str(tmp_path / "deep.bson")
    P.save_policy(path, pol)
    back = P.load_policy(path)
    assert back.num_hidden_layers == 3 and np.array_equal(back.params, pol.params)
    with pytest.raises(P.PPOError):
        P.HipPolicy(72, 128, 3, 4, dtype="bf16")          # bf16 kernels: Policy(72, hidden, 2, 4) only
    with pytest.raises(P.PPOError):
        P.HipPolicy(72, 128, 0, 4)
    with pytest.raises(P.PPOError):
        P.HipPolicy(100, 128, 2, 4)
