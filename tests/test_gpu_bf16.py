"""GPU parity tests of the bf16 compute mode (BASELINE config 5: "bf16 MLP on MFMA + fp32 GAE"; `pytest -m gpu`).

The reference is Float32 only, so this mode has no reference-held vectors: PARITY UNPINNED.  The checker is
oracle/np_oracle.py:*_bf16 -- the same rounding points (weights, layer inputs, dY, dZ2, dZ1 to bfloat16, RNE) with
float64 accumulation.  The device accumulates in fp32 inside v_mfma_f32_32x32x16_bf16, and a 1-ulp difference in an
fp32 sum can flip a bf16 rounding (2^-9 relative on that element), so the comparisons carry tolerances, written
next to each assert.  Index outputs stay exact: the sampled action is the reference's sequential fp32 CDF walk on
the device's own probabilities (checked bit for bit against the oracle sampler fed with those probabilities)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P(ppo):
    if ppo.device_count() < 1:
        pytest.fail("no HIP device: the gpu-marked tests must run on the GPU box")
    return ppo


@pytest.fixture(scope="module")
def npo():
    from oracle import np_oracle
    return np_oracle


def _masks(npo, active, Q):
    return np.stack([npo.action_mask([(int(a) >> q) & 1 for q in range(Q)]) for a in active])


@pytest.mark.parametrize("HID,H", [(128, 32), (256, 32), (128, 128), (256, 128)])
def test_bf16_forward_vs_oracle(P, npo, HID, H):
    rng = np.random.default_rng(HID + H)
    F, Q = 72, H // 4
    pol = P.HipPolicy(F, HID, 2, 4, seed=3, dtype="bf16")
    pol.params = pol.params + (rng.normal(size=pol.num_params) * 0.02).astype(np.float32)
    B = 37                                                      # ragged: not a multiple of the 8 waves of a workgroup
    states = rng.integers(-3, 7, size=(B, H, F)).astype(np.int8)
    active = rng.integers(1, 2 ** Q, size=B, dtype=np.uint64).astype(np.uint32)
    probs = P.batch_action_probabilities(pol, P.StateData(states, active)).T          # [B,A]
    assert probs.shape == (B, 4 * H)
    masks = _masks(npo, active, Q)
    want = npo.action_probabilities_bf16(pol.params, F, HID, states, masks)
    assert np.all(probs[np.isneginf(masks)] == 0.0), "masked actions have probability exactly 0"
    assert np.allclose(probs.sum(axis=1), 1.0, atol=1e-5)
    # tolerance: 5e-3 relative + 1e-6 absolute per probability (an occasional flipped bf16 rounding of a hidden unit
    # moves a logit by ~1e-4); the typical error is far smaller
    err = np.abs(probs - want)
    assert np.all(err <= 5e-3 * want + 1e-6), float((err / (want + 1e-6)).max())
    assert np.median(err[want > 0] / want[want > 0]) < 1e-4
    # and the bf16 mode is the same function as the fp32 mode at lower precision: within 5 % of the fp32 probabilities
    pol32 = P.HipPolicy(F, HID, 2, 4, seed=3)
    pol32.params = pol.params
    p32 = P.batch_action_probabilities(pol32, P.StateData(states, active)).T
    big = p32 > 1e-4
    assert np.all(np.abs(probs[big] - p32[big]) <= 0.05 * p32[big])


def test_bf16_rollout_indices_exact_and_env_path(P, orc, npo):
    """Rollout in bf16 mode: states / rewards / done follow the oracle env driven by the device's actions bit for bit;
    the action of every step is the oracle's sequential CDF walk on the device's recorded probabilities."""
    N, T, HID, max_actions = 48, 20, 256, 9
    env = P.HipVecEnv(num_envs=N, Q=8, max_actions=max_actions, seed=21, global_offset=3)
    pol = P.HipPolicy(72, HID, 2, 4, seed=4, dtype="bf16")
    ro = P.BufferRollouts()
    P.collect_rollouts_steps_(ro, env, pol, T, 0.99, record_probs=True)
    st, act = ro.state_data
    probs = ro.full_probs()                                      # [T,N,A] device probabilities
    a_dev = ro.selected_actions - 1
    masks = _masks(npo, act.reshape(-1), 8)
    want = npo.action_probabilities_bf16(pol.params, 72, HID, st.reshape(-1, 32, 72), masks).reshape(T, N, 128)
    assert np.all(np.abs(probs - want) <= 5e-3 * want + 1e-6)
    # teacher-forced replay of the env with the device's actions
    oenv = orc.Env(Q=8, max_actions=max_actions, N=N, seed=21, global_offset=3)
    oenv.reset()
    for t in range(T):
        assert np.array_equal(oenv.observe(), st[t]) and np.array_equal(oenv.active, act[t])
        tick = oenv.tick.copy()
        for n in range(N):
            u = orc.u01(orc.philox([3 + n, int(tick[n]), 0, 0], [21, 0])[0])
            a, err = orc.categorical_sample(probs[t, n], u)
            assert a == a_dev[t, n] and err == 0, "sampled index must be bit-exact"
            assert ro.selected_action_probabilities[t, n] == probs[t, n, a_dev[t, n]]
            oenv.step_one(n, int(a_dev[t, n]))
            assert oenv.reward[n] == ro.raw_rewards[t, n] and bool(oenv.done[n]) == bool(ro.terminal[t, n])
            if oenv.done[n]:
                oenv.reset_one(n)                                # auto-reset (src/rollout_buffer.jl:74-76)
    assert np.array_equal(ro.rewards, orc.compute_returns_tn(ro.raw_rewards, ro.terminal.astype(np.uint8), 0.99)), \
        "returns stay fp32/fp64 and bit-exact in bf16 mode (fp32 GAE)"


@pytest.mark.parametrize("HID", [128, 256])
def test_bf16_rollout_one_launch_equals_per_step(P, HID):
    """The whole T-step rollout in one launch (MODE 3 of the bf16 forward kernel, env state in LDS) and the three
    launches per step produce the same columns bit for bit, and leave the env in the same state."""
    cols = []
    for persistent in (False, True):
        P.set_rollout_persistent(persistent)
        try:
            env = P.HipVecEnv(num_envs=37, Q=8, max_actions=7, seed=5, global_offset=11)
            pol = P.HipPolicy(72, HID, 2, 4, seed=9, dtype="bf16")
            ro = P.BufferRollouts()
            P.collect_rollouts_steps_(ro, env, pol, 23, 0.99, record_probs=True)
            st, act = ro.state_data
            ro2 = P.BufferRollouts()
            P.collect_rollouts_steps_(ro2, env, pol, 3, 0.99)        # continues from the state the first call left
            cols.append((st, act, ro.selected_actions, ro.selected_action_probabilities, ro.raw_rewards, ro.terminal,
                         ro.rewards, ro.full_probs(), ro2.state_data[0], ro2.selected_actions))
        finally:
            P.set_rollout_persistent(None)
    for a, b in zip(*cols):
        assert np.array_equal(np.asarray(a), np.asarray(b))


def _dataset(P, N, T, HID, seed, Q=8):
    env = P.HipVecEnv(num_envs=N, Q=Q, max_actions=12, seed=seed)
    pol = P.HipPolicy(72, HID, 2, 4, seed=seed + 1, dtype="bf16")
    rng = np.random.default_rng(seed)
    pol.params = pol.params + (rng.normal(size=pol.num_params) * 0.02).astype(np.float32)
    ro = P.BufferRollouts()
    P.collect_rollouts_steps_(ro, env, pol, T, 1.0)
    return env, pol, ro, P.construct_dataset(ro)


@pytest.fixture(params=[False, True], ids=["expanded", "compact"])
def storage_mode(request, P):
    """Expanded observation rows vs env snapshots re-derived by the train forward (ppo_set_rollout_compact)."""
    P.set_rollout_compact(request.param)
    yield request.param
    P.set_rollout_compact(None)


@pytest.mark.parametrize("HID,B,Q", [(128, 24, 8), (128, 300, 8), (256, 40, 8), (256, 520, 8), (256, 30, 32), (128, 30, 32)])
def test_bf16_gradient_vs_oracle(P, npo, HID, B, Q, storage_mode):
    N, T, H = 40, 16, 4 * Q
    env, pol, ro, ds = _dataset(P, N, T, HID, seed=B + Q, Q=Q)
    rng = np.random.default_rng(B)
    sel = rng.choice(len(ds), size=B, replace=B > len(ds))
    st, act = ro.state_data
    st = st.reshape(-1, H, 72)[sel]
    masks = _masks(npo, act.reshape(-1)[sel], Q)
    a0 = (ro.selected_actions.reshape(-1)[sel] - 1).astype(np.int64)
    po = ro.selected_action_probabilities.reshape(-1)[sel]
    adv = ro.rewards.reshape(-1)[sel]
    eps = 10.0       # never clipped: a clipped/unclipped flip of a boundary sample is a whole-sample difference, not rounding
    lp, le = P.forward_backward(pol, ds, sel + 1, eps, 0.01)
    g = pol.grad()
    g16, olp, ole = npo.step_batch_grad_bf16(pol.params, 72, HID, st, masks, a0, po, adv, eps, 0.01)
    scale = np.abs(g16).max()
    assert scale > 0
    # tolerance: 1 % of max|g| per element (fp32 vs float64 accumulation in front of three bf16 roundings), and the
    # whole gradient within 0.3 % in the 2-norm
    assert np.abs(g - g16).max() <= 1e-2 * scale, float(np.abs(g - g16).max() / scale)
    assert np.linalg.norm(g - g16) <= 3e-3 * np.linalg.norm(g16), float(np.linalg.norm(g - g16) / np.linalg.norm(g16))
    assert abs(lp - olp) <= 2e-3 * (1 + abs(olp)) and abs(le - ole) <= 2e-3 * (1 + abs(ole))
    # every block of the flat gradient is populated (dW1 comes from the second kernel)
    n1 = HID * 72
    assert np.abs(g[:n1]).max() > 0 and np.abs(g[n1:n1 + HID]).max() > 0 and np.abs(g[-4:]).max() > 0
    P.forward_backward(pol, ds, sel + 1, eps, 0.01)
    assert np.array_equal(g, pol.grad()), "run-to-run bitwise reproducible"


@pytest.mark.parametrize("B", [1, 2, 5, 255, 257])
def test_bf16_gradient_ragged_batch_sizes(P, npo, B):
    """Minibatches that do not fill the 256-workgroup grid (the dW1 kernel shares the tile -> workgroup assignment)."""
    env, pol, ro, ds = _dataset(P, 30, 10, 256, seed=55)          # 300 samples
    rng = np.random.default_rng(B)
    sel = rng.integers(0, len(ds), size=B)
    st, act = ro.state_data
    masks = _masks(npo, act.reshape(-1)[sel], 8)
    a0 = (ro.selected_actions.reshape(-1)[sel] - 1).astype(np.int64)
    po, adv = ro.selected_action_probabilities.reshape(-1)[sel], ro.rewards.reshape(-1)[sel]
    P.forward_backward(pol, ds, sel + 1, 10.0, 0.01)
    g16, _, _ = npo.step_batch_grad_bf16(pol.params, 72, 256, st.reshape(-1, 32, 72)[sel], masks, a0, po, adv, 10.0, 0.01)
    assert np.abs(pol.grad() - g16).max() <= 1e-2 * np.abs(g16).max()
    assert np.linalg.norm(pol.grad() - g16) <= 3e-3 * np.linalg.norm(g16)


def test_bf16_clipped_samples_carry_only_entropy_gradient(P, npo):
    env, pol, ro, ds = _dataset(P, 16, 8, 128, seed=3)
    sel = np.arange(64)
    st, act = ro.state_data
    masks = _masks(npo, act.reshape(-1)[sel], 8)
    a0 = (ro.selected_actions.reshape(-1)[sel] - 1).astype(np.int64)
    po = ro.selected_action_probabilities.reshape(-1)[sel]
    adv = ro.rewards.reshape(-1)[sel]
    P.forward_backward(pol, ds, sel + 1, 1e-3, 0.02)            # ratio == 1 > 1 - eps for adv < 0 ... : clipped side
    g = pol.grad()
    g16, _, _ = npo.step_batch_grad_bf16(pol.params, 72, 128, st.reshape(-1, 32, 72)[sel], masks, a0, po, adv, 1e-3, 0.02)
    assert np.abs(g - g16).max() <= 1e-2 * np.abs(g16).max() + 1e-9


def test_bf16_adam_on_fp32_master_weights(P, orc):
    """Adam runs on the fp32 master parameters exactly as in fp32 mode (bit-exact vs the oracle's Adam applied to the
    device gradient); the bf16 fragment streams are re-packed by the same kernel (the next forward sees them)."""
    env, pol, ro, ds = _dataset(P, 16, 8, 128, seed=9)
    opt = P.Optimiser(P.Adam(1e-3))
    p = pol.params.copy()
    m, v, bp = np.zeros_like(p), np.zeros_like(p), np.array([0.9, 0.999])
    st, act = ro.state_data
    probe = P.StateData(st.reshape(-1, 32, 72)[:5], act.reshape(-1)[:5])
    before = P.batch_action_probabilities(pol, probe)
    sel = np.arange(1, 33)
    for it in range(2):
        P.forward_backward(pol, ds, sel, 0.05, 0.01)
        g = pol.grad()
        P.step_batch_(pol, opt, ds, sel, 0.05, 0.01)
        orc.adam_step(p, g, m, v, bp, 1e-3)
        assert np.array_equal(pol.params, p)
    after = P.batch_action_probabilities(pol, probe)
    assert not np.array_equal(before, after), "re-packed bf16 weights reach the next forward"
    ref = P.HipPolicy(72, 128, 2, 4, seed=0, dtype="bf16")
    ref.params = p
    assert np.array_equal(P.batch_action_probabilities(ref, probe), after), "packs written by k_adam == packs written by set_params"


def test_bf16_learning_signal(P):
    """PPO in bf16 mode learns the synthetic env like the fp32 mode does."""
    env = P.HipVecEnv(num_envs=256, Q=8, max_actions=32, seed=3)
    pol = P.HipPolicy(72, 128, 2, 4, seed=0, dtype="bf16")
    opt = P.Optimiser(P.Adam(3e-4))
    means = []
    for it in range(12):
        ro = P.BufferRollouts()
        P.collect_rollouts_steps_(ro, env, pol, 32, 1.0)
        means.append(float(ro.raw_rewards.mean()))
        ds = P.construct_dataset(ro)
        P.ppo_train_(pol, opt, ds, 0.1, 1024, 2, 0.01, seed=it, verbose=False)
    assert np.mean(means[-3:]) > np.mean(means[:3]) + 0.05, means


def test_bf16_config5_size_properties(P):
    """BASELINE config 5 shapes: 65536 envs, 2x256 bf16 MLP, fp32 returns.  Size-independent properties only."""
    N, T = 65536, 4
    env = P.HipVecEnv(num_envs=N, Q=8, max_actions=16, seed=5)
    pol = P.HipPolicy(72, 256, 2, 4, seed=1, dtype="bf16")
    ro = P.BufferRollouts()
    P.collect_rollouts_steps_(ro, env, pol, T, 1.0)
    assert len(ro) == N * T
    a = ro.selected_actions - 1
    st, act = ro.state_data
    assert a.min() >= 0 and a.max() < 128
    assert np.all((act.reshape(-1) >> (a.reshape(-1) // 16)) & 1 == 1), "only unmasked actions are ever sampled"
    ps = ro.selected_action_probabilities
    assert np.all(ps > 0) and np.all(ps <= 1.0)
    ds = P.construct_dataset(ro)
    opt = P.Optimiser(P.Adam(1e-4))
    h = P.ppo_train_(pol, opt, ds, 0.05, 65536, 1, 0.01, seed=0, verbose=False)
    assert np.all(np.isfinite(h[0])) and np.all(np.isfinite(h[1])) and np.all(np.isfinite(pol.params))


def test_dtype_errors(P):
    with pytest.raises(P.PPOError):
        P.HipPolicy(72, 128, 2, 4, dtype="fp8")
    with pytest.raises(P.PPOError):
        P.HipPolicy(216, 128, 2, 4, dtype="bf16")              # the bf16 kernels cover Policy(72, hidden, 2, 4): ppo_policy_set_dtype says so
