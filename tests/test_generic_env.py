"""The generic (user-defined env / policy) methods of the path: the reference's own fake env script
test/test_rollout_buffer.jl:4-50 restated (TestEnv: reward 1.0, horizon 10, policy [1,0,0]; 10 episodes), the dead
utilities permute!/shuffle!, single_trajectory_return / average_returns, and a user env whose states are StateData
feeding the MFMA training path."""
import numpy as np
import pytest


class TestEnv:                      # test/test_rollout_buffer.jl:4-12
    __test__ = False

    def __init__(self, horizon=10):
        self.num_actions, self.max_actions = 0, horizon


class TestPolicy:                   # test/test_rollout_buffer.jl:14-16 (a policy object with no parameters)
    __test__ = False


@pytest.fixture(scope="module")
def P(ppo):
    # methods of the plugin generic functions for the fake env (test/test_rollout_buffer.jl:18-39)
    if not getattr(ppo, "_testenv_registered", False):
        ppo.state.register(TestEnv)(lambda env: np.array([1, 2, 3, 4, 5], np.int64))
        ppo.reward.register(TestEnv)(lambda env: 1.0)
        ppo.is_terminal.register(TestEnv)(lambda env: env.num_actions >= env.max_actions)
        ppo.reset_.register(TestEnv)(lambda env: setattr(env, "num_actions", 0))
        ppo.step_.register(TestEnv)(lambda env, action: setattr(env, "num_actions", env.num_actions + 1))
        ppo.action_probabilities.register(TestPolicy)(lambda policy, state: np.array([1.0, 0.0, 0.0], np.float32))
        ppo._testenv_registered = True
    return ppo


def _filled(P, n):
    ro = P.BufferRollouts()
    for i in range(n):
        P.update_rollouts_(ro, np.array([i]), 0.25 * (i + 1), i + 1, float(i), i == n - 1)
    return ro


def test_update_length_permute_shuffle_host_only(P):
    ro = _filled(P, 4)
    assert len(ro) == 4
    P.permute_(ro, [4, 3, 2, 1])
    assert ro._host.selected_actions == [4, 3, 2, 1] and ro._host.terminal == [True, False, False, False]
    assert [float(x) for x in ro._host.rewards] == [3.0, 2.0, 1.0, 0.0]
    with pytest.raises(AssertionError):
        P.permute_(ro, [1, 1, 2, 3])
    P.shuffle_(ro, np.random.default_rng(0))
    assert sorted(ro._host.selected_actions) == [1, 2, 3, 4]
    ro._host.rewards.pop()
    with pytest.raises(AssertionError):
        len(ro)                                     # Base.length asserts equal column lengths (rollout_buffer.jl:40-48)


def test_host_dataset_getindex(P):
    ro = _filled(P, 5)
    ds = P.HostDataset(P, ro._host)
    s = ds[2]
    assert s["selected_action"] == 2 and s["selected_action_probability"] == np.float32(0.5) and s["returns"] == 1.0
    with pytest.raises(AssertionError):
        ds[0]
    with pytest.raises(AssertionError):
        ds[6]
    with pytest.raises(TypeError):
        ds["a"]
    with pytest.raises(P.PPOError, match="batch_state needs to be overloaded"):
        ds[[1, 2]]                                  # states are plain arrays: the user never overloaded batch_state


def test_entropy_helpers_match_the_oracle(P, orc):
    rng = np.random.default_rng(0)
    p = rng.random((16, 5)).astype(np.float32)
    p /= p.sum(axis=0, keepdims=True)
    lin = orc.linear_action_index(np.ones(5, np.int64), 16)
    _, entloss = orc.ppo_loss_with_entropy(p.T.copy(), lin, np.ones(5, np.float32), np.ones(5, np.float32), 0.05)
    assert abs(P.smoothed_entropy(p) - (-entloss)) < 1e-6
    assert abs(P.clamped_entropy(p) - P.smoothed_entropy(p)) < 1e-5      # no tiny probabilities here


@pytest.mark.gpu
def test_reference_rollout_buffer_script(P):
    """test/test_rollout_buffer.jl:41-50: 10 episodes of the fake env -> 100 samples, p = 1.0, a = 1,
    returns = ten repeats of [10, 9, ..., 1]."""
    env, policy = TestEnv(10), TestPolicy()
    rollouts = P.BufferRollouts()
    P.collect_rollouts_(rollouts, env, policy, 10, 1.0)
    h = rollouts._host
    assert len(rollouts) == 100
    assert all(a == 1 for a in h.selected_actions) and all(p == np.float32(1.0) for p in h.selected_action_probabilities)
    assert [float(r) for r in h.rewards] == list(range(10, 0, -1)) * 10
    assert h.terminal == ([False] * 9 + [True]) * 10
    assert all(np.array_equal(s, [1, 2, 3, 4, 5]) for s in h.state_data)
    ds = P.construct_dataset(rollouts)               # plain-array states: host dataset (no StateData -> not uploaded)
    assert isinstance(ds, P.HostDataset) and len(ds) == 100 and ds[1]["returns"] == 10.0
    m, s = P.average_returns(policy, env, 5)         # src/evaluate.jl:18-25, generic method
    assert m == 10.0 and s == 0.0
    P.reset_(env)
    assert P.single_trajectory_return(policy, env) == 10.0
    # an env without plugin methods fails like the reference does
    with pytest.raises(P.PPOError, match="needs to be overloaded"):
        P.collect_rollouts_(P.BufferRollouts(), object(), policy, 1, 1.0)


class GridEnv:
    """A user env producing StateData states (so the engine can train on it): deterministic feature rows, reward =
    1 for action type 0 else 0, horizon 6."""
    __test__ = False

    def __init__(self, seed):
        self.rng, self.t, self.r = np.random.default_rng(seed), 0, 0.0
        self.obs = None


@pytest.mark.gpu
def test_user_env_feeds_the_training_path(P, orc):
    if not getattr(P, "_gridenv_registered", False):
        def _reset(env):
            env.t, env.r = 0, 0.0
            env.obs = env.rng.integers(-2, 3, size=(32, 72)).astype(np.int8)
        P.reset_.register(GridEnv)(_reset)
        P.state.register(GridEnv)(lambda env: P.StateData(env.obs.copy(), np.uint32(0x3F)))
        P.reward.register(GridEnv)(lambda env: env.r)
        P.is_terminal.register(GridEnv)(lambda env: env.t >= 6)

        def _step(env, action):
            env.t += 1
            env.r = 1.0 if (action - 1) % 4 == 0 else 0.0
            env.obs = env.rng.integers(-2, 3, size=(32, 72)).astype(np.int8)
        P.step_.register(GridEnv)(_step)
        P._gridenv_registered = True
    env = GridEnv(0)
    pol = P.HipPolicy(72, 128, 2, 4, seed=2)
    ro = P.BufferRollouts()
    P.collect_rollouts_(ro, env, pol, 8, 1.0)         # HipPolicy.action_probabilities per step through the C ABI
    assert len(ro) == 48
    h = ro._host
    acts = np.array(h.selected_actions)
    assert np.all((acts >= 1) & (acts <= 96)), "quads 6 and 7 are masked"
    ds = P.construct_dataset(ro)                      # uploaded to the device buffer
    assert isinstance(ds, P.BufferDataset) and len(ds) == 48
    b = ds[[1, 2, 3]]
    assert b["state"].vertex_score.shape == (3, 32, 72) and np.array_equal(b["selected_action"], acts[:3])
    # gradient of a minibatch of these host-collected samples vs the f64 oracle
    sel = np.arange(1, 33)
    P.forward_backward(pol, ds, sel, 0.05, 0.01)
    st = np.stack([s.vertex_score for s in h.state_data])[:32]
    g64, _, _ = orc.step_batch_grad_f64(pol.params, 72, 128, st, np.full(32, 0x3F, np.uint32), (acts[:32] - 1).astype(np.int32),
                                        np.asarray(h.selected_action_probabilities, np.float32)[:32],
                                        np.asarray(h.rewards, np.float32)[:32], 0.05, 0.01)
    assert np.abs(pol.grad() - g64).max() <= 2e-5 * np.abs(g64).max() + 1e-9
    lp, le = P.step_epoch_(pol, P.Optimiser(P.Adam(1e-3)), ds, 0.05, 16, 0.01, seed=1)
    assert np.isfinite(lp) and np.isfinite(le)
