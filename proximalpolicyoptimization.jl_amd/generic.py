"""Generic (user-defined env / policy) side of the path: the reference's own control flow over the plugin functions,
for objects that are NOT the built-in batched HipVecEnv (BASELINE config 1: one env, plumbing).

    collect_step_data!     src/collect_rollouts.jl:1-15      collect_episode_data!   :17-24
    update!                src/rollout_buffer.jl:24-38       compute_state_value!    :55-64
    permute! / shuffle!    :81-93                            single_trajectory_return src/evaluate.jl:1-16

The env and the policy are whatever the user registered methods for (`state`, `reward`, `is_terminal`, `reset_`,
`step_`, `action_probabilities`); sampling is the HIP engine's categorical kernel (ppo_categorical_sample, the
reference's sequential CDF walk) fed with a host uniform, returns come from the HIP return scan
(ppo_compute_returns).  A HostRollouts of StateData states converts to the device buffer for training
(construct_dataset -> ppo_rollouts_set), so `ppo_train_` is the same MFMA path as for the batched envs.
"""
import numpy as np


class HostRollouts:
    """The reference's BufferRollouts struct itself (src/rollout_buffer.jl:1-22): five growable columns."""

    def __init__(self):
        self.state_data = []
        self.selected_action_probabilities = []
        self.selected_actions = []
        self.rewards = []
        self.terminal = []

    def __len__(self):                                            # Base.length with its @assert (:40-48)
        n = len(self.state_data)
        if not (len(self.selected_action_probabilities) == len(self.selected_actions) == len(self.rewards)
                == len(self.terminal) == n):
            raise AssertionError("BufferRollouts columns differ in length")
        return n


def update_host_(buffer, state, action_probability, action, reward, terminal):
    """update!(buffer, state, action_probability, action, reward, terminal)  (:24-38): five push!es."""
    buffer.state_data.append(state)
    buffer.selected_action_probabilities.append(np.float32(action_probability))
    buffer.selected_actions.append(int(action))
    buffer.rewards.append(np.float32(reward))
    buffer.terminal.append(bool(terminal))


def compute_state_value_(P, buffer, discount):
    """compute_state_value!(rollouts, discount) (:55-64): rewards .= compute_returns(rewards, terminal, discount)."""
    if len(buffer):
        ret = P.compute_returns(np.asarray(buffer.rewards, np.float32), np.asarray(buffer.terminal, np.uint8), discount)
        buffer.rewards = [np.float32(x) for x in ret]


def permute_(buffer, idx):
    """permute!(rollouts, idx) (:81-88), 1-based permutation like the reference."""
    ii = [int(i) - 1 for i in idx]
    if sorted(ii) != list(range(len(buffer))):
        raise AssertionError("idx is not a permutation of 1:length(rollouts)")
    for name in ("state_data", "selected_action_probabilities", "selected_actions", "rewards", "terminal"):
        col = getattr(buffer, name)
        setattr(buffer, name, [col[i] for i in ii])


def shuffle_(buffer, rng=None):
    """shuffle!(rollouts) (:90-93)."""
    rng = rng or np.random.default_rng()
    permute_(buffer, rng.permutation(len(buffer)) + 1)


def collect_step_data_(P, buffer, env, policy, rng):
    """collect_step_data!(buffer, env, policy) (src/collect_rollouts.jl:1-15)."""
    cpu_state = P.state(env)                                               # :2
    ap = np.ascontiguousarray(P.action_probabilities(policy, cpu_state), np.float32)   # :5
    a1, p_sel, err = P.categorical_sample(ap[None, :], np.array([rng.random(dtype=np.float32)], np.float32))  # :6
    if err[0]:
        raise AssertionError("ap[a] > 0.0")                                # :7
    action = int(a1[0])                                                    # 1-based like the reference
    P.step_(env, action)                                                   # :9
    r = P.reward(env)                                                      # :11
    t = P.is_terminal(env)                                                 # :12
    update_host_(buffer, cpu_state, ap[action - 1], action, r, t)          # :14
    assert float(p_sel[0]) == float(ap[action - 1])


def collect_episode_data_(P, buffer, env, policy, rng):
    """collect_episode_data!(buffer, env, policy) (:17-24)."""
    terminal = P.is_terminal(env)
    while not terminal:
        collect_step_data_(P, buffer, env, policy, rng)
        terminal = P.is_terminal(env)


def collect_rollouts_host_(P, rollouts, env, policy, num_episodes, discount, rng=None):
    """collect_rollouts!(rollouts, env, policy, num_episodes, discount) (src/rollout_buffer.jl:66-79), generic method."""
    rng = rng or np.random.default_rng()
    for _ in range(int(num_episodes)):
        P.reset_(env)                                                      # :75
        collect_episode_data_(P, rollouts, env, policy, rng)               # :76
    compute_state_value_(P, rollouts, discount)                            # :78


def single_trajectory_return(P, policy, env, rng=None):
    """single_trajectory_return(policy, env) (src/evaluate.jl:1-16): undiscounted return of one stochastic episode
    from the env's CURRENT state (the caller resets)."""
    rng = rng or np.random.default_rng()
    ret = 0.0
    done = P.is_terminal(env)
    while not done:
        ap = np.ascontiguousarray(P.action_probabilities(policy, P.state(env)), np.float32)
        a1, _, _ = P.categorical_sample(ap[None, :], np.array([rng.random(dtype=np.float32)], np.float32))
        P.step_(env, int(a1[0]))
        ret += float(P.reward(env))
        done = P.is_terminal(env)
    return ret


def average_returns_host(P, policy, env, num_trajectories, rng=None):
    """average_returns(policy, env, num_trajectories) (src/evaluate.jl:18-25), generic method: mean and sample std."""
    rets = []
    for _ in range(int(num_trajectories)):
        P.reset_(env)
        rets.append(single_trajectory_return(P, policy, env, rng))
    r = np.asarray(rets, np.float64)
    return float(r.mean()), float(r.std(ddof=1)) if r.size > 1 else float("nan")


class HostDataset:
    """BufferDataset over a HostRollouts (src/rollout_buffer.jl:95-147): getindex(Int | Vector) with the user's
    batch_state plugin, 1-based indices."""

    def __init__(self, P, rollouts):
        self._P, self.rollouts = P, rollouts

    def __len__(self):
        return len(self.rollouts)

    def __getitem__(self, idx):
        r, n = self.rollouts, len(self.rollouts)
        if isinstance(idx, (int, np.integer)):
            if not (1 <= idx <= n):
                raise AssertionError("1 <= idx <= length(rollouts)")                       # :105-106
            i = int(idx) - 1
            return {"state": r.state_data[i], "selected_action": r.selected_actions[i],
                    "selected_action_probability": r.selected_action_probabilities[i], "returns": r.rewards[i]}
        if isinstance(idx, (list, tuple, np.ndarray)):
            ii = [int(i) - 1 for i in idx]
            if ii and (min(ii) < 0 or max(ii) >= n):
                raise AssertionError("dataset index out of range")
            return {"state": self._P.batch_state([r.state_data[i] for i in ii]),                 # :122
                    "selected_action": np.array([r.selected_actions[i] for i in ii], np.int64),
                    "selected_action_probability": np.array([r.selected_action_probabilities[i] for i in ii], np.float32),
                    "returns": np.array([r.rewards[i] for i in ii], np.float32)}
        raise TypeError("Dataset index should be Int or Array, got %s" % type(idx).__name__)    # :141
