"""Two data-parallel ranks on ONE GPU (gloo for the exchange, host round trip of the gradient buffer): the real engine
through ppo_train with world = 2 -- env shards keyed by global env id, per-rank minibatches, one all-reduce per
optimiser step, identical Adam on both replicas.  The 8-GPU RCCL run cannot be rehearsed on the one-GPU box; this is
the closest the distributed control flow gets to a real execution before it."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_PER_RANK, T, HID, B = 24, 8, 128, 80          # 192 local samples: minibatches of 80, 80, 32


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, base):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch
    import torch.distributed as dist
    import ppo_amd as P
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dp = P.DataParallel(rank, world)
    off, n = dp.env_shard(world * N_PER_RANK)
    env = P.HipVecEnv(num_envs=n, Q=8, max_actions=10, seed=5, global_offset=off)
    pol = P.HipPolicy(72, HID, 2, 4, seed=3)
    opt = P.Optimiser(P.Adam(1e-3))
    ro = P.BufferRollouts()
    P.collect_rollouts_steps_(ro, env, pol, T, 1.0)
    ds = P.construct_dataset(ro)
    # 1. one optimiser step over the whole local shard: the all-reduced gradient is the mean over the union
    P.ppo_train_(pol, opt, ds, 0.05, len(ds), 1, 0.01, parallel=dp, verbose=False)
    torch.cuda.synchronize()
    grad1 = pol.grad()
    # 2. two more epochs of ragged minibatches (the last one of each epoch is short)
    perm = np.stack([np.random.default_rng(100 + e).permutation(len(ds)) + 1 for e in range(2)])   # same local order on both ranks
    ph, eh, _ = P.ppo_train_(pol, opt, ds, 0.05, B, 2, 0.01, perm=perm, parallel=dp, verbose=False)
    torch.cuda.synchronize()
    np.savez(base + "_rank%d.npz" % rank, params=pol.params, ph=ph, eh=eh, actions=ro.selected_actions,
             rewards=ro.raw_rewards, grad1=grad1)
    dist.destroy_process_group()


def test_two_ranks_one_gpu_stay_in_sync_and_match_the_union(ppo, orc, tmp_path):
    import torch.multiprocessing as mp
    base = str(tmp_path / "dp")
    mp.spawn(_rank_main, args=(2, _free_port(), base), nprocs=2, join=True)
    r0, r1 = np.load(base + "_rank0.npz"), np.load(base + "_rank1.npz")
    assert np.array_equal(r0["params"], r1["params"]), "replicas must hold bit-identical parameters after training"
    assert np.array_equal(r0["ph"], r1["ph"]) and np.array_equal(r0["eh"], r1["eh"]), "loss history is the global one"
    # the shards are the columns of the unsharded run (RNG keyed by global env id)
    P = ppo
    env = P.HipVecEnv(num_envs=2 * N_PER_RANK, Q=8, max_actions=10, seed=5)
    pol = P.HipPolicy(72, HID, 2, 4, seed=3)
    p0 = pol.params.copy()
    ro = P.BufferRollouts()
    P.collect_rollouts_steps_(ro, env, pol, T, 1.0)
    assert np.array_equal(ro.selected_actions[:, :N_PER_RANK], r0["actions"])
    assert np.array_equal(ro.selected_actions[:, N_PER_RANK:], r1["actions"])
    assert np.array_equal(ro.raw_rewards[:, N_PER_RANK:], r1["rewards"])
    # the first optimiser step of the 2-rank run used the mean gradient over the union of the two shards: the same
    # gradient from one process over all 2*N_PER_RANK envs (different slab partition -> fp32 rounding only), and from
    # the float64 oracle
    assert np.array_equal(r0["grad1"], r1["grad1"])
    ds = P.construct_dataset(ro)
    allidx = np.arange(1, len(ds) + 1)
    P.forward_backward(pol, ds, allidx, 0.05, 0.01)
    g_union = pol.grad()
    scale = np.abs(g_union).max()
    assert np.abs(r0["grad1"] - g_union).max() <= 2e-6 * scale
    st, act = ro.state_data
    g64, _, _ = orc.step_batch_grad_f64(p0, 72, HID, st.reshape(-1, 32, 72), act.reshape(-1),
                                        (ro.selected_actions.reshape(-1) - 1).astype(np.int32),
                                        ro.selected_action_probabilities.reshape(-1), ro.rewards.reshape(-1), 0.05, 0.01)
    assert np.abs(r0["grad1"] - g64).max() <= 2e-5 * np.abs(g64).max() + 1e-9
    # and training moved the replicas away from the initial parameters
    assert not np.array_equal(r0["params"], p0) and np.all(np.isfinite(r0["params"]))


@pytest.fixture(scope="module")
def P(ppo):
    if ppo.device_count() < 1:
        pytest.fail("no HIP device: the gpu-marked tests must run on the GPU box")
    return ppo


def test_native_rccl_hook_single_rank(P):
    """ppo_rccl_* (include/ppo_hip.h): a one-rank communicator; the in-library all-reduce is the identity there, so a
    training run through it must leave exactly the parameters of a run without a hook."""
    import ctypes as C
    uid = np.zeros(128, np.uint8)
    P.call("ppo_rccl_unique_id", uid.ctypes.data_as(C.c_void_p))
    assert uid.any()
    P.call("ppo_rccl_init", 0, 1, uid.ctypes.data_as(C.c_void_p))

    class Native:
        world, force_hook = 1, True

        def make_hook(self, policy):
            return P._lib.ALLREDUCE_FN(C.cast(P._lib.lib().ppo_rccl_allreduce, C.c_void_p).value)

    try:
        got = []
        for par in (None, Native()):
            env = P.HipVecEnv(num_envs=64, Q=8, max_actions=10, seed=3)
            pol = P.HipPolicy(72, 128, 2, 4, seed=5)
            opt = P.Optimiser(P.Adam(1e-3))
            ro = P.BufferRollouts()
            P.collect_rollouts_steps_(ro, env, pol, 8, 1.0)
            perm = np.stack([np.random.default_rng(e).permutation(64 * 8) + 1 for e in range(2)])   # same batches both runs
            ph, eh, _ = P.ppo_train_(pol, opt, P.construct_dataset(ro), 0.05, 128, 2, 0.01, seed=1, parallel=par,
                                     perm=perm, verbose=False)
            got.append((pol.params.copy(), ph, eh))
        assert np.array_equal(got[0][0], got[1][0]) and got[0][1] == got[1][1] and got[0][2] == got[1][2]
    finally:
        P.call("ppo_rccl_finalize")
