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
        rank, world, force_hook = 0, 1, True

        def make_hook(self, policy):
            return P._lib.ALLREDUCE_FN(C.cast(P._lib.lib().ppo_rccl_allreduce, C.c_void_p).value)

    try:
        ok = C.c_int32(0)
        P.call("ppo_rccl_self_test", C.byref(ok))
        assert ok.value == 1 and P.rccl_comm_info() == (0, 1)
        got = []
        for par in (None, Native()):
            env = P.HipVecEnv(num_envs=64, Q=8, max_actions=10, seed=3)
            pol = P.HipPolicy(72, 128, 2, 4, seed=5)
            opt = P.Optimiser(P.Adam(1e-3))
            ro = P.BufferRollouts()
            P.collect_rollouts_steps_(ro, env, pol, 8, 1.0)
            # seed-only path (device Feistel permutation): the minibatch order depends on (seed, epochs this optimiser
            # has trained) and on nothing else, so both runs see the same batches
            ph, eh, _ = P.ppo_train_(pol, opt, P.construct_dataset(ro), 0.05, 128, 2, 0.01, seed=1, parallel=par,
                                     verbose=False)
            got.append((pol.params.copy(), ph, eh))
        assert np.array_equal(got[0][0], got[1][0]) and got[0][1] == got[1][1] and got[0][2] == got[1][2]
    finally:
        P.call("ppo_rccl_finalize")


def test_same_seed_twice_in_one_process_gives_identical_parameters(P):
    """No hidden process-global state in ppo_train: (seed, fresh optimiser) fixes the minibatch order; a restored epoch
    count resumes it (ppo_adam_get/set_epoch_count)."""
    import ctypes as C

    def run(epochs_first, epochs_second, restore=None):
        env = P.HipVecEnv(num_envs=32, Q=8, max_actions=10, seed=3)
        pol = P.HipPolicy(72, 128, 2, 4, seed=5)
        opt = P.Optimiser(P.Adam(1e-3))
        ro = P.BufferRollouts()
        P.collect_rollouts_steps_(ro, env, pol, 8, 1.0)
        ds = P.construct_dataset(ro)
        P.ppo_train_(pol, opt, ds, 0.05, 64, epochs_first, 0.01, seed=9, verbose=False)
        if restore is not None:
            P.call("ppo_adam_set_epoch_count", opt.members[0]._h, restore)
        if epochs_second:
            P.ppo_train_(pol, opt, ds, 0.05, 64, epochs_second, 0.01, seed=9, verbose=False)
        n = C.c_int64(-1)
        P.call("ppo_adam_get_epoch_count", opt.members[0]._h, C.byref(n))
        return pol.params.copy(), n.value

    a, na = run(3, 0)
    b, nb = run(3, 0)
    assert np.array_equal(a, b) and na == nb == 3, "same seed, same process: identical parameters"
    c, nc = run(1, 2)
    assert np.array_equal(a, c) and nc == 3, "1 + 2 epochs continue the epoch count of 3 epochs"
    d, _ = run(1, 2, restore=0)
    assert not np.array_equal(a, d), "a rewound epoch count replays the first permutation (different batches)"


def _unequal_rank_main(rank, world, port, base):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch
    import torch.distributed as dist
    import ppo_amd as P
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dp = P.DataParallel(rank, world)
    off, n = dp.env_shard(27)                        # 14 + 13 envs: remainder rank holds one env more
    env = P.HipVecEnv(num_envs=n, Q=8, max_actions=10, seed=5, global_offset=off)
    pol = P.HipPolicy(72, HID, 2, 4, seed=3)
    opt = P.Optimiser(P.Adam(1e-3))
    ro = P.BufferRollouts()
    P.collect_rollouts_steps_(ro, env, pol, 6, 1.0)   # 84 / 78 local samples
    ds = P.construct_dataset(ro)
    # batch 40 per rank: steps (40+40, 40+38, 4+0): the last step has a zero-sample rank, the second a ragged one
    perm = np.stack([np.arange(len(ds)) + 1])
    ph, eh, _ = P.ppo_train_(pol, opt, ds, 0.05, 40, 1, 0.01, perm=perm, parallel=dp, verbose=False)
    torch.cuda.synchronize()
    st, act = ro.state_data
    np.savez(base + "_u%d.npz" % rank, params=pol.params, ph=ph, eh=eh, st=st, act=act, a=ro.selected_actions,
             p=ro.selected_action_probabilities, r=ro.rewards)
    # a batch size above the SHORTEST shard is rejected on every rank alike (no rank left waiting in a collective)
    try:
        P.ppo_train_(pol, opt, ds, 0.05, 80, 1, 0.01, parallel=dp, verbose=False)
        bad = 0
    except P.PPOError as e:
        bad = int("batch_size" in str(e))
    np.save(base + "_ubad%d.npy" % rank, np.array([bad]))
    dist.destroy_process_group()


def test_two_ranks_with_unequal_shards(ppo, orc, tmp_path):
    """Remainder envs give the ranks different dataset lengths: same number of collectives on both (no deadlock), every
    step's gradient is the exact mean over the samples both ranks contributed (checked against the float64 oracle run
    over the union batches), replicas bit-identical."""
    import torch.multiprocessing as mp
    base = str(tmp_path / "dp")
    mp.spawn(_unequal_rank_main, args=(2, _free_port(), base), nprocs=2, join=True)
    r0, r1 = np.load(base + "_u0.npz"), np.load(base + "_u1.npz")
    assert np.array_equal(r0["params"], r1["params"]) and np.array_equal(r0["ph"], r1["ph"])
    assert np.load(base + "_ubad0.npy")[0] == 1 and np.load(base + "_ubad1.npy")[0] == 1
    # oracle replay: steps over the union of the ranks' consecutive slices
    p = ppo.HipPolicy(72, HID, 2, 4, seed=3).params.copy()
    m, v, bp = np.zeros_like(p), np.zeros_like(p), np.array([0.9, 0.999])
    flat = []
    for r in (r0, r1):
        flat.append(dict(st=r["st"].reshape(-1, 32, 72), act=r["act"].reshape(-1), a=(r["a"].reshape(-1) - 1).astype(np.int32),
                         p=r["p"].reshape(-1), r=r["r"].reshape(-1)))
    lp_hist = []
    for b in range(3):
        sel = [slice(40 * b, min(40 * (b + 1), len(f["a"]))) for f in flat]
        cat = lambda k: np.concatenate([f[k][s] for f, s in zip(flat, sel)])
        g, lp, le = orc.step_batch_grad_f64(p, 72, HID, cat("st"), cat("act"), cat("a"), cat("p"), cat("r"), 0.05, 0.01)
        orc.adam_step(p, g.astype(np.float32), m, v, bp, 1e-3)
        lp_hist.append(lp)
    assert abs(np.mean(lp_hist) - r0["ph"][0]) <= 1e-5 * (1 + abs(np.mean(lp_hist)))
    assert np.abs(p - r0["params"]).max() <= 2e-5, "parameters after three union-batch steps vs the oracle loop"


def test_bench_two_ranks_rehearsal_on_one_gpu(P):
    """bench.py --gpus 2 exactly as the driver calls it (no launcher environment): two rank processes, here sharing the
    one GPU with the exchange over gloo; the line must report two ranks and carry the strong-scaling leg."""
    from test_host_logic import _run_bench
    r, out = _run_bench({"PPO_BENCH_BACKEND": "gloo", "PPO_BENCH_SHARE_GPU": "1"}, "--gpus", "2", "--steps", "1",
                        "--warmup", "1", "--envs", "64", "--t-steps", "8", "--epochs", "1", "--hid", "128",
                        "--no-cpu-baseline", timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["value"] > 0
    assert out["allreduce"] == "torch.distributed/gloo"
    assert out["strong"]["envs_per_gpu"] == 32 and out["strong"]["value"] > 0


def test_bench_stream_flag(P, tmp_path):
    """bench.py --stream DIR: the timed loop collects through a fresh DiskRollouts per iteration (BASELINE config 5 wording);
    the line says so, is flagged non-headline, and the last iteration's rollout.bin is on disk with the compact record."""
    from test_host_logic import _run_bench
    d = str(tmp_path / "stream")
    r, out = _run_bench({}, "--steps", "2", "--warmup", "1", "--envs", "96", "--t-steps", "8", "--epochs", "1",
                        "--no-cpu-baseline", "--stream", d, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert out["config"]["rollouts_streamed_to_disk"] is True and out["headline_config"] is False and out["value"] > 0
    f = os.path.join(d, "rank0", "rollout.bin")
    assert os.path.isfile(f)
    hdr = open(f, "rb").read(40)
    assert hdr[:4] == b"PPOR" and int.from_bytes(hdr[4:8], "little") == 2          # version 2: env snapshots
    assert int.from_bytes(hdr[8:16], "little") == 96                                 # N
    assert os.path.getsize(f) == 40 + 16 + 8 * 96 * (64 + 17) + 8 * 96 * 4          # header + template tag + T records + returns column
