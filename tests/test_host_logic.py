"""CPU tests of the host mirror (no GPU): plugin stubs, index helpers, data-parallel sharding and the
gradient all-reduce contract rehearsed with world_size-2 gloo (the oracle supplies the per-rank numbers)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plugin_stubs_raise_like_the_reference(ppo):
    class Foo:
        pass
    for fn, name in [(ppo.state, "state"), (ppo.reward, "reward"), (ppo.is_terminal, "is_terminal"),
                     (ppo.reset_, "reset!"), (ppo.step_, "step!"), (ppo.action_probabilities, "action_probabilities"),
                     (ppo.batch_action_probabilities, "batch_action_probabilities"), (ppo.batch_state, "batch_state"),
                     (ppo.number_of_actions_per_state, "number_of_actions_per_state"),
                     (ppo.batch_advantage, "batch_advantage"), (ppo.save_loss, "save_loss")]:
        with pytest.raises(ppo.PPOError, match="Function %s needs to be overloaded" % name.replace("!", "!")):
            fn(Foo())


def test_index_helpers(ppo, orc):
    for idx in range(1, 129):
        assert ppo.index_to_action(idx) == orc.index_to_action(idx)
    assert np.array_equal(ppo.action_mask([1, 1, 0, 0]), orc.action_mask([1, 1, 0, 0]))
    assert ppo.simplified_ppo_clip(2.0, 0.05) == orc.simplified_ppo_clip(2.0, 0.05)
    assert ppo.simplified_ppo_clip(-2.0, 0.05) == orc.simplified_ppo_clip(-2.0, 0.05)


def test_state_containers(ppo, orc):
    s1 = ppo.StateData(np.zeros((32, 72), np.int8), np.uint32(0x3F))
    s2 = ppo.StateData(np.ones((32, 72), np.int8), np.uint32(0x0F))
    b = ppo.batch_state([s1, s2])
    assert b.vertex_score.shape == (2, 32, 72) and b.action_mask.tolist() == [0x3F, 0x0F]
    assert ppo.number_of_actions_per_state(b) == 128
    m = b.mask_vector()
    assert np.array_equal(m[0], orc.action_mask([1, 1, 1, 1, 1, 1, 0, 0]))
    assert np.array_equal(m[1], orc.action_mask([1, 1, 1, 1, 0, 0, 0, 0]))
    assert np.array_equal(s1.mask_vector(), m[0])
    r = np.array([1.0, 2.0], np.float32)
    assert np.array_equal(ppo.batch_advantage(b, r), r)


def test_optimiser_composite(ppo):
    opt = ppo.Optimiser(ppo.Adam(1e-4))
    assert ppo.get_optimizer_learning_rate(opt) == 1e-4
    with pytest.raises(TypeError):
        ppo.get_optimizer_learning_rate(ppo.Adam(1e-4))


def test_env_shards_cover_everything(ppo):
    for world in (1, 2, 3, 8):
        for total in (8, 4096, 4099):
            spans = [ppo.DataParallel(r, world).env_shard(total) for r in range(world)]
            assert spans[0][0] == 0 and sum(n for _, n in spans) == total
            for (o1, n1), (o2, _) in zip(spans, spans[1:]):
                assert o1 + n1 == o2


def test_sharded_envs_match_unsharded(orc):
    """Counter RNG keyed by the GLOBAL env id: two shards of 4 envs reproduce the 8-env run column by column."""
    params = orc.glorot_params(72, 128, 2, seed=1)
    full = orc.Env(Q=8, max_actions=10, N=8, seed=3)
    full.reset()
    ref = orc.collect_rollouts_tn(full, params, 128, 12)
    for off in (0, 4):
        e = orc.Env(Q=8, max_actions=10, N=4, seed=3, global_offset=off)
        e.reset()
        part = orc.collect_rollouts_tn(e, params, 128, 12)
        for k in ("actions", "p_sel", "rewards", "done", "states"):
            assert np.array_equal(part[k], ref[k][:, off:off + 4]), k


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, out):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch
    import torch.distributed as dist
    import ppo_amd
    from oracle import oracle as orc
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    d = np.load(out + "_in.npz")
    B = d["states"].shape[0]
    dp = ppo_amd.DataParallel(rank, world)
    off, n = dp.env_shard(B)                       # here: shard the minibatch like the env batch
    sl = slice(off, off + n)
    g, lp, le = orc.step_batch_grad_f64(d["params"], 72, 128, d["states"][sl], d["active"][sl], d["actions"][sl],
                                        d["p_old"][sl], d["adv"][sl], 0.05, 0.01)
    # engine contract: each rank's buffer holds (grad, ppo, entropy) already scaled by 1/B_global
    buf = np.concatenate([g, [lp, le]]) * (n / B)
    t = torch.from_numpy(buf)
    dp.allreduce_(t)                               # one sum all-reduce per optimiser step
    if rank == 0:
        np.save(out + "_out.npy", t.numpy())
    dist.destroy_process_group()


def test_gloo_world2_gradient_allreduce(orc, tmp_path):
    import torch.multiprocessing as mp
    rng = np.random.default_rng(0)
    B = 10
    params = orc.glorot_params(72, 128, 2, seed=2)
    states = rng.integers(-3, 6, size=(B, 32, 72)).astype(np.int8)
    active = np.full(B, 0x3F, np.uint32)
    actions = rng.integers(0, 96, B).astype(np.int32)
    p_old = rng.uniform(0.005, 0.02, B).astype(np.float32)
    adv = rng.normal(size=B).astype(np.float32)
    base = str(tmp_path / "dp")
    np.savez(base + "_in.npz", params=params, states=states, active=active, actions=actions, p_old=p_old, adv=adv)
    port = _free_port()
    mp.spawn(_rank_main, args=(2, port, base), nprocs=2, join=True)
    got = np.load(base + "_out.npy")
    g, lp, le = orc.step_batch_grad_f64(params, 72, 128, states, active, actions, p_old, adv, 0.05, 0.01)
    want = np.concatenate([g, [lp, le]])
    assert np.allclose(got, want, rtol=1e-10, atol=1e-12)


def _run_bench(extra_env, *argv, timeout=300):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **extra_env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + list(argv), env=env, capture_output=True,
                       text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_gpus2_starts_two_ranks_by_itself():
    """`python bench.py --gpus 2` with no launcher environment must start two rank processes (the driver calls it
    exactly like that).  Dry run over gloo: rank launch + rendezvous + the one JSON line, no GPU."""
    r, out = _run_bench({"PPO_BENCH_DRYRUN": "1", "PPO_BENCH_BACKEND": "gloo"}, "--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    assert out is not None and out["n_gpus"] == 2 and out["dry_run"] is True
    assert sum(1 for ln in r.stdout.splitlines() if ln.startswith("{")) == 1, "exactly one JSON line"


def test_bench_multi_gpu_never_reports_a_single_rank_silently():
    """Without a GPU the rank processes fail: the launcher must exit non-zero and print no figure (round 1 printed a
    1-GPU number under --gpus N)."""
    if __import__("ppo_amd").device_count() > 0:
        pytest.skip("a GPU is present")
    r, out = _run_bench({"PPO_BENCH_BACKEND": "gloo"}, "--gpus", "2", "--steps", "1", "--warmup", "0", "--envs", "64",
                        "--no-cpu-baseline")
    assert r.returncode != 0 and out is None


def test_julia_reference_driver_is_probed_not_assumed(monkeypatch):
    """bench/julia_reference.jl cannot run here (no julia): static checks only.  It must
    include the reference's own module from a checkout given at run time (never a copy), drive ITS collect_rollouts! /
    ppo_train!, and overload exactly the plugin functions the reference declares; bench.py must leave the key out when
    `julia` or the checkout is missing."""
    import re
    src = open(os.path.join(ROOT, "bench", "julia_reference.jl")).read()
    code = re.sub(r"#[^\n]*", "", src)
    assert 'include(joinpath(REF, "src", "ProximalPolicyOptimization.jl"))' in code
    assert "PPO.collect_rollouts!(rollouts, bank, policy" in code and "PPO.ppo_train!(policy, optimizer, dataset" in code
    declared = {"state", "reward", "is_terminal", "reset!", "step!", "action_probabilities", "batch_action_probabilities",
                "batch_state", "number_of_actions_per_state", "batch_advantage", "save_loss"}      # src/ProximalPolicyOptimization.jl:16-30
    overloaded = set(re.findall(r"(?:function\s+)?PPO\.([\w!]+)\(", code)) - {"collect_rollouts!", "ppo_train!", "construct_dataset", "BufferRollouts"}
    assert overloaded <= declared and {"state", "reward", "is_terminal", "reset!", "step!", "action_probabilities",
                                       "batch_action_probabilities", "batch_state", "number_of_actions_per_state",
                                       "batch_advantage"} <= overloaded, overloaded
    assert code.count("(") == code.count(")") and code.count("[") == code.count("]")
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setenv("PPO_JULIA_REFERENCE", "/nonexistent")
    assert bench.cpu_baseline_julia() is None
    monkeypatch.setenv("PATH", "/nonexistent-bin")
    monkeypatch.setenv("PPO_JULIA_REFERENCE", ROOT)
    assert bench.cpu_baseline_julia() is None
