"""One engine per host thread ("one stream per engine handle"): the device, the stream, the kernel timers, the
RCCL communicator and the error text are thread-local in libppo_hip.so, so one process can run several engines side by
side.  Two threads, each with its own engine on the test box's one GPU, run a full PPO iteration CONCURRENTLY (ctypes
releases the GIL inside every call); each must reproduce the single-threaded run bit for bit."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _iteration(P, seed, out, key):
    try:
        P._lib.call("ppo_device_init", 0)                     # this thread's engine: its own stream on device 0
        env = P.HipVecEnv(num_envs=96, Q=8, max_actions=12, seed=seed)
        pol = P.HipPolicy(72, 128, 2, 4, seed=seed)
        ro = P.BufferRollouts()
        opt = P.Optimiser(P.Adam(1e-3))
        P.profile_enable(True)
        for it in range(3):
            P.collect_rollouts_steps_(ro, env, pol, 16, 0.99)
            ds = P.construct_dataset(ro)
            P.ppo_train_(pol, opt, ds, 0.05, 256, 2, 0.01, seed=100 + it, verbose=False)
        ms, n = P.profile_get("k_reduce_adam")                 # single-rank training: Adam rides in the slab-reduction launch
        if n == 0:
            ms, n = P.profile_get("k_adam")
        P.profile_enable(False)
        out[key] = (pol.params.copy(), ro.selected_actions.copy(), ro.rewards.copy(), n)
    except Exception as e:                                    # surfaced by the asserting thread
        out[key] = e


def test_two_engines_in_one_process(ppo):
    P = ppo
    if P.device_count() < 1:
        pytest.fail("no HIP device: the gpu-marked tests must run on the GPU box")
    ref = {}
    _iteration(P, 5, ref, "a")
    _iteration(P, 9, ref, "b")
    out = {}
    ts = [threading.Thread(target=_iteration, args=(P, 5, out, "a")), threading.Thread(target=_iteration, args=(P, 9, out, "b"))]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    for k in ("a", "b"):
        assert not isinstance(out.get(k), Exception), out.get(k)
        assert not isinstance(ref[k], Exception), ref[k]
        for x, y in zip(out[k][:3], ref[k][:3]):
            assert np.array_equal(x, y), "engine %s differs from its single-threaded run" % k
        assert out[k][3] == ref[k][3] > 0, "each engine counts its own launches (thread-local timing tables)"
    # an error raised in one thread's engine does not leak into the other's error text
    with pytest.raises(P.PPOError):
        P.HipVecEnv(num_envs=4, Q=1)                           # this thread's last error: "env_create: bad sizes"
    assert "env_create" in P._lib.last_error()
    err = {}

    def bad():
        try:
            P.HipPolicy(100, 128, 2, 4)
        except P.PPOError as e:
            err["bad"] = str(e)

    t = threading.Thread(target=bad)
    t.start()
    t.join()
    assert "policy_create" in err["bad"]
    assert "env_create" in P._lib.last_error() and "policy_create" not in P._lib.last_error()
