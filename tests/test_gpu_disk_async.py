"""Deferred finish of a streamed collection (ppo_set_disk_async): the file completes in the writer thread while the caller
trains; after disk_sync it is byte for byte the file the synchronous path writes (src/rollouts_to_disk.jl: the reference
writes synchronously, the content is what matters)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def P(ppo):
    if ppo.device_count() == 0:
        pytest.skip("no GPU")
    yield ppo
    ppo.set_disk_async(None)


@pytest.mark.parametrize("slots", [2, 16])
def test_deferred_finish_writes_the_same_file(P, tmp_path, slots):
    N, T = 96, 20
    blobs = {}
    for mode in (False, True):
        P.set_disk_async(mode)
        env = P.HipVecEnv(num_envs=N, Q=8, max_actions=9, seed=21)
        pol = P.HipPolicy(72, 128, 2, 4, seed=4)
        disk = P.DiskRollouts(str(tmp_path / ("store_%d" % mode)))
        P.collect_rollouts_steps_(disk, env, pol, T, 0.99, pinned_slots=slots)
        ds = P.construct_dataset(disk._device)                 # the columns stayed in HBM: training does not wait for the file
        opt = P.Optimiser(P.Adam(1e-3))
        P.ppo_train_(pol, opt, ds, 0.05, 128, 1, 0.01, seed=1, verbose=False)
        P.disk_sync(disk)
        blobs[mode] = (open(os.path.join(disk.state_data_directory, "rollout.bin"), "rb").read(), pol.params.copy())
        # a second collection into a fresh store right away (the reference builds a DiskRollouts per iteration)
        disk2 = P.DiskRollouts(str(tmp_path / ("store2_%d" % mode)))
        P.collect_rollouts_steps_(disk2, env, pol, T, 0.99, pinned_slots=slots)
        P.disk_sync(disk2)
        ro = P.load_disk_rollouts(disk2.state_data_directory, env)
        assert np.array_equal(ro.selected_actions, disk2._device.selected_actions)
        assert np.array_equal(ro.rewards, disk2._device.rewards)
    assert blobs[False][0] == blobs[True][0]
    assert np.array_equal(blobs[False][1], blobs[True][1])
    assert len(blobs[True][0]) == 40 + 16 + T * (N * 64 + N * 17) + T * N * 4
