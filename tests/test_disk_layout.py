"""Disk rollout store: the reference's on-disk layout (src/rollouts_to_disk.jl, src/dataset.jl).  CPU part: the
reference's own four @tests (test/test_rollout_to_disk.jl:13-23) and its BSON fixture byte for byte."""
import os

import numpy as np
import pytest


def test_bson_writer_matches_reference_fixture(ppo, golden_dir):
    raw = open(os.path.join(golden_dir, "sample_1.bson"), "rb").read()
    mine = ppo.bson_encode_state(np.array([1, 2, 3, 4, 5], np.int64))
    assert mine == raw                                   # 183 bytes, identical to BSON.@save of Int64[1,2,3,4,5]
    assert np.array_equal(ppo.bson_decode_state(raw), [1, 2, 3, 4, 5])


def test_reference_disk_tests(ppo, tmp_path):
    """test/test_rollout_to_disk.jl:13-23 restated: constructing the store wipes the directory, creates states/,
    update! writes states/sample_1.bson whose :state round-trips to [1,2,3,4,5]."""
    d = tmp_path / "rollout_to_disk"
    d.mkdir()
    f = d / "test.txt"
    f.write_text("hello")
    trajectory = ppo.DiskRollouts(str(d))
    assert not f.exists()                                                        # @test !isfile(file)
    assert (d / "states").is_dir()                                               # @test isdir(joinpath(dir, "states"))
    ppo.update_(trajectory, np.array([1, 2, 3, 4, 5], np.int64), 0.2, 1, 0.5, False)
    fp = d / "states" / "sample_1.bson"
    assert fp.is_file()                                                          # @test isfile(file_path)
    state = ppo.bson_decode_state(fp.read_bytes())
    assert np.array_equal(state, [1, 2, 3, 4, 5])                                # @test allequal(state, [1,2,3,4,5])
    assert len(trajectory) == 1
    rows = (d / "trajectory.csv").read_text().splitlines()
    assert rows[0] == "sample_names,selected_actions,selected_action_probabilities,rewards,terminal"
    assert rows[1] == "sample_1.bson,1,0.2,0.5,false"
    with pytest.raises(ppo.PPOError):
        ppo.update_(trajectory, np.zeros(3, np.int64), 1.5, 1, 0.0, False)       # @assert 0 <= p <= 1
    with pytest.raises(ppo.PPOError):
        ppo.update_(trajectory, np.zeros(3, np.int64), 0.5, 1, 0.0, 1)           # @assert terminal isa Bool


def test_state_data_roundtrip_and_dataset(ppo, tmp_path):
    d = ppo.DiskRollouts(str(tmp_path / "ds"))
    rng = np.random.default_rng(0)
    states = []
    for k in range(5):
        s = ppo.StateData(rng.integers(-3, 5, size=(32, 72)).astype(np.int8), np.uint32(0x3F >> (k % 2)))
        states.append(s)
        ppo.update_(d, s, 0.25, k + 1, 1.0, k == 4)
    # finish like write_returns_to_disk would, without the GPU: returns column by hand
    rows = open(d.trajectory_filename).read().splitlines()
    with open(d.trajectory_filename, "w") as f:
        f.write(",".join(ppo.DiskRollouts.FINAL) + "\n")
        for i, r in enumerate(rows[1:]):
            c = r.split(",")
            f.write("%s,%s,%s,%s\n" % (c[0], c[1], c[2], float(5 - i)))
    ds = ppo.DiskDataset(d.state_data_directory)
    assert len(ds) == 5
    s3 = ds[3]
    assert s3["selected_action"] == 3 and s3["returns"] == 3.0 and s3["selected_action_probability"] == 0.25
    assert np.array_equal(s3["state"].vertex_score, states[2].vertex_score)
    assert int(s3["state"].action_mask) == int(states[2].action_mask)
    b = ds[[1, 5]]
    assert b["state"].vertex_score.shape == (2, 32, 72) and b["returns"].tolist() == [5.0, 1.0]
    with pytest.raises(ppo.PPOError):
        ds[0]
    with pytest.raises(ppo.PPOError):
        ppo.DiskDataset(str(tmp_path / "missing"))
