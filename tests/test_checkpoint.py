"""Flux/BSON checkpoint interop: the reference's own `BSON.@save ... policy` file is decoded into
the flat Flux.params vector and re-encoded byte for byte; GPU part: a reference-trained checkpoint runs on the HIP
engine and a HipPolicy survives save -> load."""
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def ck(ppo):
    return ppo.checkpoint


def test_decode_reference_checkpoint(ck, golden_dir):
    raw = open(os.path.join(golden_dir, "poly-30-policy.bson"), "rb").read()
    params, fin, hid, nhl, out = ck.decode_policy(raw)
    assert (fin, hid, nhl, out) == (72, 128, 2, 4)
    want = np.load(os.path.join(golden_dir, "poly-30-policy.npz"))["params"]     # decoded independently (pymongo bson)
    assert params.dtype == np.float32 and np.array_equal(params, want)


def test_encode_reproduces_reference_bytes(ck, golden_dir):
    raw = open(os.path.join(golden_dir, "poly-30-policy.bson"), "rb").read()
    params, fin, hid, nhl, out = ck.decode_policy(raw)
    assert ck.encode_policy(params, fin, hid, nhl, out) == raw, "BSON.@save document must match BSON.jl's bytes"


@pytest.mark.parametrize("fin,hid", [(72, 256), (216, 128), (8, 32)])
def test_roundtrip_random(ck, fin, hid):
    rng = np.random.default_rng(fin + hid)
    n = sum(o * i + o for (o, i) in ck.layer_dims(fin, hid, 2, 4))
    p = rng.normal(size=n).astype(np.float32)
    got = ck.decode_policy(ck.encode_policy(p, fin, hid, 2, 4))
    assert np.array_equal(got[0], p) and got[1:] == (fin, hid, 2, 4)


def test_rejects_foreign_documents(ck, golden_dir, ppo):
    with pytest.raises(ValueError):
        ck.decode_policy(open(os.path.join(golden_dir, "sample_1.bson"), "rb").read())      # a state file, not a policy
    with pytest.raises(ValueError):
        ck.encode_policy(np.zeros(10, np.float32), 72, 128, 2, 4)                            # wrong parameter count
    with pytest.raises(ValueError):
        ck.policy_document(np.zeros(1, np.float32), 72, 128, 3, 4)                           # only L = 2 documents
    doc = ck.policy_document(np.zeros(72 * 128 + 128 + 128 * 128 + 128 + 4 * 128 + 4, np.float32), 72, 128, 2, 4)
    doc["policy"]["data"][0]["data"][0]["data"][0]["data"][2]["type"]["name"] = ["NNlib", "#relu"]
    with pytest.raises(ValueError):
        ck.decode_policy(ppo.disk._enc_doc(doc))


def test_load_policy_with_host_class(ck, golden_dir):
    class Host:
        def __init__(self, fin, hid, nhl, out):
            self.in_channels, self.hidden_channels, self.num_hidden_layers, self.num_output = fin, hid, nhl, out
            self.params = None
    pol = ck.load_policy(os.path.join(golden_dir, "poly-30-policy.bson"), Host)
    assert pol.hidden_channels == 128 and pol.params.size == 26372


@pytest.mark.gpu
def test_reference_checkpoint_runs_on_the_engine(ppo, orc, golden_dir, tmp_path):
    pol = ppo.load_policy(os.path.join(golden_dir, "poly-30-policy.bson"))
    assert (pol.in_channels, pol.hidden_channels) == (72, 128)
    rng = np.random.default_rng(0)
    states = rng.integers(-3, 7, size=(4, 32, 72)).astype(np.int8)
    active = np.array([255, 63, 1, 129], np.uint32)
    probs = ppo.batch_action_probabilities(pol, ppo.StateData(states, active)).T
    for b in range(4):
        assert np.array_equal(probs[b], orc.action_probabilities(pol.params, 72, 128, states[b], active[b], "dev"))
    # engine -> file -> engine, and the file is what BSON.jl would have written for these weights
    path = str(tmp_path / "policy.bson")
    ppo.save_policy(path, pol)
    assert open(path, "rb").read() == open(os.path.join(golden_dir, "poly-30-policy.bson"), "rb").read()
    again = ppo.load_policy(path, dtype="bf16")
    assert np.array_equal(again.params, pol.params) and again.dtype == "bf16"
