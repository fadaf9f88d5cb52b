"""The fused backward with its products on the bf16 matrix pipe as split-fp32 ("bf16x6") products
(csrc/ppo_policy_bwd_x6.hip, ppo_set_bwd_split_bf16) against the pure fp32-MFMA kernel and the float64 oracle
(Zygote's gradient restated, src/train.jl:65-79)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def P(ppo):
    if ppo.device_count() == 0:
        pytest.skip("no GPU")
    ppo.set_bwd_small_max_tiles(0)            # the fused backward at every minibatch size
    ppo.set_train_tile_max_tiles(0)
    yield ppo
    ppo.set_bwd_small_max_tiles(None)
    ppo.set_train_tile_max_tiles(None)
    ppo.set_bwd_split_bf16(None)
    ppo.set_rollout_compact(None)
    os.environ.pop("PPO_FWD_SPLIT_T2_MIN_TILES", None)
    os.environ.pop("PPO_FWD_SPLIT_T2_MIN_TILES_128", None)


def _dataset(P, N, T, HID, seed):
    env = P.HipVecEnv(num_envs=N, Q=8, max_actions=12, seed=seed)
    pol = P.HipPolicy(72, HID, 2, 4, seed=seed + 1)
    rng = np.random.default_rng(seed)
    pol.params = pol.params + (rng.normal(size=pol.num_params) * 0.02).astype(np.float32)
    ro = P.BufferRollouts()
    P.collect_rollouts_steps_(ro, env, pol, T, 1.0)
    return env, pol, ro, P.construct_dataset(ro)


def _off_the_kink(params, hid, states, delta=2e-6):
    """True per state when no hidden unit's pre-activation lies within `delta` of leakyrelu's kink: there fp32 and float64
    disagree about the sign (one unit in ~10^7), the derivative jumps 100x and a whole gradient row differs by O(1e-3 max|g|)
    between ANY two precisions -- see tests/test_gpu_deep_policy.py.  At 1700 states x 32 rows x 512 units such a unit is likely."""
    from oracle import np_oracle
    a = states.reshape(-1, 72).astype(np.float64).T
    ok = np.ones(states.shape[0], bool)
    for (W, b) in np_oracle.unpack_params(params, 72, hid, 2)[:-1]:
        z = W.astype(np.float64) @ a + b.astype(np.float64)[:, None]
        ok &= (np.abs(z).min(axis=0).reshape(states.shape[0], 32).min(axis=1) >= delta)
        a = np.where(z > 0, z, 0.01 * z)
    return ok


def _oracle_grad(orc, params, HID, ro, sel0, eps, ew):
    st, act = ro.state_data
    st = st.reshape(-1, 32, 72)[sel0]
    act = act.reshape(-1)[sel0]
    a0 = (ro.selected_actions.reshape(-1)[sel0] - 1).astype(np.int32)
    return orc.step_batch_grad_f64(params, 72, HID, st, act, a0, ro.selected_action_probabilities.reshape(-1)[sel0],
                                   ro.rewards.reshape(-1)[sel0], eps, ew)


@pytest.mark.parametrize("HID,B,compact", [(256, 401, True), (256, 600, False), (256, 333, True), (128, 1100, False), (128, 70, True)])
def test_split_backward_matches_fp32_kernel_and_f64(P, orc, HID, B, compact):
    """Same minibatch through both kernels: each within 2e-5 max|g| of the float64 gradient (the bar of every gradient
    test), the split form no further from float64 than a small multiple of the fp32 chain's own distance, the two within
    fp32 rounding of each other, and the split form bitwise reproducible.  More tiles than workgroups (HID = 256: 1701 on
    256) and fewer (70 on 512) both occur; from 1536 tiles on the HID = 256 train forward takes two tiles per workgroup pass,
    below it one: the B = 401 case moves that switch to 200 tiles (PPO_FWD_SPLIT_T2_MIN_TILES, read per launch) -- an odd count,
    the last pass has one tile."""
    P.set_rollout_compact(compact)
    if B == 401:
        os.environ["PPO_FWD_SPLIT_T2_MIN_TILES"] = "200"
    if (HID, B) == (128, 70):
        os.environ["PPO_FWD_SPLIT_T2_MIN_TILES_128"] = "32"       # (HID = 128 switches at 1024 tiles; B = 1100 takes it by itself)
    env, pol, ro, ds = _dataset(P, 48, 40 if B > 1152 else 24, HID, seed=B)
    pool = np.flatnonzero(_off_the_kink(pol.params, HID, ro.state_data[0].reshape(-1, 32, 72)))
    assert len(pool) >= 64                                           # (a minibatch may repeat samples)
    sel = pool[np.random.default_rng(B).choice(len(pool), size=B, replace=B > len(pool))] + 1
    g64, olp, ole = _oracle_grad(orc, pol.params, HID, ro, sel - 1, 0.05, 0.01)
    scale = np.abs(g64).max()
    out = {}
    for mode in (0, 1):
        P.set_bwd_split_bf16(mode)
        lp, le = P.forward_backward(pol, ds, sel, 0.05, 0.01)
        g = pol.grad()
        assert abs(lp - olp) <= 1e-5 * (1 + abs(olp)) and abs(le - ole) <= 1e-5 * (1 + abs(ole))
        out[mode] = g
        if mode == 1:
            P.forward_backward(pol, ds, sel, 0.05, 0.01)
            assert np.array_equal(g, pol.grad())
    e0 = np.abs(out[0] - g64).max() / scale
    e1 = np.abs(out[1] - g64).max() / scale
    d01 = np.abs(out[0] - out[1]).max() / scale
    n0 = np.linalg.norm(out[0] - g64) / np.linalg.norm(g64)
    n1 = np.linalg.norm(out[1] - g64) / np.linalg.norm(g64)
    rec = dict(HID=HID, B=B, compact=compact, err_fp32_mfma=e0, err_split=e1, diff=d01, l2_fp32_mfma=n0, l2_split=n1)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "split_backward_accuracy.jsonl"), "a") as f:
        f.write(json.dumps(rec) + "\n")
    assert e0 <= 2e-5 and e1 <= 2e-5, rec
    assert e1 <= 4 * e0 + 1e-6, rec
    assert d01 <= 4e-6, rec


def test_split_backward_trains_like_fp32_kernel(P):
    """Three ppo_train! epochs with each backward: the parameters stay within fp32-rounding distance (Adam's normalised
    step amplifies a gradient difference of 1e-7 max|g| to at most a few 1e-4 of a step of size eta)."""
    res = {}
    for mode in (0, 1):
        P.set_bwd_split_bf16(mode)
        env = P.HipVecEnv(num_envs=64, Q=8, max_actions=16, seed=5)
        pol = P.HipPolicy(72, 256, 2, 4, seed=6)
        opt = P.Optimiser(P.Adam(1e-3))
        ro = P.BufferRollouts()
        P.collect_rollouts_steps_(ro, env, pol, 16, 1.0)
        P.ppo_train_(pol, opt, P.construct_dataset(ro), 0.05, 256, 3, 0.01, seed=3, verbose=False)
        res[mode] = pol.params.copy()
    step = np.abs(res[0] - res[1]).max()
    assert step <= 3e-4, step        # 9-12 optimiser steps of at most 1e-3 each: the two runs stay a fraction of ONE step apart
