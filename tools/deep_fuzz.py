#!/usr/bin/env python3
"""One-off fuzz of the any-depth policy path (GPU box): random (num_hidden_layers 1..4, hidden 1..256, minibatch 1..1500,
storage form, all three kernel sets at L = 2) from real rollouts of the built-in env; rollout bit-exact against the oracle
(hidden % 32 == 0), gradient against the float64 oracle (2e-5 max|g|, states off the leakyrelu kink), reproducibility,
Adam bit-exact.  Not part of the suite (minutes of CPU oracle time)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ppo_amd as P
from oracle import oracle as orc
import importlib.util
spec = importlib.util.spec_from_file_location("t", os.path.join(ROOT, "tests", "test_gpu_deep_policy.py")); t = importlib.util.module_from_spec(spec); spec.loader.exec_module(t)

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 24):
    L = int(rng.integers(1, 5))
    hid = int(rng.choice([32, 64, 96, 128, 160, 192, 224, 256, int(rng.integers(1, 257))]))
    B = int(rng.integers(1, 1500))
    compact = bool(rng.integers(2))
    form = int(rng.integers(3)) if L == 2 else 0
    P.set_rollout_compact(compact)
    if form == 2: P.set_train_tile_max_tiles(1 << 20)
    if form == 1: P.set_bwd_small_max_tiles(1 << 20); P.set_fwd_split_max_states(512)
    try:
        N, T = 64, 28
        env = P.HipVecEnv(num_envs=N, Q=8, max_actions=10, seed=100 + trial)
        pol = P.HipPolicy(72, hid, L, 4, seed=200 + trial)
        p0 = (pol.params + (rng.normal(size=pol.num_params) * 0.02).astype(np.float32)).astype(np.float32)
        pol.params = p0
        ro = P.BufferRollouts()
        P.collect_rollouts_steps_(ro, env, pol, T, 1.0)
        st, act = ro.state_data
        bitexact = None
        if hid % 32 == 0:
            oenv = orc.Env(Q=8, max_actions=10, N=N, seed=100 + trial); oenv.reset()
            ref = orc.collect_rollouts_tn(oenv, p0, hid, T, mode_dev=True, n_hidden=L)
            bitexact = bool(np.array_equal(ro.selected_actions - 1, ref["actions"]) and np.array_equal(ro.selected_action_probabilities, ref["p_sel"]))
        ds = P.construct_dataset(ro)
        keep = np.flatnonzero(t._off_the_kink(p0, 72, hid, L, st.reshape(-1, 32, 72)))
        sel = rng.choice(keep, size=B)
        P.forward_backward(pol, ds, sel + 1, 0.05, 0.01)
        g = pol.grad()
        g64, _, _ = orc.step_batch_grad_f64(p0, 72, hid, st.reshape(-1, 32, 72)[sel], act.reshape(-1)[sel],
                                            (ro.selected_actions.reshape(-1)[sel] - 1).astype(np.int32),
                                            ro.selected_action_probabilities.reshape(-1)[sel], ro.rewards.reshape(-1)[sel], 0.05, 0.01, n_hidden=L)
        e = float(np.abs(g - g64).max() / np.abs(g64).max())
        P.forward_backward(pol, ds, sel + 1, 0.05, 0.01)
        rep = bool(np.array_equal(g, pol.grad()))
        opt = P.Optimiser(P.Adam(1e-3))
        P.step_batch_(pol, opt, ds, sel + 1, 0.05, 0.01)
        pp, mm, vv, bb = p0.copy(), np.zeros_like(p0), np.zeros_like(p0), np.array([0.9, 0.999])
        orc.adam_step(pp, pol.grad(), mm, vv, bb, 1e-3)
        adam = bool(np.array_equal(pol.params, pp))
        ok = e <= 2e-5 and rep and adam and bitexact is not False
        bad += not ok
        print("trial %2d L %d hid %3d B %4d compact %d kernels %d: grad %.1e  reproducible %s  adam-exact %s  rollout-bitexact %s  %s"
              % (trial, L, hid, B, compact, form, e, rep, adam, bitexact, "ok" if ok else "FAIL"), flush=True)
    finally:
        P.set_rollout_compact(None); P.set_train_tile_max_tiles(None); P.set_bwd_small_max_tiles(None); P.set_fwd_split_max_states(None)
print("FUZZ", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
