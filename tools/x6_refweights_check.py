#!/usr/bin/env python3
"""Diagnostic: gradient error vs the float64 checker on the reference-trained weights, every combination of train-forward
and backward form (fp32 MFMA / split-fp32)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ppo_amd as P
from oracle import oracle as orc
orc.build()
P.set_bwd_small_max_tiles(0); P.set_train_tile_max_tiles(0)
for fixture in ("poly-30-policy", "catmull-clark-policy"):
    params = np.load(os.path.join(ROOT, "tests", "golden", fixture + ".npz"))["params"]
    env = P.HipVecEnv(num_envs=32, Q=8, max_actions=12, seed=17)
    pol = P.HipPolicy(72, 128, 2, 4, seed=0)
    pol.params = params
    ro = P.BufferRollouts()
    P.collect_rollouts_steps_(ro, env, pol, 12, 1.0)
    ds = P.construct_dataset(ro)
    sel = np.random.default_rng(2).permutation(len(ds))[:200] + 1
    st, act = ro.state_data
    sel0 = sel - 1
    for eps, ew in ((0.05, 0.01), (0.2, 0.0)):
        g64, olp, ole = orc.step_batch_grad_f64(params, 72, 128, st.reshape(-1, 32, 72)[sel0], act.reshape(-1)[sel0],
                                                (ro.selected_actions.reshape(-1)[sel0] - 1).astype(np.int32),
                                                ro.selected_action_probabilities.reshape(-1)[sel0], ro.rewards.reshape(-1)[sel0], eps, ew)
        scale = np.abs(g64).max()
        for mode in (0, 1):
            P.set_bwd_split_bf16(mode)
            lp, le = P.forward_backward(pol, ds, sel, eps, ew)
            g = pol.grad()
            i = int(np.abs(g - g64).argmax())
            print(fixture, "eps", eps, "ew", ew, "split" if mode else "fp32 ", "max err / max|g| %.3e" % (np.abs(g - g64).max() / scale),
                  "l2 %.3e" % (np.linalg.norm(g - g64) / np.linalg.norm(g64)), "at", i, "g64 %.6e g %.6e" % (g64[i], g[i]),
                  "loss err %.2e %.2e" % (abs(lp - olp), abs(le - ole)), flush=True)
