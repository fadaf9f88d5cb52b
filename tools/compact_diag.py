#!/usr/bin/env python3
"""Diagnostic: gradient error of the fp32-MFMA training pass on compact rollouts by minibatch size."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ppo_amd as P
from oracle import oracle as orc
orc.build()
P.set_bwd_small_max_tiles(0); P.set_train_tile_max_tiles(0)
for compact in (True, False):
    P.set_rollout_compact(compact)
    env = P.HipVecEnv(num_envs=48, Q=8, max_actions=12, seed=5)
    pol = P.HipPolicy(72, 256, 2, 4, seed=6)
    rng = np.random.default_rng(5)
    pol.params = pol.params + (rng.normal(size=pol.num_params) * 0.02).astype(np.float32)
    ro = P.BufferRollouts()
    P.collect_rollouts_steps_(ro, env, pol, 40, 1.0)
    ds = P.construct_dataset(ro)
    st, act = ro.state_data
    for B in (300, 1024, 1025, 1100, 1536):
        sel = np.random.default_rng(B).choice(len(ds), size=B, replace=False) + 1
        sel0 = sel - 1
        g64, olp, ole = orc.step_batch_grad_f64(pol.params, 72, 256, st.reshape(-1, 32, 72)[sel0], act.reshape(-1)[sel0],
                                                (ro.selected_actions.reshape(-1)[sel0] - 1).astype(np.int32),
                                                ro.selected_action_probabilities.reshape(-1)[sel0], ro.rewards.reshape(-1)[sel0], 0.05, 0.01)
        for mode in (0, 1):
            P.set_bwd_split_bf16(mode)
            lp, le = P.forward_backward(pol, ds, sel, 0.05, 0.01)
            g = pol.grad()
            print("compact" if compact else "expanded", "B", B, "split" if mode else "fp32 ", "err %.3e" % (np.abs(g - g64).max() / np.abs(g64).max()),
                  "loss err %.2e" % abs(lp - olp), flush=True)
