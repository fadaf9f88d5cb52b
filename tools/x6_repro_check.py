#!/usr/bin/env python3
"""Diagnostic: run-to-run bitwise reproducibility of the split-fp32 training pass (gradient, dY-dependent losses)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ppo_amd as P
P.set_bwd_small_max_tiles(0); P.set_train_tile_max_tiles(0)
for HID in (128, 256):
    env = P.HipVecEnv(num_envs=48, Q=8, max_actions=12, seed=5)
    pol = P.HipPolicy(72, HID, 2, 4, seed=6)
    ro = P.BufferRollouts()
    P.collect_rollouts_steps_(ro, env, pol, 40, 1.0)
    ds = P.construct_dataset(ro)
    for B, rep in ((1100, False), (1100, True), (1536, True), (600, True), (1900, True)):
        rng = np.random.default_rng(B)
        sel = (rng.choice(len(ds), size=B, replace=False) if not rep else rng.choice(400, size=B, replace=True)) + 1
        gs = []
        for k in range(4):
            lp, le = P.forward_backward(pol, ds, sel, 0.05, 0.01)
            gs.append((pol.grad().copy(), lp, le))
        bad = [k for k in range(1, 4) if not np.array_equal(gs[0][0], gs[k][0])]
        badl = [k for k in range(1, 4) if gs[0][1:] != gs[k][1:]]
        print("HID", HID, "B", B, "repeats" if rep else "unique ", "grad differs in runs", bad, "loss differs in runs", badl,
              ("max diff %.3e" % max(np.abs(gs[0][0] - gs[k][0]).max() for k in range(1, 4))) if bad else "", flush=True)
