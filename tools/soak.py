#!/usr/bin/env python3
"""Soak: the bench iteration for the env shards of ranks 0..7 (global env offsets r*4096) and a few seeds, on one GPU.
Rare-event check (device flags, non-finite parameters) before the 8-GPU scaling run that cannot be rehearsed here."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ppo_amd as PPO

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6
bad = 0
for dtype in ("f32", "bf16"):
    for rank in range(8):
        env = PPO.HipVecEnv(num_envs=4096, Q=8, max_actions=128, seed=1234, global_offset=rank * 4096)
        pol = PPO.HipPolicy(72, 256, 2, 4, seed=0, dtype=dtype)
        opt = PPO.Optimiser(PPO.Adam(1e-4))
        for it in range(iters):
            ro = PPO.BufferRollouts()
            PPO.collect_rollouts_steps_(ro, env, pol, 128, 1.0)
            ds = PPO.construct_dataset(ro)
            PPO.ppo_train_(pol, opt, ds, 0.05, 4096, 4, 0.01, seed=1000 + it, verbose=False)
        ok = bool(np.all(np.isfinite(pol.params)))
        fl = env.error_flags()
        print(dtype, "rank", rank, "flags", fl, "finite", ok, flush=True)
        bad += (not ok) or bool(fl & ~32)
print("SOAK", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
