#!/usr/bin/env python3
"""Diagnostic: the flow of tests/test_gpu_split_backward.py (fp32-MFMA pass first, then the split pass twice) over many seeds."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ppo_amd as P
P.set_bwd_small_max_tiles(0); P.set_train_tile_max_tiles(0)
nbad = 0
for trial in range(int(os.environ.get('X6_TRIALS', '24'))):
    HID = 128 if trial % 2 == 0 else 256
    B = [1100, 700, 1536, 300][trial % 4]
    env = P.HipVecEnv(num_envs=48, Q=8, max_actions=12, seed=100 + trial)
    pol = P.HipPolicy(72, HID, 2, 4, seed=trial)
    rng = np.random.default_rng(trial)
    pol.params = pol.params + (rng.normal(size=pol.num_params) * 0.02).astype(np.float32)
    ro = P.BufferRollouts()
    P.collect_rollouts_steps_(ro, env, pol, 40, 1.0)
    ds = P.construct_dataset(ro)
    sel = rng.choice(300, size=B, replace=True) + 1
    if os.environ.get("X6_FP32_FIRST", "1") == "1":
        P.set_bwd_split_bf16(0)
        P.forward_backward(pol, ds, sel, 0.05, 0.01)
    P.set_bwd_split_bf16(1)
    outs = []
    for k in range(3):
        lp, le = P.forward_backward(pol, ds, sel, 0.05, 0.01)
        outs.append((pol.grad().copy(), lp, le))
    bad = [k for k in (1, 2) if not np.array_equal(outs[0][0], outs[k][0]) or outs[0][1:] != outs[k][1:]]
    if bad:
        nbad += 1
        d = np.abs(outs[0][0] - outs[bad[0]][0])
        print("trial", trial, "HID", HID, "B", B, "differs in runs", bad, "max diff %.3e at %d of %d, count %d" % (d.max(), d.argmax(), d.size, (d > 0).sum()),
              "loss", outs[0][1:], outs[bad[0]][1:], "idx", np.flatnonzero(d > 0)[:24].tolist(), flush=True)
print("trials with a difference:", nbad)
