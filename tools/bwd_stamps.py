#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of k_policy_bwd (needs libppo_hip_stamp.so built with -DPPO_BWD_STAMP).
Shares only -- never quote this build's run time (stamps cost cycles)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ppo_amd as PPO
PPO._lib.SO_PATH = os.path.join(ROOT, "proximalpolicyoptimization.jl_amd", "libppo_hip_stamp.so")
PPO._lib._lib = None
L = PPO._lib.lib()
env = PPO.HipVecEnv(num_envs=4096, Q=8, max_actions=128, seed=1)
pol = PPO.HipPolicy(72, 256, 2, 4, seed=0)
ro = PPO.BufferRollouts()
PPO.collect_rollouts_steps_(ro, env, pol, 4, 1.0)
ds = PPO.construct_dataset(ro)
sel = np.arange(1, 4097)
for _ in range(3):
    PPO.forward_backward(pol, ds, sel, 0.05, 0.01)
out = np.zeros(256 * 20, np.uint64)
L.ppo_debug_bwd_stamps.argtypes = [C.c_void_p]
assert L.ppo_debug_bwd_stamps(out.ctypes.data) == 0
s = out.reshape(256, 2, 10).astype(np.float64)
names = ["A transform+LDS", "barrier1", "B dX", "barrier2", "D2 late tail", "barrier3", "A wait loads", "C dW2 (+tail)", "D0 issue next loads", "D1 dW1 mfma loop"]
for wv in (0, 1):
    m = s[:, wv, :].mean(axis=0)
    print("wave %s: total %.0f cycles/WG (%.0f per tile)" % ("0" if wv == 0 else "last", m.sum(), m.sum() / 16))
    for n, v in zip(names, m):
        print("   %-14s %8.0f per tile  %5.1f %%" % (n, v / 16, 100 * v / m.sum()))
