#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of k_policy_bwd_bf16 (needs libppo_hip_bstamp.so: make -C csrc bstamp).
Shares only -- never quote this build's run time (stamps cost cycles)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ppo_amd as PPO
PPO._lib.SO_PATH = os.path.join(ROOT, "proximalpolicyoptimization.jl_amd", "libppo_hip_bstamp.so")
PPO._lib._lib = None
L = PPO._lib.lib()
env = PPO.HipVecEnv(num_envs=4096, Q=8, max_actions=128, seed=1)
pol = PPO.HipPolicy(72, 256, 2, 4, seed=0, dtype="bf16")
ro = PPO.BufferRollouts()
PPO.collect_rollouts_steps_(ro, env, pol, 4, 1.0)
ds = PPO.construct_dataset(ro)
sel = np.arange(1, 4097)
for _ in range(3):
    PPO.forward_backward(pol, ds, sel, 0.05, 0.01)
out = np.zeros(256 * 4 * 8, np.uint64)
L.ppo_debug_bf16_stamps.argtypes = [C.c_void_p]
assert L.ppo_debug_bf16_stamps(out.ctypes.data) == 0
s = out.reshape(256, 4, 8).astype(np.float64)
names = ["A wait prefetched loads", "A convert/dZ2/images", "barrier 1", "B dH1 mfma chain", "B lrelu'/image/lgkm", "C dW2",
         "C dW3 (VALU)", "C emit + end barrier"]
for wv in (0, 3):
    m = s[:, wv, :].mean(axis=0)
    print("wave %d: total %.0f cycles/WG (%.0f per tile)" % (wv, m.sum(), m.sum() / 16))
    for n, v in zip(names, m):
        print("   %-26s %8.0f per tile  %5.1f %%" % (n, v / 16, 100 * v / m.sum()))
