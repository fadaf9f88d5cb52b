#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of k_policy_fwd (needs libppo_hip_fstamp.so built with -DPPO_FWD_STAMP)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PPO_HIP_LIB"] = os.path.join(ROOT, "proximalpolicyoptimization.jl_amd", "libppo_hip_fstamp.so")
import ppo_amd as PPO
L = PPO._lib.lib()
L.ppo_debug_fwd_stamps.argtypes = [C.c_void_p]
env = PPO.HipVecEnv(num_envs=4096, Q=8, max_actions=128, seed=1)
pol = PPO.HipPolicy(72, 256, 2, 4, seed=0)
ro = PPO.BufferRollouts()
names = ["x convert/prefetch", "layer 1 (288 MFMA)", "layer 2 MFMA chains (1024)", "layer 2 epilogues (lrelu/store/L3)", "tile loop glue", "softmax+sample/loss"]
def show(tag):
    out = np.zeros(1024 * 6, np.uint64)
    assert L.ppo_debug_fwd_stamps(out.ctypes.data) == 0
    s = out.reshape(1024, 6).astype(np.float64).mean(axis=0) / 4
    print(tag, "total %.0f cycles per tile" % s.sum())
    for n, v in zip(names, s):
        print("   %-36s %8.0f  %5.1f %%" % (n, v, 100 * v / s.sum()))
PPO.collect_rollouts_steps_(ro, env, pol, 2, 1.0)
PPO.synchronize(); show("rollout (MODE 1)")
ds = PPO.construct_dataset(ro)
PPO.forward_backward(pol, ds, np.arange(1, 4097), 0.05, 0.01)
PPO.synchronize(); show("train (MODE 2)")
