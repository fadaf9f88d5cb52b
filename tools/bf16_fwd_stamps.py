#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of k_policy_fwd_bf16 (needs libppo_hip_bstamp.so: make -C csrc bstamp).
Shares only -- never quote this build's run time (stamps cost cycles)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PPO_HIP_LIB"] = os.path.join(ROOT, "proximalpolicyoptimization.jl_amd", "libppo_hip_bstamp.so")
import ppo_amd as PPO
L = PPO._lib.lib()
L.ppo_debug_bf16_fwd_stamps.argtypes = [C.c_void_p]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = PPO.HipVecEnv(num_envs=N, Q=8, max_actions=128, seed=1)
pol = PPO.HipPolicy(72, 256, 2, 4, seed=0, dtype="bf16")
ro = PPO.BufferRollouts()
names = ["prologue (W2 -> LDS, env slots)", "state rows -> operands", "layer 1", "layer 2 + 3", "softmax + sample / loss", "env step (rollout)"]
def show(tag, tiles):
    out = np.zeros(2048 * 6, np.uint64)
    assert L.ppo_debug_bf16_fwd_stamps(out.ctypes.data) == 0
    s = out.reshape(2048, 6).astype(np.float64).mean(axis=0)
    print("%s: total %.0f cycles per wave, %d tiles per wave" % (tag, s.sum(), tiles))
    for n, v in zip(names, s):
        print("   %-34s %9.0f per tile  %5.1f %%" % (n, v / tiles, 100 * v / s.sum()))
T = 8
PPO.collect_rollouts_steps_(ro, env, pol, T, 1.0)
PPO.synchronize(); show("rollout (persistent, %d steps)" % T, T * N // 2048)
ds = PPO.construct_dataset(ro)
PPO.forward_backward(pol, ds, np.arange(1, N + 1), 0.05, 0.01)
PPO.synchronize(); show("train forward", N // 2048)
