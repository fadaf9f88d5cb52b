#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of k_policy_bwd_x6 (needs libppo_hip_xstamp.so: make -C csrc xstamp).
Shares only -- never quote this build's run time (stamps cost cycles)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PPO_BWD_SPLIT_BF16"] = "1"
import ppo_amd as PPO
PPO._lib.SO_PATH = os.path.join(ROOT, "proximalpolicyoptimization.jl_amd", "libppo_hip_xstamp.so")
PPO._lib._lib = None
L = PPO._lib.lib()
hid = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nst = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
env = PPO.HipVecEnv(num_envs=4096, Q=8, max_actions=128, seed=1)
pol = PPO.HipPolicy(72, hid, 2, 4, seed=0)
ro = PPO.BufferRollouts()
PPO.collect_rollouts_steps_(ro, env, pol, 4, 1.0)
ds = PPO.construct_dataset(ro)
sel = np.arange(1, nst + 1)
for _ in range(3):
    PPO.forward_backward(pol, ds, sel, 0.05, 0.01)
nwg = 256 if hid == 256 else 512
nwg = min(nwg, nst)
out = np.zeros(512 * 24, np.uint64)
L.ppo_debug_x6_stamps.argtypes = [C.c_void_p]
assert L.ppo_debug_x6_stamps(out.ctypes.data) == 0
s = out[: nwg * 24].reshape(nwg, 2, 12).astype(np.float64)
tiles = nst / nwg
names = ["A1 wait + H1 split", "A2 dZ2 + frags", "A3 X staging", "barrier 1", "small grads (early)", "dH1 chain", "dZ1 + dW1", "small grads (late)", "barrier 2", "C prologue (dZ2^T, next loads)", "C dW2", "barrier 3"]
for wv in (0, 1):
    m = s[:, wv, :].mean(axis=0)
    print("wave %s: total %.0f cycles/WG (%.0f per tile)" % ("0" if wv == 0 else "last", m.sum(), m.sum() / tiles))
    for n, v in zip(names, m):
        print("   %-32s %8.0f per tile  %5.1f %%" % (n, v / tiles, 100 * v / m.sum()))
