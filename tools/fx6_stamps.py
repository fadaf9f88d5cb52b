#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of k_policy_fwd_train_x6t (the two-tiles-per-pass split train forward; needs
libppo_hip_fxstamp.so: make -C csrc fxstamp).  Shares only -- never quote this build's run time (stamps cost cycles).
usage: fx6_stamps.py [hid=256] [states=4096]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PPO_BWD_SPLIT_BF16"] = "1"
import ppo_amd as PPO
PPO._lib.SO_PATH = os.path.join(ROOT, "proximalpolicyoptimization.jl_amd", "libppo_hip_fxstamp.so")
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
groups = (nst + 1) // 2
nwg = min(256 if hid == 256 else 512, groups)
out = np.zeros(512 * 2 * 8, np.uint64)
L.ppo_debug_fx6_stamps.argtypes = [C.c_void_p]
assert L.ppo_debug_fx6_stamps(out.ctypes.data) == 0
s = out[: nwg * 16].reshape(nwg, 2, 8).astype(np.float64)
passes = groups / nwg
names = ["layer 1: X convert + 15 x 2 MFMAs", "H1 store, split, LDS fragments", "W2 ring fill + next X issue", "barrier 1",
         "layer 2: hid/16 k-steps x 12 MFMAs", "H2 store + layer-3 partial dots", "barrier 2", "loss tail (waves 0, 1)"]
for wv in (0, 1):
    m = s[:, wv, :].mean(axis=0)
    print("wave %s: total %.0f cycles/WG (%.0f per two-tile pass; MFMA issue of the SIMD's two waves: %d)" %
          ("0" if wv == 0 else "last", m.sum(), m.sum() / passes, 2 * 2 * (15 + hid // 16 * 6) * 32))
    for n, v in zip(names, m):
        print("   %-36s %8.0f per pass  %5.1f %%" % (n, v / passes, 100 * v / m.sum()))
