#!/usr/bin/env python3
"""Diagnostic: where a streamed config-5 iteration spends its host time (constructor / collect / train / sync)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ppo_amd as P
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
P.set_disk_async(bool(mode))
env = P.HipVecEnv(num_envs=65536, Q=8, max_actions=128, seed=1234)
pol = P.HipPolicy(72, 256, 2, 4, seed=0, dtype="bf16")
opt = P.Optimiser(P.Adam(1e-4))
for it in range(3):
    t0 = time.perf_counter()
    d = P.DiskRollouts("/tmp/ppo_async_t/r")
    t1 = time.perf_counter()
    P.collect_rollouts_steps_(d, env, pol, 128, 1.0)
    P.synchronize()
    t2 = time.perf_counter()
    ds = P.construct_dataset(d._device)
    P.ppo_train_(pol, opt, ds, 0.05, 65536, 1, 0.01, seed=it, verbose=False)
    P.synchronize()
    t3 = time.perf_counter()
    P.disk_sync(d)
    t4 = time.perf_counter()
    print("async" if mode else "sync ", "it", it, "ctor %.1f ms  collect %.1f ms  train(1 epoch) %.1f ms  sync %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3), flush=True)
