#!/usr/bin/env python3
"""End-to-end sanity at the headline size: PPO iterations of the bench workload (4096 envs x 128 steps, 2x256 MLP,
4 epochs, minibatch 4096) and the evaluator's average return (src/evaluate.jl:18-25) every few iterations, for the
fp32 and the bf16 compute mode.  Writes gpurun_out/learning_curve.json (copied to profiles/)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ppo_amd as PPO

ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 24
LAYERS = int(sys.argv[2]) if len(sys.argv) > 2 else 2          # hidden layers (1..4); bf16 mode covers 2 only
out = {"workload": "4096 envs x 128 steps, Policy(72,256,%d,4)," % LAYERS +
       " 4 epochs, minibatch 4096, gamma 1.0, eps 0.05, "
                   "entropy_weight 0.01, Adam 3e-4, synthetic rand-poly-shaped env (Q=8, max_actions 32)", "runs": {}}
for dtype in (("f32", "bf16") if LAYERS == 2 else ("f32",)):
    env = PPO.HipVecEnv(num_envs=4096, Q=8, max_actions=32, seed=7)
    ev = PPO.HipVecEnv(num_envs=1024, Q=8, max_actions=32, seed=99)
    pol = PPO.HipPolicy(72, 256, LAYERS, 4, seed=0, dtype=dtype)
    opt = PPO.Optimiser(PPO.Adam(3e-4))
    curve = []
    t0 = time.perf_counter()
    for it in range(ITERS):
        if it % 4 == 0:
            m, s = PPO.average_returns(pol, ev, 1024)
            curve.append({"iteration": it, "average_return": m, "std": s})
        ro = PPO.BufferRollouts()
        PPO.collect_rollouts_steps_(ro, env, pol, 128, 1.0)
        mean_r = float(ro.raw_rewards.mean())
        ds = PPO.construct_dataset(ro)
        ph, eh, _ = PPO.ppo_train_(pol, opt, ds, 0.05, 4096, 4, 0.01, seed=it, verbose=False)
        curve.append({"iteration": it, "mean_reward_per_step": mean_r, "ppo_loss": ph[-1], "entropy_loss": eh[-1]})
    m, s = PPO.average_returns(pol, ev, 1024)
    curve.append({"iteration": ITERS, "average_return": m, "std": s})
    PPO.synchronize()
    out["runs"][dtype] = {"curve": curve, "wall_s_incl_evaluator_and_host_copies": time.perf_counter() - t0}
    ar = [c["average_return"] for c in curve if "average_return" in c]
    print(dtype, "average return", " -> ".join("%.2f" % a for a in ar))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "learning_curve%s.json" % ("" if LAYERS == 2 else "_%dlayers" % LAYERS)), "w"), indent=1)
