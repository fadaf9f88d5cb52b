#!/usr/bin/env python3
"""Config-5 streaming check: 65536 envs, bf16 policy, every finished rollout step copied device -> pinned host
(hipMemcpyAsync on a copy stream) -> writer thread -> <dir>/rollout.bin, against the same rollout kept resident."""
import json, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ppo_amd as PPO

N, T = 65536, int(sys.argv[1]) if len(sys.argv) > 1 else 128
env = PPO.HipVecEnv(num_envs=N, Q=8, max_actions=128, seed=3)
pol = PPO.HipPolicy(72, 256, 2, 4, seed=0, dtype="bf16")
res = {"envs": N, "steps": T, "bytes_per_env_step": {"compact": 64 + 17, "expanded": 32 * 72 + 17}}
for mode in ("resident", "streamed", "streamed_expanded"):
    d = tempfile.mkdtemp(prefix="ppo_stream_")
    PPO.set_rollout_compact(False if mode == "streamed_expanded" else None)     # default: snapshots while streaming
    ro = PPO.BufferRollouts() if mode == "resident" else PPO.DiskRollouts(d)
    PPO.collect_rollouts_steps_(ro, env, pol, 2, 1.0)          # warm-up (allocations, first-touch)
    PPO.synchronize()
    ro = PPO.BufferRollouts() if mode == "resident" else PPO.DiskRollouts(d)
    t0 = time.perf_counter()
    PPO.collect_rollouts_steps_(ro, env, pol, T, 1.0)
    PPO.synchronize()
    dt = time.perf_counter() - t0
    res[mode] = {"seconds": dt, "env_steps_per_s": N * T / dt}
    if mode != "resident":
        sz = os.path.getsize(os.path.join(d, "rollout.bin"))
        res[mode]["file_bytes"] = sz
        res[mode]["GB_per_s_to_disk"] = sz / dt / 1e9
    shutil.rmtree(d, ignore_errors=True)
res["streamed_over_resident"] = res["streamed"]["env_steps_per_s"] / res["resident"]["env_steps_per_s"]
print(json.dumps(res))
