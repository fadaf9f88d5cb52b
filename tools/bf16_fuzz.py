#!/usr/bin/env python3
"""One-off fuzz of the bf16 training pass (GPU box): random minibatch sizes 1..1400 (ragged against the 256-workgroup
grid, more tiles than workgroups, both storage forms, both tile heights) against the bf16 CPU checker, same tolerances as
tests/test_gpu_bf16.py.  Not part of the suite (minutes of CPU checker time)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ppo_amd as P
from oracle import np_oracle as npo

def masks_of(act, Q):
    return np.stack([npo.action_mask([(int(a) >> q) & 1 for q in range(Q)]) for a in act])

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 14):
    Q = 8 if trial % 4 else 32
    HID = 256 if trial % 3 else 128
    B = int(rng.integers(1, 1400 if Q == 8 else 300))
    compact = bool(trial & 1)
    P.set_rollout_compact(compact)
    try:
        N, T, H = 48, 32, 4 * Q
        env = P.HipVecEnv(num_envs=N, Q=Q, max_actions=12, seed=100 + trial)
        pol = P.HipPolicy(72, HID, 2, 4, seed=200 + trial, dtype="bf16")
        pol.params = pol.params + (rng.normal(size=pol.num_params) * 0.02).astype(np.float32)
        ro = P.BufferRollouts()
        P.collect_rollouts_steps_(ro, env, pol, T, 1.0)
        ds = P.construct_dataset(ro)
        sel = rng.integers(0, len(ds), size=B)
        st, act = ro.state_data
        stb = st.reshape(-1, H, 72)[sel]
        a0 = (ro.selected_actions.reshape(-1)[sel] - 1).astype(np.int64)
        po, adv = ro.selected_action_probabilities.reshape(-1)[sel], ro.rewards.reshape(-1)[sel]
        P.forward_backward(pol, ds, sel + 1, 10.0, 0.01)
        g = pol.grad()
        g16, _, _ = npo.step_batch_grad_bf16(pol.params, 72, HID, stb, masks_of(act.reshape(-1)[sel], Q), a0, po, adv, 10.0, 0.01)
        e_max = float(np.abs(g - g16).max() / np.abs(g16).max())
        e_2 = float(np.linalg.norm(g - g16) / np.linalg.norm(g16))
        P.forward_backward(pol, ds, sel + 1, 10.0, 0.01)
        rep = bool(np.array_equal(g, pol.grad()))
        ok = e_max <= 1e-2 and e_2 <= 3e-3 and rep
        bad += not ok
        print("trial %2d HID %3d Q %2d B %4d compact %d: max %.2e  l2 %.2e  reproducible %s  %s" % (trial, HID, Q, B, compact, e_max, e_2, rep, "ok" if ok else "FAIL"), flush=True)
    finally:
        P.set_rollout_compact(None)
print("FUZZ", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
