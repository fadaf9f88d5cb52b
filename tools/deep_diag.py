#!/usr/bin/env python3
"""Diagnostic (GPU box): per-parameter-block gradient error of a deep policy against the float64 oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ppo_amd as P
from oracle import oracle as orc
import importlib.util
spec = importlib.util.spec_from_file_location("t", os.path.join(ROOT, "tests", "test_gpu_deep_policy.py")); t = importlib.util.module_from_spec(spec); spec.loader.exec_module(t)

def run(F, hid, L, B):
    rng = np.random.default_rng(B + F + hid + L)
    pol = P.HipPolicy(F, hid, L, 4, seed=2)
    p0 = (pol.params + (rng.normal(size=pol.num_params) * 0.03).astype(np.float32)).astype(np.float32)
    pol.params = p0
    states, active, actions, p_old, adv = t._random_batch(P, pol, rng, B, F)
    ro = P.BufferRollouts()
    ro.set_columns(None, states[None], active[None], actions[None].astype(np.int64) + 1, p_old[None], adv[None])
    ds = P.construct_dataset(ro)
    sel = rng.permutation(B) + 1
    lp, le = P.forward_backward(pol, ds, sel, 0.05, 0.01)
    g = pol.grad().astype(np.float64)
    s0 = sel - 1
    g64, olp, ole = orc.step_batch_grad_f64(p0, F, hid, states[s0], active[s0], actions[s0], p_old[s0], adv[s0], 0.05, 0.01, n_hidden=L)
    gm = np.abs(g64).max()
    out = ["F=%d hid=%d L=%d B=%d loss %.3e/%.3e" % (F, hid, L, B, abs(lp - olp), abs(le - ole))]
    off = 0
    dims = [(hid, F)] + [(hid, hid)] * (L - 1) + [(4, hid)]
    for k, (o, i) in enumerate(dims):
        for nm, n in (("W%d" % (k + 1), o * i), ("b%d" % (k + 1), o)):
            d = np.abs(g[off:off + n] - g64[off:off + n])
            out.append("%s %.1e@%d" % (nm, d.max() / gm, int(d.argmax())))
            off += n
    print("  ".join(out), flush=True)

import json
cases = json.loads(os.environ.get("DIAG_CASES", "[[72,256,3,200],[72,256,3,300],[72,256,2,300],[216,256,3,300]]"))
for case in cases:
    run(*case)
