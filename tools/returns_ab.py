#!/usr/bin/env python3
"""A/B of the wide return-scan tile shapes (PPO_RETURNS_VARIANT): one process per variant, HIP-event averages."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = "import sys; sys.path.insert(0, %r); import ppo_amd as P; print('%%.2f %%.2f %%.2f' %% (1e3*P.profile_returns(128, 65536, 1.0, 50), 1e3*P.profile_returns(128, 262144, 1.0, 30), 1e3*P.profile_returns(128, 16384, 1.0, 50)))" % ROOT
for v in ("0", "12832", "12816", "6416", "25616"):
    for rep in range(2):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PPO_RETURNS_VARIANT=v), capture_output=True, text=True)
        print("variant %7s: us at 65536 / 262144 / 16384 columns x 128 rows: %s" % (v, out.stdout.strip() or out.stderr[-300:]), flush=True)
