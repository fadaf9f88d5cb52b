#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2t; rm -rf $O; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "gradient_vs_f64 or q32 or any_hidden or ragged or checkpoint" > $O/gpu_tests.log 2>&1; rc=$?; tail -4 $O/gpu_tests.log; echo "tests rc=$rc"
[ $rc -eq 0 ] || exit $rc
for rep in 1 2 3; do for v in default dh2valu; do
  if [ "$v" = default ]; then unset PPO_HIP_LIB; else export PPO_HIP_LIB=$PWD/proximalpolicyoptimization.jl_amd/libppo_hip_$v.so; fi
  for h in 256 128; do
    timeout -k 10 200 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --hid $h > $O/ab_${v}_$h.json 2> $O/ab_${v}_$h.err || { tail -5 $O/ab_${v}_$h.err; exit 1; }
    python3 - $O/ab_${v}_$h.json $v $h <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d["kernels"]
print(sys.argv[2], "hid", sys.argv[3], "value %.0f"%d["value"], "bwd %.4f ms frac %.4f"%(k["k_policy_bwd"]["avg_ms"], k["k_policy_bwd"]["frac"]))
PY
  done
done; done
