#!/bin/bash
# split train forward for Q = 32 states (four tiles per state): its tests, then config 4 with / without it
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-c4a}; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "q32 or config4" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/tests.log
[ $rc -eq 0 ] || exit 1
for m in 0 100000000; do
  PPO_FWD_SPLIT_MAX_TILES=$m timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 --quads 32 --envs 8192 > $O/c4_$m.json 2> $O/c4_$m.err && python3 tools/show_bench.py $O/c4_$m.json fwdsplit=$m
done
