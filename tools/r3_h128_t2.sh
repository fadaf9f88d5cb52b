#!/bin/bash
# HID = 128: split train forward with two tiles per workgroup pass (PPO_FWD_SPLIT_T2_MIN_TILES_128) against one
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-h128_t2}; mkdir -p $O
PPO_FWD_SPLIT_T2_MIN_TILES_128=64 timeout -k 10 300 python3 -m pytest tests/test_gpu_split_backward.py -x -q -m gpu -k "128" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit 1
for r in 1 2; do for v in 0 1024; do
  PPO_FWD_SPLIT_T2_MIN_TILES_128=$v timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --hid 128 > $O/h128_${v}_$r.json 2> $O/h128_${v}_$r.err && python3 tools/show_bench.py $O/h128_${v}_$r.json hid128 t2min=$v | cut -c1-200
done; done
