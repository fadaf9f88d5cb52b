#!/bin/bash
# split-fp32 backward: its tests, A/B of the headline / HID = 128 bench, per-phase stamps
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-x6e}; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_split_backward.py -x -q -m gpu > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/tests.log
[ $rc -eq 0 ] || exit 1
for v in 0 1 1; do
  PPO_BWD_SPLIT_BF16=$v timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_$v.json 2> $O/bench_$v.err && python3 tools/show_bench.py $O/bench_$v.json
done
PPO_BWD_SPLIT_BF16=1 timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --hid 128 > $O/bench_h128_1.json 2> $O/bench_h128_1.err && python3 tools/show_bench.py $O/bench_h128_1.json
timeout -k 10 120 python3 tools/x6_stamps.py 256 > $O/stamps256.txt 2>&1; cat $O/stamps256.txt
