#!/bin/bash
# round-3: split-fp32 (bf16x6) fused backward -- gradient tests through it, then A/B of the headline bench
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-x6a}; mkdir -p $O
PPO_BWD_SPLIT_BF16=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gradient or adam or train or two_ranks or learning" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc" | tee $O/tests.rc
tail -8 $O/tests.log
[ $rc -eq 0 ] || exit 1
for v in 0 1 0 1; do
  PPO_BWD_SPLIT_BF16=$v timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_$v.json 2> $O/bench_$v.err && python3 tools/show_bench.py $O/bench_$v.json
done
PPO_BWD_SPLIT_BF16=1 timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --hid 128 > $O/bench_h128_1.json 2> $O/bench_h128_1.err && python3 tools/show_bench.py $O/bench_h128_1.json
PPO_BWD_SPLIT_BF16=0 timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --hid 128 > $O/bench_h128_0.json 2> $O/bench_h128_0.err && python3 tools/show_bench.py $O/bench_h128_0.json
