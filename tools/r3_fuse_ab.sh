#!/bin/bash
# Adam fused into the slab-reduction launch (single-rank training): tests of every training path, then A/B
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-fuse_ab}; mkdir -p $O
timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_deep_policy.py tests/test_gpu_bf16.py tests/test_gpu_engines.py tests/test_gpu_split_backward.py -x -q -m gpu -k "adam or train or epoch or iterate or learning or engines or disk" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit 1
for r in 1 2; do for v in 1 0; do
  PPO_FUSE_REDUCE_ADAM=$v timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_${v}_$r.json 2> $O/bench_${v}_$r.err && python3 tools/show_bench.py $O/bench_${v}_$r.json fuse=$v | cut -c1-260
done; done
PPO_FUSE_REDUCE_ADAM=1 timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --hid 128 > $O/h128_1.json 2> $O/h128_1.err && python3 tools/show_bench.py $O/h128_1.json hid128 fuse=1 | cut -c1-200
PPO_FUSE_REDUCE_ADAM=0 timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --hid 128 > $O/h128_0.json 2> $O/h128_0.err && python3 tools/show_bench.py $O/h128_0.json hid128 fuse=0 | cut -c1-200
