#!/bin/bash
# engines test + HID=128 small-batch A/B of the train-tile path + a deep-policy bench line
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3f; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_engines.py tests/test_gpu_two_ranks.py tests/test_gpu_deep_policy.py -x -q > $O/tests.log 2>&1; rc=$?; tail -4 $O/tests.log; [ $rc -ne 0 ] && exit $rc
for e in 512 1024; do for tt in 0 100000; do
  PPO_TRAIN_TILE_MAX_TILES=$tt timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --hid 128 --envs $e > $O/h128_${e}_tt$tt.json 2> $O/h128_${e}_tt$tt.err && python3 tools/show_bench.py $O/h128_${e}_tt$tt.json "hid128 envs=$e tt=$tt" || { tail -5 $O/h128_${e}_tt$tt.err; exit 1; }
done; done
for l in 1 3 4; do
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --layers $l > $O/layers_$l.json 2> $O/layers_$l.err && python3 tools/show_bench.py $O/layers_$l.json "layers=$l" || { tail -5 $O/layers_$l.err; exit 1; }
done
