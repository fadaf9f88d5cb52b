#!/bin/bash
# A/B of the XCD-aware block order of k_policy_wgrad (PPO_WGRAD_XCD / PPO_WGRAD_KS)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3h; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_deep_policy.py -x -q -k "gradient" > $O/tests.log 2>&1; rc=$?; tail -3 $O/tests.log; [ $rc -ne 0 ] && exit $rc
run() { name=$1; shift; env "$@" > /dev/null 2>&1; }
for cfg in "xcd0:PPO_WGRAD_XCD=0" "xcd1:PPO_WGRAD_XCD=1" "xcd1_ks24:PPO_WGRAD_XCD=1 PPO_WGRAD_KS=24" "xcd1_ks8:PPO_WGRAD_XCD=1 PPO_WGRAD_KS=8" "xcd0:PPO_WGRAD_XCD=0" "xcd1:PPO_WGRAD_XCD=1"; do
  name=${cfg%%:*}; vars=${cfg#*:}
  for e in 256 512; do
    env $vars PPO_BWD_SMALL_MAX_TILES=100000 PPO_BENCH_FORCE_DIST=1 timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --envs $e > $O/s_${e}_$name.json 2> $O/s_${e}_$name.err && python3 tools/show_bench.py $O/s_${e}_$name.json "3-product envs=$e $name" | cut -c1-330 || { tail -5 $O/s_${e}_$name.err; exit 1; }
  done
done
for cfg in "xcd0:PPO_WGRAD_XCD=0" "xcd1:PPO_WGRAD_XCD=1"; do
  name=${cfg%%:*}; vars=${cfg#*:}
  env $vars timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --layers 3 > $O/l3_$name.json 2> $O/l3_$name.err && python3 tools/show_bench.py $O/l3_$name.json "layers=3 $name" | cut -c1-330 || { tail -5 $O/l3_$name.err; exit 1; }
  env $vars PPO_BWD_SMALL_MAX_TILES=100000 timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/l2_$name.json 2> $O/l2_$name.err && python3 tools/show_bench.py $O/l2_$name.json "layers=2 three-product at 4096 $name" | cut -c1-330 || { tail -5 $O/l2_$name.err; exit 1; }
done
