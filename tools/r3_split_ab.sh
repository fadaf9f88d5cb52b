#!/bin/bash
# The split-fp32 training pass against the fp32-MFMA one, same box, alternating: its tests, the headline workload, the
# reference's own width, the strong-scaling shards (PPO_BWD_SPLIT_BF16 = 0 | 1), per-phase stamps of the split backward.
# (Replaces the one-off r3_x6*.sh scripts of the session that built the kernels; their outputs: gpurun_out/x6a .. x6o.)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-split_ab}; mkdir -p $O
timeout -k 10 400 python3 -m pytest tests/test_gpu_split_backward.py -x -q -m gpu > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/tests.log
[ $rc -eq 0 ] || exit 1
B="python3 bench.py --no-cpu-baseline --steps 5 --warmup 2"
for r in 1 2; do for v in 0 1; do
  PPO_BWD_SPLIT_BF16=$v timeout -k 10 200 $B > $O/bench_${v}_$r.json 2> $O/bench_${v}_$r.err && python3 tools/show_bench.py $O/bench_${v}_$r.json split=$v
done; done
for v in 0 1; do
  PPO_BWD_SPLIT_BF16=$v timeout -k 10 200 $B --hid 128 > $O/h128_$v.json 2> $O/h128_$v.err && python3 tools/show_bench.py $O/h128_$v.json hid128 split=$v
  for e in 512 1024; do
    PPO_BWD_SPLIT_BF16=$v PPO_BENCH_FORCE_DIST=1 timeout -k 10 200 $B --envs $e > $O/shard_${e}_$v.json 2> $O/shard_${e}_$v.err && python3 tools/show_bench.py $O/shard_${e}_$v.json envs=$e split=$v
  done
done
if [ -f proximalpolicyoptimization.jl_amd/libppo_hip_xstamp.so ]; then timeout -k 10 120 python3 tools/x6_stamps.py 256 > $O/stamps256.txt 2>&1 && cat $O/stamps256.txt; fi
