#!/bin/bash
# A/B: ring depth of the split backward's dH1 chain at HID = 256 (3, no spill: default; 6 + one spilled accumulator tile)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-x6i}; mkdir -p $O
for r in 1 2; do
for v in default rd6; do
  L=""; [ $v = rd6 ] && L=$GRAFT_REPO_ROOT/proximalpolicyoptimization.jl_amd/libppo_hip_rd6.so
  PPO_HIP_LIB=$L timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 4 --warmup 2 > $O/bench_${v}_$r.json 2> $O/bench_${v}_$r.err && python3 tools/show_bench.py $O/bench_${v}_$r.json $v
done
done
PPO_BENCH_FORCE_DIST=1 timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --envs 512 > $O/shard_512.json 2> $O/shard_512.err && python3 tools/show_bench.py $O/shard_512.json envs=512
PPO_BENCH_FORCE_DIST=1 timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --envs 1024 > $O/shard_1024.json 2> $O/shard_1024.err && python3 tools/show_bench.py $O/shard_1024.json envs=1024
