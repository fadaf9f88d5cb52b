#!/bin/bash
# A/B on one box: split backward before / after the LDS-DMA ordering fix (libppo_hip_prevdma.so = the kernel of the commit before)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-dmafix_ab}; mkdir -p $O
for r in 1 2; do for v in default prevdma; do
  L=""; [ $v = prevdma ] && L=$GRAFT_REPO_ROOT/proximalpolicyoptimization.jl_amd/libppo_hip_prevdma.so
  for e in 512 1024; do
    PPO_HIP_LIB=$L PPO_BENCH_FORCE_DIST=1 timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --envs $e > $O/s${e}_${v}_$r.json 2> $O/s${e}_${v}_$r.err && python3 tools/show_bench.py $O/s${e}_${v}_$r.json envs=$e $v | cut -c1-170
  done
  PPO_HIP_LIB=$L timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 4 --warmup 2 --hid 128 > $O/h128_${v}_$r.json 2> $O/h128_${v}_$r.err && python3 tools/show_bench.py $O/h128_${v}_$r.json hid128 $v | cut -c1-170
done; done
