#!/bin/bash
# A/B: split train forward ring 6 + 3 accumulators (default) vs ring 12 + 2 accumulators; per-phase stamps of the split backward
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-x6l}; mkdir -p $O
for r in 1 2; do
for v in default fr12; do
  L=""; [ $v = fr12 ] && L=$GRAFT_REPO_ROOT/proximalpolicyoptimization.jl_amd/libppo_hip_fr12.so
  PPO_HIP_LIB=$L timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 4 --warmup 2 > $O/bench_${v}_$r.json 2> $O/bench_${v}_$r.err && python3 tools/show_bench.py $O/bench_${v}_$r.json $v
done
done
timeout -k 10 120 python3 tools/x6_stamps.py 256 > $O/stamps256.txt 2>&1; cat $O/stamps256.txt
