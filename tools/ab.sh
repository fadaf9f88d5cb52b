#!/bin/bash
# A/B helper (GPU box): tools/ab.sh "<bench args>" variant1 variant2 ...   -- runs bench.py with
# proximalpolicyoptimization.jl_amd/libppo_hip_<variant>.so (or the default build for "default"), twice, alternating.
ARGS="$1"; shift
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = default ]; then unset PPO_HIP_LIB; else export PPO_HIP_LIB=$PWD/proximalpolicyoptimization.jl_amd/libppo_hip_$v.so; fi
    python bench.py --steps 4 --warmup 1 --no-cpu-baseline $ARGS > gpurun_out/ab_$v.log 2>&1 || { tail -5 gpurun_out/ab_$v.log; exit 1; }
    python tools/show_bench.py gpurun_out/ab_$v.log $v
  done
done
