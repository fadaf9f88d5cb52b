#!/bin/bash
# A/B on one box: HID = 256 split backward with a ring of 3 / 4 W2 pieces (run when 3 was the default; 4 is now) in the dH1 chain (libppo_hip_rd4.so: make -C csrc rd4)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-rd4}; mkdir -p $O
RD4=$GRAFT_REPO_ROOT/proximalpolicyoptimization.jl_amd/libppo_hip_rd4.so
for r in 1 2; do for v in default rd4; do
  L=""; [ $v = rd4 ] && L=$RD4
  PPO_HIP_LIB=$L timeout -k 10 60 python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/b_${v}_$r.json 2> $O/b_${v}_$r.err && python3 tools/show_bench.py $O/b_${v}_$r.json $v | cut -c1-170
done; done
PPO_HIP_LIB=$RD4 timeout -k 10 60 python3 -m pytest tests/test_gpu_split_backward.py -x -q -k "256" > $O/tests_rd4.log 2>&1; echo tests_rd4 rc=$?; tail -2 $O/tests_rd4.log
