#!/bin/bash
# one box: (1) the split-backward test file on the default build (ring of 4 at HID = 256), (2) A/B at HID = 128: ring of 6 (default) / 12
# W2 pieces in the dH1 chain (libppo_hip_r12h.so: make -C csrc r12h), (3) the HID = 128 cases of the test file on the ring of 12
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r12h}; mkdir -p $O
timeout -k 10 90 python3 -m pytest tests/test_gpu_split_backward.py -x -q > $O/tests_default.log 2>&1; echo tests_default rc=$?; tail -2 $O/tests_default.log
R=$GRAFT_REPO_ROOT/proximalpolicyoptimization.jl_amd/libppo_hip_r12h.so
for r in 1 2; do for v in default r12h; do
  L=""; [ $v = r12h ] && L=$R
  PPO_HIP_LIB=$L timeout -k 10 60 python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --hid 128 > $O/h_${v}_$r.json 2> $O/h_${v}_$r.err && python3 tools/show_bench.py $O/h_${v}_$r.json $v | cut -c1-170
done; done
PPO_HIP_LIB=$R timeout -k 10 60 python3 -m pytest tests/test_gpu_split_backward.py -x -q -k "128" > $O/tests_r12h.log 2>&1; echo tests_r12h rc=$?; tail -2 $O/tests_r12h.log
