#!/bin/bash
# after the LDS-DMA ordering fix (wait in front of the barrier, read behind it): reproducibility over many trials (default and the
# spread-DMA build that exposed the problem), the split tests, headline + HID = 128 bench
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-dmafix}; mkdir -p $O
(echo "== default"; X6_TRIALS=60 timeout -k 10 300 python3 tools/x6_repro_check2.py; echo "== spread build"; PPO_HIP_LIB=$GRAFT_REPO_ROOT/proximalpolicyoptimization.jl_amd/libppo_hip_dmaspread.so X6_TRIALS=60 timeout -k 10 300 python3 tools/x6_repro_check2.py; echo "== fp32 pass only"; PPO_BWD_SPLIT_BF16=0 X6_FP32_FIRST=0 X6_TRIALS=20 timeout -k 10 300 python3 tools/x6_repro_check2.py) 2>&1 | grep -v amdgpu.ids | cut -c1-160 | tee $O/repro.txt
timeout -k 10 400 python3 -m pytest tests/test_gpu_split_backward.py tests/test_gpu_parity.py -x -q -m gpu -k "split or gradient_vs or switch" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
for v in 1 0; do
  PPO_BWD_SPLIT_BF16=$v timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_$v.json 2> $O/bench_$v.err && python3 tools/show_bench.py $O/bench_$v.json split=$v | cut -c1-220
done
timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --hid 128 > $O/h128.json 2> $O/h128.err && python3 tools/show_bench.py $O/h128.json hid128 | cut -c1-220
PPO_BENCH_FORCE_DIST=1 timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --envs 512 > $O/s512.json 2> $O/s512.err && python3 tools/show_bench.py $O/s512.json envs=512 | cut -c1-220
