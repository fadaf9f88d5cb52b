#!/bin/bash
# first A/B of round 4, prepared at the end of round 3 (no GPU minutes left then): (1) wave priority in the MFMA loops of the split
# backward (xprio1..3), (2) transition ids of the two-tile split train forward through the scalar cache instead of vector loads that wait
# for vmcnt(0) (fsidx), (3) its layer-2 LDS reads one set ahead of their MFMAs (fzpipe; fzs = fzpipe + fsidx), (4) no reloads past the weight stream
# in the last ring round (fnodangle, xnodangle; fall = every forward knob so far), (5) the next pass's state rows requested early in layer 1
# (fxearly; fbest = fxearly + fnodangle + fzpipe).  DESIGN.md section 9 item 0 has the reasoning.
# Build first: make -C proximalpolicyoptimization.jl_amd/csrc xprio fsidx fzpipe xzpipe xnodangle fnodangle fxearly   (the .so files travel with the snapshot)
# usage (on the box): bash tools/r4_first_ab.sh [outdir] [bench|tests]   (both by default: ~10 minutes, give gpurun --timeout 1000)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-xprio}; mkdir -p $O
P=$GRAFT_REPO_ROOT/proximalpolicyoptimization.jl_amd
MODE=${2:-both}
if [ $MODE != tests ]; then
for r in 1 2; do for v in default fsidx fzpipe fzs fnodangle fall fxearly fbest xnodangle xprio1 xprio2 xprio3; do
  L=""; [ $v != default ] && L=$P/libppo_hip_$v.so
  [ -z "$L" ] || [ -f "$L" ] || { echo "missing $L"; continue; }
  PPO_HIP_LIB=$L timeout -k 10 60 python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/b_${v}_$r.json 2> $O/b_${v}_$r.err && python3 tools/show_bench.py $O/b_${v}_$r.json $v | cut -c1-170
done; done
# HID = 128: dZ2 pieces one k-step ahead in the backward chain (xzpipe), the forward variants again
for r in 1 2; do for v in default xzpipe xnodangle fzs fbest; do
  L=""; [ $v != default ] && L=$P/libppo_hip_$v.so
  [ -z "$L" ] || [ -f "$L" ] || continue
  PPO_HIP_LIB=$L timeout -k 10 60 python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --hid 128 > $O/h128_${v}_$r.json 2> $O/h128_${v}_$r.err && python3 tools/show_bench.py $O/h128_${v}_$r.json hid128 $v | cut -c1-170
done; done
# the 512-env strong-scaling shard (one-tile split forward) with and without the scalar-cache ids
for r in 1 2; do for v in default fsidx; do
  L=""; [ $v != default ] && L=$P/libppo_hip_$v.so
  [ -z "$L" ] || [ -f "$L" ] || continue
  PPO_HIP_LIB=$L PPO_BENCH_FORCE_DIST=1 timeout -k 10 60 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --envs 512 > $O/s512_${v}_$r.json 2> $O/s512_${v}_$r.err && python3 tools/show_bench.py $O/s512_${v}_$r.json envs=512 $v | cut -c1-170
done; done
fi
[ $MODE = bench ] && exit 0
for v in fsidx fzpipe fzs fnodangle fall fxearly fbest xnodangle xzpipe xprio1 xprio2 xprio3; do
  [ -f $P/libppo_hip_$v.so ] || continue
  PPO_HIP_LIB=$P/libppo_hip_$v.so timeout -k 10 90 python3 -m pytest tests/test_gpu_split_backward.py -x -q > $O/tests_$v.log 2>&1; echo tests_$v rc=$?; tail -1 $O/tests_$v.log
done
