#!/bin/bash
# per-phase clocks of the two split training kernels on one box (prepared at the end of round 3).
# Build first: make -C proximalpolicyoptimization.jl_amd/csrc xstamp fxstamp
# usage (on the box): bash tools/r4_stamps.sh [outdir]
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-stamps}; mkdir -p $O
timeout -k 10 60 python3 tools/fx6_stamps.py 256 > $O/fwd256.txt 2>&1; echo fwd256 rc=$?; cat $O/fwd256.txt | grep -v amdgpu.ids
timeout -k 10 60 python3 tools/x6_stamps.py 256 > $O/bwd256.txt 2>&1; echo bwd256 rc=$?; cat $O/bwd256.txt | grep -v amdgpu.ids
timeout -k 10 60 python3 tools/fx6_stamps.py 128 > $O/fwd128.txt 2>&1; echo fwd128 rc=$?; cat $O/fwd128.txt | grep -v amdgpu.ids
