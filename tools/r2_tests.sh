#!/bin/bash
# GPU parity tests only (one process), log under gpurun_out/
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r2t}; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q ${2:-} > $O/gpu_tests.log 2>&1; rc=$?
tail -25 $O/gpu_tests.log; echo "tests rc=$rc"; exit $rc
