#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3d
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -k "returns or gae" > gpurun_out/r3d/tests.log 2>&1; rc=$?; tail -3 gpurun_out/r3d/tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 500 python3 tools/returns_ab.py | tee gpurun_out/r3d/ab.log
