#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-x6c}; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_split_backward.py -x -q -m gpu > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -15 $O/tests.log
cat gpurun_out/split_backward_accuracy.jsonl
timeout -k 10 120 python3 tools/x6_stamps.py 256 > $O/stamps256.txt 2>&1; cat $O/stamps256.txt
timeout -k 10 120 python3 tools/x6_stamps.py 128 > $O/stamps128.txt 2>&1; cat $O/stamps128.txt
