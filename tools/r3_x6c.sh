#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-x6d}; mkdir -p $O
timeout -k 10 120 python3 tools/x6_stamps.py 256 > $O/stamps256.txt 2>&1; cat $O/stamps256.txt
timeout -k 10 120 python3 tools/x6_stamps.py 128 > $O/stamps128.txt 2>&1; cat $O/stamps128.txt
timeout -k 10 120 python3 tools/x6_stamps.py 256 512 > $O/stamps256_512.txt 2>&1; cat $O/stamps256_512.txt
