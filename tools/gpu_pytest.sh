#!/bin/bash
# tools/gpu_pytest.sh OUTDIR pytest-args...   (GPU box): run pytest with its log under gpurun_out/OUTDIR
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; shift; mkdir -p $O
timeout -k 10 1000 python3 -m pytest "$@" > $O/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -40 $O/pytest.log
exit $rc
