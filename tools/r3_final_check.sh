#!/bin/bash
# what the driver runs at round end: build check, smoke(), the whole GPU suite, the default bench
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3z; mkdir -p $O
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu --durations=40 > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log; grep -A45 "slowest" $O/tests.log | head -50; [ $rc -ne 0 ] && exit $rc
t0=$(date +%s); timeout -k 10 400 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$? in $(( $(date +%s) - t0 )) s"; cut -c1-250 $O/bench_default.json
