#!/bin/bash
# disk streaming A/B on the GPU box: disk tests, then tools/disk_stream_bench.py with 1 / 4 / 8 / 16 records per writev()
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; rm -rf $O; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "disk or stream" > $O/gpu_tests.log 2>&1; rc=$?; tail -3 $O/gpu_tests.log; echo "tests rc=$rc"
[ $rc -eq 0 ] || exit $rc
PPO_DISK_BATCH=1 timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "disk or stream" > $O/gpu_tests_c0.log 2>&1; rc=$?; tail -3 $O/gpu_tests_c0.log; echo "tests(batch=1) rc=$rc"
[ $rc -eq 0 ] || exit $rc
df -h /tmp | tail -1; nproc
for rep in 1 2; do for c in 1 4 8 16; do
  PPO_DISK_BATCH=$c timeout -k 10 300 python3 tools/disk_stream_bench.py 128 > $O/ds_$c.json 2> $O/ds_$c.err || { tail -5 $O/ds_$c.err; exit 1; }
  python3 - $O/ds_$c.json $c <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("batch", sys.argv[2], "resident %.1f M  streamed %.1f M (%.2f of resident, %.2f GB/s)  expanded %.1f M (%.2f GB/s)" % (d["resident"]["env_steps_per_s"]/1e6, d["streamed"]["env_steps_per_s"]/1e6, d["streamed_over_resident"], d["streamed"]["GB_per_s_to_disk"], d["streamed_expanded"]["env_steps_per_s"]/1e6, d["streamed_expanded"]["GB_per_s_to_disk"]))
PY
done; done
