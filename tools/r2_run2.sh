#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2c; rm -rf $O; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; tail -25 $O/gpu_tests.log; echo "tests rc=$rc"
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
PPO_ROLLOUT_COMPACT=1 timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_compact.json 2> $O/bench_compact.err || { tail -5 $O/bench_compact.err; exit 1; }
python3 - <<'PY'
import json
for f in ("bench","bench_compact"):
    d=json.loads(open("gpurun_out/r2c/%s.json"%f).read().strip().splitlines()[-1])
    print(f, "%.0f"%d["value"], "ms %.2f"%d["ms_per_step"], {k:(v.get("avg_ms"),v.get("GB/s")) for k,v in d["kernels"].items()})
PY
timeout -k 10 300 python3 tools/disk_stream_bench.py 16 > $O/disk_stream.json 2> $O/disk_stream.err || { tail -5 $O/disk_stream.err; exit 1; }
cat $O/disk_stream.json
