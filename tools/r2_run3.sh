#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2d; rm -rf $O; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "disk or stream or q32 or gradient or bitexact" > $O/gpu_tests.log 2>&1; rc=$?; tail -5 $O/gpu_tests.log; echo "tests rc=$rc"
[ $rc -eq 0 ] || exit $rc
for t in 16 128; do timeout -k 10 300 python3 tools/disk_stream_bench.py $t > $O/disk_stream_$t.json 2> $O/disk_stream_$t.err || { tail -5 $O/disk_stream_$t.err; exit 1; }; cat $O/disk_stream_$t.json; echo; done
for d in f32 bf16; do
timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --hid 128 --dtype $d > $O/bench_h128_$d.json 2> $O/bench_h128_$d.err || { tail -5 $O/bench_h128_$d.err; exit 1; }
python3 - $O/bench_h128_$d.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("hid128", d["dtype"], "%.0f"%d["value"], "ms %.2f"%d["ms_per_step"], {k:(v.get("avg_ms"),v.get("frac")) for k,v in d["kernels"].items() if "@" not in k})
PY
done
