#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2n; rm -rf $O; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "disk or stream or gradient or train or step_batch or iterate" > $O/gpu_tests.log 2>&1; rc=$?; tail -5 $O/gpu_tests.log; echo "tests rc=$rc"
[ $rc -eq 0 ] || exit $rc
for w in 1 3 6; do PPO_DISK_WRITERS=$w timeout -k 10 300 python3 tools/disk_stream_bench.py 128 > $O/disk_stream_w$w.json 2> $O/disk_stream_w$w.err || { tail -5 $O/disk_stream_w$w.err; exit 1; }
python3 - $O/disk_stream_w$w.json $w <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("writers", sys.argv[2], "resident %.1f M"%(d["resident"]["env_steps_per_s"]/1e6), "streamed %.1f M"%(d["streamed"]["env_steps_per_s"]/1e6), "%.2f GB/s"%d["streamed"]["GB_per_s_to_disk"], "ratio %.2f"%d["streamed_over_resident"], "expanded %.1f M"%(d["streamed_expanded"]["env_steps_per_s"]/1e6))
PY
done
timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r2n/bench.json").read().strip().splitlines()[-1])
print("bench %.0f"%d["value"], "ms %.2f"%d["ms_per_step"], {k:v.get("avg_ms") for k,v in d["kernels"].items() if "@" not in k}, d["roofline"]["traffic"])
PY
