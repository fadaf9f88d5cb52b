#!/bin/bash
# round-2 first GPU pass: parity tests, the default bench line, and the rank-0 shards of a strong-scaling run
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2a; rm -rf $O; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/gpu_tests.log; tail -3 $O/gpu_tests.log
timeout -k 10 400 python3 bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
cut -c1-300 $O/bench.json
for e in 512 1024 2048 4096; do
  PPO_BENCH_FORCE_DIST=1 timeout -k 10 200 python3 bench.py --steps 5 --warmup 2 --envs $e --no-cpu-baseline > $O/shard_$e.json 2> $O/shard_$e.err || { tail -5 $O/shard_$e.err; exit 1; }
  python3 - $O/shard_$e.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["config"]["envs_per_gpu"], "ms/iter %.2f"%d["ms_per_step"], d["allreduce"], {k:v["avg_ms"] for k,v in d["kernels"].items() if "avg_ms" in v and "@" not in k})
PY
done
