#!/bin/bash
# split-fp32 backward as the default: full GPU suite, headline bench line, strong-scaling shards
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-x6f}; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc" | tee $O/tests.rc; tail -6 $O/tests.log
timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err && python3 tools/show_bench.py $O/bench.json && python3 -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(json.dumps(d['roofline'], indent=1)[:1800])"
for e in 512 1024 2048; do
  PPO_BENCH_FORCE_DIST=1 timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --envs $e > $O/shard_$e.json 2> $O/shard_$e.err && python3 tools/show_bench.py $O/shard_$e.json
done
