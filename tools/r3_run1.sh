#!/bin/bash
# round-3 first GPU pass: full GPU test suite, headline bench, the strong-scaling shards (baseline for this round)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3a; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/tests.rc
tail -5 $O/tests.log
timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err && cut -c1-300 $O/bench.json
for e in 512 1024; do
  PPO_BENCH_FORCE_DIST=1 timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --envs $e > $O/shard_$e.json 2> $O/shard_$e.err && python3 tools/show_bench.py $O/shard_$e.json
done
