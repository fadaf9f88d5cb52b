#!/bin/bash
# deferred finish of the streamed rollout file: its tests, the stream tests, then config 5 streamed with and without it
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-disk1}; mkdir -p $O
timeout -k 10 400 python3 -m pytest tests/test_gpu_disk_async.py tests/test_gpu_parity.py -x -q -m gpu -k "deferred or stream or disk" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/tests.log
[ $rc -eq 0 ] || exit 1
B="python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 --dtype bf16 --envs 65536"
timeout -k 10 300 $B > $O/c5_resident.json 2> $O/c5_resident.err && python3 tools/show_bench.py $O/c5_resident.json resident
for m in 0 1; do
  PPO_DISK_ASYNC=$m timeout -k 10 300 $B --stream /tmp/ppo_bench_stream > $O/c5_stream_$m.json 2> $O/c5_stream_$m.err && python3 tools/show_bench.py $O/c5_stream_$m.json streamed async=$m
  ls -la /tmp/ppo_bench_stream/rank0/rollout.bin; rm -rf /tmp/ppo_bench_stream
done
