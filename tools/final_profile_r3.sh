#!/bin/bash
# Round-3 measurement bundle (GPU box).  Part A (default): PMC passes FIRST, then the headline bench (which reads the HBM
# traffic of those passes through PPO_PMC_TRAFFIC_FILE) and the rocprofv3 kernel stats.  tools/collect_profiles_r3.sh then
# copies part A's pmc_traffic.json to profiles/r03_pmc_traffic.json, which travels to the box of part B ("b": non-headline
# configs -- bf16, config 4/5 shapes, HID = 128, deep policy, streaming, strong-scaling shards, 2-rank rehearsal; a fresh box,
# gpurun_out/ does not travel) and is what its bench lines read.  Every line carries the figure bench.py read itself;
# nothing is filled in afterwards.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
F=gpurun_out/final3; mkdir -p $F
B="python3 bench.py --no-cpu-baseline"
if [ "${1:-a}" = "a" ]; then
  export PPO_PMC_TRAFFIC_FILE=$GRAFT_REPO_ROOT/$F/pmc_traffic.json
  rm -rf $F/trace $F/pmc
  A="--steps 1 --warmup 0 --t-steps 8 --epochs 1"
  P=$F/pmc/f32_4096
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $P/sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- $B $A > $F/pmc_sq.log 2>&1 || { tail -5 $F/pmc_sq.log; exit 1; }
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $P/grbm --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -- $B $A > $F/pmc_grbm.log 2>&1 || { tail -5 $F/pmc_grbm.log; exit 1; }
  for cfg in "f32_4096:" "f32_q32_8192:--quads 32 --envs 8192" "bf16_4096:--dtype bf16" "bf16_65536:--dtype bf16 --envs 65536" "f32_h128_4096:--hid 128" "f32_l3_4096:--layers 3"; do
    name=${cfg%%:*}; extra=${cfg#*:}
    for c in fetch:FETCH_SIZE write:WRITE_SIZE; do
      PPO_PMC_TRAFFIC_FILE=/nonexistent timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $F/pmc/$name/${c%%:*} --pmc ${c#*:} -- $B $A $extra > $F/pmc_${name}_${c%%:*}.log 2>&1 || { tail -5 $F/pmc_${name}_${c%%:*}.log; exit 1; }
    done
    echo "pmc $name done"
  done
  find $F/pmc -name "*kernel_trace.csv" -delete
  python3 tools/pmc_summary.py $F/pmc/f32_4096 > $F/pmc_summary.txt
  python3 tools/pmc_traffic2.py $F/pmc_traffic.json "f32|envs=4096|quads=8|hid=256=$F/pmc/f32_4096" "f32|envs=8192|quads=32|hid=256=$F/pmc/f32_q32_8192" \
      "bf16|envs=4096|quads=8|hid=256=$F/pmc/bf16_4096" "bf16|envs=65536|quads=8|hid=256=$F/pmc/bf16_65536" "f32|envs=4096|quads=8|hid=128=$F/pmc/f32_h128_4096" \
      "f32|envs=4096|quads=8|hid=256|layers=3=$F/pmc/f32_l3_4096"
  timeout -k 10 500 python3 bench.py --steps 5 --warmup 2 > $F/bench.json 2> $F/bench.err || { tail -5 $F/bench.err; exit 1; }
  cut -c1-200 $F/bench.json
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $F/trace -- $B --steps 3 --warmup 1 > $F/trace.log 2>&1 || { tail -5 $F/trace.log; exit 1; }
  rm -f $F/trace/*/*kernel_trace.csv
else
  timeout -k 10 300 $B --steps 3 --warmup 1 --dtype bf16 > $F/bench_bf16.json 2> $F/bench_bf16.err || { tail -5 $F/bench_bf16.err; exit 1; }
  rm -rf $F/trace_bf16
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $F/trace_bf16 -- $B --steps 2 --warmup 1 --dtype bf16 > $F/trace_bf16.log 2>&1 || { tail -5 $F/trace_bf16.log; exit 1; }
  # where the slowest launches of each hot kernel sit (round 2's stats showed one k_policy_bwd_bf16 launch 12x the average)
  python3 tools/trace_outliers.py $F/trace_bf16 > $F/trace_bf16_outliers.txt 2>&1; cat $F/trace_bf16_outliers.txt
  rm -f $F/trace_bf16/*/*kernel_trace.csv
  timeout -k 10 300 $B --steps 1 --warmup 1 --dtype bf16 --envs 65536 > $F/bench_bf16_c5.json 2> $F/bench_bf16_c5.err || { tail -5 $F/bench_bf16_c5.err; exit 1; }
  timeout -k 10 300 $B --steps 1 --warmup 1 --dtype bf16 --envs 65536 --stream /tmp/ppo_bench_stream > $F/bench_bf16_c5_streamed.json 2> $F/bench_bf16_c5_streamed.err || { tail -5 $F/bench_bf16_c5_streamed.err; exit 1; }
  rm -rf /tmp/ppo_bench_stream
  timeout -k 10 300 $B --steps 3 --warmup 1 --quads 32 --envs 8192 > $F/bench_c4.json 2> $F/bench_c4.err || { tail -5 $F/bench_c4.err; exit 1; }
  timeout -k 10 300 $B --steps 3 --warmup 1 --hid 128 > $F/bench_h128.json 2> $F/bench_h128.err || { tail -5 $F/bench_h128.err; exit 1; }
  rm -rf $F/trace_h128
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $F/trace_h128 -- $B --steps 2 --warmup 1 --hid 128 > $F/trace_h128.log 2>&1 || { tail -5 $F/trace_h128.log; exit 1; }
  rm -f $F/trace_h128/*/*kernel_trace.csv
  timeout -k 10 300 $B --steps 3 --warmup 1 --layers 3 > $F/bench_l3.json 2> $F/bench_l3.err || { tail -5 $F/bench_l3.err; exit 1; }
  timeout -k 10 300 $B --steps 3 --warmup 1 --layers 1 > $F/bench_l1.json 2> $F/bench_l1.err || { tail -5 $F/bench_l1.err; exit 1; }
  timeout -k 10 300 python3 tools/disk_stream_bench.py 128 > $F/disk_stream.json 2> $F/disk_stream.err || { tail -5 $F/disk_stream.err; exit 1; }
  for e in 256 512 1024 2048; do
    PPO_BENCH_FORCE_DIST=1 timeout -k 10 200 $B --steps 5 --warmup 2 --envs $e > $F/shard_$e.json 2> $F/shard_$e.err || { tail -5 $F/shard_$e.err; exit 1; }
  done
  PPO_TRAIN_TILE_MAX_TILES=100000 PPO_BENCH_FORCE_DIST=1 timeout -k 10 200 $B --steps 5 --warmup 2 --envs 512 > $F/shard_512_train_tile.json 2> $F/shard_512_tt.err || { tail -5 $F/shard_512_tt.err; exit 1; }
  PPO_BENCH_BACKEND=gloo PPO_BENCH_SHARE_GPU=1 timeout -k 10 300 $B --gpus 2 --steps 2 --warmup 1 > $F/bench_2ranks_shared_gpu.json 2> $F/bench_2ranks.err || { tail -5 $F/bench_2ranks.err; exit 1; }
  PPO_BENCH_BACKEND=gloo PPO_BENCH_SHARE_GPU=1 timeout -k 10 300 $B --gpus 4 --steps 2 --warmup 1 --envs 1024 > $F/bench_4ranks_shared_gpu.json 2> $F/bench_4ranks.err || { tail -5 $F/bench_4ranks.err; exit 1; }
  ls $F
fi
