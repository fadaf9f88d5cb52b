#!/bin/bash
# Copy the judged summaries of gpurun_out/final3 (tools/final_profile_r3.sh a / b) into profiles/ (tracked).  Nothing is
# edited on the way: every bench line already carries the HBM traffic of its own bundle (PPO_PMC_TRAFFIC_FILE).
set -e
F=gpurun_out/final3
cp $F/bench.json profiles/r03_bench.json
cp $(ls -t $F/trace/*/*kernel_stats.csv | head -1) profiles/r03_rocprof_kernel_stats.csv
cp $F/pmc_summary.txt profiles/r03_pmc_summary.txt
cp $F/pmc_traffic.json profiles/r03_pmc_traffic.json
for p in "bench_bf16:bench_bf16" "bench_bf16_c5:bench_bf16_65536envs" "bench_bf16_c5_streamed:bench_bf16_65536envs_streamed" "bench_c4:bench_config4_shape_8192envs" \
         "bench_h128:bench_hid128" "bench_l3:bench_3_hidden_layers" "bench_l1:bench_1_hidden_layer" "disk_stream:disk_stream_65536envs" \
         "bench_2ranks_shared_gpu:bench_2ranks_shared_gpu" "bench_4ranks_shared_gpu:bench_4ranks_shared_gpu"; do
  [ -f $F/${p%%:*}.json ] && cp $F/${p%%:*}.json profiles/r03_${p#*:}.json
done
ls $F/trace_bf16/*/*kernel_stats.csv >/dev/null 2>&1 && cp $(ls -t $F/trace_bf16/*/*kernel_stats.csv | head -1) profiles/r03_rocprof_kernel_stats_bf16.csv
ls $F/trace_h128/*/*kernel_stats.csv >/dev/null 2>&1 && cp $(ls -t $F/trace_h128/*/*kernel_stats.csv | head -1) profiles/r03_rocprof_kernel_stats_hid128.csv
[ -f $F/trace_bf16_outliers.txt ] && cp $F/trace_bf16_outliers.txt profiles/r03_trace_outliers_bf16.txt
python3 - <<'PY'
import json, os
F = "gpurun_out/final3"
rows = {}
for e in ("256", "512", "1024", "2048", "512_train_tile"):
    p = "%s/shard_%s.json" % (F, e)
    if os.path.exists(p):
        d = json.loads(open(p).read().strip().splitlines()[-1])
        rows[e] = {"ms_per_iteration": d["ms_per_step"], "env_steps_per_s": d["value"], "allreduce": d["allreduce"],
                   "kernels_avg_ms": {k: v["avg_ms"] for k, v in d["kernels"].items() if "avg_ms" in v and "@" not in k}}
if rows:
    json.dump({"what": "rank-0 shard of a strong-scaling run (4096 envs, global minibatch 4096 split N ways) measured on ONE GPU: "
                       "PPO_BENCH_FORCE_DIST=1 bench.py --envs E (one-rank in-library RCCL all-reduce per optimiser step); "
                       "512_train_tile = the same shard with PPO_TRAIN_TILE_MAX_TILES set (k_policy_train_tile + operand-layout "
                       "weight-gradient kernel: measured slower, off by default)",
               "shards": rows}, open("profiles/r03_strong_shards.json", "w"), indent=1)
PY
ls -la profiles | grep r03
