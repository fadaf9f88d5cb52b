#!/bin/bash
# Copy the judged summaries of gpurun_out/final2 (tools/final_profile_r2.sh a / b) into profiles/ (tracked).
set -e
F=gpurun_out/final2
cp $F/bench.json profiles/r02_bench.json
cp $(ls -t $F/trace/*/*kernel_stats.csv | head -1) profiles/r02_rocprof_kernel_stats.csv
cp $F/pmc_summary.txt profiles/r02_pmc_summary.txt
cp $F/pmc_traffic.json profiles/r02_pmc_traffic.json
[ -f $F/bench_bf16.json ] && cp $F/bench_bf16.json profiles/r02_bench_bf16.json
[ -f $F/bench_bf16_c5.json ] && cp $F/bench_bf16_c5.json profiles/r02_bench_bf16_65536envs.json
[ -f $F/bench_bf16_c5_streamed.json ] && cp $F/bench_bf16_c5_streamed.json profiles/r02_bench_bf16_65536envs_streamed.json
[ -f $F/bench_c4.json ] && cp $F/bench_c4.json profiles/r02_bench_config4_shape_8192envs.json
[ -f $F/bench_h128.json ] && cp $F/bench_h128.json profiles/r02_bench_hid128.json
ls $F/trace_bf16/*/*kernel_stats.csv >/dev/null 2>&1 && cp $(ls -t $F/trace_bf16/*/*kernel_stats.csv | head -1) profiles/r02_rocprof_kernel_stats_bf16.csv
ls $F/trace_h128/*/*kernel_stats.csv >/dev/null 2>&1 && cp $(ls -t $F/trace_h128/*/*kernel_stats.csv | head -1) profiles/r02_rocprof_kernel_stats_hid128.csv
[ -f $F/disk_stream.json ] && cp $F/disk_stream.json profiles/r02_disk_stream_65536envs.json
[ -f $F/bench_2ranks_shared_gpu.json ] && cp $F/bench_2ranks_shared_gpu.json profiles/r02_bench_2ranks_shared_gpu.json
[ -f $F/bench_4ranks_shared_gpu.json ] && cp $F/bench_4ranks_shared_gpu.json profiles/r02_bench_4ranks_shared_gpu.json
python3 - <<'PY'
import json, os
F = "gpurun_out/final2"
rows = {}
for e in (512, 1024, 2048):
    p = "%s/shard_%d.json" % (F, e)
    if os.path.exists(p):
        d = json.loads(open(p).read().strip().splitlines()[-1])
        rows[str(e)] = {"ms_per_iteration": d["ms_per_step"], "env_steps_per_s": d["value"], "allreduce": d["allreduce"],
                        "kernels_avg_ms": {k: v["avg_ms"] for k, v in d["kernels"].items() if "avg_ms" in v and "@" not in k}}
if rows:
    json.dump({"what": "rank-0 shard of a strong-scaling run (4096 envs, global minibatch 4096 split N ways) measured on ONE GPU: "
                       "PPO_BENCH_FORCE_DIST=1 bench.py --envs E (one-rank in-library RCCL all-reduce per optimiser step)",
               "shards": rows}, open("profiles/r02_strong_shards.json", "w"), indent=1)
PY
# the non-headline bench lines read roofline.traffic from profiles/r02_pmc_traffic.json AS IT WAS when they ran (part B runs
# before this copy): refresh the field from the PMC passes of this same bundle
python3 - <<'PY'
import json, re
pm = json.load(open("profiles/r02_pmc_traffic.json"))["launch_shapes"]
for n in ("r02_bench_bf16.json", "r02_bench_bf16_65536envs.json", "r02_bench_bf16_65536envs_streamed.json", "r02_bench_config4_shape_8192envs.json", "r02_bench_hid128.json"):
    p = "profiles/" + n
    try: lines = open(p).read().strip().splitlines()
    except FileNotFoundError: continue
    d = json.loads(lines[-1])
    src = d["roofline"].get("traffic_source") or ""
    m = re.search(r"\[([^\]]+)\]", src)
    key = m.group(1) if m else None
    ent = pm.get(key) if key else None
    if ent:
        d["roofline"]["traffic"] = ent["k_policy_bwd_hbm_bytes"]
        lines[-1] = json.dumps(d)
        open(p, "w").write("\n".join(lines) + "\n")
        print("refreshed", n, key, d["roofline"]["traffic"])
PY
ls -la profiles | grep r02
