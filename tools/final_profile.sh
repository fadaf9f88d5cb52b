#!/bin/bash
# Round-end measurement bundle (GPU box): bench line, rocprofv3 kernel-trace stats of the same command, PMC passes,
# the bf16-mode runs and the fp32-MFMA / VALU overlap micro-benchmark.  Copy the summaries into profiles/ afterwards
# (tools/collect_profiles.sh).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
F=gpurun_out/final
rm -rf $F gpurun_out/pmc; mkdir -p $F
# 1. headline bench (fp32, 4096 envs) with the cpu_baseline leg
timeout -k 10 400 python3 bench.py --steps 3 --warmup 1 > $F/bench.json 2> $F/bench.err || { tail -5 $F/bench.err; exit 1; }
tail -1 $F/bench.json | cut -c1-200
# 2. rocprofv3 kernel stats of the same command
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $F/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $F/trace.log 2>&1 || { tail -5 $F/trace.log; exit 1; }
rm -f $F/trace/*/*kernel_trace.csv   # keep the summary, drop the per-dispatch trace (large)
# 3. PMC passes (fp32 kernels; bf16 kernels: HBM traffic passes only)
bash tools/profile_pmc.sh > $F/pmc.log 2>&1 || { tail -5 $F/pmc.log; exit 1; }
rm -f gpurun_out/pmc/*/*/*kernel_trace.csv
python3 tools/pmc_summary.py gpurun_out/pmc > $F/pmc_summary.txt
python3 tools/pmc_traffic.py gpurun_out/pmc $F/pmc_traffic.json > $F/pmc_traffic.log 2>&1 || true
# 4. bf16 compute mode on the same workload and at the config-5 size (not the headline)
timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --dtype bf16 > $F/bench_bf16.json 2> $F/bench_bf16.err || { tail -5 $F/bench_bf16.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $F/trace_bf16 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --dtype bf16 > $F/trace_bf16.log 2>&1 || { tail -5 $F/trace_bf16.log; exit 1; }
rm -f $F/trace_bf16/*/*kernel_trace.csv
timeout -k 10 300 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --dtype bf16 --envs 65536 > $F/bench_bf16_c5.json 2> $F/bench_bf16_c5.err || { tail -5 $F/bench_bf16_c5.err; exit 1; }
# 5. does VALU work hide under the fp32 MFMA?  
for w in 1 2; do [ -x tools/microbench/mfma_valu_w$w ] || hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -DWPS=$w -o tools/microbench/mfma_valu_w$w tools/microbench/mfma_f32_valu_overlap.hip > /dev/null 2>&1; done
( timeout -k 5 60 ./tools/microbench/mfma_valu_w1 && timeout -k 5 60 ./tools/microbench/mfma_valu_w2 ) > $F/mfma_f32_valu_overlap.txt 2>&1 || true
ls $F $F/trace/* $F/trace_bf16/*
