#!/bin/bash
# Copy the judged summaries of gpurun_out/final (tools/final_profile.sh) into profiles/ (tracked).
set -e
F=gpurun_out/final
cp $F/bench.json profiles/r01_bench.json
cp $F/bench_bf16.json profiles/r01_bench_bf16.json
cp $F/bench_bf16_c5.json profiles/r01_bench_bf16_65536envs.json
cp $(ls -t $F/trace/*/*kernel_stats.csv | head -1) profiles/r01_rocprof_kernel_stats.csv
cp $(ls -t $F/trace_bf16/*/*kernel_stats.csv | head -1) profiles/r01_rocprof_kernel_stats_bf16.csv
cp $F/pmc_summary.txt profiles/r01_pmc_summary.txt
cp $F/pmc_traffic.json profiles/r01_pmc_traffic.json
cp $F/mfma_f32_valu_overlap.txt profiles/r01_mfma_f32_valu_overlap.txt
ls -la profiles
