#!/bin/bash
# Round-end measurement bundle (GPU box): bench line, rocprofv3 kernel-trace stats of the same command, PMC passes.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/final
timeout -k 10 400 python3 bench.py --steps 3 --warmup 1 > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err || { tail -5 gpurun_out/final/bench.err; exit 1; }
tail -1 gpurun_out/final/bench.json | cut -c1-300
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/final/trace.log 2>&1 || { tail -5 gpurun_out/final/trace.log; exit 1; }
rm -f gpurun_out/final/trace/*/*kernel_trace.csv   # keep the summary, drop the per-dispatch trace (large)
bash tools/profile_pmc.sh > gpurun_out/final/pmc.log 2>&1 || { tail -5 gpurun_out/final/pmc.log; exit 1; }
rm -f gpurun_out/pmc/*/*/*kernel_trace.csv
python3 tools/pmc_summary.py gpurun_out/pmc > gpurun_out/final/pmc_summary.txt
ls gpurun_out/final gpurun_out/final/trace/*
