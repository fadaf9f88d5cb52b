#!/bin/bash
# PMC passes for the dominant kernels (run on the GPU box via gpurun).  Counters are collected in their
# own runs (no tracing domains besides --kernel-trace), FETCH_SIZE and WRITE_SIZE in separate passes
# (TCC slot limits, MI355X_MICROARCH.md "rocprofv3 PMC slots").
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc
mkdir -p $OUT
ARGS="bench.py --steps 1 --warmup 0 --no-cpu-baseline --t-steps 8 --epochs 1"
rocprofv3 --kernel-trace --output-format csv -d $OUT/sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python3 $ARGS > $OUT/sq.log 2>&1 || { tail -5 $OUT/sq.log; exit 1; }
rocprofv3 --kernel-trace --output-format csv -d $OUT/grbm --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -- python3 $ARGS > $OUT/grbm.log 2>&1 || { tail -5 $OUT/grbm.log; exit 1; }
rocprofv3 --kernel-trace --output-format csv -d $OUT/fetch --pmc FETCH_SIZE -- python3 $ARGS > $OUT/fetch.log 2>&1 || { tail -5 $OUT/fetch.log; exit 1; }
rocprofv3 --kernel-trace --output-format csv -d $OUT/write --pmc WRITE_SIZE -- python3 $ARGS > $OUT/write.log 2>&1 || { tail -5 $OUT/write.log; exit 1; }
# bf16 compute mode: HBM traffic of its kernels (same reduced run)
rocprofv3 --kernel-trace --output-format csv -d $OUT/fetch16 --pmc FETCH_SIZE -- python3 $ARGS --dtype bf16 > $OUT/fetch16.log 2>&1 || { tail -5 $OUT/fetch16.log; exit 1; }
rocprofv3 --kernel-trace --output-format csv -d $OUT/write16 --pmc WRITE_SIZE -- python3 $ARGS --dtype bf16 > $OUT/write16.log 2>&1 || { tail -5 $OUT/write16.log; exit 1; }
find $OUT -name "*counter_collection.csv" | head
